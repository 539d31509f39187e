"""The graphed 8-utterance step trained for N steps with the derived weight layouts built inline (0) and on their branch
of the graph (1): same seeds, same kernels, same arithmetic -> the parameters must agree BIT FOR BIT afterwards (a missing
dependency between the branch and a consumer would show as a difference or as a recurrence error flag).
`python tools/check_prepare_derived.py [steps] [batch]`"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tssep_amd.hip_ops as H  # noqa: E402


def run(prepare, steps, B):
    from tssep_amd.train.graph import GraphedStep
    from tssep_amd.train.optimizer import Adam
    H.PREPARE_DERIVED = bool(prepare)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H.GEMM_PRECISION = "bf16x3"
    model = bench.build_model(bench.K_SPK).to(dev)
    opt = Adam(gradient_clipping=10.0, lr=1e-4)
    opt.set_parameters(model.parameters())
    obs, aux, tgt = bench.synth_batch(B, bench.K_SPK, bench.N_SAMPLES, seed=0)
    ex0 = dict(observation=torch.as_tensor(obs).to(dev), auxInput=torch.as_tensor(aux).to(dev),
               speaker_reverberation_early_ch0=torch.as_tensor(tgt).to(dev), reference_channel=0, dataset=["bench"] * B)
    np.random.seed(0)
    g = GraphedStep(model, opt, adopt_inputs=True)
    hits0 = H.PREPARED_HITS
    losses = []
    for i in range(steps):
        _, summary = g(dict(ex0))
        opt.step()
        if i % 50 == 0 or i == steps - 1:
            losses.append(float(summary["loss"]))
    torch.cuda.synchronize()
    H.check_cluster_errors(dev)
    return opt.flat_param.clone(), losses, H.PREPARED_HITS - hits0


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    p0, l0, h0 = run(0, steps, B)
    p1, l1, h1 = run(1, steps, B)
    print(json.dumps(dict(steps=steps, batch=B, bit_identical_parameters=bool(torch.equal(p0, p1)),
                          max_abs_diff=float((p0 - p1).abs().max()), losses_inline=l0, losses_branch=l1,
                          prepared_hits=[h0, h1], finite=bool(torch.isfinite(p1).all()))))


if __name__ == "__main__":
    main()
