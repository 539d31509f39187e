"""Parity of the HIP path against the CPU oracle on a FRESH model (bench.cpu_baseline's parity leg without the training
steps that precede it in bench.py), per-tensor gradient errors listed: `python tools/parity_once.py [batch] [steps] [training batch]`.
`steps` optimizer steps on the bench's synthetic batch first (the bench compares after ~150).  With TSSEP_HIP_LIB a
variant build of the library is compared under the same conditions."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import tssep_amd.hip_ops as H  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    TB = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = torch.device("cuda:0")
    H.GEMM_PRECISION = "bf16x3"
    model = bench.build_model(bench.K_SPK).to(dev)
    from tssep_amd.train.optimizer import Adam
    opt = Adam(gradient_clipping=10.0, lr=1e-5)
    opt.set_parameters(model.parameters())
    if steps:
        obs, aux, tgt = bench.synth_batch(TB, bench.K_SPK, bench.N_SAMPLES, seed=0)
        ex0 = dict(observation=torch.as_tensor(obs).to(dev), auxInput=torch.as_tensor(aux).to(dev),
                   speaker_reverberation_early_ch0=torch.as_tensor(tgt).to(dev), reference_channel=0, dataset=["bench"] * TB)
        np.random.seed(0)
        for _ in range(steps):
            opt.zero_grad()
            ex = dict(ex0)
            model.review(ex, model(ex))["loss"].backward()
            opt.step()
    for prod in (3, 2):
        H.WGRAD_PRODUCTS = prod
        par = bench.cpu_baseline(model, opt, parity_only=True, parity_batch=B, assert_bars=False)
        print(json.dumps(dict(lib=os.environ.get("TSSEP_HIP_LIB", "default"), steps=steps, training_batch=TB, wgrad_products=prod, **par)))
    H.WGRAD_PRODUCTS = 3


if __name__ == "__main__":
    main()
