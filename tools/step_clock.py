"""Shader clock DURING the training step (GPU box): a one-block sleeping probe on a side stream reads the
shader tick counter against the constant 100 MHz counter while the default bench step runs on the main
stream.  Prints the clock per 45-ms window next to the idle and the all-MFMA figures."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tssep_amd import _lib, hip_ops as H  # noqa: E402
from tssep_amd.train.optimizer import Adam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
H.GEMM_PRECISION = "bf16x3"
dev = torch.device("cuda", 0)
L = _lib.lib()


def probe(stream, n, iters=100000):
    outs = [torch.zeros(2, device=dev, dtype=torch.int64) for _ in range(n)]
    with torch.cuda.stream(stream):
        for o in outs:
            H.check(L.tssep_probe_clock(o.data_ptr(), 1, iters, 0, stream.cuda_stream), "probe")
    return outs


def mhz(outs):
    torch.cuda.synchronize()
    return [round(float(o[0]) / float(o[1]) * 100.0) for o in (x.cpu() for x in outs)]


side = torch.cuda.Stream()
idle = mhz(probe(side, 3))
heavy = H.probe_clock(heavy=True)
model = bench.build_model().to(dev)
opt = Adam(gradient_clipping=10.0, lr=1e-5)
opt.set_parameters(model.parameters())
obs, aux, tgt = bench.synth_batch(B, 4, 64000, seed=0)
ex0 = dict(observation=torch.as_tensor(obs).to(dev), auxInput=torch.as_tensor(aux).to(dev),
           speaker_reverberation_early_ch0=torch.as_tensor(tgt).to(dev), reference_channel=0,
           dataset=["bench"] * B)


def step():
    opt.zero_grad()
    out = model(dict(ex0))
    model.review(ex0, out)["loss"].backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
outs = probe(side, 40)                 # ~1.8 s of probes, queued now, running beside the steps
for _ in range(14):
    step()
torch.cuda.synchronize()
m = mhz(outs)
print(json.dumps({"batch": B, "idle_mhz": idle, "all_mfma_probe_mhz": round(heavy),
                  "during_step_mhz": m, "during_step_mean": round(float(np.mean(m))),
                  "during_step_min": min(m), "during_step_max": max(m)}))
