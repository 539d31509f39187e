"""PCIe-inclusive training throughput (GPU box): the default bench step fed (a) with inputs resident in
HBM (what bench.py times) and (b) with a fresh HOST batch per step through tssep_amd.dataset.DeviceLoader
(pinned staging buffers, asynchronous H2D on a copy stream, 2 batches ahead).  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tssep_amd import dataset as D, hip_ops as H  # noqa: E402
from tssep_amd.train.optimizer import Adam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H.GEMM_PRECISION = "bf16x3"
dev = torch.device("cuda", 0)
model = bench.build_model().to(dev)
opt = Adam(gradient_clipping=10.0, lr=1e-5)
opt.set_parameters(model.parameters())
obs, aux, tgt = bench.synth_batch(B, 4, 64000, seed=0)
host = dict(observation=obs, auxInput=aux, speaker_reverberation_early_ch0=tgt, reference_channel=0,
            dataset=["bench"] * B)
nbytes = obs.nbytes + aux.nbytes + tgt.nbytes


def step(ex):
    opt.zero_grad()
    out = model(dict(ex))
    model.review(ex, out)["loss"].backward()
    opt.step()
    return out


def run(batches, n):
    it = iter(batches)
    for _ in range(3):
        out = step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = step(next(it))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, int(out.mask.shape[-2])


resident = {k: (torch.as_tensor(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in host.items()}
np.random.seed(0)
dt_res, T = run((resident for _ in range(STEPS + 3)), STEPS)
dl = D.DeviceLoader(D.new([host] * (STEPS + 3)), dev, ("observation", "auxInput", "speaker_reverberation_early_ch0"))
np.random.seed(0)
dt_pipe, _ = run(dl, STEPS)
# the copy alone, for scale
pin = {k: torch.as_tensor(v).pin_memory() for k, v in (("o", obs), ("a", aux), ("t", tgt))}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    d = [v.to(dev, non_blocking=True) for v in pin.values()]
torch.cuda.synchronize()
dt_copy = (time.perf_counter() - t0) / 5
print(json.dumps({
    "batch_per_gpu": B, "steps": STEPS, "host_bytes_per_batch": nbytes,
    "resident_ms_per_step": round(dt_res * 1e3, 3), "resident_frames_per_s": round(B * T / dt_res, 1),
    "pcie_inclusive_ms_per_step": round(dt_pipe * 1e3, 3), "pcie_inclusive_frames_per_s": round(B * T / dt_pipe, 1),
    "h2d_alone_ms": round(dt_copy * 1e3, 3), "h2d_GBps": round(nbytes / dt_copy / 1e9, 2),
    "h2d_share_if_serial": round(dt_copy / dt_res, 4),
    "pinned_allocations": dl.stats["pinned_allocations"], "note":
    "same host batch object every step (the copy is real, the data generation is not part of the measurement)"}))
