"""Micro-benchmark (GPU box): the fused tail kernels -- mask head + inverse STFT (forward) and iSTFT adjoint
+ mask-head backward -- against the unfused chain, at the cfg3 / cfg5 shapes.  HIP events, inputs resident."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import functional as Fn, hip_ops as h


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for name, B, K, N in (("cfg3 b768", 768, 4, 64000), ("cfg5 b48", 48, 8, 480000)):
    T = h.stft_frames(N)
    F = 513
    _, wsyn = Fn.windows("hann", 1024, 256, torch.device("cuda", 0))
    logit = torch.randn(B, K, T, F, device="cuda")
    obs = torch.randn(B, T, F, device="cuda", dtype=torch.complex64)
    tgt = torch.randn(B, K, N, device="cuda")
    dy = torch.randn(B, K, N, device="cuda")
    res = dict(name=name, B=B, K=K, T=T)
    res["fused_fwd_ms"] = round(timeit(lambda: h.mask_istft_fwd(logit, obs, wsyn, N, tgt=tgt)), 4)
    res["fused_bwd_ms"] = round(timeit(lambda: h.mask_istft_bwd(dy, logit, obs, wsyn)), 4)
    mask, est = h.maskhead_fwd(logit, obs)
    res["maskhead_fwd_ms"] = round(timeit(lambda: h.maskhead_fwd(logit, obs)), 4)
    res["istft_fwd_ms"] = round(timeit(lambda: h.istft_fwd(est.view(B * K, T, F), wsyn, N, tgt=tgt.view(B * K, N))), 4)
    dX = h.istft_bwd(dy.view(B * K, N), wsyn, T)
    res["istft_bwd_ms"] = round(timeit(lambda: h.istft_bwd(dy.view(B * K, N), wsyn, T)), 4)
    res["maskhead_bwd_ms"] = round(timeit(lambda: h.maskhead_bwd(dX.view(B, K, T, F), None, mask, obs)), 4)
    mh = B * T * (16 * K * F + 8 * F)                       # mask head alone, per direction
    chain = mh + B * T * (8 * K * F) + 4 * B * K * N        # + the (i)STFT side: estimate + samples
    for d in ("fwd", "bwd"):
        res[f"fused_{d}_maskhead_bytes_GBps"] = round(mh / res[f"fused_{d}_ms"] / 1e6, 1)
        res[f"fused_{d}_chain_bytes_GBps"] = round(chain / res[f"fused_{d}_ms"] / 1e6, 1)
    print(json.dumps(res), flush=True)
    del logit, obs, tgt, dy, mask, est, dX
