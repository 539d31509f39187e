"""Micro-benchmark (GPU box): mask head forward/backward against the HBM roofline.
Algorithmic bytes per (b,t): 16*K*F + 8*F."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

K, T, F = int(os.environ.get("MH_K", 4)), int(os.environ.get("MH_T", 253)), 513


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for B in [int(a) for a in sys.argv[1:]] or [64, 256]:
    logit = torch.randn(B, K, T, F, device="cuda")
    obs = torch.randn(B, T, F, dtype=torch.complex64, device="cuda")
    dest = torch.randn(B, K, T, F, dtype=torch.complex64, device="cuda")
    mask, est = h.maskhead_fwd(logit, obs)
    nbytes = B * T * (16 * K * F + 8 * F)
    tf = timeit(lambda: h.maskhead_fwd(logit, obs))
    tb = timeit(lambda: h.maskhead_bwd(dest, None, mask, obs))
    src = torch.empty(nbytes // 8, device="cuda"); dst = torch.empty_like(src)
    tc = timeit(lambda: dst.copy_(src))
    print(json.dumps(dict(B=B, MB=round(nbytes / 1e6, 1), fwd_ms=round(tf, 4), fwd_TBps=round(nbytes / tf / 1e9, 3),
                          bwd_ms=round(tb, 4), bwd_TBps=round(nbytes / tb / 1e9, 3),
                          copy_same_bytes_ms=round(tc, 4), copy_TBps=round(nbytes / tc / 1e9, 3))), flush=True)
