import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
def timeit(fn, reps=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for (M, N, K) in [(2400, 513, 777216), (2400, 1280, 194304), (2400, 553, 194304), (2400, 320, 777216)]:
    A = torch.randn(K, M, device="cuda"); W = torch.randn(K, h.round_up(N, 4), device="cuda")
    for xc in ("1", "0"):
        os.environ["TSSEP_GEMM_TN_XC"] = xc
        row = {}
        for S in (8, 16, 24, 32, 40, 48):
            f = lambda: h.wgrad(A, M, W, W.shape[1], M, N, K, with_colsum=True, splitk=S)
            row[S] = round(min(timeit(f) for _ in range(3)), 3)
        print(json.dumps(dict(M=M, N=N, K=K, xc=xc, ms=row)), flush=True)
    del A, W
