# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
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
for (M, N, K) in [(320, 600, 777216), (320, 600, 194304), (513, 600, 194304), (2052, 320, 194304)]:
    A = torch.randn(K, h.round_up(M, 4), device="cuda"); W = torch.randn(K, h.round_up(N, 4), device="cuda")
    row = {}
    for S in (16, 24, 32, 40, 48, 56, 64):
        f = lambda: h.wgrad(A, A.shape[1], W, W.shape[1], M, N, K, with_colsum=True, splitk=S)
        row[S] = round(min(timeit(f) for _ in range(3)), 3)
    print(json.dumps(dict(M=M, N=N, K=K, default_S=h.pick_splitk(M, N + 1, K), ms=row)), flush=True)
    del A, W
