# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Sweep (GPU box): the 256 x 128 weight-gradient tile (TSSEP_GEMM_TN_TALL=1) against the 128 x 128 one (=2, default
for the unshifted shapes) over split-K counts, on the dW_ih shapes of the step."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

h.GEMM_PRECISION = "bf16x3"
B, T, Kspk = 768, 253, 4
R1, R4 = B * T, B * Kspk * T
SHAPES = [("W_ih pre_net", 2400, 553, R1), ("W_ih birnn0", 2400, 513, R4), ("W_ih birnn1", 2400, 320, R4),
          ("W_ih birnn2", 2400, 1280, R1), ("linear2", 2052, 320, R1)]


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(2):
        s.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best


for name, M, N, K in SHAPES:
    A = torch.randn(K, h.round_up(M, 4), device="cuda"); X = torch.randn(K, h.round_up(N, 4), device="cuda")
    row = dict(name=name, M=M, N=N, K=K)
    for tall in ("2", "1"):
        os.environ["TSSEP_GEMM_TN_TALL"] = tall
        for S in (4, 5, 6, 8, 10, 12, 16, 20, 24):
            row[f"tall{tall}_S{S}"] = round(timeit(lambda: h.wgrad(A, A.shape[1], X, X.shape[1], M, N, K, with_colsum=True, splitk=S)), 3)
    print(json.dumps(row), flush=True)
    del A, X
