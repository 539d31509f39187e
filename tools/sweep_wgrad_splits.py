# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Split-count sweep of the weight-gradient GEMMs under an environment switch (GPU box):
   python tools/sweep_wgrad_splits.py [VAR a b]     (default: TSSEP_GEMM_TN_XC 1 0)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
var, vals = (sys.argv[1], sys.argv[2:4]) if len(sys.argv) > 3 else ("TSSEP_GEMM_TN_XC", ["1", "0"])
SHIFTED = var == "TSSEP_GEMM_TN_W160"
SHAPES = [(1200, 300, 777216), (1200, 300, 194304)] if SHIFTED else \
    [(2400, 513, 777216), (2400, 1280, 194304), (2400, 553, 194304), (2400, 320, 777216)]
SPLITS = (16, 24, 32, 40, 48, 56, 64, 96) if SHIFTED else (8, 16, 24, 32, 40, 48)


def timeit(fn, reps=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for (M, N, K) in SHAPES:
    A = torch.randn(K, M, device="cuda"); W = torch.randn(K, h.round_up(N, 4), device="cuda")
    for v in vals:
        os.environ[var] = v
        row = {}
        for S in SPLITS:
            if SHIFTED:
                f = lambda: h.wgrad(A, M, W, W.shape[1], M, N, K, b_kshift=-1, kperiod=253, splitk=S)
            else:
                f = lambda: h.wgrad(A, M, W, W.shape[1], M, N, K, with_colsum=True, splitk=S)
            row[S] = round(min(timeit(f) for _ in range(3)), 3)
        print(json.dumps({"M": M, "N": N, "K": K, var: v, "ms": row}), flush=True)
    del A, W
