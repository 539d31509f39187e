# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Experiment (GPU box, TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so): where does the life of a 256 x 256 tile of the
row x row split-bf16 GEMM go?  TIMING probes -- the hacked variants compute garbage (csrc/gemm_bf16x3.hip, HACK)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

h.GEMM_PRECISION = "bf16x3"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
T, Kspk = 253, 4
SHAPES = [("birnn1 in", B * Kspk * T, 2400, 320), ("birnn0 in", B * Kspk * T, 2400, 513)]
os.environ["TSSEP_GEMM_STREAM"] = "0"
HACKS = [(0, "production"), (32, "epilogue through LDS, no global stores"), (64, "temporal stores"),
         (2, "no C stores")]


def timeit(fn, reps=6):
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
    A = torch.randn(M, h.round_up(K, 4), device="cuda"); W = torch.randn(N, h.round_up(K, 4), device="cuda")
    C = torch.empty(M, N, device="cuda")
    f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K)
    row = dict(name=name, M=M, N=N, K=K)
    for hk, label in HACKS:
        os.environ["TSSEP_GEMM_HACK"] = str(hk)
        ms = timeit(f)
        row[f"hack{hk}_ms"] = round(ms, 3)
        row[f"hack{hk}"] = label
    os.environ["TSSEP_GEMM_HACK"] = "0"
    print(json.dumps(row), flush=True)
    del A, W, C
