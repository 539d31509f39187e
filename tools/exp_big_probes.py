# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Experiment (GPU box, TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so): where does a stage of the big-tile GEMM go?
TIMING probes, garbage results (csrc/gemm_bf16x3_big.hip, PROBE)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

h.GEMM_PRECISION = "bf16x3"
B, T, Kspk = 768, 253, 4
SHAPES = [("dgrad birnn2", B * T, 1280, 2400)]
PROBES = [(0, "full"), (32, "truncating split (no v_cvt_pk_bf16_f32)"), (34, "truncating split, no global loads"), (1, "no barriers"), (2, "no global loads"), (4, "no staging"), (6, "no loads, no staging"),
          (8, "no epilogue"), (14, "MFMA + fragment reads + barriers only"), (15, "MFMA + fragment reads only"),
          (16, "no MFMA"), (30, "fragment reads + barriers only")]


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
    for pk, label in PROBES:
        os.environ["TSSEP_BIG_PROBE"] = str(pk)
        row[f"probe{pk}_ms"] = round(timeit(f), 3)
        row[f"probe{pk}"] = label
    os.environ["TSSEP_BIG_PROBE"] = "0"
    row["stages_per_cu"] = round(-(-M // 256) * -(-N // 256) / 256 * -(-K // 32), 1)
    print(json.dumps(row), flush=True)
    del A, W, C
