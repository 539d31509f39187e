# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Alternating A/B of a GEMM switch on the row x row shapes of the step whose store carries a Tanh (act 1) or the folded
Tanh backward (act 2) -- the ones the streaming / big-tile kernels do not take (GPU box):
   python tools/bench_gemm_act.py TSSEP_GEMM_NT_W160 1 0 [batch]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 768
h.GEMM_PRECISION = "bf16x3"
T, Kspk = 253, 4
R1, R4 = B * T, B * Kspk * T
SHAPES = [("proj 600->320 + tanh", R4, 320, 600, 1, False), ("proj 600->320 + tanh, combining store", R4, 320, 600, 1, True),
          ("dgrad birnn1 dx (1 - y^2)", R4, 320, 2400, 2, False), ("linear2-like plain N=320", R1, 320, 2052, 0, False)]


def timeit(fn, reps=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


tot = {va: 0.0, vb: 0.0}
os.environ.setdefault(var, va)
for name, M, N, K, act, remap in SHAPES:
    A = torch.randn(M, h.round_up(K, 4), device="cuda"); W = torch.randn(N, h.round_up(K, 4), device="cuda")
    bias = torch.randn(N, device="cuda")
    Y = torch.tanh(torch.randn(M, N, device="cuda")) if act == 2 else None
    if remap:
        C = torch.empty(M // Kspk, Kspk * N, device="cuda")
        Tq = M // Kspk // B
        f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, 0, M, N, K, bias=bias, act=act,
                           remap=dict(T=Tq, K=Kspk, sb=Tq * Kspk * N, sk=N, st=Kspk * N))
    elif act == 2:
        C = torch.empty(M, N, device="cuda")
        f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K, act=2, aux=(Y, N))
    else:
        C = torch.empty(M, N, device="cuda")
        f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K, bias=bias, act=act)
    best = {va: 1e9, vb: 1e9}
    f(); torch.cuda.synchronize()
    for _ in range(3):
        for v in (va, vb):
            os.environ[var] = v
            f(); best[v] = min(best[v], timeit(f))
    for v in (va, vb):
        tot[v] += best[v]
    print(json.dumps(dict(name=name, M=M, N=N, K=K, **{f"{var}={v}_ms": round(best[v], 3) for v in (va, vb)},
                          **{f"{var}={v}_tflops": round(2 * M * N * K / best[v] / 1e9, 1) for v in (va, vb)})), flush=True)
    del A, W, C
print(json.dumps({f"total_{var}={v}_ms": round(t, 3) for v, t in tot.items()}))
