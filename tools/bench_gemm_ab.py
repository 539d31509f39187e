"""Alternating A/B of a GEMM environment switch on the row x row shapes of the step (GPU box):
   python tools/bench_gemm_ab.py TSSEP_GEMM_STREAM 1 0 [batch]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
# round 4: the production library reads no environment variable; the TSSEP_GEMM_* switches live in the experiment
# build (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so), where they are read per call
from tssep_amd import _lib
assert not var.startswith(("TSSEP_GEMM_", "TSSEP_BIG_", "TSSEP_STREAM_")) or "exp" in os.path.basename(_lib.LIB_PATH), \
    f"{var} is a switch of the experiment build: set TSSEP_HIP_LIB=.../libtssep_hip_exp.so"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 768
KIND = sys.argv[5] if len(sys.argv) > 5 else "nt"          # "nt": forward / d(input) shapes, "tn": weight gradients
h.GEMM_PRECISION = "bf16x3"
T, Kspk = 253, 4
R1, R4 = B * T, B * Kspk * T
SHAPES = [("pre_net in", R1, 2400, 553), ("birnn0 in", R4, 2400, 513), ("birnn1 in", R4, 2400, 320),
          ("birnn2 in", R1, 2400, 1280), ("proj 600->320", R4, 320, 600), ("proj 600->513", R1, 513, 600),
          ("linear2", R1, 2052, 320), ("dgrad birnn0 dx", R4, 513, 2400), ("dgrad birnn1 dx", R4, 320, 2400),
          ("dgrad proj dh", R4, 600, 320), ("dgrad birnn2 dx", R1, 1280, 2400), ("dgrad linear2", R1, 320, 2052)]


TN_SHIFT = [("wgrad W_hh birnn0/1", 1200, 300, R4, 253), ("wgrad W_hh birnn2", 1200, 300, R1, 253)]
TN_SHAPES = [("wgrad W_ih pre_net", 2400, 553, R1, True), ("wgrad W_ih birnn0", 2400, 513, R4, True),
             ("wgrad W_ih birnn1", 2400, 320, R4, True), ("wgrad W_ih birnn2", 2400, 1280, R1, True),
             ("wgrad proj 320", 320, 600, R4, True), ("wgrad proj 513", 513, 600, R1, True),
             ("wgrad linear2", 2052, 320, R1, True)]


def timeit(fn, reps=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


tot = {va: 0.0, vb: 0.0}
os.environ.setdefault(var, va)
for shape in (SHAPES if KIND == "nt" else TN_SHIFT if KIND == "shift" else TN_SHAPES):
    if KIND == "shift":
        name, M, N, K, T = shape
        A = torch.randn(K, M, device="cuda"); W = torch.randn(K, N, device="cuda")
        C = None
        f = lambda: h.wgrad(A, M, W, N, M, N, K, b_kshift=-1, kperiod=T)
    elif KIND == "nt":
        name, M, N, K = shape
        A = torch.randn(M, h.round_up(K, 4), device="cuda"); W = torch.randn(N, h.round_up(K, 4), device="cuda")
        C = torch.empty(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
        f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K, bias=bias)
    else:
        name, M, N, K, colsum = shape
        A = torch.randn(K, h.round_up(M, 4), device="cuda"); W = torch.randn(K, h.round_up(N, 4), device="cuda")
        C = None
        f = lambda: h.wgrad(A, A.shape[1], W, W.shape[1], M, N, K, with_colsum=colsum)
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
