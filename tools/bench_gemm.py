"""Micro-benchmark (GPU box): the GEMMs of the TS-SEP step, stand-alone.   python tools/bench_gemm.py [batch] [f32|bf16x3]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K_, T = 4, 253
R1, R4 = B * T, B * K_ * T


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def run(name, M, N, K, kind):
    if kind == "nt":
        A = torch.randn(M, h.round_up(K, 4), device="cuda"); W = torch.randn(N, h.round_up(K, 4), device="cuda")
        C = torch.empty(M, N, device="cuda")
        f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K)
    elif kind == "nn":
        A = torch.randn(M, h.round_up(K, 4), device="cuda"); W = torch.randn(K, h.round_up(N, 4), device="cuda")
        C = torch.empty(M, N, device="cuda")
        f = lambda: h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K, b_kmajor=True)
    else:
        P = torch.randn(K, h.round_up(M, 4), device="cuda"); Q = torch.randn(K, h.round_up(N, 4), device="cuda")
        f = lambda: h.wgrad(P, P.shape[1], Q, Q.shape[1], M, N, K)
    ms = timeit(f)
    print(json.dumps(dict(name=name, kind=kind, M=M, N=N, K=K, ms=round(ms, 4),
                          tflops=round(2 * M * N * K / ms / 1e9, 1))), flush=True)


h.GEMM_PRECISION = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
print("precision", h.GEMM_PRECISION)
run("pre_net in", R1, 2400, 553, "nt")
run("birnn0 in", R4, 2400, 513, "nt")
run("birnn1 in", R4, 2400, 320, "nt")
run("birnn2 in", R1, 2400, 1280, "nt")
run("proj 600->320", R4, 320, 600, "nt")
run("proj 600->513", R1, 513, 600, "nt")
run("linear2", R1, 2052, 320, "nt")
run("dgrad birnn0 dx", R4, 513, 2400, "nn")
run("dgrad birnn0 dx (W^T, as on the step)", R4, 513, 2400, "nt")
run("dgrad proj dh", R4, 600, 320, "nn")
run("dgrad birnn2 dx", R1, 1280, 2400, "nn")
run("wgrad W_ih birnn0", 2400, 513, R4, "tn")
run("wgrad W_hh (1 dir)", 1200, 300, R4, "tn")
run("wgrad proj", 320, 600, R4, "tn")
run("wgrad linear2", 2052, 320, R1, "tn")
run("wgrad W_ih birnn2", 2400, 1280, R1, "tn")
