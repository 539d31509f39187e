import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
B, T, K, F, P = 768, 253, 4, 513, 320
M = B * T
A = torch.randn(M, P, device="cuda")
for N in (2052, 2048, 2049, 2176, 2304):
    W = torch.randn(N, P, device="cuda"); bias = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    row = {"N": N}
    for name, kw in (("plain", {}), ("bias", dict(bias=bias))):
        t = timeit(lambda: h.gemm(A, P, W, P, C, N, M, N, P, **kw))
        row[name + "_ms"] = round(t, 3); row[name + "_tflops"] = round(2 * M * N * P / t / 1e9, 1)
    if N == 2052:
        perm = torch.stack([torch.randperm(K) for _ in range(B)]).int().cuda()
        out = torch.empty(B, K, T, F, device="cuda")
        rm = dict(T=T, K=1, sb=K * T * F, sk=0, st=F, cm=F, co=T * F, perm=perm, perm_ld=K)
        for mode in ("1", "0"):
            os.environ["TSSEP_GEMM_REMAP_WIDE"] = mode
            t = timeit(lambda: h.gemm(A, P, W, P, out, 0, M, N, P, bias=bias, remap=rm))
            row["remap_wide" + mode + "_ms"] = round(t, 3)
    print(json.dumps(row), flush=True)
