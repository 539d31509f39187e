"""debug: where / what are the stream kernel's wrong elements"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
M, N, K = 24288, 2048, 512
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
bias = torch.randn(N, device="cuda")
with h.prefer_gemm_kernels("tall2"):
    R = torch.zeros(M, N, device="cuda"); h.gemm(A, K, W, K, R, N, M, N, K, bias=bias)
for trial in range(6):
    C = torch.full((M, N), 777.0, device="cuda")
    with h.prefer_gemm_kernels("stream"):
        h.gemm(A, K, W, K, C, N, M, N, K, bias=bias)
    bad = (C != R).nonzero()
    print("trial", trial, "bad", bad.shape[0])
    seen = set()
    for r, c in bad.tolist()[:64]:
        v = C[r, c].item()
        # where does this value come from?
        src = (R == v).nonzero()
        s = src.tolist()[:2] if src.numel() else None
        print(f"   ({r},{c}) tile ({r // 256},{c // 128}) in-tile ({r % 256},{c % 128}) got {v:.5f} want {R[r, c].item():.5f} got==elsewhere {s}")
