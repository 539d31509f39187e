"""One GEMM shape, few launches (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = os.environ.get("TSSEP_GEMM_PRECISION", "f32")
M, N, K = [int(a) for a in sys.argv[1:4]]
A = torch.randn(M, h.round_up(K, 4), device="cuda"); W = torch.randn(N, h.round_up(K, 4), device="cuda")
C = torch.empty(M, N, device="cuda")
for _ in range(3):
    h.gemm(A, A.shape[1], W, W.shape[1], C, N, M, N, K)
torch.cuda.synchronize()
