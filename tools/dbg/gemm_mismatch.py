"""debug: which NT kernel disagrees on (24288, 2048, 513) and why"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
for (M, N, K, pads) in [(24288, 2048, 513, "garbage"), (24288, 2048, 513, "zero"), (24288, 2048, 512, "none"), (2048, 2048, 513, "garbage"), (24288, 2400, 513, "garbage")]:
    torch.manual_seed(0)
    lda = h.round_up(K, 4)
    A = torch.randn(M, lda, device="cuda"); W = torch.randn(N, lda, device="cuda") / K ** 0.5
    if pads == "zero":
        A[:, K:] = 0; W[:, K:] = 0
    bias = torch.randn(N, device="cuda")
    ref = (A[:, :K].double() @ W[:, :K].double().t() + bias.double()).float()
    outs = {}
    for k in ("pipe", "tall2", "tall4", "big", "stream"):
        C = torch.zeros(M, N, device="cuda")
        with h.prefer_gemm_kernels(k):
            log = h.GEMM_LOG = []
            h.gemm(A, lda, W, lda, C, N, M, N, K, bias=bias)
            h.GEMM_LOG = None
        outs[k] = C
        err = (C - ref).abs().max().item()
        print(M, N, K, pads, k, "ran", log[0][0], "max err vs fp64", f"{err:.3e}", "equal to tall2" if k == "tall2" else bool(torch.equal(C, outs["tall2"])) if "tall2" in outs else "")
    d = (outs["big"] != outs["stream"])
    if d.any():
        idx = d.nonzero()
        print("  big != stream at", idx.shape[0], "elements; rows", idx[:, 0].min().item(), "..", idx[:, 0].max().item(), "cols", idx[:, 1].min().item(), "..", idx[:, 1].max().item(),
              "max diff", (outs["big"] - outs["stream"]).abs().max().item())
