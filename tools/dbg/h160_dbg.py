import os, sys
sys.path.insert(0, "/root/repo")
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
torch.manual_seed(17)
for (R, M, N, S, colsum) in [(2500, 300, 130, 3, True), (2500, 300, 600, 3, True), (2500, 320, 130, 3, True), (2500, 300, 130, 1, True),
                             (2512, 300, 130, 1, True), (2500, 300, 132, 1, False), (2500, 316, 600, 1, False)]:
    dz = torch.randn(R, h.round_up(M, 4), device="cuda")
    x = torch.randn(R, h.round_up(N, 4), device="cuda") / R ** 0.5
    outs = {}
    for mode in ("1", "0"):
        os.environ["TSSEP_GEMM_TN_H160"] = mode
        outs[mode] = h.wgrad(dz, dz.shape[1], x, x.shape[1], M, N, R, with_colsum=colsum, splitk=S)[0].clone()
    Nc = N + 1 if colsum else N
    ldp = outs["1"].numel() // (S * M)
    a, b = outs["1"].view(S, M, ldp)[:, :, :Nc], outs["0"].view(S, M, ldp)[:, :, :Nc]
    d = (a != b)
    print((R, M, N, S, colsum), "diff", int(d.sum()))
    if d.any():
        idx = d.nonzero()
        print("  splits", idx[:, 0].unique().tolist(), "rows", idx[:, 1].min().item(), idx[:, 1].max().item(), "nrows", idx[:, 1].unique().numel(),
              "cols", idx[:, 2].min().item(), idx[:, 2].max().item(), "ncols", idx[:, 2].unique().numel())
        ref = (dz[:, :M].double().t() @ x[:, :N].double())
        print("  err h160", (a.double().sum(0)[:, :N] - ref).abs().max().item(), "err t128", (b.double().sum(0)[:, :N] - ref).abs().max().item())
        print("  rows list", idx[:, 1].unique().tolist()[:40])
