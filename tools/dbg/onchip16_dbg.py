import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tssep_amd import hip_ops as h
torch.manual_seed(0)
N, T, Hh, groups = [int(a) for a in sys.argv[1:5]]
I = 12
G4 = 4 * Hh
wih = torch.randn(2, G4, I) * 0.3; whh = torch.randn(2, G4, Hh) * 0.1; b = torch.randn(2, 2, G4) * 0.1
plist = [wih[0], whh[0], b[0, 0], b[0, 1], wih[1], whh[1], b[1, 0], b[1, 1]]
pk = h.lstm_pack([t.cuda() for t in plist], Hh, I)
wf16 = h.lstm_pack_onchip16(whh[0].cuda(), whh[1].cuda(), Hh)
x = torch.randn(N, T, I)
ld_x = h.round_up(I, 4)
xd = torch.zeros(N * T, ld_x, device="cuda"); xd[:, :I] = x.reshape(N * T, I).cuda()
gates = torch.empty(N * T, 8 * Hh, device="cuda")
h.gemm(xd, ld_x, pk["wih_p"], pk["ld_i"], gates, 8 * Hh, N * T, 8 * Hh, I, bias=pk["bias_p"])
g2 = gates.clone()
Hp = h.round_up(Hh, 4)
cell = torch.full((N, T, 2, Hh), float("nan"), device="cuda"); hout = torch.zeros(N, T, 2 * Hp, device="cuda")
h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, wf16, N, T, Hh, groups)
torch.cuda.synchronize()
print("err flags", h._err_flag(gates.device).tolist() if hasattr(h, "_err_flag") else None)
cell2 = torch.empty_like(cell); hout2 = torch.zeros_like(hout)
h.blstm_fwd(g2, cell2, hout2, 2 * Hp, Hp, pk["whh_f"], N, T, Hh)
for name, a, bb in (("h", hout.view(N, T, 2, Hp)[..., :Hh], hout2.view(N, T, 2, Hp)[..., :Hh]), ("cell", cell, cell2),
                    ("gates", gates.view(N, T, 2, Hh, 4), g2.view(N, T, 2, Hh, 4))):
    e = (a - bb).abs()
    e = torch.nan_to_num(e, nan=9.0)
    bad = e > 1e-4
    print(name, "max err", float(e.max()), "bad", int(bad.sum()), "of", bad.numel())
    if bad.any():
        idx = bad.nonzero()
        for d in range(idx.shape[1]):
            vals = idx[:, d].unique()
            print("   dim", d, "bad values:", vals[:24].tolist(), "..." if len(vals) > 24 else "", "count", len(vals))
