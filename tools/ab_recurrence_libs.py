"""Forward + backward interleaved recurrence launches (GPU box), one library per process: alternate processes on one box for
an A/B of two builds (TSSEP_HIP_LIB).   python tools/ab_recurrence_libs.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

T, Hh = 253, 300
Hp = h.round_up(Hh, 4)
torch.manual_seed(0)
whh = [torch.randn(4 * Hh, Hh, device="cuda") * 0.05 for _ in range(2)]
wf16 = h.lstm_pack_onchip16(whh[0], whh[1], Hh)
wb16 = h.lstm_pack_onchip16_bwd(whh[0], whh[1], Hh)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


for N in [int(a) for a in sys.argv[1:]] or [768, 3072]:
    dev = torch.device("cuda", 0)
    gf, gb = h.onchip16_groups(N, Hh, dev), h.onchip16_bwd_groups(N, Hh, dev)
    gates = torch.rand(N * T, 8 * Hh, device="cuda") * 0.8 + 0.1
    cell = torch.randn(N, T, 2, Hh, device="cuda") * 0.5
    hout = torch.zeros(N, T, 2 * Hp, device="cuda")
    dh = torch.randn(N, T, 2 * Hp, device="cuda") * 0.1
    row = {"lib": os.path.basename(os.environ.get("TSSEP_HIP_LIB", "default")), "N": N, "groups": [gf, gb]}
    row["fwd_ms"] = round(timeit(lambda: h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, wf16, N, T, Hh, gf)), 3)
    row["bwd_ms"] = round(timeit(lambda: h.blstm_onchip16_bwd(gates, cell, dh, 2 * Hp, Hp, wb16, N, T, Hh, gb)), 3)
    h.cluster_error_code()      # (ablation builds compute garbage by construction)
    print(json.dumps(row), flush=True)
