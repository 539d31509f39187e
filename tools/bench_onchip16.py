"""Forward recurrence: the 32-sequence W-stationary kernel against the interleaved 16-sequence-group kernel (GPU box):
   python tools/bench_onchip16.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

T, Hh = 253, 300
Hp = h.round_up(Hh, 4)
torch.manual_seed(0)
whh = [torch.randn(4 * Hh, Hh, device="cuda") * 0.05 for _ in range(2)]
wf, wb = h.lstm_pack_onchip(whh[0], whh[1], Hh)
wf16 = h.lstm_pack_onchip16(whh[0], whh[1], Hh)
wf16q = h.lstm_pack_onchip16(whh[0], whh[1], Hh, 4)      # four-wave workgroups, ten per cluster, two per CU (round 5)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


for N in [int(a) for a in sys.argv[1:]] or [768, 3072]:
    g0 = torch.randn(N * T, 8 * Hh, device="cuda") * 0.5
    gates = g0.clone()
    cell = torch.empty(N, T, 2, Hh, device="cuda"); hout = torch.zeros(N, T, 2 * Hp, device="cuda")
    row = {"N": N, "bytes_GB": round(N * T * 2 * Hh * (16 + 16 + 4 + 4) / 1e9, 2)}

    def old():
        gates.copy_(g0); h.blstm_onchip_fwd(gates, cell, hout, 2 * Hp, Hp, wf, N, T, Hh)

    def cp():
        gates.copy_(g0)

    t_cp = timeit(cp)
    row["old_ms"] = round(timeit(old) - t_cp, 3)
    ref = hout.clone()
    for g in (1, 2, 4):
        if ((N + 15) // 16) % g:
            continue

        def new():
            gates.copy_(g0); h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, wf16, N, T, Hh, g)

        row[f"g{g}_ms"] = round(timeit(new) - t_cp, 3)
        row[f"g{g}_maxdiff"] = float((hout - ref).abs().max())
    for g in (1, 2):
        if ((N + 15) // 16) % g:
            continue

        def newq():
            gates.copy_(g0); h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, wf16q, N, T, Hh, g, waves=4)

        row[f"w4g{g}_ms"] = round(timeit(newq) - t_cp, 3)
        row[f"w4g{g}_maxdiff"] = float((hout - ref).abs().max())
    h.check_cluster_errors()
    for k in list(row):
        if k.endswith("_ms") and not k.startswith("w4"):
            row[k.replace("_ms", "_us_per_step")] = round(row[k] * 1e3 / T / max(1, -(-N * 2 // (48 * (32 if k == "old_ms" else 16 * int(k[1]))))), 2)
    print(json.dumps(row), flush=True)

# ---- backward: the 32-sequence kernel against the interleaved one (d(gates) in place on saved activations)
wb16 = h.lstm_pack_onchip16_bwd(whh[0], whh[1], Hh)
for N in [int(a) for a in sys.argv[1:]] or [768, 3072]:
    g0 = torch.rand(N * T, 8 * Hh, device="cuda") * 0.8 + 0.1
    gates = g0.clone()
    cell = torch.randn(N, T, 2, Hh, device="cuda") * 0.5
    dh = torch.randn(N, T, 2 * Hp, device="cuda") * 0.1
    row = {"N": N, "direction": "backward"}

    def cp():
        gates.copy_(g0)

    def oldb():
        gates.copy_(g0); h.blstm_onchip_bwd(gates, cell, dh, 2 * Hp, Hp, wb, N, T, Hh)

    t_cp = timeit(cp)
    row["old_ms"] = round(timeit(oldb) - t_cp, 3)
    ref = gates.clone()
    for g in (1, 2, 4):
        if ((N + 15) // 16) % g:
            continue

        def newb():
            gates.copy_(g0); h.blstm_onchip16_bwd(gates, cell, dh, 2 * Hp, Hp, wb16, N, T, Hh, g)

        row[f"g{g}_ms"] = round(timeit(newb) - t_cp, 3)
        row[f"g{g}_maxdiff"] = float((gates - ref).abs().max())
    h.check_cluster_errors()
    print(json.dumps(row), flush=True)
