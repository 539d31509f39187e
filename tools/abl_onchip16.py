"""Ablation timing of the interleaved forward recurrence (GPU box): run with TSSEP_HIP_LIB pointing at a library whose
lstm_onchip.hip was compiled with -DONCHIP16_ABL=<bits> (results wrong by construction; see the macro).
   TSSEP_HIP_LIB=tssep_amd/libtssep_hip_abl1.so python tools/abl_onchip16.py [N ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

T, Hh = 253, 300
Hp = h.round_up(Hh, 4)
torch.manual_seed(0)
whh = [torch.randn(4 * Hh, Hh, device="cuda") * 0.05 for _ in range(2)]
packs = {8: h.lstm_pack_onchip16(whh[0], whh[1], Hh, 8), 4: h.lstm_pack_onchip16(whh[0], whh[1], Hh, 4)}


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    return best


for N in [int(a) for a in sys.argv[1:]] or [32, 3072]:
    gates = torch.randn(N * T, 8 * Hh, device="cuda") * 0.5
    cell = torch.empty(N, T, 2, Hh, device="cuda"); hout = torch.zeros(N, T, 2 * Hp, device="cuda")
    row = {"lib": os.path.basename(os.environ.get("TSSEP_HIP_LIB", "default")), "N": N}
    for waves in (8, 4):
        for g in (1, 2):
            if ((N + 15) // 16) % g:
                continue
            row[f"w{waves}g{g}_ms"] = round(timeit(lambda: h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, packs[waves], N, T, Hh, g, waves=waves)), 3)
    # the same launches with the rows in TIME-MAJOR order inside groups of 32 sequences (kernel argument `layout` = 1: the 16
    # sequences a workgroup touches at one step are 16 consecutive rows instead of rows T apart) -- the access pattern only
    if N % 32 == 0:
        for g in (1, 2):
            row[f"w8g{g}_timemajor_ms"] = round(timeit(lambda: h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, packs[8], N, T, Hh, g, layout=1)), 3)
    h.cluster_error_code()
    print(json.dumps(row), flush=True)
