"""Phase timeline of the interleaved forward recurrence from a trace build (GPU box):
   (cd tssep_amd/csrc && hipcc ... -DONCHIP16_TRACE=1 -c lstm_onchip.hip ...; link as libtssep_hip_trace.so)
   TSSEP_HIP_LIB=$PWD/tssep_amd/libtssep_hip_trace.so python tools/trace_onchip16.py [N] [waves] [groups]
Stamps (s_memtime ticks, 100 MHz constant clock on gfx950 -> 10 ns) of steps 128 .. 131 of work item 0, workgroup 0 of its
cluster: 0 phase start | 1 gather decoded (exchange) / tile landed (io) | 2 first barrier passed | 3 MFMAs done | 4 cell update
done | 5 second barrier passed | 6 publish + own products (exchange) / flush issued (io) | 7 phase end."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 4
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 1
T, Hh = 253, 300
Hp = h.round_up(Hh, 4)
torch.manual_seed(0)
whh = [torch.randn(4 * Hh, Hh, device="cuda") * 0.05 for _ in range(2)]
pack = h.lstm_pack_onchip16(whh[0], whh[1], Hh, waves)
gates = torch.randn(N * T, 8 * Hh, device="cuda") * 0.5
cell = torch.empty(N, T, 2, Hh, device="cuda"); hout = torch.zeros(N, T, 2 * Hp, device="cuda")
for rep in range(3):
    h.KEEP_XBUF = []
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    h.blstm_onchip16_fwd(gates, cell, hout, 2 * Hp, Hp, pack, N, T, Hh, groups, waves=waves)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e)
    xb = h.KEEP_XBUF[0]
    tb = xb[48:48 + 76].cpu().numpy().astype("int64")      # words 96 .. : 64-bit stamps
    st = tb[:64].reshape(2, 4, 8)
    t0 = st[0, 0, 0]
    print(json.dumps(dict(N=N, waves=waves, groups=groups, ms=round(ms, 3), us_per_step=round(ms * 1e3 / T, 3),
                          exchange_wave=[[int(v - t0) for v in row] for row in st[0]],
                          io_wave=[[int(v - t0) for v in row] for row in st[1]], repolls=[int(v) for v in tb[64:68]], arrived=[int(v - t0) for v in tb[68:72]])))
h.cluster_error_code()

# ---- backward (group 0 of the bundle that serves work item 0, steps 128 and 129): stamps 0 phase start | 1 partial sums
# gathered + summed (exchange) / tiles landed (io) | 2 first barrier | 3 cell backward done | 4 second barrier | 5 deferred flush /
# request issued, MFMAs start | 6 MFMAs done | 7 partial sums published from the accumulators | 8 third barrier | 9 phase end
if len(sys.argv) > 4 and sys.argv[4] == "bwd" and waves == 8:
    wb16 = h.lstm_pack_onchip16_bwd(whh[0], whh[1], Hh)
    gates = torch.rand(N * T, 8 * Hh, device="cuda") * 0.8 + 0.1
    cell = torch.randn(N, T, 2, Hh, device="cuda") * 0.5
    dh = torch.randn(N, T, 2 * Hp, device="cuda") * 0.1
    for rep in range(3):
        h.KEEP_XBUF = []
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        h.blstm_onchip16_bwd(gates, cell, dh, 2 * Hp, Hp, wb16, N, T, Hh, groups)
        e.record(); torch.cuda.synchronize()
        tb = h.KEEP_XBUF[0][48:48 + 40].cpu().numpy().astype("int64").reshape(2, 2, 10)
        t0 = tb[0, 0, 0]
        print(json.dumps(dict(direction="backward", N=N, groups=groups, ms=round(s.elapsed_time(e), 3),
                              exchange_wave=[[int(v - t0) for v in row] for row in tb[0]],
                              io_wave=[[int(v - t0) for v in row] for row in tb[1]])))
    h.cluster_error_code()
