"""Micro-benchmark (GPU box): streaming vs cluster BLSTM recurrence, per launch, H=300, T=253."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

T, Hh, I = 253, 300, 320
torch.manual_seed(0)
lstm = torch.nn.LSTM(I, Hh, bidirectional=True, batch_first=True).cuda()
names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
plist = [getattr(lstm, n) for n in names] + [getattr(lstm, n + "_reverse") for n in names]
pk = h.lstm_pack(plist, Hh, I)
cf, cb = h.lstm_pack_cluster(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse, Hh)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


wf3, wb3 = h.lstm_pack_onchip(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse, Hh)
for N in [int(a) for a in sys.argv[1:]] or [8, 32, 64, 128, 256, 512, 1024]:
    g0 = torch.randn(N * T, 8 * Hh, device="cuda") * 0.5
    cell = torch.empty(N, T, 2, Hh, device="cuda")
    hout = torch.zeros(N, T, 2 * Hh, device="cuda")
    dh = torch.randn(N, T, 2 * Hh, device="cuda") * 0.1
    res = dict(N=N)
    for name, fwd, bwd in (("stream", lambda g: h.blstm_fwd(g, cell, hout, 2 * Hh, Hh, pk["whh_f"], N, T, Hh),
                            lambda g: h.blstm_bwd(g, cell, dh, 2 * Hh, Hh, pk["whh_b"], N, T, Hh)),
                           ("cluster2", lambda g: h.blstm_cluster_fwd(g, cell, hout, 2 * Hh, Hh, cf, N, T, Hh, 2),
                            lambda g: h.blstm_cluster_bwd(g, cell, dh, 2 * Hh, Hh, cb, N, T, Hh, 2)),
                           ):
        g = g0.clone()
        res[name + "_fwd_ms"] = round(timeit(lambda: fwd(g)), 3)
        res[name + "_bwd_ms"] = round(timeit(lambda: bwd(g)), 3)
        h.check_cluster_errors()
    g = g0.clone()
    res["onchip_fwd_ms"] = round(timeit(lambda: h.blstm_onchip_fwd(g, cell, hout, 2 * Hh, Hh, wf3, N, T, Hh)), 3)
    h.check_cluster_errors()
    res["onchip_bwd_ms"] = round(timeit(lambda: h.blstm_onchip_bwd(g, cell, dh, 2 * Hh, Hh, wb3, N, T, Hh)), 3)
    h.check_cluster_errors()
    # microseconds per time step and resident round (48 XCD-local clusters x 32 sequences x 1 direction = 768
    # sequences per round): "unloaded" = a launch of <= 32 sequences (one or two clusters, the exchange chain alone),
    # "loaded" = full rounds (all clusters streaming their activations through the same L2s / HBM)
    rounds = -(-(2 * -(-N // 32)) // 48)
    res["rounds"] = rounds
    res["onchip_fwd_us_per_step"] = round(res["onchip_fwd_ms"] * 1e3 / (T * rounds), 2)
    res["onchip_bwd_us_per_step"] = round(res["onchip_bwd_ms"] * 1e3 / (T * rounds), 2)
    for lay in [int(a) for a in os.environ.get("LAYOUTS", "").split(",") if a]:
        g = g0.clone()
        res["onchip_fwd_lay%d_ms" % lay] = round(timeit(lambda: h.blstm_onchip_fwd(g, cell, hout, 2 * Hh, Hh, wf3, N, T, Hh, lay)), 3)
        res["err%d" % lay] = int(h._err_flag(g.device)[0].item()); h._err_flag(g.device)[0].zero_()
    if os.environ.get("CROSS_XCD"):
        g = g0.clone()
        res["onchip_crossxcd_fwd_ms"] = round(timeit(lambda: h.blstm_onchip_fwd(g, cell, hout, 2 * Hh, Hh, wf3, N, T, Hh, 8)), 3)
        res["onchip_crossxcd_bwd_ms"] = round(timeit(lambda: h.blstm_onchip_bwd(g, cell, dh, 2 * Hh, Hh, wb3, N, T, Hh, 8)), 3)
        h.check_cluster_errors()
    print(json.dumps(res), flush=True)
