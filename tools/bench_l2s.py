"""GPU box: the L2-streamed sequence-parallel recurrence (lstm_l2s.hip, EXPERIMENT build: `make -C tssep_amd/csrc exp`, not in
the product or its ABI -- profiles/r6_l2s_probe.jsonl) against the streaming fp32 kernel (parity) and the W-stationary
interleaved kernels (time), per launch, H = 300, T = 253.

    TSSEP_HIP_LIB=$PWD/tssep_amd/libtssep_hip_exp.so python tools/bench_l2s.py [N ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ctypes  # noqa: E402
from tssep_amd import _lib, hip_ops as h  # noqa: E402

_L = _lib.lib()
if not hasattr(_L, "tssep_blstm_l2s_fwd"):
    sys.exit("this library has no tssep_blstm_l2s_fwd: build `make -C tssep_amd/csrc exp` and set TSSEP_HIP_LIB")
_vp, _i64, _i = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
_L.tssep_lstm_l2s_pack_floats.restype, _L.tssep_lstm_l2s_pack_floats.argtypes = _i64, [_i, _i]
_L.tssep_lstm_pack_l2s.restype, _L.tssep_lstm_pack_l2s.argtypes = _i, [_vp, _vp, _i, _vp, _vp]
_L.tssep_blstm_l2s_fwd.restype, _L.tssep_blstm_l2s_fwd.argtypes = _i, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i, _vp]
HAS8 = hasattr(_L, "tssep_blstm_l2s8_fwd")
if HAS8:
    _L.tssep_blstm_l2s8_fwd.restype, _L.tssep_blstm_l2s8_fwd.argtypes = _i, [_vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _i, _vp]


def _l2s8_fwd(gates, cell, hout, ldo, dstride, wf, N, T, H):
    h.check(_L.tssep_blstm_l2s8_fwd(gates.data_ptr(), cell.data_ptr(), hout.data_ptr(), ldo, dstride, wf.data_ptr(), N, T, H,
                                    h._stream()), "blstm_l2s8_fwd")


def _pack_l2s(w_hh_f, w_hh_r, H):
    buf = torch.empty(int(_L.tssep_lstm_l2s_pack_floats(H, 0)), device=w_hh_f.device, dtype=torch.float32)
    a, b = w_hh_f.detach().float().contiguous(), w_hh_r.detach().float().contiguous()
    h.check(_L.tssep_lstm_pack_l2s(a.data_ptr(), b.data_ptr(), H, buf.data_ptr(), h._stream()), "lstm_pack_l2s")
    return buf


def _l2s_fwd(gates, cell, hout, ldo, dstride, wf, N, T, H):
    h.check(_L.tssep_blstm_l2s_fwd(gates.data_ptr(), cell.data_ptr(), hout.data_ptr(), ldo, dstride, wf.data_ptr(), N, T, H,
                                   h._stream()), "blstm_l2s_fwd")


h.lstm_pack_l2s = lambda a, b, H, which=0: _pack_l2s(a, b, H)
h.blstm_l2s_fwd = _l2s_fwd

T, Hh, I = int(os.environ.get("L2S_T", 253)), int(os.environ.get("L2S_H", 300)), 320
torch.manual_seed(0)
lstm = torch.nn.LSTM(I, Hh, bidirectional=True, batch_first=True).cuda()
names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
plist = [getattr(lstm, n) for n in names] + [getattr(lstm, n + "_reverse") for n in names]
pk = h.lstm_pack(plist, Hh, I)
wl2s = h.lstm_pack_l2s(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse, Hh)
w16 = h.lstm_pack_onchip16(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse, Hh)
HAS_BWD = hasattr(h, "blstm_l2s_bwd")
if HAS_BWD:
    wl2sb = h.lstm_pack_l2s(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse, Hh, 1)
    w16b = h.lstm_pack_onchip16_bwd(lstm.weight_hh_l0, lstm.weight_hh_l0_reverse, Hh)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


for N in [int(a) for a in sys.argv[1:]] or [40, 3072]:
    Hp = Hh
    g0 = torch.randn(N * T, 8 * Hh, device="cuda") * 0.5
    res = dict(N=N, T=T, H=Hh)
    out = {}
    for name, fn in (("stream", lambda g, c, ho: h.blstm_fwd(g, c, ho, 2 * Hp, Hp, pk["whh_f"], N, T, Hh)),
                     ("l2s", lambda g, c, ho: h.blstm_l2s_fwd(g, c, ho, 2 * Hp, Hp, wl2s, N, T, Hh))):
        g, c, ho = g0.clone(), torch.zeros(N, T, 2, Hh, device="cuda"), torch.zeros(N, T, 2 * Hp, device="cuda")
        fn(g, c, ho)
        torch.cuda.synchronize()
        out[name] = (g, c, ho)
    for k, nm in enumerate(("gates", "cell", "h")):
        res["fwd_rel_" + nm] = rel(out["l2s"][k], out["stream"][k])
    if HAS8:      # the eight-wave K-split variant (v3)
        g, c, ho = g0.clone(), torch.zeros(N, T, 2, Hh, device="cuda"), torch.zeros(N, T, 2 * Hp, device="cuda")
        _l2s8_fwd(g, c, ho, 2 * Hp, Hp, wl2s, N, T, Hh)
        torch.cuda.synchronize()
        for k, (nm, t_) in enumerate(zip(("gates", "cell", "h"), (g, c, ho))):
            res["fwd8_rel_" + nm] = rel(t_, out["stream"][k])
    g, c, ho = g0.clone(), torch.zeros(N, T, 2, Hh, device="cuda"), torch.zeros(N, T, 2 * Hp, device="cuda")
    for rep in range(2):      # interleaved A/B
        res.setdefault("l2s_fwd_ms", []).append(round(timeit(lambda: h.blstm_l2s_fwd(g, c, ho, 2 * Hp, Hp, wl2s, N, T, Hh)), 3))
        if HAS8:
            res.setdefault("l2s8_fwd_ms", []).append(round(timeit(lambda: _l2s8_fwd(g, c, ho, 2 * Hp, Hp, wl2s, N, T, Hh)), 3))
        g16 = h.onchip16_groups(N, Hh, g.device)
        if g16:
            res.setdefault("onchip16_fwd_ms", []).append(round(timeit(
                lambda: h.blstm_onchip16_fwd(g, c, ho, 2 * Hp, Hp, w16, N, T, Hh, g16)), 3))
            h.check_cluster_errors()
    res["l2s_fwd_us_per_step"] = round(min(res["l2s_fwd_ms"]) * 1e3 / T, 2)
    if HAS8:
        res["l2s8_fwd_us_per_step"] = round(min(res["l2s8_fwd_ms"]) * 1e3 / T, 2)
    if HAS_BWD:
        dh = torch.randn(N, T, 2 * Hp, device="cuda") * 0.1
        ga, ca, _ = out["stream"]
        outb = {}
        for name, fn in (("stream", lambda g_: h.blstm_bwd(g_, ca, dh, 2 * Hp, Hp, pk["whh_b"], N, T, Hh)),
                         ("l2s", lambda g_: h.blstm_l2s_bwd(g_, ca, dh, 2 * Hp, Hp, wl2sb, N, T, Hh))):
            g_ = ga.clone()
            fn(g_)
            torch.cuda.synchronize()
            outb[name] = g_
        res["bwd_rel_dgates"] = rel(outb["l2s"], outb["stream"])
        g_ = ga.clone()
        for rep in range(2):
            res.setdefault("l2s_bwd_ms", []).append(round(timeit(lambda: h.blstm_l2s_bwd(g_, ca, dh, 2 * Hp, Hp, wl2sb, N, T, Hh)), 3))
            gb = h.onchip16_bwd_groups(N, Hh, g_.device)
            if gb:
                res.setdefault("onchip16_bwd_ms", []).append(round(timeit(
                    lambda: h.blstm_onchip16_bwd(g_, ca, dh, 2 * Hp, Hp, w16b, N, T, Hh, gb)), 3))
                h.check_cluster_errors()
        res["l2s_bwd_us_per_step"] = round(min(res["l2s_bwd_ms"]) * 1e3 / T, 2)
    print(json.dumps(res), flush=True)
