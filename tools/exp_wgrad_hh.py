"""dW_hh weight gradient (time-shifted, one direction) on the kernels that cover it, beside unshifted / unpadded
controls of the same size (GPU box):   python tools/exp_wgrad_hh.py [sequences ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

T, Hh = 253, 300
Hp = h.round_up(Hh, 4)


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


for N in [int(a) for a in sys.argv[1:]] or [3072, 768]:
    R = N * T
    dg = torch.randn(R, 8 * Hh, device="cuda") * 0.1
    hh = torch.randn(R, 2 * Hp, device="cuda") * 0.5
    big = torch.randn(R, 1280 + 320, device="cuda") * 0.1
    cases = [("dW_hh shifted 1200x300", (dg, 8 * Hh, hh, 2 * Hp, 4 * Hh, Hh, R), dict(b_kshift=-1, kperiod=T)),
             ("unshifted 1200x300", (dg, 8 * Hh, hh, 2 * Hp, 4 * Hh, Hh, R), {}),
             ("unshifted 1280x320 (no edge)", (big, 1600, (big, 1280), 1600, 1280, 320, R), {})]
    for name, args, kw in cases:
        for force in ("tn_w160", "tn_p320", "auto"):
            h.GEMM_PREFER = () if force == "auto" else (force,)
            h.GEMM_LOG = []
            try:
                part, S = h.wgrad(*args, **kw)
            except Exception as e:          # kernel does not cover the request
                print(json.dumps({"case": name, "force": force, "error": str(e)[:80]})); continue
            ran = h.GEMM_LOG[-1][0]
            h.GEMM_LOG = None
            if force != "auto" and ran != force:
                continue
            ms = timeit(lambda: h.wgrad(*args, **kw))
            M_, N_ = args[4], args[5]
            print(json.dumps({"sequences": N, "case": name, "kernel": ran, "splits": S, "ms": round(ms, 4),
                              "tflops": round(2 * M_ * N_ * R / ms / 1e9, 1)}), flush=True)
h.GEMM_PREFER = ()
