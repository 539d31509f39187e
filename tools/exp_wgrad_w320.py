"""Would the 256 x 320 weight-gradient tile pay for the dW_ih GEMMs with N = 320 q (+ the ones column)?  The same
shapes WITHOUT the ones column on tn_w160 (256 x 320 workgroups) at several split counts, beside the library's choice
WITH the ones column (GPU box):   python tools/exp_wgrad_w320.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as h

T = 253


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


SHAPES = [("dW_ih birnn0", 2400, 513, 3072 * T), ("dW_ih birnn1", 2400, 320, 3072 * T), ("dW_ih birnn2", 2400, 1280, 768 * T)]
if "--prenet" in sys.argv:      # N = 553 + 1: ragged last 320-column tile, column 552 and the ones column on the VALU
    SHAPES = [("dW_ih pre-net", 2400, 553, 768 * T), ("N 873", 2400, 873, 768 * T), ("N 552 no VALU column", 2400, 552, 768 * T)]
if "--other" in sys.argv:       # shapes of other configurations the dispatch rule also sends to the eight-wave kernel
    SHAPES = [("N 768", 2400, 768, 768 * T), ("N 1024", 2400, 1024, 768 * T), ("N 1536", 2400, 1536, 768 * T), ("N 640", 2400, 640, 3072 * T),
              ("N 2560 (8 speakers)", 2400, 2560, 768 * T), ("M 1200 N 512", 1200, 512, 3072 * T), ("small K: dW_ih birnn0", 2400, 513, 32 * T)]
for name, M, N, R in SHAPES:
    dy = torch.randn(R, M, device="cuda") * 0.1
    x = torch.randn(R, h.round_up(N + 1, 4), device="cuda") * 0.5
    h.GEMM_PREFER = ()
    h.GEMM_LOG = []
    part, S = h.wgrad(dy, M, x, x.shape[1], M, N, R, with_colsum=True)
    ran = h.GEMM_LOG[-1][0]; h.GEMM_LOG = None
    ms = timeit(lambda: h.wgrad(dy, M, x, x.shape[1], M, N, R, with_colsum=True))
    print(json.dumps({"gemm": name, "with_ones_column": True, "kernel": ran, "splits": S, "ms": round(ms, 4)}), flush=True)
    for force in ("tn_w160", "tn_p320", "tn_big"):
        if "--sweep" not in sys.argv:
            h.GEMM_PREFER = (force,)
            h.GEMM_LOG = []
            try:
                part, S = h.wgrad(dy, M, x, x.shape[1], M, N, R, with_colsum=True)
            except Exception:
                h.GEMM_LOG = None
                continue
            ran = h.GEMM_LOG[-1][0]; h.GEMM_LOG = None
            if ran == force:
                ms = timeit(lambda: h.wgrad(dy, M, x, x.shape[1], M, N, R, with_colsum=True))
                print(json.dumps({"gemm": name, "with_ones_column": True, "forced": force, "splits": S, "ms": round(ms, 4)}), flush=True)
            continue
        for S in (8, 16, 24, 32, 48, 64):
            h.GEMM_PREFER = (force,)
            h.GEMM_LOG = []
            try:
                h.wgrad(dy, M, x, x.shape[1], M, N, R, splitk=S, with_colsum=True)
            except Exception as e:
                h.GEMM_LOG = None
                continue
            ran = h.GEMM_LOG[-1][0]; h.GEMM_LOG = None
            if ran != force:
                continue
            ms = timeit(lambda: h.wgrad(dy, M, x, x.shape[1], M, N, R, splitk=S, with_colsum=True))
            print(json.dumps({"gemm": name, "with_ones_column": True, "kernel": ran, "splits": S, "ms": round(ms, 4),
                              "tflops": round(2 * M * N * R / ms / 1e9, 1)}), flush=True)
h.GEMM_PREFER = ()
