# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Split-K sweep of the step's weight-gradient GEMMs (GPU box): wgrad + split reduction, per S.
usage: python tools/sweep_splitk.py [batch]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tssep_amd import hip_ops as H

H.GEMM_PRECISION = "bf16x3"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
R1, R4 = B * 253, B * 4 * 253


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


for name, M, N, R, colsum, shift in [("W_ih pre_net", 2400, 553, R1, True, 0), ("W_ih birnn0", 2400, 513, R4, True, 0),
                                     ("W_ih birnn1", 2400, 320, R4, True, 0), ("W_ih birnn2", 2400, 1280, R1, True, 0),
                                     ("W_hh (1 dir) 4K", 1200, 300, R4, False, -1), ("W_hh (1 dir) K", 1200, 300, R1, False, -1),
                                     ("proj 513", 513, 600, R1, True, 0), ("proj 320", 320, 600, R4, True, 0),
                                     ("linear2", 2052, 320, R1, True, 0)]:
    dy = torch.randn(R, H.round_up(M, 4), device="cuda")
    x = torch.randn(R, H.round_up(N, 4), device="cuda")
    ldp = H.round_up(N + 1, 4) if colsum else N
    dw = torch.empty(M, N, device="cuda"); db = torch.empty(M, device="cuda")
    row = dict(name=name, M=M, N=N, K=R, default=H.pick_splitk(M, N + 1 if colsum else N, R))
    for S in (8, 16, 24, 32, 40, 48, 56, 64):
        def f():
            part, s = H.wgrad(dy, dy.shape[1], x, x.shape[1], M, N, R, with_colsum=colsum, splitk=S,
                              b_kshift=shift, kperiod=253 if shift else 0)
            if colsum:
                H.reduce_splits_bias(part, s, M, N, ldp, dw, db)
            else:
                H.reduce_splits(part, s, M * N, dw)
        row[f"S{S}"] = round(timeit(f), 4)
    print(json.dumps(row), flush=True)
    del dy, x
