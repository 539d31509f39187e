# HISTORICAL (round 3): toggles TSSEP_GEMM_* switches, which since round 4 exist only in the experiment build
# (make -C tssep_amd/csrc exp; TSSEP_HIP_LIB=tssep_amd/libtssep_hip_exp.so).  The numbers it produced are under profiles/r3_*.
"""Split sweep of the 320 x 128 weight-gradient tile (csrc/gemm_bf16x3_tn_h160.hip) against the 128 x 128 tile at its
own best split, alternating, on the M = 320 shapes of the step; checks that the two agree bit for bit."""
import os, sys, json
sys.path.insert(0, "/root/repo")
import torch
from tssep_amd import hip_ops as h
h.GEMM_PRECISION = "bf16x3"
def timeit(fn, reps=5):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for (M, N, K) in [(320, 600, 777216), (320, 600, 194304), (320, 100, 194304), (320, 600, 25600)]:
    A = torch.randn(K, h.round_up(M, 4), device="cuda"); W = torch.randn(K, h.round_up(N, 4), device="cuda")
    row = {}
    def run(S):
        part, s = h.wgrad(A, A.shape[1], W, W.shape[1], M, N, K, with_colsum=True, splitk=S)
        return part
    os.environ["TSSEP_GEMM_TN_H160"] = "0"; 
    s0 = h.pick_splitk(M, N + 1, K)
    ref = run(s0).view(s0, -1).sum(0)
    os.environ["TSSEP_GEMM_TN_H160"] = "1"; 
    s1 = h.pick_splitk(M, N + 1, K)
    same = bool((run(s0) == 0).sum() >= 0)
    os.environ["TSSEP_GEMM_TN_H160"] = "0"; p0 = run(32).clone()
    os.environ["TSSEP_GEMM_TN_H160"] = "1"; p1 = run(32).clone()
    bit = bool(torch.equal(p0, p1))
    for rep in range(3):
        for S in (32, 48, 64, 80, 96, 112, 128):
            if K // 16 // S < 8: continue
            for flag in ("0", "1"):
                os.environ["TSSEP_GEMM_TN_H160"] = flag
                t = timeit(lambda: run(S))
                key = ("h160" if flag == "1" else "t128") + "_S%d" % S
                row[key] = round(min(row.get(key, 1e9), t), 3)
    print(json.dumps(dict(M=M, N=N, K=K, S_128=s0, S_h160=s1, bit_identical=bit, ms=row)), flush=True)
    del A, W
