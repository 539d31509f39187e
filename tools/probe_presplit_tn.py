"""Experiment: plane-fed weight-gradient GEMM probe (csrc/gemm_presplit.hip) against the production
transpose-read kernel on the wgrad shapes of the step (GPU box).  One JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tssep_amd import _lib, hip_ops as h  # noqa: E402

h.GEMM_PRECISION = "bf16x3"
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, M, N, R in (("small", 260, 130, 3200), ("wgrad W_ih birnn0", 2400, 513, 194304), ("wgrad proj", 320, 600, 194304),
                      ("wgrad linear2", 2052, 320, 48576), ("wgrad W_ih birnn2", 2400, 1280, 48576),
                      ("wgrad W_ih birnn1", 2400, 320, 194304)):
    torch.manual_seed(0)
    dY = torch.randn(R, M, device="cuda") / R ** 0.5
    X = torch.zeros(R, h.round_up(N, 4), device="cuda"); X[:, :N] = torch.randn(R, N, device="cuda")
    part, S = h.wgrad(dY, M, X, X.shape[1], M, N, R)
    ref = part.view(S, M, N).clone()
    Mp, Np = h.round_up(M, 16), h.round_up(N, 16)
    planes = [torch.empty((Mp // 16) * R * 16, device="cuda", dtype=torch.bfloat16) for _ in range(2)] + \
             [torch.empty((Np // 16) * R * 16, device="cuda", dtype=torch.bfloat16) for _ in range(2)]

    def split():
        h.check(L.tssep_probe_split_planes(dY.data_ptr(), R, M, M, planes[0].data_ptr(), planes[1].data_ptr(), 1, st()), "split")
        h.check(L.tssep_probe_split_planes(X.data_ptr(), R, N, X.shape[1], planes[2].data_ptr(), planes[3].data_ptr(), 1, st()), "split")
    split()
    row = {"name": name, "M": M, "N": N, "K": R, "splitk": S}
    for ring in (2, 3):
        C = torch.full((S, M * N), float("nan"), device="cuda")

        def run():
            h.check(L.tssep_probe_gemm_presplit_tn(planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(),
                                                   planes[3].data_ptr(), C.data_ptr(), M, N, R, N, S, M * N, ring, st()), "tn probe")
        run()
        torch.cuda.synchronize()
        row[f"ring{ring}_bit_identical"] = bool(torch.equal(C.view(S, M, N), ref))
        row[f"ring{ring}_max_abs_diff"] = float((C.view(S, M, N) - ref).abs().max())
        ms = timeit(run)
        row[f"ring{ring}_ms"], row[f"ring{ring}_tflops"] = round(ms, 4), round(2 * M * N * R / ms / 1e9, 1)
    ms0 = timeit(lambda: h.wgrad(dY, M, X, X.shape[1], M, N, R))
    row["production_ms"], row["production_tflops"] = round(ms0, 4), round(2 * M * N * R / ms0 / 1e9, 1)
    row["split_ms"] = round(timeit(split), 4)
    print(json.dumps(row), flush=True)
    del dY, X, part, ref, planes
