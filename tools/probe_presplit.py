"""Experiment: the pre-split / asynchronous-copy GEMM probe (csrc/gemm_presplit.hip) against the
production split-bf16 GEMM on the shapes of the step (GPU box).  One JSON line per shape."""
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


for name, M, N, K in (("small", 1000, 300, 70), ("birnn1 in", 388608, 2400, 320), ("birnn0 in", 388608, 2400, 513),
                      ("birnn2 in", 97152, 2400, 1280), ("dgrad birnn0", 388608, 513, 2400),
                      ("proj", 388608, 320, 600)):
    torch.manual_seed(0)
    Kp4, Kp = h.round_up(K, 4), h.round_up(K, 16)
    A = torch.zeros(M, Kp4, device="cuda"); A[:, :K] = torch.randn(M, K, device="cuda")
    W = torch.zeros(N, Kp4, device="cuda"); W[:, :K] = torch.randn(N, K, device="cuda") / K ** 0.5
    C0 = torch.empty(M, N, device="cuda")
    h.gemm(A, Kp4, W, Kp4, C0, N, M, N, K)
    planes = [torch.empty(r, Kp, device="cuda", dtype=torch.bfloat16) for r in (M, M, N, N)]

    row = {"name": name, "M": M, "N": N, "K": K}
    for KTM, tag in ((0, "rowmajor"), (1, "ktile")):
        def split():
            h.check(L.tssep_probe_split_planes(A.data_ptr(), M, K, Kp4, planes[0].data_ptr(), planes[1].data_ptr(), KTM, st()), "split")
            h.check(L.tssep_probe_split_planes(W.data_ptr(), N, K, Kp4, planes[2].data_ptr(), planes[3].data_ptr(), KTM, st()), "split")
        split()
        for ring in (2, 3, 12, 13):
            C = torch.full((M, N), float("nan"), device="cuda")

            def run():
                h.check(L.tssep_probe_gemm_presplit(planes[0].data_ptr(), planes[1].data_ptr(), planes[2].data_ptr(),
                                                    planes[3].data_ptr(), C.data_ptr(), M, N, K, N, ring | (4096 * KTM), st()), "presplit")
            run()
            torch.cuda.synchronize()
            row[f"{tag}_ring{ring}_bit_identical"] = bool(torch.equal(C, C0))
            ms = timeit(run)
            row[f"{tag}_ring{ring}_ms"] = round(ms, 4)
            row[f"{tag}_ring{ring}_tflops"] = round(2 * M * N * K / ms / 1e9, 1)
        row[f"{tag}_split_ms"] = round(timeit(split), 4)
    ms0 = timeit(lambda: h.gemm(A, Kp4, W, Kp4, C0, N, M, N, K))
    row["production_ms"], row["production_tflops"] = round(ms0, 4), round(2 * M * N * K / ms0 / 1e9, 1)
    print(json.dumps(row), flush=True)
    del A, W, C0, planes
