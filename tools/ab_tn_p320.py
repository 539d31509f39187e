"""192 x 320 weight-gradient kernel (csrc/gemm_bf16x3_tn_p320.hip) against the other weight-gradient kernels on the
shapes with N = 320 q (+ the ones column), each at the split count the library picks for it; interleaved timing
(tools/sweep_gemm_shapes.time_calls).  `python tools/ab_tn_p320.py [batch] > profiles/rN_ab_wgrad_tn_p320.jsonl`"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tssep_amd.hip_ops as H  # noqa: E402
from sweep_gemm_shapes import time_calls  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
    R1, R4 = B * 253, B * 4 * 253
    shapes = [("dW_ih birnn1", R4, 2400, 320, True), ("dW linear2 (logit layer)", R1, 2052, 320, False),
              ("dW_ih birnn2", R1, 2400, 1280, True), ("dW_ih birnn0", R4, 2400, 513, True)]
    H.GEMM_PRECISION = "bf16x3"
    for name, R, M, N, ones in shapes:
        dY = torch.randn(R, H.round_up(M, 4), device="cuda")
        X = torch.randn(R, H.round_up(N + 1, 4), device="cuda")
        calls, splits, sums = {}, {}, {}
        for kern in ("tn_p320", "tn_big", "tn_tall", "tn"):
            log = H.GEMM_LOG = []
            with H.prefer_gemm_kernels(kern):
                part, S = H.wgrad(dY, dY.shape[1], X, X.shape[1], M, N, R, with_colsum=ones)
            H.GEMM_LOG = None
            if log[0][0] != kern:
                continue
            splits[kern] = S
            sums[kern] = part.view(S, M, -1).double().sum(0)

            def call(kern=kern):
                with H.prefer_gemm_kernels(kern):
                    H.wgrad(dY, dY.shape[1], X, X.shape[1], M, N, R, with_colsum=ones)
            calls[kern] = call
        torch.cuda.synchronize()
        ms = time_calls(calls, 5)
        rec = dict(name=name, M=M, N=N + int(ones), K=R)
        ref = sums.get("tn", next(iter(sums.values())))
        for k, v in ms.items():
            rec[k + "_ms"] = round(v, 4)
            rec[k + "_tflops"] = round(2 * M * (N + int(ones)) * R / v / 1e9, 1)
            rec[k + "_splits"] = splits[k]
            rec[k + "_max_rel_diff_vs_tn"] = float(((sums[k] - ref).abs().max() / ref.abs().max()))
        print(json.dumps(rec), flush=True)
        del dY, X, calls, sums
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
