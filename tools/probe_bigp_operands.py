"""GPU box: what staging an operand of the persistent row x row GEMM (big_p) costs -- timing probes, results wrong by
construction.  One process per library variant built with -DBIGP_PROBE_NOA / -DBIGP_PROBE_NOB
(1: the operand is neither loaded, split nor written; 2: eight 1-KB LDS-DMA copies per wave and stage in its place):

    TSSEP_HIP_LIB=$PWD/tssep_amd/libtssep_hip_probe_a1.so python tools/probe_bigp_operands.py [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tssep_amd.hip_ops as H  # noqa: E402
from sweep_gemm_shapes import time_calls  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
R1, R4 = B * 253, B * 4 * 253
shapes = [("pre_net in", R1, 2400, 553), ("birnn0 in", R4, 2400, 513), ("birnn1 in", R4, 2400, 320), ("birnn2 in", R1, 2400, 1280),
          ("dgrad birnn0 dx", R4, 513, 2400)]
H.GEMM_PRECISION = "bf16x3"
out = dict(lib=os.path.basename(os.environ.get("TSSEP_HIP_LIB", "libtssep_hip.so")), batch=B)
for name, M, N, K in shapes:
    A = torch.randn(M, H.round_up(K, 4), device="cuda")
    W = torch.randn(N, H.round_up(K, 4), device="cuda") / K ** 0.5
    C = torch.empty(M, H.round_up(N, 4), device="cuda")

    def call():
        with H.prefer_gemm_kernels("big_p"):
            H.gemm(A, A.shape[1], W, W.shape[1], C, C.shape[1], M, N, K)

    log = H.GEMM_LOG = []
    call()
    H.GEMM_LOG = None
    ms = time_calls({"big_p": call}, 5)["big_p"]
    out[name] = dict(kernel=log[0][0], ms=round(ms, 4), tflops=round(2 * M * N * K / ms / 1e9, 1))
    del A, W, C
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
