"""Persistent big-tile GEMM kernel (csrc/gemm_bf16x3_bigp.hip) against the kernels it competes with, on the row x row
shapes of the step whose store is plain / bias / bias + Tanh: interleaved timing (tools/sweep_gemm_shapes.time_calls),
bit comparison of the results.  `python tools/ab_big_p.py [batch] > profiles/rN_ab_gemm_big_p.jsonl`"""
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
    shapes = [("pre_net in", R1, 2400, 553, 0), ("birnn0 in", R4, 2400, 513, 0), ("birnn1 in", R4, 2400, 320, 0),
              ("birnn2 in", R1, 2400, 1280, 0), ("dgrad proj dh", R4, 600, 320, 0), ("dgrad proj dh (pre)", R1, 600, 513, 0),
              ("proj 600->256 tanh", R4, 256, 600, 1), ("proj 1200->512 tanh", R1, 512, 600, 1),
              ("dgrad birnn0 dx", R4, 513, 2400, 0), ("pre_net proj", R1, 513, 600, 0),
              ("proj 600->320 tanh", R4, 320, 600, 1), ("proj 1200->320 tanh (pre)", R1, 320, 600, 1)]
    H.GEMM_PRECISION = "bf16x3"
    tot = {}
    for name, M, N, K, act in shapes:
        A = torch.randn(M, H.round_up(K, 4), device="cuda")
        W = torch.randn(N, H.round_up(K, 4), device="cuda") / K ** 0.5
        bias = torch.randn(N, device="cuda")
        outs, calls = {}, {}
        probes = [p for p in os.environ.get("AB_BIGP_PROBES", "").split(",") if p]      # (experiment build, TSSEP_HIP_LIB)
        for kern in ["big_p", "big_p320"] + ["big_p@" + p for p in probes] + ["big", "stream", "nt_w160", "tall4", "tall4_xcol", "tall2"]:
            C = torch.empty(M, H.round_up(N, 4), device="cuda")
            kname, _, probe = kern.partition("@")

            def call(kname=kname, probe=probe, C=C):
                os.environ["TSSEP_BIGP_PROBE"] = probe or "0"      # "big_p@24": timing probe 24 (experiment build only)
                with H.prefer_gemm_kernels(kname):
                    H.gemm(A, A.shape[1], W, W.shape[1], C, C.shape[1], M, N, K, bias=bias, act=act)

            log = H.GEMM_LOG = []
            call()
            H.GEMM_LOG = None
            if log[0][0] != kname:
                continue
            outs[kern], calls[kern] = C, call
        torch.cuda.synchronize()
        ms = time_calls(calls, 5)
        ref = outs.get("tall2", outs.get("tall4", outs.get("big")))
        rec = dict(name=name, M=M, N=N, K=K, act=act)
        for k, v in ms.items():
            rec[k + "_ms"] = round(v, 4)
            rec[k + "_tflops"] = round(2 * M * N * K / v / 1e9, 1)
            rec[k + "_bit_identical"] = bool(torch.equal(outs[k][:, :N], ref[:, :N]))
            tot[k] = tot.get(k, 0) + v
        print(json.dumps(rec), flush=True)
        del A, W, outs, calls
        torch.cuda.empty_cache()
    print(json.dumps({"total_ms": {k: round(v, 3) for k, v in tot.items()}}))
    # the logit layer: 4 x 513 bins per frame, remapped to [B, K, T, F] with the per-utterance speaker permutation
    Kspk, F, P, T = 4, 513, 320, 253
    A = torch.randn(R1, P, device="cuda"); W = torch.randn(Kspk * F, P, device="cuda") / P ** 0.5
    bias = torch.randn(Kspk * F, device="cuda")
    perm = torch.stack([torch.randperm(Kspk) for _ in range(B)]).int().cuda()
    rm = dict(T=T, K=1, sb=Kspk * T * F, sk=0, st=F, cm=F, co=T * F, perm=perm, perm_ld=Kspk)
    outs, calls = {}, {}
    for kern in ("big_p", "big", "tall4", "tall2", "nt_w160", "pipe"):
        C = torch.empty(B, Kspk, T, F, device="cuda")

        def call(kern=kern, C=C):
            with H.prefer_gemm_kernels(kern):
                H.gemm(A, P, W, P, C, 0, R1, Kspk * F, P, bias=bias, remap=rm)

        log = H.GEMM_LOG = []
        call()
        H.GEMM_LOG = None
        if log[0][0] == kern:
            outs[kern], calls[kern] = C, call
    ms = time_calls(calls, 5)
    rec = dict(name="linear2 (logit layer, remapped)", M=R1, N=Kspk * F, K=P)
    for k, v in ms.items():
        rec[k + "_ms"] = round(v, 4)
        rec[k + "_tflops"] = round(2 * R1 * Kspk * F * P / v / 1e9, 1)
        rec[k + "_bit_identical"] = bool(torch.equal(outs[k], outs["tall2"]))
    print(json.dumps(rec), flush=True)
    # dgrad of birnn1's input: d(gates) x W_ih with the Tanh backward folded in, N = 320
    A = torch.randn(R4, 2400, device="cuda"); W = torch.randn(320, 2400, device="cuda") / 2400 ** 0.5
    Y = torch.tanh(torch.randn(R4, 320, device="cuda"))
    outs, calls = {}, {}
    for kern in ("big_p320", "nt_w160", "big", "tall2"):
        C = torch.empty(R4, 320, device="cuda")

        def call(kern=kern, C=C):
            with H.prefer_gemm_kernels(kern):
                H.gemm(A, 2400, W, 2400, C, 320, R4, 320, 2400, act=2, aux=(Y, 320))

        log = H.GEMM_LOG = []
        call()
        H.GEMM_LOG = None
        if log[0][0] == kern:
            outs[kern], calls[kern] = C, call
    ms = time_calls(calls, 5)
    rec = dict(name="dgrad birnn1 dx (folded Tanh backward)", M=R4, N=320, K=2400)
    for k, v in ms.items():
        rec[k + "_ms"] = round(v, 4)
        rec[k + "_tflops"] = round(2 * R4 * 320 * 2400 / v / 1e9, 1)
        rec[k + "_bit_identical"] = bool(torch.equal(outs[k], outs["tall2"]))
    print(json.dumps(rec), flush=True)
    del A, W, Y, outs, calls
    torch.cuda.empty_cache()
    # dgrad of birnn2's input: d(gates) x W_ih with the Tanh backward of the layer below folded into the store and the
    # speaker combination undone ([B T, K 320] -> rows (b, k, t) x 320)
    hd, G = 320, 2400
    A = torch.randn(R1, G, device="cuda"); W = torch.randn(Kspk * hd, G, device="cuda") / G ** 0.5
    Y = torch.tanh(torch.randn(R1, Kspk * hd, device="cuda"))
    rm = dict(T=T, K=1, sb=Kspk * T * hd, sk=0, st=hd, cm=hd, co=T * hd)
    outs, calls = {}, {}
    for kern in ("big_p", "big", "tall4", "tall2"):
        C = torch.empty(R1 * Kspk, hd, device="cuda")

        def call(kern=kern, C=C):
            with H.prefer_gemm_kernels(kern):
                H.gemm(A, G, W, G, C, 0, R1, Kspk * hd, G, act=2, aux=(Y, Kspk * hd), remap=rm)

        log = H.GEMM_LOG = []
        call()
        H.GEMM_LOG = None
        if log[0][0] == kern:
            outs[kern], calls[kern] = C, call
    ms = time_calls(calls, 5)
    rec = dict(name="dgrad birnn2 dx (folded Tanh backward, un-combined)", M=R1, N=Kspk * hd, K=G)
    for k, v in ms.items():
        rec[k + "_ms"] = round(v, 4)
        rec[k + "_tflops"] = round(2 * R1 * Kspk * hd * G / v / 1e9, 1)
        rec[k + "_bit_identical"] = bool(torch.equal(outs[k], outs["tall2"]))
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
