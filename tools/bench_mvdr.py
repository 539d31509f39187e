"""Microbenchmark of the mask-based MVDR beamformer (TorchBF) on one MI355X:
HIP-event time of the whole pipeline and of each stage vs the algorithmic HBM bytes, and the CPU
oracle beside it on a frequency slice.  One JSON line per configuration."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tssep_amd import _lib, hip_ops as H  # noqa: E402

HBM_PEAK = 8000.0
F64_VECTOR_PEAK = 78.6        # TFLOP/s: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz


def ev_ms(fn, iters):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    L = _lib.lib()
    for name, (B, K, M, D, T, F) in {
            "cfg3 eval: 4 spk, 6 ch, 4 s": (1, 4, 2, 6, 253, 513),
            "cfg5 eval: 8 spk, 6 ch, 30 s": (1, 8, 2, 6, 1878, 513),
            "cfg5 eval, target mask only": (1, 8, 1, 6, 1878, 513),
            "8 x cfg5 eval batched": (8, 8, 2, 6, 1878, 513)}.items():
        g = torch.Generator().manual_seed(0)
        Y = torch.randn(B, D, T, F, dtype=torch.complex128, generator=g).cuda()
        m = torch.rand(B, K, M, T, F, generator=g).cuda()
        total = ev_ms(lambda: H.mvdr_souden(m, Y, 0, check_singular=False), args.iters)
        nb = L.tssep_mvdr_partial_bytes(B, K, D, T, F)
        part = torch.empty(nb // 8, dtype=torch.float64, device="cuda")
        w = torch.empty(B * K * D * F * 2, dtype=torch.float64, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        enh = torch.empty(B, K, T, F, 2, dtype=torch.float64, device="cuda")
        Yr = torch.view_as_real(Y)
        st = torch.cuda.current_stream().cuda_stream
        t_psd = ev_ms(lambda: L.tssep_mvdr_psd(Yr.data_ptr(), m.data_ptr(), 0, part.data_ptr(),
                                               B, K, M, D, T, F, st), args.iters)
        t_w = ev_ms(lambda: L.tssep_mvdr_weights(part.data_ptr(), w.data_ptr(), info.data_ptr(), B, K,
                                                 D, T, F, 0, 1e-300, st), args.iters)
        t_app = ev_ms(lambda: L.tssep_mvdr_apply(Yr.data_ptr(), w.data_ptr(), m.data_ptr(), 0,
                                                 enh.data_ptr(), B, K, M, D, T, F, 0, 0.0, st),
                      args.iters)
        b_psd = B * T * F * (16 * D + 4 * K * M) + nb
        b_app = B * T * F * (16 * D + 16 * K)
        alg = B * T * F * (32 * D + 4 * K * M + 16 * K)
        row = {"config": name, "B": B, "K": K, "M": M, "D": D, "T": T, "F": F, "dtype": "c128",
               "total_ms": round(total, 4), "frames_per_s": round(B * T / total * 1e3, 1),
               "algorithmic_bytes": alg, "achieved_GBps": round(alg / total / 1e6, 1),
               "frac_of_hbm_peak": round(alg / total / 1e6 / HBM_PEAK, 4),
               "psd_ms": round(t_psd, 4), "psd_GBps": round(b_psd / t_psd / 1e6, 1),
               # statistics pass: 7 D^2 fp64 flop per (frame, bin, speaker) -- pair products once,
               # weighted into the target and the interference accumulators
               "psd_f64_tflops": round(B * T * F * K * 7 * D * D / t_psd / 1e9, 2),
               "psd_frac_of_f64_vector_peak": round(B * T * F * K * 7 * D * D / t_psd / 1e9 / F64_VECTOR_PEAK, 4),
               "weights_ms": round(t_w, 4), "apply_ms": round(t_app, 4),
               "apply_GBps": round(b_app / t_app / 1e6, 1),
               "apply_frac_of_hbm_peak": round(b_app / t_app / 1e6 / HBM_PEAK, 4),
               "psd_partial_bytes": nb}
        if not args.no_cpu and B == 1:
            from oracle import enhancer as oenh          # checker / CPU baseline only
            fs = 16
            mc, Yc = m[..., :fs].cpu().numpy(), Y[..., :fs].cpu().numpy()
            t0 = time.perf_counter()
            want = oenh.torch_bf(mc, Yc, 0)
            dt = time.perf_counter() - t0
            got = H.mvdr_souden(m, Y, 0)[..., :fs].cpu().numpy()
            row["cpu_oracle_frames_per_s"] = round(B * T / (dt * F / fs), 1)
            row["cpu_sample"] = f"{fs} of {F} bins, numpy complex128, scaled to {F}"
            row["max_rel_err_vs_oracle"] = float(np.abs(got - want).max() / np.abs(want).max())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
