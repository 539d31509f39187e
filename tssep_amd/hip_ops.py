"""Tensor-level wrappers over the C ABI (include/tssep_hip.h).

PyTorch is used for device memory and the current HIP stream only; every computation below
is a call into libtssep_hip.so.  Nothing here falls back to ATen math.
"""
import ctypes
import itertools
import math

import numpy as np
import torch

from . import _lib
from ._lib import GemmArgs, LstmSizes, check


# ---- optional live kernel timing (bench.py): HIP events on the launch stream -----------------
KERNEL_TIMING = False
KERNEL_TIMERS = {}      # name -> list of (start_event, end_event)
KERNEL_FLOPS = {}       # name -> algorithmic FLOPs accumulated over the timed launches
KERNEL_BYTES = {}       # name -> algorithmic HBM bytes accumulated over the timed launches


class _timed:
    def __init__(self, name, flops=0, nbytes=0):
        self.name, self.flops, self.nbytes = name, flops, nbytes

    def __enter__(self):
        if KERNEL_TIMING:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *a):
        if KERNEL_TIMING:
            self.e.record()
            KERNEL_TIMERS.setdefault(self.name, []).append((self.s, self.e))
            KERNEL_FLOPS[self.name] = KERNEL_FLOPS.get(self.name, 0) + self.flops
            KERNEL_BYTES[self.name] = KERNEL_BYTES.get(self.name, 0) + self.nbytes


KERNEL_OWN_BYTES = {}   # name -> bytes the kernel that actually runs must move (fused kernels: not the unfused chain's)


def _own_bytes(name, nbytes):
    if KERNEL_TIMING:
        KERNEL_OWN_BYTES[name] = KERNEL_OWN_BYTES.get(name, 0) + nbytes


def kernel_time_summary():
    torch.cuda.synchronize()
    return {k: (len(v), sum(s.elapsed_time(e) for s, e in v)) for k, v in KERNEL_TIMERS.items()}


def _p(t):
    """Device pointer of a tensor, or of (tensor, float_offset)."""
    if isinstance(t, tuple):
        return ctypes.c_void_p(t[0].data_ptr() + 4 * t[1])
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32, (t.device, t.dtype)
    return t


def round_up(x, m):
    return (x + m - 1) // m * m


def rows_view(t):
    """Return (tensor, ld) of a 2-D-like fp32 tensor whose rows are 16-byte aligned.
    Accepts [..., C] with unit inner stride and uniform row stride; copies into a padded
    buffer otherwise (layout glue only)."""
    _f32(t)
    c = t.shape[-1]
    t2 = t.reshape(-1, c) if t.is_contiguous() else t
    if t2.dim() == 2 and t2.stride(1) == 1 and t2.stride(0) % 4 == 0 and t2.data_ptr() % 16 == 0 \
            and (t2.shape[0] == 1 or t2.stride(0) >= c):
        return t2, t2.stride(0)
    ld = round_up(c, 4)
    buf = torch.zeros(t.numel() // c, ld, device=t.device, dtype=torch.float32)
    buf[:, :c] = t.reshape(-1, c)
    return buf, ld


def padded(rows, cols, device, zero=True):
    """[rows, cols] view of a buffer with leading dimension round_up(cols, 4).  zero: the pad columns get a defined
    value here (consumers read whole 16-byte groups); False where the producer writes them itself."""
    ld = round_up(cols, 4)
    buf = torch.empty(rows, ld, device=device, dtype=torch.float32)
    if zero and ld != cols:   # only the pad columns (the producers write the rest)
        buf[:, cols:].zero_()
    return buf, ld


# ----------------------------------------------------------------------------- probes
def probe_xcc(nblocks=256):
    out = torch.full((nblocks,), -1, device="cuda", dtype=torch.int32)
    check(_lib.lib().tssep_probe_xcc(_p(out), nblocks, _stream()), "probe_xcc")
    return out


def probe_clock(heavy=True, nblocks=1024, iters=200000):
    """-> (shader MHz, MFMA TFLOP/s of the probe) under that load."""
    out = torch.zeros(2 * nblocks, device="cuda", dtype=torch.int64)
    check(_lib.lib().tssep_probe_clock(_p(out), nblocks, iters, int(heavy), _stream()), "probe_clock")
    o = out.cpu().view(nblocks, 2).double()
    mhz = float((o[:, 0] / o[:, 1]).mean() * 100.0)
    return mhz


def probe_mfma():
    L = _lib.lib()
    o4 = torch.zeros(64, 4, device="cuda")
    o32 = torch.zeros(2, 64, 16, device="cuda")
    check(L.tssep_probe_mfma(_p(o4), _p(o32), _stream()), "probe_mfma")
    return o4.cpu(), o32.cpu()


# ------------------------------------------------------------------------------- STFT
_TABLES = {}


def fft_tables(size, device):
    key = (size, str(device))
    if key not in _TABLES:
        L = _lib.lib()
        n = 2 * (size // 2 + size // 2 + 1)
        host = np.zeros(n, dtype=np.float32)
        check(L.tssep_fft_twiddles(size, host.ctypes.data_as(ctypes.c_void_p)), "fft_twiddles")
        _TABLES[key] = torch.from_numpy(host).to(device)
    return _TABLES[key]


def stft_frames(N, size=1024, shift=256, window_length=None, pad=True, fading=True):
    return int(_lib.lib().tssep_stft_frames(N, size, shift, window_length or size, int(pad),
                                            int(fading)))


def stft_fwd(x, window, size=1024, shift=256, fading=True, T=None):
    """x [rows, N] -> complex64 [rows, T, size//2+1].  fading: size - shift zeros in front (True) or none (False);
    T: frame count when it is not the padded, full-window one (pad=False, window_length < size: the caller's
    feature extractor computes it; frames that reach past the signal read zeros)."""
    L = _lib.lib()
    x = _f32(x).contiguous()
    rows, N = x.shape
    if T is None:
        T = stft_frames(N, size, shift, None, True, fading)
    X = torch.empty(rows, T, size // 2 + 1, 2, device=x.device, dtype=torch.float32)
    check(L.tssep_stft_fwd(_p(x), rows, N, size, shift, int(fading), _p(window),
                           _p(fft_tables(size, x.device)), _p(X), T, _stream()), "stft_fwd")
    return torch.view_as_complex(X)


def istft_fwd(X, wsyn, N, size=1024, shift=256, fading=True, tgt=None):
    """X complex64 [rows, T, F] -> y [rows, N] (+ per-chunk sums of |y - tgt| when tgt given)"""
    L = _lib.lib()
    Xr = torch.view_as_real(X.contiguous())
    rows, T = X.shape[0], X.shape[1]
    y = torch.empty(rows, N, device=X.device, dtype=torch.float32)
    part = None
    if tgt is not None:
        nch = int(L.tssep_istft_chunks(N))
        part = torch.empty(rows, nch, device=X.device, dtype=torch.float32)
        tgt = _f32(tgt).contiguous()
    check(L.tssep_istft_fwd(_p(Xr), rows, T, size, shift, int(fading), _p(wsyn),
                            _p(fft_tables(size, X.device)), _p(y), N, _p(tgt), _p(part),
                            _stream()), "istft_fwd")
    return y, part


def istft_bwd(dy, wsyn, T, size=1024, shift=256, fading=True):
    L = _lib.lib()
    dy = _f32(dy).contiguous()
    rows, N = dy.shape
    dX = torch.empty(rows, T, size // 2 + 1, 2, device=dy.device, dtype=torch.float32)
    check(L.tssep_istft_bwd(_p(dy), rows, N, size, shift, int(fading), _p(wsyn),
                            _p(fft_tables(size, dy.device)), _p(dX), T, _stream()), "istft_bwd")
    return torch.view_as_complex(dX)


def mask_istft_fwd(logit, obs, wsyn, N, size=1024, shift=256, fading=True, tgt=None):
    """logit [B,K,T,F], obs complex64 [B,T,F] -> y [B,K,N] = istft(sigmoid(logit) * obs), neither the mask
    nor the masked STFT is materialised (+ per-chunk sums of |y - tgt| when tgt [B,K,N] is given)."""
    L = _lib.lib()
    B, K, T, F = logit.shape
    logit = _f32(logit).contiguous()
    obs_r = torch.view_as_real(obs.contiguous())
    y = torch.empty(B, K, N, device=logit.device, dtype=torch.float32)
    part = None
    if tgt is not None:
        part = torch.empty(B * K, int(L.tssep_istft_chunks(N)), device=logit.device, dtype=torch.float32)
        tgt = _f32(tgt).contiguous()
    # roofline bookkeeping: the UNFUSED mask head's algorithmic bytes, whatever is moved; the fused kernel's OWN
    # bytes (logit + observation in, samples out, target in when the loss sums ride along) are kept beside them
    _own_bytes("maskhead_fwd", B * T * (4 * K * F + 8 * F) + 4 * B * K * N * (2 if tgt is not None else 1))
    with _timed("maskhead_fwd", 0, B * T * (16 * K * F + 8 * F)):
        check(L.tssep_mask_istft_fwd(_p(logit), _p(obs_r), B, K, T, size, shift, int(fading), _p(wsyn),
                                     _p(fft_tables(size, logit.device)), _p(y), N, _p(tgt), _p(part),
                                     _stream()), "mask_istft_fwd")
    return y, part


def mask_istft_bwd(dy, logit, obs, wsyn, size=1024, shift=256, fading=True, loss=None, iperm=None, bt_major=False):
    """dy [B,K,N] -> dlogit [B,K,T,F] through the iSTFT adjoint and the mask head's backward.
    loss = (est, tgt, sums or None, gout): the loss gradient is formed inside the kernel from the estimate
    and the target instead of being read from `dy` (LogMAE; sums None: MAE).  bt_major: the result is laid
    out [B*T, K*F] with speaker k at position iperm[b, k] -- what the final Linear's backward reads."""
    L = _lib.lib()
    B, K, T, F = logit.shape
    obs_r = torch.view_as_real(obs.contiguous())
    dlogit = torch.empty(B * T, K * F, device=logit.device, dtype=torch.float32) if bt_major else torch.empty_like(logit)
    _own_bytes("maskhead_bwd", B * T * (8 * K * F + 8 * F) + 4 * B * K * (dy.shape[-1] if loss is None else 2 * loss[0].shape[-1]))
    with _timed("maskhead_bwd", 0, B * T * (16 * K * F + 8 * F)):
        if loss is None and not bt_major:
            dy = _f32(dy).contiguous()
            check(L.tssep_mask_istft_bwd(_p(dy), _p(logit), _p(obs_r), B, K, dy.shape[-1], size, shift,
                                         int(fading), _p(wsyn), _p(fft_tables(size, logit.device)), _p(dlogit),
                                         T, _stream()), "mask_istft_bwd")
        else:
            if loss is None:
                x, tgt, sums, gout = _f32(dy).contiguous(), None, None, None
            else:
                x, tgt, sums, gout = (_f32(loss[0]).contiguous(), _f32(loss[1]).contiguous(), loss[2],
                                      _f32(loss[3]).contiguous())
            check(L.tssep_mask_istft_bwd_loss(_p(x), _p(tgt), _p(sums), _p(gout), _p(logit), _p(obs_r), B, K,
                                              x.shape[-1], size, shift, int(fading), _p(wsyn),
                                              _p(fft_tables(size, logit.device)), _p(iperm), int(bt_major),
                                              _p(dlogit), T, _stream()), "mask_istft_bwd_loss")
    return dlogit


# --------------------------------------------------------------------------- features
STAT_AXES = {"tf": 0, "t": 1, "f": 2}


def feat_fwd(X, fb, dct, n_mfcc, top_db=80.0, statistics_axis="tf"):
    """X complex64 [B,T,F] -> (view [B,T,n_mfcc+F], ld)."""
    L = _lib.lib()
    Xr = torch.view_as_real(X.contiguous())
    B, T, F = X.shape
    n_mels = fb.shape[1] if n_mfcc else 0
    if n_mfcc:
        fb, dct = _f32(fb).contiguous(), _f32(dct).contiguous()
    D = n_mfcc + F
    ld = round_up(D, 4)
    out = torch.empty(B, T, ld, device=X.device, dtype=torch.float32)
    if ld != D:
        out[..., D:].zero_()
    stat = STAT_AXES[statistics_axis]
    ws = torch.empty(int(L.tssep_feat_workspace_bytes(B, T, max(n_mels, 1), F, stat)) // 4 + 4,
                     device=X.device, dtype=torch.float32)
    check(L.tssep_feat_fwd(_p(Xr), B, T, F, _p(fb) if n_mfcc else None,
                           _p(dct) if n_mfcc else None, n_mels, n_mfcc, float(top_db), stat, _p(out),
                           ld, _p(ws), _stream()), "feat_fwd")
    return out[..., :D], ld


# ------------------------------------------------------------------------------- GEMM
# Arithmetic of the non-recurrent GEMMs: "bf16x3" = split-bf16 on the bf16 MFMA with fp32 accumulation (fp32-class
# accuracy: masks 3e-6 from the CPU oracle, see gemm_bf16x3.hip) -- the DEFAULT since round 5: what bench.py measures is
# what `python -m tssep_amd.exp.run_tssep` trains with; "f32" = exact fp32 MFMA (the reference's GEMM arithmetic).  The
# recurrence kernel (exact fp32 or split-bf16) is chosen separately by recurrence_kernel().
# This module reads NO environment variable: every policy attribute below has its default here and is set -- recorded
# in config.yaml / log/runtime.json -- through tssep_amd.train.runtime (`eg.runtime.*`, `bench.py --runtime k=v`).
GEMM_PRECISION = "bf16x3"
_PREC = {"f32": 0, "bf16x3": 1, "bf16": 3}      # "bf16": the plain-bf16 side line (tssep_gemm_args.precision = 3)


def _gemm_args(A, lda, B, ldb, C, ldc, M, N, K, a_kmajor=False, b_kmajor=False, bias=None, act=0,
               accumulate=False, b_kshift=0, kperiod=0, remap=None, splitk=1, split_stride=0,
               b_ones_col=False, aux=None):
    def ptr(x):
        if isinstance(x, tuple):
            return x[0].data_ptr() + 4 * x[1]
        return x.data_ptr()
    g = GemmArgs()
    g.A, g.B, g.C = ptr(A), ptr(B), ptr(C)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = lda, ldb, ldc
    g.a_kmajor, g.b_kmajor = int(a_kmajor), int(b_kmajor)
    g.b_kshift, g.kperiod = b_kshift, kperiod
    g.bias = bias.data_ptr() if bias is not None else None
    g.act, g.accumulate = act, int(accumulate)
    if remap is not None:
        g.c_remap = 2 if remap.get("narrow") else 1      # (2: the 4-byte-per-lane store, the reference of the tests)
        g.c_T, g.c_K = remap["T"], remap.get("K", 1)
        g.c_sb, g.c_sk, g.c_st = remap["sb"], remap.get("sk", 0), remap["st"]
        g.c_cm, g.c_co = remap.get("cm", 0), remap.get("co", 0)
        perm = remap.get("perm")
        g.c_perm = perm.data_ptr() if perm is not None else None
        g.c_perm_ld = remap.get("perm_ld", 0)
    g.splitk, g.c_split_stride = splitk, split_stride
    g.precision = _PREC[GEMM_PRECISION]
    g.b_ones_col = int(b_ones_col)
    if act == 2:        # tanh backward folded into the store: aux = (tensor, ld) of the tanh OUTPUT, indexed like C
        assert aux is not None and splitk <= 1
        g.aux, g.ldaux = ptr(aux[0]), aux[1]
    return g


# Kernel ids of include/tssep_hip.h (TSSEP_GEMM_*).  The library chooses by itself; tests, the shape sweep and the
# A/B tools may name kernels to be tried first: GEMM_PREFER = ("stream", "tall2") launches the first one of them that
# covers the request (asked through tssep_gemm_plan) and the library's own choice when none does.
GEMM_KERNELS = {"auto": 0, "f32": 1, "pipe": 2, "tall2": 3, "tall4": 4, "tall4_xcol": 5, "big": 6, "stream": 7,
                "nt_w160": 8, "tn": 9, "tn_tall": 10, "tn_big": 11, "tn_w160": 12, "tn_h160": 13, "big_p": 14, "big_p320": 15, "tn_p320": 16}
GEMM_KERNEL_NAMES = {v: k for k, v in GEMM_KERNELS.items()}
GEMM_PREFER = ()
GEMM_LOG = None          # a list: (kernel name, M, N, K) of every launch is appended (tests)
RECURRENCE_LOG = None    # a list: dict(kernel, direction, N, T, H, groups) of every recurrence launch (Trainer: the kernel plan)


def _log_recurrence(kernel, direction, N, T, H, groups=0):
    if RECURRENCE_LOG is not None:
        RECURRENCE_LOG.append(dict(kernel=kernel, direction=direction, sequences=N, frames=T, units=H, groups=groups))


class prefer_gemm_kernels:
    """``with prefer_gemm_kernels("big"): ...`` -- see GEMM_PREFER."""

    def __init__(self, *names):
        self.names = tuple(n for n in names if n)

    def __enter__(self):
        global GEMM_PREFER
        self.old, GEMM_PREFER = GEMM_PREFER, self.names
        return self

    def __exit__(self, *exc):
        global GEMM_PREFER
        GEMM_PREFER = self.old
        return False


def gemm_plan(g, force="auto"):
    """-> name of the kernel tssep_gemm_f32_on(g, force) would launch, or None when `force` does not cover g."""
    kid = ctypes.c_int32(0)
    rc = _lib.lib().tssep_gemm_plan(ctypes.byref(g), GEMM_KERNELS[force], ctypes.byref(kid))
    return GEMM_KERNEL_NAMES[kid.value] if rc == 0 else None


def gemm_descriptor(g):
    """The request without its pointers (what the dispatcher's choice depends on): tools/sweep_gemm_shapes.py replays it."""
    d = {f: getattr(g, f) for f, _ in GemmArgs._fields_ if f not in ("A", "B", "C", "bias", "c_perm", "aux")}
    d.update(bias=bool(g.bias), perm=bool(g.c_perm), has_aux=bool(g.aux))
    return d


def _launch_gemm(g):
    kid = 0
    for name in GEMM_PREFER:
        if gemm_plan(g, name) is not None:
            kid = GEMM_KERNELS[name]
            break
    if GEMM_LOG is not None:
        GEMM_LOG.append((gemm_plan(g, GEMM_KERNEL_NAMES[kid]), g.M, g.N, g.K, gemm_descriptor(g)))
    L = _lib.lib()
    check(L.tssep_gemm_f32_on(ctypes.byref(g), kid, _stream()) if kid else L.tssep_gemm_f32(ctypes.byref(g), _stream()),
          "gemm_f32")


def gemm(A, lda, B, ldb, C, ldc, M, N, K, **kw):
    """C = epilogue(op(A) x op(B)); see include/tssep_hip.h.  A, B, C: tensors (or (tensor,
    float_offset) tuples) whose data pointers are used as given.  Keywords: a_kmajor, b_kmajor, bias, act,
    accumulate, b_kshift, kperiod, remap, splitk, split_stride, b_ones_col, aux."""
    g = _gemm_args(A, lda, B, ldb, C, ldc, M, N, K, **kw)
    with _timed("gemm_" + GEMM_PRECISION, 2 * M * N * K, 4 * (M * K + N * K + M * N * (2 if kw.get("accumulate") else 1))):
        _launch_gemm(g)


def transposed(w, rows, cols):
    """[rows, >= cols] weight view -> contiguous [cols, round_up(rows, 4)] copy (layout glue): the
    dgrad GEMMs then read BOTH operands k-contiguous with 16-byte loads (the k-major path reads
    4 bytes per lane: 242 vs ~300 TFLOP/s on the K = 2400 shapes)."""
    ld = round_up(rows, 4)
    if ld == rows:
        return w[:rows, :cols].t().contiguous(), ld
    out = torch.zeros(cols, ld, device=w.device, dtype=torch.float32)
    out[:, :rows] = w[:rows, :cols].t()
    return out, ld


def pick_splitk(M, N, K, shifted=False, ones_col=False):
    """Split count the library recommends for the weight gradient dW[M,N] = dY[K,M]^T X[K,N] (host-only query on
    a description of the request: tools and tests; `wgrad` asks with the real arguments)."""
    g = GemmArgs()
    g.A = g.B = 0x1000
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = round_up(M, 4), round_up(N, 4), round_up(N, 4)
    g.a_kmajor = g.b_kmajor = 1
    if shifted:
        g.b_kshift, g.kperiod = -1, 253
    g.b_ones_col = int(ones_col)
    g.precision = _PREC[GEMM_PRECISION]
    S = int(_lib.lib().tssep_gemm_wgrad_splits(ctypes.byref(g)))
    if S < 1:
        check(S, "gemm_wgrad_splits")
    return S


# Opt-in arithmetic of the weight gradients (bench.py's `two_product_wgrad` side line): 2 = the dY_lo * X_hi product is
# dropped (tssep_gemm_args.precision = 2); an argument of the call, not an environment variable of the library.
WGRAD_PRODUCTS = 3


def wgrad(dY, ld_dy, X, ld_x, M, N, R, b_kshift=0, kperiod=0, with_colsum=False, splitk=None):
    """dW[M,N] = dY[R,M]^T X[R,N] by split-K partials -> (partials [S, M*N], S).
    with_colsum (split-bf16 GEMM only): partials are [S, M, round_up(N+1, 4)] and column N holds
    the column sums of dY (the bias gradient), from a virtual all-ones column of X.
    The split count is the library's (tssep_gemm_wgrad_splits: it follows the kernel the library picks)."""
    Nc = N + 1 if with_colsum else N
    ldp = round_up(Nc, 4) if with_colsum else N      # 16-byte rows keep the vector epilogue
    g = _gemm_args(dY, ld_dy, X, ld_x, X, ldp, M, Nc, R, a_kmajor=True, b_kmajor=True, b_kshift=b_kshift,
                   kperiod=kperiod, splitk=8, split_stride=M * ldp, b_ones_col=with_colsum)
    if (WGRAD_PRODUCTS == 2 and GEMM_PRECISION == "bf16x3") or GEMM_PRECISION == "bf16":
        g.precision = 2      # (the plain-bf16 side line: dY as plain bf16 in the weight gradients, X keeps hi + lo)
    S = splitk
    if not S:
        S = int(_lib.lib().tssep_gemm_wgrad_splits(ctypes.byref(g)))
        if S < 1:
            check(S, "gemm_wgrad_splits")
    dev = dY[0].device if isinstance(dY, tuple) else dY.device
    part = torch.empty(S, M * ldp, device=dev, dtype=torch.float32)
    g.C, g.splitk = part.data_ptr(), S
    with _timed("gemm_" + GEMM_PRECISION, 2 * M * Nc * R, 4 * (M * R + N * R + S * M * ldp)):
        _launch_gemm(g)
    return part, S


def fused_colsum():
    """Whether weight-gradient GEMMs can also produce the bias gradient (b_ones_col)."""
    return GEMM_PRECISION in ("bf16x3", "bf16")


def reduce_splits(part, S, count, dst, accumulate=False):
    check(_lib.lib().tssep_reduce_splits(_p(part), S, count, count, _p(dst), int(accumulate),
                                         _stream()), "reduce_splits")


def reduce_splits_bias(part, S, M, N, ldp, dw, db, accumulate=False):
    """partials [S, M, ldp] of wgrad(..., with_colsum=True) -> dw [M, N] (dense) and db [M] (column N)."""
    check(_lib.lib().tssep_reduce_splits_bias(_p(part), S, M * ldp, M, N, ldp, _p(dw), N, _p(db), int(accumulate),
                                              _stream()), "reduce_splits_bias")


def colsum(A, lda, M, N, out=None, accumulate=False):
    L = _lib.lib()
    dev = A[0].device if isinstance(A, tuple) else A.device
    if out is None:
        out = torch.empty(N, device=dev, dtype=torch.float32)
    ws = torch.empty(int(L.tssep_colsum_workspace_bytes(M, N)) // 4, device=dev,
                     dtype=torch.float32)
    a = A[0].data_ptr() + 4 * A[1] if isinstance(A, tuple) else A.data_ptr()
    check(L.tssep_colsum_f32(ctypes.c_void_p(a), M, N, lda, _p(out), int(accumulate), _p(ws),
                             _stream()), "colsum")
    return out


# ---- derived weight layouts, cached across the micro-steps of a virtual minibatch ------------------
# Packed / transposed weight copies (lstm_pack, the W-stationary packs, dgrad transposes) are functions of
# the parameters only.  With virtual_minibatch_size = v (the reference's default is 12,
# tssep/train/experiment.py:135) the parameters change once per v forward/backward passes, so the copies
# are rebuilt once per optimizer step instead of once per pass.  Validity: a parameter changes either
# through torch (its ``_version`` counter moves: load_state_dict, copy_) or through the fused Adam kernel
# (raw pointer: ``weights_changed()`` is called by the optimizer).  Never used while a stream is being
# captured: a graph must contain the pack kernels, the weights differ at every replay.
WEIGHTS_VERSION = 0


def weights_changed():
    global WEIGHTS_VERSION
    WEIGHTS_VERSION += 1


_BUILDER_SEQ = itertools.count()      # order of first use: a build may read what an earlier one made
_PREPARED = None                      # while a step is being captured: (id(params[0]), tag) -> (event, value, stream, waiters)
PREPARED_HITS = 0                     # layouts handed out from there (tests)
PREPARE_DERIVED = True                # False: every layout is built where it is first used (A/B, tools)
_DERIVED_SEEN = None                  # while `record_derived` is open: (id(params[0]), tag) of every layout asked for


class record_derived:
    """``with record_derived() as seen:`` -- the layouts a step asks `derived` for (train/graph.py records them over the
    warm-up passes of ONE input signature and prepares exactly those in its graph: a builder another signature registered --
    the 32-sequence packs of a small validation batch, say -- is not rebuilt in every replay for nobody)."""

    def __enter__(self):
        global _DERIVED_SEEN
        self.old, _DERIVED_SEEN = _DERIVED_SEEN, set()
        return _DERIVED_SEEN

    def __exit__(self, *exc):
        global _DERIVED_SEEN
        _DERIVED_SEEN = self.old
        return False


def derived(tag, params, build):
    """build() -> any structure of tensors derived from `params`; memoised until a parameter changes.
    The memo lives ON the first parameter object (it dies with the module: a recycled device address of
    another model can never hit it) and is stamped with every parameter's address and version, the global
    update counter and the stream it was built on.  A `build` must reach everything it reads THROUGH the
    parameters (or through `derived` again): `prepare_derived` calls it at the start of a later step."""
    key = (id(params[0]), tag)
    if _DERIVED_SEEN is not None:
        _DERIVED_SEEN.add(key)
    if _PREPARED is not None:
        hit = _PREPARED.get(key)
        if hit is not None:
            global PREPARED_HITS
            PREPARED_HITS += 1
            ev, val, built_on, waited = hit
            cur = torch.cuda.current_stream().cuda_stream
            if cur != built_on and cur not in waited:      # (never a stream on its own event, never twice: one edge each)
                torch.cuda.current_stream().wait_event(ev)
                waited.add(cur)
            return val
    builders = params[0].__dict__.setdefault("_tssep_builders", {})      # on the parameter, like the memo below
    if tag not in builders:
        builders[tag] = (next(_BUILDER_SEQ), build)
    if torch.cuda.is_current_stream_capturing():
        return build()
    memo = params[0].__dict__.setdefault("_tssep_derived", {})
    stamp = (WEIGHTS_VERSION, torch.cuda.current_stream().cuda_stream,
             tuple((id(p), p.data_ptr(), p._version) for p in params))
    hit = memo.get(tag)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    val = build()
    memo[tag] = (stamp, val)
    return val


class prepare_derived:
    """Context of ONE step that is being captured into a hipGraph (train/graph.py): the derived weight layouts the
    previous (warm-up) steps asked for -- gate / projection packs, the W-stationary packs, the dgrad transposes; functions
    of the parameters only -- are built at the START of the step on the side stream, as a branch of the graph beside the
    STFT / feature kernels, instead of one small kernel after the other in front of the recurrences they feed
    (8 utterances per GPU: ~25 launches of 4-20 us, 0.2 ms of a 7.8-ms step).  `derived` hands out the prepared value
    after making the consumer's stream wait for the event behind it.  Outside a capture the memo above does the job."""

    def __init__(self, parameters, device, only=None):
        """only: a set of (id(parameter), tag) -- what `record_derived` saw -- restricts the branch to those layouts."""
        self.builds = sorted((seq, id(p), tag, build) for p in parameters
                             for tag, (seq, build) in p.__dict__.get("_tssep_builders", {}).items()
                             if only is None or (id(p), tag) in only)
        self.device = device

    def __enter__(self):
        global _PREPARED
        assert _PREPARED is None
        main = torch.cuda.current_stream(self.device)
        side = side_stream(self.device)
        if side is main or not PREPARE_DERIVED or not torch.cuda.is_current_stream_capturing():
            return self
        prepared = {}
        side.wait_stream(main)
        _PREPARED = prepared            # builds that go through `derived` again find what is already there
        try:
            with torch.cuda.stream(side):
                for _, pid, tag, build in self.builds:
                    val = build()
                    ev = torch.cuda.Event()
                    ev.record(side)
                    prepared[(pid, tag)] = (ev, val, side.cuda_stream, set())
        except BaseException:
            _PREPARED = None
            raise
        return self

    def __exit__(self, *exc):
        global _PREPARED
        _PREPARED = None
        return False


# ------------------------------------------------------------------------------ BLSTM
def lstm_sizes(H, I, ld_i):
    sz = LstmSizes()
    check(_lib.lib().tssep_lstm_pack_sizes(H, I, ld_i, ctypes.byref(sz)), "lstm_pack_sizes")
    return sz


def lstm_pack(params, H, I):
    """params: 8 tensors (w_ih, w_hh, b_ih, b_hh forward, then reverse), torch layout."""
    L = _lib.lib()
    ld_i = round_up(I, 4)
    sz = lstm_sizes(H, I, ld_i)
    dev = params[0].device
    buf = torch.empty(sz.wih_p + round_up(sz.bias_p, 4) + sz.whh_f + sz.whh_b, device=dev,
                      dtype=torch.float32)
    o1 = sz.wih_p
    o2 = o1 + round_up(sz.bias_p, 4)
    o3 = o2 + sz.whh_f
    wih_p, bias_p, whh_f, whh_b = buf[:o1], buf[o1:o1 + sz.bias_p], buf[o2:o3], buf[o3:]
    ps = [_f32(p.detach()).contiguous() for p in params]
    check(L.tssep_lstm_pack(*[_p(p) for p in ps], H, I, ld_i, _p(wih_p), _p(bias_p), _p(whh_f),
                            _p(whh_b), _stream()), "lstm_pack")
    return dict(wih_p=wih_p, bias_p=bias_p, whh_f=whh_f, whh_b=whh_b, ld_i=ld_i, _keep=(buf, ps))


def blstm_fwd(gates, cell, hout, ldo, dstride, whh_f, N, T, H):
    _log_recurrence("stream_f32", "fwd", N, T, H, 0)
    with _timed("blstm_fwd", 2 * 2 * N * T * 4 * H * H):
        check(_lib.lib().tssep_blstm_fwd(_p(gates), _p(cell), _p(hout), ldo, dstride, _p(whh_f), N,
                                         T, H, _stream()), "blstm_fwd")


def blstm_bwd(gates, cell, dhout, ldo, dstride, whh_b, N, T, H):
    _log_recurrence("stream_f32", "bwd", N, T, H, 0)
    with _timed("blstm_bwd", 2 * 2 * N * T * 4 * H * H):
        check(_lib.lib().tssep_blstm_bwd(_p(gates), _p(cell), _p(dhout), ldo, dstride, _p(whh_b), N,
                                         T, H, _stream()), "blstm_bwd")


# cluster (W-stationary) recurrence ------------------------------------------------------------
_ERR = {}
RECURRENCE = "auto"    # "auto" | "stream" (lstm.hip) | "cluster" (lstm_cluster.hip) | "onchip"


def _err_flag(device):
    device = torch.device(device)
    if device.index is None:      # "cuda" and "cuda:0" must name the SAME flag (a check on the wrong key sees nothing)
        device = torch.device("cuda", torch.cuda.current_device())
    key = str(device)
    if key not in _ERR:
        # [0] error flag, [1] launch epoch of the W-stationary kernels (library-maintained), [2..3] spare
        _ERR[key] = torch.zeros(4, device=device, dtype=torch.int32)
    return _ERR[key]


def cluster_error_code(device="cuda"):
    """Synchronising read-and-clear of the cluster kernels' timeout flag: 0 = no launch gave up."""
    f = _err_flag(device)
    v = int(f[0].item())
    if v:
        f[0].zero_()
    return v


def check_cluster_errors(device="cuda"):
    """Synchronising check of the cluster kernels' timeout flag (tests / end of a bench run)."""
    v = cluster_error_code(device)
    if v:
        raise RuntimeError(f"cluster recurrence kernel timed out waiting for a peer (code {v})")


def n_cus(device):
    return torch.cuda.get_device_properties(device).multi_processor_count


_WARNED = set()


def _warn_once(key, msg):
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def onchip_max_steps(N, H, backward, device=None):
    """Longest sequence (frames) the W-stationary split-bf16 recurrence takes for this launch: the 16-sequence
    kernels address (seqs - 1) T 2H 16 bytes of gates from a work item's base with 32 bits -- 14 913 frames at
    H = 300 (7 215 for the 32-sequence kernels that serve the shapes the interleaved ones do not).  Rounds 1-3
    stopped at 2 046 (an 11-bit step field in the exchange tags; it wraps now, lstm_onchip.hip)."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    g16 = 0
    if dev is not None and H % 4 == 0:
        g16 = onchip16_bwd_groups(N, H, dev) if backward else onchip16_groups(N, H, dev)
    return int(_lib.lib().tssep_lstm_onchip_max_steps(H, 16 if g16 else 32))


def recurrence_kernel(N, H, backward, T=0, device=None):
    """-> 'stream' | 'cluster' | 'onchip' for a BLSTM over N sequences (ms per launch on MI355X at H=300,
    T=253, profiles/r2_recurrence_microbench.jsonl, forward / backward):
      on-chip bf16x3 (compact granules)  1.34 / 1.32 (8 sequences), 1.58 / 1.53 (32), 1.85 / 1.74 (256),
                                         2.26 / 2.14 (768 = one resident round of 48 XCD-local clusters x
                                         32 sequences x 1 direction), 4.3 / 4.1 (1536), 8.4 / 8.1 (3072)
      fp32 cluster                       1.50 / 1.34 (8), 1.50 / 1.57 (32), 7.8 / 21.5 (768)
      streaming fp32                     4.75 / 6.0 (<= 512), 5.3 / 6.2 (768), 17.0 / 18.7 (3072)
    Policy: the on-chip kernels for every N where they exist (H >= 128, T up to `onchip_max_steps`: 14 913 frames
    = 238 s at H = 300); round 1 sent backward launches of <= 32 sequences to the fp32 cluster kernel, which the
    compact granules overtook.  The fp32 W-stationary (cluster) and streaming kernels remain selectable
    (RECURRENCE) as the exact-fp32 recurrences, and streaming is the path for H the W-stationary kernels do not
    support -- a fallback for a SUPPORTED H (sequence too long) is announced, never silent."""
    L = _lib.lib()
    h_ok = bool(L.tssep_lstm_onchip_supported(H))
    onchip_ok = h_ok and (T <= 2046 or T <= onchip_max_steps(N, H, backward, device))
    if h_ok and not onchip_ok and RECURRENCE in ("auto", "onchip") and H >= 128:
        _warn_once(("onchip_T", H, backward), f"tssep_amd: {T} frames per sequence exceed the W-stationary recurrence's "
                   f"{onchip_max_steps(N, H, backward, device)} at H = {H}: the streaming fp32 kernel runs instead (about 3x slower)")
    if RECURRENCE in ("stream", "cluster", "onchip"):
        ok = {"stream": True, "cluster": bool(L.tssep_lstm_cluster_supported(H)), "onchip": onchip_ok}[RECURRENCE]
        return RECURRENCE if ok else "stream"
    if H < 128:
        return "stream"
    if onchip_ok:
        return "onchip"
    return "stream"


def lstm_pack_cluster(w_hh_f, w_hh_r, H):
    L = _lib.lib()
    nf, nb = int(L.tssep_lstm_cluster_pack_floats(H, 0)), int(L.tssep_lstm_cluster_pack_floats(H, 1))
    buf = torch.empty(nf + nb, device=w_hh_f.device, dtype=torch.float32)
    a, b = _f32(w_hh_f.detach()).contiguous(), _f32(w_hh_r.detach()).contiguous()
    check(L.tssep_lstm_pack_cluster(_p(a), _p(b), H, _p(buf[:nf]), _p(buf[nf:]), _stream()),
          "lstm_pack_cluster")
    return buf[:nf], buf[nf:]


def blstm_cluster_fwd(gates, cell, hout, ldo, dstride, whh_cf, N, T, H, ms=2):
    _log_recurrence("cluster_f32", "fwd", N, T, H, 0)
    fence_comm(gates.device)
    L = _lib.lib()
    cus = n_cus(gates.device)
    xbuf = torch.empty(int(L.tssep_lstm_cluster_xbuf_bytes(N, H, 0, cus, ms)) // 8 + 1,
                       device=gates.device, dtype=torch.int64)
    with _timed("blstm_cluster_fwd", 2 * 2 * N * T * 4 * H * H):
        check(L.tssep_blstm_cluster_fwd(_p(gates), _p(cell), _p(hout), ldo, dstride, _p(whh_cf),
                                        _p(xbuf), _p(_err_flag(gates.device)), N, T, H, cus, ms,
                                        _stream()), "blstm_cluster_fwd")


def blstm_cluster_bwd(gates, cell, dhout, ldo, dstride, whh_cb, N, T, H, ms=2):
    _log_recurrence("cluster_f32", "bwd", N, T, H, 0)
    fence_comm(gates.device)
    L = _lib.lib()
    cus = n_cus(gates.device)
    xbuf = torch.empty(int(L.tssep_lstm_cluster_xbuf_bytes(N, H, 1, cus, ms)) // 8 + 1,
                       device=gates.device, dtype=torch.int64)
    with _timed("blstm_cluster_bwd", 2 * 2 * N * T * 4 * H * H):
        check(L.tssep_blstm_cluster_bwd(_p(gates), _p(cell), _p(dhout), ldo, dstride, _p(whh_cb),
                                        _p(xbuf), _p(_err_flag(gates.device)), N, T, H, cus, ms,
                                        _stream()), "blstm_cluster_bwd")


# on-chip-weights recurrence (bf16x3 MFMA, lstm_onchip.hip) -----------------------------------
def lstm_pack_onchip(w_hh_f, w_hh_r, H):
    L = _lib.lib()
    nf, nb = int(L.tssep_lstm_onchip_pack_floats(H, 0)), int(L.tssep_lstm_onchip_pack_floats(H, 1))
    buf = torch.empty(nf + nb, device=w_hh_f.device, dtype=torch.float32)
    a, b = _f32(w_hh_f.detach()).contiguous(), _f32(w_hh_r.detach()).contiguous()
    check(L.tssep_lstm_pack_onchip(_p(a), _p(b), H, _p(buf[:nf]), _p(buf[nf:]), _stream()),
          "lstm_pack_onchip")
    return buf[:nf], buf[nf:]


def blstm_onchip_fwd(gates, cell, hout, ldo, dstride, wf, N, T, H, layout=0):
    _log_recurrence("onchip32_bf16x3", "fwd", N, T, H, 0)
    fence_comm(gates.device)
    L = _lib.lib()
    cus = n_cus(gates.device)
    xbuf = torch.empty(int(L.tssep_lstm_onchip_xbuf_bytes(N, H, 0)) // 8 + 2, device=gates.device,
                       dtype=torch.int64)
    with _timed("blstm_onchip_fwd", 2 * 2 * N * T * 4 * H * H, N * T * 2 * H * 40):      # gates in / activations out 16 + 16, c 4, h 4 B per cell
        check(L.tssep_blstm_onchip_fwd(_p(gates), _p(cell), _p(hout), ldo, dstride, _p(wf), _p(xbuf),
                                       _p(_err_flag(gates.device)), N, T, H, cus, layout,
                                       _stream()), "blstm_onchip_fwd")


# interleaved forward (16-sequence groups in rotation); policy: runtime.onchip16 / onchip16_groups / onchip16_min_n
ONCHIP16 = True
ONCHIP16_GROUPS = 0      # 1 | 2 | 4: forced group count where it divides the number of 16-sequence groups (A/B)
ONCHIP16_MIN_N = 1       # (a one-group step is 3.5 us against 5.2: the latency regime gains too)
ONCHIP16_BWD = True
ONCHIP16_BWD_GROUPS = 2  # at most: a backward phase is paced by its three barriers and the publish (see onchip16_bwd_groups)


KEEP_XBUF = None         # a list: the exchange buffers of the interleaved forward launches are appended (trace builds, tools)


def _w16(name, restype, argtypes):
    """Entry point of the four-wave variant of the interleaved forward: EXPERIMENT build only since ABI 4 (rejected in
    round 5: 1.4-1.8x slower at every size) -- `make -C tssep_amd/csrc exp`, TSSEP_HIP_LIB=.../libtssep_hip_exp.so."""
    L = _lib.lib()
    if not hasattr(L, name):
        raise RuntimeError(f"{name}: the four-wave workgroups exist in the experiment build only (make exp, TSSEP_HIP_LIB)")
    fn = getattr(L, name)
    fn.restype, fn.argtypes = restype, argtypes
    return fn


_VP, _I64, _I = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def onchip16_groups(N, H, device, waves=8):
    if not ONCHIP16:
        return 0
    forced = ONCHIP16_GROUPS
    if waves == 8:
        g = int(_lib.lib().tssep_blstm_onchip16_groups(N, H, n_cus(device)))
    else:
        g = int(_w16("tssep_blstm_onchip16w_groups", _I, [_I64, _I, _I, _I])(N, H, n_cus(device), waves))
    if g and forced in (1, 2, 4) and ((N + 15) // 16) % forced == 0 and not (waves == 4 and forced == 4):
        return forced
    return g if N >= ONCHIP16_MIN_N else 0


def lstm_pack_onchip16(w_hh_f, w_hh_r, H, waves=8):
    L = _lib.lib()
    a, b = _f32(w_hh_f.detach()).contiguous(), _f32(w_hh_r.detach()).contiguous()
    if waves == 8:
        buf = torch.empty(int(L.tssep_lstm_onchip16_pack_floats(H)), device=w_hh_f.device, dtype=torch.float32)
        check(L.tssep_lstm_pack_onchip16(_p(a), _p(b), H, _p(buf), _stream()), "lstm_pack_onchip16")
        return buf
    buf = torch.empty(int(_w16("tssep_lstm_onchip16w_pack_floats", _I64, [_I, _I])(H, waves)), device=w_hh_f.device, dtype=torch.float32)
    check(_w16("tssep_lstm_pack_onchip16w", _I, [_VP, _VP, _I, _I, _VP, _VP])(_p(a), _p(b), H, waves, _p(buf), _stream()),
          "lstm_pack_onchip16")
    return buf


def blstm_onchip16_fwd(gates, cell, hout, ldo, dstride, wf16, N, T, H, groups, layout=0, waves=8):
    """wf16: the pack made for the SAME `waves` (lstm_pack_onchip16); waves = 4: experiment build only (`_w16`)."""
    _log_recurrence("onchip16_bf16x3" + ("_w4" if waves == 4 else ""), "fwd", N, T, H, groups)
    fence_comm(gates.device)
    L = _lib.lib()
    cus = n_cus(gates.device)
    nbytes = int(L.tssep_lstm_onchip16_xbuf_bytes(N, H)) if waves == 8 else \
        int(_w16("tssep_lstm_onchip16w_xbuf_bytes", _I64, [_I64, _I, _I])(N, H, waves))
    xbuf = torch.empty(nbytes // 8 + 2, device=gates.device, dtype=torch.int64)
    if KEEP_XBUF is not None:
        KEEP_XBUF.append(xbuf)
    with _timed("blstm_onchip_fwd", 2 * 2 * N * T * 4 * H * H, N * T * 2 * H * 40):      # gates in / activations out 16 + 16, c 4, h 4 B per cell
        if waves == 8:
            rc = L.tssep_blstm_onchip16_fwd(_p(gates), _p(cell), _p(hout), ldo, dstride, _p(wf16), _p(xbuf),
                                            _p(_err_flag(gates.device)), N, T, H, cus, layout, groups, _stream())
        else:
            rc = _w16("tssep_blstm_onchip16w_fwd", _I, [_VP, _VP, _VP, _I64, _I64, _VP, _VP, _VP, _I64, _I64, _I, _I, _I, _I, _I, _VP])(
                _p(gates), _p(cell), _p(hout), ldo, dstride, _p(wf16), _p(xbuf), _p(_err_flag(gates.device)), N, T, H, cus,
                layout, groups, waves, _stream())
        check(rc, "blstm_onchip16_fwd")


def onchip16_bwd_groups(N, H, device):
    """group count of the interleaved backward (0: use the 32-sequence kernel; policy runtime.onchip16_bwd)"""
    if not ONCHIP16_BWD or not (256 < H <= 320):
        return 0
    # (two groups at most: a backward phase is paced by its three barriers and the publish, 7.1 against 7.8 ms per launch
    # at 3 072 sequences with four -- profiles/r3_onchip16_backward.jsonl)
    return min(onchip16_groups(N, H, device), ONCHIP16_BWD_GROUPS)


def lstm_pack_onchip16_bwd(w_hh_f, w_hh_r, H):
    L = _lib.lib()
    buf = torch.empty(int(L.tssep_lstm_onchip16_bwd_pack_floats(H)), device=w_hh_f.device, dtype=torch.float32)
    a, b = _f32(w_hh_f.detach()).contiguous(), _f32(w_hh_r.detach()).contiguous()
    check(L.tssep_lstm_pack_onchip16_bwd(_p(a), _p(b), H, _p(buf), _stream()), "lstm_pack_onchip16_bwd")
    return buf


def blstm_onchip16_bwd(gates, cell, dhout, ldo, dstride, wb16, N, T, H, groups, layout=0):
    _log_recurrence("onchip16_bf16x3", "bwd", N, T, H, groups)
    fence_comm(gates.device)
    L = _lib.lib()
    cus = n_cus(gates.device)
    xbuf = torch.empty(int(L.tssep_lstm_onchip16_bwd_xbuf_bytes(N, H)) // 8 + 2, device=gates.device, dtype=torch.int64)
    if KEEP_XBUF is not None:
        KEEP_XBUF.append(xbuf)
    with _timed("blstm_onchip_bwd", 2 * 2 * N * T * 4 * H * H, N * T * 2 * H * 44):      # activations 16, c 4 + 4, dh 4 in, d(gates) 16 out
        check(L.tssep_blstm_onchip16_bwd(_p(gates), _p(cell), _p(dhout), ldo, dstride, _p(wb16), _p(xbuf),
                                         _p(_err_flag(gates.device)), N, T, H, cus, layout, groups, _stream()),
              "blstm_onchip16_bwd")


def blstm_onchip_bwd(gates, cell, dhout, ldo, dstride, wb, N, T, H, layout=0):
    _log_recurrence("onchip32_bf16x3", "bwd", N, T, H, 0)
    fence_comm(gates.device)
    L = _lib.lib()
    cus = n_cus(gates.device)
    xbuf = torch.empty(int(L.tssep_lstm_onchip_xbuf_bytes(N, H, 1)) // 8 + 2, device=gates.device,
                       dtype=torch.int64)
    with _timed("blstm_onchip_bwd", 2 * 2 * N * T * 4 * H * H, N * T * 2 * H * 44):      # activations 16, c 4 + 4, dh 4 in, d(gates) 16 out
        check(L.tssep_blstm_onchip_bwd(_p(gates), _p(cell), _p(dhout), ldo, dstride, _p(wb), _p(xbuf),
                                       _p(_err_flag(gates.device)), N, T, H, cus, layout,
                                       _stream()), "blstm_onchip_bwd")


def lstm_unpack(src, ld, nsplit, split_stride, H, ncols, dst_f, dst_r, accumulate=False):
    check(_lib.lib().tssep_lstm_unpack(_p(src), ld, nsplit, split_stride, H, ncols, _p(dst_f),
                                       _p(dst_r), int(accumulate), _stream()), "lstm_unpack")


# ---- side stream for weight gradients ---------------------------------------------------------
# Weight-gradient GEMMs are off the critical path of backward (nothing downstream reads them):
# with a GradBucket attached they are accumulated straight into the flat gradient buffer on a
# second HIP stream, overlapping the T-sequential recurrences of the layers still to come.
_SIDE = {}
OVERLAP_WGRAD = True
FOLD_TANH = True         # Tanh backward inside the consumer's d(input) GEMM store
FOLD_TAIL = 1            # LogMAE / MAE backward and the logit un-map inside the fused tail's backward (0: off; 2: the loss only, 3: the un-map only -- experiments)
SIDE_STREAM = True
SIDE_STREAM_MAX_SEQS = 512   # layers with more sequences (x 253 frames) stay on the compute stream: beyond this the GEMMs of the two streams only slow each other
ACTIVE_SINK = 0          # which gradient bucket the running micro-batch accumulates into
GRAPH_STEP = "auto"      # Trainer: hipGraph replay of forward + loss + backward ("auto": batches of <= GRAPH_MAX_UTTERANCES utterances)
GRAPH_MAX_UTTERANCES = 32


BUCKETED_ALLREDUCE = False     # per-layer gradient segments reduced during the backward (distributed.GradBucket; default off)
COMM_PENDING = {}              # str(device) -> the communication stream with gradient reductions in flight


def fence_comm(device):
    """In front of every W-stationary recurrence launch: an RCCL kernel must not run beside it (include/tssep_hip.h:
    the clusters need all their workgroups resident) -- the compute stream waits for the reductions queued so far."""
    st = COMM_PENDING.pop(str(torch.device(device) if not isinstance(device, torch.device) else device), None)
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)


def side_stream(device, rows=0):
    """The weight-gradient stream paired with the CURRENT stream (one per compute stream).  Measured
    (batch sweep, MI355X): +7 % at batch 64, +2 % at 192, 0 at 384 -- where the uncontended GEMMs
    run at 213 instead of 124 TFLOP/s --, -0.6 % for the 768-sequence layers of batch 768, so layers
    with more than 512 sequences stay on the compute stream."""
    cur = torch.cuda.current_stream(device)
    if not SIDE_STREAM or rows > SIDE_STREAM_MAX_SEQS * 253:
        return cur
    key = (str(device), cur.cuda_stream)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def join_side_stream(device=None):
    """Make the current stream wait for the gradient work queued on ITS side stream (the pairing is per
    compute stream, so a stream that is being captured into a hipGraph only ever joins the stream it
    forked itself)."""
    cur = torch.cuda.current_stream(device)
    st = _SIDE.get((str(cur.device), cur.cuda_stream))
    if st is not None:
        cur.wait_stream(st)


# ------------------------------------------------------------------------ elementwise
def tanh_bwd(dy, y, rows, P, K, T, combined):
    dz = torch.empty(rows, P, device=dy.device, dtype=torch.float32)
    check(_lib.lib().tssep_tanh_bwd(_p(dy), _p(y), _p(dz), rows, P, K, T, int(combined),
                                    _stream()), "tanh_bwd")
    return dz


def cond_fwd(pre, ld_pre, aux, B, K, T, F, trials, combination):
    """-> (xs view [B*trials*K*T, W], ld)"""
    L = _lib.lib()
    aux2, ld_aux = rows_view(aux)
    E = aux.shape[-1]
    W = F if combination == "mul" else F + E
    # (the 16-byte path of cond_mul_fwd_kernel writes the pad columns itself: products of the inputs' zero pads)
    writes_pads = combination == "mul" and F <= 1024 and ld_pre % 4 == 0 and ld_pre >= round_up(F, 4) \
        and pre.data_ptr() % 16 == 0 and aux2.data_ptr() % 16 == 0
    xs, ld = padded(B * trials * K * T, W, pre.device, zero=not writes_pads)
    if combination == "mul":
        assert E == F, (E, F)
        check(L.tssep_cond_mul_fwd(_p(pre), ld_pre, _p(aux2), ld_aux, _p(xs), ld, B, K, T, F,
                                   trials, _stream()), "cond_mul_fwd")
    else:
        check(L.tssep_cond_cat_fwd(_p(pre), ld_pre, _p(aux2), ld_aux, _p(xs), ld, B, K, T, F, E,
                                   trials, _stream()), "cond_cat_fwd")
    return xs, ld, (aux2, ld_aux)


def cond_bwd(dxs, ld_dxs, auxinfo, B, K, T, F, trials, combination):
    L = _lib.lib()
    aux2, ld_aux = auxinfo
    # (cond_mul_bwd_v4_kernel sums the pad columns too: zero in both inputs)
    writes_pads = combination == "mul" and ld_dxs % 4 == 0 and ld_dxs >= round_up(F, 4) and dxs.data_ptr() % 16 == 0 \
        and aux2.data_ptr() % 16 == 0
    dpre, ld = padded(B * T, F, dxs.device, zero=not writes_pads)
    if combination == "mul":
        check(L.tssep_cond_mul_bwd(_p(dxs), ld_dxs, _p(aux2), ld_aux, _p(dpre), ld, B, K, T, F,
                                   trials, _stream()), "cond_mul_bwd")
    else:
        check(L.tssep_cond_cat_bwd(_p(dxs), ld_dxs, _p(dpre), ld, B, K, T, F, trials, _stream()),
              "cond_cat_bwd")
    return dpre, ld


# -------------------------------------------------------------------------- mask head
def maskhead_fwd(logit, obs):
    """logit [B,K,T,F] fp32, obs complex64 [B,T,F] -> mask [B,K,T,F], est complex64 [B,K,T,F]"""
    B, K, T, F = logit.shape
    logit = _f32(logit).contiguous()
    obs_r = torch.view_as_real(obs.contiguous())
    mask = torch.empty_like(logit)
    est = torch.empty(B, K, T, F, 2, device=logit.device, dtype=torch.float32)
    with _timed("maskhead_fwd", 0, B * T * (16 * K * F + 8 * F)):
        check(_lib.lib().tssep_maskhead_fwd(_p(logit), _p(obs_r), _p(mask), _p(est), B, K, T, F,
                                            _stream()), "maskhead_fwd")
    return mask, torch.view_as_complex(est)


def maskhead_bwd(dest, dmask, mask, obs):
    B, K, T, F = mask.shape
    dest_r = torch.view_as_real(dest.contiguous())
    obs_r = torch.view_as_real(obs.contiguous())
    dlogit = torch.empty_like(mask)
    if dmask is not None:
        dmask = _f32(dmask).contiguous()
    with _timed("maskhead_bwd", 0, B * T * (16 * K * F + 8 * F)):
        check(_lib.lib().tssep_maskhead_bwd(_p(dest_r), _p(dmask), _p(mask), _p(obs_r), _p(dlogit),
                                            B, K, T, F, _stream()), "maskhead_bwd")
    return dlogit


def mask_mul_fwd(mask, obs):
    B, K, T, F = mask.shape
    est = torch.empty(B, K, T, F, 2, device=mask.device, dtype=torch.float32)
    check(_lib.lib().tssep_mask_mul_fwd(_p(_f32(mask).contiguous()),
                                        _p(torch.view_as_real(obs.contiguous())), _p(est), B, K, T,
                                        F, _stream()), "mask_mul_fwd")
    return torch.view_as_complex(est)


def mask_mul_bwd(dest, obs):
    B, K, T, F = dest.shape
    dmask = torch.empty(B, K, T, F, device=dest.device, dtype=torch.float32)
    check(_lib.lib().tssep_mask_mul_bwd(_p(torch.view_as_real(dest.contiguous())),
                                        _p(torch.view_as_real(obs.contiguous())), _p(dmask), B, K,
                                        T, F, _stream()), "mask_mul_bwd")
    return dmask


# ----------------------------------------------------------------------------- beamformer
def mvdr_souden(masks, obs, reference_channel, eps=None, masking=False, masking_eps=0.0,
                check_singular=True):
    """TorchBF('mvdr_souden'), tssep/train/enhancer.py:215-265.
    masks [B,K,M,T,F] fp32|fp64 (M = 1|2), obs [B,D,T,F] complex128 -> enh [B,K,T,F] complex128.
    eps None = torch.finfo(float64).tiny, as the reference.  Raises torch.linalg.LinAlgError for a
    singular interference PSD like torch.linalg.solve does (one host sync; check_singular=False
    skips it)."""
    L = _lib.lib()
    assert masks.is_cuda and obs.is_cuda, (masks.device, obs.device)
    assert obs.dtype == torch.complex128, obs.dtype
    if masks.dtype not in (torch.float32, torch.float64):
        masks = masks.to(torch.float64)
    B, K, M, T, F = masks.shape
    Bo, D, To, Fo = obs.shape
    assert (B, T, F) == (Bo, To, Fo), (masks.shape, obs.shape)
    if M not in (1, 2):
        raise ValueError(masks.shape)
    masks = masks.contiguous()
    obs_r = torch.view_as_real(obs.contiguous())
    ws_bytes = L.tssep_mvdr_workspace_bytes(B, K, D, T, F)
    if ws_bytes <= 0:
        raise RuntimeError(f"mvdr_souden: unsupported shape masks {tuple(masks.shape)} "
                           f"obs {tuple(obs.shape)} (at most 8 channels)")
    ws = torch.empty(ws_bytes // 8 + 2, device=obs.device, dtype=torch.float64)
    info = torch.empty(1, device=obs.device, dtype=torch.int32)
    enh = torch.empty(B, K, T, F, 2, device=obs.device, dtype=torch.float64)
    eps = float(torch.finfo(torch.float64).tiny) if eps is None else float(eps)
    # torch.clamp(mask, min=masking_eps) compares in the mask's dtype (enhancer.py:261-263)
    masking_eps = float(torch.tensor(float(masking_eps), dtype=masks.dtype))
    msz = masks.element_size()
    nbytes = B * T * F * (32 * D + K * M * msz + 16 * K + (K * msz if masking else 0))
    with _timed("mvdr_souden", 0, nbytes):
        check(L.tssep_mvdr_souden_fwd(_p(obs_r), _p(masks), int(masks.dtype == torch.float64),
                                      _p(enh), _p(ws), _p(info), B, K, M, D, T, F,
                                      int(reference_channel), eps, int(bool(masking)),
                                      float(masking_eps), _stream()), "mvdr_souden")
    if check_singular:
        n = int(info.item())
        if n:
            raise torch.linalg.LinAlgError(
                f"mvdr_souden: the solver failed because the interference PSD matrix is singular "
                f"({n} of {B * K * F} (batch, speaker, frequency) systems)")
    return torch.view_as_complex(enh)


# ----------------------------------------------------------------------------- losses
def logmae_fwd(est, tgt):
    L = _lib.lib()
    B, K, N = est.shape
    est, tgt = _f32(est).contiguous(), _f32(tgt).contiguous()
    loss = torch.empty(B, device=est.device, dtype=torch.float32)
    sums = torch.empty(B, device=est.device, dtype=torch.float32)
    ws = torch.empty(int(L.tssep_logmae_workspace_bytes(B, K, N)) // 4, device=est.device,
                     dtype=torch.float32)
    check(L.tssep_logmae_fwd(_p(est), _p(tgt), B, K, N, _p(loss), _p(sums), _p(ws), _stream()),
          "logmae_fwd")
    return loss, sums


def logmae_finalize(part, B, K, N):
    L = _lib.lib()
    loss = torch.empty(B, device=part.device, dtype=torch.float32)
    sums = torch.empty(B, device=part.device, dtype=torch.float32)
    check(L.tssep_logmae_finalize(_p(part), B, K, part.shape[-1], N, _p(loss), _p(sums),
                                  _stream()), "logmae_finalize")
    return loss, sums


def logmae_bwd(est, tgt, sums, gout):
    B, K, N = est.shape
    dest = torch.empty_like(est)
    check(_lib.lib().tssep_logmae_bwd(_p(est), _p(tgt), _p(sums), _p(_f32(gout).contiguous()), B,
                                      K, N, _p(dest), _stream()), "logmae_bwd")
    return dest


def vadbce_fwd(logit, vad):
    L = _lib.lib()
    B, K, T, F = logit.shape
    logit, vad = _f32(logit).contiguous(), _f32(vad).contiguous()
    loss = torch.empty(B, device=logit.device, dtype=torch.float32)
    xmean = torch.empty(B, K, T, device=logit.device, dtype=torch.float32)
    ws = torch.empty(B * K * T, device=logit.device, dtype=torch.float32)
    check(L.tssep_vadbce_fwd(_p(logit), _p(vad), B, K, T, F, _p(loss), _p(xmean), _p(ws),
                             _stream()), "vadbce_fwd")
    return loss, xmean


def vadbce_bwd(xmean, vad, gout, F):
    B, K, T = xmean.shape
    dlogit = torch.empty(B, K, T, F, device=xmean.device, dtype=torch.float32)
    check(_lib.lib().tssep_vadbce_bwd(_p(xmean), _p(_f32(vad).contiguous()),
                                      _p(_f32(gout).contiguous()), B, K, T, F, _p(dlogit),
                                      _stream()), "vadbce_bwd")
    return dlogit


def logit_map_fwd(raw, perm, iperm, B, trials, K, T, F, Fr, spk_rows):
    out = torch.empty(B, K, T, F, device=raw.device, dtype=torch.float32)
    check(_lib.lib().tssep_logit_map_fwd(_p(raw), _p(perm), _p(iperm), B, trials, K, T, F, Fr,
                                         int(spk_rows), _p(out), _stream()), "logit_map_fwd")
    return out


def logit_map_bwd(dout, perm, iperm, B, trials, K, T, F, Fr, spk_rows):
    n = B * trials * K * T * Fr
    draw = torch.empty(n, device=dout.device, dtype=torch.float32)
    check(_lib.lib().tssep_logit_map_bwd(_p(_f32(dout).contiguous()), _p(perm), _p(iperm), B,
                                         trials, K, T, F, Fr, int(spk_rows), _p(draw), _stream()),
          "logit_map_bwd")
    return draw
