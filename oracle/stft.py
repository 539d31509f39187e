"""STFT / iSTFT with paderbox semantics (oracle, CPU).

The reference calls ``fe.stft`` / ``fe.istft`` (tssep/train/model.py:503-504,
661-664), which come from padertorch.contrib.cb.feature_extractor.STFT on top of
paderbox.transform.module_stft (padertorch==0.0.1 / paderbox==0.0.8,
requirements.txt:16-17) -- neither is in /root/reference.  This restates the
published algorithm; it is pinned by the reference's own doctest numbers
(tssep/train/feature_extractor.py:197-202, tssep/train/model.py:480,563-575).
"""
import math

import numpy as np
import torch
from scipy.signal import get_window


def fade_widths(window_length, shift, fading):
    """(zeros in front, zeros behind) of paderbox stft's ``fading``: True / 'full' = window_length - shift on both
    sides; 'half' = half of that in front, the other half (the odd sample) behind; None / False = none.
    ('half' is restated from the published paderbox 0.0.8 and pinned only through the frame counts of
    oracle/stft_vad.py -- no reference test transforms with it: parity unpinned for that option.)"""
    if fading in (None, False):
        return 0, 0
    p = window_length - shift
    if fading == "half":
        return p // 2, p - p // 2
    return p, p


def num_frames(num_samples, size=1024, shift=256, window_length=None,
               pad=True, fading=True):
    """Frame count (tssep/train/model.py:480: 80000 -> 316)."""
    window_length = size if window_length is None else window_length
    n = num_samples + sum(fade_widths(window_length, shift, fading))
    if pad:
        return max(int(math.ceil((n - window_length) / shift)), 0) + 1
    return (n - window_length) // shift + 1


def analysis_window(window, window_length):
    """Periodic window, scipy ``fftbins=True`` (paderbox stft default)."""
    return get_window(window, window_length, fftbins=True)


def synthesis_window(window, window_length, shift):
    """Biorthogonal synthesis window: w / sum_i shift_i(w^2)."""
    w = analysis_window(window, window_length).astype(np.float64)
    denom = np.zeros(window_length, dtype=np.float64)
    w2 = w ** 2
    for i in range(-(window_length // shift) - 1, window_length // shift + 2):
        off = i * shift
        lo, hi = max(0, off), min(window_length, window_length + off)
        if lo < hi:
            denom[lo:hi] += w2[lo - off:hi - off]
    return w / denom


def stft(x, size=1024, shift=256, window="hann", window_length=None,
         pad=True, fading=True):
    """x[..., N] real -> X[..., T, size//2+1] complex.  No scaling."""
    is_np = isinstance(x, np.ndarray)
    xt = torch.as_tensor(x)
    window_length = size if window_length is None else window_length
    lead, tail = fade_widths(window_length, shift, fading)
    if lead or tail:
        xt = torch.nn.functional.pad(xt, (lead, tail))
    n = xt.shape[-1]
    if pad:
        frames = max(int(math.ceil((n - window_length) / shift)), 0) + 1
        need = (frames - 1) * shift + window_length
        if need > n:
            xt = torch.nn.functional.pad(xt, (0, need - n))
    else:
        frames = (n - window_length) // shift + 1
    seg = xt.unfold(-1, window_length, shift)[..., :frames, :]
    w = torch.as_tensor(analysis_window(window, window_length), dtype=xt.dtype)
    X = torch.fft.rfft(seg * w, n=size, dim=-1)
    return X.numpy() if is_np else X


def istft(X, size=1024, shift=256, window="hann", window_length=None,
          fading=True, num_samples=None):
    """X[..., T, F] complex -> x[..., N] real (inverse of :func:`stft`)."""
    is_np = isinstance(X, np.ndarray)
    Xt = torch.as_tensor(X)
    window_length = size if window_length is None else window_length
    rdtype = torch.float64 if Xt.dtype == torch.complex128 else torch.float32
    seg = torch.fft.irfft(Xt, n=size, dim=-1)[..., :window_length]
    wsyn = torch.as_tensor(synthesis_window(window, window_length, shift),
                           dtype=rdtype)
    seg = seg * wsyn
    T = seg.shape[-2]
    length = (T - 1) * shift + window_length
    lead = seg.shape[:-2]
    # overlap-add: fold is the exact adjoint of unfold
    cols = seg.reshape(-1, T, window_length).transpose(1, 2)
    out = torch.nn.functional.fold(
        cols, output_size=(1, length), kernel_size=(1, window_length),
        stride=(1, shift)).reshape(*lead, length)
    lead, tail = fade_widths(window_length, shift, fading)
    out = out[..., lead:length - tail]
    if num_samples is not None:
        out = out[..., :num_samples]
    return out.numpy() if is_np else out
