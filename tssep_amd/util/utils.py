"""Sample-level VAD -> STFT-frame VAD (tssep/util/utils.py:11-77).

The reference delegates the frame bookkeeping to paderbox (absent; parity unpinned):
a frame is active when any sample under its window is active, with the same fading / padding
as the STFT.  Host-side numpy (target preparation, not on the GPU hot path)."""
import numpy as np
import torch


def stft_vad(vad, window_length, shift, fading=True):
    is_torch = isinstance(vad, torch.Tensor)
    dev = vad.device if is_torch else None
    v = vad.detach().cpu().numpy() if is_torch else np.asarray(vad)
    v = v.astype(bool)
    n = v.shape[-1]
    pad = window_length - shift if fading else 0
    total = n + 2 * pad
    frames = max(int(np.ceil((total - window_length) / shift)), 0) + 1
    need = (frames - 1) * shift + window_length
    vp = np.zeros(v.shape[:-1] + (need,), dtype=bool)
    vp[..., pad:pad + n] = v
    c = np.concatenate([np.zeros(v.shape[:-1] + (1,), dtype=np.int64), np.cumsum(vp, -1)], -1)
    starts = np.arange(frames) * shift
    out = (c[..., starts + window_length] - c[..., starts]) > 0
    out = out.astype(np.float32)
    return torch.as_tensor(out, device=dev) if is_torch else out
