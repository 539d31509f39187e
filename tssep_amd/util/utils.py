"""Sample-level VAD -> STFT-frame VAD (tssep/util/utils.py:11-77).

Reference structure: for every row, the runs of active samples (``ArrayInterval.normalized_intervals``,
utils.py:45-47) have their start and (exclusive) end mapped through paderbox's
``sample_index_to_stft_frame_index`` (utils.py:53-64) and the frames ``[f(start), f(end))`` are set
(utils.py:66-67); the number of frames is ``_samples_to_stft_frames(N, window_length, shift, pad=True,
fading)`` (utils.py:34-42).  paderbox 0.0.8 is absent here: both helpers are restated from the published
package (oracle/stft_vad.py states them as plain loops; this file is the vectorised product version,
tests/test_host_logic.py compares the two).  Host-side numpy -- target preparation, not on the GPU hot
path (the reference bypasses it when ``ex['Vad']`` is present, loss.py:134).
"""
import math

import numpy as np
import torch


def samples_to_stft_frames(samples, size, shift, pad=True, fading=True):
    """paderbox.transform.module_stft._samples_to_stft_frames: frames of an STFT over `samples` samples
    (window `size`, hop `shift`); fading pads ``size - shift`` samples on both sides ('half': one side's
    worth in total)."""
    if fading not in (None, False):
        samples = samples + (1 if fading == "half" else 2) * (size - shift)
    frames = (samples - size + shift) / shift
    return math.ceil(frames) if pad else math.floor(frames)


def sample_index_to_stft_frame_index(sample, window_length, shift, fading=True):
    """paderbox.transform.module_stft.sample_index_to_stft_frame_index: the frame whose centre is
    nearest to the sample (0 for samples before the first centre); scalar or array."""
    if fading not in (None, False):
        pad_width = window_length - shift
        if fading == "half":
            pad_width //= 2
        sample = np.asarray(sample) + pad_width
    return np.maximum(0, (np.asarray(sample) - window_length // 2) // shift)


def _rows_to_frames(v, window_length, shift, fading):
    """bool [R, N] -> bool [R, frames].  Run starts / ends from one diff over the padded rows; every run
    marks +1 at f(start) and -1 at f(end), a cumulative sum > 0 is the union of the frame ranges."""
    R, N = v.shape
    frames = samples_to_stft_frames(N, window_length, shift, pad=True, fading=fading)
    edge = np.diff(np.pad(v.astype(np.int8), ((0, 0), (1, 1))), axis=-1)        # [R, N+1]
    r_s, s = np.nonzero(edge == 1)
    r_e, e = np.nonzero(edge == -1)               # exclusive ends, same run order per row as starts
    mark = np.zeros((R, frames + 1), dtype=np.int32)
    fs = np.minimum(sample_index_to_stft_frame_index(s, window_length, shift, fading), frames)
    fe = np.minimum(sample_index_to_stft_frame_index(e, window_length, shift, fading), frames)
    np.add.at(mark, (r_s, fs), 1)
    np.add.at(mark, (r_e, fe), -1)
    return np.cumsum(mark[:, :frames], axis=-1) > 0


def stft_vad(vad, window_length, shift, fading=True):
    """Move a sample activity to a frame activity.  torch in -> float32 tensor on the same device
    (utils.py:25-28), numpy in -> bool array, list / tuple -> list (utils.py:72-73)."""
    if isinstance(vad, torch.Tensor):
        out = stft_vad(vad.detach().cpu().numpy(), window_length, shift, fading)
        return torch.as_tensor(out, dtype=torch.float32).to(vad.device)
    if isinstance(vad, np.ndarray):
        v = vad.astype(bool)
        out = _rows_to_frames(v.reshape(-1, v.shape[-1]), window_length, shift, fading)
        return out.reshape(*v.shape[:-1], out.shape[-1])
    if isinstance(vad, (tuple, list)):
        return [stft_vad(v, window_length, shift, fading) for v in vad]
    raise TypeError(vad)
