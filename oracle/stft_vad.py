"""TEST INFRASTRUCTURE ONLY.  Plain-loop restatement of ``stft_vad`` (tssep/util/utils.py:11-77) and of the
two paderbox 0.0.8 helpers it imports (utils.py:20-23: ``paderbox.transform.module_stft.
_samples_to_stft_frames`` and ``sample_index_to_stft_frame_index``).

paderbox is a third-party dependency of the reference (requirements.txt:17, ``paderbox==0.0.8``), absent
from /root/reference and not installable here, and no test or doctest of the reference asserts a value of
``stft_vad`` -> **parity unpinned** for the helpers' arithmetic (restated from the published package:
frame count ``ceil((N + 2*(W - S) - W + S) / S)``, which the reference's own frame-count doctests pin --
model.py:480 80000 -> 316, feature_extractor.py:200 10000 -> 43 -- and the frame index
``max(0, (n + (W - S) - W//2) // S)``).  The STRUCTURE follows the reference line by line: runs of active
samples -> start / exclusive end through the frame-index map -> ``ai[start:end] = True``.
"""
import math

import numpy as np


def samples_to_stft_frames(samples, size, shift, pad=True, fading=True):
    # paderbox module_stft._samples_to_stft_frames; called at utils.py:34-42 with pad=True
    if fading not in (None, False):
        pad_width = size - shift
        samples = samples + (1 + (fading != "half")) * pad_width
    frames = (samples - size + shift) / shift
    return math.ceil(frames) if pad else math.floor(frames)


def sample_index_to_stft_frame_index(sample, window_length, shift, fading=True):
    # paderbox module_stft.sample_index_to_stft_frame_index; called at utils.py:53-64
    if fading not in (None, False):
        pad_width = window_length - shift
        if fading == "half":
            pad_width //= 2
        sample = sample + pad_width
    if sample < window_length // 2:
        return 0
    return (sample - window_length // 2) // shift


def normalized_intervals(a):
    # paderbox ArrayInterval(a).normalized_intervals: maximal runs [start, end) of True samples
    runs, start = [], None
    for i, x in enumerate(a):
        if x and start is None:
            start = i
        elif not x and start is not None:
            runs.append((start, i))
            start = None
    if start is not None:
        runs.append((start, len(a)))
    return runs


def stft_vad(vad, window_length, shift, fading=True):
    """bool [..., N] -> bool [..., frames]  (utils.py:30-71, the numpy branch)"""
    vad = np.asarray(vad).astype(bool)
    N = vad.shape[-1]
    frames = samples_to_stft_frames(N, size=window_length, shift=shift, pad=True, fading=fading)
    out = np.zeros(vad.shape[:-1] + (frames,), dtype=bool)
    for idx in np.ndindex(vad.shape[:-1]):
        for start, end in normalized_intervals(vad[idx]):
            fs = sample_index_to_stft_frame_index(start, window_length, shift, fading)
            fe = sample_index_to_stft_frame_index(end, window_length, shift, fading)
            out[idx][fs:fe] = True
    return out
