"""Feature extractors (oracle, CPU): TorchMFCC, Log1pMaxNormAbsSTFT, concat.

Follows tssep/train/feature_extractor_torchaudio.py:22-106 (TorchMFCC),
tssep/train/feature_extractor.py:233-248 (Log1pMaxNormAbsSTFT) and :352-360
(ConcaternatedSTFTFeatures).  The mel filterbank / AmplitudeToDB / DCT live in
torchaudio==2.0.2 (absent): restated from its published algorithm, parity
unpinned (no reference test asserts an MFCC value).
"""
import math

import numpy as np
import torch


def _hz_to_mel(freq, mel_scale="htk"):
    """torchaudio.functional._hz_to_mel (2.0.2)."""
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    if freq >= 1000.0:
        mels = 1000.0 / f_sp + math.log(freq / 1000.0) / (math.log(6.4) / 27.0)
    return mels


def _mel_to_hz(mels, mel_scale="htk"):
    if mel_scale == "htk":
        return 700.0 * (10 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_mel = 1000.0 / f_sp
    log_t = mels >= min_log_mel
    freqs[log_t] = 1000.0 * torch.exp((math.log(6.4) / 27.0) * (mels[log_t] - min_log_mel))
    return freqs


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio.functional.melscale_fbanks(norm=None | 'slaney', mel_scale='htk' | 'slaney')."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min, mel_scale), _hz_to_mel(f_max, mel_scale), n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)  # [n_freqs, n_mels]
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb


def create_dct(n_mfcc, n_mels, norm="ortho"):
    """torchaudio.functional.create_dct -> [n_mels, n_mfcc]."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    if norm is None:
        dct *= 2.0
    else:
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


def mfcc_tables(size=1024, sample_rate=16000, n_mfcc=40, f_min=40.0,
                f_max=-400.0, n_mels=40, dct_norm="ortho", mel_norm=None, mel_scale="htk"):
    """fb[F, n_mels], dct[n_mels, n_mfcc] as built at
    feature_extractor_torchaudio.py:57-85 (negative f_max wraps to sr+f_max)."""
    if f_max and f_max < 0:
        f_max = sample_rate + f_max
    fb = melscale_fbanks(size // 2 + 1, f_min, f_max, n_mels, sample_rate, mel_norm, mel_scale)
    return fb, create_dct(n_mfcc, n_mels, dct_norm)


def amplitude_to_db_power(x, top_db=80.0, db_max=None):
    """torchaudio AmplitudeToDB('power', top_db) incl. its batching quirk:
    3-D input is treated as the channels of ONE item (max over everything),
    4-D as [batch, channel, freq, time] (max per batch item).
    ``db_max`` (checker-only extension, not in the reference): the 3-D input is
    a SLICE of a larger batch whose dB maximum is given -- the floor is then the
    one the reference would have applied to the whole batch (`mel_db_max`)."""
    x_db = 10.0 * torch.log10(torch.clamp(x, min=1e-10))
    shape = x_db.size()
    packed_channels = shape[-3] if x_db.dim() > 2 else 1
    v = x_db.reshape(-1, packed_channels, shape[-2], shape[-1])
    top = v.amax(dim=(-3, -2, -1)) if db_max is None else torch.full((v.shape[0],), float(db_max))
    v = torch.max(v, (top - top_db).view(-1, 1, 1, 1))
    return v.reshape(shape)


def mel_db_max(X, fb):
    """Maximum of 10 log10(mel power) over a whole batch X[B, T, F]: what
    `amplitude_to_db_power` takes its floor from for a 3-D input."""
    power = abs(X).to(torch.float32) ** 2
    mel = torch.matmul(power, fb)
    return float((10.0 * torch.log10(torch.clamp(mel, min=1e-10))).amax())


def torch_mfcc(X, fb, dct, top_db=80.0, db_max=None, log_mels=False):
    """feature_extractor_torchaudio.py:93-106.  X[..., T, F] complex."""
    power = abs(X.transpose(-1, -2)).to(torch.float32) ** 2   # [..., F, T]
    mel = torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)
    if log_mels:                                              # :98-100
        mel = torch.log(mel + 1e-6)
    else:
        mel = amplitude_to_db_power(mel, top_db, db_max)      # [..., n_mels, T]
    return torch.matmul(mel.transpose(-1, -2), dct)           # [..., T, n_mfcc]


def log1p_max_norm_abs(X, statistics_axis="tf"):
    """feature_extractor.py:233-248 (torch branch) / :249-263 (numpy)."""
    if isinstance(X, np.ndarray):
        s = np.abs(X)
        axis = {"tf": (-2, -1), "t": -2, "f": -1}[statistics_axis]
        return np.log1p(s * ((np.e - 1) / np.amax(s, keepdims=True, axis=axis)))
    s = abs(X)
    dim = {"tf": (-2, -1), "t": -2, "f": -1}[statistics_axis]
    s = s * ((np.e - 1) / torch.amax(s, keepdim=True, dim=dim))
    return torch.log1p(s)


def concat_features(X, fb, dct, db_max=None):
    """ConcaternatedSTFTFeatures(TorchMFCC, Log1pMaxNormAbsSTFT)
    (feature_extractor.py:352-360, init_cfg_common.yaml:13-50) -> [..., T, 553]."""
    return torch.concat([torch_mfcc(X, fb, dct, db_max=db_max), log1p_max_norm_abs(X)], dim=-1)
