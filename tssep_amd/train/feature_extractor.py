"""STFT feature extractors -- drop-ins for tssep/train/feature_extractor.py (and the
padertorch ``STFT`` base it star-imports, feature_extractor.py:8) on the HIP kernels.

Implemented: ``STFT`` (base: stft / istft / __call__), ``Log1pMaxNormAbsSTFT`` (:183-263),
``ConcaternatedSTFTFeatures`` (:290-367), ``TorchMFCC`` (feature_extractor_torchaudio.py).
Inputs must be CUDA tensors: there is no CPU path in this package.
"""
import math

import numpy as np
import torch

from .. import functional as Fn
from .. import hip_ops as H
from ..configurable import Configurable


def _cuda_f32(x):
    if isinstance(x, np.ndarray):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        raise RuntimeError("tssep_amd runs on the GPU only: move the input to cuda first "
                           "(no CPU fallback)")
    return x.to(torch.float32)


class STFT(Configurable):
    """padertorch.contrib.cb.feature_extractor.STFT on paderbox.transform.module_stft (star-imported at
    feature_extractor.py:8): every option the reference's configs may set (init_cfg_common.yaml:33-43) --
    ``window_length <= size`` (the transform is zero-padded), ``pad`` (False: only full frames), ``fading`` in
    {True / 'full', 'half', False / None}.  The FFT plan is 1024 / 256 (any other: TSSEP_E_UNSUPPORTED from
    the library, naming the plan)."""

    def __init__(self, size=1024, shift=256, window_length=None, pad=True, fading=True,
                 output_size=None, window="blackman"):
        if window_length is None:
            window_length = size
        if not 0 < window_length <= size:
            raise ValueError(f"window_length = {window_length} must lie in (0, size = {size}]")
        if fading not in (None, False, True, "full", "half"):
            raise ValueError(f"fading = {fading!r}: None, False, True, 'full' or 'half' (paderbox stft)")
        self.size, self.shift, self.window_length = size, shift, window_length
        self.pad, self.fading, self.window = pad, fading, window
        self.output_size = self._get_output_size(output_size)

    @property
    def frequencies(self):
        return self.size // 2 + 1

    def _get_output_size(self, output_size):
        if output_size is None:
            return self.frequencies
        return output_size

    def _windows(self, device):
        return Fn.windows(self.window, self.size, self.shift, device, self.window_length)

    def _fade(self):
        """(zeros in front, zeros behind) that `fading` adds to a signal (paderbox stft): 'full' = window_length -
        shift on both sides, 'half' = half of that (the odd sample behind)."""
        if self.fading in (None, False):
            return 0, 0
        p = self.window_length - self.shift
        if self.fading == "half":
            return p // 2, p - p // 2
        return p, p

    def _plain(self):
        """The configuration the fused kernels are built for: the 1024 / 256 plan, full fading, a full-size window."""
        return self.window_length == self.size and self.fading in (True, "full") and self._check_plan() == 1

    def _check_plan(self):
        """1: the plan specialised to the shipped configs (1024 / 256: fused mask head + inverse STFT); 2: the general plan
        (stft_generic.hip: even sizes with size / 2 = 2^a 3^b 5^c <= 2048, shift <= min(size, 512) -- 512 / 128, TorchMFCC's
        own default 400 / 200, ...); anything else raises, naming the plan."""
        from .. import _lib
        plan = int(_lib.lib().tssep_stft_plan(int(self.size), int(self.shift)))
        if not plan:
            raise RuntimeError(f"unsupported FFT plan size={self.size}, shift={self.shift}: libtssep_hip.so builds even sizes "
                               "with size / 2 = 2^a 3^b 5^c <= 2048 and shift <= min(size, 512) whose two LDS lines per "
                               "wave fit the device (TSSEP_E_UNSUPPORTED)")
        return plan

    FEATURE_MAX_BINS = 1032      # MAXF of csrc/feat.hip: the feature kernels keep one LDS line of F bins per wave

    def _check_feature_bins(self):
        """The feature kernels (tssep_feat_fwd) take at most FEATURE_MAX_BINS frequency bins (size <= 2062): say so up
        front instead of TSSEP_E_SHAPE out of the kernel launch after the STFT plan was accepted."""
        F = self.size // 2 + 1
        if F > self.FEATURE_MAX_BINS:
            raise RuntimeError(f"{type(self).__name__}: size={self.size} gives {F} frequency bins; the feature kernels of "
                               f"libtssep_hip.so take at most {self.FEATURE_MAX_BINS} (size <= {2 * (self.FEATURE_MAX_BINS - 1)}); "
                               "the STFT / iSTFT themselves support this plan")

    def frames(self, num_samples):
        lead, tail = self._fade()
        n = num_samples + lead + tail
        if self.pad:
            return max(-(-(n - self.window_length) // self.shift), 0) + 1
        if n < self.window_length:
            raise ValueError(f"pad=False: {num_samples} samples (+ fading) are shorter than one window")
        return (n - self.window_length) // self.shift + 1

    def stft(self, signal):
        """[..., N] -> complex64 [..., T, F]  (fe.stft, model.py:503-504)"""
        self._check_plan()
        x = _cuda_f32(signal)
        w, _ = self._windows(x.device)
        x2 = x.reshape(-1, x.shape[-1])
        T = self.frames(x.shape[-1])
        if self._plain():
            X = H.stft_fwd(x2, w, self.size, self.shift, True, T=T)
        else:
            # the kernel knows "size - shift zeros in front" or none: any other fade is laid out here (zero padding is
            # layout, the transform stays the library's); frames reaching past the signal read zeros (pad=True) or
            # do not exist (pad=False: T is smaller)
            lead, tail = self._fade()
            if lead or tail:
                x2 = torch.nn.functional.pad(x2, (lead, tail))
            X = H.stft_fwd(x2, w, self.size, self.shift, False, T=T)
        return X.reshape(*x.shape[:-1], X.shape[-2], X.shape[-1])

    def _num_samples(self, T, num_samples):
        lead, tail = self._fade()
        full = (T - 1) * self.shift + self.window_length - lead - tail
        return full if num_samples is None else min(num_samples, full)

    def istft(self, signal, num_samples=None):
        """complex [..., T, F] -> [..., N]  (fe.istft, model.py:661-664); differentiable."""
        self._check_plan()
        if self.window_length % self.shift:
            raise ValueError("istft: window_length must be a multiple of shift (paderbox's biorthogonal window)")
        T = signal.shape[-2]
        N = self._num_samples(T, num_samples)
        _, wsyn = self._windows(signal.device)
        lead, _ = self._fade()
        lead0 = self.size - self.shift                 # what the kernel drops in front
        if lead == lead0:
            return Fn.istft(signal, wsyn, N, self.size, self.shift, True)
        # another fade: k empty frames in front move the overlap-add by k hops, d more samples are computed and cut
        # (layout only; autograd takes pad and slice back)
        k = -(-(lead0 - lead) // self.shift)
        d = lead - lead0 + k * self.shift
        Xp = torch.nn.functional.pad(torch.view_as_real(signal), (0, 0, 0, 0, k, 0))
        y = Fn.istft(torch.view_as_complex(Xp), wsyn, N + d, self.size, self.shift, True)
        return y[..., d:]

    def masked_istft(self, logit, observation, num_samples=None, target=None):
        """istft(sigmoid(logit) * observation) as one fused kernel each way (functional.mask_istft):
        logit [B,K,T,F], observation complex [B,T,F] -> [B,K,N]; ``target`` [B,K,N] (optional) lets the
        forward accumulate the |estimate - target| sums a LogMAE / MAE loss needs."""
        if not self._plain():         # the fused tail is built for the shipped STFT; any other: mask head, then istft
            _, est = Fn.mask_head(logit, observation)
            return self.istft(est, num_samples=num_samples)
        N = self._num_samples(logit.shape[-2], num_samples)
        _, wsyn = self._windows(logit.device)
        return Fn.mask_istft(logit, observation, wsyn, N, self.size, self.shift, True, tgt=target)

    def stft_to_feature(self, stft_signals):
        raise NotImplementedError(type(self))

    def __call__(self, signal):
        return self.stft_to_feature(self.stft(signal))

    def sample_index_to_frame_index(self, sample_index):
        """paderbox STFT.sample_index_to_frame_index (feature_extractor.py:208,306)."""
        from ..util.utils import sample_index_to_stft_frame_index
        return sample_index_to_stft_frame_index(sample_index, self.window_length, self.shift, self.fading)


class Log1pMaxNormAbsSTFT(STFT):
    def __init__(self, size=1024, shift=256, window_length=None, pad=True, fading=True,
                 output_size=None, window="blackman", statistics_axis="tf"):
        super().__init__(size=size, shift=shift, window_length=window_length, pad=pad,
                         fading=fading, output_size=output_size, window=window)
        if statistics_axis not in H.STAT_AXES:
            raise ValueError(f"statistics_axis = {statistics_axis!r}: 'tf', 't' or 'f' (feature_extractor.py:239-242)")
        self.statistics_axis = statistics_axis
        self._check_feature_bins()

    def stft_to_feature(self, stft_signals):
        X = stft_signals
        lead = X.shape[:-2]
        out, _ = H.feat_fwd(X.reshape(-1, X.shape[-2], X.shape[-1]).contiguous(), None, None, 0,
                            statistics_axis=self.statistics_axis)
        return out.reshape(*lead, X.shape[-2], X.shape[-1])


class TorchMFCC(STFT, torch.nn.Module):
    """feature_extractor_torchaudio.py:11-106; filterbank / DCT tables restated from
    torchaudio 2.0.2 (absent here; parity unpinned, see oracle/features.py)."""

    def __init__(self, size=400, shift=200, window_length=None, pad=True, fading=True,
                 output_size=None, window="hann", sample_rate: int = 16000, n_mfcc: int = 40,
                 dct_norm: str = "ortho", log_mels: bool = False, f_min: float = 40,
                 f_max: float = -400, n_mels: int = 40, mel_norm: str = None,
                 mel_scale: str = "htk"):
        torch.nn.Module.__init__(self)
        self.n_mfcc = n_mfcc
        STFT.__init__(self, size=size, shift=shift, window_length=window_length, pad=pad,
                      fading=fading, output_size=output_size, window=window)
        # every option of feature_extractor_torchaudio.py:22-39 (round 5; the shipped configs use the defaults): the mel scale
        # ('htk' | 'slaney'), its area normalisation (None | 'slaney'), the DCT normalisation ('ortho' | None) are TABLES
        # handed to tssep_feat_fwd, `log_mels` is an argument of the kernel -- restated from torchaudio 2.0.2 (absent here:
        # parity unpinned like the defaults, oracle/features.py)
        if mel_scale not in ("htk", "slaney"):
            raise ValueError('mel_scale should be one of "htk" or "slaney"')        # (torchaudio's message)
        if mel_norm is not None and mel_norm != "slaney":
            raise ValueError('norm must be one of None or "slaney"')
        if dct_norm is not None and dct_norm != "ortho":
            raise ValueError("norm must be either 'ortho' or None")
        self.sample_rate, self.f_min = sample_rate, f_min
        if f_max and f_max < 0:
            f_max = sample_rate + f_max                       # :57-58
        self.f_max, self.n_mels = f_max, n_mels
        self.dct_norm, self.mel_norm, self.top_db, self.log_mels = dct_norm, mel_norm, 80, log_mels
        # module tree of the reference (feature_extractor_torchaudio.py:69-85): a parameter-free
        # ``amplitude_to_DB``, ``mel_scale`` holding the persistent buffer ``fb`` and the buffer
        # ``dct_mat`` -> checkpoint keys ``<fe>.dct_mat`` and ``<fe>.mel_scale.fb``, so a checkpoint the
        # reference wrote loads with strict=True (init_cfg_tssep.yaml:22)
        self.amplitude_to_DB = _AmplitudeToDB("power", self.top_db)
        self.mel_scale = _MelScale(n_mels, sample_rate, f_min, f_max, size // 2 + 1, mel_norm, mel_scale)
        self.register_buffer("dct_mat", _create_dct(n_mfcc, n_mels, dct_norm))

    @property
    def fb(self):
        return self.mel_scale.fb

    def _get_output_size(self, output_size):
        return self.n_mfcc if output_size is None else output_size

    def stft_to_feature(self, stft_signals):
        X = stft_signals
        if X.dim() == 2:        # un-batched example: the dB floor is per utterance (torchaudio 2-D case)
            return self.stft_to_feature(X[None])[0]
        assert X.dim() == 3, X.shape
        self._check_feature_bins()
        out, _ = H.feat_fwd(X, self.fb, self.dct_mat, self.n_mfcc, self._db_arg)
        return out[..., :self.n_mfcc]

    @property
    def _db_arg(self):
        """top_db for tssep_feat_fwd; negative = `log_mels` (log(mel + 1e-6), no floor: include/tssep_hip.h)."""
        return -1.0 if self.log_mels else float(self.top_db)


class _MelScale(torch.nn.Module):
    """Holder of the mel filterbank under torchaudio's key (``MelScale.fb``, a persistent buffer
    [n_freqs, n_mels]); the filtering itself runs inside tssep_feat_fwd."""

    def __init__(self, n_mels, sample_rate, f_min, f_max, n_stft, norm=None, mel_scale="htk"):
        super().__init__()
        self.n_mels, self.sample_rate, self.f_min, self.f_max = n_mels, sample_rate, f_min, f_max
        self.norm, self.mel_scale = norm, mel_scale
        self.register_buffer("fb", _melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate, norm, mel_scale))


class _AmplitudeToDB(torch.nn.Module):
    """Parameter- and buffer-free, like torchaudio's; keeps the attribute name of the reference."""

    def __init__(self, stype, top_db):
        super().__init__()
        self.stype, self.top_db = stype, top_db


def _hz_to_mel(freq, mel_scale="htk"):
    """torchaudio.functional._hz_to_mel (2.0.2): HTK 2595 log10(1 + f / 700); Slaney: linear below 1 kHz
    (200 / 3 Hz per mel), logarithmic above (step log(6.4) / 27)."""
    if mel_scale == "htk":
        return 2595.0 * math.log10(1.0 + freq / 700.0)
    f_sp = 200.0 / 3
    mels = freq / f_sp
    min_log_hz = 1000.0
    if freq >= min_log_hz:
        mels = min_log_hz / f_sp + math.log(freq / min_log_hz) / (math.log(6.4) / 27.0)
    return mels


def _mel_to_hz(mels, mel_scale="htk"):
    if mel_scale == "htk":
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_mel = 1000.0 / f_sp
    log_t = mels >= min_log_mel
    freqs[log_t] = 1000.0 * torch.exp((math.log(6.4) / 27.0) * (mels[log_t] - min_log_mel))
    return freqs


def _melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio.functional.melscale_fbanks (2.0.2) -> [n_freqs, n_mels]; norm='slaney': every triangle divided by the
    width of its mel band (area normalisation)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_mel(f_min, mel_scale), _hz_to_mel(f_max, mel_scale), n_mels + 2)
    f_pts = _mel_to_hz(m_pts, mel_scale)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    if norm == "slaney":
        fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])).unsqueeze(0)
    return fb.contiguous()


def _create_dct(n_mfcc, n_mels, norm="ortho"):
    """torchaudio.functional.create_dct -> [n_mels, n_mfcc] (norm None: the DCT-II scaled by 2)."""
    n = torch.arange(float(n_mels))
    k = torch.arange(float(n_mfcc)).unsqueeze(1)
    dct = torch.cos(math.pi / float(n_mels) * (n + 0.5) * k)
    if norm is None:
        dct *= 2.0
    else:
        dct[0] *= 1.0 / math.sqrt(2.0)
        dct *= math.sqrt(2.0 / float(n_mels))
    return dct.t().contiguous()


class ConcaternatedSTFTFeatures(STFT, torch.nn.Module):
    @classmethod
    def finalize_dogmatic_config(cls, config):               # feature_extractor.py:308-321
        for fe in ["fe1", "fe2"]:
            if isinstance(config.get(fe), dict):
                for k in ("size", "shift", "pad", "fading", "window"):
                    if k in config:
                        config[fe][k] = config[k]
                if config.get("window_length") is not None:
                    config[fe]["window_length"] = config["window_length"]

    def __init__(self, fe1, fe2, output_size=None, size=1024, shift=256, window="blackman",
                 window_length=None, pad=True, fading=True):
        torch.nn.Module.__init__(self)
        self._tmp = [fe1, fe2]
        STFT.__init__(self, size=size, shift=shift, window_length=window_length, pad=pad,
                      fading=fading, output_size=output_size, window=window)
        self.fe1, self.fe2 = fe1, fe2

    def _get_output_size(self, output_size):
        fe1, fe2 = self._tmp
        if output_size is None:
            return fe1._get_output_size(None) + fe2._get_output_size(None)
        return output_size

    def stft_to_feature(self, stft_signals):
        X = stft_signals
        if X.dim() == 2:
            return self.stft_to_feature(X[None])[0]
        if isinstance(self.fe1, TorchMFCC) and isinstance(self.fe2, Log1pMaxNormAbsSTFT) \
                and X.dim() == 3:
            # one fused pass pair writes [mfcc | log1p] side by side (feature_extractor.py:352-360)
            out, _ = H.feat_fwd(X, self.fe1.fb, self.fe1.dct_mat, self.fe1.n_mfcc, self.fe1._db_arg,
                                statistics_axis=self.fe2.statistics_axis)
            return out
        return torch.concat([self.fe1.stft_to_feature(X), self.fe2.stft_to_feature(X)], dim=-1)
