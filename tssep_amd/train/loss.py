"""Losses -- drop-ins for tssep/train/loss.py on the HIP kernels: ``LogMAE`` (:219-247) and
``VADSigmoidBCE`` (:272-345) with their ``from_ex_out`` glue (:89-99, :118-146).
MSE / MAE / FreqMSE / SignalAndVADSigmoidBCE are not selected by any shipped config."""
import torch

from .. import functional as Fn
from ..configurable import Configurable


class ABC(Configurable, torch.nn.Module):
    def __init__(self, target: str = "speaker_reverberation_early_ch0", pit: bool = False):
        super().__init__()
        if pit:
            raise NotImplementedError("pit=True (every shipped config uses pit: false)")
        self.target, self.pit = target, pit

    def _upper(self, s):
        return s[0].upper() + s[1:]

    def targets(self, lower=False, upper=False):         # loss.py:30-40
        if lower:
            assert not upper
            return tuple(t.lower() for t in self.targets())
        if upper:
            return tuple(self._upper(t) for t in self.targets())
        return (self.target,)

    @property
    def name(self):
        return self.__class__.__name__

    def forward(self, estimate, target):
        assert estimate.shape == target.shape, (estimate.shape, target.shape)
        return self.loss_fn(estimate, target)

    def update_summary(self, summary, ex, out, model):
        pass


class TimeDomain(ABC):
    def from_ex_out(self, ex, out, model, summary):      # loss.py:90-99
        return self(out.time_estimate, ex[self.target])


class LogMAE(TimeDomain):
    def loss_fn(self, estimate, target):
        """log10(sum_k mean_n |e - t|) -> [B]  (loss.py:244-247)"""
        if estimate.dim() == 2:
            return Fn.log_mae(estimate[None], target[None])[0]
        return Fn.log_mae(estimate, target)


class MAE(TimeDomain):
    def loss_fn(self, estimate, target):
        """sum_k mean_n |e - t| -> [B]  (loss.py:194-216)"""
        if estimate.dim() == 2:
            return Fn.mae(estimate[None], target[None])[0]
        return Fn.mae(estimate, target)


class LogitsSTFTDomain(ABC):
    def from_ex_out(self, ex, out, model, summary):      # loss.py:122-146
        estimate = torch.squeeze(out.logit, dim=-3)
        assert self.target[0].isupper(), self.target
        if self.target not in ex:
            if self.target == "Vad":
                from ..util.utils import stft_vad
                ex[self.target] = stft_vad(ex[self.target.lower()], model.fe.window_length,
                                           model.fe.shift, model.fe.fading)
            else:
                raise NotImplementedError(self.target)
        return self(estimate, ex[self.target])

    def update_summary(self, summary, ex, out, model):   # loss.py:148-169: the mask image framed by the target activity
        import einops
        target_vad = einops.repeat(self.prepare_target(torch.as_tensor(ex[self.target])).to(out.mask.device, torch.float32),
                                   "... spk time -> ... spk mask time freq", freq=40, mask=out.mask.shape[-3])
        masks = torch.concat([target_vad, out.mask.detach(), target_vad], dim=-1)
        summary.add_mask_image(f"{model.enhancer.name}_mask", masks,
                               rearrange="... spk mask time freq -> ... time (spk mask freq)", batch_first=True)


class VADSigmoidBCE(LogitsSTFTDomain):
    def __init__(self, target: str = "Vad", pit: bool = False, magnitude_threshold: float = 0.05):
        super().__init__(target=target, pit=pit)
        assert 0 < magnitude_threshold < 1, magnitude_threshold
        self.magnitude_threshold = magnitude_threshold

    def prepare_target(self, target, dtype=None):
        if self.target in ["vad", "Vad"]:
            return target
        raise NotImplementedError("STFT-magnitude VAD targets (loss.py:316-327) are off the hot path")

    def forward(self, estimate, target):                 # loss.py:329-345
        if not isinstance(target, torch.Tensor):
            target = torch.stack(target)
        if self.target not in ["vad", "Vad"]:
            raise NotImplementedError(self.target)
        target = target.to(device=estimate.device, dtype=torch.float32)
        if estimate.dim() == 3:
            return Fn.vad_bce(estimate[None], target[None])[0]
        return Fn.vad_bce(estimate, target)
