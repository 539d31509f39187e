"""Enhancers -- drop-in for tssep/train/enhancer.py:21-265: the training-time classes (Dummy,
Nothing, Masking) and the eval-time mask-based MVDR beamformer TorchBF.
WPE / ClassicBF_np of the reference (numpy + nara_wpe / pb_bss on the CPU) are out of scope."""
import numpy as np
import torch

from .. import hip_ops as H
from ..configurable import Configurable


class ABC(Configurable):
    @property
    def name(self):
        return self.__class__.__name__


def _observation(masks, ex):
    """reference-channel selection shared by Nothing / Masking (enhancer.py:50-69,79-95)."""
    reference_channel = ex["reference_channel"]
    Observation = ex["Observation"]
    batched = {4: False, 5: True}[len(masks.shape)]
    if reference_channel is None:
        assert len(Observation.shape) == (3 if batched else 2), Observation.shape
    else:
        assert len(Observation.shape) == (4 if batched else 3), Observation.shape
        Observation = Observation[..., reference_channel, :, :]
    if isinstance(Observation, np.ndarray):
        Observation = torch.tensor(Observation, device=masks.device)
    return Observation, batched


class Dummy(ABC):
    def __call__(self, masks: torch.Tensor, ex, model):
        return None


class Nothing(ABC):
    def __call__(self, masks: torch.Tensor, ex, model):
        """the observation of the reference channel, with a speaker axis (enhancer.py:44-70)"""
        Observation, _ = _observation(masks, ex)
        return Observation[..., None, :, :]


class Masking(ABC):
    def __call__(self, masks: torch.Tensor, ex, model):
        """masks [B,K,1,T,F] -> complex [B,K,T,F] = Obs[ref] * mask (enhancer.py:98-100).
        Standalone form (mask tensor in).  ``Model.forward`` does not come through here for the
        Masking enhancer: it fuses sigmoid + product into one mask-head kernel."""
        Observation, batched = _observation(masks, ex)
        m = torch.squeeze(masks, dim=-3)
        if not batched:
            m, Observation = m[None], Observation[None]
        est = _MaskMul.apply(m.contiguous(), Observation.contiguous())
        return est if batched else est[0]


class _MaskMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mask, obs):
        ctx.save_for_backward(obs)
        return H.mask_mul_fwd(mask, obs)

    @staticmethod
    def backward(ctx, dest):
        (obs,) = ctx.saved_tensors
        return H.mask_mul_bwd(dest, obs), None


def trace(input, axis1=-2, axis2=-1):
    """batched trace (enhancer.py:103-137)"""
    assert input.shape[axis1] == input.shape[axis2], input.shape
    return torch.diagonal(input, dim1=axis1, dim2=axis2).sum(-1)


class TorchBF(ABC):
    """Mask-based MVDR (Souden) beamformer in complex128 -- enhancer.py:140-265, same constructor
    arguments, same call signature, same checks.  One fused pipeline of three HIP kernels
    (statistics, per-bin solve, filtering; csrc/mvdr.hip); forward only -- the reference uses it
    at evaluation time."""

    def __init__(self, bf="mvdr_souden", masking=False, masking_eps=0.0, eps=None):
        super().__init__()
        assert bf == "mvdr_souden", (bf, "Only mvdr_souden is implemented")
        self.bf = bf
        self.eps = eps
        self.masking = masking
        self.masking_eps = masking_eps

    def __call__(self, masks, ex, model):
        """masks [(B,) K, M, T, F] with M = 2 (target, interference) or 1 (interference = 1 - m);
        ex['Observation'] [(B,) D, T, F] complex128 -> [(B,) K, T, F] complex128."""
        batched = {4: False, 5: True}[len(masks.shape)]
        reference_channel = ex["reference_channel"]
        Observation = ex["Observation"]
        assert len(Observation.shape) == (4 if batched else 3), Observation.shape
        assert Observation.dtype == torch.complex128, Observation.dtype
        if masks.shape[-3] not in (1, 2):
            raise ValueError(masks.shape)
        if masks.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("TorchBF is an evaluation-time enhancer here: call it under "
                                      "torch.no_grad() (no backward kernel)")
        Observation = Observation.to(masks.device)
        if not batched:
            masks, Observation = masks[None], Observation[None]
        enh = H.mvdr_souden(masks.detach(), Observation, reference_channel, eps=self.eps,
                            masking=self.masking, masking_eps=self.masking_eps)
        return enh if batched else enh[0]
