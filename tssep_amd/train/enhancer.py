"""Masking enhancer -- drop-in for tssep/train/enhancer.py:21-100 (training-time classes).
The eval-time beamformers / WPE of the reference are out of the hot-path scope."""
import torch

from .. import hip_ops as H
from ..configurable import Configurable


class ABC(Configurable):
    @property
    def name(self):
        return self.__class__.__name__


class Masking(ABC):
    def __call__(self, masks: torch.Tensor, ex, model):
        """masks [B,K,1,T,F] -> complex [B,K,T,F] = Obs[ref] * mask (enhancer.py:98-100).
        Standalone form (mask tensor in).  ``Model.forward`` does not come through here for the
        Masking enhancer: it fuses sigmoid + product into one mask-head kernel."""
        reference_channel = ex["reference_channel"]
        Observation = ex["Observation"]
        batched = {4: False, 5: True}[len(masks.shape)]
        if reference_channel is not None:
            Observation = Observation[..., reference_channel, :, :]
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
