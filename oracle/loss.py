"""Masking enhancer and the two shipped losses -- oracle, CPU."""
import torch


def masking(mask, Observation, reference_channel=0):
    """tssep/train/enhancer.py:73-100.  mask [B,K,M=1,T,F], Observation
    [B,C,T,F] complex -> [B,K,T,F] complex."""
    obs = Observation[..., reference_channel, :, :]
    return obs[..., None, :, :] * torch.squeeze(mask, dim=-3)


def log_mae(estimate, target):
    """tssep/train/loss.py:244-247: log10(sum_k mean_n |e-t|) -> [B]."""
    return torch.log10((estimate - target).abs().mean(dim=-1).sum(dim=-1))


def mae(estimate, target):
    """tssep/train/loss.py:214-216."""
    return (estimate - target).abs().mean(dim=-1).sum(dim=-1)


def vad_sigmoid_bce(logit, vad):
    """tssep/train/loss.py:329-345 + :302-310 with target 'Vad':
    logit [..., K, T, F] (mask axis already squeezed, loss.py:132),
    vad [..., K, T] -> mean over (T, K) of BCE-with-logits(mean_f logit)."""
    x = logit.mean(dim=-1)
    y = vad.to(x.dtype)
    l = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    return l.mean(dim=(-1, -2))


def vad2sep_broadcast(state_dict, shapes,
                      bcast=("mask_estimator.post_net.linear2.weight",
                             "mask_estimator.post_net.linear2.bias")):
    """tssep/train/init_ckpt.py:54-89 (mode='repeat')."""
    out = dict(state_dict)
    for k in bcast:
        p = out[k]
        for i, (actual, desired) in enumerate(zip(p.shape, shapes[k])):
            if actual < desired:
                assert desired % actual == 0
                p = torch.repeat_interleave(p, desired // actual, dim=i)
        out[k] = p
    return out
