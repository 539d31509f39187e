"""MaskEstimator_v2.forward -- oracle, CPU (batched input only).

Follows tssep/train/net.py:809-986 for ``aux_net=None`` and no normalizers
(every shipped config, init_cfg_common.yaml:70,79-80):
speaker shuffle :821-856, pre-net :858-860, conditioning mul :871-874 /
cat :879-894, averaged permutations :900-955, post-net built at :603-668,
un-permute :957-967, sigmoid :981-986.
"""
import numpy as np
import torch

from .rnnp import rnnp


def mask_estimator_forward(p, xs, aux, *, odim, nmask=1, combination="mul",
                           ts_vad=False, output_resolution="tf",
                           random_speaker_order=True,
                           num_averaged_permutations=1, layers=3,
                           prefix="mask_estimator.", fast=False, perm=None):
    """p: dict name->tensor (state_dict names, App. C of SURVEY.md).
    xs [B,T,D]; aux: list (B) of lists (K) of [E] tensors, or tensor [B,K,E].
    ``perm``: optional explicit permutations [B,K]; otherwise drawn from the
    global np.random exactly like net.py:824-826 (one permutation per batch
    entry, in batch order).  Returns dict(mask, logit, embedding, perm)."""
    assert xs.dim() == 3, xs.shape
    if not isinstance(aux, torch.Tensor):
        aux = torch.stack([torch.stack(list(a), 0) if isinstance(a, (list, tuple))
                           else a for a in aux], 0)
    B, K = aux.shape[:2]
    if random_speaker_order:
        if perm is None:
            perm = [np.random.permutation(K) for _ in range(B)]
        perm = np.asarray(perm)
        iperm = np.argsort(perm, axis=-1)
        aux = torch.stack([a[torch.as_tensor(q)] for a, q in zip(aux, perm)], 0)

    xs = rnnp(xs, p, prefix + "pre_net.", fast)                    # [B,T,odim]
    aux4 = aux.unsqueeze(-2)                                         # [B,K,1,E]
    if combination == "mul":
        xs = xs[..., None, :, :] * aux4
    elif combination == "cat":
        T = xs.shape[-2]
        xs = torch.concat([xs[:, None].expand(B, K, T, xs.shape[-1]),
                           aux4.expand(B, K, T, aux4.shape[-1])], dim=-1)
    else:
        raise NotImplementedError(combination)

    trials = num_averaged_permutations
    if trials > 1:
        idx = ((np.arange(K)[:, None] + np.arange(K)[None, :]) % K)[:trials].ravel()
        xs = xs[:, idx].reshape(B, trials, K, *xs.shape[-2:]) \
            .reshape(B * trials, K, *xs.shape[-2:])

    for l in range(layers):
        if l == layers - 1 and ts_vad is not False:
            # Rearrange '... spk time feature -> ... 1 time (spk feature)'
            n, k, t, f = xs.shape
            xs = xs.permute(0, 2, 1, 3).reshape(n, 1, t, k * f)
        xs = rnnp(xs, p, prefix + f"post_net.birnn{l}.", fast)
        if l < layers - 1:
            xs = torch.tanh(xs)                       # Dropout(p=0) is identity
    w, b = p[prefix + f"post_net.linear{layers - 1}.weight"], \
        p[prefix + f"post_net.linear{layers - 1}.bias"]
    xs = xs @ w.t() + b
    n, _, t, _ = xs.shape
    if output_resolution == "tf":
        if ts_vad is False:   # '... spk time (mask freq) -> ... spk mask time freq'
            logit = xs.reshape(n, xs.shape[1], t, nmask, odim).permute(0, 1, 3, 2, 4)
        else:                 # '... 1 time (spk mask freq) -> ... spk mask time freq'
            logit = xs.reshape(n, t, ts_vad, nmask, odim).permute(0, 2, 3, 1, 4)
    elif output_resolution == "t":
        if ts_vad is False:   # '... spk time mask -> ... spk mask time freq'
            logit = xs.permute(0, 1, 3, 2)[..., None].expand(n, xs.shape[1], nmask, t, odim)
        else:                 # '... 1 time (spk mask) -> ... spk mask time freq'
            logit = xs.reshape(n, t, ts_vad, nmask).permute(0, 2, 3, 1)[..., None] \
                .expand(n, ts_vad, nmask, t, odim)
    else:
        raise ValueError(output_resolution)

    if trials > 1:
        logit = logit.reshape(B, trials * K, *logit.shape[2:])
        revert_idx = np.argsort(idx.ravel())
        logit = logit[:, revert_idx]
        logit = logit.reshape(B, K, trials, *logit.shape[2:]).mean(dim=2)

    if random_speaker_order:
        logit = logit[np.arange(B)[:, None], iperm]
    return dict(mask=torch.sigmoid(logit), logit=logit, embedding=aux4, perm=perm)
