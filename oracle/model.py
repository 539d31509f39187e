"""End-to-end hot path -- oracle, CPU: Model.forward + loss part of Model.review
(tssep/train/model.py:465-536, 653-669)."""
import torch

from . import features, loss as oloss, net, stft as ostft


def init_mask_estimator_params(*, idim, odim, units, projs, layers=3,
                               combination="mul", aux_size=None, ts_vad=False,
                               output_resolution="tf", nmask=1,
                               prefix="mask_estimator.", dtype=torch.float32):
    """Parameters with torch's default init, created in the registration order
    of MaskEstimator_v2.__init__ (net.py:544-666) so that a given
    torch.manual_seed yields the same values as the reference module."""
    p = {}

    def add_rnnp(name, i, hdim):
        lstm = torch.nn.LSTM(i, units, num_layers=1, bidirectional=True,
                             batch_first=True)
        lin = torch.nn.Linear(2 * units, hdim)
        for k, v in lstm.named_parameters():
            p[f"{prefix}{name}.net.0.{k}"] = v.detach().to(dtype)
        p[f"{prefix}{name}.net.1.weight"] = lin.weight.detach().to(dtype)
        p[f"{prefix}{name}.net.1.bias"] = lin.bias.detach().to(dtype)

    add_rnnp("pre_net", idim, odim)
    first = odim + (aux_size if combination == "cat" else 0)
    ts_factor = 1
    for l in range(layers):
        if l == layers - 1 and ts_vad is not False:
            ts_factor = ts_vad
        add_rnnp(f"post_net.birnn{l}", (first if l == 0 else projs) * ts_factor, projs)
    out = (odim if output_resolution == "tf" else 1) * nmask * ts_factor
    lin = torch.nn.Linear(projs, out)
    p[f"{prefix}post_net.linear{layers - 1}.weight"] = lin.weight.detach().to(dtype)
    p[f"{prefix}post_net.linear{layers - 1}.bias"] = lin.bias.detach().to(dtype)
    return p


def forward_loss(p, observation, aux, target, *, cfg, loss="LogMAE",
                 fe_tables=None, window="hann", size=1024, shift=256,
                 mfcc=True, perm=None, fast=False, mel_db_max=None):
    """observation [B,1,N] f32, aux [B,K,E], target [B,K,N] (LogMAE) or
    Vad [B,K,T] (VADSigmoidBCE).  ``cfg`` = MaskEstimator kwargs for
    :func:`oracle.net.mask_estimator_forward`.  Returns dict with
    Observation, Input, mask, logit, stft_estimate, time_estimate, loss[B].
    ``mel_db_max`` (checker-only): the batch is a slice of a larger one whose mel-dB
    maximum is given (features.mel_db_max) -- AmplitudeToDB's floor is batch-global."""
    X = ostft.stft(observation, size=size, shift=shift, window=window)  # [B,1,T,F]
    Xr = X[..., 0, :, :]
    if mfcc:
        fb, dct = fe_tables if fe_tables is not None else features.mfcc_tables(size)
        inp = features.concat_features(Xr, fb, dct, db_max=mel_db_max).to(torch.float32)
    else:
        inp = features.log1p_max_norm_abs(Xr).to(torch.float32)
    out = net.mask_estimator_forward(p, inp, aux, perm=perm, fast=fast, **cfg)
    est = oloss.masking(out["mask"], X, 0)
    y = ostft.istft(est, size=size, shift=shift, window=window,
                    num_samples=observation.shape[-1])
    if loss == "LogMAE":
        lv = oloss.log_mae(y, target)
    elif loss == "VADSigmoidBCE":
        lv = oloss.vad_sigmoid_bce(torch.squeeze(out["logit"], dim=-3), target)
    else:
        raise ValueError(loss)
    return dict(Observation=X, Input=inp, mask=out["mask"], logit=out["logit"],
                stft_estimate=est, time_estimate=y, loss=lv, perm=out["perm"])
