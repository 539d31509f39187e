"""BLSTM + projection ("RNNP", one layer) -- oracle, CPU.

Restates what tssep/train/rnnp.py:88-96,146-168 executes for elayers=1:
``torch.nn.LSTM(I, H, bidirectional=True, batch_first=True)`` followed by
``Linear(2H, hdim)``; 4-D inputs are flattened ``(batch speaker)`` (:131-134).
The LSTM cell is written out explicitly (gate row blocks i,f,g,o; zero initial
state, rnnp.py:121) so the oracle is an independent statement of the math;
``tests/test_oracle.py`` also checks it against ``torch.nn.LSTM``.
"""
import torch


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """x[N,T,I] -> h[N,T,H] for one direction."""
    N, T, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(N, H)
    c = x.new_zeros(N, H)
    out = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gx[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(H, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return torch.stack(out, dim=1)


def blstm(x, p, prefix):
    """Bidirectional LSTM with torch parameter names under ``prefix``."""
    f = lstm_direction(x, p[prefix + "weight_ih_l0"], p[prefix + "weight_hh_l0"],
                       p[prefix + "bias_ih_l0"], p[prefix + "bias_hh_l0"])
    b = lstm_direction(x, p[prefix + "weight_ih_l0_reverse"],
                       p[prefix + "weight_hh_l0_reverse"],
                       p[prefix + "bias_ih_l0_reverse"],
                       p[prefix + "bias_hh_l0_reverse"], reverse=True)
    return torch.cat([f, b], dim=-1)


def blstm_fast(x, p, prefix):
    """Same math through ATen's fused CPU LSTM (what the reference itself runs
    on a CPU); used for the timed cpu_baseline and for large oracle cases."""
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    flat = [p[prefix + n] for n in names] + [p[prefix + n + "_reverse"] for n in names]
    N = x.shape[0]
    H = flat[1].shape[1]
    z = x.new_zeros(2, N, H)
    out, _, _ = torch.lstm(x, (z, z), flat, True, 1, 0.0, False, True, True)
    return out


def rnnp(x, p, prefix, fast=False):
    """RNNP_packed.forward for elayers=1 (rnnp.py:111-173).
    x: [N,T,I], [T,I] or [B,K,T,I];  params ``prefix+'net.0.*'``, ``'net.1.*'``."""
    shape = x.shape
    if x.dim() == 4:
        h = x.reshape(shape[0] * shape[1], shape[2], shape[3])
    elif x.dim() == 2:
        h = x[None]
    else:
        h = x
    h = (blstm_fast if fast else blstm)(h, p, prefix + "net.0.")
    h = h @ p[prefix + "net.1.weight"].t() + p[prefix + "net.1.bias"]
    if x.dim() == 4:
        h = h.reshape(shape[0], shape[1], shape[2], -1)
    elif x.dim() == 2:
        h = h[0]
    return h
