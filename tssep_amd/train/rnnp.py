"""RNNP_packed -- drop-in for tssep/train/rnnp.py:11-173 on the HIP kernels.

Same constructor, attribute names and ``state_dict`` keys (``net.0.weight_ih_l0`` ...): the
``torch.nn.LSTM`` / ``Linear`` / ``Tanh`` members are kept as PARAMETER CONTAINERS so that
initialisation order (torch.manual_seed parity) and checkpoints are interchangeable; their
``forward`` is never called -- the math runs in libtssep_hip.so.
"""
import torch

from .. import functional as Fn


class RNNP_packed(torch.nn.Module):
    def __init__(self, idim, elayers, cdim, hdim, dropout, typ="blstm", return_states=False):
        super().__init__()
        if typ != "blstm":
            raise NotImplementedError(f"typ={typ!r}: only 'blstm' is on the hot path (rnnp.py:40)")
        if dropout != 0:
            raise NotImplementedError("dropout > 0 (every shipped config uses 0)")
        assert not return_states, return_states          # asserted off at rnnp.py:121
        bidir = True
        net = []
        for i in range(elayers):
            inputdim = idim if i == 0 else hdim
            net.append(torch.nn.LSTM(inputdim, cdim, num_layers=1, bidirectional=bidir,
                                     batch_first=True))
            net.append(torch.nn.Linear(2 * cdim, hdim))
            if i < elayers - 1:
                net.append(torch.nn.Dropout(p=dropout))
                net.append(torch.nn.Tanh())
        self.net = torch.nn.ModuleList(net)
        self.elayers, self.cdim, self.typ, self.bidir = elayers, cdim, typ, bidir
        self.dropout, self.return_states = dropout, return_states
        self.hdim = hdim

    def forward_rows(self, rows, N, T, final_act=0, combine=0, in_tanh=0, next_folds=False):
        """rows: [N*T, I] (rows (n,t)) -> [N*T, hdim]; ``final_act``/``combine`` fuse the Tanh
        that follows this module in the post-net and the speaker-combination rearrange.  ``in_tanh``: the
        rows are the output of such a fused Tanh of the module in front (K > 1: in its speaker-combined
        layout) -- its backward is folded into this module's first d(input) GEMM; ``next_folds``: the module
        behind does the same for this module's final Tanh (functional.rnnp_layer)."""
        h = rows
        for i in range(self.elayers):
            lstm, lin = self.net[4 * i], self.net[4 * i + 1]
            last = i == self.elayers - 1
            fold_ok = self.hdim % 4 == 0 and Fn.H.FOLD_TANH
            h = Fn.rnnp_layer(h, lstm, lin, N, T, act=(final_act if last else 1),
                              combine=(combine if last else 0),
                              in_tanh=(in_tanh if i == 0 else (1 if fold_ok else 0)),
                              dz_given=((next_folds and final_act == 1) if last else fold_ok))
        return h

    def forward(self, xs_pack, prev_state=None):
        assert prev_state is None, prev_state                       # rnnp.py:121
        if isinstance(xs_pack, torch.nn.utils.rnn.PackedSequence):
            raise NotImplementedError("PackedSequence (unimplemented in the reference too, rnnp.py:124-129)")
        shape = xs_pack.shape
        if len(shape) == 4:
            N, T = shape[0] * shape[1], shape[2]
        elif len(shape) == 3:
            N, T = shape[0], shape[1]
        elif len(shape) == 2:
            N, T = 1, shape[0]
        else:
            raise KeyError(len(shape))
        y = self.forward_rows(xs_pack.reshape(N * T, shape[-1]), N, T)
        return y.reshape(*shape[:-1], y.shape[-1])
