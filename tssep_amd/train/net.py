"""MaskEstimator_v2 -- drop-in for tssep/train/net.py:333-986 on the HIP kernels.

Constructor signature, module tree and parameter registration order follow the reference
(net.py:501-669) so ``state_dict`` keys, default-init random streams and ``config.yaml`` are
interchangeable.  The einops layout changes of the reference (speaker combination, final
rearrange, trial mean, speaker un-permutation) are not separate copies here: they are folded
into the producing GEMM's store or one fused map kernel.
"""
import collections
import dataclasses

import numpy as np
import torch

from .. import functional as Fn
from ..configurable import Configurable
from .rnnp import RNNP_packed


@dataclasses.dataclass
class Output:                      # net.py:240-247
    mask: torch.Tensor
    logit: torch.Tensor
    embedding: torch.Tensor = None
    vad_mask: torch.Tensor = None
    vad_logit: torch.Tensor = None


class Sequential(torch.nn.Sequential):        # net.py:190-237 (container role only)
    pass


class _Marker(torch.nn.Module):
    """Parameter-free stand-in for the einops layers of the reference post-net (keeps the
    ``post_net`` key numbering: rearrange2, linear2, rearrange3 ...)."""

    def __init__(self, pattern):
        super().__init__()
        self.pattern = pattern

    def extra_repr(self):
        return repr(self.pattern)


class MaskEstimator_v2(Configurable, torch.nn.Module):
    @classmethod
    def finalize_dogmatic_config(cls, config):        # net.py:342-499
        if config.get("aux_net") is not None:
            raise NotImplementedError("aux_net (null in every shipped config, init_cfg_common.yaml:70)")
        if config.get("combination", "cat") == "cat" and config.get("aux_net_output_size") is None:
            config["aux_net_output_size"] = 100

    def __init__(self, *, idim=80, odim=None, layers=3, units=300, projs=320, dropout=0, nmask=1,
                 pre_net="RNNP", aux_net=None, aux_net_output_size=None, combination: str = "cat",
                 ts_vad=False, output_resolution: str = "tf", random_speaker_order=True,
                 num_averaged_permutations=1, input_normalizer=None, aux_normalizer=None,
                 explicit_vad=False):
        super().__init__()
        if odim is None:
            odim = idim
        if aux_net is not None or input_normalizer is not None or aux_normalizer is not None:
            raise NotImplementedError("aux_net / normalizers are outside the hot path (SURVEY 2.1 #2)")
        if explicit_vad:
            raise NotImplementedError("explicit_vad (off in every shipped config)")
        if nmask != 1:
            raise NotImplementedError("nmask != 1 (Masking enhancer uses 1, model.py:138-145)")
        self.odim, self.nmask = odim, nmask
        self.output_resolution = output_resolution
        self.random_speaker_order = random_speaker_order
        self.num_averaged_permutations = num_averaged_permutations
        self.ts_vad = ts_vad
        self.input_normalizer, self.aux_normalizer = input_normalizer, aux_normalizer
        self.explicit_vad = explicit_vad
        self.layers, self.projs = layers, projs
        if not self.ts_vad:
            assert self.num_averaged_permutations == 1, (self.ts_vad, self.num_averaged_permutations)
        if pre_net == "RNNP":
            self.pre_net = RNNP_packed(idim=idim, elayers=1, cdim=units, hdim=odim, dropout=dropout,
                                       typ="blstm")
        elif pre_net in [None, False]:
            raise NotImplementedError("pre_net=None (every shipped config uses 'RNNP')")
        else:
            raise ValueError(pre_net)
        self.aux_net = aux_net
        self.combination = combination

        data = collections.OrderedDict()
        counter = [0]

        def put(key, value):                       # SequentialDict of net.py:562-578
            for counter[0] in range(counter[0], 100):
                k = f"{key}{counter[0]}"
                if k not in data:
                    data[k] = value
                    return
            raise RuntimeError(key)

        ts_factor = 1
        if combination == "cat":
            assert aux_net_output_size is not None, (combination, aux_net_output_size)
            first_birnn_idim = odim + aux_net_output_size
        elif combination in ["mul"]:
            first_birnn_idim = odim
        elif combination == "film":
            raise NotImplementedError(combination)          # net.py:875-878
        else:
            raise ValueError(combination)
        for l in range(layers):
            if l == layers - 1 and ts_vad is not False:
                assert 2 < ts_vad < 20, ts_vad               # net.py:607
                put("rearrange", _Marker("... spk time feature -> ... 1 time (spk feature)"))
                ts_factor = ts_vad
            put("birnn", RNNP_packed(idim=(first_birnn_idim if l == 0 else projs) * ts_factor,
                                     elayers=1, cdim=units, hdim=projs, dropout=dropout, typ="blstm"))
            if l < layers - 1:
                put("dropout", torch.nn.Dropout(p=dropout))
                put("activation", torch.nn.Tanh())
        if output_resolution == "tf":
            final_out_features = odim * nmask * ts_factor
        elif output_resolution == "t":
            final_out_features = nmask * ts_factor
        else:
            raise ValueError(output_resolution)
        put("linear", torch.nn.Linear(in_features=projs, out_features=final_out_features))
        put("rearrange", _Marker("final einops rearrange / reduce-repeat (net.py:631-659)"))
        self.post_net = Sequential(data)
        self.final_activation = torch.nn.Sigmoid()
        self._birnn_keys = [k for k in data if k.startswith("birnn")]
        self._linear_key = [k for k in data if k.startswith("linear")][0]
        if ts_vad is not False and layers < 2:
            raise NotImplementedError("ts_vad with a single post-net layer")

    @property
    def _birnns(self):
        return [self.post_net._modules[k] for k in self._birnn_keys]

    @property
    def _linear(self):
        return self.post_net._modules[self._linear_key]

    def extra_repr(self) -> str:
        return f"combination={self.combination!r},"

    # hook for hipGraph replay (tssep_amd.train.graph.GraphedStep): a callable (B, K, device) ->
    # (perm, iperm) device int32 [B, K] that replaces the host draw + H2D copy below
    permutation_source = None

    @staticmethod
    def draw_permutations(B, K):
        """One np.random.permutation(K) per batch entry, in batch order, from the GLOBAL numpy RNG
        (net.py:824-826) -> int32 [2, B, K]: the permutations and their inverses (net.py:827-831)."""
        perm = np.stack([np.random.permutation(K) for _ in range(B)])
        return np.stack([perm, np.argsort(perm, axis=-1)]).astype(np.int32)

    def _speaker_permutations(self, B, K, dev):
        if self.permutation_source is not None:
            return self.permutation_source(B, K, dev)
        # kernels want: output index of the speaker at shuffled position s == perm[b][s]
        both = torch.as_tensor(self.draw_permutations(B, K)).to(dev)              # one H2D copy
        return both[0], both[1]

    # ----------------------------------------------------------------------------------
    def logits(self, xs, aux):
        """-> (logit [B,K,T,F], embedding [B,K,1,E]).  Batched input only."""
        if xs.dim() == 2:
            lg, emb = self.logits(xs[None], [aux])
            return lg[0], emb[0]
        assert xs.dim() == 3, xs.shape
        if isinstance(aux, (tuple, list)):
            aux = torch.stack([torch.stack(list(a), 0) if isinstance(a, (tuple, list)) else a
                               for a in aux], 0)
        B, T = xs.shape[0], xs.shape[1]
        K = aux.shape[1]
        dev = xs.device
        perm_d = iperm_d = None
        if self.random_speaker_order:
            perm_d, iperm_d = self._speaker_permutations(B, K, dev)
            # shuffled aux[b, s] = aux[b, perm[b, s]]: one gather for the whole batch
            aux = torch.gather(aux, 1, perm_d.long()[..., None].expand(-1, -1, aux.shape[-1]))
        aux = aux.to(torch.float32).contiguous()
        if self.ts_vad is not False:
            assert K == self.ts_vad, (K, self.ts_vad)
        trials = self.num_averaged_permutations
        F = self.odim
        pre = self.pre_net.forward_rows(xs.reshape(B * T, xs.shape[-1]), B, T)       # [B*T, odim]
        h = Fn.condition(pre, aux, B, K, T, trials, self.combination)               # rows (b,tr,k,t)
        nb = len(self._birnns)
        # the Tanh between two post-net modules runs forward in the producer's projection store and backward in
        # the consumer's d(input) GEMM store (functional.rnnp_layer): `fold` = both ends agree on it
        prev_tanh = 0
        for l, birnn in enumerate(self._birnns):
            last = l == nb - 1
            fold = (not last) and birnn.hdim % 4 == 0 and Fn.H.FOLD_TANH
            if last and self.ts_vad is not False:
                h = birnn.forward_rows(h, B * trials, T, in_tanh=prev_tanh)        # combined input
            else:
                nxt_combined = (l == nb - 2) and self.ts_vad is not False
                h = birnn.forward_rows(h, B * trials * K, T, final_act=0 if last else 1,
                                       combine=K if nxt_combined else 0, in_tanh=prev_tanh, next_folds=fold)
                prev_tanh = (K if nxt_combined else 1) if fold else 0
        Fr = F if self.output_resolution == "tf" else 1
        logit = Fn.head(h, self._linear, perm_d, iperm_d, B, K, T, F, trials, Fr,
                        spk_rows=self.ts_vad is False)
        return logit, aux.unsqueeze(-2)

    def forward(self, xs, aux=None) -> Output:
        logit, emb = self.logits(xs, aux)
        mask = Fn.sigmoid(logit)
        u = -3
        return Output(mask=mask.unsqueeze(u), logit=logit.unsqueeze(u), embedding=emb)
