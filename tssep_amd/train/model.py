"""Model -- drop-in for tssep/train/model.py: ``forward`` (:465-536) and ``review`` (:653-752: the loss part on the
HIP kernels, the snapshot branch as plain tensor glue); the dataset helpers follow :182-370 stage by stage on
``tssep_amd.dataset`` (lazy map / shuffle / batch / threaded prefetch, pinned asynchronous H2D)."""
import dataclasses
import functools
import os

import numpy as np
import torch

from .. import functional as Fn
from ..configurable import Configurable
from . import enhancer as _enh, feature_extractor as _fe, loss as _loss, net as _net
from ..data import DummyReader


class ReviewSummary(dict):
    """The few members of padertorch's ReviewSummary the reference's review touches."""

    def add_to_loss(self, value):
        self["loss"] = self["loss"] + value if "loss" in self else value

    def add_scalar(self, name, *value):
        self.setdefault("scalars", {}).setdefault(name, []).extend(
            float(v) if not isinstance(v, torch.Tensor) else v.detach() for v in value)

    def add_histogram(self, name, values):
        self.setdefault("histograms", {}).setdefault(name, []).append(
            values.detach() if isinstance(values, torch.Tensor) else values)

    # ---- snapshot members (model.py:692-752 calls them when ``create_snapshot`` is set).  padertorch is not in this
    # image: the reference's versions render colour images for tensorboard; these keep the DATA such an image shows --
    # the selected batch entry, rearranged as asked -- so that a writer can render it (parity of the rendering: unpinned)
    @staticmethod
    def _select(signal, batch_first):
        signal = signal.detach() if isinstance(signal, torch.Tensor) else torch.as_tensor(np.asarray(signal))
        if batch_first is True:
            return signal[0]
        if batch_first is False:
            return signal[:, 0]
        return signal

    def add_audio(self, name, signal, sampling_rate=16000, batch_first=None, normalize=True):
        audio = self._select(signal, batch_first).to(torch.float32).reshape(-1)
        if normalize:
            audio = audio * (0.95 / torch.clamp(audio.abs().max(), min=1e-30))
        self.setdefault("audios", {})[name] = (audio.cpu(), sampling_rate)

    def _image(self, name, signal, batch_first, rearrange):
        signal = self._select(signal, batch_first)
        if rearrange is not None:
            import einops
            signal = einops.rearrange(signal, rearrange)
        assert signal.dim() == 2, (name, tuple(signal.shape))
        self.setdefault("images", {})[name] = signal.transpose(0, 1).flip(0).cpu()      # [freq (top = high), time]

    def add_stft_image(self, name, signal, batch_first=None, rearrange=None):
        signal = signal.detach() if isinstance(signal, torch.Tensor) else torch.as_tensor(np.asarray(signal))
        mag = signal.abs().to(torch.float32)
        mag = torch.log10(torch.clamp(mag / torch.clamp(mag.max(), min=1e-30), min=1e-3)) / 3 + 1     # 60 dB below the peak -> [0, 1]
        self._image(name, torch.clamp(mag, 0, 1), batch_first, rearrange)

    def add_mask_image(self, name, mask, batch_first=None, rearrange=None):
        mask = mask.detach() if isinstance(mask, torch.Tensor) else torch.as_tensor(np.asarray(mask))
        self._image(name, torch.clamp(mask.to(torch.float32), 0, 1), batch_first, rearrange)


class Model(Configurable, torch.nn.Module):
    @classmethod
    def finalize_dogmatic_config(cls, config):              # model.py:71-149
        config.setdefault("fe", None)
        if config.get("fe") is None:
            config["fe"] = dict(factory=_fe.Log1pMaxNormAbsSTFT, size=1024, shift=256, window="hann")
        if config.get("reader") is None:
            config["reader"] = dict(factory=DummyReader)
        if config.get("enhancer") is None:
            config["enhancer"] = dict(factory=_enh.Masking)
        if config.get("loss") is None:
            config["loss"] = dict(factory=_loss.LogMAE)
        if config.get("mask_estimator") is None:
            fe = Configurable.from_config(_fe.Log1pMaxNormAbsSTFT.get_config(
                {k: v for k, v in config["fe"].items() if k != "factory"})) \
                if not hasattr(config["fe"], "output_size") else config["fe"]
            config["mask_estimator"] = dict(factory=_net.MaskEstimator_v2, idim=fe.output_size,
                                            odim=fe.frequencies, nmask=1)

    def __init__(self, fe=None, reader=None, mask_estimator=None, enhancer=None, loss=None):
        super().__init__()
        self.fe = fe
        self.reader = reader
        self.mask_estimator = mask_estimator
        self.enhancer = enhancer
        self.loss = loss
        self.create_snapshot = False

    # ------------------------------------------------------------------ data helpers
    def example_to_device(self, ex, device):                # model.py:166-180
        for k in {"Input", "observation", "auxInput", *self.loss.targets(lower=True),
                  *self.loss.targets()}:
            if k in ex:
                v = ex[k]
                if isinstance(v, np.ndarray):
                    v = torch.as_tensor(v)
                if isinstance(v, torch.Tensor):
                    ex[k] = v.to(device)
        return ex

    def collate_fn(self, exs):                              # model.py:339-370
        ex = {k: [e[k] for e in exs] for k in exs[0]}
        ex["reference_channel"] = exs[0]["reference_channel"]
        for k in ("observation", "auxInput", "vad", "Vad", *self.loss.targets(lower=True)):
            if k in ex:
                ex[k] = torch.as_tensor(np.stack([np.asarray(v) for v in ex[k]]))
        return ex

    def prepare_dataset(self, dataset_name, device, training=False, review=True, batch_size=None,
                        prefetch=True, reader=None, sort=False, verbose=False, load_keys=None):
        """model.py:182-337, same stages in the same order: reader -> prepare -> shuffle(reshuffle) when
        training -> batch + collate -> threaded prefetch -> device.  The device stage is a
        ``DeviceLoader`` (pinned staging buffers, asynchronous H2D on a copy stream, 2 batches ahead)
        instead of ``example_to_device`` in a prefetch thread."""
        from .. import dataset as D
        if reader is None:
            reader = self.reader
        pre_load_apply = None
        if sort:                                            # longest first: an OOM shows up early
            def get_num_samples(ex):
                if "end" in ex and "start" in ex:
                    return ex["end"] - ex["start"]
                n = ex["num_samples"]
                if isinstance(n, dict):
                    return n["observation"] if "observation" in n else max(n["original_source"])
                return n

            def pre_load_apply(ds):
                return D.new(ds).copy(freeze=True).sort(get_num_samples, reverse=True)
        if load_keys is None:
            load_keys = ["observation", *self.loss.targets(lower=True)]
        ds = D.new(reader(dataset_name, pre_load_apply=pre_load_apply, load_keys=load_keys))
        if training:            # data parallel: one process per GPU, each trains on its share of the utterances
            from .. import distributed as _dist
            ds = ds.shard(_dist.get_rank(), _dist.world_size())

        targets = tuple(self.loss.targets())
        passthrough = ("example_id", "dataset", "gender", "auxInput", "vad", "framewise_embeddings",
                       "framewise_embeddings_stride")

        def pick_target(audio, name):
            """The loss target `name` out of the loaded audio (model.py:254-283): signals are stored
            under the lower-case key (multi-channel ones reduced to the reference channel), the
            frame-level 'Vad' under its own name; (None, None) when the reader did not load it."""
            key = name.lower()
            if key in audio:
                value = audio[key]
                if isinstance(value, np.ndarray) and value.ndim == 3:
                    value = value[:, 0]
                return key, value
            if name == "Vad" and name in audio:
                return name, audio[name]
            return None, None

        def prepare(ex):
            audio = ex.get("audio_data")
            r = {"reference_channel": 0}
            if audio is not None and "observation" in audio:
                r["observation"] = audio["observation"]
            elif "Input" in ex:
                r["Input"] = ex["Input"]
            else:
                raise KeyError("observation")
            for name in targets:
                key, value = pick_target(audio or {}, name)
                if key is not None:
                    r[key] = value
                elif audio is None or name == "Vad":
                    if self.training:                      # a missing entry only passes outside training
                        raise KeyError(name if audio is not None else "audio_data")
                elif review:
                    raise Exception(
                        f"The reader did not load {name.lower()!r} although the loss asks for it; load it, "
                        "or call prepare_dataset(review=False) when the target is not needed.")
            r.update((k, ex[k]) for k in passthrough if k in ex)
            if verbose:
                r["verbose"] = ex
            return r

        ds = ds.map(prepare)
        if training and not sort:
            ds = ds.shuffle(reshuffle=True)
        if batch_size is not None:
            ds = ds.batch(batch_size).map(self.collate_fn)
        if prefetch:
            threads = int(os.environ.get("SLURM_CPUS_PER_TASK", 6))
            ds = ds.prefetch(threads, threads * 2, catch_filter_exception=True)
        elif training:
            ds = ds.catch()
        if device is not None:
            keys = ("Input", "observation", "auxInput", "vad", *self.loss.targets(lower=True),
                    *self.loss.targets())
            if batch_size is None:                          # single examples: convert, then copy
                ds = ds.map(lambda ex: {k: (torch.as_tensor(v) if isinstance(v, np.ndarray)
                                            and v.dtype != object else v) for k, v in ex.items()})
            if prefetch:
                return D.DeviceLoader(ds, device, keys, depth=2)
            ds = ds.map(functools.partial(self.example_to_device, device=device))
        return ds

    def prepare_train_dataset(self, device, batch_size=None, prefetch=True, reader=None, sort=False):
        return self.prepare_dataset(self.reader.train_dataset_name if reader is None
                                    else reader.train_dataset_name, device, training=True,
                                    batch_size=batch_size, prefetch=prefetch, reader=reader, sort=sort)

    def prepare_validate_dataset(self, device, batch_size=None, prefetch=True, reader=None, sort=False):
        return self.prepare_dataset(self.reader.validate_dataset_name if reader is None
                                    else reader.validate_dataset_name, device, training=False,
                                    batch_size=batch_size, prefetch=prefetch, reader=reader, sort=sort)

    # ------------------------------------------------------------------------ forward
    class ForwardOutput:
        """The fields of the reference's dataclass (model.py:454-463): mask, logit, embedding,
        stft_estimate, time_estimate, vad_mask, vad_logit.  ``mask`` and ``stft_estimate`` are computed on
        first access: the training step needs neither -- the loss path runs sigmoid, masking and the
        inverse STFT as one fused kernel from ``logit`` (functional.mask_istft) -- so the two largest
        tensors of the step ([B,K,T,F] fp32 and complex64) are only materialised for callers that look
        at them (snapshots, evaluation, custom losses).  Accessing them gives exactly the tensors the
        eager chain would have produced, connected to the same autograd graph."""

        def __init__(self, mask=None, logit=None, embedding=None, stft_estimate=None, time_estimate=None,
                     vad_mask=None, vad_logit=None, _lazy=None):
            self._mask, self._stft_estimate, self._lazy, self._lazy0 = mask, stft_estimate, _lazy, _lazy
            self.logit, self.embedding, self.time_estimate = logit, embedding, time_estimate
            self.vad_mask, self.vad_logit = vad_mask, vad_logit

        def _materialise(self):
            if self._lazy is not None:
                lazy, self._lazy = self._lazy, None
                mask, est = lazy()
                if self._mask is None:
                    self._mask = mask
                if self._stft_estimate is None:
                    self._stft_estimate = est

        @property
        def mask(self):
            if self._mask is None:
                self._materialise()
            return self._mask

        @mask.setter
        def mask(self, value):
            self._mask = value

        @property
        def stft_estimate(self):
            if self._stft_estimate is None:
                self._materialise()
            return self._stft_estimate

        @stft_estimate.setter
        def stft_estimate(self, value):
            self._stft_estimate = value

        @property
        def materialised(self):
            return self._lazy is None

        def fresh(self):
            """A new view of the same tensors with mask / stft_estimate un-materialised again (hipGraph
            replay: the static logit holds new values after every replay)."""
            import copy
            other = copy.copy(self)
            if other._lazy0 is not None:
                other._mask = other._stft_estimate = None
                other._lazy = other._lazy0
            return other

        def __repr__(self):
            def sh(t):
                return None if t is None else tuple(t.shape)
            return (f"ForwardOutput(mask={'<lazy>' if self._mask is None and self._lazy else sh(self._mask)}, "
                    f"logit={sh(self.logit)}, embedding={sh(self.embedding)}, "
                    f"stft_estimate={'<lazy>' if self._stft_estimate is None and self._lazy else sh(self._stft_estimate)}, "
                    f"time_estimate={sh(self.time_estimate)})")

    def forward(self, ex, feature_transform=None) -> "Model.ForwardOutput":
        ex["AuxInput"] = [a for a in ex["auxInput"]]
        if not isinstance(ex["reference_channel"], int):
            raise NotImplementedError(type(ex["reference_channel"]), ex["reference_channel"])
        ref = ex["reference_channel"]
        if "Input" in ex:
            pass
        elif "Observation" in ex:
            ex["Input"] = self.fe.stft_to_feature(ex["Observation"][..., ref, :, :]).to(torch.float32)
        elif hasattr(self.fe, "stft"):
            ex["Observation"] = self.fe.stft(ex["observation"])
            ex["Input"] = self.fe.stft_to_feature(ex["Observation"][..., ref, :, :]).to(torch.float32)
        else:
            ex["Input"] = self.fe(ex["observation"][..., ref, :]).to(torch.float32)
        if feature_transform is not None:
            ex["Input"] = feature_transform(ex["Input"])
        ex = self.reader.data_hooks.pre_net(ex)

        aux = ex["auxInput"] if isinstance(ex["auxInput"], torch.Tensor) else ex["AuxInput"]
        batched = ex["Input"].dim() == 3
        logit, emb = self.mask_estimator.logits(ex["Input"], aux)
        logit4 = logit if batched else logit[None]
        out = self.ForwardOutput(logit=logit.unsqueeze(-3), embedding=emb)
        if "Observation" in ex and isinstance(self.enhancer, _enh.Masking):
            obs = ex["Observation"][..., ref, :, :]
            obs3 = (obs if batched else obs[None]).contiguous()
            out._fusable = (logit4, obs3, batched)           # review() runs the fused chain from here

            def lazy():                                     # sigmoid (net.py:983) + Masking (enhancer.py:98-100)
                mask, est = Fn.mask_head(logit4, obs3)
                if not batched:
                    mask, est = mask[0], est[0]
                return mask.unsqueeze(-3), est
        elif "Observation" in ex:
            def lazy():
                mask = Fn.sigmoid(logit4)
                mask = (mask if batched else mask[0]).unsqueeze(-3)
                return mask, self.enhancer(mask, ex, self)
        else:
            assert isinstance(self.loss, _loss.VADSigmoidBCE), type(self.loss)

            def lazy():
                mask = Fn.sigmoid(logit4)
                return (mask if batched else mask[0]).unsqueeze(-3), None
        out._lazy = out._lazy0 = lazy
        return out

    # ------------------------------------------------------------------------- review
    def review(self, ex, out: "Model.ForwardOutput"):
        summary = ReviewSummary()
        if hasattr(self.fe, "istft") and "observation" in ex:          # model.py:661-664
            n = ex["observation"].shape[-1]
            fus = getattr(out, "_fusable", None)
            if fus is not None and not out.materialised and hasattr(self.fe, "masked_istft"):
                # nobody looked at mask / stft_estimate: sigmoid -> masking -> istft in one kernel, with
                # the |estimate - target| partial sums of a time-domain loss on the side
                logit4, obs3, batched = fus
                tgt = ex.get(getattr(self.loss, "target", None)) if isinstance(self.loss, _loss.TimeDomain) else None
                if isinstance(tgt, torch.Tensor):
                    tgt = tgt if batched else tgt[None]
                else:
                    tgt = None
                te = self.fe.masked_istft(logit4, obs3, num_samples=n, target=tgt)
                out.time_estimate = te if batched else te[0]
            else:
                out.time_estimate = self.fe.istft(out.stft_estimate, num_samples=n)
        loss_value = self.loss.from_ex_out(ex, out, self, summary)
        # (`loss_weight`, checker-only, not a key of the reference: per-utterance weights of the batch sum -- with 0 / 1
        # weights the backward of a FULL batch, on the kernels a full batch selects, yields the gradient of a slice of it,
        # which is what bench.py compares with the CPU oracle's backward over that slice)
        w = ex.get("loss_weight")
        summary.add_to_loss(loss_value.sum() if w is None else (loss_value * w.to(loss_value)).sum())      # model.py:669
        with torch.no_grad():
            name = self.loss.name
            if loss_value.ndim == 0:                                      # model.py:672-679
                summary.add_scalar(f'{ex["dataset"]}_{name}', loss_value)
                summary.add_histogram(f'hist_{ex["dataset"]}_{name}', loss_value)
            elif loss_value.ndim == 1:                                    # model.py:680-686
                for dataset_name, lv in zip(ex["dataset"], loss_value):
                    summary.add_scalar(f"{dataset_name}_{name}", lv)
                    summary.add_histogram(f"hist_{dataset_name}_{name}", lv)
            else:
                raise NotImplementedError(loss_value.ndim, loss_value.shape)
            has_batch_dim = loss_value.ndim == 1
            if self.create_snapshot:                                      # model.py:692-752
                enh = self.enhancer.name
                if out.time_estimate is not None:
                    for i, e in enumerate(out.time_estimate):
                        summary.add_audio(f"{enh}_audio_est_{i}", e, sampling_rate=self.reader.sample_rate, batch_first=True)
                if "observation" in ex:
                    summary.add_audio(f"{enh}_audio_observation", ex["observation"][ex["reference_channel"]],
                                      sampling_rate=self.reader.sample_rate, batch_first=True)
                if "Observation" in ex:
                    summary.add_stft_image(f"{enh}_Observation", ex["Observation"][ex["reference_channel"]], batch_first=True)
                dataset_name = ex["dataset"][0] if has_batch_dim else ex["dataset"]
                masks = out.mask                      # (materialises mask / stft_estimate: the snapshot looks at them)
                summary.add_mask_image(f"{dataset_name}_{enh}_mask", masks[0] if has_batch_dim else masks,
                                       rearrange="spk mask time freq -> time (spk mask freq)", batch_first=None)
                if out.stft_estimate is not None:
                    summary.add_stft_image(f"{enh}_stft_estimate", out.stft_estimate,
                                           rearrange="... spk time freq -> ... time (spk freq)", batch_first=True)
                for target_name in self.loss.targets(upper=True):
                    if target_name == "Vad":
                        import einops
                        target = einops.repeat(torch.as_tensor(ex[target_name]), "... -> ... freq", freq=40)
                    elif target_name in ex:
                        target = ex[target_name]
                    else:
                        target = self.fe.stft(ex[target_name.lower()])
                    summary.add_stft_image(f"{enh}_target_{target_name}", target,
                                           rearrange="... spk time freq -> ... time (spk freq)", batch_first=True)
                self.loss.update_summary(summary, ex, out, self)
        return summary
