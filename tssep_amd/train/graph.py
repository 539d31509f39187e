"""hipGraph replay of the training step's device work for the launch-bound regime.

At 8 utterances per GPU (the per-GPU shard of BASELINE configs[3]) a step is ~170 kernel launches, most of
them far shorter than the time the host needs to issue one; the T-sequential recurrences aside, the GPU
waits for Python.  ``GraphedStep`` captures

    zero the flat gradient -> Model.forward -> Model.review (loss) -> backward -> join the side stream

ONCE per input signature into a hipGraph (torch.cuda.CUDAGraph: stream capture of every launch the C ABI
makes on torch's current stream, including the forked weight-gradient stream) and replays it per step.
The optimizer step (gradient all-reduce over ranks, global-norm clip + Adam: two launches) stays outside
the graph: its bias correction depends on the host-side step count and the all-reduce belongs to RCCL.

What would otherwise be frozen into the graph is routed through device memory:
  * the speaker permutations of ``random_speaker_order`` (net.py:821-831) are still drawn from the global
    ``np.random`` on the host -- one permutation per batch entry and step, the reference's consumption
    order -- and copied into a static device buffer before each replay
    (``MaskEstimator_v2.permutation_source``);
  * the W-stationary recurrences take their launch epoch from device memory (csrc/common.h);
  * inputs live in static device tensors (a batch that already IS the static tensor is not copied).

The derived weight layouts (packs, transposes: ~25 small launches) are built at the start of the captured step on the
weight-gradient stream, beside the STFT / feature kernels, not in front of each recurrence (``hip_ops.prepare_derived``).
"""
import numpy as np
import torch

from .. import hip_ops as H

_INPUT_KEYS = ("observation", "auxInput", "Input", "Vad", "vad")


class GraphedStep:
    def __init__(self, model, optimizer, warmup=2, adopt_inputs=False, zero_grad=True, max_graphs=0):
        """adopt_inputs: the tensors of the first batch of a signature BECOME the static inputs (no
        clone, no per-step copy while the caller keeps passing the same tensors -- benchmarks).
        zero_grad=False (Trainer): the graph does not clear the gradient bucket -- the caller accumulates the micro-steps
        of a virtual minibatch (tssep/train/experiment.py:135-151) and clears it at the boundaries itself; the passes a
        capture needs then leave the bucket exactly as they found it.
        max_graphs > 0: input signatures beyond that many run eagerly (a corpus of many chunk lengths must not pile up
        graphs, each with its own activation pool)."""
        self.model, self.optimizer, self.warmup = model, optimizer, int(warmup)
        self.adopt_inputs = bool(adopt_inputs)
        self.zero_grad, self.max_graphs = bool(zero_grad), int(max_graphs)
        self._graphs = {}
        self.replays = 0
        self.eager_steps = 0

    # ------------------------------------------------------------------------------ helpers
    def _tensor_keys(self, ex):
        keys = [k for k in (*_INPUT_KEYS, *self.model.loss.targets(lower=True), *self.model.loss.targets())
                if isinstance(ex.get(k), torch.Tensor)]
        return tuple(dict.fromkeys(keys))

    def _signature(self, ex):
        # (non-tensor entries that steer the computation are part of the signature: a batch with another reference
        # channel must not replay the graph captured for the first one)
        ref = ex.get("reference_channel")
        return tuple((k, tuple(ex[k].shape), ex[k].dtype) for k in self._tensor_keys(ex)) + \
            (("training", self.model.training), ("reference_channel", ref if isinstance(ref, (int, type(None))) else repr(ref)))

    def _eager(self, ex, derived_tags=None):
        if self.zero_grad:
            self.optimizer.zero_grad()
        # (while capturing: the weight packs / transposes of the whole step as a branch of the graph, hip_ops.py)
        with H.prepare_derived(list(self.model.parameters()), next(self.model.parameters()).device, only=derived_tags):
            out = self.model(ex)
            summary = self.model.review(ex, out)
            summary["loss"].backward()
        self.optimizer.bucket.sync()
        return out, summary

    def _host_side_targets(self, ex):
        """What the loss would compute on the HOST inside `review` (loss.py:122-146: the frame activity `Vad` from the
        sample activity `vad`, util/utils.stft_vad -- numpy, as in the reference) is computed here, in front of the graph,
        and enters it as one more static input: a device-to-host copy cannot be captured."""
        loss, fe = self.model.loss, getattr(self.model, "fe", None)
        tgt = getattr(loss, "target", None)
        if tgt == "Vad" and tgt not in ex and ex.get("vad") is not None and fe is not None:
            from ..util.utils import stft_vad
            dev = next(self.model.parameters()).device
            v = stft_vad(ex["vad"], fe.window_length, fe.shift, fe.fading)
            ex = dict(ex)
            ex[tgt] = torch.as_tensor(np.asarray(v) if not isinstance(v, torch.Tensor) else v, dtype=torch.float32).to(dev)
        return ex

    def usable(self, ex):
        """Whether this example can go through a graph at all: device-resident inputs, no snapshot this step (the snapshot
        branch of `review` copies to the host)."""
        return (isinstance(ex.get("observation", ex.get("Input")), torch.Tensor)
                and ex.get("observation", ex.get("Input")).is_cuda
                and isinstance(ex.get("auxInput"), torch.Tensor) and ex["auxInput"].is_cuda
                and not getattr(self.model, "create_snapshot", False) and not H.KERNEL_TIMING)

    # ------------------------------------------------------------------------------ capture
    def _capture(self, ex):
        if H.KERNEL_TIMING:
            raise RuntimeError("per-kernel event timing cannot be captured into a graph")
        me = self.model.mask_estimator
        dev = next(self.model.parameters()).device
        st = {"keys": self._tensor_keys(ex)}
        st["static"] = {k: (ex[k].detach() if self.adopt_inputs else ex[k].detach().clone()) for k in st["keys"]}
        st["rest"] = {k: v for k, v in ex.items() if k not in st["static"]}
        B = ex["auxInput"].shape[0] if isinstance(ex.get("auxInput"), torch.Tensor) and ex["auxInput"].dim() == 3 else 1
        K = ex["auxInput"].shape[-2]
        shuffled = bool(getattr(me, "random_speaker_order", False))
        if shuffled:
            st["perm_dev"] = torch.zeros(2, B, K, device=dev, dtype=torch.int32)
            st["perm_pinned"] = [torch.zeros(2, B, K, dtype=torch.int32).pin_memory() for _ in range(4)]
            st["perm_events"] = [None] * 4
            st["perm_shape"] = (B, K)
            # warm-up and capture read the static buffer; a valid permutation must be in it.  Drawn with np.random's
            # state put back: the step this capture serves draws ITS permutation in __call__, so a run through graphs
            # consumes the global RNG exactly like the eager run -- one permutation per batch entry and step (net.py:824-826)
            rng = np.random.get_state()
            st["perm_dev"].copy_(torch.as_tensor(me.draw_permutations(B, K)))
            np.random.set_state(rng)

            def source(b, k, d, _p=st["perm_dev"]):
                assert (b, k) == tuple(_p.shape[1:]), ((b, k), _p.shape)
                return _p[0], _p[1]
            st["source"] = source

        def run(tags=None):
            return self._eager({**st["rest"], **st["static"]}, tags)

        prev = me.permutation_source
        if shuffled:
            me.permutation_source = st["source"]
        # (zero_grad=False: the warm-up and capture passes accumulate into the caller's half-filled bucket -- put it back)
        kept = None if self.zero_grad else [f.clone() for f in self.optimizer.bucket.flats]
        try:
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s), H.record_derived() as seen:
                for _ in range(self.warmup):       # same stream as the capture: side stream, caches, tables exist
                    run()
            st["derived_tags"] = frozenset(seen) if self.warmup > 0 else None     # the layouts THIS signature reads
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            H.check_cluster_errors(dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                st["out"], st["summary"] = run(st["derived_tags"])
            st["graph"] = g
            st["per_example"] = self._per_example_scalars(st["summary"], st["rest"].get("dataset"))
        finally:
            me.permutation_source = prev
            if kept is not None:
                torch.cuda.synchronize(dev)
                for f, k in zip(self.optimizer.bucket.flats, kept):
                    f.copy_(k)
        return st

    def _per_example_scalars(self, summary, datasets):
        """The captured summary keys its per-utterance loss scalars by the FIRST batch's dataset names (model.py:672-686).
        -> the scalar / histogram tensors in batch order (None when the summary is not of that form), so that a replay can
        hand out a summary named after the CURRENT batch's `dataset` entries."""
        name = self.model.loss.name
        if isinstance(datasets, str):
            datasets = [datasets]
        if not isinstance(datasets, (list, tuple)):
            return None
        queues = {k: list(v) for k, v in summary.get("scalars", {}).items()}
        hqueues = {k: list(v) for k, v in summary.get("histograms", {}).items()}
        items = []
        for d in datasets:
            q, hq = queues.get(f"{d}_{name}"), hqueues.get(f"hist_{d}_{name}")
            if not q or not hq:
                return None
            items.append((q.pop(0), hq.pop(0)))
        if any(queues.values()) or any(hqueues.values()):
            return None                                   # something else was logged: keep the captured summary as it is
        return items

    def _summary_for(self, st, ex):
        """The summary of a replay: the static loss buffers of the graph (overwritten by the next replay -- read or clone
        them before the next call) under the names of THIS batch."""
        items, datasets = st.get("per_example"), ex.get("dataset")
        if isinstance(datasets, str):
            datasets = [datasets]
        if items is None or not isinstance(datasets, (list, tuple)) or len(datasets) != len(items):
            return st["summary"]
        name = self.model.loss.name
        summary = type(st["summary"])()
        summary["loss"] = st["summary"]["loss"]
        for d, (sc, hist) in zip(datasets, items):
            summary.setdefault("scalars", {}).setdefault(f"{d}_{name}", []).append(sc)
            summary.setdefault("histograms", {}).setdefault(f"hist_{d}_{name}", []).append(hist)
        return summary

    # ------------------------------------------------------------------------------- replay
    def __call__(self, ex):
        """-> (ForwardOutput, ReviewSummary) of this step; their tensors are STATIC buffers that the next
        call overwrites (lazily computed fields -- mask, stft_estimate -- are evaluated from them on
        access).  Gradients are in the optimizer's flat bucket afterwards (call ``optimizer.step()``)."""
        ex = self._host_side_targets(ex)
        if not self.zero_grad:
            # an EAGER micro-step of the same virtual minibatch may still be accumulating weight gradients into the bucket on
            # the side stream; the captured wgrad nodes += into the same views, ordered only against the launch stream, and
            # a CAPTURE saves and restores the half-filled bucket around its warm-up passes (the test of the
            # mixed routing lost the eager step's weight gradients in exactly that copy).  A stream wait, no host sync.
            self.optimizer.bucket.sync()
        sig = self._signature(ex)
        st = self._graphs.get(sig)
        if st is None:
            if self.max_graphs and len(self._graphs) >= self.max_graphs:
                self.eager_steps += 1
                return self._eager(ex)
            st = self._graphs[sig] = self._capture(ex)
        for k in st["keys"]:
            if ex[k].data_ptr() != st["static"][k].data_ptr():
                st["static"][k].copy_(ex[k], non_blocking=True)
        if "perm_dev" in st:
            slot = self.replays % len(st["perm_pinned"])
            ev = st["perm_events"][slot]
            if ev is not None:
                ev.synchronize()                                   # the copy that last read this buffer
            st["perm_pinned"][slot].copy_(torch.from_numpy(
                self.model.mask_estimator.draw_permutations(*st["perm_shape"])))
            st["perm_dev"].copy_(st["perm_pinned"][slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st["perm_events"][slot] = ev
        st["graph"].replay()
        self.replays += 1
        out = st["out"]
        return (out.fresh() if hasattr(out, "fresh") else out), self._summary_for(st, ex)
