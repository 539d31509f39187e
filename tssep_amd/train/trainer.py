"""Minimal trainer -- the part of ``padertorch.train.trainer.Trainer`` the reference relies on
(tssep/train/experiment.py:219-320): iterate the training set, ``model(ex)`` + ``model.review``,
``loss.backward()``, optimizer step every ``virtual_minibatch_size`` examples, validation +
checkpoints every ``checkpoint_trigger`` iterations (``checkpoints/ckpt_<iter>.pth`` with the
``model`` / ``optimizer`` / ``iteration`` / ``epoch`` keys, ``ckpt_latest.pth`` and
``ckpt_best_loss.pth`` links), resume from ``ckpt_latest.pth``.

Data parallel (new here; the reference refuses more than one GPU, experiment.py:181-184): one process per
GPU under torchrun.  Every rank trains on its share of the utterances (``Dataset.shard``), the flat
gradient is all-reduced (SUM, the loss is summed over the batch: model.py:669) inside ``optimizer.step()``
on the last micro-step of a virtual minibatch, parameters start from rank 0's values, and only rank 0
validates and writes checkpoints."""
import os
from pathlib import Path

import numpy as np
import torch

from .. import distributed as _dist
from ..configurable import Configurable


class _closing:
    """contextlib.closing for iterators that may not have ``close`` (lists, tuples)."""

    def __init__(self, it):
        self.it = it

    def __enter__(self):
        return self.it

    def __exit__(self, *exc):
        close = getattr(self.it, "close", None)
        if close is not None:
            close()
        return False


class Trainer(Configurable):
    def __init__(self, model, storage_dir, optimizer, summary_trigger=(1, "epoch"),
                 checkpoint_trigger=(1, "epoch"), stop_trigger=(1, "epoch"),
                 virtual_minibatch_size=1):
        self.model = model
        self.storage_dir = Path(storage_dir).expanduser().resolve()
        self.optimizer = optimizer
        self.summary_trigger = tuple(summary_trigger)
        self.checkpoint_trigger = tuple(checkpoint_trigger)
        self.stop_trigger = tuple(stop_trigger)
        self.virtual_minibatch_size = virtual_minibatch_size
        self.iteration, self.epoch = 0, 0
        self.validation_dataset = None
        self.best_loss = float("inf")
        self.history = []

    @property
    def checkpoint_dir(self):
        return self.storage_dir / "checkpoints"

    def register_validation_hook(self, validation_iterator, max_checkpoints=None, **_):
        self.validation_dataset = validation_iterator

    def test_run(self, train_iterator, validation_iterator, **_):
        """Pre-flight (experiment.py:259-292): one training and one validation example."""
        self.model.train()
        for ex in train_iterator[:1]:
            self.model.review(ex, self.model(ex))["loss"].backward()
        self.model.eval()
        with torch.no_grad():
            for ex in validation_iterator[:1]:
                self.model.review(ex, self.model(ex))
        for p in self.model.parameters():
            p.grad = None

    def _triggered(self, trigger):
        n, unit = trigger
        return unit == "iteration" and self.iteration % n == 0

    def check_device_errors(self):
        """Raise if a W-stationary recurrence launch gave up on a peer (include/tssep_hip.h: err[0]):
        everything computed since is garbage and must not be trained on or checkpointed.  One host
        sync; called where the loop synchronises anyway (summary scalars, validation, checkpoints).
        Local to this rank: use ``agree_on_failure`` where every rank passes."""
        p = next(self.model.parameters(), None)
        if p is not None and p.is_cuda:
            from .. import hip_ops
            hip_ops.check_cluster_errors(p.device)

    def agree_on_failure(self, error=None):
        """COLLECTIVE (every rank, same iteration): this rank's device error flag, or the exception
        `error` it has already caught, is exchanged (all-reduce MAX) so that all ranks raise together --
        a rank that raised alone would leave its peers in the next gradient all-reduce until the
        watchdog fires.  Also the barrier that keeps the peers until rank 0 has validated / written."""
        if error is None:
            try:
                self.check_device_errors()
            except RuntimeError as e:
                error = e
        if _dist.world_size() == 1:
            if error is not None:
                raise error
            return
        p = next(self.model.parameters(), None)
        _, who = _dist.agree_on_failure(0 if error is None else 1, device=p.device if p is not None and p.is_cuda else None)
        if error is not None:
            raise error
        if who >= 0:
            raise RuntimeError(f"rank {_dist.get_rank()}: leaving because rank {who} failed")

    @staticmethod
    def _utterances(ex):
        obs = ex.get("observation", ex.get("Input"))
        return int(obs.shape[0]) if isinstance(obs, torch.Tensor) and obs.dim() == 3 else 1

    def _log_kernel_plan(self, gemm_log, recurrence_log, chief):
        """Once per run: the kernels behind the first training step -- GEMM kernel per request (the library's own choice,
        `tssep_gemm_plan`), recurrence family and group counts -- and the runtime policy they ran under."""
        from . import runtime as _runtime
        plan = _runtime.summarise_plan(gemm_log, recurrence_log)
        self.kernel_plan = plan
        if not chief:
            return
        import json
        (self.storage_dir / "log").mkdir(parents=True, exist_ok=True)
        (self.storage_dir / "log" / "kernel_plan.json").write_text(json.dumps(
            dict(policy=_runtime.current(), **plan), indent=1))
        rec = sorted({f"{r['kernel']}/{r['direction']}" for r in plan["recurrence"]})
        print(f"tssep_amd: arithmetic {_runtime.current()['gemm_precision']}; GEMM kernels "
              f"{dict(sorted(plan['gemm'].items()))}; recurrences {rec} (log/kernel_plan.json)", flush=True)

    def validate(self):
        self.model.eval()
        losses = []
        with torch.no_grad():
            for ex in self.validation_dataset:
                losses.append(self.model.review(ex, self.model(ex))["loss"].detach())
        self.model.train()
        value = float(torch.stack(losses).mean()) if losses else float("nan")
        self.check_device_errors()
        return value

    def save_checkpoint(self, val_loss):
        self.check_device_errors()
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        path = self.checkpoint_dir / f"ckpt_{self.iteration}.pth"
        torch.save({"model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                    "optimizer": self.optimizer.state_dict(), "iteration": self.iteration,
                    "epoch": self.epoch}, path)
        links = ["ckpt_latest.pth"]
        if val_loss == val_loss and val_loss <= self.best_loss:
            self.best_loss = val_loss
            links.append("ckpt_best_loss.pth")
        for name in links:
            link = self.checkpoint_dir / name
            if link.is_symlink() or link.exists():
                link.unlink()
            os.symlink(path.name, link)
        return path

    def _chief_checkpoint(self, do_it):
        """Rank 0 validates and writes the checkpoint; EVERY rank calls this and leaves it together
        (``agree_on_failure`` is the barrier, and carries a failure of the chief to the others)."""
        error = None
        # (all ranks agree FIRST that nobody's device flag is up: a checkpoint trigger without a summary trigger must not
        # let the chief write parameters of a step some rank has flagged as garbage)
        self.agree_on_failure()
        if do_it:
            try:
                self.save_checkpoint(self.validate())
            except Exception as e:                    # noqa: BLE001 -- re-raised on every rank below
                error = e
        self.agree_on_failure(error)

    def load_checkpoint(self, path):
        sd = torch.load(str(path), map_location="cpu")
        self.model.load_state_dict(sd["model"])
        if "optimizer" in sd and self.optimizer.bucket is not None:
            self.optimizer.load_state_dict(sd["optimizer"])
        self.iteration, self.epoch = sd.get("iteration", 0), sd.get("epoch", 0)

    def train(self, train_dataset, resume=False, device=0, **_):
        self.model.to(torch.device("cuda", device) if isinstance(device, int) else device)
        self.model.train()
        self.optimizer.set_parameters(self.model.parameters())
        if resume:
            self.load_checkpoint(self.checkpoint_dir / "ckpt_latest.pth")
        rank, world = _dist.get_rank(), _dist.world_size()
        if world > 1:
            # identical replicas: rank 0's parameters (and, after a resume, its Adam moments); a speaker
            # permutation / shuffle stream of its own per rank, derived from the current state so that a
            # seeded run stays reproducible
            for t in (self.optimizer.flat_param, self.optimizer.exp_avg, self.optimizer.exp_avg_sq):
                _dist.broadcast_(t, src=0)
            np.random.seed((int(np.random.get_state()[1][0]) + 7919 * rank) & 0x7FFFFFFF)
        chief = rank == 0
        # hipGraph replay of forward + loss + backward (train/graph.py) for launch-bound micro-steps: policy
        # runtime.graph_step = "auto" (batches of at most runtime.graph_max_utterances utterances: 8 x 4 s run 6.9 instead
        # of 8.6 ms), "on", "off".  One graph per input shape, at most eight (further shapes run eagerly).  The gradient
        # bucket stays the trainer's: cleared at the boundaries of the virtual minibatch below, never by the graph; the
        # speaker permutations are drawn from np.random per step exactly as in the eager step, so a seeded run computes
        # the same losses and parameters either way (tests: bit-identical over 24 iterations).
        from .. import hip_ops as _H
        gstep = None
        on_gpu = next(self.model.parameters()).is_cuda
        bucketed = world > 1 and _H.BUCKETED_ALLREDUCE
        if bucketed:      # per-layer segments, reduced during the last micro-step's backward (runtime.bucketed_allreduce)
            self.optimizer.bucket.set_segments(_dist.layer_groups(self.model.named_parameters()))
        if _H.GRAPH_STEP != "off" and on_gpu and hasattr(getattr(self.model, "mask_estimator", None), "permutation_source"):
            from .graph import GraphedStep
            gstep = GraphedStep(self.model, self.optimizer, zero_grad=False, max_graphs=8)
        self.graph_step = gstep
        plan_logged = not on_gpu
        stop_n, stop_unit = self.stop_trigger
        assert stop_unit == "iteration", self.stop_trigger
        self.optimizer.zero_grad()
        while self.iteration < stop_n:
            # Every rank must run the same number of optimizer steps per epoch: one batch more on one rank is
            # that rank alone in an all-reduce, for ever.  `Dataset.shard` equalises the SOURCE examples; when
            # the length of the prepared dataset is known it is compared once per epoch, when a stage may drop
            # examples (`catch`) the ranks agree before every micro-step on whether all of them still have data
            # (one 16-byte all-reduce on the host-side control group, distributed.py: no GPU sync, the launch-ahead
            # of the host survives) and end the epoch together at the first one that has not.  All ranks leave at
            # the same `iteration`, so a half-filled virtual minibatch is CARRIED into the next epoch (consistent
            # partial sums on every rank), exactly as on one GPU and in the reference's trainer.
            agree_per_step = False
            if world > 1:
                try:
                    n_batches = len(train_dataset)
                except TypeError:
                    n_batches = None
                known = _dist.same_on_all_ranks(-1 if n_batches is None else 1)
                if known and n_batches is not None:
                    if not _dist.same_on_all_ranks(n_batches):
                        raise RuntimeError(f"rank {rank}: {n_batches} training batches in this epoch, another "
                                           "rank has a different count (shard BEFORE any stage that drops examples)")
                else:
                    agree_per_step = True
            batches = iter(train_dataset)
            # (closed when the epoch ends early -- stop trigger, exception -- so that the loader's producer / prefetch
            # threads stop NOW: a generator left open keeps daemon threads inside HIP calls until the interpreter tears the
            # runtime down under them -- `terminate called without an active exception` at exit, one run in six of the toy
            # experiments)
            with _closing(batches):
                while True:
                    ex = next(batches, None)
                    boundary = (self.iteration + 1) % self.virtual_minibatch_size == 0
                    if bucketed and boundary and ex is not None:
                        self.optimizer.bucket.arm()
                    if agree_per_step:
                        if not _dist.same_on_all_ranks(0 if ex is None else 1) or ex is None:
                            break
                    elif ex is None:
                        break
                    if not plan_logged:
                        # the first micro-step of a run, eagerly, with the launch logs on: which kernels the library picks
                        # for THIS model and batch (tssep_gemm_plan) and which recurrence family runs -> log/kernel_plan.json
                        _H.GEMM_LOG, _H.RECURRENCE_LOG = [], []
                        try:
                            summary = self.model.review(ex, self.model(ex))
                            summary["loss"].backward()
                        finally:
                            glog, rlog, _H.GEMM_LOG, _H.RECURRENCE_LOG = _H.GEMM_LOG, _H.RECURRENCE_LOG, None, None
                        plan_logged = True
                        self._log_kernel_plan(glog, rlog, chief)
                    elif gstep is not None and gstep.usable(ex) and (
                            _H.GRAPH_STEP == "on" or self._utterances(ex) <= _H.GRAPH_MAX_UTTERANCES):
                        _, summary = gstep(ex)           # forward + loss + backward as one replayed hipGraph
                    else:
                        summary = self.model.review(ex, self.model(ex))
                        summary["loss"].backward()
                    self.iteration += 1
                    if boundary:
                        self.optimizer.step()            # all-reduce(SUM) over ranks, clip, Adam
                        self.optimizer.zero_grad()
                    if self._triggered(self.summary_trigger):
                        self.history.append((self.iteration, float(summary["loss"])))     # host sync
                        self.agree_on_failure()
                    if self._triggered(self.checkpoint_trigger) and self.validation_dataset is not None:
                        self._chief_checkpoint(chief)
                    if self.iteration >= stop_n:
                        break
            self.epoch += 1
        self.agree_on_failure()
        if self.validation_dataset is not None:
            need = chief and not (self.checkpoint_dir / "ckpt_latest.pth").exists()
            self._chief_checkpoint(need)
        if chief:                                    # the summary scalars, for runs driven as child processes
            import json
            (self.storage_dir / "log").mkdir(parents=True, exist_ok=True)
            extra = {} if gstep is None else dict(graph_replays=gstep.replays, graph_eager_steps=gstep.eager_steps,
                                                  graphs=len(gstep._graphs))
            (self.storage_dir / "log" / "history.json").write_text(json.dumps(
                dict(iteration=self.iteration, epoch=self.epoch, loss=self.history, **extra)))
        if world > 1:
            # identical replicas fed with summed gradients must still be identical: anything else is a
            # lost or doubled all-reduce
            if not _dist.replicas_agree(self.optimizer.flat_param):
                raise RuntimeError(f"rank {rank}: the parameter replicas diverged during data-parallel training")
            torch.distributed.barrier()              # nobody leaves before rank 0 has written its files
        return self.history
