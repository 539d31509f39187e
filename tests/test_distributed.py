"""N > 1 path on CPU: world_size-2 ``gloo`` runs of the flat gradient bucket (one all-reduce(SUM)) and of
the utterance sharding.

* ``test_gradient_sum_allreduce_world2``: the bucket mechanics on a small ``nn.Linear``.
* ``test_model_level_data_parallel_equivalence``: the REAL TS-SEP step (STFT -> features -> RNNP stack ->
  mask head -> iSTFT -> LogMAE, the CPU oracle's restatement of tssep/train/model.py:465-536, 653-669) on
  two ranks, each with its shard of the utterances: because the loss is SUMMED over the batch
  (model.py:669) and utterances are independent, the all-reduced flat gradient must equal the
  single-process gradient of the concatenated batch.  (The same statement on the HIP path, two processes
  on one GPU: tests/test_gpu_modules.py::test_data_parallel_hip_step_matches_single_process.)
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tssep_amd.distributed import GradBucket, equal_shard, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(name, rank, world, port, q, *args):
    """Process target: run the named worker; an exception travels to the parent as the rank's result, so the
    test fails at once with the worker's traceback instead of waiting for a result that never comes."""
    import traceback
    os.environ["TSSEP_DIST_BACKEND"] = "gloo"      # CPU tensors: never nccl, also on a box that has a GPU
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    try:
        globals()[name](rank, world, port, q, *args)
    except BaseException:                           # noqa: BLE001
        q.put((rank, "__error__", traceback.format_exc()))


def _spawn(worker, world, *args):
    """Start `world` ranks and collect one result per rank.  The rendezvous port is found by binding to 0 and
    closing again, so another process can take it before gloo does (seen once in ~30 runs of this suite on a busy
    box): ONLY that failure (address in use) is repeated, on a fresh port; any other exception of a worker
    fails the test with the worker's traceback, and so does a rank that dies without reporting."""
    import queue
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_entry, args=(worker.__name__, r, world, port, q, *args)) for r in range(world)]
        for p in procs:
            p.start()
        res, failure = [], None
        try:
            while len(res) < world and failure is None:
                try:
                    r = q.get(timeout=5)
                except queue.Empty:
                    if all(not p.is_alive() for p in procs) and q.empty():
                        failure = "a rank died without reporting (exit codes %s)" % [p.exitcode for p in procs]
                    continue
                if len(r) == 3 and isinstance(r[1], str) and r[1] == "__error__":
                    failure = f"rank {r[0]} raised:\n{r[2]}"
                else:
                    res.append(r)
        finally:
            for p in procs:
                p.join(60 if failure is None else 1)
                if p.is_alive():
                    p.terminate()
                    p.join(10)
        if failure is None:
            return sorted(res, key=lambda r: r[0])
        if "EADDRINUSE" in failure or "Address already in use" in failure:
            continue
        raise AssertionError(failure)
    raise AssertionError(failure)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10
    bucket = GradBucket(lin.parameters())
    lo, hi = shard_range(8, rank, world)
    bucket.zero()
    lin(x[lo:hi]).abs().sum().backward()          # loss summed over the shard
    bucket.all_reduce()
    # (numpy, not torch: a tensor travels through the queue as a shared-memory handle that dies with this process)
    # (the flat buffer lays every tensor on a 256-byte boundary with zero gaps: compare the tensors, check the gaps)
    grads = torch.cat([p_.grad.flatten() for p_ in bucket.params])
    assert abs(float(bucket.flat.abs().sum()) - float(grads.abs().sum())) <= 1e-5 * float(grads.abs().sum())
    assert all((p_.grad.data_ptr() - bucket.flat.data_ptr()) % 256 == 0 for p_ in bucket.params)
    q.put((rank, grads.detach().numpy().copy(), float(bucket.global_norm())))
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_equal_shard_gives_every_rank_the_same_number_of_steps():
    for n, w in ((10, 2), (10, 4), (64, 8), (9, 8)):
        spans = [equal_shard(n, r, w) for r in range(w)]
        assert len({hi - lo for lo, hi in spans}) == 1 and spans[-1][1] <= n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    try:
        equal_shard(3, 0, 8)
    except ValueError:
        pass
    else:
        raise AssertionError("3 examples over 8 ranks must be refused")


def test_dataset_shard_stage():
    from tssep_amd import dataset as D
    ds = D.new([{"i": i} for i in range(10)]).map(lambda e: e["i"] * 10)
    assert list(ds.shard(0, 1)) == list(ds)
    a, b = list(ds.shard(0, 2)), list(ds.shard(1, 2))
    assert a == [0, 10, 20, 30, 40] and b == [50, 60, 70, 80, 90]
    assert [len(list(ds.shard(r, 4))) for r in range(4)] == [2, 2, 2, 2]
    assert [len(b_) for b_ in ds.shard(1, 2).batch(2)] == [2, 2, 1]


def _guard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    bucket = GradBucket(lin.parameters())
    out = []
    for bad_rank in (1, None):
        bucket.zero()
        lin(torch.ones(4, 6) * (rank + 1)).sum().backward()
        flag = torch.tensor([5 if rank == bad_rank else 0, 7, 0, 0], dtype=torch.int32)      # err[0]; err[1] (the epoch) never counts
        bucket.set_guard(flag)
        bucket.all_reduce()
        out.append((float(bucket.guard[0]), float(bucket.guard[1:].abs().sum()), float(bucket.flat.abs().sum())))
    q.put((rank, out, int(bucket._full.numel() - bucket.flat.numel())))
    dist.destroy_process_group()


def test_failure_flag_rides_in_the_gradient_allreduce_world2():
    """the guard slot behind the flat gradient carries every rank's "my gradient is garbage" flag through the
    SAME all-reduce(SUM): non-zero on both ranks when one rank set it, zero when none did; the gradient itself
    is untouched by the slot."""
    res = _spawn(_guard_worker, 2)
    for rank, (bad, good), slot in res:
        assert slot == 64
        assert bad[0] == 5.0 and bad[1] == 0.0, (rank, bad)
        assert good[0] == 0.0 and good[1] == 0.0, (rank, good)
        assert bad[2] == good[2] > 0


def _segments_worker(rank, world, port, q):
    from tssep_amd.distributed import layer_groups
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def build():
        torch.manual_seed(3)
        m = torch.nn.Sequential()
        m.add_module("a", torch.nn.Linear(7, 5))
        m.add_module("b", torch.nn.Linear(5, 9))
        m.add_module("c", torch.nn.Linear(9, 3))
        return m

    x = torch.randn(16, 7, generator=torch.Generator().manual_seed(10 + rank))
    res = {}
    for mode in ("flat", "all_layers", "some_layers", "unarmed"):
        m = build()
        b = GradBucket(m.parameters())
        if mode != "flat":
            b.set_segments(layer_groups(m.named_parameters()))
            assert len(b.segments) == 3 and b.segments[0][0] == 0
        b.zero()
        if mode in ("all_layers", "some_layers"):
            b.arm()
        m(x).pow(2).sum().backward()
        b.set_guard(torch.tensor([rank], dtype=torch.int32))      # rank 1 "failed": the slot must still arrive
        # the layers report in reverse order, as a backward does; "some": the first layer never reports (unfused path)
        for name in ("c", "b", "a")[:3 if mode != "some_layers" else 2]:
            b.notify(list(getattr(m, name).parameters()))
        order = list(b._reduced)
        b.all_reduce()
        res[mode] = (b._full.clone().numpy(), order, list(getattr(b, "last_reduction_order", []) or []))
    q.put((rank, res))
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_the_flat_one_world2():
    """per-layer segments reduced as their layer reports (reverse layer order) + the rest at the end give
    the flat all-reduce's buffer BIT FOR BIT -- gradient, alignment gaps and the guard slot -- whether all layers report,
    only some do, or the bucket was never armed (then nothing is reduced early)."""
    res = _spawn(_segments_worker, 2)
    for rank, r in res:
        flat = r["flat"][0]
        assert float(np.abs(flat).sum()) > 0 and flat[-64] == 1.0          # guard slot: 0 + 1
        for mode in ("all_layers", "some_layers", "unarmed"):
            assert np.array_equal(r[mode][0], flat), (rank, mode)
        assert r["all_layers"][1] == [2, 1, 0] and r["some_layers"][1] == [2, 1] and r["unarmed"][1] == []
        assert r["flat"][1] == []
    assert np.array_equal(res[0][1]["flat"][0], res[1][1]["flat"][0])


def test_layer_groups_of_the_mask_estimator_are_its_five_layers():
    """`distributed.layer_groups` on the real TS-SEP mask estimator (net.py:809-986): pre-net, birnn0 / 1 / 2 (nn.LSTM +
    projection = ten tensors each, the unit one RNNP layer's backward completes together) and the logit layer; the groups are
    contiguous in the flat layout, cover every parameter once, and an un-armed bucket never reduces early."""
    from tssep_amd.distributed import layer_groups
    from tssep_amd.train import net
    m = net.MaskEstimator_v2(idim=553, odim=513, units=12, projs=16, combination="mul", aux_net_output_size=513, ts_vad=4,
                             output_resolution="tf")
    groups = layer_groups(m.named_parameters())
    assert [len(g) for g in groups] == [10, 10, 10, 10, 2]
    assert sum(len(g) for g in groups) == len(list(m.parameters()))
    b = GradBucket(m.parameters())
    b.set_segments(groups)
    assert len(b.segments) == 5 and b.segments[0][0] == 0 and b.segments[-1][1] == b.flat.numel()
    assert all(a[1] == c[0] for a, c in zip(b.segments, b.segments[1:]))
    b.notify(groups[-1])                       # not armed (no process group, not the last micro-step): nothing happens
    assert b._reduced == [] and b._armed is False
    b.arm()
    assert b._armed is False                   # a single process never arms


def test_gradient_sum_allreduce_world2():
    res = _spawn(_worker, 2)
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10
    lin(x).abs().sum().backward()
    ref = torch.cat([lin.weight.grad.flatten(), lin.bias.grad.flatten()])
    for _, flat, norm in res:
        torch.testing.assert_close(torch.as_tensor(flat), ref)
        assert abs(norm - float(ref.norm())) < 1e-4


# ---- the real model --------------------------------------------------------------------------------
_CFG = dict(odim=33, combination="mul", ts_vad=4, output_resolution="tf", random_speaker_order=False)
_DIMS = dict(idim=33, odim=33, units=6, projs=7, combination="mul", aux_size=33, ts_vad=4)
_B, _K, _N = 4, 4, 400


def _toy_batch():
    rng = np.random.RandomState(7)
    tgt = 0.1 * rng.randn(_B, _K, _N).astype(np.float32)
    obs = tgt.sum(1, keepdims=True) + 0.01 * rng.rand(_B, 1, _N).astype(np.float32)
    aux = rng.rand(_B, _K, 33).astype(np.float32)
    return [torch.as_tensor(a) for a in (obs, aux, tgt)]


def _oracle_params():
    from oracle import model as omodel
    torch.manual_seed(3)
    p = omodel.init_mask_estimator_params(**_DIMS)
    return [v.requires_grad_() for v in p.values()], p


def _oracle_loss(p, obs, aux, tgt):
    from oracle import model as omodel
    # log1p features only: the MFCC dB floor is taken over the LOCAL batch, which couples
    # the utterances of a shard on purpose -- DP then equals per-shard reference runs, not one big batch
    return omodel.forward_loss(p, obs, aux, tgt, cfg=_CFG, mfcc=False, size=64, shift=16)["loss"].sum()


def _model_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    leaves, p = _oracle_params()
    bucket = GradBucket(leaves)
    lo, hi = shard_range(_B, rank, world)
    obs, aux, tgt = (t[lo:hi] for t in _toy_batch())
    bucket.zero()
    loss = _oracle_loss(p, obs, aux, tgt)
    loss.backward()
    bucket.all_reduce()
    q.put((rank, torch.cat([p_.grad.flatten() for p_ in bucket.params]).detach().numpy().copy(), float(loss)))
    dist.destroy_process_group()


def test_model_level_data_parallel_equivalence():
    res = _spawn(_model_worker, 2)
    leaves, p = _oracle_params()
    loss = _oracle_loss(p, *_toy_batch())
    loss.backward()
    ref = torch.cat([v.grad.flatten() for v in leaves])
    assert float(ref.abs().max()) > 1e-4
    for _, flat, _ in res:
        torch.testing.assert_close(torch.as_tensor(flat), ref, rtol=2e-4, atol=2e-6 * float(ref.abs().max()) + 1e-9)
    torch.testing.assert_close(torch.tensor(sum(r[2] for r in res)), torch.tensor(float(loss.detach())), rtol=1e-5, atol=1e-6)


def _replica_worker(rank, world, port, q):
    from tssep_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    assert D.init_from_env() == (rank, world, rank) and D.get_rank() == rank and D.world_size() == world
    t = torch.arange(5.0) + 10 * rank
    same_before = D.replicas_agree(t)
    D.broadcast_(t, src=0)
    q.put((rank, same_before, D.replicas_agree(t), t.tolist()))
    dist.destroy_process_group()


def test_broadcast_and_replica_check_world2():
    res = _spawn(_replica_worker, 2)
    for rank, before, after, values in res:
        assert before is False and after is True and values == [0.0, 1.0, 2.0, 3.0, 4.0]


def _failure_worker(rank, world, port, q):
    from tssep_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env()
    fine = D.agree_on_failure(0)
    one_failed = D.agree_on_failure(3 if rank == 1 else 0)
    q.put((rank, fine, one_failed, D.same_on_all_ranks(17), D.same_on_all_ranks(5 + rank)))
    dist.destroy_process_group()


def test_failure_flag_and_batch_count_exchange_world2():
    """a rank that raises alone leaves its peers in the next all-reduce until the watchdog fires;
    every rank must learn of a failure (and of unequal batch counts) at the same program point."""
    for rank, fine, one_failed, same, differ in _spawn(_failure_worker, 2):
        assert fine == (0, -1) and one_failed == (3, 1) and same is True and differ is False


def _trainer_failure_worker(rank, world, port, q):
    """The Trainer's collective check: rank 1's device flag is set -> BOTH ranks raise."""
    from tssep_amd import distributed as D
    from tssep_amd.train.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env()
    t = Trainer(model=torch.nn.Linear(2, 2), storage_dir="/tmp", optimizer=None)
    t.agree_on_failure()                                    # nobody failed: returns
    if rank == 1:
        def boom():
            raise RuntimeError("cluster recurrence kernel timed out waiting for a peer (code 7)")
        t.check_device_errors = boom
    try:
        t.agree_on_failure()
        msg = "returned"
    except RuntimeError as e:
        msg = str(e)
    # the chief's checkpoint block: rank 0 fails inside it, rank 1 (which does not enter it) leaves too
    t.check_device_errors = lambda: None
    t.save_checkpoint = lambda v: (_ for _ in ()).throw(OSError("disk full"))
    t.validate = lambda: 0.0
    try:
        t._chief_checkpoint(rank == 0)
        msg2 = "returned"
    except (RuntimeError, OSError) as e:
        msg2 = str(e)
    q.put((rank, msg, msg2))
    dist.destroy_process_group()


def test_trainer_ranks_fail_together_world2():
    res = _spawn(_trainer_failure_worker, 2)
    assert "rank 1 failed" in res[0][1] and "timed out" in res[1][1]
    assert "disk full" in res[0][2] and "rank 0 failed" in res[1][2]


def test_bucket_grads_are_views():
    lin = torch.nn.Linear(4, 2)
    b = GradBucket(lin.parameters())
    lin(torch.ones(3, 4)).sum().backward()
    assert lin.weight.grad.data_ptr() == b.flat.data_ptr()
    assert float(b.flat.abs().sum()) > 0
    b.zero()
    assert float(lin.weight.grad.abs().sum()) == 0


class _Ragged:
    """A prepared dataset whose length is unknown (a `catch` stage may drop examples) and differs per rank."""
    def __init__(self, n):
        self.n = n

    def __len__(self):
        raise TypeError("the length of a dataset that may drop examples is unknown")

    def __iter__(self):
        return iter([torch.full((2, 3), float(i + 1)) for i in range(self.n)])


class _ToyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(3, 1)

    def forward(self, ex):
        return self.lin(ex)

    def review(self, ex, out):
        return dict(loss=out.sum())


class _Sgd:
    """Stands in for the fused optimizer (which needs the GPU library): flat bucket + all-reduce + step count."""
    def __init__(self):
        self.steps = 0
        self.micro = []

    def set_parameters(self, params):
        self.bucket = GradBucket(list(params))
        self.flat_param = torch.cat([p.data.reshape(-1) for p in self.bucket.params])
        self.exp_avg = torch.zeros(1)
        self.exp_avg_sq = torch.zeros(1)

    def zero_grad(self):
        self.bucket.zero()

    def step(self):
        self.bucket.all_reduce()
        self.steps += 1
        # micro-batches in this step: d(loss)/d(bias) of _ToyModel is 2 rows per micro-batch and rank
        self.micro.append(int(round(float(self.bucket.params[-1].grad.sum()) / (2 * dist.get_world_size()))))


def _ragged_worker(rank, world, port, q, tmp):
    from tssep_amd import distributed as D
    from tssep_amd.train.trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env()
    torch.manual_seed(0)
    t = Trainer(model=_ToyModel(), storage_dir=tmp, optimizer=_Sgd(), stop_trigger=(7, "iteration"),
                summary_trigger=(100, "iteration"), checkpoint_trigger=(100, "iteration"), virtual_minibatch_size=2)
    t.model.to = lambda *_a, **_k: t.model                    # CPU run of the loop (the product path is GPU only)
    t.train(_Ragged(5 if rank == 0 else 3), device="cpu")     # rank 1 runs out of batches two steps early
    q.put((rank, t.iteration, t.epoch, t.optimizer.steps, t.optimizer.micro))
    dist.destroy_process_group()


def test_ranks_with_different_batch_counts_end_the_epoch_together(tmp_path):
    """`equal_shard` equalises source examples only; a later stage that drops examples gives the ranks
    different batch counts and the one with more waits for ever in the all-reduce.  With an unknown length the
    ranks agree per micro-step: both stop each epoch after rank 1's 3 batches, so both take the same number of
    optimizer steps and reach the stop trigger together.  ADVICE r3: the half-filled virtual minibatch of 2 is
    CARRIED over the epoch boundary (as on one GPU and in the reference), so 7 micro-steps are exactly 3 steps
    and every step saw two micro-batches."""
    res = _spawn(_ragged_worker, 2, str(tmp_path))
    assert res[0][1:] == res[1][1:], res
    assert res[0][1] == 7 and res[0][3] == 3 and res[0][4] == [2, 2, 2], res
