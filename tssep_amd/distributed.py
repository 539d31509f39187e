"""Data-parallel gradient exchange: one process per GPU, one flat fp32 bucket, one
all-reduce(SUM) over RCCL/xGMI per optimizer step.

The reference has no distributed path (tssep/train/experiment.py:181-184 refuses >1 GPU).  Its
loss is SUMMED over the batch (tssep/train/model.py:669), so gradients of the shards are summed,
not averaged, to equal a single-process run over the concatenated batch.  Utterances are
independent in forward and backward, so the data path needs no other collective.
"""
import weakref

import torch
import torch.distributed as dist


FLAT_ALIGN = 64      # elements: every tensor of a flat parameter / gradient buffer starts on a 256-byte boundary


def flat_offsets(params, align=FLAT_ALIGN):
    """-> (offset of each tensor, total length) of the flat buffers the optimizer and the gradient bucket lay the
    parameters out in.  Each tensor starts on a multiple of `align` elements: the GEMMs read weights with 16-byte loads
    straight out of the flat buffer (a bias of 513 elements in front of a weight used to knock it off that alignment,
    and every use of the weight paid a padded copy: hip_ops.rows_view).  The gaps stay zero in all buffers (gradient 0,
    moments 0, update 0), so norms, all-reduce and Adam are those of the parameters."""
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += -(-p.numel() // align) * align
    return offsets, off


def layer_groups(named_parameters):
    """-> lists of parameters, one per layer in forward (= registration) order, for GradBucket.set_segments: the two
    modules of an RNNP layer (`<layer>.net.0` nn.LSTM + `<layer>.net.1` projection, tssep/train/rnnp.py:88-96) are one
    group -- its backward completes all ten tensors together --, every other module is its own."""
    groups, last = [], None
    for name, p in named_parameters:
        if not p.requires_grad:
            continue
        key = name.split(".net.")[0] if ".net." in name else name.rsplit(".", 1)[0]
        if key != last:
            groups.append([])
            last = key
        groups[-1].append(p)
    return groups


GUARD_SLOT = 64          # floats (256 bytes) behind the gradient: the failure flag that rides in the all-reduce


class GradBucket:
    """Flat view over all parameter gradients: grads live inside ONE contiguous buffer, so the
    all-reduce needs no pack/unpack copies.

    ``replicas`` > 1 keeps one extra flat buffer per concurrently running micro-batch (gradient
    accumulation over micro-batches on separate HIP streams, the reference's
    ``virtual_minibatch_size``, tssep/train/experiment.py:135-151): every micro-batch accumulates
    into its own buffer (no read-modify-write races between streams) and ``reduce_replicas`` folds
    them into buffer 0 = ``p.grad``."""

    def __init__(self, params, replicas=1):
        self.params = [p for p in params if p.requires_grad]
        offsets, n = flat_offsets(self.params)
        dev, dt = self.params[0].device, self.params[0].dtype
        # GUARD_SLOT floats behind buffer 0 travel with the all-reduce: element 0 is this rank's "my gradient is garbage"
        # flag (a W-stationary recurrence launch gave up, err[0] of include/tssep_hip.h) as a float -- after the SUM it
        # is non-zero on EVERY rank when any rank set it, which is what the guarded Adam launch tests (a
        # rank-local flag let the peers apply a gradient that already contained the failed rank's garbage)
        self._full = torch.zeros(n + GUARD_SLOT, device=dev, dtype=dt)
        self.guard = self._full[n:]
        self.flats = [self._full[:n]] + [torch.zeros(n, device=dev, dtype=dt) for _ in range(replicas - 1)]
        self.flat = self.flats[0]
        for p, off in zip(self.params, offsets):
            views = [f[off:off + p.numel()].view_as(p) for f in self.flats]
            p.grad = views[0]
            # the HIP backward may accumulate straight into these views (off the autograd engine,
            # on a side stream): see tssep_amd.functional._grad_sink
            p._tssep_grad_sinks = views
            p._tssep_grad_sink = views[0]
            p._tssep_bucket = weakref.ref(self)
        self._offsets, self._n = offsets, n
        self.segments = []          # [(start, end)] element ranges of `_full`, one per layer, in forward order (set_segments)
        self._seg_of = {}           # id(parameter) -> segment index
        self._seg_params = []       # per segment: ids of its parameters
        self._armed = False         # only the LAST micro-step of a virtual minibatch may reduce early (arm)
        self._reported = []         # per segment: ids reported complete in this backward
        self._reduced = []          # segment indices already all-reduced (in launch order)
        self.comm_stream = None

    # ---- per-layer segments, reduced as soon as a layer's gradients are complete (round 6) -------------
    # Policy `runtime.bucketed_allreduce` (default OFF: no multi-GPU box has run this yet).  The backward of a layer
    # reports its parameters (`notify`, from tssep_amd.functional where the weight gradients are accumulated straight
    # into the bucket); when every parameter of a segment has reported, the bucket is armed and more than one rank takes
    # part, the segment is all-reduced at once on a COMMUNICATION stream that waits for the compute stream and its
    # weight-gradient side stream -- layers are reported in reverse order, so the reductions queue in reverse layer order
    # while the backward of the layers in front is still running.  `all_reduce()` then reduces what is left (layers that
    # never reported: unfused paths, graph replays; the guard slot) and joins the communication stream.
    # What an RCCL kernel may run beside: the GEMMs and the fused tail of the backward -- NOT a W-stationary recurrence
    # (include/tssep_hip.h, concurrency contract: its clusters need all their workgroups resident): every W-stationary
    # launch first waits for the communication stream (hip_ops.fence_comm), so a reduction overlaps the dz / dh GEMMs
    # between two recurrences and no more.
    def set_segments(self, groups):
        """groups: lists of parameters, one per layer, in forward order; each group must be contiguous in the flat layout."""
        index = {id(p): i for i, p in enumerate(self.params)}
        self.segments, self._seg_of, self._seg_params = [], {}, []
        for g in groups:
            ids = sorted(index[id(p)] for p in g if id(p) in index)
            if not ids:
                continue
            assert ids == list(range(ids[0], ids[-1] + 1)), "a segment's parameters must be neighbours in the flat buffer"
            start = self._offsets[ids[0]]
            end = self._offsets[ids[-1] + 1] if ids[-1] + 1 < len(self.params) else self._n
            k = len(self.segments)
            self.segments.append((start, end))
            self._seg_params.append({id(self.params[i]) for i in ids})
            for i in ids:
                self._seg_of[id(self.params[i])] = k
        self._reset_step()

    def _reset_step(self):
        self._reported = [set() for _ in self.segments]
        self._reduced, self._armed = [], False

    def arm(self):
        """The backward that follows is the LAST of its virtual minibatch: complete segments may be reduced at once."""
        self._armed = bool(self.segments) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def notify(self, params):
        """The gradients of `params` are complete for this backward (queued on the current stream / its side stream)."""
        if not self._armed:
            return
        if self.flat.is_cuda and torch.cuda.is_current_stream_capturing():
            return                                   # (a captured step reduces after the replay, in all_reduce)
        for p in params:
            k = self._seg_of.get(id(p))
            if k is None or k in self._reduced:
                continue
            self._reported[k].add(id(p))
            if self._reported[k] == self._seg_params[k]:
                self._launch(k)

    def _reduce_range(self, start, end, group=None):
        """SUM over ranks of `_full[start:end]`, enqueued behind the CURRENT stream; the current stream waits for it
        (a synchronous collective: with nccl = RCCL the kernel runs on the library's own stream, which first waits for
        the current stream, and the current stream then waits for that kernel -- so "the current stream" is a handle a
        later `wait_stream` can order against; an async work object would leave the kernel on a stream nobody here sees)."""
        buf = self._full[start:end]
        if buf.is_cuda and dist.get_backend(group) == "gloo":       # test configuration: staged through the host
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(host)
            return
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)

    def _launch(self, k):
        start, end = self.segments[k]
        if self.flat.is_cuda:
            from . import hip_ops
            dev = self.flat.device
            cur = torch.cuda.current_stream(dev)
            comm = self.comm_stream = self.comm_stream or torch.cuda.Stream(device=dev)
            comm.wait_stream(cur)
            side = hip_ops._SIDE.get((str(cur.device), cur.cuda_stream))
            if side is not None:
                comm.wait_stream(side)
            with torch.cuda.stream(comm):
                self._reduce_range(start, end)         # the COMMUNICATION stream waits for the collective, not the compute stream
            hip_ops.COMM_PENDING[str(dev)] = comm      # the next W-stationary launch waits for it (fence_comm)
        else:
            self._reduce_range(start, end)
        self._reduced.append(k)

    def reduce_replicas(self):
        for f in self.flats[1:]:
            self.flat.add_(f)

    def zero(self):
        for f in self.flats:
            f.zero_()
        if self.segments:
            self._reset_step()

    def sync(self):
        """Wait for gradient work queued on the side stream (no-op on CPU / when unused)."""
        if self.flat.is_cuda:
            from . import hip_ops
            hip_ops.join_side_stream(self.flat.device)

    def set_guard(self, flag):
        """`flag`: device int32 tensor whose element 0 is non-zero when this rank's gradient must not be applied; stored as
        a float in the guard slot (one converting copy, no host sync; error codes are small positive integers, so a SUM
        over ranks is non-zero exactly when some rank's flag was).  None clears the slot."""
        if flag is None:
            self.guard.zero_()
        else:
            self.guard[:1].copy_(flag[:1])

    def all_reduce(self, group=None):
        """SUM over ranks of the gradient AND the guard slot behind it: one collective -- or, with segments reduced early
        (`arm` / `notify`), one collective per remaining contiguous range, last layers first, then a join."""
        self.sync()
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return None
        if not self._reduced:
            self._reduce_range(0, self._full.numel(), group)
            return None
        done = sorted(self.segments[k] for k in self._reduced)
        rest, pos = [], 0
        for a, b in done + [(self._full.numel(), self._full.numel())]:
            if a > pos:
                rest.append((pos, a))
            pos = max(pos, b)
        for a, b in reversed(rest):
            self._reduce_range(a, b, group)
        if self.comm_stream is not None and self.flat.is_cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)      # the early segments
            from . import hip_ops
            hip_ops.COMM_PENDING.pop(str(self.flat.device), None)
        self.last_reduction_order = list(self._reduced)
        self._reduced, self._armed = [], False
        self._reported = [set() for _ in self.segments]
        return None

    def global_norm(self):
        return torch.linalg.vector_norm(self.flat)


def env_rank():
    """(rank, world, local_rank) of this process from the torchrun environment (1 process = 1 GPU)."""
    import os
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def init_from_env(backend=None):
    """Join the job the environment describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): select the
    rank's GPU and, for WORLD_SIZE > 1, create the process group (``nccl`` = RCCL when a GPU is
    present, ``gloo`` otherwise).  Idempotent.  -> (rank, world, local_rank)."""
    import os
    rank, world, local_rank = env_rank()
    on_gpu = torch.cuda.is_available()
    if on_gpu:
        if os.environ.get("TSSEP_DIST_BACKEND") == "gloo":       # several ranks may share a GPU there
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # TSSEP_DIST_BACKEND=gloo: several ranks on ONE GPU (tests, debugging; RCCL refuses that)
        backend = backend or os.environ.get("TSSEP_DIST_BACKEND") or ("nccl" if on_gpu else "gloo")
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        # Only rank 0 validates and writes checkpoints while its peers wait in the next collective: the
        # watchdog's default (10 min for nccl) is shorter than a validation pass over a real corpus.
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("TSSEP_DIST_TIMEOUT_S", 4 * 3600)))
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout, **kw)
        _make_control_group(backend, timeout)
    return rank, world, local_rank


# Control plane: the small integers the ranks agree on per micro-step (has-data flags, failure codes) travel
# over a HOST-side gloo group next to the RCCL data plane.  On the nccl group the same 16-byte all-reduce is a
# GPU collective queued behind the previous step's kernels and read back with a host sync: the host would lose
# its launch-ahead on every micro-step.  Without the group (creation failed / TSSEP_DIST_CONTROL=0)
# the exchanges fall back to the default group, correct but synchronising.
_CONTROL = None


def _make_control_group(backend, timeout):
    """COLLECTIVE (every rank of the default group).  `dist.new_group` may fail on SOME ranks only (interface selection),
    and TSSEP_DIST_CONTROL may differ between them: ranks that exchanged control values on different groups would wait for
    each other until the watchdog fires.  The outcome is therefore agreed over the default group (all-reduce MIN of a
    success flag) and the control group is used by all ranks or by none."""
    global _CONTROL
    import os
    _CONTROL = None
    if backend != "nccl":
        return
    group, err = None, None
    want = os.environ.get("TSSEP_DIST_CONTROL", "1") != "0"
    wanted = torch.tensor([1 if want else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(wanted, op=dist.ReduceOp.MIN)            # new_group is itself collective: all ranks try, or none
    if int(wanted.item()):
        try:
            group = dist.new_group(backend="gloo", timeout=timeout)
        except Exception as e:                               # noqa: BLE001 -- agreed on below
            err = e
    ok = torch.tensor([1 if group is not None else 0], dtype=torch.int32, device=wanted.device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()):
        _CONTROL = group
    elif want:
        import warnings
        warnings.warn(f"tssep_amd.distributed: no host-side control group on every rank ({err or 'a peer has none'}); "
                      "control exchanges will synchronise with the GPU stream")


def _control_tensor(values, device=None):
    """(tensor, group) for a control exchange: host memory on the gloo control / default group, device memory
    only when the one group there is, is RCCL."""
    t = torch.tensor([int(v) for v in values], dtype=torch.int64)
    if _CONTROL is not None:
        return t, _CONTROL
    if dist.get_backend() == "nccl":
        t = t.to(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
    return t, None


def agree_on_failure(code, device=None):
    """Collective: MAX over ranks of a small non-negative failure code (0 = fine).  Every rank learns that
    SOME rank failed and can leave together instead of waiting in the next all-reduce until the watchdog
    fires.  Also a barrier.  -> (max code, rank that reported it or -1)."""
    if world_size() == 1:
        return int(code), (0 if code else -1)
    rank = get_rank()
    t, group = _control_tensor([code, (rank + 1) if code else 0], device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    worst, who = (int(v) for v in t.cpu())
    return worst, who - 1


def same_on_all_ranks(value):
    """Collective: True when the integer `value` is the same on every rank."""
    if world_size() == 1:
        return True
    t, group = _control_tensor([value, -int(value)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    hi, neg_lo = (int(v) for v in t.cpu())
    return hi == -neg_lo


def _staged(tensor):
    """gloo moves host memory: device tensors are staged through the host there (RCCL works in place)."""
    return tensor.is_cuda and dist.get_backend() == "gloo"


def broadcast_(tensor, src=0):
    """In-place broadcast of `tensor` from rank `src` (no-op without a process group)."""
    if world_size() == 1:
        return tensor
    if _staged(tensor):
        host = tensor.cpu()
        dist.broadcast(host, src=src)
        tensor.copy_(host)
    else:
        dist.broadcast(tensor, src=src)
    return tensor


def replicas_agree(tensor):
    """True when `tensor` holds the same values on every rank (two scalar all-reduces of its fp64 sum and
    its fp64 sum of squares: MIN and MAX must coincide)."""
    if world_size() == 1:
        return True
    t = tensor.detach().double()
    stat = torch.stack([t.sum(), (t * t).sum()])
    if _staged(stat):
        stat = stat.cpu()
    lo, hi = stat.clone(), stat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo.cpu(), hi.cpu()))


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def equal_shard(n_items, rank, world):
    """[lo, hi) of the rank's TRAINING shard: every rank gets floor(n / world) units (the remainder is
    dropped), because all ranks must run the same number of optimizer steps -- a rank with one batch
    more would wait for ever in the gradient all-reduce."""
    per = n_items // world
    if per == 0:
        raise ValueError(f"{n_items} training examples cannot be sharded over {world} ranks")
    return rank * per, (rank + 1) * per


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items units for `rank` (pure data parallel)."""
    per, rem = divmod(n_items, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)
