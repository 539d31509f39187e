"""Input pipeline: the lazy-dataset calls the reference's ``Model.prepare_dataset`` makes
(tssep/train/model.py:182-337: ``map``, ``shuffle(reshuffle=True)``, ``batch``, ``prefetch``, ``catch``,
``sort``, ``copy``; the third-party ``lazy_dataset`` package is absent) and the device stage that replaces
``pt.data.example_to_device`` + ``prefetch(1, 2)`` (model.py:166-180, 324-335).

MI355X-first where it matters: host batches are staged in PINNED buffers that are reused, copied with
asynchronous H2D transfers on a dedicated copy stream, and handed to the compute stream through events --
the training step never waits for PCIe unless the producer is slower than the GPU (``DeviceLoader``).
Everything else is plain host logic (threads, order-preserving queues)."""
import collections
import queue
import threading

import numpy as np
import torch


class FilterException(Exception):
    """Raised by a map function to drop an example (caught by ``catch()`` / ``prefetch(catch_filter_exception=True)``)."""


def new(examples):
    """lazy_dataset.new: a dataset over a list, or over the values of a dict (keys become example ids)."""
    if isinstance(examples, Dataset):
        return examples
    if isinstance(examples, dict):
        examples = list(examples.values())
    return Dataset(list(examples))


class Dataset:
    """An indexable source plus a chain of lazy stages.  Iterating evaluates the chain; every ``__iter__``
    is a new pass (a ``shuffle(reshuffle=True)`` stage draws a new order per pass, like lazy_dataset)."""

    def __init__(self, source, stages=()):
        self._source = source
        self._stages = tuple(stages)

    def _with(self, stage):
        return Dataset(self._source, self._stages + (stage,))

    # ---- the calls model.py makes
    def map(self, fn):
        return self._with(("map", fn))

    def shuffle(self, reshuffle=False, rng=None):
        state = {"rng": rng if rng is not None else np.random, "perm": None}
        return self._with(("shuffle", bool(reshuffle), state))

    def sort(self, key, reverse=False):
        return Dataset(sorted(list(self), key=key, reverse=reverse))

    def copy(self, freeze=False):
        return Dataset(list(self)) if freeze else Dataset(self._source, self._stages)

    def shard(self, rank, world):
        """The rank's share of the SOURCE examples for data-parallel training (equal shares, remainder
        dropped: tssep_amd.distributed.equal_shard); later stages see only that share."""
        if world == 1:
            return self
        from .distributed import equal_shard
        items = list(self._source)
        lo, hi = equal_shard(len(items), rank, world)
        return Dataset(items[lo:hi], self._stages)

    def batch(self, batch_size, drop_last=False):
        return self._with(("batch", int(batch_size), bool(drop_last)))

    def catch(self, exceptions=(FilterException,)):
        return self._with(("catch", tuple(exceptions)))

    def prefetch(self, num_workers, buffer_size, backend="t", catch_filter_exception=False):
        if backend != "t":
            raise NotImplementedError("prefetch backend %r (threads only)" % (backend,))
        return self._with(("prefetch", int(num_workers), int(buffer_size), bool(catch_filter_exception)))

    # ---- evaluation
    def __iter__(self):
        it = iter(self._source)
        pending_maps = []          # maps directly in front of a prefetch stage run inside its workers

        def flush(it_, maps):
            for fn in maps:
                it_ = map(fn, it_)
            return it_

        for st in self._stages:
            kind = st[0]
            if kind == "map":
                pending_maps.append(st[1])
                continue
            if kind == "prefetch":
                it = _threaded(it, list(pending_maps), st[1], st[2], st[3])
                pending_maps = []
                continue
            if kind == "shuffle":
                # per-example maps commute with a permutation: order first, map lazily afterwards
                items = list(it)
                state = st[2]
                if st[1] or state["perm"] is None or len(state["perm"]) != len(items):
                    state["perm"] = state["rng"].permutation(len(items))
                it = (items[i] for i in state["perm"])
                continue
            it = flush(it, pending_maps)
            pending_maps = []
            if kind == "batch":
                it = _batched(it, st[1], st[2])
            elif kind == "catch":
                it = _catching(it, st[1])
        return flush(it, pending_maps)

    def __len__(self):
        n = len(self._source)
        for st in self._stages:
            if st[0] == "batch":
                n = n // st[1] if st[2] else -(-n // st[1])
            elif st[0] in ("catch",) or (st[0] == "prefetch" and st[3]):
                raise TypeError("the length of a dataset that may drop examples is unknown")
        return n

    def __getitem__(self, item):
        if isinstance(item, slice):
            out = []
            stop = item.stop
            for i, ex in enumerate(self):
                if stop is not None and i >= stop:
                    break
                out.append(ex)
            return out[item.start or 0::item.step or 1] if (item.start or item.step) else out
        if item < 0:
            return list(self)[item]
        for i, ex in enumerate(self):
            if i == item:
                return ex
        raise IndexError(item)


def _batched(it, n, drop_last):
    buf = []
    for ex in it:
        buf.append(ex)
        if len(buf) == n:
            yield buf
            buf = []
    if buf and not drop_last:
        yield buf


def _catching(it, exceptions):
    it = iter(it)
    while True:
        try:
            yield next(it)
        except StopIteration:
            return
        except exceptions:
            continue


_END = object()


def _threaded(it, maps, num_workers, buffer_size, catch_filter):
    """Order-preserving threaded map: ``num_workers`` threads apply ``maps`` to the items of ``it``; at most
    ``buffer_size`` results are in flight.  Exceptions surface in the consumer at the item's position."""
    if num_workers < 1 or buffer_size < 1:
        raise ValueError((num_workers, buffer_size))
    src = iter(it)
    src_lock = threading.Lock()
    slots = threading.Semaphore(buffer_size)
    results = {}
    cond = threading.Condition()
    state = {"next_in": 0, "done_in": False, "stop": False}

    def work():
        while True:
            slots.acquire()
            with src_lock:
                if state["stop"] or state["done_in"]:
                    slots.release()
                    return
                try:
                    item = next(src)
                except StopIteration:
                    state["done_in"] = True
                    slots.release()
                    with cond:
                        cond.notify_all()
                    return
                except BaseException as e:                 # the source itself failed: deliver in order
                    idx = state["next_in"]
                    state["next_in"] += 1
                    state["done_in"] = True
                    with cond:
                        results[idx] = (False, e)
                        cond.notify_all()
                    return
                idx = state["next_in"]
                state["next_in"] += 1
            try:
                for fn in maps:
                    item = fn(item)
                res = (True, item)
            except BaseException as e:
                res = (False, e)
            with cond:
                results[idx] = res
                cond.notify_all()

    threads = [threading.Thread(target=work, daemon=True) for _ in range(num_workers)]
    for t in threads:
        t.start()
    try:
        out = 0
        while True:
            with cond:
                while out not in results and not (state["done_in"] and out >= state["next_in"]):
                    cond.wait(0.05)
                if out not in results:
                    return
                ok, val = results.pop(out)
            out += 1
            slots.release()
            if ok:
                yield val
            elif catch_filter and isinstance(val, FilterException):
                continue
            else:
                raise val
    finally:
        state["stop"] = True
        for _ in threads:
            slots.release()
        for t in threads:                                    # (bounded: a map stage stuck in user code must not hang the exit)
            t.join(timeout=2.0)


class DeviceLoader:
    """Moves host batches to the GPU ahead of the consumer.

    ``source``: iterable of dict batches (numpy arrays / torch CPU tensors under ``keys``; everything else
    passes through).  A producer thread copies each array into a reusable PINNED staging buffer, issues the
    H2D copy on a dedicated stream and records an event; ``__next__`` makes the consumer's current stream
    wait for that event (no host synchronisation) and returns device tensors.  ``depth`` batches are in
    flight (the reference keeps 2, model.py:333-335)."""

    def __init__(self, source, device, keys, depth=2, copy_threads=6):
        self.source, self.device, self.keys, self.depth = source, torch.device(device), tuple(keys), int(depth)
        self.copy_threads = int(copy_threads)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceLoader needs a GPU device (tssep_amd has no CPU path)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.stats = collections.Counter()

    def __len__(self):
        return len(self.source)

    def __getitem__(self, item):
        if isinstance(item, slice) and item.start in (None, 0) and item.step in (None, 1):
            out = []
            for i, ex in enumerate(self):
                if item.stop is not None and i >= item.stop:
                    break
                out.append(ex)
            return out
        if isinstance(item, int) and item >= 0:              # ds[i]: the reference indexes datasets too
            for i, ex in enumerate(self):
                if i == item:
                    return ex
            raise IndexError(item)
        raise TypeError("DeviceLoader supports iteration, [:n] and [i >= 0]")

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        copy_stream = torch.cuda.Stream(device=self.device)
        import concurrent.futures
        pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.copy_threads)
        ring = [dict() for _ in range(self.depth + 1)]        # pinned buffers per slot: key -> (tensor, event)

        def stage(slot, key, value):
            t = torch.as_tensor(np.ascontiguousarray(value)) if isinstance(value, np.ndarray) else value.contiguous()
            buf, ev = ring[slot].get(key, (None, None))
            if buf is None or buf.dtype != t.dtype or buf.numel() < t.numel():
                buf = torch.empty(t.numel(), dtype=t.dtype).pin_memory()
                self.stats["pinned_allocations"] += 1
            elif ev is not None:
                ev.synchronize()                                # the copy that last read this buffer is done
            view = buf[:t.numel()].view(t.shape)
            nbytes = t.numel() * t.element_size()
            if nbytes >= (32 << 20) and t.dim() >= 1 and t.shape[0] >= 2 * self.copy_threads:
                # large tensors: the pageable -> pinned copy is the slow leg (one core moves ~8 GB/s, a
                # batch of 768 utterances is 1 GB); chunks along the batch axis on a few threads
                # (Tensor.copy_ releases the GIL), each chunk's DMA issued as soon as it is staged
                # allocated UNDER the copy stream: the block then belongs to that stream's pool, so the
                # allocator cannot hand it out again (to the next batch's DMA) while kernels of a step the
                # host has already queued still read it -- the consumer's record_stream(cur) below defers
                # the reuse until its stream has passed this point.  (Allocated on the default stream the
                # block was recycled at once and the next DMA overwrote a batch still in use.)
                with torch.cuda.stream(copy_stream):
                    dev = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                bounds = np.linspace(0, t.shape[0], self.copy_threads + 1).astype(int)

                def chunk(i):
                    a, b = int(bounds[i]), int(bounds[i + 1])
                    view[a:b].copy_(t[a:b])
                    with torch.cuda.stream(copy_stream):
                        dev[a:b].copy_(view[a:b], non_blocking=True)
                list(pool.map(chunk, range(self.copy_threads)))
                with torch.cuda.stream(copy_stream):
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
            else:
                view.copy_(t)
                with torch.cuda.stream(copy_stream):
                    dev = view.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
            ring[slot][key] = (buf, ev)
            self.stats["h2d_bytes"] += t.numel() * t.element_size()
            return dev, ev

        def produce():
            try:
                torch.cuda.set_device(self.device)
                for n, ex in enumerate(self.source):
                    if stop.is_set():
                        return
                    ex = dict(ex)
                    events = []
                    for k in self.keys:
                        v = ex.get(k)
                        if isinstance(v, np.ndarray) and v.dtype != object or (isinstance(v, torch.Tensor) and not v.is_cuda):
                            ex[k], ev = stage(n % len(ring), k, v)
                            events.append(ev)
                    while not stop.is_set():
                        try:
                            q.put((ex, events), timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(_END)
            except BaseException as e:                          # surfaces in the consumer
                q.put(e)

        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                ex, events = item
                cur = torch.cuda.current_stream(self.device)
                for ev in events:
                    cur.wait_event(ev)
                for k in self.keys:
                    v = ex.get(k)
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(cur)
                self.stats["batches"] += 1
                yield ex
        finally:
            # stop the producer and WAIT for it (bounded): a daemon thread still inside a pinned-memory / HIP call when the
            # interpreter finalises takes the process down (`terminate called without an active exception`)
            stop.set()
            try:
                while True:
                    q.get_nowait()                      # (a producer blocked on the full queue sees the stop flag at once)
            except queue.Empty:
                pass
            th.join(timeout=10.0)
            pool.shutdown(wait=True)
