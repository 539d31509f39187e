"""N > 1 path on CPU: world_size-2 gloo run of the gradient bucket (SUM all-reduce) and the
utterance sharding.  Utterances are independent, so summed shard gradients must equal the
gradient of the concatenated batch (checked here on the oracle's loss, which is summed over the
batch like tssep/train/model.py:669)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tssep_amd.distributed import GradBucket, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
    q.put((rank, bucket.flat.clone(), float(bucket.global_norm())))
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_gradient_sum_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(30)
    torch.manual_seed(0)
    lin = torch.nn.Linear(6, 3)
    x = torch.arange(8 * 6, dtype=torch.float32).view(8, 6) / 10
    lin(x).abs().sum().backward()
    ref = torch.cat([lin.weight.grad.flatten(), lin.bias.grad.flatten()])
    for _, flat, norm in res:
        torch.testing.assert_close(flat, ref)
        assert abs(norm - float(ref.norm())) < 1e-4


def test_bucket_grads_are_views():
    lin = torch.nn.Linear(4, 2)
    b = GradBucket(lin.parameters())
    lin(torch.ones(3, 4)).sum().backward()
    assert lin.weight.grad.data_ptr() == b.flat.data_ptr()
    assert float(b.flat.abs().sum()) > 0
    b.zero()
    assert float(lin.weight.grad.abs().sum()) == 0
