"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: image sharding, the single flat gradient
all-reduce, scalar means and metric gathering.  The render kernels themselves need no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_from_image_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # --- sharding: dim-0 split like DataParallel's scatter
        batch = torch.arange(10).view(5, 2)
        mine, none = parallel.shard_batch(batch, None)
        assert none is None
        a, b = parallel.shard_range(5)
        assert torch.equal(mine, batch[a:b])
        everyone = parallel.gather_metrics(mine.float())
        assert torch.equal(everyone, batch.float()), everyone
        # --- data-parallel gradient: loss is a SUM over images, each rank differentiates its shard
        torch.manual_seed(0)
        w = torch.nn.Parameter(torch.randn(4, 3))
        bias = torch.nn.Parameter(torch.randn(3))
        unused = torch.nn.Parameter(torch.randn(2))
        x = torch.randn(6, 4)
        xs = parallel.shard_batch(x)
        ((xs @ w + bias) ** 2).sum().backward()
        nbytes = parallel.allreduce_gradients([w, bias, unused])
        assert nbytes == (12 + 3) * 4
        w_ref = w.detach().clone().requires_grad_()
        b_ref = bias.detach().clone().requires_grad_()
        ((x @ w_ref + b_ref) ** 2).sum().backward()
        assert torch.allclose(w.grad, w_ref.grad, atol=1e-5) and torch.allclose(bias.grad, b_ref.grad, atol=1e-5)
        assert unused.grad is None
        m = parallel.allreduce_scalar_mean(float(rank + 1))
        assert abs(m - 1.5) < 1e-12
        q.put((rank, 'ok'))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_single_process_is_identity():
    t = torch.arange(6).view(3, 2)
    assert torch.equal(parallel.shard_batch(t), t)
    assert parallel.shard_range(7, rank=1, world_size=4) == (2, 4)
    assert parallel.shard_range(7, rank=3, world_size=4) == (6, 7)
    assert parallel.allreduce_gradients([torch.nn.Parameter(torch.zeros(2))]) == 0
