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


def _bucket_worker(rank, world, port, q, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3), torch.nn.Tanh(),
                                  torch.nn.Linear(3, 2))
        unused = torch.nn.Parameter(torch.randn(3))
        params = list(net.parameters()) + [unused]
        # tiny buckets: several collectives, launched while backward is still running
        gb = parallel.GradientBuckets(params, bucket_bytes=64, mode=mode)
        assert len(gb.buckets) > 2
        x = torch.randn(5, 7)                       # uneven batch: 3 images on rank 0, 2 on rank 1
        xs = parallel.shard_batch(x)
        assert xs.shape[0] == (3 if rank == 0 else 2)
        ref = [p.detach().clone().requires_grad_() for p in net.parameters()]

        def fwd(ps, inp):
            h = torch.tanh(inp @ ps[0].t() + ps[1])
            h = torch.tanh(h @ ps[2].t() + ps[3])
            return ((h @ ps[4].t() + ps[5]) ** 2).sum()
        for step in range(2):                      # second step: views survive zero_grad()
            gb.zero_grad()
            fwd(list(net.parameters()), xs).backward()
            assert gb.launched_in_backward >= 1     # overlap: some buckets went out before finish()
            nbytes = gb.finish()
            assert nbytes == gb.nbytes
            for r in ref:
                r.grad = None
            fwd(ref, x).backward()
            for p, r in zip(net.parameters(), ref):
                assert torch.allclose(p.grad, r.grad, atol=1e-5), (step, p.grad, r.grad)
                assert p.grad.data_ptr() == gb._ptr[id(p)]
            assert torch.equal(unused.grad, torch.zeros(3))
        # ParallelModel + shard_batch with the uneven batch: every rank renders its own images
        calls = []

        def fake_render(model, h, w, cam, focal, center, bbox, c, spr, **kw):
            calls.append((cam.shape[0], h, spr))
            return (cam.sum(dim=(1, 2)),)
        pm = parallel.ParallelModel(16, model=net, render=fake_render, depth_samples_per_ray=8)
        cam = torch.arange(5 * 16, dtype=torch.float32).view(5, 4, 4)
        mine = pm(parallel.shard_batch(cam), None, None, None, None, res_multiplier=2, ray_multiplier=2)[0]
        assert calls == [(3 if rank == 0 else 2, 32, 16)]
        assert torch.equal(parallel.gather_metrics(mine), cam.sum(dim=(1, 2)))
        q.put((rank, 'ok'))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run_two(target, *extra):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + extra) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_gradient_buckets_all_reduce_uneven_batch():
    _run_two(_bucket_worker, 'all_reduce')


def test_gradient_buckets_reduce_scatter():
    _run_two(_bucket_worker, 'reduce_scatter')


def _accumulate_worker(rank, world, port, q):
    """Two backward() calls per optimiser step, as run.py:1044 / 1110-1139 do: all but the last under no_sync(); a
    delay between them gives an (erroneously) early collective time to finish - the failure mode of round 2's code."""
    import time
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
        gb = parallel.GradientBuckets(list(net.parameters()), bucket_bytes=64)
        x = torch.randn(6, 7)
        xs = parallel.shard_batch(x)
        ref = [p.detach().clone().requires_grad_() for p in net.parameters()]

        def losses(ps, inp):
            h = torch.tanh(inp @ ps[0].t() + ps[1])
            o = h @ ps[2].t() + ps[3]
            return (o ** 2).sum(), o.abs().sum()
        for step in range(2):
            gb.zero_grad()
            la, lb = losses(list(net.parameters()), xs)
            with gb.no_sync():
                la.backward(retain_graph=True)
            assert gb.launched_in_backward == 0 and not any(gb._launched)
            time.sleep(0.3 if rank == 0 else 0.0)
            lb.backward()
            assert gb.launched_in_backward >= 1
            gb.finish()
            for r in ref:
                r.grad = None
            ra, rb = losses(ref, x)
            (ra + rb).backward()
            for p, r in zip(net.parameters(), ref):
                assert torch.allclose(p.grad, r.grad, atol=1e-5), (step, (p.grad - r.grad).abs().max())
        # misuse: a second synchronising backward in the same step must raise, not corrupt
        gb.zero_grad()
        la, lb = losses(list(net.parameters()), xs)
        la.backward(retain_graph=True)
        try:
            lb.backward()
            raised = False
        except RuntimeError as e:
            raised = 'no_sync' in str(e)
        gb.finish()
        assert raised
        # ... and so must no_sync() after collectives went out
        try:
            with gb.no_sync():
                pass
            raised = False
        except RuntimeError:
            raised = True
        assert raised
        # sub-group: padding / averaging follow the GROUP's size, not the default group's
        # (new_group is collective: both groups are built on every rank)
        g0, g1 = dist.new_group([0]), dist.new_group([1])
        mine = g0 if rank == 0 else g1
        w = torch.nn.Parameter(torch.full((3,), float(rank + 1)))
        gs = parallel.GradientBuckets([w], mode='reduce_scatter', average=True, group=mine)
        assert gs.world_size == 1 and gs.buckets[0]['flat'].numel() == 3
        (w * w).sum().backward()
        gs.finish()
        assert torch.allclose(w.grad, 2 * w.detach())
        # one image over both ranks: row bands (whole 8-row strips) rendered locally, gathered to the full image
        H, W = 40, 6
        img = torch.arange(2 * H * W * 3, dtype=torch.float32).view(2, H, W, 3)
        r0, r1 = parallel.shard_rows(H)
        assert (r0, r1) == ((0, 16) if rank == 0 else (16, 40)) and r0 % 8 == 0
        seen = []

        def fake_rows(a, b):
            seen.append((a, b))
            return img[:, a:b], img[:, a:b, :, 0]
        full, first = parallel.render_image_rows(fake_rows, H)
        assert seen == [(r0, r1)] and torch.equal(full, img) and torch.equal(first, img[..., 0])
        # fewer strips than ranks: one rank gets the strip, the other an empty band - which must not reach the renderer
        # (the kernels refuse height 0) and must still take part in the gather instead of leaving the others hanging in it
        assert parallel.shard_rows(8, rank=0, world_size=2) == (0, 0) and parallel.shard_rows(8, rank=1, world_size=2) == (0, 8)
        small = torch.arange(2 * 8 * W * 3, dtype=torch.float32).view(2, 8, W, 3)
        calls = []

        def strict_rows(a, b):
            assert b > a, 'render_rows called with an empty band'
            calls.append((a, b))
            return small[:, a:b], None, small[:, a:b, :, 0].to(torch.int32)
        full, none, first = parallel.render_image_rows(strict_rows, 8)
        assert calls == ([] if rank == 0 else [(0, 8)]) and none is None
        assert torch.equal(full, small) and torch.equal(first, small[..., 0].to(torch.int32))
        one = parallel.render_image_rows(lambda a, b: strict_rows(a, b)[0], 8)
        assert torch.is_tensor(one) and torch.equal(one, small)
        # ... and when the bands are rendered with row_window_sync (the render reduces its miss-fill cells over the ranks),
        # the rank without rows enters the same reduction with the neutral element instead of going straight to the gather
        def synced_rows(a, b):
            cells_ = torch.zeros(16, dtype=torch.uint8)
            cells_[:12].view(torch.int32).copy_(torch.tensor([5, 9, 3], dtype=torch.int32))
            parallel.allreduce_ray_setup(cells_)
            assert cells_[:12].view(torch.int32).tolist() == [5, 9, 3]          # the other rank added nothing
            return small[:, a:b]
        joined = parallel.render_image_rows(synced_rows, 8, on_empty=lambda: parallel.join_ray_setup('cpu'))
        assert torch.equal(joined, small)
        # the miss-fill cells of a row-sharded render: max of the two order-preserving keys (unsigned), sum of the count
        ws = torch.zeros(64, dtype=torch.uint8)
        cells = ws[:12].view(torch.int32)
        mine3 = [0x7F000000 + 5, 0xC0000001, 7] if rank == 0 else [0x80000002, 0x40000000, 11]
        cells.copy_(torch.tensor([v - 2 ** 32 if v >= 2 ** 31 else v for v in mine3], dtype=torch.int32))
        parallel.allreduce_ray_setup(ws)
        got = [int(v) & 0xFFFFFFFF for v in ws[:12].view(torch.int32).tolist()]
        assert got == [0x80000002, 0xC0000001, 18], [hex(v) for v in got]
        assert not ws[12:].any()
        # one-shot form keeps a staging buffer per gradient size (G and D steps alternate)
        a, b = torch.nn.Parameter(torch.ones(5)), torch.nn.Parameter(torch.ones(9))
        for _ in range(2):
            for prm in (a, b):
                prm.grad = torch.full_like(prm, float(rank + 1))
                parallel.allreduce_gradients([prm])
                assert torch.equal(prm.grad, torch.full_like(prm, 3.0))
        assert {k[1] for k in parallel._FLAT_CACHE} >= {5, 9}
        q.put((rank, 'ok'))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_gradient_buckets_two_backwards_per_step():
    _run_two(_accumulate_worker)


def test_gradient_buckets_single_process():
    w = torch.nn.Parameter(torch.randn(4, 3))
    b = torch.nn.Parameter(torch.randn(3))
    gb = parallel.GradientBuckets([w, b])
    (w.sum() + b.sum()).backward()
    assert gb.finish() == 0 and torch.equal(w.grad, torch.ones(4, 3))
    gb.zero_grad()
    assert torch.equal(w.grad, torch.zeros(4, 3)) and w.grad.data_ptr() == gb._ptr[id(w)]


def test_stand_in_basis_mix_matches_einsum():
    """tools/basis_mix.py (the stand-in plane producers' only heavy op): values and both gradients against einsum."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from basis_mix import basis_mix
    torch.manual_seed(3)
    c = torch.randn(3, 4, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, 5, 6, 7, dtype=torch.float64, requires_grad=True)
    assert torch.allclose(basis_mix(c, b), torch.einsum('bk,kchw->bchw', c, b))
    assert torch.autograd.gradcheck(basis_mix, (c, b))
