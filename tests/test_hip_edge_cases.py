"""Edge cases of the fused renderer and the field query against the oracle (GPU): smallest and odd image sizes,
minimum / odd sample counts, the largest attention table, the smallest and a large plane resolution, single
points, rays that graze cube edges, and inputs the reference rejects."""
import pytest
import torch

from parity_util import err
from stand_in import look_at_cameras
from nerf_from_image_amd import ops
from oracle import nfi_oracle as orc

pytestmark = pytest.mark.gpu


def scene(B, A, PR, seed, use_sdf=True):
    g = torch.Generator().manual_seed(seed)
    n_out = 1 + A if A > 0 else 4
    planes = torch.randn(B, 3, 32, PR, PR, generator=g)
    if PR >= 64:      # white noise at this resolution is a field no sampler resolves: band-limit it like a real plane producer
        low = torch.randn(B * 3, 32, 16, 16, generator=g)
        planes = torch.nn.functional.interpolate(low, size=(PR, PR), mode='bilinear', align_corners=True).view(B, 3, 32, PR, PR) \
            + 0.05 * planes
    d = dict(planes=planes, w1=torch.randn(64, 32, generator=g),
             b1=0.3 * torch.randn(64, generator=g), w2=torch.randn(n_out, 64, generator=g),
             b2=0.3 * torch.randn(n_out, generator=g), beta=torch.tensor([0.1]), alpha=torch.tensor([0.2]),
             att=(torch.rand(B, A, 3, generator=g) * 2 - 1) if A > 0 else None)
    return d, g


def both(d, cam, focal, H, W, S, A, noise_c, noise_f, dev, fine=True, white=True, use_sdf=True):
    mv = lambda t: None if t is None else t.to(dev)
    texels = ops.planes_to_texels(d['planes'].to(dev))
    image = ops.decoder_pack(mv(d['w1']), mv(d['b1']), mv(d['w2']), mv(d['b2']), A)
    r = ops.render_fwd(mv(cam), mv(focal), H, W, S, texels, image, 0.55, A, mv(d['att']), use_sdf, mv(d['beta']),
                       mv(d['alpha']), noise_coarse=mv(noise_c), noise_fine=mv(noise_f), fine_sampling=fine,
                       white_background=white, skip_missed_rays=True)
    with torch.no_grad():
        o = orc.render(d['planes'], d['w1'], d['b1'], d['w2'], d['b2'], cam, focal, H, W, S, 0.55, white_background=white,
                       fine_sampling=fine, noise_coarse=noise_c, noise_fine=noise_f, use_sdf=use_sdf, beta=d['beta'],
                       alpha=d['alpha'], attention_values=d['att'])
    return r, o


@pytest.mark.parametrize('H,W,S,A,PR,B', [(1, 1, 4, 10, 16, 1), (5, 7, 5, 14, 2, 2), (3, 9, 64, 1, 33, 1),
                                           (2, 2, 65, 10, 8, 3), (4, 6, 127, 0, 16, 1), (8, 8, 17, 10, 512, 1)])
def test_render_shapes_and_limits(gpu_device, H, W, S, A, PR, B):
    d, g = scene(B, A, PR, 11 * H + S + A)
    cam = look_at_cameras(B, 1.4, g)
    focal = torch.full((B,), 2.5)          # narrow enough for the single off-centre ray of a 1x1 image to hit the cube
    noise_c = torch.rand(B, H, W, S, generator=g)
    noise_f = torch.rand(B * H * W, S, generator=g)
    r, o = both(d, cam, focal, H, W, S, A, noise_c, noise_f, gpu_device, white=(S % 2 == 0))
    for k in ('rgb', 'depth', 'mask'):
        e = err(r[k], o[k])
        assert e['max'] <= 1e-4 and e['nonfinite'] == 0, (k, e)
    # the same shapes with the extra maps composited by the kernel (1 ... 14 attention values: the per-wave LDS table's
    # size; 1x1 images; 65 and 127 samples: the two-slot kernel): same pixels, maps against the oracle
    mv = lambda t: None if t is None else t.to(gpu_device)
    texels = ops.planes_to_texels(d['planes'].to(gpu_device))
    image = ops.decoder_pack(mv(d['w1']), mv(d['b1']), mv(d['w2']), mv(d['b2']), A)
    m = ops.render_fwd(mv(cam), mv(focal), H, W, S, texels, image, 0.55, A, mv(d['att']), True, mv(d['beta']), mv(d['alpha']),
                       noise_coarse=mv(noise_c), noise_fine=mv(noise_f), white_background=(S % 2 == 0), skip_missed_rays=True,
                       want_semantics=A > 0, want_coords=True, want_normals=True)
    for k in ('rgb', 'depth', 'mask'):
        assert torch.equal(m[k], r[k]), k
    with torch.no_grad():
        o2 = orc.render(d['planes'], d['w1'], d['b1'], d['w2'], d['b2'], cam, focal, H, W, S, 0.55, white_background=(S % 2 == 0),
                        noise_coarse=noise_c, noise_fine=noise_f, use_sdf=True, beta=d['beta'], alpha=d['alpha'],
                        attention_values=d['att'], want_semantics=A > 0)
    if A > 0:
        assert err(m['semantics'], o2['semantics'])['max'] <= 1e-5
    pts = o2['ro'].unsqueeze(-2) + o2['rd'].unsqueeze(-2) * o2['t_sorted'].unsqueeze(-1)
    assert err(m['coords'], (o2['weights'].unsqueeze(-1) * pts).sum(-2))['max'] <= 1e-5
    assert torch.isfinite(m['normals']).all() and float(m['normals'].abs().max()) <= 1.0 + 1e-4 + (1.0 if S % 2 == 0 else 0.0)


def test_density_branch_coarse_only_odd_sizes(gpu_device):
    d, g = scene(2, 0, 24, 5)
    cam = look_at_cameras(2, 1.8, g)
    focal = torch.full((2,), 0.9)
    H, W, S = 7, 3, 31
    noise_c = torch.rand(2, H, W, S, generator=g)
    r, o = both(d, cam, focal, H, W, S, 0, noise_c, None, gpu_device, fine=False, use_sdf=False)
    for k in ('rgb', 'depth', 'mask'):
        assert err(r[k], o[k])['max'] <= 1e-4, k


def test_single_point_and_cube_corners(gpu_device):
    d, g = scene(1, 10, 16, 3)
    dev = gpu_device
    texels = ops.planes_to_texels(d['planes'].to(dev))
    image = ops.decoder_pack(d['w1'].to(dev), d['b1'].to(dev), d['w2'].to(dev), d['b2'].to(dev), 10)
    r = 0.55
    for x in (torch.zeros(1, 1, 3), torch.tensor([[[r, r, r]]]), torch.tensor([[[-r, r, -r], [r, -r, 0.0], [0.0, 0.0, -r]]]),
              torch.tensor([[[r * (1 + 2e-7), 0.0, 0.0]]])):
        q = ops.field_query(x.to(dev), texels, image, r, 10, d['att'].to(dev), True, d['beta'].to(dev), d['alpha'].to(dev),
                            want_sdf=True, want_outside=True)
        ref = orc.field_query(d['planes'], d['w1'], d['b1'], d['w2'], d['b2'], x, r, True, d['beta'], d['alpha'], d['att'])
        assert torch.equal(q['outside'].cpu().float(), ref['outside'])
        assert err(q['sdf'], ref['sdf'])['max'] <= 1e-5 and err(q['rgb'], ref['rgb'])['max'] <= 1e-4
        rel = ((q['sigma'].cpu() - ref['sigma']).abs() / ref['sigma'].abs().clamp_min(1.0)).max().item()
        assert rel <= 3e-5, rel          # relative sigma bound, as in test_hip_parity.py


def test_rejected_inputs(gpu_device):
    d, g = scene(1, 10, 16, 4)
    dev = gpu_device
    texels = ops.planes_to_texels(d['planes'].to(dev))
    image = ops.decoder_pack(d['w1'].to(dev), d['b1'].to(dev), d['w2'].to(dev), d['b2'].to(dev), 10)
    cam = look_at_cameras(1, 1.4, g).to(dev)
    focal = torch.full((1,), 1.0, device=dev)
    args = (texels, image, 0.55, 10, d['att'].to(dev), True, d['beta'].to(dev), d['alpha'].to(dev))
    with pytest.raises(RuntimeError):
        ops.render_fwd(cam, focal, 4, 4, 3, *args)            # fewer than 4 samples
    with pytest.raises(RuntimeError):
        ops.render_fwd(cam, focal, 4, 4, 129, *args)          # more than 128 samples per pass with fine sampling
    with pytest.raises(RuntimeError):
        ops.render_fwd(cam, focal, 4, 4, 513, *args, fine_sampling=False)     # more than 512 in a single pass
    with pytest.raises(RuntimeError):
        ops.render_fwd(cam, focal, 4, 4, 129, *args, fine_sampling=False, want_semantics=True)   # extra maps: <= 128
    assert ops.render_fwd(cam, focal, 4, 4, 129, *args, fine_sampling=False)['rgb'].isfinite().all()
    with pytest.raises((RuntimeError, TypeError, ValueError)):
        ops.render_fwd(cam.cpu(), focal.cpu(), 4, 4, 8, *args)   # CPU tensors: no CPU path
    with pytest.raises((RuntimeError, ValueError)):
        ops.decoder_pack(d['w1'].to(dev), d['b1'].to(dev), d['w2'][:5].to(dev), d['b2'][:5].to(dev), 10)
    with pytest.raises(RuntimeError):
        ops.planes_to_texels(torch.randn(1, 3, 32, 1, 1, device=dev))     # plane_res < 2


def test_no_ray_meets_the_cube(gpu_device):
    """A batch whose rays all miss the scene cube: the reference fails (min() of an empty selection,
    lib/nerf_utils.py:258).  ops.render_fwd(strict=True) - what the drop-in render() passes by default
    (strict_near_far) - raises like it; without strict the fused kernel returns the background image."""
    d, g = scene(1, 10, 16, 11)
    dev = gpu_device
    texels = ops.planes_to_texels(d['planes'].to(dev))
    image = ops.decoder_pack(d['w1'].to(dev), d['b1'].to(dev), d['w2'].to(dev), d['b2'].to(dev), 10)
    cam = look_at_cameras(1, 1.8, g)
    cam[:, :3, 3] += 10.0 * cam[:, :3, 0]            # ten units to the side, same viewing direction: every LINE misses the cube
    cam, focal = cam.to(dev), torch.full((1,), 1.0, device=dev)
    args = (texels, image, 0.55, 10, d['att'].to(dev), True, d['beta'].to(dev), d['alpha'].to(dev))
    with pytest.raises(RuntimeError, match='no ray intersects the scene cube'):
        ops.render_fwd(cam, focal, 8, 8, 16, *args, strict=True)
    out = ops.render_fwd(cam, focal, 8, 8, 16, *args, strict=False, want_coords=True, want_semantics=True)
    assert float(out['mask'].abs().max()) == 0.0 and float((out['rgb'] - 1.0).abs().max()) == 0.0
    assert float(out['coords'].abs().max()) == 0.0 and float(out['semantics'].abs().max()) == 0.0
    # and a batch with at least one hit does not raise
    good = look_at_cameras(1, 1.8, g).to(dev)
    ok = ops.render_fwd(good, focal, 8, 8, 16, *args, strict=True)
    assert float(ok['mask'].max()) > 0.0
    # strict=True reads the set-up's hit count BEFORE the render is launched (split launch): same image as the one call
    one = ops.render_fwd(good, focal, 8, 8, 16, *args, strict=False)
    assert all(torch.equal(ok[k], one[k]) for k in ('rgb', 'depth', 'mask'))
    # the training stash redirects the ray set-up (no split): the check comes after the launch, still raises
    with pytest.raises(RuntimeError, match='no ray intersects the scene cube'):
        ops.render_fwd(cam, focal, 8, 8, 16, *args, strict=True, stash=True)
    # 'deferred': nothing is read back at the call; the NEXT strict call on the device (or flush_strict) raises
    ops.flush_strict(dev)
    ops.render_fwd(cam, focal, 8, 8, 16, *args, strict='deferred')               # no hit, no exception yet
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match='an earlier batch'):
        ops.render_fwd(good, focal, 8, 8, 16, *args, strict='deferred')
    ops.render_fwd(good, focal, 8, 8, 16, *args, strict='deferred')
    ops.flush_strict(dev)                                                        # a batch with hits: silent
    ops.render_fwd(cam, focal, 8, 8, 16, *args, strict='deferred')
    with pytest.raises(RuntimeError, match='an earlier batch'):
        ops.flush_strict(dev)
    ops.flush_strict(dev)                                                        # the queue was cleared by the raise
    # 'after': ONE launch (no split), the counter is read back behind the render kernel - raises for THIS batch
    with pytest.raises(RuntimeError, match='no ray intersects the scene cube'):
        ops.render_fwd(cam, focal, 8, 8, 16, *args, strict='after')
    two = ops.render_fwd(good, focal, 8, 8, 16, *args, strict='after')
    assert all(torch.equal(two[k], one[k]) for k in ('rgb', 'depth', 'mask'))
    # a device named without an index (torch.device('cuda')) is the current device: the same queue as the tensors' own
    # 'cuda:N' (flush_strict(torch.device('cuda')) used to look up a key of its own and never raise)
    ops.render_fwd(cam, focal, 8, 8, 16, *args, strict='deferred')
    with torch.cuda.device(dev), pytest.raises(RuntimeError, match='an earlier batch'):
        ops.flush_strict(torch.device('cuda'))
    # the STAGED path (nerf_utils.compute_near_far_planes -> ops.near_far): 'deferred' defers there too instead of taking
    # the synchronous branch because the string is truthy
    ro, rd = ops.raygen(8, 8, focal, cam, normalize=True)
    ops.near_far(ro, rd, 0.55, strict='deferred')                                # no hit, no exception, no host read
    with pytest.raises(RuntimeError, match='an earlier batch'):
        ops.flush_strict(dev)
    with pytest.raises(RuntimeError, match='no ray intersects the scene cube'):
        ops.near_far(ro, rd, 0.55, strict=True)
    ro, rd = ops.raygen(8, 8, focal, good, normalize=True)
    ops.near_far(ro, rd, 0.55, strict='deferred')
    ops.flush_strict(dev)


def test_a_band_of_background_rows_is_not_an_error(gpu_device):
    """row_window without row_window_sync under the default options: the top band of a centred object holds no ray that
    meets the cube - legitimate for a band (the image has hits), so the windowed call must not apply the batch check
    to its own counter (it used to raise there and leave the other ranks hanging in the gather)."""
    import types
    import nerf_from_image_amd.render as nfi_render
    from nerf_from_image_amd.generator import FusedField
    d, g = scene(1, 10, 16, 11)
    dev = gpu_device
    texels = ops.planes_to_texels(d['planes'].to(dev))
    image = ops.decoder_pack(d['w1'].to(dev), d['b1'].to(dev), d['w2'].to(dev), d['b2'].to(dev), 10)
    fused = FusedField(texels, image, d['att'].to(dev), 10, True, d['beta'].to(dev), d['alpha'].to(dev), 0.2)
    fused.ray_features, fused.bbox_overlay = None, False

    def model(viewdir, c, req, extra):
        smp = lambda x, r=None: None
        smp.fused = fused
        return {'sampler': smp}
    cam = look_at_cameras(1, 2.0, g).to(dev)
    focal = torch.full((1,), 1.0, device=dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dcfg = {'scene_range': 0.2, 'white_background': True}                      # a small cube: the outer bands see none of it
    full = nfi_render.make_render(cfg, dcfg)(model, 64, 64, cam, focal, None, None, None, 16, randomize=False)
    assert float(full[2][:, :8].abs().max()) == 0.0 and float(full[2].max()) > 0.0
    top = nfi_render.make_render(cfg, dcfg, row_window=(0, 8))(model, 64, 64, cam, focal, None, None, None, 16, randomize=False)
    assert top[0].shape == (1, 8, 64, 3) and torch.equal(top[0], full[0][:, :8]) and torch.equal(top[2], full[2][:, :8])


@pytest.mark.parametrize('H,W,S,B', [(16, 16, 8, 1), (16, 48, 16, 3), (32, 16, 65, 2), (64, 64, 128, 1), (48, 80, 20, 9)])
def test_work_queues_cover_every_ray_exactly_once(gpu_device, H, W, S, B):
    """Per-XCD block queues with stealing (fewer blocks than XCDs, non-square images, the wide kernel, more scenes
    than XCDs) against the single work counter: the images must be bit-identical, every pixel written."""
    d, g = scene(B, 10, 32, 7 * H + W + S)
    dev = gpu_device
    cam = look_at_cameras(B, 1.4, g).to(dev)
    focal = torch.full((B,), 1.5, device=dev)
    texels = ops.planes_to_texels(d['planes'].to(dev))
    image = ops.decoder_pack(d['w1'].to(dev), d['b1'].to(dev), d['w2'].to(dev), d['b2'].to(dev), 10)
    noise_c = torch.rand(B, H, W, S, generator=g).to(dev)
    noise_f = torch.rand(B * H * W, S, generator=g).to(dev)
    outs = []
    for tuning in (0, 16, 32, 64 + 128, 96 + 256):
        r = ops.render_fwd(cam, focal, H, W, S, texels, image, 0.55, 10, d['att'].to(dev), True, d['beta'].to(dev),
                           d['alpha'].to(dev), noise_coarse=noise_c, noise_fine=noise_f, tuning=tuning)
        # poison check: outputs are torch.empty - a ray nobody marched would leave garbage / NaN behind
        assert all(torch.isfinite(r[k]).all() for k in ('rgb', 'depth', 'mask'))
        outs.append(r)
    for r in outs[1:]:
        for k in ('rgb', 'depth', 'mask'):
            assert torch.equal(r[k], outs[0][k]), k
    assert outs[0]['mask'].max() > 0.1


def test_two_host_threads_on_their_own_streams(gpu_device):
    """under nn.DataParallel the reference calls render() from one Python thread per replica, concurrently - the
    library has to be re-entrant (no global mutable state, the caller's stream, thread-local error text).  One GPU here, so
    the two threads share the device and differ in stream and scene: every launch of each thread (fused render, and the field
    query forward + backward of a training step) has to reproduce, bit for bit, what the same thread's work gives when run alone."""
    import threading
    import nerf_from_image_amd.generator as nfi_gen
    dev = gpu_device
    H = W = 48
    S, A = 32, 10
    jobs = []
    for seed in (5, 6):
        d, g = scene(2, A, 64, seed)
        cam = look_at_cameras(2, 2.0, g)
        focal = torch.full((2,), 1.0254)
        mv = lambda t: t.to(dev)
        jobs.append(dict(texels=ops.planes_to_texels(mv(d['planes'])), planes=mv(d['planes']),
                         image=ops.decoder_pack(mv(d['w1']), mv(d['b1']), mv(d['w2']), mv(d['b2']), A),
                         w=[mv(d[k]) for k in ('w1', 'b1', 'w2', 'b2')], att=mv(d['att']), beta=mv(d['beta']), alpha=mv(d['alpha']),
                         cam=mv(cam), focal=mv(focal), nc=mv(torch.rand(2, H, W, S, generator=g)), nf=mv(torch.rand(2 * H * W, S, generator=g)),
                         pts=mv((torch.rand(2, 20000, 3, generator=g) * 2 - 1) * 0.6), cot=mv(torch.randn(2, 20000, 4, generator=g))))

    class Dec(torch.nn.Module):                      # (what make_sampler reads: .net[0] / .net[2] with weight and bias)
        def __init__(self, w):
            super().__init__()
            self.net = torch.nn.ModuleList([torch.nn.Linear(32, 64), torch.nn.Identity(), torch.nn.Linear(64, 1 + A)])
            with torch.no_grad():
                self.net[0].weight.copy_(w[0]); self.net[0].bias.copy_(w[1]); self.net[2].weight.copy_(w[2]); self.net[2].bias.copy_(w[3])

    def work(j, rounds, out):
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            dec = Dec(j['w']).to(dev)
            res = []
            for _ in range(rounds):
                r = ops.render_fwd(j['cam'], j['focal'], H, W, S, j['texels'], j['image'], 0.55, A, j['att'], True, j['beta'], j['alpha'],
                                   noise_coarse=j['nc'], noise_fine=j['nf'], fine_sampling=True, white_background=True,
                                   skip_missed_rays=True, strict=False)
                planes = j['planes'].clone().requires_grad_(True)
                pts = j['pts'].clone().requires_grad_(True)
                smp = nfi_gen.make_sampler(planes, dec, 0.55, A, j['att'], True, j['beta'], j['alpha'])
                q = smp(pts, ['sigma', 'rgb'])
                ((q['sigma'] * j['cot'][..., 0]).sum() + (q['rgb'] * j['cot'][..., 1:]).sum()).backward()
                res.append((r['rgb'].clone(), r['depth'].clone(), r['mask'].clone(), planes.grad.clone(), pts.grad.clone(),
                            dec.net[0].weight.grad.clone()))
                dec.zero_grad(set_to_none=True)
            stream.synchronize()
        out.append(res)
    alone = []
    for j in jobs:
        work(j, 2, alone)
    together = [[], []]
    failures = []

    def guarded(i):
        try:
            work(jobs[i], 12, together[i])
        except BaseException as e:                    # noqa: BLE001 - re-raised below, on the main thread
            failures.append(e)
    threads = [threading.Thread(target=guarded, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if failures:
        raise failures[0]
    def same(a, b, k):
        # outputs 3 and 5, the plane and weight gradients: sums that end in float atomics (order-dependent in the last bits)
        return torch.equal(a, b) if k not in (3, 5) else float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())   # (measured 1e-6)
    for i in range(2):
        for k, (a, b) in enumerate(zip(alone[i][0], alone[i][1])):                           # (reproducible to begin with)
            assert same(a, b, k), ('alone, job %d output %d' % (i, k), float((a - b).abs().max()), float(a.abs().max()))
        for r, res in enumerate(together[i][0]):
            for k, (a, b) in enumerate(zip(res, alone[i][0])):
                assert same(a, b, k), ('thread %d round %d output %d' % (i, r, k), float((a - b).abs().max()))
    assert not torch.equal(together[0][0][0][0], together[1][0][0][0])                         # (two different scenes)
