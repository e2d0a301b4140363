"""Plane-producer hand-off: the fused tail of the last synthesis block writes texels directly
(interleaved layout = channels-last [B,96,R,R]), and every field kernel reads / differentiates that layout in place.

CPU: the oracle restatement against the committed vector from the live SynthesisBlock; with the reference sources, the
block wrapper's glue on the real SynthesisNetwork (kernel replaced by the oracle function).  GPU: the HIP kernels."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import nfi_oracle_neighbours as orn

from oracle import reference

REF = reference.root()          # the checkout, or the copy oracle/make_ref.py staged (GPU box); None: neither


def gold():
    z = np.load(os.path.join(GOLDEN, 'neighbours.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files if k.startswith('handoff_')}


def test_oracle_tail_matches_committed_reference_vector():
    t = gold()
    got = orn.torgb_upsample_add(t['handoff_x'], t['handoff_styles'], t['handoff_weight'], t['handoff_bias'], t['handoff_prev'])
    assert torch.allclose(got, t['handoff_ref'], rtol=0, atol=1e-5)


@pytest.mark.skipif(REF is None, reason='reference sources not available (oracle/make_ref.py)')
def test_fused_block_glue_on_the_real_synthesis_network(monkeypatch):
    """fuse_last_block on the reference's SynthesisNetwork: same state_dict keys, and - with the kernel swapped for
    the oracle function on CPU - the same image as the unfused network, bit for bit (style computation, ws iteration,
    upsample + add order)."""
    ref_sg = reference.modules().stylegan
    import nerf_from_image_amd.handoff as handoff
    torch.manual_seed(0)
    net = ref_sg.SynthesisNetwork(w_dim=32, img_resolution=32, img_channels=96, channel_base=512, channel_max=32,
                                  use_noise=False).eval()
    ws = torch.randn(2, net.num_ws, 32)
    with torch.no_grad():
        ref = net(ws)
    keys = list(net.state_dict().keys())
    monkeypatch.setattr(handoff, 'torgb_upsample_add', orn.torgb_upsample_add)
    blk = handoff.fuse_last_block(net)
    assert blk is net.b32 and list(net.state_dict().keys()) == keys
    with torch.no_grad():
        got = net(ws)
    assert torch.equal(got, ref)
    handoff.unfuse_last_block(net)
    with torch.no_grad():
        assert torch.equal(net(ws), ref)
    with pytest.raises(AttributeError):
        handoff.fuse_last_block(torch.nn.Linear(2, 2))


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_torgb_texels_forward_and_layout(gpu_device):
    from nerf_from_image_amd import ops
    t = {k: v.to(gpu_device) for k, v in gold().items()}
    out = ops.torgb_texels(t['handoff_x'], t['handoff_styles'], t['handoff_weight'], t['handoff_bias'], t['handoff_prev'])
    assert out.shape == (2, 96, 16, 16) and out.is_contiguous(memory_format=torch.channels_last)
    scale = t['handoff_ref'].abs().max().item()
    assert (out - t['handoff_ref']).abs().max().item() <= 2e-6 * scale + 1e-6
    tex = ops.planes_view_as_texels(out.view(2, 3, 32, 16, 16))
    assert tex is not None and tex.data_ptr() == out.data_ptr() and ops.texel_layout_of(tex) == ops.TEXELS_INTERLEAVED
    # no previous image (a network whose last block is also its first)
    y = ops.torgb_texels(t['handoff_x'], t['handoff_styles'], t['handoff_weight'], t['handoff_bias'], None)
    ref = orn.torgb_upsample_add(t['handoff_x'].cpu(), t['handoff_styles'].cpu(), t['handoff_weight'].cpu(),
                                 t['handoff_bias'].cpu(), None)
    assert (y.cpu() - ref).abs().max().item() <= 2e-6 * scale + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('B,Cin,R', [(2, 32, 16), (2, 128, 64), (1, 48, 24), (1, 256, 32)])
def test_torgb_texels_backward_against_float64_autograd(gpu_device, B, Cin, R):
    from nerf_from_image_amd import handoff
    g = torch.Generator().manual_seed(B * 1000 + Cin + R)
    x = torch.randn(B, Cin, R, R, generator=g)
    s = torch.randn(B, Cin, generator=g) / Cin ** 0.5
    w = torch.randn(96, Cin, 1, 1, generator=g)
    bias = torch.randn(96, generator=g)
    prev = torch.randn(B, 96, R // 2, R // 2, generator=g)
    wgt = torch.randn(B, 96, R, R, generator=g)
    leaves64 = [v.double().requires_grad_() for v in (x, s, w, bias, prev)]
    ref = orn.torgb_upsample_add(*leaves64)
    ref_g = torch.autograd.grad((ref * wgt.double()).sum(), leaves64)
    leaves = [v.to(gpu_device).requires_grad_() for v in (x, s, w, bias, prev)]
    out = handoff.torgb_upsample_add(*leaves)
    scale = ref.abs().max().item()
    assert (out.detach().cpu().double() - ref.detach()).abs().max().item() <= 3e-6 * scale
    got_g = torch.autograd.grad((out * wgt.to(gpu_device)).sum(), leaves)
    for name, a, b in zip(('x', 'styles', 'weight', 'bias', 'previous image'), got_g, ref_g):
        assert a.shape == b.shape, name
        err = (a.cpu().double() - b).abs().max().item()
        assert err <= 2e-5 * b.abs().max().item(), (name, err, b.abs().max().item())


def _field_inputs(dev, B=2, R=24, seed=1):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 96, R, R, generator=g)                       # the producer's [B,96,R,R] output
    d = dict(img=img, w1=torch.randn(64, 32, generator=g), b1=0.3 * torch.randn(64, generator=g),
             w2=torch.randn(11, 64, generator=g), b2=0.3 * torch.randn(11, generator=g),
             att=torch.rand(B, 10, 3, generator=g) * 2 - 1, beta=torch.tensor([0.1]), alpha=torch.tensor([0.05]))
    return {k: v.to(dev) for k, v in d.items()}, g


@pytest.mark.gpu
def test_every_field_kernel_reads_the_interleaved_layout_in_place(gpu_device):
    """planar texels (nfi_planes_to_texels of an NCHW image) vs the zero-copy view of the same image stored
    channels-last: fused render and field query bit-identical, gradients equal up to the order of the atomics."""
    from nerf_from_image_amd import ops
    from nerf_from_image_amd.field_backward import field_query_bwd
    from stand_in import look_at_cameras
    dev = gpu_device
    d, g = _field_inputs(dev)
    B, R = 2, 24
    planes_nchw = d['img'].view(B, 3, 32, R, R)
    planar = ops.planes_to_texels(planes_nchw.contiguous())
    img_cl = d['img'].contiguous(memory_format=torch.channels_last)
    inter = ops.planes_view_as_texels(img_cl.view(B, 3, 32, R, R))
    assert inter is not None and inter.data_ptr() == img_cl.data_ptr() and tuple(inter.shape) == (B, R, R, 3, 32)
    assert ops.planes_view_as_texels(planes_nchw) is None
    image = ops.decoder_pack(d['w1'], d['b1'], d['w2'], d['b2'], 10)
    cam = look_at_cameras(B, 1.6, g).to(dev)
    focal = torch.full((B,), 1.0254, device=dev)
    H = W = 32
    S = 32
    nc, nf = torch.rand(B, H, W, S, generator=g).to(dev), torch.rand(B * H * W, S, generator=g).to(dev)
    for tdt, conv in ((ops.TEXEL_F32, lambda t: t), (ops.TEXEL_BF16, lambda t: t.to(torch.bfloat16))):
        a = ops.render_fwd(cam, focal, H, W, S, ops.planes_to_texels(planes_nchw.contiguous(), tdt),
                           ops.decoder_pack(d['w1'], d['b1'], d['w2'], d['b2'], 10, tdt), 0.55, 10, d['att'], True, d['beta'],
                           d['alpha'], noise_coarse=nc, noise_fine=nf)
        b = ops.render_fwd(cam, focal, H, W, S, conv(inter), ops.decoder_pack(d['w1'], d['b1'], d['w2'], d['b2'], 10, tdt),
                           0.55, 10, d['att'], True, d['beta'], d['alpha'], noise_coarse=nc, noise_fine=nf)
        for k in ('rgb', 'depth', 'mask'):
            assert torch.equal(a[k], b[k]), (tdt, k)
        assert a['mask'].mean() > 0.02
    x = ((torch.rand(B, 70000, 3, generator=g) * 2 - 1) * 0.6).to(dev)
    qa = ops.field_query(x, planar, image, 0.55, 10, d['att'], True, d['beta'], d['alpha'], want_sdf=True)
    qb = ops.field_query(x, inter, image, 0.55, 10, d['att'], True, d['beta'], d['alpha'], want_sdf=True)
    for k in ('sigma', 'rgb', 'sdf'):
        assert torch.equal(qa[k], qb[k]), k
    gs, gc = torch.randn(B, 70000, generator=g).to(dev), torch.randn(B, 70000, 3, generator=g).to(dev)
    for mode in (0, 1):                                    # per-point atomics, binned scatter
        ga = field_query_bwd(x, planar, image, d['w1'], d['w2'], 0.55, 10, d['att'], True, d['beta'], d['alpha'], gs, gc,
                             want_points=True, scatter_mode=mode)
        gb = field_query_bwd(x, inter, image, d['w1'], d['w2'], 0.55, 10, d['att'], True, d['beta'], d['alpha'], gs, gc,
                             want_points=True, scatter_mode=mode)
        assert tuple(gb['g_texels'].shape) == (B, R, R, 3, 32)
        pa, pb = ops.texel_grad_to_planes(ga['g_texels']), ops.texel_grad_to_planes(gb['g_texels'])
        assert pb.shape == pa.shape and pb.data_ptr() == gb['g_texels'].data_ptr()      # a view, no kernel
        assert (pa - pb).abs().max().item() <= 1e-5 * pa.abs().max().item(), mode
        assert torch.equal(ga['g_points'], gb['g_points'])
        for k in ('g_w1', 'g_b2', 'g_beta'):
            assert (ga[k] - gb[k]).abs().max().item() <= 1e-4 * ga[k].abs().max().item() + 1e-6, (mode, k)
    # regulariser operator (sdf + spatial gradient and its double backward)
    pts = ((torch.rand(B, 5000, 3, generator=g) * 2 - 1) * 0.5).to(dev)
    sa = ops.sdf_gradient_fwd(pts, planar, d['w1'], d['b1'], d['w2'], d['b2'], 0.55)
    sb = ops.sdf_gradient_fwd(pts, inter, d['w1'], d['b1'], d['w2'], d['b2'], 0.55)
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])
    gd, gg = torch.randn(B, 5000, generator=g).to(dev), torch.randn(B, 5000, 3, generator=g).to(dev)
    ra = ops.sdf_gradient_bwd(pts, planar, d['w1'], d['b1'], d['w2'], d['b2'], 0.55, gd, gg)
    rb = ops.sdf_gradient_bwd(pts, inter, d['w1'], d['b1'], d['w2'], d['b2'], 0.55, gd, gg)
    pa, pb = ops.texel_grad_to_planes(ra['g_texels']), ops.texel_grad_to_planes(rb['g_texels'])
    assert (pa - pb).abs().max().item() <= 1e-5 * pa.abs().max().item()


@pytest.mark.gpu
def test_generator_with_fused_handoff_runs_without_any_layout_kernel(gpu_device, monkeypatch):
    """A generator whose synthesis network has the reference's last-block structure, attached with
    fused_handoff=True: planes_to_texels / texels_to_planes are never launched (forward, fused render, backward), and
    images and gradients equal the unfused model's."""
    import copy
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    from nerf_from_image_amd import ops
    from stand_in import StandInGenerator, StyleLikeSynthesis, look_at_cameras
    dev = gpu_device
    torch.manual_seed(4)
    base = StandInGenerator(0.55, attention_values=10, use_sdf=True, plane_res=32)
    base.synthesis_network = StyleLikeSynthesis(32, channels=32)
    base = base.to(dev).eval()
    fused = copy.deepcopy(base)
    nfi_gen.attach(base)
    nfi_gen.attach(fused, fused_handoff=True)
    g = torch.Generator().manual_seed(2)
    B, H, W, S = 2, 24, 24, 32
    cam = look_at_cameras(B, 1.6, g).to(dev)
    focal = torch.full((B,), 1.0254, device=dev)
    z = torch.randn(B, 512, generator=g).to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 0.55, 'white_background': True})
    draws = [torch.rand(B, H, W, S, generator=g).to(dev), torch.rand(B * H * W, S, generator=g).to(dev)]

    def run(model, grad):
        it = iter(draws)
        real = torch.rand
        torch.rand = lambda *a, **k: next(it)
        try:
            if not grad:
                with torch.no_grad():
                    return render(model, H, W, cam, focal, None, None, z, S)[:3]
            zz = z.clone().requires_grad_()
            rgb, _, mask, _, _, _ = render(model, H, W, cam, focal, None, None, zz, S)
            last = getattr(model.synthesis_network, 'b32')
            gr = torch.autograd.grad(rgb.sum() + 2 * mask.sum(), [zz, last.torgb.weight, last.torgb.bias, last.conv1.weight,
                                                                  model.synthesis_network.b16.torgb.weight])
            return (rgb, mask) + tuple(gr)
        finally:
            torch.rand = real
    ref_img, ref_grad = run(base, False), run(base, True)

    def boom(*a, **k):
        raise AssertionError('a layout kernel was launched on the fused hand-off path')
    monkeypatch.setattr(ops, 'planes_to_texels', boom)
    monkeypatch.setattr(ops, 'texels_to_planes', boom)
    got_img, got_grad = run(fused, False), run(fused, True)
    for a, b in zip(got_img, ref_img):
        assert (a - b).abs().max().item() <= 2e-5
    assert ref_img[2].mean() > 0.02
    for name, a, b in zip(('rgb', 'mask', 'z', 'torgb.weight', 'torgb.bias', 'conv1.weight', 'previous block'), got_grad, ref_grad):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item() + 1e-6, name
