"""Neighbours of the renderer in the inversion loop: the augmentation warp of run.py::augment_impl
and the PSNR / IoU monitors of lib/metrics.py.

CPU: the oracle restatement against the committed vectors (tests/golden/neighbours.npz, written from the live
functions by oracle/make_golden.py) and - where the reference sources exist - the host-side pose algebra and RNG draw order
of nerf_from_image_amd.augment against the live augment_impl.  GPU: the HIP kernels against the oracle through the
C ABI (forward, backward = adjoint identity and autograd of the oracle, drop-in API)."""
import ast
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
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_oracle_matches_committed_reference_vectors():
    t = gold()
    for tag, white in (('black', False), ('white', True)):
        got = orn.warp_images(t['warp_%s_img' % tag], t['warp_%s_rot' % tag], t['warp_%s_scale' % tag],
                              t['warp_%s_trans' % tag], white)
        assert torch.allclose(got, t['warp_%s_ref' % tag], rtol=0, atol=2e-6)
        # image 0 carries the identity transform
        assert torch.allclose(got[0], t['warp_%s_img' % tag][0], atol=1e-5)
    assert torch.allclose(orn.psnr(t['metric_pred'], t['metric_target']), t['metric_psnr'], rtol=0, atol=1e-5)
    assert torch.equal(orn.iou(t['metric_mask_a'], t['metric_mask_b']), t['metric_iou'])


@pytest.mark.skipif(REF is None, reason='reference sources not available (oracle/make_ref.py)')
def test_pose_algebra_and_draw_order_match_live_reference():
    pose_utils = reference.modules().pose_utils
    import torch.nn.functional as F
    import nerf_from_image_amd.augment as aug
    tree = ast.parse(open(os.path.join(REF, 'run.py')).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'augment_impl'][0]
    env = {'torch': torch, 'np': np, 'F': F, 'pose_utils': pose_utils,
           'args': types.SimpleNamespace(supervise_alpha=False), 'dataset_config': {'white_background': False}}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'run.py::augment_impl', 'exec'), env)
    aug.configure(env['args'], env['dataset_config'])
    g = torch.Generator().manual_seed(0)
    for ortho in (False, True):
        pose = torch.eye(4).repeat(5, 1, 1)
        pose[:, :3, :3] = torch.linalg.qr(torch.randn(5, 3, 3, generator=g))[0]
        pose[:, :3, 3] = torch.randn(5, 3, generator=g)
        if ortho:
            pose[:, 3, 3] = 1 + 0.3 * torch.rand(5, generator=g)
        focal = None if ortho else 1 + 0.2 * torch.rand(5, generator=g)
        for p, disable_scale in ((1.0, False), (0.5, True)):
            torch.manual_seed(11)
            _, pr, fr, tr = env['augment_impl'](None, pose.clone(), None if ortho else focal.clone(), p, disable_scale)
            torch.manual_seed(11)
            _, pm, fm, tm = aug.augment_impl(None, pose.clone(), None if ortho else focal.clone(), p, disable_scale)
            assert all(torch.equal(a, b) for a, b in zip(tr, tm)), 'random draws differ'
            assert torch.equal(pr, pm) and (ortho or torch.equal(fr, fm))
    # p == 0 without a cached transform is the identity (run.py:803-804)
    assert aug.augment(None, pose, None, 0)[1] is pose


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('tag,white', [('black', False), ('white', True)])
def test_warp_forward_against_reference_vectors(gpu_device, tag, white):
    from nerf_from_image_amd import ops
    t = gold()
    d = lambda k: t['warp_%s_%s' % (tag, k)].to(gpu_device)
    got = ops.affine_warp(d('img'), d('rot'), d('scale'), d('trans'), white)
    err = (got.cpu() - t['warp_%s_ref' % tag]).abs().max().item()
    assert err <= 2e-5, err                      # bilinear blend of values in [-1,1]; coordinates differ by ~1e-6 px
    # scale=None means 1
    one = ops.affine_warp(d('img'), d('rot'), None, d('trans'), white)
    ref1 = orn.warp_images(t['warp_%s_img' % tag], t['warp_%s_rot' % tag], torch.ones(6), t['warp_%s_trans' % tag], white)
    assert (one.cpu() - ref1).abs().max().item() <= 2e-5


@pytest.mark.gpu
def test_warp_backward_is_the_adjoint_and_matches_autograd(gpu_device):
    from nerf_from_image_amd import ops
    import nerf_from_image_amd.augment as aug
    g = torch.Generator().manual_seed(3)
    bs, C, H, W = 30, 6, 32, 32                                        # 15 augmentations x 2 images, cat(pred, target)
    img = (torch.rand(bs, C, H, W, generator=g) * 2 - 1)
    rot = (torch.rand(bs, generator=g) - 0.5) * 2 * np.pi
    scale = torch.exp2(torch.randn(bs, generator=g) * 0.2)
    trans = torch.randn(bs, 2, generator=g) * 0.1
    w = torch.randn(bs, C, H, W, generator=g)
    x64 = img.double().requires_grad_()
    ref_out = orn.warp_images(x64, rot.double(), scale.double(), trans.double(), True)
    ref_g, = torch.autograd.grad((ref_out * w.double()).sum(), x64)
    aug.configure(types.SimpleNamespace(supervise_alpha=False), {'white_background': True})
    x = img.to(gpu_device).requires_grad_()
    out, _, _ = aug.augment(x, None, None, 1.0, cached_tform=(rot.to(gpu_device), scale.to(gpu_device), trans.to(gpu_device)))
    assert (out.detach().cpu().double() - ref_out.detach()).abs().max().item() <= 2e-5
    got_g, = torch.autograd.grad((out * w.to(gpu_device)).sum(), x)
    assert (got_g.cpu().double() - ref_g).abs().max().item() <= 1e-4 * ref_g.abs().max().item()
    # adjoint identity <warp(x), y> == <x, warp^T(y)> (black background: the map is linear)
    y = torch.randn(bs, C, H, W, generator=g).to(gpu_device)
    a = (ops.affine_warp(x.detach(), rot.to(gpu_device), scale.to(gpu_device), trans.to(gpu_device), False) * y).double().sum()
    b = (x.detach() * ops.affine_warp_bwd(y, rot.to(gpu_device), scale.to(gpu_device), trans.to(gpu_device), False)).double().sum()
    assert abs(float(a - b)) <= 1e-6 * abs(float(a)) + 1e-4


@pytest.mark.gpu
def test_augment_draws_and_dropin(gpu_device):
    """Same seed -> the same transform as the oracle fed with the recorded draws; p=0 is the identity."""
    import nerf_from_image_amd.augment as aug
    aug.configure(types.SimpleNamespace(supervise_alpha=True), {'white_background': False})
    img = torch.rand(4, 3, 16, 16, device=gpu_device)
    assert aug.augment(img, None, None, 0)[0] is img
    torch.manual_seed(5)
    out, pose, focal, tform = aug.augment(img, None, None, 0.8, return_tform=True)
    assert pose is None and focal is None
    rot, scale, trans = (t.cpu() for t in tform)
    ref = orn.warp_images(img.cpu(), rot, scale, trans, False)
    assert (out.cpu() - ref).abs().max().item() <= 2e-5
    again, _, _ = aug.augment(img, None, None, 0.8, cached_tform=tform)
    assert torch.equal(again, out)


@pytest.mark.gpu
def test_psnr_and_iou_against_reference_vectors(gpu_device):
    import nerf_from_image_amd.metrics as nfi_metrics
    t = gold()
    pred, target = t['metric_pred'].to(gpu_device), t['metric_target'].to(gpu_device)
    per = nfi_metrics.psnr(pred, target, reduction='none')
    assert (per.cpu() - t['metric_psnr']).abs().max().item() <= 2e-5 * 60
    assert float(per[4]) == 60.0
    assert abs(float(nfi_metrics.psnr(pred, target)) - float(t['metric_psnr'].mean())) <= 1e-4
    # channel-last layout gives the same number (the mean runs over everything but the batch)
    per2 = nfi_metrics.psnr(pred.permute(0, 2, 3, 1).contiguous(), target.permute(0, 2, 3, 1).contiguous(), reduction='none')
    assert (per2 - per).abs().max().item() <= 1e-4
    a, b = t['metric_mask_a'].to(gpu_device), t['metric_mask_b'].to(gpu_device)
    iou = nfi_metrics.iou(a, b, reduction='none')
    assert torch.equal(iou.cpu(), t['metric_iou'])
    assert torch.equal(nfi_metrics.iou(a.unsqueeze(1), b.unsqueeze(1), reduction='none').cpu(), t['metric_iou'])
    p2, i2 = nfi_metrics.psnr_and_iou(pred, target, a, b)
    assert torch.equal(p2, per) and torch.equal(i2, iou)
    with pytest.raises(AssertionError):                       # range check of lib/metrics.py:22-27
        nfi_metrics.psnr(pred * 2, target)
    with pytest.raises(AssertionError):
        nfi_metrics.psnr(pred[:, :2], target[:, :2])          # not an RGB image


@pytest.mark.gpu
@pytest.mark.parametrize('white', [False, True])
def test_augment_against_the_live_reference_on_the_gpu(gpu_device, white):
    """run.py::augment / augment_impl (run.py:720-817, AST-sliced) on PyTorch-ROCm against the drop-in on the same device under
    the same seed: the inversion loop's call (15 augmentations of cat(prediction, target), run.py:2216-2225, with the
    gradient back to the prediction) and the training loop's (image + pose + focal, run.py:937)."""
    if REF is None:
        pytest.skip('reference sources not staged (oracle/make_ref.py)')
    import torch.nn.functional as F
    import nerf_from_image_amd.augment as aug
    cfg = types.SimpleNamespace(supervise_alpha=False), {'white_background': white}
    env = reference.slice_functions('run.py', ['augment_impl', 'augment'],
                                    {'torch': torch, 'np': np, 'F': F, 'pose_utils': reference.modules().pose_utils,
                                     'args': cfg[0], 'dataset_config': cfg[1]})
    aug.configure(*cfg)
    g = torch.Generator(device=gpu_device).manual_seed(17)
    cat = torch.rand((2, 6, 128, 128), device=gpu_device, generator=g) * 2 - 1
    cot = torch.randn((30, 6, 128, 128), device=gpu_device, generator=g)
    res = []
    for fn in (env['augment'], aug.augment):
        x = cat.clone().requires_grad_(True)
        stack = x.unsqueeze(1).expand(-1, 15, -1, -1, -1).contiguous().flatten(0, 1)
        torch.manual_seed(23)
        out, pose, focal = fn(stack, None, None, 1.0)
        assert pose is None and focal is None
        (out * cot).sum().backward()
        res.append((out.detach(), x.grad))
    (o_r, g_r), (o, gx) = res
    # white-noise images: neighbouring texels differ by up to 2, the sampling coordinates by an ulp of [-1, 1] (8e-6 px at
    # 128 px; ATen builds them with a batched matmul of the base grid, the kernel per pixel) on each axis: 3.1e-5 measured
    assert (o - o_r).abs().max().item() <= 6e-5
    assert (gx - g_r).abs().max().item() <= 6e-5 * g_r.abs().max().item()
    # image + pose + focal, perspective and orthographic, p < 1 (some images stay as they are), with the transform returned
    for ortho in (False, True):
        pose = torch.eye(4, device=gpu_device).repeat(6, 1, 1)
        pose[:, :3, :3] = torch.linalg.qr(torch.randn((6, 3, 3), device=gpu_device, generator=g))[0]
        pose[:, :3, 3] = torch.randn((6, 3), device=gpu_device, generator=g)
        if ortho:
            pose[:, 3, 3] = 1 + 0.3 * torch.rand(6, device=gpu_device, generator=g)
        focal = None if ortho else 1 + 0.2 * torch.rand(6, device=gpu_device, generator=g)
        img = torch.rand((6, 4 if not white else 3, 64, 64), device=gpu_device, generator=g) * 2 - 1
        outs = []
        for fn in (env['augment'], aug.augment):
            torch.manual_seed(29)
            outs.append(fn(img.clone(), pose.clone(), None if focal is None else focal.clone(), 0.6, False, None, True))
        (i_r, p_r, f_r, t_r), (i_m, p_m, f_m, t_m) = outs
        assert all(torch.equal(a, b) for a, b in zip(t_r, t_m)), 'random draws differ'
        assert (i_m - i_r).abs().max().item() <= 6e-5
        assert (p_m - p_r).abs().max().item() <= 1e-5
        assert (f_r is None and f_m is None) or (f_m - f_r).abs().max().item() <= 1e-6


@pytest.mark.gpu
def test_psnr_and_iou_against_the_live_reference_on_the_gpu(gpu_device):
    """lib/metrics.py::psnr / iou (30-45, 79-94, AST-sliced: the module itself needs lpips and skimage) on PyTorch-ROCm against
    the HIP monitors on render-sized inputs, both layouts and both reductions."""
    if REF is None:
        pytest.skip('reference sources not staged (oracle/make_ref.py)')
    import nerf_from_image_amd.metrics as nfi_metrics
    env = reference.slice_functions('lib/metrics.py', ['range_check', 'psnr', 'iou'], {'torch': torch})
    g = torch.Generator(device=gpu_device).manual_seed(31)
    target = torch.rand((16, 128, 128, 3), device=gpu_device, generator=g)
    noise = torch.randn((16, 128, 128, 3), device=gpu_device, generator=g)
    level = torch.logspace(-4, -1, 16, device=gpu_device).view(16, 1, 1, 1)
    pred = (target + noise * level).clamp(-0.05, 1.05)               # (inside the range check's margin, outside [0, 1])
    pred[3] = target[3]                                              # a perfect image: the 60 dB clamp
    for a, b in ((pred, target), (pred.permute(0, 3, 1, 2).contiguous(), target.permute(0, 3, 1, 2).contiguous())):
        ref_each, got_each = env['psnr'](a, b, reduction='none'), nfi_metrics.psnr(a, b, reduction='none')
        assert (got_each - ref_each).abs().max().item() <= 2e-4, (got_each, ref_each)          # dB, out of 20 ... 60
        assert float(got_each[3]) == 60.0 == float(ref_each[3])
        assert abs(float(nfi_metrics.psnr(a, b)) - float(env['psnr'](a, b))) <= 2e-4
    ma = torch.rand((16, 128, 128), device=gpu_device, generator=g)
    mb = (ma + 0.3 * torch.randn((16, 128, 128), device=gpu_device, generator=g)).clamp(0, 1)
    mb[5], ma[5] = 0.0, 0.0                                          # empty masks: (0 + eps) / (0 + eps) = 1
    for a, b in ((ma, mb), (ma.unsqueeze(1), mb.unsqueeze(1))):
        assert torch.equal(nfi_metrics.iou(a, b, reduction='none'), env['iou'](a, b, reduction='none'))
        assert abs(float(nfi_metrics.iou(a, b)) - float(env['iou'](a, b))) <= 1e-6
    assert float(nfi_metrics.iou(ma, mb, reduction='none')[5]) == 1.0
