"""Generate tests/golden/*.npz from the LIVE reference and pin the oracle to it.

TEST INFRASTRUCTURE ONLY (see oracle/nfi_oracle.py header).

Runs where the reference sources exist (/root/reference in the build container, else the staged oracle/_ref).  It
  1. imports lib.nerf_utils / models.generator from /root/reference with
     PYTORCH_JIT=0 (scripted functions become plain Python, so the two
     torch.rand* draws can be intercepted),
  2. obtains ``render`` by AST-slicing run.py (run.py is a script: argparse at
     import time) and exec-ing it with stub ``args`` / ``dataset_config``,
  3. wraps the nerf_utils stage functions and the sampler closure to record
     every stage boundary,
  4. runs a set of small seeded cases, asserts that oracle/nfi_oracle.py
     reproduces every recorded tensor BIT FOR BIT on CPU, and
  5. writes inputs + reference outputs to tests/golden/.

Usage:  PYTORCH_JIT=0 python oracle/make_golden.py
"""
import os
import sys

os.environ['PYTORCH_JIT'] = '0'
import ast
import math
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import reference                  # noqa: E402

REF = reference.root()                        # /root/reference, or the copy oracle/make_ref.py staged
assert REF is not None, 'no reference sources (neither /root/reference nor oracle/_ref)'
sys.path.insert(0, REF)
warnings.filterwarnings('ignore')

from lib import nerf_utils as ref_nu          # noqa: E402
from models import generator as ref_gen       # noqa: E402
from oracle import nfi_oracle as orc          # noqa: E402


def load_reference_render(cfg_args, dataset_config):
    src = open(os.path.join(REF, 'run.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'render'][0]
    mod = ast.Module(body=[fn], type_ignores=[])
    import torch.nn.functional as F
    g = {'torch': torch, 'F': F, 'nerf_utils': ref_nu, 'args': cfg_args,
         'dataset_config': dataset_config}
    exec(compile(mod, 'run.py::render', 'exec'), g)
    return g['render'], g


class Recorder:
    """Wraps module-level functions to record their outputs."""

    def __init__(self):
        self.log = {}
        self._orig = {}

    def wrap(self, module, name, keys):
        orig = getattr(module, name)
        self._orig[(module, name)] = orig

        def wrapped(*a, **k):
            out = orig(*a, **k)
            outs = out if isinstance(out, tuple) else (out,)
            for key, val in zip(keys, outs):
                if key is not None and val is not None:
                    self.log.setdefault(key, []).append(val.detach().clone())
            return out
        setattr(module, name, wrapped)

    def restore(self):
        for (module, name), orig in self._orig.items():
            setattr(module, name, orig)


class NoiseTap:
    """Intercepts torch.rand / torch.rand_like in draw order (render draws
    rand_like [B,H,W,S] first, then rand [N,S])."""

    def __init__(self, gen):
        self.gen = gen
        self.draws = []

    def __enter__(self):
        self._rand, self._rand_like = torch.rand, torch.rand_like

        def rand(*size, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (list, tuple)) else size
            out = self._rand(list(size), generator=self.gen, dtype=kw.get('dtype', torch.float32))
            self.draws.append(out.clone())
            return out

        def rand_like(t, **kw):
            out = self._rand(list(t.shape), generator=self.gen, dtype=t.dtype)
            self.draws.append(out.clone())
            return out
        torch.rand, torch.rand_like = rand, rand_like
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like = self._rand, self._rand_like


def look_at_cameras(n, radius, gen, ortho=False):
    """Random cameras on a sphere looking at the origin (OpenGL convention:
    camera looks down -z, as nerf_utils.get_ray_bundle expects)."""
    v = torch.randn(n, 3, generator=gen)
    eye = radius * v / v.norm(dim=-1, keepdim=True)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    up = torch.tensor([0., 0., 1.]).expand(n, 3)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    true_up = torch.cross(right, fwd, dim=-1)
    cam = torch.eye(4).repeat(n, 1, 1)
    cam[:, :3, 0] = right
    cam[:, :3, 1] = true_up
    cam[:, :3, 2] = -fwd
    cam[:, :3, 3] = eye
    return cam


CASES = {
    # name: dict(...)
    'persp_white_fine_rand': dict(B=2, H=12, W=12, S=16, scene_range=0.55, radius=2.0, focal=1.0254,
                                  white=True, fine=True, randomize=True, sdf=True, A=10, alpha=0.05,
                                  beta=0.1, bbox=False, ortho=False),
    'persp_bbox_black_fine_rand': dict(B=2, H=10, W=14, S=16, scene_range=1.4, radius=2.0, focal=1.0,
                                       white=False, fine=True, randomize=True, sdf=True, A=10, alpha=0.02,
                                       beta=0.07, bbox=True, ortho=False),
    'ortho_fine_det': dict(B=2, H=12, W=12, S=16, scene_range=2.0, radius=3.0, focal=None,
                           white=False, fine=True, randomize=False, sdf=True, A=10, alpha=0.03,
                           beta=0.1, bbox=False, ortho=True),
    'density_rgb_coarse_only': dict(B=2, H=12, W=12, S=32, scene_range=0.55, radius=2.0, focal=1.0254,
                                    white=True, fine=False, randomize=True, sdf=False, A=0, alpha=1.0,
                                    beta=0.1, bbox=False, ortho=False),
    'persp_s64_fine_rand': dict(B=1, H=8, W=8, S=64, scene_range=0.55, radius=1.3, focal=1.0254,
                                white=True, fine=True, randomize=True, sdf=True, A=10, alpha=0.02,
                                beta=0.1, bbox=False, ortho=False),
    # BASELINE cfg5 (ray_multiplier=2): 128 + 128 samples per ray, and a ragged 96 + 96
    'persp_s128_fine_rand': dict(B=1, H=6, W=6, S=128, scene_range=0.55, radius=1.3, focal=1.0254,
                                 white=True, fine=True, randomize=True, sdf=True, A=10, alpha=0.02,
                                 beta=0.1, bbox=False, ortho=False),
    # --use_viewdir (carla): per-ray ViewDirectionMapper feature added to a 32-wide decoder output
    'persp_viewdir_fine_rand': dict(B=2, H=8, W=8, S=16, scene_range=0.55, radius=1.6, focal=1.0254,
                                    white=True, fine=True, randomize=True, sdf=True, A=10, alpha=0.05,
                                    beta=0.1, bbox=False, ortho=False, viewdir=True),
    'viewdir_density_rgb_coarse_only': dict(B=1, H=8, W=8, S=24, scene_range=0.55, radius=1.6, focal=1.0254,
                                            white=False, fine=False, randomize=True, sdf=False, A=0, alpha=1.0,
                                            beta=0.1, bbox=False, ortho=False, viewdir=True),
    # principal-point shift (`center`, nerf_utils.py:42-46; only data/loaders.py:185 ever sets it)
    'persp_center_fine_rand': dict(B=2, H=10, W=12, S=16, scene_range=0.55, radius=2.0, focal=1.0254,
                                   white=True, fine=True, randomize=True, sdf=True, A=10, alpha=0.05,
                                   beta=0.1, bbox=False, ortho=False, center=True),
    # compute_coords (the query points composited in the semantics slot, run.py:337-338) + force_no_cam_grad
    'persp_coords_black_fine_rand': dict(B=2, H=8, W=8, S=16, scene_range=0.55, radius=1.6, focal=1.0254,
                                         white=False, fine=True, randomize=True, sdf=True, A=10, alpha=0.05,
                                         beta=0.1, bbox=False, ortho=False, coords=True),
    # one pass without fine sampling at the inversion loop's sample count (run.py:512-514 x ray_multiplier 4, run.py:2271)
    'persp_s512_coarse_only_rand': dict(B=1, H=5, W=4, S=512, scene_range=0.55, radius=1.3, focal=1.0254,
                                        white=False, fine=False, randomize=True, sdf=True, A=10, alpha=0.02,
                                        beta=0.1, bbox=False, ortho=False),
    'persp_s96_black_fine_det': dict(B=1, H=6, W=6, S=96, scene_range=0.55, radius=1.3, focal=1.0254,
                                     white=False, fine=True, randomize=False, sdf=True, A=10, alpha=0.02,
                                     beta=0.1, bbox=False, ortho=False),
}

PLANE_RES = 32
PLANE_CH = 32


def make_planes(n_scenes, seed):
    g = torch.Generator().manual_seed(seed)
    # smooth-ish random planes: low-res noise upsampled + a little high-res noise
    low = torch.randn(n_scenes * 3, PLANE_CH, 8, 8, generator=g)
    up = torch.nn.functional.interpolate(low, size=(PLANE_RES, PLANE_RES), mode='bilinear', align_corners=True)
    hi = 0.25 * torch.randn(n_scenes * 3, PLANE_CH, PLANE_RES, PLANE_RES, generator=g)
    return (up + hi).view(n_scenes, 3, PLANE_CH, PLANE_RES, PLANE_RES).contiguous()


def run_case(name, c, planes_all):
    g = torch.Generator().manual_seed(sum(ord(ch) * (i + 1) for i, ch in enumerate(name)))
    B, H, W, S, A = c['B'], c['H'], c['W'], c['S'], c['A']
    planes = planes_all[:B]
    vd = bool(c.get('viewdir', False))
    cfg_args = types.SimpleNamespace(use_viewdir=vd, use_sdf=c['sdf'], attention_values=A,
                                     fine_sampling=c['fine'])
    dataset_config = {'scene_range': c['scene_range'], 'white_background': c['white']}
    render, _ = load_reference_render(cfg_args, dataset_config)

    torch.manual_seed(1234)
    gen = ref_gen.Generator(512, c['scene_range'], attention_values=A, use_sdf=c['sdf'],
                            disable_stylegan_noise=True, use_viewdir=vd)
    gen.eval()
    # plane producer stub: the StyleGAN2 synthesis network is outside the hot path
    class PlaneStub(torch.nn.Module):
        def forward(self, ws, **kw):
            return planes.reshape(B, 3 * PLANE_CH, PLANE_RES, PLANE_RES)
    gen.synthesis_network = PlaneStub()
    with torch.no_grad():
        gen.decoder.net[0].weight.copy_(torch.randn(64, 32, generator=g))
        gen.decoder.net[0].bias.copy_(0.5 * torch.randn(64, generator=g))
        gen.decoder.net[2].weight.copy_(torch.randn(gen.decoder.net[2].weight.shape[0], 64, generator=g))
        gen.decoder.net[2].bias.copy_(0.5 * torch.randn(gen.decoder.net[2].bias.shape[0], generator=g))
        if c['sdf']:
            gen.beta.fill_(c['beta'])
            gen.alpha.fill_(c['alpha'])
        if vd:     # the output layer is zero-initialised in the reference (generator.py:217-219)
            gen.viewdir_mapper.output.weight.copy_(torch.randn(gen.viewdir_mapper.output.weight.shape, generator=g))
            gen.viewdir_mapper.output.bias.copy_(0.5 * torch.randn(gen.viewdir_mapper.output.bias.shape, generator=g))
    ray_feature = []
    if vd:
        gen.viewdir_mapper.fc6.register_forward_hook(lambda m, i, o_: ray_feature.append(o_.detach().clone()))
    cam = look_at_cameras(B, c['radius'], g)
    focal = None if c['ortho'] else torch.full((B,), c['focal']) * (1 + 0.05 * torch.randn(B, generator=g))
    bbox = None
    if c['bbox']:
        start = -0.8 + 0.2 * torch.rand(B, 2, generator=g)
        extent = 1.4 + 0.4 * torch.rand(B, 2, generator=g)
        bbox = torch.stack((start, extent), dim=1)           # [B,2,2]
    center = None
    if c.get('center'):
        center = 0.5 + 0.15 * (torch.rand(B, 2, generator=g) - 0.5)       # principal point around the image centre
    coords = bool(c.get('coords', False))
    num_ws = 15 if A > 0 else 14
    ws = torch.randn(B, num_ws, 512, generator=g)
    att = None
    extra_in = {}
    if A > 0:
        att = torch.sigmoid(torch.randn(B, A, 3, generator=g)) * 2.004 - 1.002
        extra_in = {'attention_values': att}

    rec = Recorder()
    rec.wrap(ref_nu, 'get_ray_bundle', ['ro', 'rd_raw'])
    rec.wrap(ref_nu, 'compute_near_far_planes', ['near', 'far'])
    rec.wrap(ref_nu, 'compute_query_points_from_rays', ['x_coarse', 't_coarse'])
    rec.wrap(ref_nu, 'render_volume_density_weights_only', ['weights_coarse'])
    rec.wrap(ref_nu, 'sample_pdf', ['t_fine'])
    sampler_log = []
    orig_forward = gen.forward

    def fwd(*a, **k):
        out = orig_forward(*a, **k)
        smp = out['sampler']

        def tapped(x_in, req=['sigma', 'rgb']):
            r = smp(x_in, req)
            sampler_log.append({k2: v.detach().clone() for k2, v in r.items()})
            return r
        out['sampler'] = tapped
        return out
    gen.forward = fwd

    noise_gen = torch.Generator().manual_seed(4321)
    with torch.no_grad(), NoiseTap(noise_gen) as tap:
        rgb, depth, mask, normals, sem, _ = render(
            gen, H, W, cam, focal, center, bbox, ws, S, randomize=c['randomize'],
            compute_semantics=(A > 0 and not coords), compute_coords=coords, extra_model_inputs=extra_in,
            force_no_cam_grad=coords)
    rec.restore()
    noise_c = tap.draws[0] if c['randomize'] else None
    noise_f = tap.draws[1] if (c['randomize'] and c['fine']) else None

    dec = gen.decoder.net
    w1, b1, w2, b2 = dec[0].weight.detach(), dec[0].bias.detach(), dec[2].weight.detach(), dec[2].bias.detach()
    beta = gen.beta.detach() if c['sdf'] else None
    alpha = gen.alpha.detach() if c['sdf'] else None

    viewdir = None
    if vd:
        viewdir = dict(x=ray_feature[0].reshape(B, H * W, 32), w3=gen.viewdir_mapper.output.weight.detach(),
                       b3=gen.viewdir_mapper.output.bias.detach())
    # ---- oracle must reproduce the reference bit for bit -------------------
    with torch.no_grad():
        o = orc.render(planes, w1, b1, w2, b2, cam, focal, H, W, S, c['scene_range'],
                       white_background=c['white'], fine_sampling=c['fine'], bbox=bbox, center=center,
                       want_coords=coords, noise_coarse=noise_c, noise_fine=noise_f, use_sdf=c['sdf'], beta=beta,
                       alpha=alpha, attention_values=att, want_semantics=(A > 0), viewdir=viewdir)

    def same(a, b, what):
        assert a.shape == b.shape, (name, what, a.shape, b.shape)
        assert torch.equal(a, b), (name, what, (a - b).abs().max().item())
    same(o['rgb'], rgb, 'rgb')
    same(o['depth'], depth, 'depth')
    same(o['mask'], mask, 'mask')
    if A > 0 or coords:
        same(o['semantics'], sem, 'semantics / coords map')
    same(o['ro'], rec.log['ro'][0].expand_as(o['ro']), 'ro')
    same(o['near'], rec.log['near'][0], 'near')
    same(o['far'], rec.log['far'][0], 'far')
    same(o['t_coarse'], rec.log['t_coarse'][0], 't_coarse')
    shp = o['t_coarse'].shape
    same(o['sigma_coarse'], sampler_log[0]['sigma'].view(*shp), 'sigma_coarse')
    same(o['rgb_coarse'], sampler_log[0]['rgb'].view(*shp, 3), 'rgb_coarse')
    if c['fine']:
        same(o['weights_coarse'], rec.log['weights_coarse'][0].flatten(0, 2), 'weights_coarse')
        same(o['t_fine'].flatten(0, 2), rec.log['t_fine'][0], 't_fine')
        same(o['sigma_fine'], sampler_log[1]['sigma'].view(*shp), 'sigma_fine')
        same(o['rgb_fine'], sampler_log[1]['rgb'].view(*shp, 3), 'rgb_fine')

    out = dict(cam2world=cam, w1=w1, b1=b1, w2=w2, b2=b2,
               ref_rgb=rgb, ref_depth=depth, ref_mask=mask,
               ref_ro=o['ro'], ref_rd=o['rd'], ref_near=o['near'], ref_far=o['far'], ref_hit=o['hit'],
               ref_t_coarse=o['t_coarse'], ref_sigma_coarse=o['sigma_coarse'],
               ref_rgb_coarse=o['rgb_coarse'], ref_outside_coarse=o['outside_coarse'],
               ref_sdf_coarse=o['sdf_coarse'], ref_weights=o['weights'])
    if focal is not None:
        out['focal'] = focal
    if bbox is not None:
        out['bbox'] = bbox
    if center is not None:
        out['center'] = center
    if att is not None:
        out['attention_values'] = att
        out['ref_coords_map' if coords else 'ref_semantics'] = sem
    if c['sdf']:
        out['beta'] = beta
        out['alpha'] = alpha
    if vd:
        out.update(viewdir_x=viewdir['x'], w3=viewdir['w3'], b3=viewdir['b3'])
    if noise_c is not None:
        out['noise_coarse'] = noise_c
    if noise_f is not None:
        out['noise_fine'] = noise_f
    if c['fine']:
        out.update(ref_weights_coarse=o['weights_coarse'], ref_weights_smooth=o['weights_smooth'],
                   ref_cdf=o['cdf'], ref_inds=o['inds'], ref_t_fine=o['t_fine'],
                   ref_sigma_fine=o['sigma_fine'], ref_rgb_fine=o['rgb_fine'],
                   ref_perm=o['perm'], ref_t_sorted=o['t_sorted'])
    meta = {k: v for k, v in c.items()}
    return out, meta


def check_cfg1():
    """BASELINE cfg1 at its exact shape on the CPU: the REAL Generator (StyleGAN2 synthesis network and all, random
    init, seed 1234) renders 4 scenes at 64x64 with 32 coarse samples (and once more with 32 + 32) through the live
    run.py::render; the oracle, fed the planes / colour table the generator produced and the same two noise draws,
    must reproduce rgb, depth and mask bit for bit.  Nothing is written."""
    B, R, S = 4, 64, 32
    for fine in (False, True):
        cfg_args = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=fine)
        dataset_config = {'scene_range': 0.55, 'white_background': True}
        render, _ = load_reference_render(cfg_args, dataset_config)
        torch.manual_seed(1234)
        gen = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True).eval()
        g = torch.Generator().manual_seed(5)
        cam = look_at_cameras(B, 2.0, g)
        focal = torch.full((B,), 1.0254)
        z = torch.randn(B, 512, generator=g)
        with torch.no_grad():
            # the random-init field is almost empty: centre the distance output so that surfaces exist
            probe = (torch.rand(B, 2048, 3, generator=g) * 2 - 1) * 0.55
            d = gen(None, z, ['sampler'])['sampler'](probe, ['sdf_distance'])['sdf_distance']
            gen.decoder.net[2].bias[0] -= d.median()
            gen.alpha.fill_(0.05)
        seen = {}
        hook = gen.synthesis_network.register_forward_hook(lambda m, i, o_: seen.__setitem__('planes', o_.detach()))
        noise_gen = torch.Generator().manual_seed(4321)
        with torch.no_grad(), NoiseTap(noise_gen) as tap:
            rgb, depth, mask, _, _, extra = render(gen, R, R, cam, focal, None, None, z, S,
                                                   extra_model_outputs=['attention_values'])
        hook.remove()
        planes = seen['planes'].view(B, 3, 32, 256, 256)
        dec = gen.decoder.net
        with torch.no_grad():
            o = orc.render(planes, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, cam, focal, R, R, S, 0.55,
                           white_background=True, fine_sampling=fine, noise_coarse=tap.draws[0],
                           noise_fine=tap.draws[1] if fine else None, use_sdf=True, beta=gen.beta, alpha=gen.alpha,
                           attention_values=extra['attention_values'])
        for k, ref in (('rgb', rgb), ('depth', depth), ('mask', mask)):
            assert torch.equal(o[k], ref), ('cfg1', fine, k, (o[k] - ref).abs().max().item())
        print('cfg1 (B=4, 64x64, 32%s samples, real Generator): oracle == reference bit for bit; mask mean %.3f' % (
            '+32' if fine else '', mask.mean().item()))


def check_normals():
    """The NORMAL MAP of the oracle pinned to the live reference on the CPU: the real Generator renders 2 scenes at 32x32 with
    16 + 16 samples through run.py::render(compute_normals=True) - normals = normalize(autograd of the SDF w.r.t. the query
    points), models/generator.py:599-623, composited with detached weights + white background, lib/nerf_utils.py:146-159 -;
    the oracle's map is what tests/parity_util.oracle_normal_map builds (autograd of the oracle's own field at the oracle's
    samples, its weights, its permutation).  Same samples bit for bit (same noise), so the maps agree to rounding: the
    HIP-vs-oracle bound of tests/test_hip_parity.py is then a bound against the reference.  Nothing is written."""
    B, R, S = 2, 32, 16
    cfg_args = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    dataset_config = {'scene_range': 0.55, 'white_background': True}
    render, _ = load_reference_render(cfg_args, dataset_config)
    torch.manual_seed(1234)
    gen = ref_gen.Generator(512, 0.55, attention_values=10, use_sdf=True, disable_stylegan_noise=True).eval()
    g = torch.Generator().manual_seed(6)
    cam = look_at_cameras(B, 2.0, g)
    focal = torch.full((B,), 1.0254)
    z = torch.randn(B, 512, generator=g)
    with torch.no_grad():
        probe = (torch.rand(B, 2048, 3, generator=g) * 2 - 1) * 0.55
        d = gen(None, z, ['sampler'])['sampler'](probe, ['sdf_distance'])['sdf_distance']
        gen.decoder.net[2].bias[0] -= d.median()
        gen.alpha.fill_(0.05)
    gen.requires_grad_(False)
    seen = {}
    hook = gen.synthesis_network.register_forward_hook(lambda m, i, o_: seen.__setitem__('planes', o_.detach()))
    noise_gen = torch.Generator().manual_seed(4321)
    with NoiseTap(noise_gen) as tap:                                     # (grad mode ON: the reference asserts it)
        rgb, depth, mask, normals, _, extra = render(gen, R, R, cam, focal, None, None, z, S, compute_normals=True,
                                                     extra_model_outputs=['attention_values'])
    hook.remove()
    assert normals is not None and normals.shape == (B, R, R, 3)
    planes = seen['planes'].view(B, 3, 32, 256, 256)
    dec = gen.decoder.net
    w = [t.detach() for t in (dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias)]
    att = extra['attention_values'].detach()
    with torch.no_grad():
        o = orc.render(planes, *w, cam, focal, R, R, S, 0.55, white_background=True, fine_sampling=True,
                       noise_coarse=tap.draws[0], noise_fine=tap.draws[1], use_sdf=True, beta=gen.beta.detach(),
                       alpha=gen.alpha.detach(), attention_values=att)
    for k, ref in (('rgb', rgb), ('depth', depth), ('mask', mask)):
        assert torch.equal(o[k], ref.detach()), ('normals case', k, (o[k] - ref).abs().max().item())

    def oracle_normals(pts):
        p = pts.clone().requires_grad_()
        q = orc.field_query(planes, *w, p, 0.55, True, gen.beta.detach(), gen.alpha.detach(), att)
        gx, = torch.autograd.grad(q['sdf'].sum(), p)
        return torch.nn.functional.normalize(gx, dim=-1)
    n_c = oracle_normals(orc.points_on_rays(o['ro'], o['rd'], o['t_coarse']).reshape(B, -1, 3)).view(B, R, R, -1, 3)
    n_f = oracle_normals(orc.points_on_rays(o['ro'], o['rd'], o['t_fine']).reshape(B, -1, 3)).view(B, R, R, -1, 3)
    n_all = torch.cat((n_c, n_f), dim=-2).gather(-2, o['perm'].unsqueeze(-1).expand(-1, -1, -1, -1, 3))
    ours = (o['weights'][..., None] * n_all).sum(dim=-2) + (1. - o['mask'][..., None])
    err = (ours - normals.detach()).abs().max().item()
    assert err <= 2e-6, ('normal map', err)
    print('normal map (B=2, 32x32, 16+16 samples, real Generator): oracle == reference to %.1e (rgb / depth / mask bit for bit); '
          'mask mean %.3f' % (err, mask.mean().item()))


def slice_functions(path, names, g):
    """exec the named top-level functions of a reference file that cannot be imported as a module."""
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), (path, names)
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), g)
    return g


def neighbours(gold):
    """Pins oracle/nfi_oracle_neighbours.py to the live augment_impl (run.py) and psnr / iou (lib/metrics.py) and
    writes tests/golden/neighbours.npz."""
    import torch.nn.functional as F
    from lib import pose_utils
    from oracle import nfi_oracle_neighbours as orn
    out = {}
    g = torch.Generator().manual_seed(2024)
    for white in (False, True):
        env = {'torch': torch, 'np': np, 'F': F, 'pose_utils': pose_utils,
               'args': types.SimpleNamespace(supervise_alpha=not white), 'dataset_config': {'white_background': white}}
        slice_functions(os.path.join(REF, 'run.py'), ['augment_impl'], env)
        bs, C, H, W = 6, 6, 20, 28
        img = torch.rand(bs, C, H, W, generator=g) * 2 - 1
        rot = (torch.rand(bs, generator=g) - 0.5) * 2 * np.pi
        scale = torch.exp2(torch.randn(bs, generator=g) * 0.2)
        trans = torch.randn(bs, 2, generator=g) * 0.1
        rot[0], scale[0], trans[0] = 0.0, 1.0, 0.0              # identity transform
        ref, _, _, _ = env['augment_impl'](img.clone(), None, None, 1.0, cached_tform=(rot, scale, trans))
        mine = orn.warp_images(img, rot, scale, trans, white)
        assert torch.equal(ref, mine), ('warp', white, (ref - mine).abs().max().item())
        tag = 'white' if white else 'black'
        out.update({'warp_%s_img' % tag: img, 'warp_%s_rot' % tag: rot, 'warp_%s_scale' % tag: scale,
                    'warp_%s_trans' % tag: trans, 'warp_%s_ref' % tag: ref})
    env = slice_functions(os.path.join(REF, 'lib', 'metrics.py'), ['range_check', 'psnr', 'iou'], {'torch': torch})
    pred = torch.rand(5, 3, 24, 24, generator=g) * 1.08 - 0.04          # inside the range check, outside [0,1] in places
    target = (pred + 0.1 * torch.randn(5, 3, 24, 24, generator=g)).clamp(-0.05, 1.05)
    target[4] = pred[4].clamp(0, 1)                                      # perfect reconstruction -> clamped at 60 dB
    ref_psnr = env['psnr'](pred, target, reduction='none')
    assert torch.equal(ref_psnr, orn.psnr(pred, target)) and float(ref_psnr[4]) == 60.0
    assert torch.equal(env['psnr'](pred, target), orn.psnr(pred, target).mean())
    a = torch.rand(5, 24, 24, generator=g)
    b = (a + 0.3 * torch.randn(5, 24, 24, generator=g)).clamp(0, 1)
    a[3], b[3] = 0.1, 0.2                                                # both masks empty: iou = 1
    ref_iou = env['iou'](a, b, reduction='none')
    assert torch.equal(ref_iou, orn.iou(a, b)) and float(ref_iou[3]) == 1.0
    assert torch.equal(env['iou'](a.unsqueeze(1), b.unsqueeze(1), reduction='none'), ref_iou)
    out.update(metric_pred=pred, metric_target=target, metric_psnr=ref_psnr, metric_mask_a=a, metric_mask_b=b,
               metric_iou=ref_iou)
    # ---- the tail of the last synthesis block (stylegan.py:383-435) against the live SynthesisBlock ----
    from models import stylegan as ref_sg
    torch.manual_seed(77)
    blk = ref_sg.SynthesisBlock(32, 32, w_dim=24, resolution=16, img_channels=96, is_last=True, use_noise=False).eval()
    with torch.no_grad():
        blk.torgb.bias.normal_()
    x_in = torch.randn(2, 32, 8, 8, generator=g)
    img_prev = torch.randn(2, 96, 8, 8, generator=g)
    ws = torch.randn(2, 3, 24, generator=g)
    seen = {}
    hk = blk.conv1.register_forward_hook(lambda m, i, o_: seen.__setitem__('x', o_.detach().clone()))
    with torch.no_grad():
        _, img = blk(x_in, img_prev.clone(), ws)
        styles = blk.torgb.affine(ws[:, 2]) * blk.torgb.weight_gain
        mine = orn.torgb_upsample_add(seen['x'], styles, blk.torgb.weight, blk.torgb.bias, img_prev)
    hk.remove()
    assert torch.equal(img, mine), ('torgb tail', (img - mine).abs().max().item())
    out.update(handoff_x=seen['x'], handoff_styles=styles.detach(), handoff_weight=blk.torgb.weight.detach().reshape(96, 32),
               handoff_bias=blk.torgb.bias.detach(), handoff_prev=img_prev, handoff_ref=img)
    np.savez(os.path.join(gold, 'neighbours.npz'), **{k: v.numpy() for k, v in out.items()})
    print('neighbours                   ok  (augment_impl warp black/white, psnr, iou, synthesis-block tail pinned to the live code)')


def main():
    if '--cfg1' in sys.argv:
        return check_cfg1()
    if '--normals' in sys.argv:
        return check_normals()
    import json
    gold = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(gold, exist_ok=True)
    planes = make_planes(2, 777)
    np.savez(os.path.join(gold, 'planes.npz'), planes=planes.numpy())
    metas = {}
    for name, c in CASES.items():
        out, meta = run_case(name, c, planes)
        np.savez(os.path.join(gold, name + '.npz'), **{k: v.numpy() for k, v in out.items()})
        metas[name] = meta
        print('%-28s ok  mask mean %.3f  rgb mean %.3f  hit %.2f' %
              (name, out['ref_mask'].mean().item(), out['ref_rgb'].mean().item(),
               out['ref_hit'].float().mean().item()))
    neighbours(gold)
    json.dump(metas, open(os.path.join(gold, 'cases.json'), 'w'), indent=1, sort_keys=True)
    print('torch', torch.__version__, '-> wrote', gold)


if __name__ == '__main__':
    main()
