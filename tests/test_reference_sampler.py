"""The `sampler(x_in, request_sampler_outputs)` closure (models/generator.py:587-681) that `attach()` puts behind
`Generator.forward(...)['sampler']`, against the REAL closure of the untouched reference Generator on PyTorch-ROCm: same
planes, same points (some outside the scene cube), every request the reference's callers make - values, and gradients to
the planes, the decoder's weights, alpha / beta, the colour table's producer and the points themselves.

The bounds are relative L2 errors (values: max |d| relative to the largest magnitude), measured on MI355X and asserted at
about twice the measured figure."""
import pytest
import torch

from oracle import reference

import reference_cases as rc

pytestmark = pytest.mark.gpu

REQUESTS = {
    'sigma_rgb': ['sigma', 'rgb'],                                               # run.py:229 (the render itself)
    'all': ['sdf_distance', 'sigma', 'rgb', 'semantics', 'coords'],              # run.py:230-237 + the regulariser's request
    'sigma_only': ['sigma'],                                                     # (no gradient to the colour table's producer)
    'rgb_only': ['rgb'],                                                         # (none to alpha / beta)
    'sdf_only': ['sdf_distance'],                                                # generator.py:505-585's request
}


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _rel_max(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _run(model, sc, planes96, x, request, cot, viewdir=None):
    model.requires_grad_(True)
    for p in model.parameters():
        p.grad = None
    pl = planes96.clone().requires_grad_(True)
    pts = x.clone().requires_grad_(True)
    with rc.frozen_producer(model, pl):
        sampler = model(viewdir, sc.ws, ['sampler'])['sampler']
        out = sampler(pts, request)
    assert sorted(out) == sorted(request), (sorted(out), request)
    loss = sum((out[k] * cot[k]).sum() for k in request)
    loss.backward()
    grads = {'planes': pl.grad, 'points': pts.grad}
    for name, p in model.named_parameters():
        if p.grad is not None and not name.startswith(('synthesis_network', 'mapping_network')):
            grads[name] = p.grad.clone()
    model.requires_grad_(False)
    return {k: v.detach() for k, v in out.items()}, grads


@pytest.mark.parametrize('geometry,request_name', [('chairs', 'sigma_rgb'), ('chairs', 'all'), ('chairs', 'sigma_only'),
                                                   ('chairs', 'rgb_only'), ('chairs', 'sdf_only'),
                                                   ('density', 'sigma_rgb'), ('carla', 'sigma_rgb'), ('p3d', 'all')])
def test_sampler_closure_values_and_gradients(gpu_device, geometry, request_name):
    if not reference.available():
        pytest.skip('reference sources not staged: run oracle/make_ref.py (or __graft_entry__.build()) where /root/reference exists')
    sc = rc.build_scene(geometry, 2, gpu_device)
    request = [r for r in REQUESTS[request_name]]
    B, R, S = 2, 24, 16
    g = torch.Generator(device=gpu_device).manual_seed(3)
    x = (torch.rand((B, R, R, S, 3), device=gpu_device, generator=g) * 2 - 1) * sc.g['scene_range'] * 1.15   # ~1/3 outside the cube
    viewdir = None
    if sc.g.get('viewdir'):
        viewdir = torch.nn.functional.normalize(torch.randn((B, R, R, 1, 3), device=gpu_device, generator=g), dim=-1)
    with torch.no_grad():
        planes96 = sc.gen.synthesis_network(sc.ws[:, :14])
        probe = sc.gen(viewdir, sc.ws, ['sampler'])['sampler'](x, request)
    cot = {k: torch.randn(probe[k].shape, device=gpu_device, generator=g) for k in request}
    with rc.deterministic_producer():
        ref_out, ref_g = _run(sc.gen, sc, planes96, x, request, cot, viewdir)
        hip_out, hip_g = _run(sc.hip, sc, planes96, x, request, cot, viewdir)
    report = {}
    for k in request:
        assert hip_out[k].shape == ref_out[k].shape, k
        report[k] = _rel_max(hip_out[k], ref_out[k])
    assert sorted(hip_g) == sorted(ref_g), (sorted(hip_g), sorted(ref_g))
    for k in ref_g:
        assert hip_g[k].shape == ref_g[k].shape, k
        report['g_' + k] = _rel_l2(hip_g[k], ref_g[k])
    print('SAMPLER', geometry, request_name, {k: '%.2e' % v for k, v in report.items()})
    for k in request:
        assert report[k] <= VALUE_BOUND, (k, report)
    for k in ref_g:
        assert report['g_' + k] <= GRADIENT_BOUND.get(k, GRADIENT_BOUND['*']), (k, report)


# measured (MI355X, two sessions): values <= 4.1e-6 (sigma; everything else <= 1.1e-6), gradients <= 3.7e-5 (the output layer's
# bias; planes <= 1.8e-5, points <= 7.2e-6, the colour table's producer <= 1.7e-6)
VALUE_BOUND = 1e-5
GRADIENT_BOUND = {'*': 1e-4, 'points': 3e-5}


def test_sampler_normals_request(gpu_device):
    """`normals` (generator.py:599-623): eval only, needs grad mode, every other output comes back detached."""
    if not reference.available():
        pytest.skip('reference sources not staged')
    sc = rc.build_scene('chairs', 2, gpu_device)
    g = torch.Generator(device=gpu_device).manual_seed(4)
    x = (torch.rand((2, 16, 16, 8, 3), device=gpu_device, generator=g) * 2 - 1) * sc.g['scene_range'] * 0.98
    outs = []
    for model in (sc.gen, sc.hip):
        sampler = model(None, sc.ws, ['sampler'])['sampler']
        outs.append(sampler(x.clone(), ['sigma', 'rgb', 'normals']))
    r, h = outs
    assert sorted(r) == sorted(h)
    for k in r:
        assert h[k].shape == r[k].shape and not h[k].requires_grad and not r[k].requires_grad, k
    assert _rel_max(h['sigma'], r['sigma']) <= 1e-5 and _rel_max(h['rgb'], r['rgb']) <= 1e-5
    # unit vectors from fp32 differences of neighbouring texels: compare where the gradient is not tiny
    cos = (h['normals'] * r['normals']).sum(-1)
    assert float((cos > 0.9999).float().mean()) > 0.999, float((cos > 0.9999).float().mean())
