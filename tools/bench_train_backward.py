"""Timing of the field backward + binned plane-gradient scatter on the REAL inputs of a cfg4-like training step
(4 images, 128 x 128 orthographic rays, 64 + 64 samples: the points, upstream gradients and zero rows that the one-node
render of nerf_from_image_amd.render hands to field_query_bwd), replayed N times with HIP events.

    python tools/bench_train_backward.py [N]            NFI_PROBE_LIBRARY=<variant .so> selects a variant build
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
from nerf_from_image_amd import _lib  # noqa: E402
if os.environ.get('NFI_PROBE_LIBRARY'):
    _lib.LIBRARY = os.environ['NFI_PROBE_LIBRARY']


def capture(dev, batch=4, res=128, samples=64, plane_res=256):
    from stand_in import StandInGenerator, look_at_cameras
    import nerf_from_image_amd.generator as nfi_gen
    import nerf_from_image_amd.render as nfi_render
    torch.manual_seed(0)
    model = StandInGenerator(2.0, attention_values=10, use_sdf=True, plane_res=plane_res).to(dev).train()
    with torch.no_grad():
        model.alpha.fill_(0.05)
    nfi_gen.attach(model)
    g = torch.Generator().manual_seed(1)
    cam = look_at_cameras(batch, 3.0, g).to(dev)
    z = torch.randn(batch, 512, generator=g).to(dev)
    target_rgb = (torch.rand(batch, res, res, 3, generator=g) * 2 - 1).to(dev)
    target_mask = (torch.rand(batch, res, res, generator=g) > 0.5).float().to(dev)
    cfg = types.SimpleNamespace(use_viewdir=False, use_sdf=True, attention_values=10, fine_sampling=True)
    render = nfi_render.make_render(cfg, {'scene_range': 2.0, 'white_background': False})
    calls = []
    real = nfi_render.field_query_bwd

    def spy(*a, **k):
        calls.append((a, k))
        return real(*a, **k)
    nfi_render.field_query_bwd = spy
    rgb, _, mask, _, _, _ = render(model, res, res, cam, None, None, None, z, samples)
    loss = ((rgb - target_rgb) ** 2).mean() + ((mask - target_mask) ** 2).mean()
    loss.backward()
    nfi_render.field_query_bwd = real
    assert len(calls) == 1
    return real, calls[0]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda:0')
    fn, (a, k) = capture(dev)
    a = tuple(t.detach() if torch.is_tensor(t) else t for t in a)
    if os.environ.get('NFI_NO_RAY_ORDER'):
        k = {n: v for n, v in k.items() if n != 'ray_order'}
    print('ray_order hint:', k.get('ray_order'))
    exact = fn(*a, **dict(k, scatter_mode=0))['g_texels'].double()
    modes = [int(m) for m in os.environ.get('NFI_SCATTER_MODES', '1').split(',')]
    for rnd in range(2 if len(modes) > 1 else 1):          # (two alternating rounds when modes are compared)
        for mode in modes:
            km = dict(k, scatter_mode=mode)
            out = fn(*a, **km)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            for i in range(n):
                ev[i].record()
                out = fn(*a, **km)
            ev[n].record()
            torch.cuda.synchronize()
            ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
            # the plane gradient against the per-point atomic scatter (scatter_mode 0: fp32 rows never leave the registers)
            got = out['g_texels'].double()
            err_max = float((got - exact).abs().max() / exact.abs().max())
            err_l2 = float((got - exact).norm() / exact.norm())
            pts = a[0].shape[0] * a[0].shape[1]
            gs = a[11]
            print('%s scatter_mode %d: %.1f M points (%.1f %% with a non-zero sigma gradient): field backward + scatter %.3f ms median '
                  '(min %.3f)  |g_texels| %.9e  sum %.9e  |g_w1| %.9e  vs atomic scatter: max %.2e of max, l2 %.2e' % (
                      os.path.basename(_lib.LIBRARY), mode, pts / 1e6, 100.0 * float((gs != 0).float().mean()), ms[n // 2], ms[0],
                      float(out['g_texels'].double().norm()), float(out['g_texels'].double().sum()),
                      float(out['g_w1'].double().norm()), err_max, err_l2))


if __name__ == '__main__':
    main()
