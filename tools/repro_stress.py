"""Bit-reproducibility stress of the forward kernels on the workloads bench.py times (run on the GPU box):

    python tools/repro_stress.py [launches=1000]

Each workload is launched `launches` times with identical inputs; every output is compared bit for bit with the
majority (element-wise median) of the first five launches.  Nothing here uses atomics on floats, so ANY difference is
a wrong result of the kind HISTORY.md "Determinism" describes.  Prints one line per workload and a JSON summary.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from nerf_from_image_amd import ops  # noqa: E402


def stress(name, fn, n, keys):
    first = [fn() for _ in range(5)]
    ref = {k: torch.stack([f[k].clone().view(torch.int32) for f in first]).median(dim=0).values for k in keys}
    bad = torch.zeros((), dtype=torch.int64, device='cuda')
    launches_bad = torch.zeros((), dtype=torch.int64, device='cuda')
    for _ in range(n):
        out = fn()
        cnt = sum((out[k].view(torch.int32) != ref[k]).sum() for k in keys)
        bad += cnt
        launches_bad += (cnt > 0).long()
    torch.cuda.synchronize()
    r = {'launches': n, 'differing_elements': int(bad), 'launches_with_a_difference': int(launches_bad),
         'elements_per_launch': int(sum(ref[k].numel() for k in keys))}
    print('%-44s %s' % (name, r), flush=True)
    return r


def render_case(dev, n_img, radius, tdt, R=bench.R, S=bench.S, tuning=0, **render_kw):
    dd = bench.synthetic_inputs(n_img, 4321, dev)
    g = torch.Generator().manual_seed(77)
    dd['cam'] = bench.cameras(n_img, radius, g).to(dev)
    texels = ops.planes_to_texels(dd['planes'], tdt)
    image = ops.decoder_pack(dd['w1'], dd['b1'], dd['w2'], dd['b2'], bench.A, tdt)
    gn = torch.Generator(device=dev).manual_seed(99)
    nc = torch.rand((n_img, R, R, S), device=dev, generator=gn)
    nf = torch.rand((n_img * R * R, S), device=dev, generator=gn)
    state = {'ws': None}

    def fn():
        out = ops.render_fwd(dd['cam'], dd['focal'], R, R, S, texels, image, bench.SCENE_RANGE, bench.A, dd['att'], True,
                             dd['beta'], dd['alpha'], noise_coarse=nc, noise_fine=nf, workspace=state['ws'], tuning=tuning, **render_kw)
        state['ws'] = out['_workspace']
        return out
    return fn


def field_case(dev, mlp_precision):
    dd = bench.synthetic_inputs(2, 99, dev)
    texels = ops.planes_to_texels(dd['planes'])
    image = ops.decoder_pack(dd['w1'], dd['b1'], dd['w2'], dd['b2'], bench.A)
    g = torch.Generator(device=dev).manual_seed(5)
    x = (torch.rand((2, 1 << 20, 3), device=dev, generator=g) * 2 - 1) * bench.SCENE_RANGE * 1.1

    def fn():
        return ops.field_query(x, texels, image, bench.SCENE_RANGE, bench.A, dd['att'], True, dd['beta'], dd['alpha'],
                               want_sdf=True, mlp_precision=mlp_precision)
    return fn


def sdf_gradient_cases(dev):
    dd = bench.synthetic_inputs(4, 7, dev)
    texels = ops.planes_to_texels(dd['planes'])
    g = torch.Generator(device=dev).manual_seed(6)
    x = (torch.rand((4, 31 ** 3, 3), device=dev, generator=g) * 2 - 1) * bench.SCENE_RANGE * 0.99
    w2, b2 = dd['w2'], dd['b2']

    def fwd():
        s, gr = ops.sdf_gradient_fwd(x, texels, dd['w1'], dd['b1'], w2, b2, bench.SCENE_RANGE)
        return {'sdf': s, 'gradient': gr}
    return fwd


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dev = torch.device('cuda:0')
    res = {}
    res['render_b8_chairs'] = stress('render 8 x 128^2 64+64 chairs-like', render_case(dev, 8, bench.RADIUS, ops.TEXEL_F32), n,
                                     ('rgb', 'depth', 'mask'))
    res['render_b8_all_hit'] = stress('render 8 x 128^2 64+64 every ray hits', render_case(dev, 8, 1.3, ops.TEXEL_F32), n,
                                      ('rgb', 'depth', 'mask'))
    res['render_b8_all_hit_exact_fp32_mlp'] = stress('render 8 x 128^2 every ray hits, exact-fp32 MLP',
                                                     render_case(dev, 8, 1.3, ops.TEXEL_F32, tuning=8), max(n // 4, 50),
                                                     ('rgb', 'depth', 'mask'))
    res['render_cfg5_fp16'] = stress('render 2 x 256^2 128+128 fp16 texels (cfg5)',
                                     render_case(dev, 2, bench.RADIUS, ops.TEXEL_F16, R=256, S=128), max(n // 2, 50),
                                     ('rgb', 'depth', 'mask'))
    # round 5: the packed bf16 tile (three workgroups per CU) and the 128 + 128 kernel with its 16-bit semantics table
    res['render_b8_chairs_bf16'] = stress('render 8 x 128^2 64+64 chairs-like, bf16 texels',
                                          render_case(dev, 8, bench.RADIUS, ops.TEXEL_BF16), n, ('rgb', 'depth', 'mask'))
    res['render_b8_all_hit_bf16'] = stress('render 8 x 128^2 64+64 every ray hits, bf16 texels',
                                           render_case(dev, 8, 1.3, ops.TEXEL_BF16), max(n // 2, 50), ('rgb', 'depth', 'mask'))
    res['render_cfg5_semantics'] = stress('render 2 x 256^2 128+128 + semantics map (cfg5)',
                                          render_case(dev, 2, bench.RADIUS, ops.TEXEL_F32, R=256, S=128, want_semantics=True),
                                          max(n // 4, 50), ('rgb', 'depth', 'mask', 'semantics'))
    # round 6: the normal map's split-fp16 contraction (new K = 32 MFMAs next to packed-fp32 code), fp32 / bf16 texels, 128 + 128
    res['render_b8_chairs_normals'] = stress('render 8 x 128^2 64+64 chairs-like + normals map',
                                             render_case(dev, 8, bench.RADIUS, ops.TEXEL_F32, want_normals=True), max(n // 2, 50),
                                             ('rgb', 'depth', 'mask', 'normals'))
    res['render_b8_all_hit_normals_bf16'] = stress('render 8 x 128^2 64+64 every ray hits + normals map, bf16 texels',
                                                   render_case(dev, 8, 1.3, ops.TEXEL_BF16, want_normals=True), max(n // 4, 50),
                                                   ('rgb', 'depth', 'mask', 'normals'))
    res['render_cfg5_normals_semantics'] = stress('render 2 x 256^2 128+128 + normals + semantics maps (cfg5)',
                                                  render_case(dev, 2, bench.RADIUS, ops.TEXEL_F32, R=256, S=128, want_normals=True,
                                                              want_semantics=True), max(n // 4, 50),
                                                  ('rgb', 'depth', 'mask', 'normals', 'semantics'))
    res['field_query_exact'] = stress('field_query_kernel 2 x 1 Mi points, exact fp32', field_case(dev, 0), max(n // 2, 50),
                                      ('sigma', 'rgb', 'sdf'))
    res['field_query_split'] = stress('field_query_kernel 2 x 1 Mi points, split fp16', field_case(dev, 1), max(n // 2, 50),
                                      ('sigma', 'rgb', 'sdf'))
    res['sdf_gradient_fwd'] = stress('sdf_gradient_fwd_kernel 4 x 31^3 points', sdf_gradient_cases(dev), n, ('sdf', 'gradient'))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
