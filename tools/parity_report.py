"""Measured parity of the HIP path against the oracle, per case (GPU box): max |sigma| error absolute and relative to
max(1, sigma), searchsorted-index flip rate, sort-permutation flip rate, max |rgb / depth / mask| error.  The asserts
in tests/ are set to about twice these numbers.   python tools/parity_report.py > gpurun_out/parity_report.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def main():
    from conftest import golden_case_names, load_golden
    from parity_util import err, hip_render, oracle_normal_map, oracle_render
    from nerf_from_image_amd import ops
    from oracle import nfi_oracle as orc
    import test_hip_full_size as fs
    dev = torch.device('cuda:0')
    rep = {}
    for name in golden_case_names():
        meta, t = load_golden(name)
        if meta['S'] > 128 and meta['fine']:
            continue                      # (no such case: with fine sampling a pass holds at most 128 samples)
        o = oracle_render(meta, t, 'cpu')
        r = hip_render(meta, t, dev, taps=ops.TAP_NAMES)
        e = {k: err(r[k], o[k])['max'] for k in ('rgb', 'depth', 'mask')}
        sc, so = r['sigma_coarse'].cpu(), o['sigma_coarse']
        e['sigma_abs'] = float((sc - so).abs().max())
        e['sigma_rel'] = float(((sc - so).abs() / so.abs().clamp_min(1.0)).max())
        e['sigma_max'] = float(so.max())
        if meta['fine']:
            e['perm_flip_rate'] = float((r['perm'].cpu().long() != o['perm']).float().mean())
            n = o['weights_coarse'].shape[0]
            u = t['noise_fine'].to(dev) if 'noise_fine' in t else orc.deterministic_u(n, meta['S'], o['weights_coarse']).to(dev)
            _, taps = ops.resample(o['sigma_coarse'].to(dev), o['rd'].to(dev), o['t_coarse'].to(dev), u, want_taps=True)
            e['inds_flip_rate'] = float((taps['inds'].cpu() != o['inds']).float().mean())
            e['t_fine'] = err(r['t_fine'], o['t_fine'])['max']
        if meta['sdf'] and 'viewdir_x' not in t and meta['S'] <= 128:
            # the composited normal map (fused kernel, analytic derivative) against autograd of the oracle's distance
            ref_map = oracle_normal_map(meta, t, o)
            e['normal_map'] = err(hip_render(meta, t, dev, skip_missed_rays=True, want_normals=True)['normals'], ref_map)['max']
            e16 = err(hip_render(meta, t, dev, skip_missed_rays=True, texel_dtype=ops.TEXEL_F16, want_normals=True)['normals'], ref_map)
            e['normal_map_fp16_texels_max'], e['normal_map_fp16_texels_mean'] = e16['max'], e16['mean']
        rep[name] = e
    for tag, B, radius, seed, R, S in (('cfg2_b1', 1, 1.6, 1234, 128, 64), ('cfg2_b8_chairs', 8, 2.0, 1234, 128, 64),
                                       ('cfg2_b8_all_hit', 8, 1.3, 77, 128, 64), ('cfg5_b1', 1, 1.6, 1234, 256, 128)):
        d = fs.make_inputs(B, dev, radius=radius, seed=seed, R=R, S=S)
        r = fs.hip(d, taps=('perm', 't_fine', 'sigma_coarse'))
        o = fs.oracle(d, dev)
        e = {k: err(r[k], o[k])['max'] for k in ('rgb', 'depth', 'mask')}
        e['perm_flip_rate'] = float((r['perm'].long() != o['perm']).float().mean())
        e['t_fine'] = err(r['t_fine'], o['t_fine'])['max']
        so = o['sigma_coarse']
        rel = (r['sigma_coarse'] - so).abs() / so.abs().clamp_min(1.0)
        # a coarse sample within an ulp of a cube face can be classified inside by one implementation and outside by the
        # other (sigma vs exactly 0): those few samples are counted separately, the maximum is taken over the rest
        face = ((r['sigma_coarse'] == 0) != (so == 0))
        e['sigma_cube_face_flips'] = int(face.sum())
        e['sigma_samples'] = int(so.numel())
        e['sigma_rel'] = float(rel[~face].max())
        e['sigma_abs'] = float((r['sigma_coarse'] - so).abs()[~face].max())
        e['mask_mean'] = float(o['mask'].mean())
        rep[tag + '_vs_pytorch_rocm_oracle'] = e
        del d, r, o
        torch.cuda.empty_cache()
    print(json.dumps(rep, indent=1))


if __name__ == '__main__':
    main()
