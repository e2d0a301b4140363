"""Stage-by-stage error report of the HIP path against the oracle on the golden cases.
Diagnostics only (prints a table; asserts nothing).  Run on the GPU box:
    python tools/gpu_diag.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

from conftest import golden_case_names, load_golden  # noqa: E402
from parity_util import err, hip_field_setup, hip_render, oracle_render  # noqa: E402
from nerf_from_image_amd import ops  # noqa: E402
from oracle import nfi_oracle as orc  # noqa: E402


def line(name, e):
    print('   %-22s max %.3e  mean %.3e  exact %.4f  nonfinite %d' % (name, e['max'], e['mean'], e['exact'], e['nonfinite']))


def main():
    dev = torch.device('cuda:0')
    print(torch.cuda.get_device_name(0))
    for name in golden_case_names():
        meta, t = load_golden(name)
        print('== %s %s' % (name, {k: meta[k] for k in ('B', 'H', 'W', 'S', 'A', 'fine', 'sdf', 'white')}))
        o = oracle_render(meta, t, 'cpu')
        og = oracle_render(meta, t, dev)
        g = lambda k: t[k].to(dev) if k in t else None
        # --- rays
        ro, rd = ops.raygen(meta['H'], meta['W'], g('focal'), g('cam2world'), g('bbox'), None, normalize=True)
        line('ro', err(ro, o['ro'])); line('rd', err(rd, o['rd']))
        near, far, hit = ops.near_far(o['ro'].to(dev).contiguous(), o['rd'].to(dev), meta['scene_range'])
        line('near|oracle rays', err(near, o['near'])); line('far|oracle rays', err(far, o['far']))
        line('hit', err(hit.float(), o['hit'].float()))
        pts, dep = ops.stratified_points(o['ro'].to(dev).contiguous(), o['rd'].to(dev), o['near'].to(dev), o['far'].to(dev),
                                         meta['S'], g('noise_coarse'))
        line('t_coarse|oracle in', err(dep, o['t_coarse']))
        xo = orc.points_on_rays(o['ro'], o['rd'], o['t_coarse'])
        line('x_coarse|oracle in', err(pts, xo))
        # --- field
        texels, image = hip_field_setup(meta, t, dev)
        B = meta['B']
        fq = ops.field_query(xo.reshape(B, -1, 3).to(dev), texels, image, meta['scene_range'], meta['A'],
                             g('attention_values'), meta['sdf'], g('beta'), g('alpha'), want_sdf=True,
                             want_semantics=meta['A'] > 0, want_outside=True)
        line('field sdf', err(fq['sdf'], o['sdf_coarse'].reshape(B, -1)))
        line('field sigma', err(fq['sigma'], o['sigma_coarse'].reshape(B, -1)))
        line('field rgb', err(fq['rgb'], o['rgb_coarse'].reshape(B, -1, 3)))
        line('field outside', err(fq['outside'].float(), o['outside_coarse'].reshape(B, -1)))
        line('  (oracle gpu-vs-cpu sigma)', err(og['sigma_coarse'], o['sigma_coarse']))
        line('  (oracle gpu-vs-cpu rgb)', err(og['rgb'], o['rgb']))
        if meta['fine']:
            w = ops.ray_weights(o['sigma_coarse'].to(dev), o['rd'].to(dev), o['t_coarse'].to(dev))
            line('weights|oracle in', err(w.flatten(0, 2), o['weights_coarse']))
            S = meta['S']
            u = g('noise_fine') if 'noise_fine' in t else orc.deterministic_u(o['weights_coarse'].shape[0], S, o['weights_coarse']).to(dev)
            fine, taps = ops.resample(o['sigma_coarse'].to(dev), o['rd'].to(dev), o['t_coarse'].to(dev), u, want_taps=True)
            line('smooth', err(taps['smooth'], o['weights_smooth']))
            line('cdf', err(taps['cdf'], o['cdf']))
            line('inds', err(taps['inds'].float(), o['inds'].float()))
            exp_inds = torch.searchsorted(taps['cdf'].contiguous(), u.contiguous(), right=True)
            line('inds|kernel cdf', err(taps['inds'].float(), exp_inds.float()))
            line('t_fine', err(fine, o['t_fine'].flatten(0, 2)))
            rgbm, depm, maskm, _, ct = ops.composite(
                o['rd'].to(dev), o['t_coarse'].to(dev), o['sigma_coarse'].to(dev), o['rgb_coarse'].to(dev),
                o['t_fine'].to(dev), o['sigma_fine'].to(dev), o['rgb_fine'].to(dev), white_background=meta['white'], want_taps=True)
            line('composite rgb|oracle in', err(rgbm, o['rgb'])); line('composite depth', err(depm, o['depth']))
            line('composite mask', err(maskm, o['mask']))
            line('t_sorted', err(ct['depth_sorted'], o['t_sorted'])); line('perm', err(ct['perm'].float(), o['perm'].float()))
            line('weights', err(ct['weights'], o['weights']))
        # --- fused
        r = hip_render(meta, t, dev, taps=ops.TAP_NAMES)
        for k_h, k_o in (('rgb', 'rgb'), ('depth', 'depth'), ('mask', 'mask'), ('ray_directions', 'rd'), ('near_plane', 'near'),
                         ('far_plane', 'far'), ('t_coarse', 't_coarse'), ('sigma_coarse', 'sigma_coarse'),
                         ('rgb_coarse', 'rgb_coarse'), ('t_fine', 't_fine'), ('sigma_fine', 'sigma_fine'),
                         ('t_sorted', 't_sorted'), ('weights', 'weights'), ('perm', 'perm')):
            if k_h in r and k_o in o:
                line('fused ' + k_h, err(r[k_h], o[k_o]))
        r2 = hip_render(meta, t, dev, taps=(), skip_missed_rays=True)
        line('fused(skip) rgb', err(r2['rgb'], o['rgb'])); line('fused(skip) mask', err(r2['mask'], o['mask']))
        line('fused(skip) vs fused rgb', err(r2['rgb'], r['rgb']))
    torch.cuda.synchronize()
    print('done')


if __name__ == '__main__':
    main()
