"""Shared helpers for the HIP-vs-oracle parity tests (GPU box) and diagnostics."""
import torch

from oracle import nfi_oracle as orc
from nerf_from_image_amd import ops


def oracle_render(meta, t, device='cpu'):
    """Runs the oracle on a golden case on `device` (CPU = the reference's CPU numerics)."""
    g = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in t.items()}
    with torch.no_grad():
        return orc.render(
            g['planes'], g['w1'], g['b1'], g['w2'], g['b2'], g['cam2world'], g.get('focal'),
            meta['H'], meta['W'], meta['S'], meta['scene_range'], white_background=meta['white'],
            fine_sampling=meta['fine'], bbox=g.get('bbox'), center=g.get('center'),
            want_coords=bool(meta.get('coords')), noise_coarse=g.get('noise_coarse'),
            noise_fine=g.get('noise_fine'), use_sdf=meta['sdf'], beta=g.get('beta'), alpha=g.get('alpha'),
            attention_values=g.get('attention_values'), want_semantics=meta['A'] > 0, viewdir=viewdir_of(g))


def viewdir_of(t):
    """dict(x, w3, b3) of a --use_viewdir golden case, else None."""
    return dict(x=t['viewdir_x'], w3=t['w3'], b3=t['b3']) if 'viewdir_x' in t else None


def hip_field_setup(meta, t, dev, texel_dtype=ops.TEXEL_F32):
    texels = ops.planes_to_texels(t['planes'].to(dev), texel_dtype)
    if 'viewdir_x' in t:
        return texels, ops.decoder_pack_viewdir(*(t[k].to(dev) for k in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3')), meta['A'],
                                                texel_dtype)
    image = ops.decoder_pack(t['w1'].to(dev), t['b1'].to(dev), t['w2'].to(dev), t['b2'].to(dev), meta['A'], texel_dtype)
    return texels, image


def hip_render(meta, t, dev, taps=(), skip_missed_rays=False, texel_dtype=ops.TEXEL_F32, **kw):
    texels, image = hip_field_setup(meta, t, dev, texel_dtype)
    g = lambda k: t[k].to(dev) if k in t else None
    return ops.render_fwd(
        g('cam2world'), g('focal'), meta['H'], meta['W'], meta['S'], texels, image, meta['scene_range'], meta['A'],
        attention_values=g('attention_values'), use_sdf=meta['sdf'], beta=g('beta'), alpha=g('alpha'),
        bbox=g('bbox'), center=g('center'), noise_coarse=g('noise_coarse'), noise_fine=g('noise_fine'), fine_sampling=meta['fine'],
        white_background=meta['white'], taps=taps, skip_missed_rays=skip_missed_rays,
        ray_features=ops.pad_ray_features(t['viewdir_x'].to(dev)) if 'viewdir_x' in t else None, **kw)


def err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    d = (a - b).abs()
    fin = torch.isfinite(d)
    return dict(max=float(d[fin].max()) if fin.any() else 0.0, mean=float(d[fin].mean()) if fin.any() else 0.0,
                exact=float((a == b).float().mean()), nonfinite=int((~fin).sum()))
