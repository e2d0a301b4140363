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


def oracle_normal_map(meta, t, o):
    """The composited normal map of a golden case as the reference computes it (generator.py:599-623 + lib/nerf_utils.py:
    149-151, 159): autograd of the oracle's distance at every sample, normalised, composited with the oracle's weights."""
    # (--use_viewdir cases: the distance is row 0 of the 33-row second layer; the colour rows do not matter here, so the
    #  oracle is asked with the first four rows as a plain A = 0 decoder)
    vd = 'viewdir_x' in t
    w2, b2, att = (t['w2'][:4], t['b2'][:4], None) if vd else (t['w2'], t['b2'], t.get('attention_values'))

    def oracle_normals(pts):
        p = pts.clone().requires_grad_()
        q = orc.field_query(t['planes'], t['w1'], t['b1'], w2, b2, p, meta['scene_range'], True, t['beta'], t['alpha'], att)
        gx, = torch.autograd.grad(q['sdf'].sum(), p)
        return torch.nn.functional.normalize(gx, dim=-1)
    B, H, W = meta['B'], meta['H'], meta['W']
    n_all = oracle_normals(orc.points_on_rays(o['ro'], o['rd'], o['t_coarse']).reshape(B, -1, 3)).view(B, H, W, -1, 3)
    if meta['fine']:
        n_f = oracle_normals(orc.points_on_rays(o['ro'], o['rd'], o['t_fine']).reshape(B, -1, 3)).view(B, H, W, -1, 3)
        n_all = torch.cat((n_all, n_f), dim=-2).gather(-2, o['perm'].unsqueeze(-1).expand(-1, -1, -1, -1, 3))
    ref_map = (o['weights'][..., None] * n_all).sum(dim=-2)
    if meta['white']:
        ref_map = ref_map + (1. - o['mask'][..., None])
    return ref_map
