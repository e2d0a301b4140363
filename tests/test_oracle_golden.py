"""The oracle must reproduce the committed reference vectors bit for bit on CPU.

tests/golden/*.npz were written by oracle/make_golden.py from the LIVE
reference (run.py::render + lib/nerf_utils.py + models/generator.py)."""
import pytest
import torch

from conftest import golden_case_names, load_golden
from oracle import nfi_oracle as orc


def run_oracle(meta, t):
    return orc.render(
        t['planes'], t['w1'], t['b1'], t['w2'], t['b2'], t['cam2world'], t.get('focal'),
        meta['H'], meta['W'], meta['S'], meta['scene_range'], white_background=meta['white'],
        fine_sampling=meta['fine'], bbox=t.get('bbox'), center=t.get('center'),
        want_coords=bool(meta.get('coords')), noise_coarse=t.get('noise_coarse'),
        noise_fine=t.get('noise_fine'), use_sdf=meta['sdf'], beta=t.get('beta'), alpha=t.get('alpha'),
        attention_values=t.get('attention_values'), want_semantics=meta['A'] > 0,
        viewdir=dict(x=t['viewdir_x'], w3=t['w3'], b3=t['b3']) if 'viewdir_x' in t else None)


@pytest.mark.parametrize('name', golden_case_names())
def test_oracle_matches_reference_vectors(name):
    meta, t = load_golden(name)
    with torch.no_grad():
        o = run_oracle(meta, t)
    keys = ['rgb', 'depth', 'mask', 'ro', 'rd', 'near', 'far', 'hit', 't_coarse', 'sigma_coarse',
            'rgb_coarse', 'outside_coarse', 'sdf_coarse', 'weights']
    if meta['fine']:
        keys += ['weights_coarse', 'weights_smooth', 'cdf', 'inds', 't_fine', 'sigma_fine', 'rgb_fine',
                 'perm', 't_sorted']
    if meta.get('coords'):
        keys += ['coords_map']
        o['coords_map'] = o['semantics']
    elif meta['A'] > 0:
        keys += ['semantics']
    for k in keys:
        ref = t['ref_' + k]
        got = o[k]
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        if got.dtype.is_floating_point:
            # same torch build => bit-exact; other builds: ATen kernels may differ in the last ulp
            assert torch.allclose(got, ref, rtol=0, atol=2e-6), (k, (got - ref).abs().max().item())
        else:
            mism = (got != ref).float().mean().item()
            assert mism <= 1e-3, (k, mism)


def test_no_hit_raises_like_reference():
    ro = torch.tensor([[[[5., 5., 5.]]]])
    rd = torch.tensor([[[[0., 0., 1.]]]])      # line misses the cube
    with pytest.raises(RuntimeError):
        orc.near_far(ro, rd, 0.5)
