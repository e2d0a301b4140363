"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports
every symbol include/nfi_hip.h declares, and rejects bad arguments without touching a GPU."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from nerf_from_image_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    return _lib.load()


def test_header_declares_the_expected_entry_points():
    src = open(_lib.HEADER).read()
    declared = set(re.findall(r'\b(nfi_[a-z_0-9]+)\s*\(', src))
    assert declared == set(_lib.FUNCTIONS), declared ^ set(_lib.FUNCTIONS)
    for name in ('nfi_render_fwd', 'nfi_field_query_fwd', 'nfi_raygen', 'nfi_near_far', 'nfi_sample_pdf',
                 'nfi_composite_fwd', 'nfi_planes_to_texels', 'nfi_decoder_pack', 'nfi_decoder_pack_viewdir',
                 'nfi_field_query_bwd', 'nfi_field_bwd_workspace_bytes', 'nfi_composite_bwd', 'nfi_points_bwd',
                 'nfi_raygen_bwd', 'nfi_bbox_overlay', 'nfi_resample', 'nfi_ray_weights', 'nfi_sdf_gradient_fwd',
                 'nfi_sdf_gradient_bwd', 'nfi_render_setup'):
        assert name in declared


def test_library_exports_every_declared_symbol(lib):
    for name in _lib.FUNCTIONS:
        assert hasattr(lib, name), name
    assert lib.nfi_version() >= 100
    assert lib.nfi_decoder_image_floats() == 6224
    assert lib.nfi_decoder_image_floats_viewdir() == 6016
    # workspace of the binned scatter: feature gradients (128 B/point) + order + ranks + two cell tables
    n_ws = _lib.struct_query('nfi_field_bwd_workspace_bytes', 'nfi_field_bwd_args', n_scenes=2, points_per_scene=1000,
                             plane_res=16, scatter_mode=1)
    assert n_ws >= 2 * 1000 * (128 + 24) + 2 * 3 * 2 * 256 * 4
    # 8 floats + 1 byte per ray + reduce words
    assert lib.nfi_render_workspace_bytes(16384) >= 16384 * 33


def test_library_contains_gfx950_code_objects_only(lib):
    blob = open(_lib.LIBRARY, 'rb').read()
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_integration_md_stub_matches_the_header():
    """The ctypes stub INTEGRATION.md shows a maintainer (section 2) lists nfi_field_args field for field."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = text[text.index('class nfi_field_args'):text.index('def field_query')]
    names = re.findall(r"\('(\w+)', ctypes\.", block)
    assert names == [n for n, _ in _lib.STRUCT_FIELDS['nfi_field_args']], names


def test_struct_layout_matches_header():
    # spot checks: field order and the natural-alignment size ctypes derives from the parsed header
    f = [n for n, _ in _lib.STRUCT_FIELDS['nfi_render_args']]
    assert f[:4] == ['n_scenes', 'height', 'width', 'n_samples'] and f[-13:] == ['profile_cycles', 'ray_features', 'termination_eps', 'texel_layout', 'clock_probe', 'row_offset', 'full_height',
                                                                                          'stash_t', 'stash_sigma', 'stash_rgb', 'rays_ready', 'coords', 'normals']
    assert ctypes.sizeof(_lib.STRUCTS['nfi_sample_pdf_args']) == 8 + 4 + 4 + 3 * 8 + 8 + 3 * 8
    for name, st in _lib.STRUCTS.items():
        assert ctypes.sizeof(st) % 8 == 0 or ctypes.sizeof(st) % 4 == 0, name


def test_bad_arguments_are_rejected_before_any_launch(lib):
    rc = lib.nfi_decoder_pack(None, None, None, None, 10, 0, None, None)
    assert rc == -1 and b'null' in lib.nfi_last_error()
    rc = lib.nfi_planes_to_texels(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 4096, 0, None)
    assert rc == -1 and b'plane_res' in lib.nfi_last_error()
    for S, fine in ((200, 1), (513, 0), (3, 0)):        # at most 128 per pass with fine sampling, 512 in a single pass
        a = _lib.make_args('nfi_render_args', n_scenes=1, height=4, width=4, n_samples=S, fine_sampling=fine, cam2world=16,
                           rgb=16, depth=16, mask=16, workspace=16)
        rc = lib.nfi_render_fwd(ctypes.byref(a), None)
        assert rc == -1 and b'n_samples' in lib.nfi_last_error()
    a = _lib.make_args('nfi_composite_args', n_rays=4, n_a=100, n_b=100, ray_directions=16, depth_a=16, sigma_a=16,
                       rgb_a=16, rgb_map=16, depth_map=16, mask=16)
    assert lib.nfi_composite_fwd(ctypes.byref(a), None) == -1


def test_product_path_has_no_cpu_fallback():
    from nerf_from_image_amd import ops
    with pytest.raises(RuntimeError, match='GPU'):
        ops.planes_to_texels(torch.zeros(1, 3, 32, 4, 4))
    # nothing under the package may import the oracle
    pkg = os.path.join(ROOT, 'nerf_from_image_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            assert 'oracle' not in open(os.path.join(pkg, fn)).read().replace('no oracle', ''), fn


def test_bench_roofline_is_reproducible_from_the_committed_profile():
    """bench.py's roofline object (no GPU needed for the arithmetic): `frac` is SURVEY.md 8(d)'s own figure - algorithmic
    decoder FLOPs per launch / kernel time against the fp32 matrix / vector peak - from the live kernel time and ray count
    alone; the vector-ALU pipe fraction (the pipe that binds) and the byte-side levels come from the committed rocprofv3
    PMC profile it names, each against its own peak; the ANY-issue proxy is a side field, never the fraction."""
    import json
    import bench
    src, prof = bench.load_pmc_profile()
    assert src in bench.PMC_PROFILES and prof == json.load(open(os.path.join(ROOT, src)))
    kernel_ms, marched = prof['kernel_ns_in_clock_pass'] * 1e-6, prof['rays_marched_per_launch']
    r = bench.roofline(kernel_ms, marched, 8)
    assert r['source'] == src and r['bound'] == 'valu' and r['valu_frac'] == r['valu_pipe']['frac'] and r['unit'] == 'TFLOP/s' and r['peak'] == 157.3
    assert abs(r['achieved'] - 704512 * marched / (kernel_ms * 1e-3) / 1e12) < 1e-9
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['frac'] == r['frac_flops'] and 0.2 < r['frac'] < 1.0
    assert r['traffic'] == prof['fabric_bytes_per_launch']
    # the vector ALU alone (the pipe that binds) at the profile's own kernel time is the profile's VALU fraction
    assert abs(r['valu_pipe']['frac'] - prof['valu_frac']) < 0.01 * prof['valu_frac'] and 0.4 < r['valu_pipe']['frac'] < 1.0
    assert abs(r['any_issue_proxy']['frac'] - prof['issue_frac']) < 0.01 * prof['issue_frac']
    assert r['valu_pipe']['frac'] < r['any_issue_proxy']['frac']
    assert 1.5 < r['waves_per_simd'] <= 2.0
    # priced at a shader clock measured in the timed launches, the pipe fraction scales with the clock ratio; the FLOP
    # fraction does not depend on the clock
    live = bench.roofline(kernel_ms, marched, 8, live_clock_hz=2.4e9)
    assert abs(live['valu_pipe']['frac'] - r['valu_pipe']['frac'] * prof['shader_clock_hz'] / 2.4e9) < 1e-9
    assert 'live' in live['shader_clock_source'] and live['frac'] == r['frac']
    for level in ('hbm_compulsory', 'l2_requests', 'fabric'):
        assert 0.0 < r['levels'][level]['frac'] < 1.0, level
    assert r['levels']['gather_stream_algorithmic']['x_hbm_peak'] > 1.0      # cache-served: why HBM is not the bound
