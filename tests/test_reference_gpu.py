"""The drop-in on the REAL reference classes, on MI355X: models/generator.py::Generator (StyleGAN2 plane producer, texture
mapper, ViewDirectionMapper - all PyTorch-ROCm) with `attach()`ed HIP sampler, rendered by nerf_from_image_amd.render,
against the same Generator rendered by the reference's own run.py::render (AST-sliced) on PyTorch-ROCm and on the CPU -
same weights, cameras and noise.  The reference sources are oracle/_ref (staged by oracle/make_ref.py; the GPU box has
no /root/reference) or the checkout itself.

Budget (BASELINE north star): rgb / depth / mask within 1e-4 of the reference.  Against the reference's CPU numerics (the
pinned side) that is asserted as is.  The reference's GPU path differs from its own CPU path by more than that on a
handful of pixels (ATen's GPU elementwise kernels contract a*b+c into FMAs, an ulp on a query point moves a sample across
a cube face or a texel boundary); there the HIP result has to be as close to the GPU reference as the CPU reference is
(+1e-4), and the number of pixels over 1e-4 is bounded."""
import pytest
import torch

from oracle import reference

import reference_cases as rc

pytestmark = pytest.mark.gpu
BUDGET = 1e-4


def _require_reference():
    # The staged copy ships with the snapshot (git-ignored like the .so, not gpurun-ignored): on the GPU box these tests RUN
    # (0 skipped in profiles/r6/pytest_gpu_run1.log).  A snapshot made without it - a bare clone, where build() found no
    # /root/reference to stage from - is reported as a skip with the recipe's name rather than as a dozen failures.
    if not reference.available():
        pytest.skip('reference sources not staged: run oracle/make_ref.py (or __graft_entry__.build()) where /root/reference exists')


def _check(rep, maps=('rgb', 'depth', 'mask')):
    for k in maps:
        assert rep['vs_reference_cpu'][k] <= BUDGET, (k, 'vs the reference on the CPU', rep)
        assert rep['vs_reference_gpu'][k] <= rep['reference_cpu_vs_gpu_gap'][k] + BUDGET, (k, 'vs the reference on this GPU', rep)
    assert rep['mask_mean'] > 0.1, rep          # the scene renders surfaces


@pytest.mark.parametrize('geometry,batch', [('chairs', 1), ('chairs', 8), ('p3d', 16), ('cub', 4), ('density', 4)])
def test_render_matches_the_real_reference(gpu_device, geometry, batch):
    """cfg2 (B = 1 and 8), a p3d_car-like cfg3 batch (scene_range 1.4, black background, crop bbox, B = 16), an
    orthographic cub-like cfg4 batch, and the Generator's other branches (`use_sdf=False`: sigma = softplus(d - 1),
    `attention_values=0`: rgb = wide_sigmoid_rescaled(features); models/generator.py:637-641, 665-666) on the chairs
    cameras - all 128 x 128 rays, 64 + 64 samples."""
    _require_reference()
    sc = rc.build_scene(geometry, batch, gpu_device)
    rep = rc.compare(sc, 128, 64, cpu_images=2)
    _check(rep)
    # the handful of pixels the GPU reference itself moves (see the module docstring)
    n_pix = batch * 128 * 128
    assert rep['pixels_over_1e-4_vs_reference_gpu']['rgb'] <= max(2, 2e-5 * n_pix), rep


# 16-bit texel STORAGE against the fp32 reference (run.py:176-350 on fp32 planes): what the storage type costs, measured on
# MI355X (profiles/r6/reference_parity.json, `configs_vs_fp32_reference`) and asserted at twice the measured figure.
# name -> (max |d rgb|, |d depth|, |d mask|), (mean |d rgb|, |d depth|, |d mask|) as MEASURED
STORAGE_DEVIATION = {
    # BASELINE cfg2 as worded ("bf16"): 8 significand bits per texel - 21 x the fp32 budget at the worst pixel
    'cfg2_b8_128px_64+64_bf16_texels': ((2.09e-3, 4.99e-3, 1.91e-3), (4.3e-5, 8.0e-5, 3.9e-5)),
    # the same configuration on fp16 texels (11 bits, the fast storage: three workgroups per CU)
    'cfg2_b8_128px_64+64_fp16_texels': ((2.70e-4, 5.46e-4, 2.39e-4), (6.6e-6, 1.29e-5, 6.3e-6)),
    # BASELINE cfg5 ("fp16 render ... 256^2, 128 fine samples")
    'cfg5_b2_256px_128+128_fp16_texels': ((1.73e-4, 3.62e-4, 1.48e-4), (5.5e-6, 1.11e-5, 5.2e-6)),
}


def _check_storage(rep, measured):
    (mx, mn) = measured
    for which in ('vs_reference_gpu', 'vs_reference_cpu'):
        for k, m in zip(('rgb', 'depth', 'mask'), mx):
            assert rep[which][k] <= 2.0 * m, (which, k, rep[which][k], 'measured', m)
    for which in ('mean_abs_vs_reference_gpu', 'mean_abs_vs_reference_cpu'):
        for k, m in zip(('rgb', 'depth', 'mask'), mn):
            assert rep[which][k] <= 2.0 * m, (which, k, rep[which][k], 'measured', m)
    assert rep['mask_mean'] > 0.1, rep


def test_sphere_pretrained_generator_matches_the_real_reference(gpu_device):
    """SURVEY 8(d)'s "G-sphere": a default-initialised generator renders an almost empty scene, so the reference's own
    `pretrain_sdf` (run.py:824-866) is run first - 80 Adam steps against the unit sphere here, 300 in tools/g_sphere.py - and
    the drop-in is compared with the untouched reference on the result: every ray ends on an opaque surface."""
    _require_reference()
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import g_sphere
    state = g_sphere.pretrained_state(gpu_device, 80, 4)
    sc = g_sphere.scene_from_state(state, gpu_device, 4)
    rep = rc.compare(sc, 128, 64, cpu_images=1)
    _check(rep)
    assert rep['mask_mean'] > 0.9 and max(rep['pixels_over_1e-4_vs_reference_gpu'].values()) <= 2, rep


def test_class_conditional_generator_takes_z_and_labels(gpu_device):
    """--use_class: `Generator(num_classes=...)` called with model_input = (z, labels) (models/generator.py:428-446: class
    embedding, conditional mapping network) - the attached forward hands the tuple to the reference's own latent handling;
    against the untouched Generator + reference render on this GPU, and with the latents ws given instead."""
    _require_reference()
    sc = rc.build_scene('classes', 4, gpu_device)
    assert sc.gen.num_classes == 5 and sc.labels is not None
    rep_ws = rc.compare(sc, 128, 64, cpu_images=1)
    _check(rep_ws)
    import copy
    tup = copy.copy(sc)
    tup.ws = (sc.z, sc.labels)                    # what run.py passes in the class-conditional training / eval loops
    rep = rc.compare(tup, 128, 64, cpu_images=0)
    for k in ('rgb', 'depth', 'mask'):
        assert rep['vs_reference_gpu'][k] <= rep_ws['reference_cpu_vs_gpu_gap'][k] + BUDGET, (k, rep)
        assert abs(rep['vs_reference_gpu'][k] - rep_ws['vs_reference_gpu'][k]) <= 1e-5, (k, rep, rep_ws)   # the same render


def _check_gpu_only(rep, n_pix):
    """Against the reference on this GPU alone (model inputs the CPU leg cannot be fed with): within the budget but for the
    handful of pixels the GPU reference moves against its own CPU path (module docstring)."""
    for k in ('rgb', 'depth', 'mask'):
        assert rep['pixels_over_1e-4_vs_reference_gpu'][k] <= max(2, 2e-5 * n_pix), (k, rep)
        assert rep['vs_reference_gpu'][k] <= 5e-3, (k, rep)
    assert rep['mask_mean'] > 0.1, rep


def test_encoder_generator_and_the_remaining_model_inputs(gpu_device):
    """--use_encoder: `Generator(use_encoder=True)` is called with model_input = (z, image) (models/generator.py:423-426:
    ResidualEncoder -> conditional mapping network), and run.py's ParallelModel reaches its `.emb` directly
    (`encoder_output=True`, run.py:594-595) and its regulariser outputs (`pretrain_sdf=True`, run.py:589-593).  Then the two
    model inputs not exercised elsewhere: `freeze_noise` on a generator WITH StyleGAN2 noise (noise_mode 'const',
    generator.py:471-474) and latents given as ws [B,1,512] (broadcast to all layers, generator.py:439-442)."""
    _require_reference()
    import copy
    import nerf_from_image_amd.render as nfi_render
    sc = rc.build_scene('encoder', 2, gpu_device)
    assert sc.gen.use_encoder and sc.image is not None
    tup = copy.copy(sc)
    tup.ws = (sc.z, sc.image)
    _check_gpu_only(rc.compare(tup, 128, 64, cpu_images=0), 2 * 128 * 128)
    ref_render, _ = reference.load_render(sc.args, sc.dcfg)
    pm_ref = rc.parallel_model(ref_render, sc.gen, 64, 32)
    pm_hip = rc.parallel_model(nfi_render.make_render(sc.args, sc.dcfg), sc.hip, 64, 32)
    with torch.no_grad():
        # (the reference's own module on both sides; MIOpen's convolutions are not bit-reproducible between two module copies)
        emb_hip, emb_ref = (pm(None, None, None, None, sc.image, encoder_output=True) for pm in (pm_hip, pm_ref))
        assert emb_hip.shape == emb_ref.shape == (2, 512) and rc.max_err(emb_hip, emb_ref) <= 1e-5
    losses = []
    for pm, model in ((pm_ref, sc.gen), (pm_hip, sc.hip)):
        model.train()
        torch.manual_seed(41)
        losses.append(pm(None, None, None, None, (sc.z, sc.image), pretrain_sdf=True))
        model.eval()
    assert sorted(losses[0]) == sorted(losses[1]) == ['sdf_distance_loss', 'sdf_eikonal_loss']
    for k in losses[0]:
        assert losses[1][k].shape == losses[0][k].shape == (2,)
        assert rc.max_err(losses[1][k], losses[0][k]) <= 1e-4 * float(losses[0][k].abs().max()), (k, losses)

    noisy = rc.build_scene('chairs', 2, gpu_device, stylegan_noise=True)
    frozen = rc.compare(noisy, 128, 64, cpu_images=1, extra_model_inputs={'freeze_noise': True})
    _check(frozen)
    noise = rc.draw_noise(noisy, 128, 64)
    first, second = (rc.hip_render(noisy, 128, 64, noise, extra_model_inputs={'freeze_noise': True}) for _ in range(2))
    drawn = rc.hip_render(noisy, 128, 64, noise)
    assert rc.max_err(first[0], second[0]) <= BUDGET < 1e-2 < rc.max_err(first[0], drawn[0])    # the input is what freezes it

    one_w = copy.copy(noisy)
    one_w.ws = noisy.ws[:, :1].contiguous()                                      # [B,1,512]
    _check_gpu_only(rc.compare(one_w, 128, 64, cpu_images=0, extra_model_inputs={'freeze_noise': True}), 2 * 128 * 128)


def test_more_call_patterns_of_render_match_the_real_reference(gpu_device):
    """Call patterns of run.py::render / Generator.forward beyond the BASELINE configurations, each against the untouched
    reference on this GPU (and its CPU path): the orthographic camera WITH a crop box (cub's loader passes one), deterministic
    sampling (`randomize=False`: no stratified jitter, linspace in sample_pdf), `extra_model_outputs=['attention_values']`
    (slot 5 of the tuple), the 'bbox' visualisation overlay (generator.py:645-659, needs compute_coords), the normal and
    semantic maps of the first eval batch together (run.py:2036-2051)."""
    _require_reference()
    sc = rc.build_scene('cub_bbox', 4, gpu_device)
    _check(rc.compare(sc, 128, 64, cpu_images=1))
    _check(rc.compare(sc, 128, 64, cpu_images=1, randomize=False))
    sc = rc.build_scene('p3d', 2, gpu_device)
    noise = rc.draw_noise(sc, 64, 32)
    ours = rc.hip_render(sc, 64, 32, noise, extra_model_outputs=['attention_values'])
    ref = rc.reference_render(sc, 64, 32, noise, extra_model_outputs=['attention_values'])
    assert set(ours[5]) == set(ref[5]) == {'attention_values'} and torch.equal(ours[5]['attention_values'], ref[5]['attention_values'])
    rep = rc.compare(sc, 64, 32, cpu_images=1, compute_coords=True, extra_model_outputs=['bbox'])
    _check(rep, ('rgb', 'depth', 'mask', 'extra'))
    sc = rc.build_scene('chairs', 2, gpu_device)
    rep = rc.compare(sc, 64, 32, cpu_images=1, grad=True, compute_normals=True, compute_semantics=True)
    _check(rep, ('rgb', 'depth', 'mask', 'extra'))
    assert rep['vs_reference_cpu']['normals'] <= 3e-3, rep
    # error behaviour: a batch in which NO ray meets the scene cube - the reference fails on min() of an empty selection
    # (lib/nerf_utils.py:258), the drop-in raises the same exception type
    import copy
    lost = copy.copy(sc)
    lost.cam = sc.cam.clone()
    lost.cam[:, :3, 3] += 10.0 * sc.cam[:, :3, 0]          # ten units sideways, same viewing direction
    noise = rc.draw_noise(sc, 64, 32)
    with pytest.raises(RuntimeError):
        rc.reference_render(lost, 64, 32, noise)
    with pytest.raises(RuntimeError, match='no ray intersects the scene cube'):
        rc.hip_render(lost, 64, 32, noise)


def test_stylegan_noise_draws_interleave_like_the_reference(gpu_device):
    """A generator WITH per-layer StyleGAN2 noise (`disable_stylegan_noise=False`, the training default): after the same
    torch.manual_seed the drop-in must consume PyTorch's Philox stream in the reference's order - stratified jitter
    (nerf_utils.py:115) BEFORE the model is called, the synthesis network's randn draws inside it, the inverse-CDF draw
    (nerf_utils.py:202) after it - or every layer's noise, and with it every plane, differs."""
    _require_reference()
    sc = rc.build_scene('chairs', 2, gpu_device, stylegan_noise=True)
    for mod in (sc.gen, sc.hip):
        mod.train()                                  # (noise_mode 'random' draws; eval would draw too: use_noise is the switch)
    torch.manual_seed(777)
    ours = rc.hip_render(sc, 128, 64, None)
    torch.manual_seed(777)
    ref = rc.reference_render(sc, 128, 64, None)
    torch.manual_seed(778)
    other = rc.reference_render(sc, 128, 64, None)
    assert rc.max_err(ref[0], other[0]) > 1e-2           # the noise matters: another seed, another image
    with torch.no_grad():                                # ... and so do the per-layer draws alone (noise_strength 0.1, reference_cases)
        planes = []
        for seed in (1, 2):
            torch.manual_seed(seed)
            planes.append(sc.gen.synthesis_network(sc.ws[:, :14]))
    assert rc.max_err(planes[0], planes[1]) > 1e-2
    for k, a, b in zip(('rgb', 'depth', 'mask'), ours[:3], ref[:3]):
        over = int(((a - b).abs() > BUDGET).sum())
        # 3 values of 98 304 over the budget (2.0e-4 at most) measured: the pixels the GPU reference moves against its own CPU path
        assert over <= 8 and rc.max_err(a, b) < 5e-3, (k, rc.max_err(a, b), over)
    # the same with the tail of the last synthesis block fused into the texel hand-off (attach(fused_handoff=True)): the block's
    # two noise draws come before the fused part
    import copy
    import nerf_from_image_amd.generator as nfi_gen
    fused = copy.copy(sc)
    fused.hip = nfi_gen.attach(copy.deepcopy(sc.gen), fused_handoff=True).train()
    torch.manual_seed(777)
    theirs = rc.hip_render(fused, 128, 64, None)
    for k, a, b in zip(('rgb', 'depth', 'mask'), theirs[:3], ref[:3]):
        over = int(((a - b).abs() > BUDGET).sum())
        assert over <= 8 and rc.max_err(a, b) < 5e-3, (k, 'fused hand-off', rc.max_err(a, b), over)


def test_cfg1_shape_coarse_only_matches_the_real_reference(gpu_device):
    """BASELINE cfg1's shape - 4 scenes, 64 x 64 rays, 32 coarse samples, no fine pass (`--fine_sampling` off: ONE stratified
    draw, no resampling, run.py:261 skipped) - on the real class: the single-pass fused kernel against the reference."""
    _require_reference()
    sc = rc.build_scene('chairs', 4, gpu_device, fine_sampling=False)
    rep = rc.compare(sc, 64, 32, cpu_images=2)
    _check(rep)
    assert max(rep['pixels_over_1e-4_vs_reference_gpu'].values()) <= 2, rep


@pytest.mark.parametrize('case', ['cfg5_b2_256px_128+128_fp32_texels', 'cfg2_b8_128px_64+64_fp32_texels_term1e-5',
                                  'cfg5_b2_256px_128+128_fp32_texels_term1e-5'])
def test_cfg5_shape_and_termination_match_the_real_reference(gpu_device, case):
    """BASELINE cfg5's shape (256 x 256 rays, 128 + 128 samples: run.py's res_multiplier = ray_multiplier = 2, 598-605) on
    fp32 texels, and the fine-pass termination + compaction of cfg5 (termination_eps = 1e-5) at both shapes: inside the
    1e-4 budget of the exact fp32 reference."""
    _require_reference()
    rep = rc.config_case(case, gpu_device, cpu_images=1)
    _check(rep)
    n_pix = rc.CONFIG_CASES[case][1] * rc.CONFIG_CASES[case][2] ** 2
    assert max(rep['pixels_over_1e-4_vs_reference_gpu'].values()) <= max(2, 2e-5 * n_pix), rep


@pytest.mark.parametrize('case', sorted(STORAGE_DEVIATION))
def test_16_bit_texel_storage_deviation_from_the_fp32_reference(gpu_device, case):
    """cfg2 as BASELINE words it (bf16) and cfg5 (fp16): the HIP twin stores the planes in 16 bits, the reference renders the
    fp32 planes.  The deviation is the storage type's, not the kernels' (against the oracle on the SAME rounded planes the
    16-bit kernels are inside 5e-6: tests/test_hip_parity.py) - stated here, asserted at 2 x the measured value, with and
    without fine-pass termination."""
    _require_reference()
    scenes = {}
    _check_storage(rc.config_case(case, gpu_device, cpu_images=1, scenes=scenes), STORAGE_DEVIATION[case])
    if case + '_term1e-5' in rc.CONFIG_CASES:
        _check_storage(rc.config_case(case + '_term1e-5', gpu_device, cpu_images=1, scenes=scenes), STORAGE_DEVIATION[case])


def test_cfg5_through_run_py_parallel_model(gpu_device):
    """cfg5 the way run.py reaches it: ParallelModel(128, ...) called with res_multiplier = 2, ray_multiplier = 2
    (run.py:598-605) - the drop-in on fp32 and on fp16 texels against the untouched ParallelModel + reference render."""
    _require_reference()
    import nerf_from_image_amd.render as nfi_render
    from nerf_from_image_amd import ops
    sc = rc.build_scene('chairs', 2, gpu_device)
    ref_render, _ = reference.load_render(sc.args, sc.dcfg, unscripted_stages=True)
    pm_ref = rc.parallel_model(ref_render, sc.gen, 128, 64)
    noise = rc.draw_noise(sc, 256, 128)
    kw = dict(use_ema=True, res_multiplier=2, ray_multiplier=2)
    with torch.no_grad():
        with rc.ReplayNoise(noise):
            b = pm_ref(sc.cam, sc.focal, None, sc.bbox, sc.ws, **kw)
        for texels, bound in ((ops.TEXEL_F32, (2e-4, 3e-4, 2e-4)),                # measured 7.4e-5 / 1.3e-4 / 6.4e-5 (GPU vs GPU)
                              (ops.TEXEL_F16, tuple(2 * m for m in STORAGE_DEVIATION['cfg5_b2_256px_128+128_fp16_texels'][0]))):
            twin = sc if texels == ops.TEXEL_F32 else rc.with_texels(sc, texels)
            pm_hip = rc.parallel_model(nfi_render.make_render(sc.args, sc.dcfg), twin.hip, 128, 64)
            with rc.ReplayNoise(noise):
                a = pm_hip(sc.cam, sc.focal, None, sc.bbox, sc.ws, **kw)
            assert a[0].shape == b[0].shape == (2, 256, 256, 3)
            for k, x, y, lim in zip(('rgb', 'depth', 'mask'), a[:3], b[:3], bound):
                assert rc.max_err(x, y) <= lim, (k, texels, rc.max_err(x, y), lim)


def test_extra_maps_match_the_real_reference(gpu_device):
    """compute_semantics (every inversion eval batch, run.py:2036-2051) and compute_coords (every encoder-training
    iteration, run.py:1639-1646): the composited map in slot 4 of the tuple."""
    _require_reference()
    sc = rc.build_scene('p3d', 4, gpu_device)
    rep = rc.compare(sc, 128, 64, cpu_images=1, compute_semantics=True)
    _check(rep, ('rgb', 'depth', 'mask', 'extra'))
    rep = rc.compare(sc, 128, 64, cpu_images=1, compute_coords=True)
    _check(rep, ('rgb', 'depth', 'mask', 'extra'))
    # `extra_model_inputs` of Generator.forward (generator.py:419-421, 452-464): the caller's own colour table, and a bias
    # on the texture mapper's - the attached forward leaves both to the reference's code and takes the table it returns
    g = torch.Generator().manual_seed(3)
    table = (torch.rand(4, 10, 3, generator=g) * 2 - 1).to(gpu_device)
    for inputs in ({'attention_values': table}, {'attention_values_bias': 0.3 * table}):
        rep = rc.compare(sc, 128, 64, cpu_images=1, compute_semantics=True, extra_model_inputs=inputs)
        _check(rep, ('rgb', 'depth', 'mask', 'extra'))


def test_normal_map_matches_the_real_reference(gpu_device):
    """compute_normals (run.py:1444-1454): the reference differentiates the SDF by autograd (generator.py:599-623), the
    kernels analytically; maps agree to 3e-3 (the normalisation amplifies the decoder's 1e-5 where |grad| is small)."""
    _require_reference()
    sc = rc.build_scene('chairs', 2, gpu_device)
    rep = rc.compare(sc, 64, 32, cpu_images=2, grad=True, compute_normals=True)       # (the reference's sampler needs autograd)
    _check(rep)
    assert rep['vs_reference_cpu']['normals'] <= 3e-3, rep
    assert rep['vs_reference_gpu']['normals'] <= rep['reference_cpu_vs_gpu_gap']['normals'] + 3e-3, rep


def test_view_direction_decoder_matches_the_real_reference(gpu_device):
    """--use_viewdir (carla): the reference's ViewDirectionMapper runs in PyTorch, its closure (generator.py:243-251) in
    the kernels."""
    _require_reference()
    sc = rc.build_scene('carla', 2, gpu_device)
    rep = rc.compare(sc, 64, 32, cpu_images=1)
    _check(rep)
    # the eval callers on carla (run.py:1444-1454, 2036-2051): semantics / coords with the view-direction decoder, composited
    # by the fused kernel itself since round 5, the normal map since round 6 (ONE render launch)
    for kw in (dict(compute_semantics=True), dict(compute_coords=True)):
        rep = rc.compare(sc, 64, 32, cpu_images=1, **kw)
        _check(rep, ('rgb', 'depth', 'mask', 'extra'))
    rep = rc.compare(sc, 64, 32, cpu_images=1, grad=True, compute_normals=True)       # (the reference's sampler needs autograd)
    _check(rep)
    assert rep['vs_reference_cpu']['normals'] <= 3e-3, rep
    assert rep['vs_reference_gpu']['normals'] <= rep['reference_cpu_vs_gpu_gap']['normals'] + 3e-3, rep


# Gradients with the producer's convolutions on MIOpen's deterministic solvers (rc.deterministic_producer): inside one
# process every figure repeats to all printed digits (profiles/r6/gradient_spread.json, three runs; without the flag d loss /
# d latents moves by 1e-5 ... 2e-3 between runs in BOTH implementations - the producer's atomic split-K weight gradients).
# From one process to the next MIOpen may still pick another (deterministic) solver, the planes then differ in the last bits
# and a few samples flip sides of a texel edge: the camera / focal figures - sums over the image that cancel ~1000 : 1 - move
# by a factor 2-4; and the latents' gradient - which runs through the producer's own backward in both implementations - was
# seen at 1.9e-6 ... 2.2e-5 (MIOpen may serve the two backwards of one process from different solvers once its find
# database has learnt the shapes).  Relative L2 of the HIP gradient against the fp32 reference's: the MAXIMUM over the
# round's sessions (profiles/r6/); asserted at 3 x.
LATENTS_MEASURED = 2.2e-5            # any d / d latents figure, any geometry: the largest seen (regulariser branch, session r6r)
GRADIENT_MEASURED = {
    #          d/d latents  d/d planes  d/d camera  d/d focal
    'chairs': dict(g_ws=7.8e-6, g_planes=4.9e-5, g_cam=7.7e-5, g_focal=3.5e-4),
    'p3d': dict(g_ws=7.7e-6, g_planes=5.4e-5, g_cam=5.3e-4, g_focal=4.0e-4),
    # (orthographic: the reference's own fp32 sum of the camera gradient is 6e-4 off float64, ours 4e-5 - below)
    'cub': dict(g_ws=5.4e-6, g_planes=5.5e-5, g_cam=6.3e-4),
    'carla': dict(g_ws=3.8e-6, g_planes=3.5e-5, g_cam=3.6e-5, g_focal=1.6e-5),
    # density branch + direct colour head (one session: 3.7e-6 / 6.1e-5 / 2.7e-5 / 7.6e-5; the cameras are 'chairs', whose
    # camera / focal spread over the sessions is taken over)
    'density': dict(g_ws=3.8e-6, g_planes=6.1e-5, g_cam=7.7e-5, g_focal=3.5e-4),
}
# HIP's distance from the float64 reference over the fp32 reference's distance from it, at most (renderer only, same planes):
# planes 1.0 - 2.0 measured (the backward's split-fp16 operands are scaled per tile: an entry is resolved to 2^-22 of its
# tile's largest), camera 0.07 - 1.0, focal 1.0 - 4.2 (22 significand bits against fp32's 24 in the coordinate gradients of
# every sample, summed over an image under a crop box with ~1000 : 1 cancellation; absolute: 2.8e-4 ... 6.6e-4 of the
# gradient's norm, the fp32 reference 7e-5 ... 3e-4)
FLOAT64_RATIO = dict(g_planes=2.5, g_cam=2.5, g_focal=6.0)
# ... + a floor below which a ratio of two such distances says nothing (seen: camera 4.9e-5 against 1.4e-5 in one session,
# 3.9e-5 against 6.0e-4 in another)
FLOAT64_FLOOR = dict(g_planes=1e-5, g_cam=1e-4, g_focal=1e-4)


@pytest.mark.parametrize('geometry', ['chairs', 'p3d', 'cub', 'carla', 'density'])
def test_gradients_match_the_real_reference(gpu_device, geometry):
    """Forward + backward through the real plane producer: d loss / d ws (through the StyleGAN2 synthesis network and
    the texture mapper), d loss / d planes (what the renderer hands the producer's backward), d loss / d camera matrix,
    d loss / d focal - the leaves the inversion loop optimises (run.py:2264-2299).

    Two comparators.  (1) The fp32 reference, at 3 x the measured figure.  (2) The reference in FLOAT64 on the same device
    (rc.as_double) as ground truth for the RENDERER: the float64 run renders the very planes the fp32 runs rendered, so the
    producer's own rounding is common to all three; the HIP gradient may be at most FLOAT64_RATIO times as far from it as
    the fp32 reference is."""
    _require_reference()
    with rc.deterministic_producer():
        sc = rc.build_scene(geometry, 2, gpu_device)
        # (carla: --use_viewdir - the camera gradient also flows through the PyTorch ViewDirectionMapper's view directions)
        rep = rc.gradients(sc, 128, 64) if geometry != 'carla' else rc.gradients(sc, 64, 32)
    assert abs(rep['loss_hip'] - rep['loss_reference']) <= 1e-5 * abs(rep['loss_reference']), rep
    for k, measured in GRADIENT_MEASURED[geometry].items():
        measured = max(measured, LATENTS_MEASURED) if k == 'g_ws' else measured
        assert rep[k] <= 3.0 * measured + 1e-6, (k, rep[k], 'measured', measured, rep)
    ours, theirs = rep['renderer_only_hip_vs_float64'], rep['renderer_only_reference_vs_float64']
    for k in ours:
        assert ours[k] <= FLOAT64_RATIO[k] * theirs[k] + FLOAT64_FLOOR[k], (k, 'vs float64: HIP', ours[k], 'fp32 reference', theirs[k], rep)
    # the whole graph in float64 (producer included): both fp32 implementations carry the producer's rounding
    for k in ('g_ws', 'g_planes'):
        assert rep['hip_vs_float64'][k] <= 2.5 * rep['reference_vs_float64'][k] + 1e-5, (k, rep)


@pytest.mark.parametrize('samples', [128, 256])
def test_single_pass_render_and_gradients_match_the_real_reference(gpu_device, samples):
    """`--fine_sampling` off: ONE pass of 128 samples (training) or 4 x 64 = 256 (`ray_multiplier=4` of the inversion loop,
    run.py:2271) - the single-list fused kernels (render_fwd_wide / render_fwd_long) and their one-node backward against the
    real reference: forward inside the budget, gradients (measured, one session: latents 1.3e-5, planes 2.0e-5 / 1.6e-5,
    camera 8e-6 / 1.2e-5, focal 3e-6 / 1.1e-5; asserted at 3 x resp. the 1e-4 below which the image-wide sums say nothing)."""
    _require_reference()
    with rc.deterministic_producer():
        sc = rc.build_scene('p3d', 2, gpu_device, fine_sampling=False)
        _check(rc.compare(sc, 64, samples, cpu_images=1))
        rep = rc.gradients(sc, 64, samples)
    assert abs(rep['loss_hip'] - rep['loss_reference']) <= 1e-5 * abs(rep['loss_reference']), rep
    assert rep['g_ws'] <= 3 * LATENTS_MEASURED and rep['g_planes'] <= 6.5e-5 and rep['g_cam'] <= 1e-4 and rep['g_focal'] <= 1e-4, rep
    ours, theirs = rep['renderer_only_hip_vs_float64'], rep['renderer_only_reference_vs_float64']
    for k in ours:
        assert ours[k] <= FLOAT64_RATIO[k] * theirs[k] + FLOAT64_FLOOR[k], (k, ours[k], theirs[k], rep)


def test_semantic_map_gradient_matches_the_real_reference(gpu_device):
    """`compute_semantics=True` WITH a gradient (the staged path: one launch per stage, every per-sample tensor an autograd
    tensor): a loss on rgb, mask and the composited semantic map, gradients w.r.t. latents / planes / camera / focal against
    the real reference (measured, one session: 5.9e-6 / 5.6e-5 / 8.2e-5 / 3.7e-5 - asserted against the p3d row of
    GRADIENT_MEASURED, whose spread over the sessions is known)."""
    _require_reference()
    with rc.deterministic_producer():
        sc = rc.build_scene('p3d', 2, gpu_device)
        rep = rc.gradients(sc, 128, 64, compute_semantics=True)
    assert abs(rep['loss_hip'] - rep['loss_reference']) <= 1e-5 * abs(rep['loss_reference']), rep
    for k, measured in GRADIENT_MEASURED['p3d'].items():
        measured = max(measured, LATENTS_MEASURED) if k == 'g_ws' else measured
        assert rep[k] <= 3.0 * measured + 1e-6, (k, rep[k], rep)


def test_force_no_cam_grad_matches_the_real_reference(gpu_device):
    """`force_no_cam_grad=True` (run.py:211-214; the eval renders and --no_optimize_pose inversion, run.py:1262, 2045,
    2274): the coarse query points, the depths and the ray directions are detached - but run.py:286-288 builds the FINE
    pass's points from the undetached ray origins, so a camera that requires grad still receives the gradient of the fine
    samples' origins (its translation column), and the focal length none.  The drop-in reproduces exactly that; the
    latents' gradient is the reference's."""
    _require_reference()
    import copy
    with rc.deterministic_producer():
        sc = rc.build_scene('p3d', 2, gpu_device)
        noise = rc.draw_noise(sc, 128, 64)
        gw = torch.Generator(device=gpu_device).manual_seed(5)
        w_rgb = torch.randn((2, 128, 128, 3), device=gpu_device, generator=gw)
        grads = {}
        for which in ('hip', 'ref'):
            ws = sc.ws.detach().clone().requires_grad_()
            cam, focal = sc.cam.detach().clone().requires_grad_(), sc.focal.detach().clone().requires_grad_()
            if which == 'hip':
                out = rc.hip_render(sc, 128, 64, noise, grad=True, ws=ws, cam=cam, focal=focal, force_no_cam_grad=True)
            else:
                twin = copy.copy(sc)
                twin.ws, twin.cam, twin.focal = ws, cam, focal
                out = rc.reference_render(twin, 128, 64, noise, grad=True, force_no_cam_grad=True)
            ((out[0] * w_rgb).sum() + out[2].sum()).backward()
            assert focal.grad is None or float(focal.grad.abs().max()) == 0.0, which
            assert cam.grad is not None and float(cam.grad[:, :3, :3].abs().max()) == 0.0, which     # rotation: nothing
            assert float(cam.grad[:, :3, 3].abs().max()) > 0.0, which                                   # translation: the fine origins
            grads[which] = (ws.grad, cam.grad[:, :3, 3])
    assert rc.rel_err(grads['hip'][0], grads['ref'][0]) <= 3 * LATENTS_MEASURED, rc.rel_err(grads['hip'][0], grads['ref'][0])
    # (the camera figure is a sum over the image like GRADIENT_MEASURED['p3d']['g_cam'], 5.3e-4)
    assert rc.rel_err(grads['hip'][1], grads['ref'][1]) <= 3 * 5.3e-4, rc.rel_err(grads['hip'][1], grads['ref'][1])


def test_inversion_steps_match_the_real_reference(gpu_device):
    """BASELINE cfg3's loop on the real Generator at size and length (p3d_car-like geometry, 4 images x 128 x 128 x (64 + 64),
    30 steps of Adam on latents + camera + focal, run.py:2232-2299, --inv_steps 30) with a synthetic target (no p3d_car
    data / checkpoint exists offline): at every point of the REFERENCE's trajectory the HIP path gives the same loss and
    gradients, and its own trajectory reaches the same PSNR / IoU.  Producer on deterministic MIOpen solvers: the figures
    repeat (measured over the 30 steps, profiles/r6/reference_parity.json: loss <= 2.1e-5, d/d latents <= 2.3e-4, camera <=
    1.1e-4, focal <= 5.5e-4 - the spikes are single samples that the two fp32 forwards place on different sides of a texel
    edge; PSNR 33.98 -> 45.77 dB: reference 45.77057 dB / IoU 0.97519, HIP 45.77055 / 0.97519)."""
    _require_reference()
    with rc.deterministic_producer():
        sc = rc.build_scene('p3d', 4, gpu_device)
        r = rc.inversion(sc, 128, 64, steps=30)
    for a in r['along_reference_trajectory']:
        assert a['loss_rel'] <= 6.5e-5 and a['g_ws'] <= 7e-4 and a['g_cam'] <= 3.3e-4 and a.get('g_focal', 0.0) <= 1.7e-3, a
    (l0, p0, i0), (l1, p1, i1) = r['reference'][0], r['reference'][-1]
    (h0, q0, j0), (h1, q1, j1) = r['hip'][0], r['hip'][-1]
    assert l1 < l0 and h1 < h0 and p1 > p0 + 5.0, (r['reference'], r['hip'])                 # both descend (+ 11.8 dB)
    # free-running trajectories (each its own Adam; chaotic): measured 1.5e-5 ... 2.8e-3 dB apart after the 30 steps, IoU identical
    assert abs(q1 - p1) <= 8.5e-3 and abs(j1 - i1) <= 1e-4, (r['reference'][-1], r['hip'][-1])


def test_regulariser_branch_on_the_real_generator(gpu_device):
    """The G step's regularisers (run.py:974-979, 1011-1028; generator.py:505-585) on the real Generator in training mode:
    `attach(model, hip_regularisers=True)` serves eikonal / distance / total-variation / entropy from the HIP kernels
    (`nfi_sdf_gradient_fwd/bwd`: the eikonal term's backward is the reference's DOUBLE backward through lib/ops.grid_sample2d)
    - same seed, same two draws - against the reference's own forward: losses and gradients w.r.t. the latents (through the
    StyleGAN2 synthesis network), the decoder and beta.  Producer on deterministic MIOpen solvers; bounds = 3 x the maximum
    measured over the round's sessions (losses 7.9e-7; latents 1.9e-6 ... 2.2e-5 - 1.0e-3 without the flag -, W1 5.1e-6,
    b1 9.8e-7, W2 8.9e-7, beta 1.7e-6)."""
    _require_reference()
    with rc.deterministic_producer():
        sc = rc.build_scene('cub', 2, gpu_device)
        rep = rc.regularisers(sc)
    assert max(rep['loss_rel'].values()) <= 2e-6, rep
    for k, bound in (('ws', 3 * LATENTS_MEASURED), ('w1', 1.6e-5), ('b1', 3e-6), ('w2', 3e-6), ('beta', 5e-6)):
        assert rep['grad_rel_l2'][k] <= bound, (k, rep)
    # against the reference in float64: the latents' gradient of BOTH fp32 implementations is 1.08e-3 from it (the producer)
    assert rep['hip_vs_float64']['ws'] <= 1.5 * rep['reference_vs_float64']['ws'] + 1e-6, rep


@pytest.mark.parametrize('fused_handoff,one_forward,path_length,noisy', [(False, False, False, False), (True, False, False, False),
                                                                         (True, True, False, False), (True, True, True, False),
                                                                         (True, True, True, True)])
def test_generator_training_step_on_the_real_generator(gpu_device, fused_handoff, one_forward, path_length, noisy):
    """BASELINE cfg4's generator step on the real class (cub-like: orthographic, scene_range 2.0, black background, 4 images
    x 128 x 128 x (64 + 64), model.train(), latents through the mapping network, image + alpha loss + eikonal / distance
    regularisers, ONE backward): the gradient of EVERY generator parameter - mapping network, StyleGAN2 synthesis, texture
    mapper, decoder, beta, alpha - against the reference's own render + forward; with `fused_handoff=True` the last
    synthesis block's tail (upsample + torgb + add, stylegan.py:424-443) runs in nfi_torgb_texels_fwd / _bwd as well.
    Producer on deterministic MIOpen solvers; measured: loss 3.1e-7, all 116 tensors 8.0e-6, worst significant tensor
    6.6e-5 (decoder.net.0.weight); both fp32 implementations 9.5e-5 from the float64 reference."""
    _require_reference()
    with rc.deterministic_producer():
        # (noisy: per-layer StyleGAN2 noise on - run.py's training default - with non-zero strengths, drawn by torch.randn from
        #  the same seed in both; the strengths are parameters too.  No float64 leg: a float64 producer draws other numbers.)
        sc = rc.build_scene('cub', 4, gpu_device, stylegan_noise=noisy)
        # (one_forward: as run.py's G loop has it, 966-986 - ONE Generator.forward inside render() serves the sampler and the
        #  eikonal / total-variation / entropy regularisers, its volume draw between render's two noise draws)
        rep = rc.training_step(sc, 128, 64, fused_handoff=fused_handoff, one_forward=one_forward, path_length=path_length,
                               float64=not noisy)
    assert abs(rep['loss_hip'] - rep['loss_reference']) <= 1e-6 * abs(rep['loss_reference']), rep
    # (all parameters: 8.0e-6 in every session so far; most of them sit behind the producer's backward like the latents)
    assert rep['n_parameter_tensors'] > 100 and rep['grad_rel_l2_all_parameters'] <= 3 * LATENTS_MEASURED, rep
    assert rep['worst_tensor_rel_l2'] <= 2e-4, rep
    if noisy:
        assert rep['noise_strength_tensors'] == 13 and rep['noise_strength_grad_rel_l2'] <= 2e-4, rep
        return
    h, r = rep['hip_vs_float64'], rep['reference_vs_float64']
    assert h['all_parameters'] <= 1.5 * r['all_parameters'] and h['worst_tensor_rel_l2'] <= 1.5 * r['worst_tensor_rel_l2'] + 1e-5, rep


def test_run_py_parallel_model_calls_the_drop_in_unchanged(gpu_device):
    """INTEGRATION.md section 1 in action: run.py's own `ParallelModel` (560-617, AST-sliced - run.py cannot be imported)
    with nothing changed but the module-level name `render` it calls, and the attach()ed Generator as its model: the
    call run.py's training / eval / inversion loops make (use_ema, resolution / ray multipliers, compute_semantics, the
    per-replica loss closure) returns what the untouched ParallelModel + reference render + reference Generator return."""
    _require_reference()
    import nerf_from_image_amd.render as nfi_render
    sc = rc.build_scene('p3d', 2, gpu_device)
    ref_render, _ = reference.load_render(sc.args, sc.dcfg, unscripted_stages=True)

    pm_ref = rc.parallel_model(ref_render, sc.gen, 64, 32)
    pm_hip = rc.parallel_model(nfi_render.make_render(sc.args, sc.dcfg), sc.hip, 64, 32)
    closure_seen = []

    def closure(pm, rgb, alpha, semantics, extra_outputs, weight=1.0):
        closure_seen.append(pm)
        return weight * rgb.mean(dim=(1, 2, 3)) + alpha.mean(dim=(1, 2))
    for kw in (dict(use_ema=True), dict(use_ema=True, ray_multiplier=2, compute_semantics=True),
               dict(use_ema=False, closure=closure, closure_params={'weight': 0.5})):
        res = 64 * int(kw.get('res_multiplier', 1))
        samples = 32 * int(kw.get('ray_multiplier', 1))
        noise = rc.draw_noise(sc, res, samples)
        with torch.no_grad():
            with rc.ReplayNoise(noise):
                a = pm_hip(sc.cam, sc.focal, None, sc.bbox, sc.ws, **kw)
            with rc.ReplayNoise(noise):
                b = pm_ref(sc.cam, sc.focal, None, sc.bbox, sc.ws, **kw)
        if 'closure' in kw:
            assert a.shape == b.shape == (2,) and rc.max_err(a, b) <= BUDGET
            continue
        for k, x, y in zip(('rgb', 'depth', 'mask', 'normals', 'extra'), a[:5], b[:5]):
            assert (x is None) == (y is None), k
            if x is not None:
                assert x.shape == y.shape and rc.max_err(x, y) <= BUDGET, (k, kw, rc.max_err(x, y))
    assert closure_seen == [pm_hip, pm_ref]


def test_same_seed_gives_the_reference_noise(gpu_device):
    """No interception: the SCRIPTED reference (as run.py runs it) and the drop-in after the same torch.manual_seed draw
    the same two noise tensors from PyTorch-ROCm's Philox stream (same shapes, same order: lib/nerf_utils.py:115, 202)."""
    _require_reference()
    sc = rc.build_scene('chairs', 2, gpu_device)
    torch.manual_seed(4242)
    ours = rc.hip_render(sc, 128, 64, None)
    torch.manual_seed(4242)
    ref = rc.reference_render(sc, 128, 64, None)
    for k, a, b in zip(('rgb', 'depth', 'mask'), ours[:3], ref[:3]):
        over = int(((a - b).abs() > BUDGET).sum())
        assert over <= 2 and rc.max_err(a, b) < 5e-3, (k, rc.max_err(a, b), over)
