"""The legs of bench.py that are NOT the timed headline: the CPU baseline and parity figure (the reference itself, from the
staged oracle/_ref; the oracle restatement where that is missing), the reference renderer on this GPU under PyTorch-ROCm (the
>= 10 x target's denominator), and the `extras` block (render-only rates per storage type / workload, the composited extra maps, the
staged-path comparison, the end-to-end legs on the real Generator, the 16-bit storage deviation).

TEST / MEASUREMENT INFRASTRUCTURE: this is the only place outside tests/ where bench.py reaches oracle/ - after the timed
region, as the checker and the baseline, never as the thing measured (see oracle/nfi_oracle.py header)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench as _b  # noqa: E402  (the workload definition: constants, synthetic_inputs, cameras, time_render)

R, S, A, SCENE_RANGE, RADIUS = _b.R, _b.S, _b.A, _b.SCENE_RANGE, _b.RADIUS
synthetic_inputs, cameras, stats, time_render = _b.synthetic_inputs, _b.cameras, _b.stats, _b.time_render


def reference_renderer(d, device, scripted=True):
    """run.py::render (AST-sliced) on the real models/generator.py::Generator carrying this workload's field tensors,
    plane producer frozen to the synthetic planes (render only).  Sources: oracle/reference.py (the checkout, or the
    copy oracle/make_ref.py staged for the GPU box).  Returns call(cam, focal, n_images) -> 6-tuple, or None."""
    from oracle import reference
    if not reference.available():
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import reference_cases as rc
    gen = rc.generator_from_tensors(d['planes'], d['w1'], d['b1'], d['w2'], d['b2'], d['beta'], d['alpha'], SCENE_RANGE, device)
    ren, _ = reference.load_render(reference.render_args(), {'scene_range': SCENE_RANGE, 'white_background': True},
                                   unscripted_stages=not scripted)
    att = d['att'].to(device)

    def call(cam, focal, n):
        return ren(gen, R, R, cam, focal, None, None, rc.dummy_ws(n, device), S, extra_model_inputs={'attention_values': att[:n]})
    return call


def cpu_baseline_and_parity(seed, dev, ops, texels='fp32'):
    """CPU baseline on one image of the workload, and the bench's own parity figure on that image.

    kind 'reference': the reference itself - run.py::render + the real Generator's sampler, TorchScript on as run.py has
    it - on this box's host cores (sources: oracle/_ref, staged by oracle/make_ref.py).  Without the sources: kind 'port',
    the oracle (the CPU restatement pinned bit-exactly to the reference).  The same image, same noise, rendered by the
    timed HIP code path is compared with the oracle AND - where available - with the reference itself (unscripted stage
    functions, so that the noise can be handed over)."""
    from oracle import nfi_oracle as orc
    d = synthetic_inputs(1, seed, 'cpu')
    if texels != 'fp32':
        # 16-bit plane storage: the oracle gets the SAME planes the kernels gather from (rounded to the storage type)
        d['planes'] = d['planes'].to(torch.float16 if texels == 'fp16' else torch.bfloat16).to(torch.float32)
    g = torch.Generator().manual_seed(seed + 1)
    nc = torch.rand(1, R, R, S, generator=g)
    nf = torch.rand(R * R, S, generator=g)

    def oracle_once():
        return orc.render(d['planes'], d['w1'], d['b1'], d['w2'], d['b2'], d['cam'], d['focal'], R, R, S, SCENE_RANGE,
                          white_background=True, noise_coarse=nc, noise_fine=nf, use_sdf=True, beta=d['beta'],
                          alpha=d['alpha'], attention_values=d['att'])
    ref_call = reference_renderer(d, 'cpu', scripted=True)
    times = []
    with torch.no_grad():
        for i in range(4):
            t0 = time.perf_counter()
            if ref_call is not None:
                ref_call(d['cam'], d['focal'], 1)
            else:
                oracle_once()
            if i > 0:
                times.append(time.perf_counter() - t0)
        ref = oracle_once()
    med = sorted(times)[len(times) // 2]
    base = {'value': R * R / med, 'unit': 'rays/s', 'cores': torch.get_num_threads(),
            'kind': 'reference' if ref_call is not None else 'port',
            'sample': ('run.py::render + the real Generator sampler (oracle/_ref), ' if ref_call is not None else 'oracle restatement, ') +
                      '1 image 128x128, 64+64 samples, planes precomputed (render only), fp32, 1 warm-up + 3 timed runs, median'}
    dd = {k: v.to(dev) for k, v in d.items()}
    tdt = {'fp32': ops.TEXEL_F32, 'fp16': ops.TEXEL_F16, 'bf16': ops.TEXEL_BF16}[texels]
    texel_t = ops.planes_to_texels(dd['planes'], tdt)
    image = ops.decoder_pack(dd['w1'], dd['b1'], dd['w2'], dd['b2'], A, tdt)
    out = ops.render_fwd(dd['cam'], dd['focal'], R, R, S, texel_t, image, SCENE_RANGE, A, dd['att'], True, dd['beta'],
                         dd['alpha'], noise_coarse=nc.to(dev), noise_fine=nf.to(dev), fine_sampling=True,
                         white_background=True, skip_missed_rays=True)
    keys = ('rgb', 'depth', 'mask')
    vs_cpu = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in keys}
    parity = dict(vs_cpu)                    # top-level rgb / depth / mask: against the pinned CPU oracle
    parity.update(budget=1e-4, against='CPU oracle (reference ATen numerics, pinned to the live reference), 1 image of this '
                                       'workload, same noise' + ('' if texels == 'fp32' else ', planes rounded to the %s storage' % texels),
                  ok=bool(max(vs_cpu.values()) <= 1e-4 and all(bool(torch.isfinite(out[k]).all()) for k in keys)),
                  mask_mean=float(ref['mask'].mean()))
    # the reference itself, same image, same noise: on the CPU (must equal the oracle bit for bit) and on this GPU
    # (PyTorch-ROCm: its elementwise kernels contract a*b+c into FMAs, so it differs from its own CPU path: gap printed)
    if ref_call is not None:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import reference_cases as rc
        plain_cpu = reference_renderer(d, 'cpu', scripted=False)
        plain_gpu = reference_renderer(dd, dev, scripted=False)
        with torch.no_grad():
            with rc.ReplayNoise([nc, nf]):
                r_cpu = dict(zip(keys, plain_cpu(d['cam'], d['focal'], 1)[:3]))
            with rc.ReplayNoise([nc.to(dev), nf.to(dev)]):
                r_gpu = dict(zip(keys, plain_gpu(dd['cam'], dd['focal'], 1)[:3]))
        vs_gpu = {k: float((out[k] - r_gpu[k]).abs().max()) for k in keys}
        gap = {k: float((r_gpu[k].cpu() - r_cpu[k]).abs().max()) for k in keys}
        parity.update(vs_reference_cpu={k: float((out[k].cpu() - r_cpu[k]).abs().max()) for k in keys},
                      oracle_equals_reference_cpu_bit_for_bit=bool(all(torch.equal(r_cpu[k], ref[k]) for k in keys)),
                      vs_reference_pytorch_rocm=vs_gpu, reference_cpu_vs_pytorch_rocm_gap=gap,
                      ok_vs_reference_pytorch_rocm=bool(all(vs_gpu[k] <= gap[k] + 1e-4 for k in keys)))
    else:
        with torch.no_grad():
            ref_gpu = orc.render(dd['planes'], dd['w1'], dd['b1'], dd['w2'], dd['b2'], dd['cam'], dd['focal'], R, R, S, SCENE_RANGE,
                                 white_background=True, noise_coarse=nc.to(dev), noise_fine=nf.to(dev), use_sdf=True,
                                 beta=dd['beta'], alpha=dd['alpha'], attention_values=dd['att'])
        vs_gpu = {k: float((out[k] - ref_gpu[k]).abs().max()) for k in keys}
        gap = {k: float((ref_gpu[k].cpu() - ref[k]).abs().max()) for k in keys}
        parity.update(vs_pytorch_rocm_oracle=vs_gpu, oracle_cpu_vs_pytorch_rocm_gap=gap,
                      ok_vs_pytorch_rocm=bool(all(vs_gpu[k] <= gap[k] + 1e-4 for k in keys)))
    return base, parity


def time_staged_semantics(ops, dev, n_img, radius, iters=20):
    """render(compute_semantics=True) as the STAGED path runs it (nerf_from_image_amd/render.py: ray set-up, near/far,
    stratified points, field query with semantics, resampling, fine points, second field query, merge + composite of
    rgb and the A-channel map): the per-sample tensors go through HBM.  Same inputs as time_render."""
    from nerf_from_image_amd import nerf_utils
    dd = synthetic_inputs(n_img, 4321, dev)
    g = torch.Generator().manual_seed(77)
    dd['cam'] = cameras(n_img, radius, g).to(dev)
    texels = ops.planes_to_texels(dd['planes'])
    image = ops.decoder_pack(dd['w1'], dd['b1'], dd['w2'], dd['b2'], A)
    gn = torch.Generator(device=dev).manual_seed(99)
    nc = torch.rand((n_img, R, R, S), device=dev, generator=gn)
    nf = torch.rand((n_img * R * R, S), device=dev, generator=gn)

    def query(pts):
        q = ops.field_query(pts.reshape(n_img, -1, 3), texels, image, SCENE_RANGE, A, dd['att'], True, dd['beta'], dd['alpha'],
                            want_semantics=True, mlp_precision=1)
        shp = pts.shape[:-1]
        return q['sigma'].view(*shp), q['rgb'].view(*shp, 3), q['semantics'].view(*shp, A)

    def once():
        ro, rd = nerf_utils.get_ray_bundle_normalized(R, R, dd['focal'], dd['cam'], None, None)
        near, far = nerf_utils.compute_near_far_planes(ro, rd, SCENE_RANGE, strict=False)
        pts, dep = nerf_utils.compute_query_points_from_rays(ro, rd, near, far, S, randomize=True, noise=nc)
        sig, rgb, sem = query(pts)
        z, _ = ops.resample(sig, rd, dep, nf)
        z = z.view(*dep.shape[:3], S)
        sig_f, rgb_f, sem_f = query(nerf_utils.points_on_rays(ro, rd, z))
        return nerf_utils.merge_and_composite(rd, dep, sig, rgb, z, sig_f, rgb_f, None, None, sem, sem_f, white_background=True)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    with torch.no_grad():
        for i in range(iters + 3):
            if i >= 3:
                evs[i - 3].record()
            once()
        evs[iters].record()
    torch.cuda.synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(iters)]
    return {'rays_per_s': n_img * R * R * iters / (sum(per) * 1e-3), 'ms': stats(per), 'iters': iters}


def pytorch_rocm_reference(dev):
    """The >= 10x target's denominator: the reference renderer on PyTorch-ROCm on this GPU - run.py::render + the real
    Generator's sampler (TorchScript on, as run.py runs it), planes precomputed (render only), B = 1 / 4 / 8 images of
    the workload, HIP events, best of 3 after 2 warm-up calls.  Without the staged sources: the oracle's op sequence."""
    from oracle import nfi_oracle as orc
    dd = synthetic_inputs(8, 4321, dev)
    call = reference_renderer(dd, dev, scripted=True)
    res = {'unit': 'rays/s', 'kind': 'reference' if call is not None else 'port'}
    for n in (1, 4, 8):
        cam, focal = dd['cam'][:n].contiguous(), dd['focal'][:n].contiguous()
        times = []
        with torch.no_grad():
            for i in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                if call is not None:
                    call(cam, focal, n)
                else:
                    orc.render(dd['planes'][:n], dd['w1'], dd['b1'], dd['w2'], dd['b2'], cam, focal, R, R, S, SCENE_RANGE,
                               white_background=True, noise_coarse=torch.rand((n, R, R, S), device=dev),
                               noise_fine=torch.rand((n * R * R, S), device=dev), use_sdf=True, beta=dd['beta'],
                               alpha=dd['alpha'], attention_values=dd['att'][:n])
                b.record()
                torch.cuda.synchronize()
                times.append(a.elapsed_time(b) * 1e-3)
        res['b%d' % n] = n * R * R / min(times[2:])
    res['value'] = max(res['b1'], res['b4'], res['b8'])
    res['sample'] = (('run.py::render (oracle/_ref) + the real Generator sampler' if call is not None else
                      'oracle (reference ATen op sequence)') +
                     ' on this GPU under PyTorch-ROCm, 1 / 4 / 8 images 128x128, 64+64, render only, fp32, best of 3 '
                     'after 2 warm-up calls, HIP events; value = the best of the three batch sizes')
    torch.cuda.empty_cache()
    return res


def extras(dev, ops):
    """Untimed-side measurements reported next to the headline (never part of `value`)."""
    ex = {'render_only': {}}
    cases = {
        'b1_chairs_fp32_texels': (1, RADIUS, ops.TEXEL_F32, {}),
        'b8_chairs_fp32_texels': (8, RADIUS, ops.TEXEL_F32, {}),
        'b8_all_rays_hit_fp32_texels': (8, 1.3, ops.TEXEL_F32, {}),
        'b8_chairs_bf16_texels': (8, RADIUS, ops.TEXEL_BF16, {}),
        # fp16 plane storage (fp32 arithmetic): packed texels blended with v_fma_mix_f32, three blocks per CU
        'b8_chairs_fp16_texels': (8, RADIUS, ops.TEXEL_F16, {}),
        'b8_all_rays_hit_fp16_texels': (8, 1.3, ops.TEXEL_F16, {}),
        # the same kernels with the decoder MLP on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) instead of split fp16
        'b8_chairs_fp32_texels_mlp_exact_fp32': (8, RADIUS, ops.TEXEL_F32, {'tuning': 8}),
        'b8_all_rays_hit_fp32_texels_mlp_exact_fp32': (8, 1.3, ops.TEXEL_F32, {'tuning': 8}),
        # BASELINE config 5 geometry on one GPU: 256x256 rays, 128 + 128 samples per ray
        'b2_cfg5_256px_128+128_fp32_texels': (2, RADIUS, ops.TEXEL_F32, {'R': 256, 'S': 128}),
        'b2_cfg5_256px_128+128_fp16_texels': (2, RADIUS, ops.TEXEL_F16, {'R': 256, 'S': 128}),
    }
    exact_out = {}
    for name, (n_img, radius, tdt, kw) in cases.items():
        ex['render_only'][name], exact_out[name] = time_render(ops, dev, n_img, radius, tdt, **kw)
    # the composited extra maps of run.py:312-338 from the SAME fused launch (compute_coords: every encoder-training
    # iteration, run.py:1639-1646; compute_semantics: every inversion eval batch, run.py:2036-2051), with the staged path
    # (one launch per stage, every per-sample tensor through HBM: what these calls cost before round 4) beside them
    maps = {}
    for name in ('b8_chairs_fp32_texels', 'b8_all_rays_hit_fp32_texels', 'b2_cfg5_256px_128+128_fp32_texels'):
        n_img, radius, tdt, kw = cases[name]
        for label, mkw in (('coords', dict(want_coords=True)), ('semantics', dict(want_semantics=True)),
                           ('normals', dict(want_normals=True)),
                           ('normals+semantics', dict(want_normals=True, want_semantics=True))):   # the first eval batch, run.py:2036-2051
            r, out = time_render(ops, dev, n_img, radius, tdt, **mkw, **kw)
            r['x_plain_rate'] = r['rays_per_s'] / ex['render_only'][name]['rays_per_s']
            r['rgb_bit_identical_to_plain'] = bool(torch.equal(out['rgb'], exact_out[name]['rgb']))
            maps['%s_%s' % (name, label)] = r
    maps['b8_chairs_fp32_texels_semantics_staged_path'] = time_staged_semantics(ops, dev, 8, RADIUS)
    maps['b8_chairs_fp32_texels_semantics_staged_path']['x_plain_rate'] = (
        maps['b8_chairs_fp32_texels_semantics_staged_path']['rays_per_s'] / ex['render_only']['b8_chairs_fp32_texels']['rays_per_s'])
    ex['extra_maps_fused'] = maps
    del exact_out
    ex['pytorch_rocm_reference_path'] = pytorch_rocm_reference(dev)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    # END TO END on the real reference classes, plane producer included (SURVEY.md 8(d) metric (ii)): render() incl.
    # Generator.forward at cfg2 B = 1 / 4 / 8, one --run_inversion step (cfg3 shape), one generator-side GAN step (cfg4
    # shape) - the HIP drop-in against the untouched reference on this GPU, same weights (tools/end_to_end.py; the kernel
    # split of the same legs: profiles/r6/e2e/)
    from oracle import reference
    if reference.available():
        try:
            import end_to_end
            ex['end_to_end_real_generator'] = end_to_end.summary(dev, quick=True)
        except Exception as e:      # reported, never fatal for the headline line
            ex['end_to_end_real_generator'] = {'error': repr(e)}
        # what 16-bit plane STORAGE costs against the fp32 reference (BASELINE words cfg2 with bf16, cfg5 with fp16)
        try:
            import reference_cases as rc
            dev_ = {}
            scenes = {}
            for name in ('cfg2_b8_128px_64+64_bf16_texels', 'cfg2_b8_128px_64+64_fp16_texels'):
                r = rc.config_case(name, dev, cpu_images=0, scenes=scenes)
                dev_[name] = {'max_abs': r['vs_reference_gpu'], 'mean_abs': r['mean_abs_vs_reference_gpu'],
                              'against': 'run.py::render + the real Generator on fp32 planes, PyTorch-ROCm, same noise'}
            ex['texel_storage_vs_fp32_reference'] = dev_
        except Exception as e:
            ex['texel_storage_vs_fp32_reference'] = {'error': repr(e)}
        torch.cuda.empty_cache()
    else:
        ex['end_to_end_real_generator'] = {'error': 'reference sources not staged (oracle/make_ref.py)'}
    return ex


