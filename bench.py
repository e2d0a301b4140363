"""Headline benchmark: rendered rays/s of the volumetric-rendering hot path (BASELINE.json cfg2).

  python bench.py --gpus N --steps K --warmup W            (N>1: python -m torch.distributed.run --nnodes=1
                                                             --nproc-per-node N ... bench.py --gpus N ...)
  python bench.py --mode train --gpus N ...                 cfg4-like generator training step with the RCCL
                                                             gradient all-reduce (tools/train_bench.py)

One step (render mode) = one pass of the hot path over one batch per GPU: triplane hand-off (NCHW -> channel-last
texels), decoder operand packing, the reference's two torch.rand draws, ray set-up and the fused render of
IMAGES_PER_GPU images at 128x128 with 64 coarse + 64 fine samples (fp32 planes, fp32 accumulation; decoder MLP
operands as split fp16 hi+lo pairs, see config.mlp).  Planes, decoder weights and cameras are synthetic and already
resident in HBM.  Images are sharded across ranks (weak scaling, no collective on the render path: SURVEY.md 8(e)).

`value` is the SERIAL schedule: one stream, every step after the previous one - what a caller of the drop-in render()
gets.  The two-stream schedule of round 3 (the next steps' fronts prepared on a second stream) is still timed, as the side
field `value_pipelined`; it is a property of this script, not of the API.

Before the W warm-up steps the script runs the same step untimed for `--prewarm-ms` (default 300 ms): the caching allocator
and the shader clock (2.30 GHz in the first steps after idle, 2.38 GHz settled) are then in steady state, and a short run -
the driver's `--steps 20 --warmup 5` - reads the same rate as a long one (141 M rays/s; 134 M without it).  Reported in
`prewarm`.

Printed JSON (rank 0, one line) carries, besides the contract fields:
  config.value_mlp_exact_fp32 / .value_all_rays_hit / .value_pipelined - the same whole step, same K, same protocol, with
                      the decoder MLP on exact-fp32 MFMA (tuning bit 3) / with cameras at radius 1.3 (every ray crosses
                      the scene cube, nothing is skipped) / under the two-stream schedule (scalars inside `config`, where
                      the driver's parser keeps them);
  config.x_pytorch_rocm_reference - `value` over the reference renderer's rays/s on this very GPU (run.py::render + the
                      real Generator sampler under PyTorch-ROCm, best of B = 1 / 4 / 8; extras.pytorch_rocm_reference_path);
  ms_per_step_stats - min / median / max over the K timed steps (HIP events per step);
  roofline          - the fused render kernel against SURVEY.md 8(d)'s arithmetic: achieved = algorithmic decoder FLOPs
                      (704 512 per marched ray x rays marched per launch) / kernel duration (live HIP events on the launch
                      stream), peak = 157.3 TFLOP/s (fp32 matrix / vector peak), `traffic` = fabric bytes per launch
                      (FETCH_SIZE x2 + WRITE_SIZE from the committed rocprofv3 PMC profile named in `source`, scaled by
                      the live ray count).  The pipe that binds by the counters is the vector ALU: `valu_pipe.frac`
                      = 4 x SQ_ACTIVE_INST_VALU per marched ray x rays / kernel time / (1024 SIMDs x the shader clock
                      measured in the timed launches), also as the scalar `valu_frac`; `bound` is 'valu' (the MFMA pipe is
                      busy 9 % of the time).  `any_issue_proxy` (SQ_ACTIVE_INST_ANY: VALU + LDS + VMEM + SALU
                      issue of the resident waves, which overlap) is a utilisation figure, not a bound.  `levels` carries
                      the byte-side figures, each against its own peak: L2 request bytes (34.5 TB/s), fabric bytes
                      (8 TB/s), compulsory HBM bytes, the cache-served algorithmic gather stream (196 608 B per marched
                      ray: above the HBM peak, so not a bound);
  per_rank          - ms per step, kernel ms, fraction of rays marched for every rank (N > 1);
  parity            - max |error| of ONE image of this very workload rendered by the timed code path against the CPU
                      oracle (untimed; budget 1e-4 on rgb / depth / mask);
  cpu_baseline      - kind 'reference': run.py::render + the real Generator's sampler (oracle/_ref, staged by
                      oracle/make_ref.py) timed on this box's host cores on ONE image of the same workload; kind 'port' (the
                      oracle restatement) only where the staged sources are missing.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (only the extras' end-to-end legs and --mode train run convolutions: MIOpen starts from an empty user database - a fresh
#  box's state - whatever ran on this box before; the timed render step contains no MIOpen kernel)
if 'MIOPEN_USER_DB_PATH' not in os.environ:
    import tempfile
    os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='nfi_miopen_db_')

import torch  # noqa: E402

R, S, A, PLANE_RES = 128, 64, 10, 256
IMAGES_PER_GPU = 8
SCENE_RANGE, RADIUS, FOCAL = 0.55, 2.0, 1.0254      # shapenet_chairs-like
GATHER_BYTES_PER_RAY = 196608                        # 128 points x 12 texels x 128 B
HBM_PEAK_GBS = 8000.0
L2_PEAK_GBS = 34500.0                                # MI355X_MICROARCH.md, L2 (per XCD, aggregate)
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_F16_PEAK_TFLOPS = 2500.0
MLP_FLOP_PER_RAY = 704512
N_SIMD = 1024
PMC_PROFILES = ('profiles/r6/pmc_render_fwd.json', 'profiles/r5/pmc_render_fwd.json', 'profiles/r4/pmc_render_fwd.json', 'profiles/r3/pmc_render_fwd.json', 'profiles/r2/pmc_render_fwd.json', 'profiles/r1/pmc_render_fwd_derived.json')


def cameras(n, radius, gen):
    v = torch.randn(n, 3, generator=gen)
    eye = radius * v / v.norm(dim=-1, keepdim=True)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)
    up = torch.tensor([0., 0., 1.]).expand(n, 3)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    tup = torch.cross(right, fwd, dim=-1)
    cam = torch.eye(4).repeat(n, 1, 1)
    cam[:, :3, 0] = right
    cam[:, :3, 1] = tup
    cam[:, :3, 2] = -fwd
    cam[:, :3, 3] = eye
    return cam


def synthetic_inputs(n_images, seed, dev):
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(n_images * 3, 32, 16, 16, generator=g)
    planes = torch.nn.functional.interpolate(low, size=(PLANE_RES, PLANE_RES), mode='bilinear', align_corners=True)
    planes = (planes + 0.2 * torch.randn(n_images * 3, 32, PLANE_RES, PLANE_RES, generator=g))
    d = dict(planes=planes.view(n_images, 3, 32, PLANE_RES, PLANE_RES).contiguous(),
             w1=torch.randn(64, 32, generator=g), b1=0.3 * torch.randn(64, generator=g),
             w2=torch.randn(1 + A, 64, generator=g), b2=0.3 * torch.randn(1 + A, generator=g),
             att=torch.rand(n_images, A, 3, generator=g) * 2 - 1, beta=torch.tensor([0.1]), alpha=torch.tensor([0.05]),
             cam=cameras(n_images, RADIUS, g), focal=torch.full((n_images,), FOCAL))
    # centre the decoder's distance output (sign change inside the cube => surfaces to render); done
    # with a closed-form shift of b2[0] from a CPU probe of the first-layer statistics
    feat = planes.view(n_images, 3, 32, -1).mean(dim=(1, 3))                 # mean plane feature per scene
    hid = torch.nn.functional.softplus(feat.mean(0) @ (d['w1'] / 32 ** 0.5).t() + d['b1'])
    d['b2'][0] -= float((hid @ (d['w2'][0] / 8.0)) + d['b2'][0])
    return {k: v.to(dev) for k, v in d.items()}


class HipEvents:
    """Raw hipEvent pair (the render kernel is bracketed inside the C ABI call, on its stream)."""

    def __init__(self):
        self.hip = ctypes.CDLL('libamdhip64.so')
        self.a, self.b = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(self.a)) == 0
        assert self.hip.hipEventCreate(ctypes.byref(self.b)) == 0

    def pair(self):
        return (self.a.value, self.b.value)

    def elapsed_ms(self):
        ms = ctypes.c_float()
        self.hip.hipEventSynchronize(self.b)
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), self.a, self.b) == 0
        return ms.value


def stats(xs):
    xs = sorted(xs)
    return {'min': xs[0], 'median': xs[len(xs) // 2], 'max': xs[-1]}


def time_render(ops, dev, n_img, radius, texel_dtype, iters=50, R=R, S=S, tuning=0, **render_kw):
    """Render-only rays/s of one configuration: HIP events around every call (on the launch stream), `iters` calls."""
    dd = synthetic_inputs(n_img, 4321, dev)
    g = torch.Generator().manual_seed(77)
    dd['cam'] = cameras(n_img, radius, g).to(dev)
    texels = ops.planes_to_texels(dd['planes'], texel_dtype)
    image = ops.decoder_pack(dd['w1'], dd['b1'], dd['w2'], dd['b2'], A, texel_dtype)
    gn = torch.Generator(device=dev).manual_seed(99)          # the same noise for every variant of a configuration
    nc = torch.rand((n_img, R, R, S), device=dev, generator=gn)
    nf = torch.rand((n_img * R * R, S), device=dev, generator=gn)
    ws = None
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    for i in range(iters + 5):
        if i >= 5:
            evs[i - 5].record()
        out = ops.render_fwd(dd['cam'], dd['focal'], R, R, S, texels, image, SCENE_RANGE, A, dd['att'], True,
                             dd['beta'], dd['alpha'], noise_coarse=nc, noise_fine=nf, workspace=ws, tuning=tuning,
                             **render_kw)
        ws = out['_workspace']
    evs[iters].record()
    torch.cuda.synchronize()
    per = [evs[i].elapsed_time(evs[i + 1]) for i in range(iters)]
    n = n_img * R * R
    return {'rays_per_s': n * iters / (sum(per) * 1e-3), 'ms': stats(per), 'iters': iters}, out


def cpu_baseline_and_parity(seed, dev, ops, texels='fp32'):
    """The CPU baseline (kind 'reference': run.py::render + the real Generator's sampler from the staged oracle/_ref) and the
    bench's own parity figure - tools/bench_legs.py; after the timed region, never part of `value`."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_legs
    return bench_legs.cpu_baseline_and_parity(seed, dev, ops, texels)


def extras(dev, ops):
    """Untimed-side measurements reported next to the headline (tools/bench_legs.py; never part of `value`)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench_legs
    return bench_legs.extras(dev, ops)


def load_pmc_profile(texels='fp32'):
    for rel in (PMC_PROFILES if texels == 'fp32' else ('profiles/r5/pmc_render_fwd_%s.json' % texels, 'profiles/r4/pmc_render_fwd_%s.json' % texels, 'profiles/r3/pmc_render_fwd_%s.json' % texels)):
        p = os.path.join(ROOT, rel)
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                if 'issue_cycles_per_marched_ray' in d:
                    return rel, d
            except Exception:
                pass
    return None, None


def roofline(kernel_ms, marched, n_images, live_clock_hz=None, texels='fp32'):
    """Roofline object of the fused render kernel (see the module docstring).  live_clock_hz: the shader clock measured
    INSIDE the timed launches (nfi_render_args.clock_probe: s_memtime / s_memrealtime of one persistent wave); the peak
    is priced at it, so that kernel time and clock come from the same run (without it: the profile's own clock)."""
    t = kernel_ms * 1e-3
    src, prof = load_pmc_profile(texels)
    gather_gbs = marched * GATHER_BYTES_PER_RAY / t / 1e9
    compulsory = n_images * (3 * 32 * PLANE_RES * PLANE_RES * 4 + R * R * (2 * S * 4 + 33 + 20))   # planes + noise + ray set-up + outputs
    mlp_tflops = marched * MLP_FLOP_PER_RAY / t / 1e12
    levels = {
        'hbm_compulsory': {'bytes_per_launch': compulsory, 'achieved': compulsory / t / 1e9, 'peak': HBM_PEAK_GBS,
                           'unit': 'GB/s', 'frac': compulsory / t / 1e9 / HBM_PEAK_GBS},
        'gather_stream_algorithmic': {'bytes_per_launch': marched * GATHER_BYTES_PER_RAY, 'achieved': gather_gbs,
                                      'unit': 'GB/s', 'x_hbm_peak': gather_gbs / HBM_PEAK_GBS,
                                      'note': 'served by L1/L2/Infinity Cache, NOT a bound (exceeds the HBM peak)'},
        'mfma': {'achieved': mlp_tflops, 'unit': 'TFLOP/s', 'peak_fp32': MFMA_F32_PEAK_TFLOPS,
                 'frac_of_fp32_matrix_peak': mlp_tflops / MFMA_F32_PEAK_TFLOPS,
                 'note': 'useful MLP FLOPs (704 512 per ray); issued as split-fp16 MFMA: 3 products per fp32-equivalent '
                         'FLOP against the 2.5 PFLOP/s fp16 peak',
                 'frac_of_fp16_matrix_peak_issued': 3 * mlp_tflops / MFMA_F16_PEAK_TFLOPS},
    }
    # algorithmic decoder FLOPs per launch / kernel time against the fp32 matrix / vector peak
    r = {'bound': 'valu', 'kernel': 'render_fwd_kernel', 'kernel_ms': kernel_ms, 'rays_marched_per_launch': marched,
         'achieved': mlp_tflops, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': mlp_tflops / MFMA_F32_PEAK_TFLOPS,
         'frac_flops': mlp_tflops / MFMA_F32_PEAK_TFLOPS, 'flop_per_marched_ray': MLP_FLOP_PER_RAY, 'traffic': None,
         'source': src, 'levels': levels,
         'note': 'frac = 704 512 decoder FLOP x rays marched / kernel time / 157.3 TFLOP/s (SURVEY.md 8(d)); the binding pipe '
                 'by the counters is the vector ALU (valu_pipe.frac): gather address + bilinear blend, hi/lo splits, '
                 'quarter-rate softplus transcendentals and the per-ray stages all issue there, and on gfx950 MFMA time is '
                 'VALU time (tools/probes/mfma_valu_overlap.hip)'}
    if prof is None:
        r.update(valu_pipe=None, note_pmc='no PMC profile with issue_cycles_per_marched_ray under profiles/: run '
                                          'tools/gpu_session.sh <tag> pmc (tools/pmc_collect.py)')
        return r
    clk = live_clock_hz or prof['shader_clock_hz']
    peak_cycles = N_SIMD * clk
    units = prof['rays_marched_per_launch']
    valu_per_ray = prof.get('valu_cycles_per_marched_ray') or (
        4.0 * prof['raw']['SQ_ACTIVE_INST_VALU'] / units if prof.get('raw', {}).get('SQ_ACTIVE_INST_VALU') else None)
    scale = marched / prof['rays_marched_per_launch']                 # byte counters scale with the rays marched
    l2 = prof['l2_request_bytes_per_launch'] * scale
    fab = prof['fabric_bytes_per_launch'] * scale
    levels['l2_requests'] = {'bytes_per_launch': l2, 'achieved': l2 / t / 1e9, 'peak': L2_PEAK_GBS, 'unit': 'GB/s',
                             'frac': l2 / t / 1e9 / L2_PEAK_GBS, 'hit_rate': prof.get('l2_hit_rate')}
    levels['fabric'] = {'bytes_per_launch': fab, 'achieved': fab / t / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': fab / t / 1e9 / HBM_PEAK_GBS, 'x_compulsory': fab / compulsory,
                        'note': 'FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; Infinity-Cache hits included; WRITE_SIZE '
                                'is mostly the write-back of the preceding kernels\' dirty lines (texel hand-off, rand)'}
    any_cycles = prof['issue_cycles_per_marched_ray'] * marched / t
    r.update(traffic=fab, shader_clock_hz=clk,
             valu_frac=None if valu_per_ray is None else valu_per_ray * marched / t / peak_cycles,
             shader_clock_source='live: s_memtime / s_memrealtime of a persistent wave of the timed launches'
             if live_clock_hz else 'the PMC profile (GRBM_GUI_ACTIVE / 8 / kernel time of its clock pass)',
             valu_pipe={'cycles_per_marched_ray': valu_per_ray,
                        'achieved_gcycles_per_s': None if valu_per_ray is None else valu_per_ray * marched / t / 1e9,
                        'peak_gcycles_per_s': peak_cycles / 1e9,
                        'frac': None if valu_per_ray is None else valu_per_ray * marched / t / peak_cycles,
                        'note': '4 x SQ_ACTIVE_INST_VALU per marched ray (PMC profile) x rays marched (live) / kernel time '
                                '(live) over 1024 SIMDs x the live shader clock: the binding pipe'},
             any_issue_proxy={'cycles_per_marched_ray': prof['issue_cycles_per_marched_ray'],
                              'frac': any_cycles / peak_cycles, 'frac_in_profile_run': prof.get('issue_frac'),
                              'note': '4 x SQ_ACTIVE_INST_ANY: VALU + LDS + VMEM + SALU issue of the resident waves, which '
                                      'overlap - a utilisation proxy (passes 1 at three waves per SIMD), NOT a bound'},
             waves_per_simd=prof.get('waves_per_simd'),
             mfma_busy_frac=prof.get('mfma_busy_frac'), wait_frac_of_wave_time=prof.get('wait_frac_of_wave_time'))
    return r


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (how the driver calls it): re-run this very command line under
    torch.distributed.run, one process per GPU on 127.0.0.1; rank 0 prints the JSON line, the exit code is passed on."""
    import socket
    import subprocess
    avail = torch.cuda.device_count()
    if n > avail:
        raise SystemExit('--gpus %d but only %d GPU(s) are visible' % (n, avail))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--mode', choices=('render', 'train'), default='render')
    ap.add_argument('--images-per-gpu', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle leg (and the parity figure)')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--no-skip', action='store_true', help='march rays that miss the scene cube too')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL even with one rank (exercises the N>1 code path)')
    ap.add_argument('--bucket-mb', type=int, default=32, help='train mode: gradient bucket size')
    ap.add_argument('--reduce-mode', choices=('all_reduce', 'reduce_scatter'), default='all_reduce')
    ap.add_argument('--no-overlap', action='store_true', help='train mode: launch the collectives after backward')
    ap.add_argument('--texels', choices=('fp32', 'fp16', 'bf16'), default='fp32',
                    help='storage type of the texels the kernels gather from (arithmetic stays fp32)')
    ap.add_argument('--pipelined', action='store_true',
                    help='render mode: `value` under the two-stream schedule instead of the serial one (HISTORY.md 4.3)')
    ap.add_argument('--prewarm-ms', type=float, default=300.0,
                    help='render mode: untimed steps for this long before the W warm-up steps (allocator and shader clock '
                         'in steady state, HISTORY.md 4.3); 0 switches it off')
    ap.add_argument('--no-variants', action='store_true',
                    help='render mode: skip the value_mlp_exact_fp32 / value_all_rays_hit / value_pipelined legs')
    ap.add_argument('--prefetch-depth', type=int, default=2,
                    help='render mode, two-stream schedule: how many steps ahead the front of a step is prepared')
    ap.add_argument('--render-streams', type=int, default=None,
                    help='render mode, two-stream schedule: consecutive render kernels alternate over this many streams '
                         '(default 1; 2 with fp16 texels)')
    ap.add_argument('--serial', action='store_true', help='render mode: one stream, every step after the previous one (the default)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.force_dist):
        return self_launch(args.gpus)
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # the library is prebuilt in-tree; if it ever has to be (re)built, exactly one rank does it
    import __graft_entry__ as entry
    if local_rank == 0:
        entry.build()
    if use_dist:
        dist.barrier()
    from nerf_from_image_amd import ops

    if args.mode == 'train':
        return train_mode(args, dev, rank, world, use_dist)

    B = args.images_per_gpu or IMAGES_PER_GPU
    d = synthetic_inputs(B, 1234 + rank, dev)
    # the every-ray-hits variant of the workload: the same scenes seen from radius 1.3 (the cube fills the image)
    d_hit = dict(d, cam=cameras(B, 1.3, torch.Generator().manual_seed(77 + rank)).to(dev))
    n_rays = B * R * R
    ev = HipEvents()
    probe = torch.zeros(2, dtype=torch.int64, device=dev)
    tdt = {'fp32': ops.TEXEL_F32, 'fp16': ops.TEXEL_F16, 'bf16': ops.TEXEL_BF16}[args.texels]

    def prepare(v, slot_ws=None):
        """Everything of a step in front of the render kernel: texel hand-off of the producer's planes, decoder operand
        image, the two noise draws, the ray set-up (rays, scene-cube test, miss-fill reduction) into the slot's workspace."""
        dv, vt = v['d'], v.get('tdt', tdt)
        return dict(texels=ops.planes_to_texels(dv['planes'], vt), image=ops.decoder_pack(dv['w1'], dv['b1'], dv['w2'], dv['b2'], A, vt),
                    noise_c=torch.rand((B, R, R, S), dtype=torch.float32, device=dev),
                    noise_f=torch.rand([n_rays, S], dtype=torch.float32, device=dev),
                    ws=ops.render_setup(dv['cam'], dv['focal'], R, R, SCENE_RANGE, workspace=slot_ws))

    def render(v, pre, timed_kernel=False):
        dv = v['d']
        return ops.render_fwd(dv['cam'], dv['focal'], R, R, S, pre['texels'], pre['image'], SCENE_RANGE, A, dv['att'], True,
                              dv['beta'], dv['alpha'], noise_coarse=pre['noise_c'], noise_fine=pre['noise_f'],
                              fine_sampling=True, white_background=True, skip_missed_rays=not args.no_skip,
                              workspace=pre['ws'], rays_ready=True, events=ev.pair() if timed_kernel else None,
                              clock_probe=probe if timed_kernel else None, tuning=v['tuning'])

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    prep_stream = torch.cuda.Stream(device=dev)
    n_render_streams = args.render_streams or (2 if args.texels == 'fp16' else 1)
    render_streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(max(1, n_render_streams) - 1)]

    def run_steps(v, n, pipelined, marks=None, timed_kernel=False, after=None):
        """n steps of workload variant v.  Serial: one stream, what a caller of render() gets.  Pipelined: the front of step
        i + depth runs on a second HIP stream while step i renders (depth + 1 slots; a slot is refilled only after the
        render that read it has finished).  Every step does all of its work either way."""
        main = torch.cuda.current_stream(dev)
        if not pipelined:
            for i in range(n):
                pre = prepare(v, v.get('ws'))
                v['ws'] = pre['ws']
                out = render(v, pre, timed_kernel)
                if marks is not None:
                    marks[i + 1].record()
                if after is not None:
                    after()
            return out
        depth = max(1, args.prefetch_depth)
        slots = [{} for _ in range(depth + 1)]

        def fill(i):
            slot = slots[i % (depth + 1)]
            with torch.cuda.stream(prep_stream):
                if 'done' in slot:
                    prep_stream.wait_event(slot['done'])
                slot['pre'] = prepare(v, slot['pre']['ws'] if 'pre' in slot else None)
                slot['ready'] = torch.cuda.Event()
                slot['ready'].record(prep_stream)
        prep_stream.wait_stream(main)
        for j in range(min(depth, n)):
            fill(j)
        for rs in render_streams[1:]:
            rs.wait_stream(main)
        for i in range(n):
            if i + depth < n:
                fill(i + depth)
            slot = slots[i % (depth + 1)]
            rs = render_streams[i % len(render_streams)]
            rs.wait_event(slot['ready'])
            with torch.cuda.stream(rs):
                out = render(v, slot['pre'], timed_kernel)
                # one event per step: the step's timing mark also says 'this slot may be refilled'
                slot['done'] = marks[i + 1] if marks is not None else torch.cuda.Event()
                slot['done'].record(rs)
            if after is not None:
                after()
        for rs in render_streams[1:]:
            main.wait_stream(rs)
        main.wait_stream(prep_stream)
        return out

    def timed(v, pipelined):
        """W untimed + exactly K timed steps of variant v, barrier + synchronize on both sides, max over ranks."""
        run_steps(v, args.warmup, pipelined)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        fence()
        t0 = time.perf_counter()
        marks[0].record()
        run_steps(v, args.steps, pipelined, marks)
        fence()
        local = time.perf_counter() - t0
        per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        elapsed = local
        if use_dist:
            t = torch.tensor([local], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, per_step, local

    headline = {'d': d, 'tuning': 0}
    pipelined = bool(args.pipelined) and not args.serial
    prewarm_steps = 0
    if args.prewarm_ms > 0:
        # untimed: bring the caching allocator and the shader clock to steady state (see --prewarm-ms)
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            run_steps(headline, 10, pipelined)
            torch.cuda.synchronize()
            prewarm_steps += 10
    elapsed, per_step, local_elapsed = timed(headline, pipelined)

    variants = {}
    if not args.no_variants:
        legs = [('all_rays_hit', {'d': d_hit, 'tuning': 0}, pipelined),
                ('pipelined' if not pipelined else 'serial', {'d': d, 'tuning': 0}, not pipelined)]
        if args.texels == 'fp32':
            # (the exact-fp32 MLP is built for fp32 texels: with 16-bit storage the texels set the precision)
            legs.insert(0, ('mlp_exact_fp32', {'d': d, 'tuning': 8}, pipelined))
            # BASELINE cfg2 names bf16: the same whole step with the triplanes handed over as bf16 texels (arithmetic fp32)
            legs.append(('bf16_texels', {'d': d, 'tuning': 0, 'tdt': ops.TEXEL_BF16}, pipelined))
        for name, v, pl in legs:
            e_v, per_v, _ = timed(v, pl)
            variants[name] = {'value': world * n_rays * args.steps / e_v, 'ms_per_step': e_v / args.steps * 1e3,
                              'ms_per_step_stats': stats(per_v)}

    # ---- dominant kernel, timed live with HIP events on its own stream (untimed extra launches) ----
    k_ms, k_clk = [], []

    def read_kernel_events():
        k_ms.append(ev.elapsed_ms())
        cyc, ticks = (int(v) for v in probe.tolist())
        if ticks > 0:
            k_clk.append(cyc / ticks * 1e8)                # s_memrealtime ticks at 100 MHz
    run_steps(headline, min(50, max(5, args.steps)), pipelined, timed_kernel=True, after=read_kernel_events)    # same schedule as the timed steps
    kernel_ms = sum(k_ms) / len(k_ms)
    live_clock = sum(k_clk) / len(k_clk) if k_clk else None
    # rays the kernel marches = rays whose line meets the cube inflated by 1e-4 (the kernel's own skip test, fp32)
    import numpy as np
    wide = float(np.float32(SCENE_RANGE) * np.float32(1.0001))

    def marched_rays(dv):
        ro, rd = ops.raygen(R, R, dv['focal'], dv['cam'], normalize=True)
        return int(ops.near_far(ro, rd, wide, strict=False)[2].sum().item()) if not args.no_skip else n_rays
    marched = marched_rays(d)
    mine = {'rank': rank, 'ms_per_step': local_elapsed / args.steps * 1e3, 'ms_per_step_stats': stats(per_step),
            'kernel_ms': kernel_ms, 'rays_marched_fraction': marched / n_rays, 'shader_clock_hz': live_clock}
    per_rank = [mine]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        value = world * n_rays * args.steps / elapsed
        two_stream = ('two HIP streams: texel hand-off + decoder pack + noise draws + ray set-up of step i+%d overlap the '
                      'render kernel of step i (%d slots; every step does all of its work)%s'
                      % (max(1, args.prefetch_depth), max(1, args.prefetch_depth) + 1,
                         '; consecutive render kernels alternate over %d streams' % len(render_streams) if len(render_streams) > 1 else ''))
        res = {
            'metric': 'rendered rays/sec (128x128, 64+64 samples)', 'value': value, 'unit': 'rays/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'ms_per_step_stats': stats(per_step),
            'prewarm': {'ms': args.prewarm_ms, 'untimed_steps': prewarm_steps,
                        'why': 'allocator + shader clock in steady state before the W warm-up and K timed steps'},
            'schedule': two_stream if pipelined else 'one stream, serial steps: what a caller of the drop-in render() gets',
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (decoder MLP operands split-fp16 hi+lo, 22 bits; strict fp32: config.value_mlp_exact_fp32)',
            'data': 'synthetic',
            'config': {'workload': 'cfg2: shapenet_chairs-like forward render, %d images/GPU, 128x128 rays/image, '
                                   '64 coarse + 64 fine samples, 3x256x256x32 fp32 triplanes, SDF decoder, A=10; '
                                   'step = texel hand-off + decoder pack + 2 rand draws + ray set-up + fused render kernel'
                                   % B,
                       'mlp': 'split-fp16: decoder MLP operands as fp16 hi+lo pairs (22 significand bits), products '
                              'hi*hi + hi*lo + lo*hi accumulated in fp32 on v_mfma_f32_16x16x32_f16; everything else '
                              'fp32 (the same whole step with the exact-fp32 MFMA: value_mlp_exact_fp32)',
                       'texel_storage': args.texels + (' (arithmetic fp32)' if args.texels != 'fp32' else ''),
                       'images_per_gpu': B, 'resolution': R, 'samples': '64+64', 'plane_res': PLANE_RES,
                       'camera_radius': RADIUS, 'scene_range': SCENE_RANGE, 'rays_marched_fraction': marched / n_rays,
                       'skip_missed_rays': not args.no_skip, 'sharding': 'images across ranks, no collective'},
        }
        if variants:
            # scalar keys inside `config`: the driver's parser keeps config / roofline / cpu_baseline and drops unknown
            # top-level keys, so the strict-fp32 and all-rays-hit whole-step rates live here
            other = 'pipelined' if not pipelined else 'serial'
            cfg = res['config']
            cfg['value_serial' if not pipelined else 'value_pipelined'] = value
            cfg['value_' + other] = variants[other]['value']
            if 'mlp_exact_fp32' in variants:
                cfg['value_mlp_exact_fp32'] = variants['mlp_exact_fp32']['value']
            cfg['value_all_rays_hit'] = variants['all_rays_hit']['value']
            if 'bf16_texels' in variants:
                cfg['value_bf16_texels'] = variants['bf16_texels']['value']
            cfg['rays_marched_fraction_all_rays_hit'] = marched_rays(d_hit) / n_rays
            res['variants'] = dict(variants, note='the SAME whole step, K, warm-up and max-over-ranks protocol as `value`: '
                                   'mlp_exact_fp32 = decoder MLP on v_mfma_f32_16x16x4_f32 (tuning bit 3), the strictly-fp32 '
                                   'rate; all_rays_hit = cameras at radius 1.3, every ray crosses the scene cube; bf16_texels = the triplanes handed over '
                                   'as bf16 texels (the storage type BASELINE cfg2 names; arithmetic fp32, parity against the rounded planes); %s = %s'
                                   % (other, two_stream if other == 'pipelined' else 'one stream'))
        res['roofline'] = roofline(kernel_ms, marched, B, live_clock, args.texels)
        res['kernel_ms_stats'] = stats(k_ms)
        res['per_rank'] = per_rank
        if world == 1 and not args.no_extras:
            res['extras'] = extras(dev, ops)      # before the CPU leg: its OpenMP workers keep spinning for a while
            ref_gpu = res['extras'].get('pytorch_rocm_reference_path') or {}
            if ref_gpu.get('value'):
                # the north star's ">= 10x the reference PyTorch-ROCm renderer at 1 GPU": this step's whole-job rate over
                # the reference's best batch size on the same GPU (render only, which favours the reference: `value`
                # also pays for the texel hand-off, the decoder pack and the noise draws)
                res['config']['pytorch_rocm_reference_rays_per_s'] = ref_gpu['value']
                res['config']['pytorch_rocm_reference_kind'] = ref_gpu['kind']
                res['config']['x_pytorch_rocm_reference'] = value / ref_gpu['value']
            e2e = (res['extras'].get('end_to_end_real_generator') or {}).get('render_incl_synthesis', {}).get('b%d' % B)
            if e2e:
                # SURVEY.md 8(d) metric (ii): render() INCLUDING the real plane producer (StyleGAN2 synthesis, PyTorch-ROCm /
                # MIOpen in both implementations), HIP drop-in vs the untouched reference, and the share of that
                # end-to-end call this bench's whole render step (hand-off + pack + noise + set-up + kernel) accounts for
                cfg = res['config']
                cfg['end_to_end_rays_per_s'] = e2e['hip_fp32_texels_rays_per_s']
                cfg['end_to_end_reference_rays_per_s'] = e2e['reference_rays_per_s']
                cfg['x_reference_end_to_end'] = e2e['x_reference_fp32_texels']
                cfg['renderer_share_of_end_to_end_render'] = res['ms_per_step'] / e2e['hip_fp32_texels_ms']
                b1 = (res['extras'].get('end_to_end_real_generator') or {}).get('render_incl_synthesis', {}).get('b1') or {}
                if 'hip_fp32_texels_hip_graph_rays_per_s' in b1:
                    # one image per call is launch bound (~380 producer launches): the whole call as ONE HIP graph
                    # (nerf_from_image_amd/graphs.py) against the eager drop-in and the reference
                    cfg['end_to_end_b1_rays_per_s'] = b1['hip_fp32_texels_rays_per_s']
                    cfg['end_to_end_b1_hip_graph_rays_per_s'] = b1['hip_fp32_texels_hip_graph_rays_per_s']
                    cfg['x_reference_end_to_end_b1_hip_graph'] = b1['x_reference_hip_graph']
            dev16 = res['extras'].get('texel_storage_vs_fp32_reference') or {}
            for tx in ('bf16', 'fp16'):
                r16 = dev16.get('cfg2_b8_128px_64+64_%s_texels' % tx)
                if r16:
                    res['config']['%s_vs_fp32_reference_rgb' % tx] = r16['max_abs']['rgb']
                    res['config']['%s_vs_fp32_reference_rgb_mean' % tx] = r16['mean_abs']['rgb']
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'], res['parity'] = cpu_baseline_and_parity(1234, dev, ops, args.texels)
        else:
            res['cpu_baseline'] = None
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


def train_mode(args, dev, rank, world, use_dist):
    """cfg4-like generator training step (tools/train_bench.py): render fwd + regularisers + bwd + gradient
    all-reduce (GradientBuckets over RCCL) + Adam, 4 images per GPU."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import train_bench
    import torch.distributed as dist
    B = args.images_per_gpu or 4
    r = train_bench.run(dev, steps=args.steps, warmup=args.warmup, batch=B, bucket_mb=args.bucket_mb,
                        reduce_mode=args.reduce_mode, overlap=not args.no_overlap, use_dist=use_dist, texels=args.texels)
    elapsed = r['elapsed_s']
    per_rank = [r]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        per_rank = [None] * world
        dist.all_gather_object(per_rank, r)
    if rank == 0:
        alone = r['allreduce_alone_ms']
        res = {
            'metric': 'training rays/sec (cfg4-like generator step: render fwd + regularisers + bwd + gradient all-reduce + Adam)',
            'value': world * r['rays_per_step'] * args.steps / elapsed, 'unit': 'rays/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (decoder MLP operands split-fp16 hi+lo, 22 bits, forward and backward recompute; gradients, plane-'
                     'gradient image and optimiser fp32)',
            'data': 'synthetic',
            'config': {'workload': 'cfg4-like: %d images/GPU, 128x128 orthographic rays, 64+64 samples, scene_range 2.0, '
                                   'black background, image + alpha loss, eikonal + distance regularisers, gradient '
                                   'all-reduce of a generator-sized fp32 set (%d parameters), fused Adam'
                                   % (B, r['n_params']),
                       'collective': 'none (single rank)' if not use_dist else
                                     'RCCL %s, %d buckets of <= %d MiB, %s' % (
                                         args.reduce_mode, r['n_buckets'], args.bucket_mb,
                                         'launched after backward' if args.no_overlap else
                                         'launched from post-accumulate hooks during backward'),
                       'gradient_bytes_per_step': r['gradient_bytes'],
                       'texel_storage': args.texels + (' (arithmetic and plane gradient fp32)' if args.texels != 'fp32' else '')},
            'per_rank': [{k: pr[k] for k in ('ms_per_step', 'fwd_bwd_ms', 'allreduce_exposed_ms', 'optimiser_ms',
                                             'allreduce_alone_ms', 'buckets_launched_in_backward', 'loss')}
                         for pr in per_rank],
            'allreduce_bus_gbs_alone': (2 * (world - 1) / world * r['gradient_bytes'] / (alone * 1e-3) / 1e9)
            if (alone and world > 1) else None,
            'cpu_baseline': None,
        }
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
