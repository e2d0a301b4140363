"""Cuts a rocprofv3 --kernel-trace of `tools/end_to_end.py --leg ... --markers` into renderer / producer / other GPU time.

  python tools/phase_split.py <dir with *_kernel_trace.csv> <sidecar.json written by end_to_end.py> > split.json

The step launches one marker kernel (an elementwise op nothing else uses) at every phase boundary; every dispatch between
two markers belongs to the phase the first one opened (one stream: dispatch order = execution order).  On top of the
phases, a kernel that comes out of libnfi_hip.so (its name is a __global__ of nerf_from_image_amd/csrc) is the renderer's
wherever it runs - the drop-in transposes planes to texels and packs the decoder inside Generator.forward, and serves the
regulariser branch from its own kernels.  Groups:
  renderer  ray set-up .. compositing, their backward, (drop-in) every libnfi_hip.so kernel
  producer  Generator.forward up to the end of the synthesis network (mapping network, texture mapper, StyleGAN2
            synthesis) and its backward, minus libnfi_hip.so kernels
  forward_rest  the remainder of Generator.forward (the regulariser branch / path length in PyTorch) and its backward
  other     loss, optimiser, gradient clipping, zero_grad
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# phase opened by each marker -> group
GROUP = {'step_begin': 'other_then_renderer', 'model_begin': 'producer', 'synth_end': 'forward_rest', 'model_end': 'renderer',
         'render_end': 'other', 'loss_bwd_end': 'renderer', 'reg_bwd_begin': 'forward_rest', 'producer_bwd_begin': 'producer',
         'bwd_end': 'other'}


def own_kernels():
    names = set()
    for f in glob.glob(os.path.join(ROOT, 'nerf_from_image_amd', 'csrc', '*')):
        text = open(f).read()
        for m in re.finditer(r'__global__', text):
            k = re.search(r'\b([a-z_0-9]*kernel[a-z_0-9]*)\s*\(', text[m.end():m.end() + 400])
            if k:
                names.add(k.group(1))
    return names


def main():
    trace_dir, sidecar = sys.argv[1], json.load(open(sys.argv[2]))
    files = glob.glob(os.path.join(trace_dir, '**', '*kernel_trace.csv'), recursive=True)
    assert files, 'no kernel_trace.csv under ' + trace_dir
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # PyTorch-ROCm names the op's kernel '<op>_kernel_cuda(...)' inside a vectorized_elementwise_kernel<> or, for the ops it
    # compiles per dtype, '<op>_kernel_vectorized4_kernel' / '<op>_vectorized4_kernel'
    marker_of = {re.compile(r'(^|[^a-z])%s(_kernel|_vectorized)' % op): name for name, op in sidecar['markers'].items()}
    own = own_kernels()
    own_re = re.compile(r'\b(%s)\b' % '|'.join(sorted(own)))
    warmup, iters = sidecar['warmup'], sidecar['iters']
    step, phase, seen_in_step = -1, None, set()
    acc, own_in, launches, by_kernel = {}, {}, {}, {}
    t_first = t_last = None
    for t0, t1, name in rows:
        mk = next((v for k, v in marker_of.items() if k.search(name)), None)
        if mk is not None:
            if mk == 'step_begin':
                step += 1
                seen_in_step = set()
            if mk in ('loss_bwd_end', 'reg_bwd_begin', 'producer_bwd_begin') and mk in seen_in_step:
                continue                                   # several tensors carry the same boundary: the first one counts
            seen_in_step.add(mk)
            phase = mk
            continue
        if step < warmup or step >= warmup + iters or phase is None:
            continue
        t_first = t0 if t_first is None else t_first
        t_last = t1
        d = (t1 - t0) * 1e-6
        mine = bool(own_re.search(name))
        acc[phase] = acc.get(phase, 0.0) + d
        launches[phase] = launches.get(phase, 0) + 1
        if mine:
            own_in[phase] = own_in.get(phase, 0.0) + d
        key = ('own: ' if mine else phase + ': ') + name[:110]
        by_kernel[key] = by_kernel.get(key, 0.0) + d
    if step + 1 < warmup + iters:
        ops = '|'.join(sidecar['markers'].values())
        cand = sorted({n[:200] for _, _, n in rows if re.search(ops, n, re.I)})
        print('phase_split: %d steps found, %d expected; %d dispatches; kernel names that mention a marker op:\n  %s'
              % (step + 1, warmup + iters, len(rows), '\n  '.join(cand[:40])), file=sys.stderr)
        small = sorted({n[:160] for t0, t1, n in rows if t1 - t0 < 3000})[:80]
        print('short kernels:\n  ' + '\n  '.join(small), file=sys.stderr)
        sys.exit(1)
    groups = {'renderer': 0.0, 'producer': 0.0, 'forward_rest': 0.0, 'other': 0.0}
    for ph, ms in acc.items():
        mine = own_in.get(ph, 0.0)
        groups['renderer'] += mine
        g = GROUP[ph]
        if g == 'other_then_renderer':
            # step_begin .. model_begin: zero_grad (fill kernels), then render()'s ray set-up and first noise draw
            g = 'renderer'
        groups[g] += ms - mine
    per = {k: v / iters for k, v in groups.items()}
    total = sum(per.values())
    top = sorted(by_kernel.items(), key=lambda kv: -kv[1])[:14]
    out = {'leg': sidecar['result'].get('leg'), 'impl': sidecar['result'].get('impl'), 'options': {k: v for k, v in sidecar['result'].items()
                                                                                                   if k in ('texels', 'fused_handoff', 'path_length', 'hip_regularisers', 'batch')},
           'step_ms_events_median': sidecar['result']['ms_median'], 'steps': iters,
           'gpu_busy_ms_per_step': total, 'wall_ms_per_step_in_trace': (t_last - t_first) * 1e-6 / iters,
           'group_ms_per_step': per, 'group_share_of_gpu_time': {k: v / total for k, v in per.items()},
           'phase_ms_per_step': {k: v / iters for k, v in acc.items()}, 'phase_launches_per_step': {k: v / iters for k, v in launches.items()},
           'libnfi_kernels_ms_per_step': sum(own_in.values()) / iters,
           'top_kernels_ms_per_step': [[k, v / iters] for k, v in top]}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
