"""rocprofv3 counter collection for one kernel of a command, in SEPARATE passes (kernel trace only - never combined with
sys/hip/hsa tracing), plus the derived per-launch figures bench.py's roofline reads.

  python tools/pmc_collect.py --kernel render_fwd_kernel --out gpurun_out/pmc_render_fwd.json \
         --marched-from-bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras

Passes: (0) kernel trace + stats, (1) FETCH_SIZE, (2) WRITE_SIZE, (3) TCC_HIT_sum TCC_MISS_sum,
(4) SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES, (5) SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM, (6) SQ_INSTS_VALU SQ_INSTS_MFMA
SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR.

Units / corrections (MI355X_MICROARCH.md, HBM section; checked against the kernel duration):
  FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE counts 128-byte requests at 64 B on gfx950 -> x2;
  SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count in units of 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

PASSES = [
    'FETCH_SIZE',
    'WRITE_SIZE',
    'TCC_HIT_sum TCC_MISS_sum',
    'SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES',
    'SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM',
    'SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR',
]


def run(cmd, log):
    with open(log, 'w') as f:
        return subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=600, cwd='/tmp')   # rocprofv3 wants a writable cwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', required=True, help='substring of the kernel name')
    ap.add_argument('--out', required=True)
    ap.add_argument('--workdir', default=None)
    ap.add_argument('--marched-from-bench', action='store_true',
                    help='the command is bench.py: take rays_marched_per_launch from its JSON line')
    ap.add_argument('--units', type=float, default=None, help='units (rays / points) processed per launch, if known')
    ap.add_argument('--pick', choices=('mean', 'max'), default='mean',
                    help='how to combine the launches of the kernel (max: the launch with the largest counter value)')
    ap.add_argument('cmd', nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == '--' else a.cmd
    work = a.workdir or (os.path.splitext(a.out)[0] + '_passes')
    os.makedirs(work, exist_ok=True)
    os.environ['TMPDIR'] = '/tmp'
    res = {'command': ' '.join(cmd), 'kernel_filter': a.kernel, 'raw': {}}

    # pass 0: kernel trace + stats
    d0 = os.path.join(work, 'p0')
    run(['rocprofv3', '--kernel-trace', '--stats', '--output-format', 'csv', '-d', d0, '-o', 'b', '--'] + cmd,
        os.path.join(work, 'p0.log'))
    units = a.units
    if a.marched_from_bench:
        for line in open(os.path.join(work, 'p0.log')):
            if line.startswith('{') and '"roofline"' in line:
                units = json.loads(line)['roofline']['rays_marched_per_launch']
    durs = []
    for f in glob.glob(os.path.join(d0, '**', '*kernel_trace.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if a.kernel in r['Kernel_Name']:
                durs.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    stats_rows = []
    for f in glob.glob(os.path.join(d0, '**', '*kernel_stats.csv'), recursive=True):
        stats_rows = list(csv.DictReader(open(f)))
    if durs:
        res['kernel_ns_mean'] = sum(durs) / len(durs)
        res['kernel_ns_min'] = min(durs)
        res['kernel_launches'] = len(durs)
    res['kernel_stats_top'] = [{k: r[k] for k in ('Name', 'Calls', 'AverageNs', 'Percentage')} for r in stats_rows[:12]]
    for i, c in enumerate(PASSES, 1):
        d = os.path.join(work, 'p%d' % i)
        run(['rocprofv3', '--kernel-trace', '--pmc'] + c.split() + ['--output-format', 'csv', '-d', d, '-o', 'b', '--'] + cmd,
            os.path.join(work, 'p%d.log' % i))
        acc = collections.defaultdict(list)
        pd = []
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if a.kernel in r['Kernel_Name']:
                    acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                if a.kernel in r['Kernel_Name']:
                    pd.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        for k, v in acc.items():
            res['raw'][k] = max(v) if a.pick == 'max' else sum(v) / len(v)
        if pd and 'GRBM_GUI_ACTIVE' in acc:
            res['kernel_ns_in_clock_pass'] = max(pd) if a.pick == 'max' else sum(pd) / len(pd)
    raw = res['raw']
    if units:
        res['rays_marched_per_launch' if a.marched_from_bench else 'units_per_launch'] = units
    g = raw.get
    if g('GRBM_GUI_ACTIVE') and res.get('kernel_ns_in_clock_pass'):
        cyc = g('GRBM_GUI_ACTIVE') / 8.0
        res['elapsed_cycles'] = cyc
        res['shader_clock_hz'] = cyc / (res['kernel_ns_in_clock_pass'] * 1e-9)
    if g('SQ_ACTIVE_INST_ANY'):
        issue = 4.0 * g('SQ_ACTIVE_INST_ANY')
        if units:
            res['issue_cycles_per_marched_ray' if a.marched_from_bench else 'issue_cycles_per_unit'] = issue / units
        if g('SQ_WAVE_CYCLES') and g('SQ_WAVES'):
            # waves resident per SIMD on average = wave-cycles / (SIMDs x elapsed); issue fraction per SIMD
            res['issue_frac_of_wave_time'] = g('SQ_ACTIVE_INST_ANY') / g('SQ_WAVE_CYCLES')
            res['wait_frac_of_wave_time'] = (g('SQ_WAIT_ANY') or 0.0) / g('SQ_WAVE_CYCLES')
        if res.get('elapsed_cycles'):
            res['issue_frac'] = issue / (1024.0 * res['elapsed_cycles'])
        if g('SQ_ACTIVE_INST_VALU'):
            valu = 4.0 * g('SQ_ACTIVE_INST_VALU')
            if units:
                res['valu_cycles_per_marched_ray' if a.marched_from_bench else 'valu_cycles_per_unit'] = valu / units
            if res.get('elapsed_cycles'):
                res['valu_frac'] = valu / (1024.0 * res['elapsed_cycles'])
        if g('SQ_WAVE_CYCLES') and res.get('elapsed_cycles'):
            res['waves_per_simd'] = 4.0 * g('SQ_WAVE_CYCLES') / (1024.0 * res['elapsed_cycles'])
    if g('TCC_HIT_sum') is not None and g('TCC_MISS_sum') is not None:
        res['l2_request_bytes_per_launch'] = (g('TCC_HIT_sum') + g('TCC_MISS_sum')) * 128.0
        res['l2_hit_rate'] = g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1.0)
    if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None:
        res['fetch_bytes_corrected'] = g('FETCH_SIZE') * 1024.0 * 2.0
        res['write_bytes'] = g('WRITE_SIZE') * 1024.0
        res['fabric_bytes_per_launch'] = res['fetch_bytes_corrected'] + res['write_bytes']
    if g('SQ_VALU_MFMA_BUSY_CYCLES') and res.get('elapsed_cycles'):
        res['mfma_busy_frac'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * res['elapsed_cycles'])   # this one counts single cycles
    json.dump(res, open(a.out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k not in ('raw', 'kernel_stats_top')}, indent=1))


if __name__ == '__main__':
    main()
