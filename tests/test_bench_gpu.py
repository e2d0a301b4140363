"""bench.py contract on the GPU box: one JSON line with the required keys, and the N>1 code path
(RCCL process group, barrier, MAX all-reduce of the elapsed time) exercised with a single rank under
torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_process(gpu_device):
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                        '--images-per-gpu', '2'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 3 and j['value'] > 1e6 and j['dtype'] == 'f32'
    rf = j['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9


def test_bench_distributed_code_path(gpu_device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', '29517', 'bench.py', '--gpus', '1', '--steps', '3',
                        '--warmup', '1', '--no-cpu-baseline', '--images-per-gpu', '2', '--force-dist'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = _last_json(r.stdout)
    assert j['n_gpus'] == 1 and j['value'] > 1e6 and j['scaling'] == 'weak'
