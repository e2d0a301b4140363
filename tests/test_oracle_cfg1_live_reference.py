"""BASELINE cfg1 at its exact shape (B=4, 64x64, 32 coarse samples; and 32+32) on the CPU: the oracle against the
LIVE reference - real Generator incl. the StyleGAN2 synthesis network, run.py::render AST-sliced - bit for bit.
Runs in a subprocess because the reference's scripted functions must be imported with PYTORCH_JIT=0 (the noise
draws are intercepted).  Needs the reference sources (the checkout, or the staged oracle/_ref)."""
import os
import subprocess
import sys

import pytest

from oracle import reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not reference.available(), reason='reference sources not available (oracle/make_ref.py)')
def test_cfg1_oracle_equals_live_reference():
    env = dict(os.environ, PYTORCH_JIT='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'make_golden.py'), '--cfg1'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('oracle == reference bit for bit') == 2, r.stdout


@pytest.mark.skipif(not reference.available(), reason='reference sources not available (oracle/make_ref.py)')
def test_oracle_normal_map_equals_live_reference():
    """The composited normal map as tests/parity_util.oracle_normal_map builds it - autograd of the oracle's field at the oracle's
    samples, its weights and permutation - against run.py::render(compute_normals=True) on the real Generator (CPU, same
    noise): the same samples bit for bit, so the HIP-vs-oracle bound of tests/test_hip_parity.py (3e-5) is a bound against
    the reference at equal samples."""
    env = dict(os.environ, PYTORCH_JIT='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'make_golden.py'), '--normals'], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'normal map (B=2, 32x32, 16+16 samples, real Generator): oracle == reference' in r.stdout, r.stdout
