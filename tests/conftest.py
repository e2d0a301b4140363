import json
import os
import sys
import tempfile

# A private MIOpen user database for the test process (and the processes it starts): the gradient tests run the plane
# producer on MIOpen's deterministic solvers, and what MIOpen records while they run must not follow a LATER process - the
# end-to-end legs of bench.py - around through ~/.config/miopen.  (Set before the first convolution initialises MIOpen.)
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='nfi_miopen_db_'))

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden_case_names():
    return sorted(json.load(open(os.path.join(GOLDEN, 'cases.json'))).keys())


def load_golden(name, device='cpu'):
    """Returns (meta dict, tensors dict) for a committed golden case."""
    meta = json.load(open(os.path.join(GOLDEN, 'cases.json')))[name]
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    t = {k: torch.from_numpy(z[k]).to(device) for k in z.files}
    planes = torch.from_numpy(np.load(os.path.join(GOLDEN, 'planes.npz'))['planes'])
    t['planes'] = planes[:meta['B']].contiguous().to(device)
    return meta, t


@pytest.fixture(scope='session')
def gpu_device(request):
    if not torch.cuda.is_available():
        # the harness fails loudly.  A run that ASKS for the GPU tests (-m gpu) on a box without a GPU is
        # an error, not a pass with skips; only a run that merely collects them (no -m expression) skips.
        expr = request.config.getoption('-m') or ''
        if 'gpu' in expr and 'not gpu' not in expr:
            pytest.fail('GPU tests requested (-m gpu) but no GPU is visible (torch.cuda.is_available() is False)', pytrace=False)
        pytest.skip('no GPU visible')
    return torch.device('cuda:0')
