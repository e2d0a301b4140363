"""Run-to-run spread of the gradient comparisons of tests/reference_cases.py on this GPU: `repeats` fresh runs of
rc.gradients() per geometry, with the producer's convolutions as PyTorch picks them and with
torch.backends.cudnn.deterministic (MIOpen: no atomic split-K weight-gradient solvers).  JSON on stdout
(profiles/r6/gradient_spread.json).  Test infrastructure."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import reference_cases as rc  # noqa: E402


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device('cuda:0')
    out = {}
    for det in (False, True):
        for geometry, res, samples in (('chairs', 128, 64), ('p3d', 128, 64), ('cub', 128, 64), ('carla', 64, 32)):
            rows = []
            for _ in range(repeats):
                with torch.backends.cudnn.flags(enabled=True, deterministic=det, benchmark=False):
                    r = rc.gradients(rc.build_scene(geometry, 2, dev), res, samples)
                rows.append({k: v for k, v in r.items() if not k.startswith('loss')})
                torch.cuda.empty_cache()
            out['%s%s' % (geometry, '_deterministic_producer' if det else '')] = rows
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
