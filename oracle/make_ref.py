"""Stage the reference checkout for the GPU box: /root/reference  ->  oracle/_ref/   (git-ignored, shipped by gpurun).

TEST INFRASTRUCTURE ONLY (see oracle/nfi_oracle.py header): nothing under nerf_from_image_amd/ and no timed region of
bench.py may import what this writes.

The reference is pure Python, so there is nothing to compile: "building" oracle/_ref means placing the importable
files of the hot path where the GPU box (which has no /root/reference) can import them, exactly as the in-tree
libnfi_hip.so travels there.  The sources stay OUT of the repository's history: oracle/_ref/ is listed in .gitignore
(and not in .gpurunignore), and this recipe is the only thing committed.  `__graft_entry__.build()` runs it whenever
/root/reference is present; on the GPU box build() finds the staged copy and leaves it alone.

Staged (paths relative to the reference root):
  run.py                 render (176-350), ParallelModel (560-617), augment_impl (720-815): AST-sliced by the tests,
                         never imported (argparse at import time, run.py:42)
  lib/nerf_utils.py      the ray / sampling / compositing stage functions
  lib/ops.py             grid_sample2d (double-differentiable gather of the regulariser branch)
  lib/pose_utils.py      pose algebra of the augmentation test
  lib/metrics.py         psnr / iou (AST-sliced: the module imports lpips / skimage, absent here)
  models/generator.py    Generator, TriplanarDecoder, ViewDirectionMapper, the sampler closure
  models/stylegan.py     the plane producer (StyleGAN2 synthesis), EqualizedLinear

Usage:  python oracle/make_ref.py [--check]
"""
import hashlib
import json
import os
import shutil
import sys

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
DST = os.path.join(HERE, '_ref')
FILES = ['run.py', 'lib/nerf_utils.py', 'lib/ops.py', 'lib/pose_utils.py', 'lib/metrics.py', 'models/generator.py',
         'models/stylegan.py']


def _sha(path):
    return hashlib.sha256(open(path, 'rb').read()).hexdigest()


def staged_ok():
    """True when oracle/_ref holds every file of the manifest with the recorded hash."""
    man = os.path.join(DST, 'MANIFEST.json')
    if not os.path.exists(man):
        return False
    try:
        m = json.load(open(man))
        return all(os.path.exists(os.path.join(DST, f)) and _sha(os.path.join(DST, f)) == h for f, h in m['files'].items()) \
            and set(m['files']) == set(FILES)
    except Exception:
        return False


def stage(force=False):
    """Copies the files above from /root/reference into oracle/_ref/ and writes MANIFEST.json (sha256 per file).
    Returns the staged root, or None when there is neither a checkout nor a staged copy."""
    if not os.path.isdir(REF):
        return DST if staged_ok() else None
    if not force and staged_ok():
        m = json.load(open(os.path.join(DST, 'MANIFEST.json')))
        if all(_sha(os.path.join(REF, f)) == h for f, h in m['files'].items()):
            return DST
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    hashes = {}
    for f in FILES:
        dst = os.path.join(DST, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(REF, f), dst)
        hashes[f] = _sha(dst)
    with open(os.path.join(DST, 'MANIFEST.json'), 'w') as fh:
        json.dump({'source': REF, 'files': hashes,
                   'note': 'staged by oracle/make_ref.py; git-ignored; test infrastructure only'}, fh, indent=1)
    return DST


if __name__ == '__main__':
    if '--check' in sys.argv:
        print('oracle/_ref staged and intact' if staged_ok() else 'oracle/_ref missing or stale')
        sys.exit(0 if staged_ok() else 1)
    root = stage(force='--force' in sys.argv)
    print('staged reference at', root)
    sys.exit(0 if root else 1)
