"""oracle/make_ref.py + oracle/reference.py: the staged copy of the reference's importable hot-path files (what the GPU box
gets instead of /root/reference) - manifest, integrity check, loader fallback, and that it stays out of the history."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import make_ref, reference  # noqa: E402

pytestmark = pytest.mark.skipif(not reference.available(), reason='reference sources not available (oracle/make_ref.py)')


def test_staged_copy_is_complete_and_intact():
    root = make_ref.stage()
    assert root == make_ref.DST and make_ref.staged_ok()
    man = json.load(open(os.path.join(root, 'MANIFEST.json')))
    assert sorted(man['files']) == sorted(make_ref.FILES)
    for f, h in man['files'].items():
        assert make_ref._sha(os.path.join(root, f)) == h, f
    if os.path.isdir(make_ref.REF):                       # (build container) byte-identical to the checkout
        for f in make_ref.FILES:
            assert open(os.path.join(root, f), 'rb').read() == open(os.path.join(make_ref.REF, f), 'rb').read(), f


def test_staged_copy_is_ignored_by_git_but_not_by_gpurun():
    assert 'oracle/_ref/' in open(os.path.join(ROOT, '.gitignore')).read().split()
    if os.path.isdir(os.path.join(ROOT, '.git')):          # (a checkout; an archive / GPU-box snapshot has no .git)
        r = subprocess.run(['git', 'check-ignore', '-q', 'oracle/_ref/run.py'], cwd=ROOT)
        assert r.returncode == 0, 'oracle/_ref must be git-ignored (reference sources never enter the history)'
        tracked = subprocess.run(['git', 'ls-files', 'oracle/_ref'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        assert tracked == '', tracked
    ignore = open(os.path.join(ROOT, '.gpurunignore')).read()
    assert 'oracle' not in ignore and '_ref' not in ignore          # it has to travel to the GPU box with the snapshot


def test_loader_falls_back_to_the_staged_copy():
    env = dict(os.environ, NFI_REFERENCE_CHECKOUT='/nonexistent')
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from oracle import reference as r\n'
            'assert r.root() == r.STAGED, r.root()\n'
            'm = r.modules(); ren, env = r.load_render(r.render_args(), {"scene_range": 0.55, "white_background": True})\n'
            'assert callable(ren) and hasattr(m.generator, "Generator") and hasattr(r.nerf_utils_unscripted(), "sample_pdf")\n'
            'print("ok")' % ROOT)
    p = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'ok' in p.stdout, p.stderr[-2000:]


def test_render_strictness_option_values():
    import types
    import nerf_from_image_amd.render as nfi_render
    for v, want in ((True, True), (False, False), ('deferred', 'deferred'), ('after', 'after'), (1, True)):
        assert nfi_render._strict(types.SimpleNamespace(strict_near_far=v)) == want
    with pytest.raises(TypeError):
        nfi_render.make_render(None, None, no_such_option=1)
