"""Access to the LIVE reference code for the tests and the bench's baseline legs.

TEST INFRASTRUCTURE ONLY (see oracle/nfi_oracle.py header): imported by tests/, by bench.py's `cpu_baseline` /
`pytorch_rocm_reference_path` legs (after the timed region) and by oracle/make_golden.py; never by the product.

The reference checkout is /root/reference in the build container; on the GPU box it is the copy staged by
oracle/make_ref.py under oracle/_ref/ (git-ignored, shipped with the snapshot).  `root()` names whichever exists.

  modules()                 lib.nerf_utils, lib.ops, lib.pose_utils, models.generator, models.stylegan as a namespace
  nerf_utils_unscripted()   lib/nerf_utils.py once more with TorchScript off (noise interception)
  load_render(args, cfg)    run.py::render (176-350), AST-sliced (run.py is a script: argparse at import, run.py:42)
  slice_functions(...)      any other top-level def / class of a reference file, the same way
"""
import ast
import os
import sys
import types
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CHECKOUT = os.environ.get('NFI_REFERENCE_CHECKOUT', '/root/reference')      # (the override lets a test exercise the staged copy)
STAGED = os.path.join(HERE, '_ref')
_cache = {}


def root():
    """Path of the reference sources, or None (neither the checkout nor a staged copy)."""
    if os.path.isdir(CHECKOUT):
        return CHECKOUT
    if os.path.exists(os.path.join(STAGED, 'MANIFEST.json')) and os.path.exists(os.path.join(STAGED, 'run.py')):
        return STAGED
    return None


def available():
    return root() is not None


def modules():
    """The importable reference modules, imported the way run.py imports them (TorchScript as the process has it: on
    by default - what the reference runs with -, off under PYTORCH_JIT=0 as oracle/make_golden.py sets it)."""
    r = root()
    if r is None:
        raise RuntimeError('no reference sources: neither %s nor %s (run oracle/make_ref.py where the checkout exists)'
                           % (CHECKOUT, STAGED))
    if 'mods' not in _cache:
        sys.path.insert(0, r)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                from lib import nerf_utils, ops, pose_utils
                from models import generator, stylegan
        finally:
            sys.path.remove(r)
        _cache['mods'] = types.SimpleNamespace(nerf_utils=nerf_utils, ops=ops, pose_utils=pose_utils, generator=generator,
                                               stylegan=stylegan, root=r)
    return _cache['mods']


def nerf_utils_unscripted():
    """A second, private import of lib/nerf_utils.py with TorchScript disabled: its @torch.jit.script stage functions
    are then plain Python, so that a test can hand BOTH implementations the same noise by intercepting torch.rand /
    torch.rand_like (scripted code calls aten::rand directly and cannot be intercepted)."""
    if 'nu_plain' not in _cache:
        import importlib.util
        state = torch.jit._state
        was_enabled = bool(state._enabled)
        state.disable()
        try:
            spec = importlib.util.spec_from_file_location('nfi_reference_nerf_utils_unscripted',
                                                          os.path.join(root(), 'lib', 'nerf_utils.py'))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        finally:
            if was_enabled:
                state.enable()
        _cache['nu_plain'] = mod
    return _cache['nu_plain']


def slice_functions(rel_path, names, env):
    """exec()s the named top-level defs / classes of a reference file into `env` and returns it."""
    path = os.path.join(root(), rel_path)
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    missing = set(names) - {n.name for n in body}
    if missing:
        raise KeyError('%s has no top-level %s' % (rel_path, sorted(missing)))
    exec(compile(ast.Module(body=body, type_ignores=[]), rel_path + '::' + '+'.join(names), 'exec'), env)
    return env


def load_render(args, dataset_config, unscripted_stages=False):
    """run.py::render bound to `args` (use_viewdir, use_sdf, attention_values, fine_sampling) and `dataset_config`
    (scene_range, white_background): the globals it reads.  unscripted_stages: bind it to nerf_utils_unscripted()
    (noise interception).  Returns (render, its globals dict)."""
    import torch.nn.functional as F
    nu = nerf_utils_unscripted() if unscripted_stages else modules().nerf_utils
    env = {'torch': torch, 'F': F, 'nerf_utils': nu, 'args': args, 'dataset_config': dataset_config}
    slice_functions('run.py', ['render'], env)
    return env['render'], env


def render_args(fine_sampling=True, use_sdf=True, attention_values=10, use_viewdir=False):
    return types.SimpleNamespace(fine_sampling=fine_sampling, use_sdf=use_sdf, attention_values=attention_values,
                                 use_viewdir=use_viewdir)
