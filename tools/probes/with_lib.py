"""Runs a tool of this repository against another build of the library:
    python tools/probes/with_lib.py build/ab/libnfi_<name>.so tools/inversion_synthetic.py --hip-only"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nerf_from_image_amd import _lib  # noqa: E402

_lib.LIBRARY = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.abspath(sys.argv[0])))
runpy.run_path(sys.argv[0], run_name='__main__')
