"""ctypes binding of libnfi_hip.so (the C ABI declared in include/nfi_hip.h).

The argument structs are generated from the header at import time, so the
Python side cannot drift from the C side.  There is no CPU fallback: if the
library is missing or a call fails, a RuntimeError is raised (the reference's
error convention is Python exceptions).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(_ROOT, 'include', 'nfi_hip.h')
LIBRARY = os.path.join(_HERE, 'libnfi_hip.so')

_SCALARS = {
    'int': ctypes.c_int, 'float': ctypes.c_float, 'int64_t': ctypes.c_int64, 'int32_t': ctypes.c_int32,
    'uint32_t': ctypes.c_uint32, 'uint8_t': ctypes.c_uint8, 'size_t': ctypes.c_size_t,
}


def _strip_comments(src):
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    return re.sub(r'//[^\n]*', ' ', src)


def parse_header(path=HEADER):
    """Returns (structs, functions): {name: [(field, ctype)]}, {name: (restype, [argtypes])}."""
    src = _strip_comments(open(path).read())
    structs = {}
    for m in re.finditer(r'typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
        fields = []
        for decl in m.group(2).split(';'):
            decl = decl.strip()
            if not decl:
                continue
            dm = re.match(r'(const\s+)?(\w+)\s*(.*)', decl, flags=re.S)
            base, rest = dm.group(2), dm.group(3)
            for item in rest.split(','):
                item = item.strip()
                ptr = item.count('*')
                name = item.replace('*', '').strip()
                if ptr or base == 'void':
                    fields.append((name, ctypes.c_void_p))
                else:
                    fields.append((name, _SCALARS[base]))
        structs[m.group(3)] = fields
    functions = {}
    for m in re.finditer(r'(?:^|\n)\s*(const\s+char\s*\*|int|size_t)\s+(nfi_\w+)\s*\(([^)]*)\)\s*;', src):
        ret = {'int': ctypes.c_int, 'size_t': ctypes.c_size_t}.get(m.group(1).strip(), ctypes.c_char_p)
        args = []
        params = m.group(3).strip()
        if params and params != 'void':
            for p in params.split(','):
                p = p.strip()
                if '*' in p or p.startswith('nfi_stream_t'):
                    args.append(ctypes.c_void_p)
                else:
                    args.append(_SCALARS[p.split()[0]])
        functions[m.group(2)] = (ret, args)
    return structs, functions


STRUCT_FIELDS, FUNCTIONS = parse_header()


def _make_struct(name, fields):
    return type(name, (ctypes.Structure,), {'_fields_': fields})


STRUCTS = {n: _make_struct(n, f) for n, f in STRUCT_FIELDS.items()}

_lib = None


def load():
    """Loads libnfi_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY):
        raise RuntimeError(
            'libnfi_hip.so not found at %s: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(there is no CPU fallback for the HIP path)' % LIBRARY)
    # torch first: its wheel carries its own HIP runtime (libamdhip64), and a process must end up with ONE - loaded the other
    # way round (this library and /opt/rocm's runtime, then torch's) the launches here fail with 'no ROCm-capable device'
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIBRARY)
    for fname, (ret, args) in FUNCTIONS.items():
        fn = getattr(lib, fname)  # AttributeError here = header/library mismatch
        fn.restype = ret
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().nfi_last_error().decode()


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, last_error()))


def ptr(t):
    """Device pointer of a tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def make_args(struct_name, **kw):
    s = STRUCTS[struct_name]()
    known = {f for f, _ in STRUCT_FIELDS[struct_name]}
    for k, v in kw.items():
        if k not in known:
            raise KeyError('%s has no field %s' % (struct_name, k))
        if hasattr(v, 'data_ptr'):
            v = v.data_ptr()
        setattr(s, k, v)
    return s


def call_struct(fname, struct_name, stream, **kw):
    lib = load()
    a = make_args(struct_name, **kw)
    check(getattr(lib, fname)(ctypes.byref(a), ctypes.c_void_p(stream)), fname)


def struct_query(fname, struct_name, **kw):
    """size_t f(const struct*) style helpers (workspace sizes)."""
    lib = load()
    a = make_args(struct_name, **kw)
    return int(getattr(lib, fname)(ctypes.byref(a)))
