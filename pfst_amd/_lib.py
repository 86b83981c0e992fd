"""ctypes binding of libpfst_hip.so.  Signatures are parsed from include/pfst_hip.h, so the header is
the single source of truth for the C ABI.  Loading fails LOUDLY: there is no CPU / eager fallback."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, '..', 'include', 'pfst_hip.h')
LIB_PATH = os.environ.get('PFST_HIP_LIB') or os.path.join(HERE, 'libpfst_hip.so')   # override: A/B builds of the kernels

_SCALARS = {'int': ctypes.c_int, 'long long': ctypes.c_longlong, 'float': ctypes.c_float, 'double': ctypes.c_double}


class PfstHipError(RuntimeError):
    pass


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every `pfst_*` declaration."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    decls = {}
    for m in re.finditer(r'(const\s+char\s*\*|int)\s+(pfst_\w+)\s*\(([^)]*)\)\s*;', text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if 'char' in ret else ctypes.c_int
        argl = []
        if args and args != 'void':
            for a in args.split(','):
                a = ' '.join(a.split())
                an = a.split()[-1].lstrip('*')
                ty = a[:a.rfind(an)].strip()
                if '*' in ty or ty == 'pfst_stream_t':
                    argl.append((ctypes.c_void_p, an))
                else:
                    argl.append((_SCALARS[ty.replace('const ', '')], an))
        decls[name] = (restype, argl)
    return decls


_lib = None
_decls = None


def lib():
    global _lib, _decls
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PfstHipError(f'{LIB_PATH} is missing: run `python -m pfst_amd.build` (hipcc, gfx950). '
                               'pfst_amd has no fallback path.')
        L = ctypes.CDLL(LIB_PATH)
        _decls = parse_header()
        for name, (restype, args) in _decls.items():
            fn = getattr(L, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype = restype
            fn.argtypes = [a[0] for a in args]
        _lib = L
    return _lib


def call(name, *args):
    """Invoke an int-returning entry point; raise with the library's message on failure."""
    L = lib()
    # ctypes accepts SURPLUS arguments on a cdecl function without a word (the declared ones are converted, the rest passed through): an edit
    # that appends an argument to the wrong call site would shift e.g. the stream out of its slot -- refuse any count but the header's
    want = len(_decls[name][1])
    if len(args) != want:
        raise TypeError(f'{name} takes {want} arguments (include/pfst_hip.h), got {len(args)}')
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise PfstHipError(f'{name} failed ({rc}): {L.pfst_last_error().decode()}')
