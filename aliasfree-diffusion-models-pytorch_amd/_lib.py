"""ctypes binding of libafd_hip.so (the C ABI declared in include/afd.h).

The argument types are parsed from the header itself, so the binding cannot drift from it.
There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "afd.h")
LIBPATH = os.environ.get("AFD_LIBPATH") or os.path.join(_PKG, "libafd_hip.so")     # (AFD_LIBPATH: another build of the SAME library, for same-box A/B runs)

_CTYPES = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
}
_DECL = re.compile(r"^(int|size_t|const char\*)\s+(afd_\w+)\s*\(([^;]*?)\)\s*;", re.M | re.S)


def _strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    out = {}
    for ret, name, args in _DECL.findall(_strip_comments(open(path).read())):
        argtypes = []
        args = " ".join(args.split())
        if args not in ("", "void"):
            for a in args.split(","):
                a = a.strip()
                if "*" in a or a.startswith("afd_stream_t"):
                    argtypes.append(ctypes.c_void_p)
                else:
                    base = a.split()[0] if not a.startswith("unsigned") else a
                    argtypes.append(_CTYPES[base])
        restype = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[ret]
        out[name] = (restype, argtypes)
    return out


_VALUE_RETURNING = {"afd_device_count", "afd_tok_supported", "afd_conv3x3_weight_kinds", "afd_conv_wgrad_form"}      # int results, not status codes


class AfdError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIBPATH):
            raise AfdError(
                f"{LIBPATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback.")
        self.cdll = ctypes.CDLL(LIBPATH)
        self.sigs = parse_header()
        for name, (restype, argtypes) in self.sigs.items():
            fn = getattr(self.cdll, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = restype, argtypes
            if restype is ctypes.c_int and name not in _VALUE_RETURNING:
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)

    def _checked(self, name, fn):
        last_error = self.cdll.afd_last_error
        last_error.restype = ctypes.c_char_p

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise AfdError(f"{name} failed (code {rc}): {last_error().decode()}")
        call.__name__ = name
        return call


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
