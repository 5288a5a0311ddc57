"""ctypes binding of the libvaecap C ABI (include/vaecap.h).

The prototypes are parsed from the header itself, so the header stays the single
source of truth.  There is NO fallback: if the shared library is missing or a
call fails, an exception is raised (the product path never computes on the CPU).
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "vaecap.h")
LIB_PATH = os.path.join(_HERE, "lib", "libvaecap.so")

_CTYPES = {
    "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p, "const float*": ctypes.c_void_p,
    "int32_t*": ctypes.c_void_p, "const int32_t*": ctypes.c_void_p,
    "uint32_t*": ctypes.c_void_p, "const uint32_t*": ctypes.c_void_p, "double*": ctypes.c_void_p, "const double*": ctypes.c_void_p, "double": ctypes.c_double,
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "size_t": ctypes.c_size_t, "uint64_t": ctypes.c_uint64,
    "const char*": ctypes.c_char_p,
    "void**": ctypes.c_void_p, "int*": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: (restype_str, [(type_str, arg_name), ...])} for every vc_* prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\s*\*)\s+(vc_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        ret = re.sub(r"\s*\*", "*", ret)
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"^(.*?)(\w+)$", a)
                typ = mm.group(1).strip().replace(" *", "*")
                alist.append((typ, mm.group(2)))
        protos[name] = (ret, alist)
    return protos


class VaecapError(RuntimeError):
    pass


class Lib(object):
    """Checked access: lib.vc_xxx(...) raises VaecapError on a non-zero return code."""

    def __init__(self, cdll, protos):
        self._cdll = cdll
        self._protos = protos
        self._cache = {}
        cdll.vc_last_error.restype = ctypes.c_char_p

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        fn = self._cache.get(name)
        if fn is None:
            if name not in self._protos:
                raise AttributeError("%s is not declared in include/vaecap.h" % name)
            ret, args = self._protos[name]
            raw = getattr(self._cdll, name)
            raw.argtypes = [_CTYPES[t] for t, _ in args]
            raw.restype = _CTYPES[ret]
            if ret == "int" and name not in ("vc_abi_version", "vc_sumsq_blocks", "vc_embedding_index_max_vocab", "vc_comm_available", "vc_trace_available", "vc_adam_blocks", "vc_gemm_get_precision") and not name.endswith("_supported") and not name.endswith("_preferred"):  # value-returning ints
                def fn(*a, _raw=raw, _name=name):
                    rc = _raw(*a)
                    if rc != 0:
                        msg = self._cdll.vc_last_error()
                        raise VaecapError("%s failed (code %d): %s" % (_name, rc, msg.decode() if msg else ""))
                    return 0
            else:
                fn = raw
            self._cache[name] = fn
        return fn


_LIB = None


def load(path=None):
    """Load libvaecap.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    # PyTorch-ROCm bundles its own libamdhip64; load it FIRST so that libvaecap binds to the same HIP
    # runtime (two runtimes in one process each enumerate the GPU and the second one sees no device).
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise VaecapError("libvaecap.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % p)
    lib = Lib(ctypes.CDLL(p), parse_header())
    if path is None:
        _LIB = lib
    return lib


def ptr(t):
    """Device pointer of a torch tensor (or None / int passthrough)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()
