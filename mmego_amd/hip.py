"""ctypes binding of libmmego_hip.so (the C ABI declared in include/mmego_hip.h).

The prototypes are parsed from the header itself, so binding and header cannot drift.  There is NO
fallback: if the library is missing or a kernel launch fails, a RuntimeError is raised.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mmego_hip.h")
LIBPATH = os.path.join(_HERE, "lib", "libmmego_hip.so")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int mmego_*(...)` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(mmego_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            name = re.search(r"(\w+)$", a).group(1)
            typ = a[: -len(name)].strip()
            args.append((ctypes.c_void_p if "*" in typ else _CT[typ.replace("const ", "")], name))
        protos[m.group(1)] = args
    return protos


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise RuntimeError("libmmego_hip.so is not built (%s). Run `python -m mmego_amd.build` "
                               "(or __graft_entry__.build()); there is no CPU fallback." % LIBPATH)
        _lib = ctypes.CDLL(LIBPATH)
        _protos = parse_header()
        for name, args in _protos.items():
            fn = getattr(_lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [t for t, _ in args]
    return _lib


def _conv(v):
    if isinstance(v, torch.Tensor):
        return v.data_ptr()
    return v


def stream_handle():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Launch `mmego_<name>` on torch's current stream.  Tensors are passed as device pointers."""
    fn = getattr(lib(), "mmego_" + name)
    rc = fn(stream_handle(), *[_conv(a) for a in args])
    if rc != 0:
        raise RuntimeError("mmego_%s failed: %s" % (name, "bad argument" if rc < 0 else "hipError %d" % rc))


def graph_dA_nblk(G):
    return lib().mmego_graph_dA_nblk(G)


def colstats_nblk(rows):
    return lib().mmego_colstats_nblk(rows)
