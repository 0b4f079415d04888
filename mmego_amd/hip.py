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
LIBPATH = os.environ.get("MMEGO_HIP_LIB") or os.path.join(_HERE, "lib", "libmmego_hip.so")    # (MMEGO_HIP_LIB: A/B builds of scripts/)

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
    if isinstance(v, (ctypes.Array, ctypes.Structure)):   # a host-side descriptor (table): kept alive by whoever holds the argument list
        return ctypes.addressof(v)
    return v


def ptr(t):
    """Device address of a tensor (or None / an integer address) for a descriptor field."""
    if t is None:
        return None
    return t.data_ptr() if isinstance(t, torch.Tensor) else int(t)


def is_bf16_mfma_entry(name):
    """True for the entry points (name without the mmego_ prefix) whose kernels run bf16 MFMAs or belong to their chains: everything
    exported by csrc/split3.hip, bf16.hip and *_bf16.hip (tests/test_host_cpu.py checks that no other file contains a bf16 MFMA and
    that every entry point of those files matches).  What the exclusivity check of plan.unordered_with() calls an aggressor."""
    return name.startswith("split3_") or "bf16" in name


def stream_handle():
    return torch.cuda.current_stream().cuda_stream


class GemmDesc(ctypes.Structure):
    """MmegoGemmDesc of include/mmego_hip.h: one mmego_gemm argument set (the stream aside)."""
    _fields_ = [("A", ctypes.c_void_p), ("sam", ctypes.c_long), ("sak", ctypes.c_long),
                ("B", ctypes.c_void_p), ("sbk", ctypes.c_long), ("sbn", ctypes.c_long),
                ("C", ctypes.c_void_p), ("scm", ctypes.c_long), ("scn", ctypes.c_long),
                ("bias", ctypes.c_void_p),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("nbatch", ctypes.c_int),
                ("sAb", ctypes.c_long), ("sBb", ctypes.c_long), ("sCb", ctypes.c_long),
                ("relu", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("splitk_ws", ctypes.c_void_p), ("nsplit", ctypes.c_int),
                ("sBiasb", ctypes.c_long),
                ("cmul", ctypes.c_void_p), ("asum", ctypes.c_void_p)]


class BnRef(ctypes.Structure):
    """MmegoBnRef of include/mmego_hip.h: a BatchNorm whose batch statistics arrive as partial records."""
    _fields_ = [("rec", ctypes.c_void_p), ("nrec", ctypes.c_int), ("rows_per_rec", ctypes.c_int),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p),
                ("momentum", ctypes.c_float), ("eps", ctypes.c_float), ("state", ctypes.c_void_p)]

    @staticmethod
    def of(bn, state, rec=None, nrec=0, rows_per_rec=0):
        return BnRef(ptr(rec), int(nrec), int(rows_per_rec), ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                     float(bn.momentum), float(bn.eps), ptr(state))


class GcnFront(ctypes.Structure):
    """MmegoGcnFront of include/mmego_hip.h."""
    _fields_ = [("X1", ctypes.c_void_p), ("ld1", ctypes.c_long), ("X2", ctypes.c_void_p), ("ld2", ctypes.c_long), ("in_mode", ctypes.c_int),
                ("bn1", BnRef), ("bn2", BnRef), ("xact", ctypes.c_void_p),
                ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("cin", ctypes.c_int), ("nout", ctypes.c_int),
                ("mix", ctypes.c_int), ("K", ctypes.c_int), ("cout", ctypes.c_int), ("A", ctypes.c_void_p), ("importance", ctypes.c_void_p),
                ("Z", ctypes.c_void_p), ("ldz", ctypes.c_long), ("Y", ctypes.c_void_p), ("recY", ctypes.c_void_p), ("recR", ctypes.c_void_p),
                ("outT", ctypes.c_void_p), ("T", ctypes.c_int), ("F", ctypes.c_long), ("V", ctypes.c_int)]


class Pack(ctypes.Structure):
    """MmegoPack of include/mmego_hip.h: one weight re-layout of mmego_pack_multi."""
    _fields_ = [("W", ctypes.c_void_p), ("Wp", ctypes.c_void_p), ("Co", ctypes.c_int), ("Ci", ctypes.c_int), ("taps", ctypes.c_int),
                ("kind", ctypes.c_int)]


class DwRed(ctypes.Structure):
    """MmegoDwRed of include/mmego_hip.h: one layer's weight-gradient partials to be summed by mmego_mlp_dw_reduce_multi."""
    _fields_ = [("part", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int), ("rows", ctypes.c_long),
                ("nblk", ctypes.c_int), ("stride", ctypes.c_long)]


class Lstm64Fwd(ctypes.Structure):
    """MmegoLstm64Fwd of include/mmego_hip.h: one stack's layer for mmego_lstm64_forward_multi."""
    _P2 = ctypes.c_void_p * 2
    _fields_ = [("xproj", _P2), ("xs", ctypes.c_long), ("whh", _P2), ("bhh", _P2), ("h0", _P2), ("c0", _P2),
                ("out", ctypes.c_void_p), ("os", ctypes.c_long), ("hn", _P2), ("cn", _P2), ("gates", _P2), ("cst", _P2), ("hprev", _P2),
                ("B", ctypes.c_int), ("T", ctypes.c_int), ("drop_y", ctypes.c_void_p), ("drop_mask", ctypes.c_void_p),
                ("drop_p", ctypes.c_float), ("seed_ctr", ctypes.c_void_p), ("salt", ctypes.c_uint)]


class Lstm64Bwd(ctypes.Structure):
    """MmegoLstm64Bwd of include/mmego_hip.h."""
    _P2 = ctypes.c_void_p * 2
    _fields_ = [("dout", ctypes.c_void_p), ("dos", ctypes.c_long), ("gates", _P2), ("cst", _P2), ("c0", _P2), ("whh", _P2),
                ("dgates", _P2), ("dgs", ctypes.c_long), ("B", ctypes.c_int), ("T", ctypes.c_int)]


def pair2(a, b):
    """Two device addresses as a descriptor's pointer pair."""
    return (ctypes.c_void_p * 2)(ptr(a), ptr(b))


class Slab(ctypes.Structure):
    """MmegoSlab of include/mmego_hip.h: one deferred partial-product sum."""
    _fields_ = [("ws", ctypes.c_void_p), ("out", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("asum", ctypes.c_void_p),
                ("kind", ctypes.c_int), ("nsplit", ctypes.c_int), ("M", ctypes.c_int), ("N", ctypes.c_int), ("taps", ctypes.c_int),
                ("scm", ctypes.c_long)]


_gemm_rec = None
_gemm_rec_outs = None


class gemm_group:
    """Context: the mmego_gemm calls made inside are collected and issued as ONE mmego_gemm_group launch at exit; every other
    launch inside the context runs immediately.  Contract (checked):
      * independent products only -- a product recorded here is not executed before the context ends, so nothing inside the
        context may read OR write what a recorded product writes (its C and asum): a pass-through launch, or a later recorded
        product, one of whose pointer arguments -- a tensor (any view) or a raw integer address -- falls inside the byte extent
        of a deferred output raises;
      * the group is issued on the stream that was current at entry: a stream switch inside the context raises at exit;
      * split-K products share ops.scratch: the group entry point runs a group containing one as separate launches in recorded
        order on that one stream (mmego_gemm_group's fallback), which keeps the scratch reuse stream-ordered."""

    def __enter__(self):
        global _gemm_rec, _gemm_rec_outs
        if _gemm_rec is not None:
            raise RuntimeError("gemm_group contexts do not nest")
        _gemm_rec, _gemm_rec_outs = [], []
        self._stream = stream_handle()
        return self

    def __exit__(self, et, ev, tb):
        global _gemm_rec, _gemm_rec_outs
        rec, _gemm_rec, _gemm_rec_outs = _gemm_rec, None, None
        if et is None and rec:
            if stream_handle() != self._stream:
                raise RuntimeError("gemm_group: the current stream changed inside the context; the deferred products would be "
                                   "issued on another stream than the launches around them")
            for i in range(0, len(rec), 10):
                part = rec[i:i + 10]
                arr = (GemmDesc * len(part))()
                for dsc, a in zip(arr, part):
                    for (fname, _), v in zip(GemmDesc._fields_, a):
                        setattr(dsc, fname, _conv(v))
                call("gemm_group", len(part), arr)          # (the array object, not its address: a recorded call keeps it alive)
        return False


# (mmego_gemm's argument list behind the stream: C at 6 with strides at 7 / 8, M N K nbatch at 10..13, sCb at 16, asum at 23)


def _launch(name, *args):
    """The one place a kernel is launched: `mmego_<name>` on torch's current stream, tensors passed as device pointers.
    (plan.StepPlan swaps this function for a recorder while it records a step.)"""
    fn = getattr(lib(), "mmego_" + name)
    rc = fn(stream_handle(), *[_conv(a) for a in args])
    if rc != 0:
        raise RuntimeError("mmego_%s failed: %s" % (name, "bad argument" if rc < 0 else "hipError %d" % rc))


def _ptr_of(a):
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    if isinstance(a, int) and a >= (1 << 20):        # a raw device address (sizes, strides and flags are far below any mapping)
        return a
    return None


def _gemm_out_extents(args):
    """Byte ranges [lo, hi) a recorded mmego_gemm writes: C (M x N x nbatch through its strides) and asum (M floats per batch)."""
    out = []
    M, N, nb = int(args[10]), int(args[11]), max(1, int(args[13]))
    c = _ptr_of(args[6])
    if c is not None:
        span = (M - 1) * abs(int(args[7])) + (N - 1) * abs(int(args[8])) + 1
        for b in range(nb):              # (per batch: the batches of a pair product may lie apart, with other leaves' slots between)
            lo = c + 4 * b * int(args[16])
            out.append((lo, lo + 4 * span))
    s = _ptr_of(args[23])
    if s is not None:
        out.append((s, s + 4 * M * nb))
    return out


def call(name, *args):
    """Launch `mmego_<name>` on torch's current stream (inside a gemm_group context: defer the mmego_gemm calls)."""
    if _gemm_rec is not None and name != "gemm_group":
        for a in args:
            p = _ptr_of(a)
            if p is not None and any(lo <= p < hi for lo, hi in _gemm_rec_outs):
                raise RuntimeError("gemm_group: mmego_%s takes the output of a product that is still deferred inside this context "
                                   "(only independent leaves may be grouped)" % name)
        if name == "gemm":
            _gemm_rec_outs.extend(_gemm_out_extents(args))
            _gemm_rec.append(args)
            return
    _launch(name, *args)


def graph_dA_nblk(G):
    return lib().mmego_graph_dA_nblk(G)


def colstats_nblk(rows):
    return lib().mmego_colstats_nblk(rows)
