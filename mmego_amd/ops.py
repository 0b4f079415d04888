"""Typed host wrappers over the C ABI: shape/stride/dtype checks on the host, then one kernel launch.

Every function here runs on the GPU through libmmego_hip.so; nothing computes with torch ops.  Tensors
are fp32 CUDA(HIP) tensors; 2-D arguments may be strided views (e.g. a column slice of a wider buffer,
or ``W.t()``) -- the element strides go straight to the kernel.
"""
import os as _os

import torch

from . import hip


def _chk(t, nd=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError("expected an fp32 device tensor, got %r" % (type(t),))
    if nd is not None and t.dim() != nd:
        raise ValueError("expected %d-D tensor, got shape %s" % (nd, tuple(t.shape)))
    return t


def _rows(t):
    """2-D row view [rows, C] with unit column stride of a contiguous-by-rows tensor."""
    _chk(t, 2)
    if t.stride(1) != 1:
        raise ValueError("row tensor needs unit column stride")
    return t


class Arena:
    """Named scratch buffers that persist across calls (static addresses: HIP-graph friendly).

    A buffer is never replaced: asking for a known name with another shape (a net used at a second minibatch size -- the short
    last minibatch of an epoch, an evaluation pass) creates a second buffer beside the first.  Captured HIP graphs keep replaying
    on the addresses they were captured with, so freeing or recycling a buffer they use would let them scribble over whatever
    the allocator puts there next."""

    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, key, shape, dtype=torch.float32, zero=False):
        shape = tuple(int(s) for s in shape)
        slot = (key, shape, dtype)
        t = self.bufs.get(slot)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self.bufs[slot] = t
        if zero:
            fill(t, 0.0) if dtype == torch.float32 else t.zero_()
        return t

    def has(self, key):
        return any(k[0] == key for k in self.bufs)


_scratch = {}
_retired = []

_capture_origin = {"stream": None}


class capture:
    """`with ops.capture(graph[, stream=...])`: torch.cuda.graph plus the bookkeeping of the capture's ORIGIN stream.

    On this ROCm an event recorded on a stream that was itself forked from the capturing stream, and waited on by another
    non-origin stream, aborts the process at capture (scripts/repro_nested_capture_fork.py), so code that wants a side stream
    under capture may only fork from the origin (capture_can_fork()).  The origin is known exactly between the begin and the
    end of THIS context -- set on entry, cleared on exit, never left behind for a later capture -- and the engines only read
    it.  A capture started any other way (plain torch.cuda.graph) has no recorded origin: capture_can_fork() is False there
    and every caller takes its single-chain form."""

    def __init__(self, graph, **kw):
        self._ctx = torch.cuda.graph(graph, **kw)

    def __enter__(self):
        if _capture_origin["stream"] is not None:
            raise RuntimeError("ops.capture does not nest")
        self._ctx.__enter__()
        _capture_origin["stream"] = torch.cuda.current_stream().cuda_stream
        return self

    def __exit__(self, et, ev, tb):
        _capture_origin["stream"] = None
        return self._ctx.__exit__(et, ev, tb)


_no_fork = [0]


class no_fork:
    """`with ops.no_fork()`: every fork site of the package (blocks.lstm_recurrence's two chains, the engines' stage branches) takes
    its single-chain form -- whatever is launched inside runs as ONE dependency chain on the current stream.  The engines of
    train_step.py open it whenever a net of theirs runs bf16-MFMA kernels (precision "split3" / "bf16"): no other kernel is
    then ever resident on a CU beside a bf16-MFMA workgroup."""

    def __enter__(self):
        _no_fork[0] += 1
        return self

    def __exit__(self, et, ev, tb):
        _no_fork[0] -= 1
        return False


def capture_can_fork():
    """True when a side stream may be forked from the current stream: always when running eagerly; under graph capture only on
    the origin stream of an ops.capture context; never inside ops.no_fork()."""
    if _no_fork[0]:
        return False
    if not torch.cuda.is_current_stream_capturing():
        return True
    return _capture_origin["stream"] is not None and torch.cuda.current_stream().cuda_stream == _capture_origin["stream"]


def scratch(device, n):
    """Grow-only fp32 scratch per (device, stream) (split-K slabs, reduction partials).  Stream-ordered reuse is safe."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    t = _scratch.get(key)
    if t is None or t.numel() < n:
        if t is not None:
            _retired.append(t)          # captured graphs may still replay on the smaller buffer: never hand its memory back
        t = torch.empty(max(int(n), 1 << 20), dtype=torch.float32, device=device)
        _scratch[key] = t
    return t


def mm(A, B, C, bias=None, relu=False, accumulate=False, nsplit=1, cmul=None, asum=None, cmask=None):
    """C[M,N] (+)= A[M,K] @ B[K,N] (+bias)(relu), then (*= cmul) or (= 0 where cmask <= 0) -- cmul / cmask: a tensor with C's
    shape and strides; A, B, C are 2-D views with arbitrary strides.  asum [M] (+)= row sums of A (see asum_ok)."""
    if cmask is not None:
        if cmul is not None or relu:
            raise ValueError("mm: cmask excludes cmul and relu")
        cmul, relu = cmask, 2
    _chk(A, 2), _chk(B, 2), _chk(C, 2)
    M, K = A.shape
    K2, N = B.shape
    if K2 != K or tuple(C.shape) != (M, N):
        raise ValueError("mm shape mismatch %s @ %s -> %s" % (tuple(A.shape), tuple(B.shape), tuple(C.shape)))
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous with N elements")
    ws = None
    if nsplit > 1:
        ws = scratch(A.device, nsplit * (M * N + (M if asum is not None else 0)))
    if asum is not None and (asum.numel() != M or not asum.is_contiguous() or not asum_ok(M, N, K, nsplit, over_tile=True)):
        raise ValueError("mm: asum needs M contiguous elements and a small-product shape (ops.asum_ok)")
    if cmul is not None and (tuple(cmul.shape) != (M, N) or cmul.stride() != C.stride() or nsplit > 1):
        raise ValueError("mm: cmul needs C's shape and strides, and an unsplit product")
    hip.call("gemm", A, A.stride(0), A.stride(1), B, B.stride(0), B.stride(1), C, C.stride(0), C.stride(1), bias,
             M, N, K, 1, 0, 0, 0, int(relu), int(accumulate), ws, nsplit, 0, cmul, asum)
    return C


def bmm(A, B, C, accumulate=False, bias=None):
    """Batched C[b] (+)= A[b] @ B[b] (+ bias[n]); 3-D views, batch stride may be 0 (broadcast operand)."""
    _chk(A, 3), _chk(B, 3), _chk(C, 3)
    nb, M, K = A.shape
    if B.shape[0] != nb or C.shape[0] != nb or B.shape[1] != K or tuple(C.shape[1:]) != (M, B.shape[2]):
        raise ValueError("bmm shape mismatch")
    if bias is not None and (bias.numel() != B.shape[2] or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous with N elements")
    hip.call("gemm", A, A.stride(1), A.stride(2), B, B.stride(1), B.stride(2), C, C.stride(1), C.stride(2), bias,
             M, B.shape[2], K, nb, A.stride(0), B.stride(0), C.stride(0), 0, int(accumulate), None, 1, 0, None, None)
    return C


def mm_two(A0, B0, C0, A1, B1, C1, bias0=None, bias1=None, cmul0=None, cmul1=None):
    """C0 = A0 @ B0 (+ bias0)(* cmul0) and C1 = A1 @ B1 (+ bias1)(* cmul1) -- two INDEPENDENT products of one shape -- as ONE batched
    launch: the batch strides are the distances between the operands (any two fp32 tensors are a whole number of elements apart).  The
    multipliers must lie as far apart as the outputs (two halves of one buffer).  Falls back to two launches when the layouts differ."""
    same = (A0.shape == A1.shape and B0.shape == B1.shape and C0.shape == C1.shape and A0.stride() == A1.stride() and B0.stride() == B1.stride()
            and C0.stride() == C1.stride() and (bias0 is None) == (bias1 is None) and (cmul0 is None) == (cmul1 is None))
    dA, dB, dC = A1.data_ptr() - A0.data_ptr(), B1.data_ptr() - B0.data_ptr(), C1.data_ptr() - C0.data_ptr()
    if same and cmul0 is not None:
        same = (cmul0.shape == C0.shape and cmul1.shape == C0.shape and cmul0.stride() == C0.stride() and cmul1.stride() == C0.stride()
                and cmul1.data_ptr() - cmul0.data_ptr() == dC)
    if same and bias0 is not None:
        same = bias0.is_contiguous() and bias1.is_contiguous() and bias0.numel() == B0.shape[1] == bias1.numel()
    if not same or dA % 16 or dB % 16 or dC % 16:
        mm(A0, B0, C0, bias=bias0, cmul=cmul0)
        mm(A1, B1, C1, bias=bias1, cmul=cmul1)
        return False
    _chk(A0, 2), _chk(B0, 2), _chk(C0, 2), _chk(A1, 2), _chk(B1, 2), _chk(C1, 2)
    M, K = A0.shape
    K2, N = B0.shape
    if K2 != K or tuple(C0.shape) != (M, N):
        raise ValueError("mm_two shape mismatch")
    dbias = (bias1.data_ptr() - bias0.data_ptr()) // 4 if bias0 is not None else 0
    hip.call("gemm", A0, A0.stride(0), A0.stride(1), B0, B0.stride(0), B0.stride(1), C0, C0.stride(0), C0.stride(1), bias0,
             M, N, K, 2, dA // 4, dB // 4, dC // 4, 0, 0, None, 1, dbias, cmul0, None)
    return True


def linear_pair(x, W0, W1, b0, b1, out, ncol):
    """out[:, :ncol] = x W0^T + b0 and out[:, ncol:2 ncol] = x W1^T + b1 as ONE batched product (the two directions' input
    projections of a BiLSTM layer: 2 x 1280 tiles = exactly five rounds of the persistent GEMM grid instead of 2 x 2.5)."""
    _chk(x, 2), _chk(out, 2)
    rows, K = x.shape
    ok = (W0.shape == W1.shape == (ncol, K) and W0.is_contiguous() and W1.is_contiguous() and b0.is_contiguous() and b1.is_contiguous()
          and x.stride(1) == 1 and out.stride(1) == 1 and out.shape[1] >= 2 * ncol)
    dW, dB = (W1.data_ptr() - W0.data_ptr()) // 4, (b1.data_ptr() - b0.data_ptr()) // 4
    if not ok or (W1.data_ptr() - W0.data_ptr()) % 16 or (b1.data_ptr() - b0.data_ptr()) % 4:
        linear(x, W0, b0, out[:, :ncol])
        linear(x, W1, b1, out[:, ncol:2 * ncol])
        return out
    hip.call("gemm", x, x.stride(0), 1, W0, 1, K, out, out.stride(0), 1, b0, rows, ncol, K, 2, 0, dW, ncol, 0, 0, None, 1, dB, None, None)
    return out


def stacked(a, b):
    """One [na + nb, ...] view over ``a`` and ``b`` when ``b`` sits directly behind ``a`` in memory (two tensors laid out back to
    back in a flat parameter / gradient buffer: params.FlatParams, a module's flat_param_order), else None."""
    if (a.dtype != b.dtype or tuple(a.shape[1:]) != tuple(b.shape[1:]) or not a.is_contiguous() or not b.is_contiguous()
            or b.data_ptr() != a.data_ptr() + a.numel() * a.element_size()
            or a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()):     # (neighbours by chance in the allocator are not one buffer)
        return None
    return torch.as_strided(a.detach(), (a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), a.stride())


def linear(x, W, b, out, relu=False):
    """out[rows,N] = x[rows,K] @ W[N,K]^T + b"""
    return mm(x, W.view(W.shape[0], -1).t(), out, bias=b, relu=relu)


_SPLIT_K_MIN = 128          # k per slab at least
_SPLIT_K_FROM = 1024        # split from this K on


def pick_split(M, N, K):
    """Split-K factor of a weight-gradient product: these have few output tiles and a long K (the batch rows), and a
    workgroup's k-loop is latency-bound per 64-k step, so K is cut down to one or two steps per workgroup as long as the
    grid stays within ~2 workgroups per CU."""
    if K <= _SPLIT_K_FROM:
        return 1        # short K: the K-quartered small-product kernel does it in one launch (no slab-reduce launch behind it)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    want = max(1, 512 // tiles)
    return int(max(1, min(want, K // _SPLIT_K_MIN)))


def chain_split(M, N, K):
    """Split-K factor for a product on a serial critical path (the recurrent dh = dgates . W_hh of the LSTM backward): few
    output tiles and a long K, so the K range is cut until about two workgroups per CU are busy (>= 512 k per slab)."""
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    return int(max(1, min(512 // max(tiles, 1), K // 512)))


_KQ_MAX = 512               # gemm.hip GEMM_KQ_MAX
_PAIR_SPLIT_WGS = 512


def asum_ok(M, N, K, nsplit, nbatch=1, over_tile=False):
    """Whether mmego_gemm takes the K-quartered small-product kernel for this shape (its dispatch rule, gemm.hip), the one that
    can return the row sums of its A operand beside the product."""
    kchunk = -(-(-(-K // nsplit)) // 16) * 16
    wgs64 = ((M + 63) // 64) * ((N + 63) // 64) * nsplit * nbatch
    tile = M % 64 == 0 and N % 64 == 0 and K % 64 == 0 and (M // 64) * (N // 64) * nsplit * nbatch >= 256
    return wgs64 <= _KQ_MAX and kchunk >= 64 and (over_tile or not tile) and M * N < (1 << 16)


def grad_weight(dY, X, dW, db=None, prefer_fused=False):
    """dW[N,K] = dY[rows,N]^T @ X[rows,K]  (fixed-order split over rows); db[N] (optional) = column sums of dY, the bias
    gradient: from the same launch where the product runs on the small-product kernel, a column-sum launch otherwise."""
    W2 = dW.view(dW.shape[0], -1)
    nsplit = pick_split(W2.shape[0], W2.shape[1], X.shape[0])
    # (prefer_fused: take the small-product kernel with the bias sums even where the tile kernel would be picked for the product
    # alone -- product + slab reduce instead of product + slab reduce + two column-sum launches)
    fused = db is not None and db.is_contiguous() and asum_ok(W2.shape[0], W2.shape[1], X.shape[0], nsplit, over_tile=prefer_fused)
    mm(dY.t(), X, W2, nsplit=nsplit, asum=db if fused else None)
    if db is not None and not fused:
        colsum(dY, db)
    return dW


class SlabList:
    """The deferred partial-product sums of one backward pass: split-K weight gradients (grad_weight_deferred), the temporal
    convolution's weight-gradient slabs and the edge-importance partials are left in buffers of their own (an Arena, not the shared
    scratch, which the next product would overwrite) and added by ONE mmego_slab_reduce launch at the end of the pass -- every such
    product used to be followed by a reduce launch of its own in the backward pass's dependent chain."""

    def __init__(self):
        self.items = []

    def add(self, ws, out, kind, nsplit, M, N, scm=0, taps=1, scale=None, asum=None):
        self.items.append(hip.Slab(hip.ptr(ws), hip.ptr(out), hip.ptr(scale), hip.ptr(asum), int(kind), int(nsplit), int(M), int(N), int(taps),
                                   int(scm)))

    def run(self):
        for i in range(0, len(self.items), 24):
            part = self.items[i:i + 24]
            arr = (hip.Slab * len(part))(*part)
            hip.call("slab_reduce", len(part), arr)
        self.items = []


def grad_weight_deferred(dY, X, dW, slabs, arena, key, db=None):
    """grad_weight whose split-K sum is left to ``slabs`` (a SlabList the caller runs at the end of its backward pass); db: the
    bias gradient rides along as the product's slab row sums where the small-product kernel takes the shape."""
    W2 = dW.view(dW.shape[0], -1)
    M, N = W2.shape
    K = X.shape[0]
    nsplit = pick_split(M, N, K)
    fused = db is not None and db.is_contiguous() and asum_ok(M, N, K, nsplit, over_tile=True)
    if nsplit == 1:
        mm(dY.t(), X, W2, asum=db if fused else None)
    else:
        _chk(dY, 2), _chk(X, 2)
        if dY.shape[0] != K or dY.shape[1] != M or X.shape[1] != N or not W2.is_contiguous():
            raise ValueError("grad_weight_deferred shape mismatch")
        ws = arena.get(key, (nsplit * (M * N + (M if fused else 0)),))
        hip.call("gemm", dY, dY.stride(1), dY.stride(0), X, X.stride(0), X.stride(1), W2, N, 1, None, M, N, K, 1, 0, 0, 0, 0, 2, ws, nsplit, 0,
                 None, db if fused else None)
        slabs.add(ws, W2, 0, nsplit, M, N, scm=N, asum=db if fused else None)
    if db is not None and not fused:
        colsum(dY, db)
    return dW


def grad_weight_pair(dY, ncol, X, dW0, dW1, X1=None, db=None):
    """dW0 = dY[:, :ncol]^T @ X and dW1 = dY[:, ncol:2 ncol]^T @ (X1 or X) as ONE batched product (the two directions' weight
    gradients of a BiLSTM layer; their slots in the flat gradient buffer are a fixed distance apart).  Falls back to two
    products when the layout does not allow it.  db [2 ncol] (optional, contiguous): the column sums of dY (both directions'
    bias gradients) from the same launch; -> True when db was written."""
    rows, K = X.shape
    Xb = X if X1 is None else X1
    W0, W1 = dW0.view(dW0.shape[0], -1), dW1.view(dW1.shape[0], -1)
    dist, xdist = W1.data_ptr() - W0.data_ptr(), Xb.data_ptr() - X.data_ptr()
    ok = (W0.shape == W1.shape == (ncol, K) and W0.is_contiguous() and W1.is_contiguous() and dY.stride(1) == 1 and X.stride(1) == 1
          and Xb.shape == X.shape and Xb.stride() == X.stride() and dist % 16 == 0 and xdist % 16 == 0)
    if ok and pick_split(ncol, K, rows) > 1:
        # long K (stage-1 IMU_Net: 10 240 rows): the two directions as one batched product keep twice the tiles in flight, so the
        # K range is cut half as often as for one direction alone (or not at all) -- one launch + at most one slab reduce instead
        # of two of each
        tiles = ((ncol + 63) // 64) * ((K + 63) // 64) * 2
        nsplit = int(max(1, min(_PAIR_SPLIT_WGS // tiles, rows // _SPLIT_K_MIN)))
        ws = scratch(dY.device, nsplit * 2 * ncol * K) if nsplit > 1 else None
        hip.call("gemm", dY, 1, dY.stride(0), X, X.stride(0), 1, W0, K, 1, None, ncol, K, rows, 2, ncol, xdist // 4, dist // 4, 0, 0, ws,
                 nsplit, 0, None, None)
        return False
    if not ok:
        grad_weight(dY[:, :ncol], X, dW0)
        grad_weight(dY[:, ncol:2 * ncol], Xb, dW1)
        return False
    fused = db is not None and db.is_contiguous() and db.numel() == 2 * ncol and asum_ok(ncol, K, rows, 1, nbatch=2)
    hip.call("gemm", dY, 1, dY.stride(0), X, X.stride(0), 1, W0, K, 1, None, ncol, K, rows, 2, ncol, xdist // 4, dist // 4, 0, 0, None, 1, 0,
             None, db if fused else None)
    return fused


def grad_input(dY, W, dX, accumulate=False, cmul=None, cmask=None):
    """dX[rows,K] (+)= dY[rows,N] @ W[N,K], then (*= cmul) or (= 0 where cmask <= 0: ReLU backward, cmask = the forward output)"""
    return mm(dY, W.view(W.shape[0], -1), dX, accumulate=accumulate, cmul=cmul, cmask=cmask)


def grad_input_slabs(arena, key, dY, W, dX):
    """dX[rows, In] = dY[rows, K] @ W[K, In] for outputs that are FEWER than 256 tiles of 320 x 256 (stage-1 IMU_Net training: 10 240 x
    {512, 1024} over K = 4096): against W^T both operands are k-contiguous, and with K cut into 256 / tiles slabs the product runs on
    gemm_tile_big_kernel, one workgroup per CU (r06: 0.85 instead of 0.52-0.63 of the fp32 peak on the 128 x 128 walk); the slabs are
    added in order by a streaming sum.  -> False (nothing done) where the shape does not fit."""
    rows, K = dY.shape
    W2 = W.view(W.shape[0], -1)
    In = W2.shape[1]
    if (rows % 320 or In % 256 or K % 64 or W2.shape[0] != K or dY.stride(1) != 1 or not W2.is_contiguous() or not dX.is_contiguous()
            or tuple(dX.shape) != (rows, In)):
        return False
    tiles = (rows // 320) * (In // 256)
    ns = 256 // tiles if tiles < 256 else 0
    if ns < 2 or tiles * ns != 256 or K // ns < 512:
        return False
    kchunk = -(-(-(-K // ns)) // 64) * 64
    if (ns - 1) * kchunk >= K or K - (ns - 1) * kchunk < 64:
        return False
    WT = arena.get("%s.wT" % key, (In, K))
    hip.call("transpose_batched", W2, WT, 1, K, In)
    ws = arena.get("%s.ws" % key, (ns * rows * In,))
    hip.call("gemm", dY, dY.stride(0), 1, WT, 1, K, dX, In, 1, None, rows, In, K, 1, 0, 0, 0, 0, 2, ws, ns, 0, None, None)
    hip.call("split3_slab_sum", ws, ns, rows * In, dX)
    return True


def colsum(X, out, accumulate=False, out2=None, scale=None):
    """out[c] (+)= sum_r X[r, c] (* scale[c], for up to 1024 rows); out2 (optional) gets a copy of the result."""
    X = _rows(X)
    rows, C = X.shape
    ws = scratch(X.device, C * hip.colstats_nblk(rows))
    hip.call("colsum", X, X.stride(0), rows, C, ws, out, out2, int(accumulate), scale)
    return out


def fill(t, v):
    if not t.is_contiguous():
        raise ValueError("fill needs a contiguous tensor")
    hip.call("fill", t, t.numel(), float(v))
    return t


def copy2d(X, Y, accumulate=False):
    X, Y = _rows(X), _rows(Y)
    if X.shape != Y.shape:
        raise ValueError("copy2d shape mismatch")
    hip.call("copy2d", X, X.stride(0), Y, Y.stride(0), X.shape[0], X.shape[1], int(accumulate))
    return Y


def gather_rows(X, idx, Y):
    """Y[i, :] = X[idx[i], :] for contiguous fp32 X [n, W], int64 idx [m], Y [m, W]."""
    _chk(X, 2), _chk(Y, 2)
    if not (X.is_contiguous() and Y.is_contiguous() and idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()):
        raise ValueError("gather_rows needs contiguous fp32 X/Y and a contiguous int64 device index")
    if Y.shape != (idx.numel(), X.shape[1]):
        raise ValueError("gather_rows shape mismatch")
    hip.call("gather_rows", X, X.shape[0], X.shape[1], idx, idx.numel(), Y)
    return Y


def relu_mask_(G, H):
    G, H = _rows(G), _rows(H)
    hip.call("relu_mask", G, G.stride(0), H, H.stride(0), G.shape[0], G.shape[1])
    return G


class BnState:
    """mean / invstd / a / b vectors of one BatchNorm application (kept for backward)."""

    def __init__(self, arena, key, C):
        v = arena.get(key, (4, C))
        self.all = v
        self.mean, self.invstd, self.a, self.b = v[0], v[1], v[2], v[3]
        self.C = C


def bn_stats(arena, key, X, bn, training):
    """Compute the affine of a BatchNorm over the rows of X (train: batch stats + running update)."""
    X = _rows(X)
    rows, C = X.shape
    st = BnState(arena, key, C)
    if training:
        ws = scratch(X.device, 3 * C * hip.colstats_nblk(rows))
        hip.call("bn_train_stats", X, X.stride(0), rows, C, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                 float(bn.momentum), float(bn.eps), ws, st.mean, st.invstd, st.a, st.b)
        # num_batches_tracked is bumped once per forward for the whole net (FlatParams.tick_args)
    else:
        hip.call("bn_eval_affine", C, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                 st.mean, st.invstd, st.a, st.b)
    return st


def bn_stats_pair(arena, key1, X1, bn1, key2, X2, bn2, training):
    """bn_stats of two same-shape tensors; in training one pair of launches instead of two (same results)."""
    X1, X2 = _rows(X1), _rows(X2)
    rows, C = X1.shape
    tc = 8
    while tc < C and tc < 64:
        tc <<= 1
    if not training or X2.shape != X1.shape or C % tc:
        return bn_stats(arena, key1, X1, bn1, training), bn_stats(arena, key2, X2, bn2, training)
    st1, st2 = BnState(arena, key1, C), BnState(arena, key2, C)
    ws = scratch(X1.device, 6 * C * hip.colstats_nblk(rows))
    hip.call("bn_train_stats_pair", rows, C, ws,
             X1, X1.stride(0), bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, float(bn1.momentum), float(bn1.eps),
             st1.mean, st1.invstd, st1.a, st1.b,
             X2, X2.stride(0), bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, float(bn2.momentum), float(bn2.eps),
             st2.mean, st2.invstd, st2.a, st2.b)
    return st1, st2


def affine_act(X, st, Y, relu=True, X2=None, st2=None):
    X, Y = _rows(X), _rows(Y)
    rows, C = X.shape
    if X2 is not None:
        X2 = _rows(X2)
        hip.call("affine_act", X, X.stride(0), st.mean, st.a, st.b, X2, X2.stride(0), st2.mean, st2.a, st2.b, Y,
                 Y.stride(0), rows, C, int(relu))
    else:
        hip.call("affine_act", X, X.stride(0), st.mean, st.a, st.b, None, 0, None, None, None, Y, Y.stride(0), rows, C,
                 int(relu))
    return Y


def bn_backward(dY, Ymask, X, st, dgamma, dbeta, dX):
    """BatchNorm(train) backward through an optional ReLU (mask = Ymask > 0)."""
    dY, X, dX = _rows(dY), _rows(X), _rows(dX)
    rows, C = X.shape
    ws = scratch(X.device, 2 * C * hip.colstats_nblk(rows) + 2 * C)
    c12 = ws[2 * C * hip.colstats_nblk(rows):]
    if Ymask is not None:
        Ymask = _rows(Ymask)
        hip.call("bn_backward", dY, dY.stride(0), Ymask, Ymask.stride(0), X, X.stride(0), st.mean, st.invstd, st.a, rows,
                 C, ws, c12, dgamma, dbeta, dX, dX.stride(0))
    else:
        hip.call("bn_backward", dY, dY.stride(0), None, 0, X, X.stride(0), st.mean, st.invstd, st.a, rows, C, ws, c12,
                 dgamma, dbeta, dX, dX.stride(0))
    return dX


def bn_backward_pair(dY, Ymask, X1, st1, dgamma1, dbeta1, dX1, X2, st2, dgamma2, dbeta2, dX2):
    """Two BatchNorm(train) backwards that share dY and the ReLU mask, in the launches of one (same bits as two bn_backward calls)."""
    dY, Ymask, X1, dX1, X2, dX2 = (_rows(t) for t in (dY, Ymask, X1, dX1, X2, dX2))
    rows, C = X1.shape
    tc = 8
    while tc < C and tc < 64:
        tc <<= 1
    if X2.shape != X1.shape or C % tc:
        bn_backward(dY, Ymask, X1, st1, dgamma1, dbeta1, dX1)
        bn_backward(dY, Ymask, X2, st2, dgamma2, dbeta2, dX2)
        return
    nblk = hip.colstats_nblk(rows)
    ws = scratch(X1.device, 4 * C * nblk + 4 * C)
    hip.call("bn_backward_pair", dY, dY.stride(0), Ymask, Ymask.stride(0), rows, C, ws, ws[4 * C * nblk:],
             X1, X1.stride(0), st1.mean, st1.invstd, st1.a, dgamma1, dbeta1, dX1, dX1.stride(0),
             X2, X2.stride(0), st2.mean, st2.invstd, st2.a, dgamma2, dbeta2, dX2, dX2.stride(0))


def transform2h_(pts, R, t, src=None, keep=None, feats=None, nfeat=0):
    """In place on pts [F, P, C] (or any contiguous view of it).  src (optional): the points are read from there -- [F, P, 3]
    (xyz only) or a tensor of pts' shape (whole rows: the other channels are copied along).  keep (optional, pts' shape) and
    feats (optional 2-D [F*P, >= nfeat] view with unit column stride: its first nfeat columns) receive the transformed rows too."""
    _chk(pts)
    if not pts.is_contiguous():
        raise ValueError("transform2h_ needs a contiguous point tensor")
    F = R.numel() // 9
    C = pts.shape[-1]
    P = pts.numel() // (F * C)
    src_ld = 0
    if src is not None:
        if not (src.is_contiguous() and src.dtype == torch.float32):
            raise ValueError("transform2h_: src must be a contiguous fp32 tensor")
        if src.numel() * C == 3 * pts.numel():
            src_ld = 3
        elif src.numel() == pts.numel():
            src_ld = C
        else:
            raise ValueError("transform2h_: src must hold one xyz row or one whole row per point")
    if keep is not None and not (keep.is_contiguous() and keep.dtype == torch.float32 and keep.numel() == pts.numel()):
        raise ValueError("transform2h_: keep must be a contiguous fp32 tensor of pts' size")
    ldf = 0
    if feats is not None:
        if not (feats.dim() == 2 and feats.stride(1) == 1 and feats.shape[0] * C == pts.numel() and 1 <= nfeat <= min(C, feats.shape[1])):
            raise ValueError("transform2h_: feats must be a [points, >= nfeat] view with unit column stride")
        ldf = feats.stride(0)
    if not (R.is_contiguous() and t.is_contiguous() and t.numel() == 3 * F):
        raise ValueError("R/t must be contiguous [F,3,3]/[F,3]")
    hip.call("transform2h", pts, F, P, C, R, t, src, src_ld, keep, feats, ldf, nfeat)
    return pts


def rotate_points(inp, out, R, t=None, transpose=True):
    F = R.numel() // 9
    P = inp.numel() // (3 * F)
    if not (inp.is_contiguous() and out.is_contiguous() and R.is_contiguous()):
        raise ValueError("rotate_points needs contiguous tensors")
    hip.call("rotate_points", inp, out, F, P, R, t, int(transpose), int(t is not None))
    return out
