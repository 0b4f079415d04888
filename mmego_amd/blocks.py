"""Forward/backward building blocks of the nets, as sequences of HIP kernel launches.

Each block keeps what its backward needs in the per-net Arena under a string key.  `G(p)` maps a
parameter to its slot in the flat gradient buffer.  No torch compute ops are used here.
"""
import os

import torch
import torch.nn as nn

from . import hip, ops


class LstmParams(nn.Module):
    """Parameter container with nn.LSTM's names, shapes, registration order and init (uniform +-1/sqrt(H)),
    so state_dicts and seeded initial weights are interchangeable with torch's nn.LSTM
    (reference Net/Upper_Net.py:333, Net/Lower_Net.py:91, Net/IMU_Net.py:58-62).  Gate order i,f,g,o."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.0, bidirectional=True):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.dropout = float(dropout)
        self.num_directions = 2 if bidirectional else 1
        H = hidden_size
        for layer in range(num_layers):
            for d in range(self.num_directions):
                sfx = "_reverse" if d == 1 else ""
                In = input_size if layer == 0 else H * self.num_directions
                self.register_parameter("weight_ih_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H, In)))
                self.register_parameter("weight_hh_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H, H)))
                self.register_parameter("bias_ih_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H)))
                self.register_parameter("bias_hh_l%d%s" % (layer, sfx), nn.Parameter(torch.empty(4 * H)))
        stdv = 1.0 / (H ** 0.5) if H > 0 else 0.0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)

    def w(self, kind, layer, d):
        return getattr(self, "%s_l%d%s" % (kind, layer, "_reverse" if d == 1 else ""))


# ---------------------------------------------------------------------------------------------------
# three (k=1 conv, BatchNorm, ReLU) stages over rows
# ---------------------------------------------------------------------------------------------------
def _mlp3_layers(mod):
    return ((mod.conv1, mod.cb1), (mod.conv2, mod.cb2), (mod.conv3, mod.cb3))


def _mlp3_fused_train(mod, x):
    dims = [mod.conv1.weight.numel() // mod.conv1.weight.shape[0]] + [c.weight.shape[0] for c in (mod.conv1, mod.conv2, mod.conv3)]
    return max(dims) <= 64 and x.stride(1) == 1


def _mlp3_forward_fused(ar, key, mod, x, out_last, pool=None):
    """Training forward of the three stages as 3 + 1 launches (mlp_train.hip): each product applies the PREVIOUS stage's
    BatchNorm + ReLU while loading its operand and leaves its own column statistics as per-workgroup partials; only the
    pre-BN tensors z1, z2, z3 and the last stage's output exist in memory."""
    rows = x.shape[0]
    nblk = hip.lib().mmego_mlp_train_nblk(rows)
    cur, part, bn_prev, st_prev = x, None, None, None
    for i, (conv, bn) in enumerate(_mlp3_layers(mod), 1):
        C = conv.weight.shape[0]
        K = conv.weight.numel() // C
        z = ar.get("%s.z%d" % (key, i), (rows, C))
        part_i = ar.get("%s.sp%d" % (key, i), (nblk * 2 * 64,), dtype=torch.float64)
        if part is None:
            hip.call("mlp_fwd_layer", cur, cur.stride(0), rows, K, None, None, None, 0.0, None, None, 0.0, None,
                     conv.weight, conv.bias, C, z, z.stride(0), part_i)
        else:
            hip.call("mlp_fwd_layer", cur, cur.stride(0), rows, K, part, bn_prev.weight, bn_prev.bias, float(bn_prev.eps),
                     bn_prev.running_mean, bn_prev.running_var, float(bn_prev.momentum), st_prev.all,
                     conv.weight, conv.bias, C, z, z.stride(0), part_i)
        cur, part, bn_prev, st_prev = z, part_i, bn, ops.BnState(ar, "%s.bn%d" % (key, i), C)
    if pool is not None:
        # the last stage's BatchNorm + ReLU inside the frame-wise softmax pooling behind it (local.hip pool128_bn_act): the activated
        # tensor is not stored
        lin, vec, attn = pool
        hip.call("pool128_bn_act", cur, cur.stride(0), rows, part, bn_prev.weight, bn_prev.bias, float(bn_prev.eps), bn_prev.running_mean,
                 bn_prev.running_var, float(bn_prev.momentum), st_prev.all, lin.weight, lin.bias, vec, attn)
        return None
    hip.call("mlp_bn_act", cur, cur.stride(0), rows, cur.shape[1], part, bn_prev.weight, bn_prev.bias, float(bn_prev.eps),
             bn_prev.running_mean, bn_prev.running_var, float(bn_prev.momentum), st_prev.all, out_last, out_last.stride(0))
    return out_last


def _mlp3_backward_fused(ar, key, mod, x, dy3, G, need_dx, have_sums=False, gather=None):
    """Backward of the three stages as 1 + 3 + 1 launches: per stage ONE pass computes dz, dX (the next stage's dy), the
    per-workgroup dW partial and the BatchNorm sums of the stage below; a last launch sums the dW partials of all three.
    have_sums: the producer of dy3 has left the last stage's (sum g, sum g xhat) partials in "<key>.gp3" already (mmego_pool8_backward).
    gather = (rows, gidx, feats, ldf, anchors, N, D): the first stage's input is not in memory (x is None) but gathered on the fly."""
    rows = x.shape[0] if gather is None else gather[0]
    nblk = hip.lib().mmego_mlp_train_nblk(rows)
    layers = _mlp3_layers(mod)
    dims = [layers[0][0].weight.numel() // layers[0][0].weight.shape[0]] + [c.weight.shape[0] for c, _ in layers]
    zs = [None] + [ar.get("%s.z%d" % (key, i), (rows, dims[i])) for i in (1, 2, 3)]
    sts = [None] + [ops.BnState(ar, "%s.bn%d" % (key, i), dims[i]) for i in (1, 2, 3)]
    gp = [None] + [ar.get("%s.gp%d" % (key, i), (nblk * 2 * 64,), dtype=torch.float64) for i in (1, 2, 3)]
    dwp = [None] + [ar.get("%s.dwp%d" % (key, i), (nblk * 4096,)) for i in (1, 2, 3)]
    if not have_sums:
        hip.call("mlp_bn_bwd_reduce", dy3, dy3.stride(0), zs[3], zs[3].stride(0), rows, dims[3], sts[3].all, gp[3])
    dy = dy3
    for i in (3, 2, 1):
        conv, bn = layers[i - 1]
        xin = zs[i - 1] if i > 1 else x
        want_dx = i > 1 or need_dx
        dprev = ar.get("%s.dy%d" % (key, i - 1), (rows, dims[i - 1])) if want_dx else None
        if i == 1 and gather is not None:
            _, gidx, feats, ldf, anchors, Npts, D = gather
            hip.call("mlp_bwd_layer_gather", dy, dy.stride(0), zs[i], zs[i].stride(0), rows, dims[i], sts[i].all, gp[i], G(bn.weight), G(bn.bias),
                     gidx, feats, ldf, anchors, Npts, D, conv.weight, dprev, dprev.stride(0) if want_dx else 0, dwp[i])
        else:
            hip.call("mlp_bwd_layer", dy, dy.stride(0), zs[i], zs[i].stride(0), rows, dims[i], sts[i].all, gp[i], G(bn.weight), G(bn.bias),
                     xin, xin.stride(0), dims[i - 1], sts[i - 1].all if i > 1 else None, conv.weight, dprev,
                     dprev.stride(0) if want_dx else 0, gp[i - 1] if i > 1 else None, dwp[i])
        dy = dprev
    descs = [hip.DwRed(hip.ptr(dwp[i]), hip.ptr(G(layers[i - 1][0].weight)), dims[i], dims[i - 1], rows, 0, 0) for i in (1, 2, 3)]
    if _dw_group is not None:
        _dw_group.extend(descs)                              # summed by ONE launch at the end of the backward pass (dw_reduce_group)
        _dw_keep.extend(dwp[1:])
    else:
        hip.call("mlp_dw_reduce", rows, 3, dwp[1], G(layers[0][0].weight), dims[1], dims[0], dwp[2], G(layers[1][0].weight), dims[2],
                 dims[1], dwp[3], G(layers[2][0].weight), dims[3], dims[2])
    return dy if need_dx else None


_dw_group = None
_dw_keep = []


class dw_reduce_group:
    """Context around a backward pass with several fused pointwise-MLP chains (Upper_Net: GlobalPointNet + PointNet; UpperNetwlocal: three):
    their weight-gradient partials are summed by ONE mmego_mlp_dw_reduce_multi launch when the context ends instead of one launch per
    chain (same sums, same order: the kernel works layer by layer)."""

    def __enter__(self):
        global _dw_group
        if _dw_group is not None:
            raise RuntimeError("dw_reduce_group contexts do not nest")
        _dw_group = []
        return self

    def __exit__(self, et, ev, tb):
        global _dw_group
        descs, _dw_group = _dw_group, None
        try:
            if et is None:
                for i in range(0, len(descs), 9):
                    part = descs[i:i + 9]
                    hip.call("mlp_dw_reduce_multi", len(part), (hip.DwRed * len(part))(*part))
        finally:
            del _dw_keep[:]
        return False


_POOL_FUSED = True        # (False: the pooling as its own launches -- what other shapes take; tests compare the two)


def pool128_fusable(mod, x, P, training):
    """GlobalPointNet's pooling fused into the train-mode chain's last stage (and its backward): 128 points per frame, 64 channels,
    the fused PointNet kernels in use, and a row count mlp_train's partition cuts into whole frame pairs."""
    return bool(_POOL_FUSED and training and P == 128 and mod.conv3.weight.shape[0] == 64 and _mlp3_fused_train(mod, x)
                and hip.lib().mmego_pool128_ok(x.shape[0]))


def mlp3_forward(ar, key, mod, x, out_last, training, pool=None, bf16=False, front=None):
    """pool = (attention Linear, vec, attn) with pool128_fusable(...): the softmax pooling behind the chain runs inside its last launch
    (out_last is not written); returns None then.
    front (optional; [rows, n <= 4] columns that end where out_last's begin, same row stride): receives x[:, :n] -- Lower_Net's
    cat(xyz, features) rows.  The one-kernel eval path writes them in the same launch as the features (whole lines for the L2 instead
    of 12 bytes per 512-byte row from another launch); the other paths copy."""
    rows = x.shape[0]
    cur = x

    def copy_front():
        if front is not None:
            ops.copy2d(x[:, :front.shape[1]], front)
    if training and _mlp3_fused_train(mod, x) and out_last.stride(1) == 1:
        copy_front()
        return _mlp3_forward_fused(ar, key, mod, x, out_last, pool=pool)
    if pool is not None:
        raise ValueError("mlp3_forward: pool needs the fused train-mode chain (check pool128_fusable)")
    if not training:
        # eval: BatchNorm folded into the convs (bn_fold_linear), then the three stages in ONE kernel whose intermediates
        # stay in LDS (mlp3.hip) -- no pre-BN tensors, no per-point 32/48/64-channel activations in HBM
        layers = _mlp3_layers(mod)
        dims = [conv.weight.shape[0] for conv, _ in layers]
        Cin = layers[0][0].weight.numel() // dims[0]
        if (Cin <= 32 and dims[0] <= 32 and dims[1] <= 64 and dims[2] <= 64 and x.stride(1) == 1
                and out_last.stride(1) == 1 and len({bn.eps for _, bn in layers}) == 1):
            # BatchNorm folded by the kernel itself while it stages the weights: no fold launches
            bnp = torch.tensor([t.data_ptr() for _, bn in layers for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)],
                               dtype=torch.int64)
            wb = [v for (conv, _), C in zip(layers, dims) for v in (conv.weight, conv.bias, C)]
            pre = 0
            if front is not None:
                pre = front.shape[1]
                if not (front.stride(0) == out_last.stride(0) and front.stride(1) == 1 and pre <= min(4, Cin)
                        and front.data_ptr() + 4 * pre == out_last.data_ptr()):
                    raise ValueError("mlp3_forward: front must be the columns right in front of out_last")
            hip.call("mlp3_eval_bf16" if bf16 else "mlp3_eval", x, x.stride(0), rows, Cin, *wb, out_last, out_last.stride(0), bnp,
                     float(layers[0][1].eps), pre)
            return out_last
        copy_front()
        folded = []
        for i, (conv, bn) in enumerate(layers, 1):
            C = conv.weight.shape[0]
            K = conv.weight.numel() // C
            wf, bf = ar.get("%s.wf%d" % (key, i), (C, K)), ar.get("%s.bf%d" % (key, i), (C,))
            hip.call("bn_fold_linear", conv.weight, conv.bias, C, K, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                     float(bn.eps), wf, bf)
            folded += [wf, bf, C]
        for i in range(3):                                  # wider layers: one product per stage, bias + ReLU in its epilogue
            wf, bf, C = folded[3 * i:3 * i + 3]
            y = out_last if i == 2 else ar.get("%s.y%d" % (key, i + 1), (rows, C))
            ops.linear(cur, wf, bf, y, relu=True)
            cur = y
        return cur
    copy_front()
    for i, (conv, bn) in enumerate(_mlp3_layers(mod), 1):
        C = conv.weight.shape[0]
        z = ar.get("%s.z%d" % (key, i), (rows, C))
        ops.linear(cur, conv.weight, conv.bias, z)
        st = ops.bn_stats(ar, "%s.bn%d" % (key, i), z, bn, training)
        y = out_last if i == 3 else ar.get("%s.y%d" % (key, i), (rows, C))
        ops.affine_act(z, st, y, relu=True)
        cur = y
    return cur


def mlp3_backward(ar, key, mod, x, y3, dy3, G, need_dx):
    if _mlp3_fused_train(mod, x) and dy3.stride(1) == 1:
        return _mlp3_backward_fused(ar, key, mod, x, dy3, G, need_dx)
    rows = x.shape[0]
    layers = _mlp3_layers(mod)
    dy = dy3
    for i in (3, 2, 1):
        conv, bn = layers[i - 1]
        C = conv.weight.shape[0]
        z = ar.get("%s.z%d" % (key, i), (rows, C))
        y = y3 if i == 3 else ar.get("%s.y%d" % (key, i), (rows, C))
        st = ops.BnState(ar, "%s.bn%d" % (key, i), C)
        dz = ar.get("%s.dz%d" % (key, i), (rows, C))
        ops.bn_backward(dy, y, z, st, G(bn.weight), G(bn.bias), dz)
        inp = x if i == 1 else ar.get("%s.y%d" % (key, i - 1), (rows, layers[i - 2][0].weight.shape[0]))
        ops.grad_weight(dz, inp, G(conv.weight))
        # (no bias gradient: a bias in front of a batch-statistics BatchNorm has EXACTLY zero gradient -- the column sums of
        #  dz vanish identically -- so its slot in the flat gradient buffer simply stays 0; the reference's autograd produces
        #  rounding noise there, which cannot change any output)
        if i > 1 or need_dx:
            dprev = ar.get("%s.dy%d" % (key, i - 1), (rows, inp.shape[1]))
            ops.grad_input(dz, conv.weight, dprev)
            dy = dprev
    return dy if need_dx else None


# ---------------------------------------------------------------------------------------------------
# Linear (+ReLU) with backward
# ---------------------------------------------------------------------------------------------------
def linear_backward(dy, x, lin, G, dx=None, accumulate_dx=False, bias_grad=True, relu_input=False, leaves=None):
    """Gradients of y = x W^T + b: writes G(W), G(b); fills dx if given.  bias_grad=False: the layer feeds a batch-statistics
    BatchNorm directly, so the bias gradient is identically zero and its slot in the gradient buffer stays 0.
    relu_input: x is the output of a ReLU whose backward is applied to dx right here (dx = 0 where x <= 0), in the epilogue of
    the product that computes dx.  leaves (optional list): the weight / bias gradient -- a leaf of the backward pass -- is not
    computed here but appended as a closure, for the caller to run later (run_leaves; dy and x must stay untouched until then)."""
    def weight_grad():
        ops.grad_weight(dy, x, G(lin.weight), db=G(lin.bias) if lin.bias is not None and bias_grad else None)
    if leaves is None:
        weight_grad()
    else:
        leaves.append(weight_grad)
    if dx is not None:
        masked = relu_input and x.shape == dx.shape and x.stride() == dx.stride()
        ops.grad_input(dy, lin.weight, dx, accumulate=accumulate_dx, cmask=x if masked else None)
        if relu_input and not masked:
            ops.relu_mask_(dx, x)
    return dx


# ---------------------------------------------------------------------------------------------------
# LocalVoxelNet's training step on its own kernels (vox.hip)
# ---------------------------------------------------------------------------------------------------
def vox_fusable(mod, x):
    """Conv3d(64, 96, k = 3 over the whole 3x3x3 grid) -> 1x1x1 convs 96 -> 128 -> 64 with train-mode BatchNorm + ReLU (Net/Upper_Net.py:
    180-205) over x [rows, 1728]: the shapes mmego_vox_* are built for."""
    layers = _mlp3_layers(mod)
    C = [conv.weight.shape[0] for conv, _ in layers]
    K = layers[0][0].weight.numel() // C[0]
    ok = (x.dim() == 2 and x.shape[1] == K and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
          and layers[1][0].weight.numel() == C[1] * C[0] and layers[2][0].weight.numel() == C[2] * C[1]
          and all(conv.weight.is_contiguous() and conv.bias is not None for conv, _ in layers))
    return bool(ok and hip.lib().mmego_vox_ok(x.shape[0], K, C[0], C[1], C[2]))


def _vox_buffers(ar, key, mod, rows):
    layers = _mlp3_layers(mod)
    C = [conv.weight.shape[0] for conv, _ in layers]
    nrt = (rows + 15) // 16
    z = [ar.get("%s.z%d" % (key, i + 1), (rows, C[i])) for i in range(3)]
    y = [ar.get("%s.y%d" % (key, i + 1), (rows, C[i])) for i in range(2)]
    rec = [ar.get("%s.rec%d" % (key, i + 1), (nrt, C[i], 2)) for i in range(3)]
    st = [ops.BnState(ar, "%s.bn%d" % (key, i + 1), C[i]) for i in range(3)]
    return layers, C, z, y, rec, st


def voxel_forward(ar, key, mod, x, out):
    """Train-mode LocalVoxelNet forward, x [rows, 1728] -> out [rows, 64]: 4 launches (a launch boundary only where a BatchNorm needs
    every row's statistics; vox.hip).  Keeps z1..z3, y1, y2 and the BatchNorm states for voxel_backward."""
    rows, K = x.shape
    layers, C, z, y, rec, st = _vox_buffers(ar, key, mod, rows)
    (c1, b1), (c2, b2), (c3, b3) = layers
    hip.call("vox_l1_fwd", x, x.stride(0), rows, K, c1.weight, c1.bias, C[0], z[0], rec[0])
    hip.call("vox_mid_fwd", z[0], rec[0], hip.BnRef.of(b1, st[0].all), rows, C[0], y[0], c2.weight, c2.bias, C[1], z[1], rec[1])
    hip.call("vox_mid_fwd", z[1], rec[1], hip.BnRef.of(b2, st[1].all), rows, C[1], y[1], c3.weight, c3.bias, C[2], z[2], rec[2])
    hip.call("vox_out_fwd", z[2], rec[2], hip.BnRef.of(b3, st[2].all), rows, C[2], out, out.stride(0))
    return out


def voxel_backward(ar, key, mod, x, out, dout, G):
    """Backward of voxel_forward: 5 launches; -> dx [rows, 1728].  Weight and BatchNorm parameter gradients are assigned; the conv biases
    get none (exactly zero in front of a batch-statistics BatchNorm, as in mlp3_backward)."""
    rows, K = x.shape
    layers, C, z, y, rec, st = _vox_buffers(ar, key, mod, rows)
    (c1, b1), (c2, b2), (c3, b3) = layers
    nrt = (rows + 15) // 16
    g = [ar.get("%s.g%d" % (key, i + 1), (rows, C[i])) for i in range(3)]
    dz = [ar.get("%s.dz%d" % (key, i + 1), (rows, C[i])) for i in range(3)]
    prt = [ar.get("%s.prt%d" % (key, i + 1), (nrt, C[i], 2)) for i in range(3)]
    dx = ar.get("%s.dx" % key, (rows, K))
    hip.call("vox_bwd_sums", dout, dout.stride(0), out, out.stride(0), z[2], st[2].all, rows, C[2], g[2], prt[2])
    hip.call("vox_mid_bwd", g[2], z[2], st[2].all, prt[2], rows, C[2], dz[2], G(b3.weight), G(b3.bias), c3.weight, C[1], y[1], z[1],
             st[1].all, g[1], prt[1])
    hip.call("vox_mid_bwd", g[1], z[1], st[1].all, prt[1], rows, C[1], dz[1], G(b2.weight), G(b2.bias), c2.weight, C[0], y[0], z[0],
             st[0].all, g[0], prt[0])
    hip.call("vox_l1_bwd", g[0], z[0], st[0].all, prt[0], rows, C[0], dz[0], G(b1.weight), G(b1.bias), c1.weight, K, dx, K)
    hip.call("vox_dw", rows, dz[0], x, x.stride(0), G(c1.weight), C[0], K, dz[1], y[0], G(c2.weight), C[1], dz[2], y[1], G(c3.weight), C[2])
    return dx


# ---------------------------------------------------------------------------------------------------
# 3-layer bidirectional LSTM, H = 64 (persistent sequence kernels)
# ---------------------------------------------------------------------------------------------------
def lstm64_forward(ar, key, lstm, x, B, T, h0, c0, stash, p_drop, seed_ctr, salt=0):
    """x [B*T, In] rows (b*T+t) -> out [B*T,128] (arena), hn, cn [2L,B,64] (fresh tensors).  With ``stash`` and p_drop > 0 the
    inter-layer dropout is applied by the layer kernel itself while it stores its outputs (mask from seed_ctr, which the net's
    once-per-forward tick advances: FlatParams.tick_args; ``salt`` separates the LSTM stacks of one net)."""
    return lstm64_forward_multi(ar, [(key, lstm, x, h0, c0, p_drop, seed_ctr, salt)], B, T, stash)[0]


def _l64_bufs(ar, keys, name, l, shape):
    """One arena buffer per stack -- for TWO stacks the halves of one buffer, so that a pair of tensors of the two stacks is always the
    same distance apart (ops.mm_two: the stacks' products of a layer as one batched launch, masks included)."""
    if len(keys) == 2:
        t = ar.get("%s+%s.%s%d" % (keys[0], keys[1], name, l), (2,) + tuple(shape))
        return [t[0], t[1]]
    return [ar.get("%s.%s%d" % (k, name, l), shape) for k in keys]


def _l64_wih(lstm, l):
    """(W_ih of both directions as one [512, In] matrix, their biases as one [512] vector) where they are neighbours in memory (the
    nets' flat parameter order), else None."""
    W = ops.stacked(lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1))
    b = ops.stacked(lstm.w("bias_ih", l, 0), lstm.w("bias_ih", l, 1))
    return (W, b) if W is not None and b is not None else None


def lstm64_forward_multi(ar, stacks, B, T, stash, last_out=None):
    """lstm64_forward for several INDEPENDENT stacks of the same depth and (B, T) -- stacks: (key, lstm, x, h0, c0, p_drop, seed_ctr, salt)
    each -- with ONE sequence-kernel launch per layer for all of them (mmego_lstm64_forward_multi: grid z = stack).  -> [(out, hn, cn)].
    last_out: per stack a [B*T, 128] view (any row stride) the LAST layer writes instead of its arena slot (the callers' concatenation)."""
    L = stacks[0][1].num_layers
    if any(st[1].num_layers != L for st in stacks):
        raise ValueError("lstm64_forward_multi: stacks of different depth")
    dev = stacks[0][2].device
    hns = [torch.empty((2 * L, B, 64), dtype=torch.float32, device=dev) for _ in stacks]
    cns = [torch.empty((2 * L, B, 64), dtype=torch.float32, device=dev) for _ in stacks]
    curs = [st[2] for st in stacks]
    outs = [None] * len(stacks)
    keys = [st[0] for st in stacks]
    for l in range(L):
        calls = []
        xps = _l64_bufs(ar, keys, "xp", l, (B * T, 512))
        outl = _l64_bufs(ar, keys, "out", l, (B * T, 128))
        dol = _l64_bufs(ar, keys, "do", l, (B * T, 128)) if stash and l < L - 1 else None
        mkl = _l64_bufs(ar, keys, "mk", l, (B * T, 128)) if stash and l < L - 1 else None
        wih = [_l64_wih(st[1], l) for st in stacks]
        if len(stacks) == 2 and all(w is not None for w in wih):
            # both stacks' input projections (both directions each: N = 512) as one batched product
            ops.mm_two(curs[0], wih[0][0].t(), xps[0], curs[1], wih[1][0].t(), xps[1], bias0=wih[0][1], bias1=wih[1][1])
        else:
            for i, (key, lstm, _, h0, c0, p_drop, seed_ctr, salt) in enumerate(stacks):
                ops.linear_pair(curs[i], lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1), lstm.w("bias_ih", l, 0), lstm.w("bias_ih", l, 1),
                                xps[i], 256)
        for i, (key, lstm, _, h0, c0, p_drop, seed_ctr, salt) in enumerate(stacks):
            hn, cn = hns[i], cns[i]
            xp = xps[i]
            out = last_out[i] if (last_out is not None and l == L - 1) else outl[i]
            if stash:
                gates = ar.get("%s.g%d" % (key, l), (2, T, B, 256))
                cst = ar.get("%s.c%d" % (key, l), (2, T, B, 64))
                hprev = ar.get("%s.hp%d" % (key, l), (2, B * T, 64))
                st = (gates[0], gates[1], cst[0], cst[1], hprev[0], hprev[1])
            else:
                st = (None,) * 6
            h00 = h0[2 * l] if h0 is not None else None
            h01 = h0[2 * l + 1] if h0 is not None else None
            c00 = c0[2 * l] if c0 is not None else None
            c01 = c0[2 * l + 1] if c0 is not None else None
            outs[i] = curs[i] = out
            drop = (None, None, 0.0, None, 0)
            if stash and p_drop > 0.0 and l < L - 1:
                curs[i] = dol[i]
                drop = (curs[i], mkl[i], float(p_drop), seed_ctr, 8 * salt + l)
            calls.append((B, T, xp, xp[:, 256:], 512, lstm.w("weight_hh", l, 0), lstm.w("weight_hh", l, 1),
                          lstm.w("bias_hh", l, 0), lstm.w("bias_hh", l, 1), h00, h01, c00, c01, out, out.stride(0), hn[2 * l], hn[2 * l + 1],
                          cn[2 * l], cn[2 * l + 1], *st, *drop))
        # one launch for all the stacks when their options agree (the kernel's instantiation is per launch)
        if len(calls) > 1 and len({(c[25] is None) for c in calls}) == 1:
            P = hip.pair2
            descs = (hip.Lstm64Fwd * len(calls))(*[
                hip.Lstm64Fwd(P(c[2], c[3]), c[4], P(c[5], c[6]), P(c[7], c[8]), P(c[9], c[10]), P(c[11], c[12]), hip.ptr(c[13]), c[14],
                              P(c[15], c[16]), P(c[17], c[18]), P(c[19], c[20]), P(c[21], c[22]), P(c[23], c[24]), B, T,
                              hip.ptr(c[25]), hip.ptr(c[26]), c[27], hip.ptr(c[28]), c[29]) for c in calls])
            hip.call("lstm64_forward_multi", len(calls), descs)
        else:
            for c in calls:
                hip.call("lstm64_forward", *c)
    return [(outs[i], hns[i], cns[i]) for i in range(len(stacks))]


def run_leaves(leaves):
    """Run deferred leaf closures (weight gradients): their small-tile products share launches (hip.gemm_group)."""
    if leaves:
        with hip.gemm_group():
            for fn in leaves:
                fn()
        del leaves[:]


def lstm64_backward(ar, key, lstm, x, B, T, c0, dout, G, p_drop, need_dx, leaves=None):
    """Backward of the three-layer BiLSTM(64).  The weight-gradient products of a layer are LEAVES (nothing in the backward pass
    reads them) while the rest is one dependent chain of small kernels: the six of them are issued behind the last layer as ONE
    grouped launch (hip.gemm_group / mmego_gemm_group) instead of sitting two by two between the chain's kernels."""
    return lstm64_backward_multi(ar, [(key, lstm, x, c0, dout, p_drop)], B, T, G, need_dx, leaves)[0]


def lstm64_backward_multi(ar, stacks, B, T, G, need_dx, leaves=None):
    """lstm64_backward for several independent stacks -- (key, lstm, x, c0, dout, p_drop) each -- one backward-through-time launch per
    layer for all of them (mmego_lstm64_backward_multi), every stack's weight-gradient leaves in one grouped launch.  -> [dx or None]."""
    L = stacks[0][1].num_layers
    if any(st[1].num_layers != L for st in stacks):
        raise ValueError("lstm64_backward_multi: stacks of different depth")
    d_curs = [st[4] for st in stacks]
    own = leaves is None             # (a caller's list: the caller runs it, together with its other leaves)
    if own:
        leaves = []
    keys = [st[0] for st in stacks]
    for l in range(L - 1, -1, -1):
        calls, after = [], []
        dgl = _l64_bufs(ar, keys, "dg", l, (B * T, 512))
        inps, masks = [], []
        for i, (key, lstm, x, c0, _, p_drop) in enumerate(stacks):
            if l == 0:
                inps.append(x)
            else:
                inps.append(_l64_bufs(ar, keys, "do" if p_drop > 0.0 else "out", l - 1, (B * T, 128))[i])
            masks.append(_l64_bufs(ar, keys, "mk", l - 1, (B * T, 128))[i] if l > 0 and p_drop > 0.0 else None)
        dxl = _l64_bufs(ar, keys, "dx", l, (B * T, inps[0].shape[1])) if (l > 0 or need_dx) and len({t.shape[1] for t in inps}) == 1 else None
        for i, (key, lstm, x, c0, _, p_drop) in enumerate(stacks):
            inp = inps[i]
            gates = ar.get("%s.g%d" % (key, l), (2, T, B, 256))
            cst = ar.get("%s.c%d" % (key, l), (2, T, B, 64))
            hprev = ar.get("%s.hp%d" % (key, l), (2, B * T, 64))
            dg = dgl[i]
            c00 = c0[2 * l] if c0 is not None else None
            c01 = c0[2 * l + 1] if c0 is not None else None
            d_cur = d_curs[i]
            calls.append((B, T, d_cur, d_cur.stride(0), gates[0], gates[1], cst[0], cst[1], c00, c01,
                          lstm.w("weight_hh", l, 0), lstm.w("weight_hh", l, 1), dg, dg[:, 256:], 512))
            # weight gradients of both directions per launch (batch dimension = direction)
            # the bias gradients (row sums of the gate gradients; bias_ih and bias_hh of a direction share them) ride on the two
            # batched weight-gradient products where the directions' bias tensors are neighbours in the flat buffer
            def weight_grads(l=l, dg=dg, inp=inp, hprev=hprev, lstm=lstm):
                bi = ops.stacked(G(lstm.w("bias_ih", l, 0)), G(lstm.w("bias_ih", l, 1)))
                bh = ops.stacked(G(lstm.w("bias_hh", l, 0)), G(lstm.w("bias_hh", l, 1)))
                got_i = ops.grad_weight_pair(dg, 256, inp, G(lstm.w("weight_ih", l, 0)), G(lstm.w("weight_ih", l, 1)), db=bi)
                got_h = ops.grad_weight_pair(dg, 256, hprev[0], G(lstm.w("weight_hh", l, 0)), G(lstm.w("weight_hh", l, 1)), X1=hprev[1], db=bh)
                if got_i and got_h:
                    pass
                elif B * T <= 1024:     # the four bias gradients (bias_ih = bias_hh per direction) in one launch
                    hip.call("colsum_pair", dg, dg.stride(0), B * T, 256, G(lstm.w("bias_ih", l, 0)), G(lstm.w("bias_hh", l, 0)),
                             G(lstm.w("bias_ih", l, 1)), G(lstm.w("bias_hh", l, 1)), 0)
                else:
                    for d in range(2):
                        ops.colsum(dg[:, d * 256:(d + 1) * 256], G(lstm.w("bias_ih", l, d)), out2=G(lstm.w("bias_hh", l, d)))
            leaves.append(weight_grads)
            if l > 0 or need_dx:
                def input_grad(i=i, key=key, lstm=lstm, inp=inp, dg=dg, l=l):
                    dinp = dxl[i] if dxl is not None else ar.get("%s.dx%d" % (key, l), (B * T, inp.shape[1]))
                    # (the inter-layer dropout mask is applied by the last product's epilogue)
                    mask = masks[i]
                    Wc = ops.stacked(lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1))
                    if Wc is not None:      # both directions' input weights back to back (the net's flat_param_order): one product
                        ops.grad_input(dg, Wc, dinp, cmul=mask)
                    else:
                        ops.grad_input(dg[:, :256], lstm.w("weight_ih", l, 0), dinp)
                        ops.grad_input(dg[:, 256:], lstm.w("weight_ih", l, 1), dinp, accumulate=True, cmul=mask)
                    d_curs[i] = dinp
                after.append(input_grad)
        if len(calls) > 1:
            P = hip.pair2
            descs = (hip.Lstm64Bwd * len(calls))(*[
                hip.Lstm64Bwd(hip.ptr(c[2]), c[3], P(c[4], c[5]), P(c[6], c[7]), P(c[8], c[9]), P(c[10], c[11]), P(c[12], c[13]), c[14], B, T)
                for c in calls])
            hip.call("lstm64_backward_multi", len(calls), descs)
        else:
            hip.call("lstm64_backward", *calls[0])
        wih = [_l64_wih(st[1], l) for st in stacks]
        if len(after) == 2 and dxl is not None and all(w is not None for w in wih):
            # both stacks' input gradients (dropout masks in the epilogue) as one batched product
            ops.mm_two(dgl[0], wih[0][0], dxl[0], dgl[1], wih[1][0], dxl[1], cmul0=masks[0], cmul1=masks[1])
            d_curs[0], d_curs[1] = dxl[0], dxl[1]
        else:
            for fn in after:
                fn()
    if own:
        run_leaves(leaves)
    return [d if need_dx else None for d in d_curs]


# ---------------------------------------------------------------------------------------------------
# attention pooling over the points of each group
# ---------------------------------------------------------------------------------------------------
def attn_pool_forward(X, lin, G_, P, C, vec, attn):
    hip.call("attn_pool_forward", X, lin.weight, lin.bias, G_, P, C, vec, attn)


def pool128_backward_fused(ar, key, mod, lin, attn, vec, dvec, rows, dY, G):
    """Backward of the fused pooling (pool128_bn_act): row gradients with the activated rows recomputed from "<key>.z3", the stage's
    BatchNorm sums into "<key>.gp3" (what _mlp3_backward_fused(have_sums=True) expects) and the attention parameter gradients."""
    nblk = hip.lib().mmego_mlp_train_nblk(rows)
    z3 = ar.get("%s.z3" % key, (rows, 64))
    gp3 = ar.get("%s.gp3" % key, (nblk * 2 * 64,), dtype=torch.float64)
    awp = ar.get("%s.awp" % key, (nblk, 128))
    hip.call("pool128_backward", z3, 64, rows, ops.BnState(ar, "%s.bn3" % key, 64).all, attn, vec, dvec, lin.weight, dY, dY.stride(0), gp3, awp)
    gw, gb = G(lin.weight).view(-1), G(lin.bias)
    if _dw_group is not None:
        # the attention parameters' per-workgroup partials join the pass's one reduce launch (dw_reduce_group)
        _dw_group.append(hip.DwRed(hip.ptr(awp), hip.ptr(gw), 1, 64, 0, nblk, 128))
        _dw_group.append(hip.DwRed(hip.ptr(awp[:, 64:]), hip.ptr(gb), 1, 1, 0, nblk, 128))
        _dw_keep.append(awp)
    elif gb.data_ptr() == gw.data_ptr() + 4 * gw.numel():     # weight and bias gradient slots back to back: one column sum
        ops.colsum(awp[:, :65], torch.as_strided(gw, (65,), (1,)))
    else:
        ops.colsum(awp[:, :64], gw)
        ops.colsum(awp[:, 64:65], gb)


def attn_pool_backward(ar, key, X, lin, attn, dvec, G_, P, C, dX, G):
    pdw = ar.get("%s.pdw" % key, (G_, C))
    pdb = ar.get("%s.pdb" % key, (G_, 1))
    hip.call("attn_pool_backward", X, lin.weight, attn, dvec, G_, P, C, dX, pdw, pdb)
    if G_ <= 1024:          # both parameter gradients' column sums in one launch
        hip.call("colsum2", pdw, C, G_, C, G(lin.weight).view(-1), pdb, 1, G_, 1, G(lin.bias))
    else:
        ops.colsum(pdw, G(lin.weight).view(-1))
        ops.colsum(pdb, G(lin.bias))


# ---------------------------------------------------------------------------------------------------
# generic-H step-kernel LSTM stack (forward only): IMU_Net's rnn_fast / rnn_slow
# ---------------------------------------------------------------------------------------------------
_LSTM_TWO_CHAINS = True
_side_streams = {}


class two_chains:
    """Context: switch the two-chain recurrence (lstm_recurrence) on or off.  The engines that run OTHER work beside an IMU_Net
    forward (train_step.ConcurrentStages / PipelinedStages: the other stage's small-kernel tail fills the recurrence's gaps
    already, and competes for the CUs' second workgroup slot) switch it off: measured 6.11 ms (off) against 6.19 ms (on) per
    U+L step, while a forward that runs alone gains (sequential U+L step 7.66 -> 7.51 ms, IMU-shared 4.70 -> 4.52 ms)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        global _LSTM_TWO_CHAINS
        self.was, _LSTM_TWO_CHAINS = _LSTM_TWO_CHAINS, bool(self.on)
        return self

    def __exit__(self, *exc):
        global _LSTM_TWO_CHAINS
        _LSTM_TWO_CHAINS = self.was
        return False


_LSTM_SEQ_XCD = True


class seq_xcd:
    """Context: allow (True) or forbid (False) the persistent rnn_slow launch (mmego_lstm_seq_xcd) for the forwards issued inside.
    A launch's 256 workgroups wait for each other, so every launch that may be in flight at once needs its own co-resident set:
    an engine that runs k forwards side by side (train_step.PipelinedStages) allows it for the first
    `mmego_lstm_seq_xcd_slots()` of them only; the others take the launch-per-timestep form (same results to fp32 rounding)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        global _LSTM_SEQ_XCD
        self.was, _LSTM_SEQ_XCD = _LSTM_SEQ_XCD, bool(self.on) and _LSTM_SEQ_XCD
        return self

    def __exit__(self, *exc):
        global _LSTM_SEQ_XCD
        _LSTM_SEQ_XCD = self.was
        return False


def seq_xcd_sync(ar, key):
    """The synchronisation words and the exchange buffer of mmego_lstm_seq_xcd for one BiLSTM stack of one net (its arena): zero when
    created; word 9 of the first is the kernel's sticky error flag (seq_xcd_errors), word 10 its launch generation."""
    name = "%s.seqsync" % key
    fresh = not ar.has(name)
    t = ar.get(name, (16,), dtype=torch.int32)
    x = ar.get("%s.seqxbuf" % key, (2, 8, 16, 512), dtype=torch.int64)
    if fresh:
        t.zero_()
        x.zero_()
        _seq_sync_bufs.append(t)
    return t, x


_seq_sync_bufs = []


def seq_xcd_errors(ar=None):
    """Sum of the error words of the mmego_lstm_seq_xcd sync buffers of arena `ar` (None: of every buffer handed out in this process);
    0 = every launch so far made progress.  One host read per buffer: for tests, the smoke check and the end of a benchmark, not
    for the step loop."""
    if ar is None:
        return sum(int(t[9].item()) for t in _seq_sync_bufs)
    return sum(int(t[9].item()) for (name, _, _), t in ar.bufs.items() if name.endswith(".seqsync"))


def seq_xcd_raise():
    """Raise if any persistent rnn_slow launch of this process reported that its workgroups were not co-resident (results invalid).
    The trainers call it once per epoch and every evaluation pass once at its end (processors.py); one host read per buffer."""
    n = seq_xcd_errors()
    if n:
        raise RuntimeError("mmego_lstm_seq_xcd: %d persistent recurrence buffer(s) report a launch whose workgroups could not all be "
                           "resident (shared / partitioned device?) -- the head poses computed since the last check are invalid. "
                           "blocks.seq_xcd(False) runs the recurrence as one launch per timestep." % n)


def _side_stream(cur):
    """One extra stream per launching stream (the second direction's chain of a BiLSTM layer's recurrence)."""
    st = _side_streams.get(cur.cuda_stream)
    if st is None:
        st = _side_streams[cur.cuda_stream] = torch.cuda.Stream()
    return st


def lstm_recurrence(ar, key, lstm, l, xp, out, Bn, T, gst=None, cst=None):
    """The recurrent half of BiLSTM layer ``l``: xp [Bn*T, 8H] (input projections of both directions, rows b*T+t) ->
    out [Bn*T, 2H]; gst / cst [2, T, Bn, 4H / H]: stashes for backward.

    With >= 128 rows the two directions run as TWO CHAINS of single-direction mmego_lstm_step launches on two streams (two
    parallel branches under graph capture; only where a side stream may be forked: ops.capture_can_fork): they are independent dependency chains, each launch covers the chip once
    (64 x 16 tiles, lstm_step.hip), so a CU holds one workgroup of each direction, and while one direction sits in its launch
    gap / cold-L2 prologue / cell update the other one's product loop owns the matrix pipe: 21.7 us per timestep against
    23.4 us with both directions in one launch (Bn = H = 512, scripts/bench_lstm_step.py --chains).  Same arithmetic in the
    same order: results are bit-identical (tests/test_hip_parity.py)."""
    H = lstm.hidden_size
    w0, w1 = lstm.w("weight_hh", l, 0), lstm.w("weight_hh", l, 1)
    b0, b1 = lstm.w("bias_hh", l, 0), lstm.w("bias_hh", l, 1)
    if (_LSTM_SEQ_XCD and gst is None and cst is None and T > 1 and w0.is_contiguous() and w1.is_contiguous()
            and hip.lib().mmego_lstm_seq_xcd_ok(Bn, H, T)):
        # <= 64 rows (IMU_Net's rnn_slow): the layer's whole recurrence as ONE persistent launch with stationary weights
        # (lstm_seq.hip): 2 launches per forward instead of 2 T
        sync, xbuf = seq_xcd_sync(ar, key)
        hip.call("lstm_seq_xcd", xp, xp.stride(0), w0, w1, b0, b1, out, out.stride(0), sync, xbuf, Bn, H, T)
        return
    c = ar.get("%s.c" % key, (2, Bn, H))
    xp_p, out_p = xp.data_ptr(), out.data_ptr()
    xs, os_ = T * 8 * H, T * 2 * H       # row strides between consecutive batch rows b
    if _LSTM_TWO_CHAINS and T > 1 and Bn >= 128 and ops.capture_can_fork():
        cur = torch.cuda.current_stream()
        side = _side_stream(cur)
        # the product-less first timestep of BOTH directions as one (elementwise) launch, in front of the fork
        hip.call("lstm_step", 2, Bn, H, 1, None, None, os_, w0, w1, b0, b1, xp_p, xp_p + 4 * ((T - 1) * 8 * H + 4 * H), xs,
                 out_p, out_p + 4 * ((T - 1) * 2 * H + H), os_, c[0], c[1],
                 None if gst is None else gst[0, 0], None if gst is None else gst[1, T - 1],
                 None if cst is None else cst[0, 0], None if cst is None else cst[1, T - 1])
        side.wait_stream(cur)
        for s in range(1, T):
            t0, t1 = s, T - 1 - s
            hip.call("lstm_step", 1, Bn, H, int(s == 0), out_p + 4 * ((t0 - 1) * 2 * H) if s > 0 else None, None, os_, w0, None, b0, None,
                     xp_p + 4 * (t0 * 8 * H), None, xs, out_p + 4 * (t0 * 2 * H), None, os_, c[0], None,
                     None if gst is None else gst[0, t0], None, None if cst is None else cst[0, t0], None)
            with torch.cuda.stream(side):
                hip.call("lstm_step", 1, Bn, H, int(s == 0), out_p + 4 * ((t1 + 1) * 2 * H + H) if s > 0 else None, None, os_, w1, None,
                         b1, None, xp_p + 4 * (t1 * 8 * H + 4 * H), None, xs, out_p + 4 * (t1 * 2 * H + H), None, os_, c[1], None,
                         None if gst is None else gst[1, t1], None, None if cst is None else cst[1, t1], None)
        cur.wait_stream(side)
        return
    for s in range(T):
        t0, t1 = s, T - 1 - s
        hp0 = out_p + 4 * ((t0 - 1) * 2 * H) if s > 0 else None          # h_{t-1} of each direction
        hp1 = out_p + 4 * ((t1 + 1) * 2 * H + H) if s > 0 else None
        hip.call("lstm_step", 2, Bn, H, int(s == 0), hp0, hp1, os_, w0, w1, b0, b1,
                 xp_p + 4 * (t0 * 8 * H), xp_p + 4 * (t1 * 8 * H + 4 * H), xs,
                 out_p + 4 * (t0 * 2 * H), out_p + 4 * (t1 * 2 * H + H), os_, c[0], c[1],
                 None if gst is None else gst[0, t0], None if gst is None else gst[1, t1],
                 None if cst is None else cst[0, t0], None if cst is None else cst[1, t1])


def lstm_steps_forward(ar, key, lstm, x, Bn, T):
    """x [Bn*T, In] rows (b*T+t) -> out [Bn*T, 2H] of the last layer (eval mode: no dropout)."""
    H = lstm.hidden_size
    cur = x
    out = None
    for l in range(lstm.num_layers):
        xp = ar.get("%s.xp%d" % (key, l), (Bn * T, 8 * H))
        ops.linear_pair(cur, lstm.w("weight_ih", l, 0), lstm.w("weight_ih", l, 1), lstm.w("bias_ih", l, 0), lstm.w("bias_ih", l, 1), xp, 4 * H)
        out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        lstm_recurrence(ar, key, lstm, l, xp, out, Bn, T)
        cur = out
    return out


# ---------------------------------------------------------------------------------------------------
# bf16-operand / fp32-accumulate BiLSTM forward (opt-in precision mode of the frozen IMU_Net, bf16.hip)
# ---------------------------------------------------------------------------------------------------
def cvt_bf16(x, out):
    """out (torch.bfloat16 storage) = round-to-nearest-even(x); 2-D row-strided views."""
    if x.dim() != 2 or out.shape != x.shape or x.stride(1) != 1 or out.stride(1) != 1 or out.dtype != torch.bfloat16:
        raise ValueError("cvt_bf16 needs 2-D unit-column-stride fp32 input and a bf16 output of the same shape")
    hip.call("cvt_bf16", x, x.stride(0), x.shape[0], x.shape[1], out, out.stride(0))
    return out


def lstm_bf16_weights(lstm):
    """Per layer: (W_ih of both directions stacked [8H, In] bf16, b_ih + b_hh stacked [8H] fp32, W_hh bf16 per direction).
    Built once per parameter version (the IMU_Net of stages 2/3 is frozen)."""
    ver = tuple(p._version for p in lstm.parameters()) + tuple(p.data_ptr() for p in lstm.parameters())
    cache = getattr(lstm, "_bf16_cache", None)
    if cache is not None and cache[0] == ver:
        return cache[1]
    H = lstm.hidden_size
    layers = []
    for l in range(lstm.num_layers):
        In = lstm.w("weight_ih", l, 0).shape[1]
        dev = lstm.w("weight_ih", l, 0).device
        wih = torch.empty((8 * H, In), dtype=torch.bfloat16, device=dev)
        bias = torch.empty((8 * H,), dtype=torch.float32, device=dev)
        whh = []
        for d in range(2):
            cvt_bf16(lstm.w("weight_ih", l, d).detach(), wih[4 * H * d:4 * H * (d + 1)])
            hip.call("add", lstm.w("bias_ih", l, d).detach(), lstm.w("bias_hh", l, d).detach(), bias[4 * H * d:], 4 * H)
            wb = cvt_bf16(lstm.w("weight_hh", l, d).detach(), torch.empty((4 * H, H), dtype=torch.bfloat16, device=dev))
            # fragment-major [hidden block][gate][16-k step][k half][unit][8 k] (include/mmego_hip.h, mmego_lstm_step_bf16):
            # a one-time re-layout of frozen weights
            whh.append(wb.view(4, H // 32, 32, H // 16, 2, 8).permute(1, 0, 3, 4, 2, 5).contiguous())
        layers.append((wih, bias, whh[0], whh[1]))
    lstm._bf16_cache = (ver, layers)
    return layers


def _frag_major_w(wb, H):
    """[4H, K] bf16 row-major -> fragment-major [hidden block][gate][16-k step][k half][unit][8 k] (include/mmego_hip.h)."""
    K = wb.shape[1]
    return wb.view(4, H // 32, 32, K // 16, 2, 8).permute(1, 0, 3, 4, 2, 5).contiguous()


def lstm_bf16_weights_fused(lstm):
    """Per layer: ([per direction: list of fragment-major W_ih column blocks], [W_hh fragment-major per direction], bias [2][4H])
    for the fused projection + recurrence step (large batches)."""
    ver = tuple(p._version for p in lstm.parameters()) + tuple(p.data_ptr() for p in lstm.parameters())
    cache = getattr(lstm, "_bf16_fused_cache", None)
    if cache is not None and cache[0] == ver:
        return cache[1]
    H = lstm.hidden_size
    base = lstm_bf16_weights(lstm)
    layers = []
    for l in range(lstm.num_layers):
        wih, bias, whh0, whh1 = base[l]
        segs = []
        for d in range(2):
            wd = wih[4 * H * d:4 * H * (d + 1)]
            cols = [wd] if l == 0 else [wd[:, :H], wd[:, H:]]
            segs.append([_frag_major_w(c.contiguous(), H) for c in cols])
        layers.append((segs, (whh0, whh1), bias))
    lstm._bf16_fused_cache = (ver, layers)
    return layers


FUSED_MIN_ROWS = 2048


def fused_input_fragments(ar, key, Bn, T, In):
    """The layer-0 operand buffer of lstm_steps_forward_bf16_fused: T fragment-major [Bp x In] bf16 matrices (a producer that writes
    them itself -- IMU_Net's fc1, mmego_fc_relu_bf16_frag_tm -- passes the filled buffer as `xfrag`)."""
    Bp = (Bn + 31) // 32 * 32
    return ar.get("%s.xfrag" % key, (T, Bp * In), dtype=torch.bfloat16), Bp


def lstm_steps_forward_bf16_fused(ar, key, lstm, x, Bn, T, xfrag=None, pool=None):
    """Large-batch form of lstm_steps_forward_bf16: no projection tensor; every step multiplies [x_t | h_{t-1}] by [W_ih | W_hh]
    (bf16.hip, lstm_step_bf16_fused_kernel).  Layer inputs and all h_t live fragment-major, one [Bp x K] matrix per timestep.
    x [Bn*T, In] fp32 rows (b*T+t), or None with `xfrag` = the layer-0 operand already in that layout.
    pool = (attention Linear, vec [Bn, 2H], attn [Bn, T]): the softmax pooling over the T timesteps behind the stack reads the last
    layer's bf16 fragments (mmego_attn_pool_frag_bf16) -- no fp32 output is written at all; returns None then."""
    H = lstm.hidden_size
    W = lstm_bf16_weights_fused(lstm)
    In = lstm.input_size
    xf, Bp = fused_input_fragments(ar, key, Bn, T, In)
    if xfrag is None:
        hip.call("cvt_bf16_frag_tm", x, x.stride(0), Bn, T, In, xf, Bp)
    elif xfrag.data_ptr() != xf.data_ptr():
        raise ValueError("lstm_steps_forward_bf16_fused: xfrag must be the buffer of fused_input_fragments")
    out = None
    prev = None
    for l in range(lstm.num_layers):
        segs, whh, bias = W[l]
        pooled = pool is not None and bool(hip.lib().mmego_attn_pool_frag_bf16_ok(H))
        last = l == lstm.num_layers - 1 and not pooled           # (the layer that writes the fp32 output)
        hf = ar.get("%s.hfall%d" % (key, l), (T, 2, Bp * H), dtype=torch.bfloat16)      # h_t of every timestep, fragment-major
        c = ar.get("%s.c" % key, (2, Bn, H))
        if last:
            out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        out_p = out.data_ptr() if last else 0
        os_ = T * 2 * H
        for s in range(T):
            t0, t1 = s, T - 1 - s
            if l == 0:
                a = (xf[t0], xf[t1], segs[0][0], segs[1][0], In, None, None, None, None, 0)
                nseg = 1
            else:
                a = (prev[t0, 0], prev[t1, 0], segs[0][0], segs[1][0], H, prev[t0, 1], prev[t1, 1], segs[0][1], segs[1][1], H)
                nseg = 2
            hip.call("lstm_step_bf16_fused", 2, Bn, H, int(s == 0), nseg, *a,
                     hf[t0 - 1, 0] if s > 0 else None, hf[t1 + 1, 1] if s > 0 else None, whh[0], whh[1], bias,
                     out_p + 4 * (t0 * 2 * H) if last else None, out_p + 4 * (t1 * 2 * H + H) if last else None, os_,
                     hf[t0, 0], hf[t1, 1], c[0], c[1])
        prev = hf
    if pool is not None:
        lin, vec, attn = pool
        if out is None:
            hip.call("attn_pool_frag_bf16", prev, T, Bp, Bn, H, lin.weight, lin.bias, vec, attn)
            return None
        attn_pool_forward(out, lin, Bn, T, 2 * H, vec, attn)
        return None
    return out


def lstm_steps_forward_bf16(ar, key, lstm, x, Bn, T):
    """As lstm_steps_forward with bf16 product operands: x [Bn*T, In] fp32 rows (b*T+t) -> out [Bn*T, 2H] fp32 of the last
    layer (same row order).  Gate pre-activations, cell state and outputs are fp32.  Internal layouts (include/mmego_hip.h):
    projection operands time-major bf16 [T][Bp][In], projections tile-major fp32, recurrent operands fragment-major."""
    H = lstm.hidden_size
    In = x.shape[1]
    if Bn >= FUSED_MIN_ROWS and In % 64 == 0:
        return lstm_steps_forward_bf16_fused(ar, key, lstm, x, Bn, T)
    W = lstm_bf16_weights(lstm)
    Bp = (Bn + 31) // 32 * 32
    cur = ar.get("%s.xtm" % key, (T * Bp, In), dtype=torch.bfloat16)
    hip.call("cvt_bf16_tm", x, x.stride(0), Bn, T, In, cur, Bp)
    out = None
    for l in range(lstm.num_layers):
        wih, bias, whh0, whh1 = W[l]
        xpf = ar.get("%s.xpf%d" % (key, l), (T * Bp * 8 * H,))
        hip.call("gemm_bf16", cur, cur.stride(0), wih, wih.stride(0), None, 0, None, 0, xpf, bias, T * Bp, 8 * H, cur.shape[1], 0)
        out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        last = l == lstm.num_layers - 1
        outb = None if last else ar.get("%s.outb%d" % (key, l), (T * Bp, 2 * H), dtype=torch.bfloat16)   # time-major
        c = ar.get("%s.c" % key, (2, Bn, H))
        hf = ar.get("%s.hfrag" % key, (2, 2, Bp, H), dtype=torch.bfloat16)   # [ping-pong][direction]
        out_p = out.data_ptr()
        outb_p = 0 if last else outb.data_ptr()
        os_ = T * 2 * H                      # fp32 output: row stride between consecutive batch rows b
        for s in range(T):
            t0, t1 = s, T - 1 - s
            prev, nxt = hf[(s + 1) & 1], hf[s & 1]
            hip.call("lstm_step_bf16", 2, Bn, H, int(s == 0), prev[0] if s > 0 else None, prev[1] if s > 0 else None,
                     whh0, whh1, xpf, t0 * (Bp // 32), t1 * (Bp // 32),
                     out_p + 4 * (t0 * 2 * H), out_p + 4 * (t1 * 2 * H + H), os_,
                     None if last else outb_p + 2 * (t0 * Bp * 2 * H), None if last else outb_p + 2 * (t1 * Bp * 2 * H + H), 2 * H,
                     nxt[0], nxt[1], c[0], c[1])
        cur = outb
    return out


# ---------------------------------------------------------------------------------------------------
# fp32-ACCURATE BiLSTM forward on the bf16 matrix pipe (opt-in mode "split3" of the frozen IMU_Net, split3.hip): every
# fp32 operand as three bf16 pieces (exact), six piece products per product, fp32 accumulation
# ---------------------------------------------------------------------------------------------------
SPLIT3_NPROD = int(os.environ.get("MMEGO_SPLIT3_NPROD", "6"))       # 6: dropped terms <= 2^-24 relative; 9: all piece products
SPLIT3_WM = 0             # projection tile rows / 64 (0: the library's choice)
# True: a layer's recurrence as two chains of single-direction launches on 16-unit workgroups (mmego_split3_step16).  Off: 4 % faster
# per forward when the forward runs alone, no faster inside a step.  It is the form in which the co-residency corruption of r05 showed
# most often (a kernel WITH packed-fp32 instructions on a CU that also holds one of its workgroups: DESIGN.md section 7d).
SPLIT3_TWO_CHAINS = False        # (a measurement / test hook, see split3_two_chains(); no environment variable since r06)


class split3_two_chains:
    """`with blocks.split3_two_chains(True)`: rnn_fast's split3 recurrences as two chains of the 16-unit step kernel (two bf16-MFMA
    workgroups per CU) where forking is allowed.  The form in which the co-residency corruption of r05 showed most often; kept for
    its reproducers and regression tests, not a user setting."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        global SPLIT3_TWO_CHAINS
        self.was, SPLIT3_TWO_CHAINS = SPLIT3_TWO_CHAINS, self.on
        return self

    def __exit__(self, et, ev, tb):
        global SPLIT3_TWO_CHAINS
        SPLIT3_TWO_CHAINS = self.was
        return False


def split3_buffer(ar, name, Rp, K):
    """A "sfrag" operand buffer (include/mmego_hip.h): Rp x K values as three bf16 pieces, [Rp / 32][K / 16][3][64][8]."""
    return ar.get(name, (Rp // 32, K // 16, 3, 64, 8), dtype=torch.bfloat16)


def split3_cvt(x, out=None, Rp=None, tm=None):
    """fp32 [rows, K] (unit column stride) -> sfrag pieces.  tm = (Bn, T, Bp): rows b*T + t in, rows t*Bp + b out."""
    rows, K = x.shape
    if x.stride(1) != 1 or x.dtype != torch.float32 or K % 16:
        raise ValueError("split3_cvt needs fp32 rows with unit column stride and K % 16 == 0")
    if tm is not None:
        Bn, T, Bp = tm
        Rp = T * Bp
    else:
        Bn = T = Bp = 0
        Rp = (rows + 31) // 32 * 32 if Rp is None else Rp
    if out is None:
        out = torch.empty((Rp // 32, K // 16, 3, 64, 8), dtype=torch.bfloat16, device=x.device)
    hip.call("split3_cvt", x, x.stride(0), rows, K, int(tm is not None), Bn, T, Bp, Rp, out)
    return out


def split3_cvt_t(x, out=None, shift=0, T=0):
    """fp32 [R, C] (unit column stride, R % 16 == 0) -> the sfrag pieces of x^T ([C rounded up to 32] rows, K = R).
    T > 0: column r of x^T is row r + shift of x inside r's T-row sequence, zero outside (h_{t-1} / h_{t+1} without a copy)."""
    R, C = x.shape
    if x.stride(1) != 1 or x.dtype != torch.float32 or R % 16:
        raise ValueError("split3_cvt_t needs fp32 rows with unit column stride and R % 16 == 0")
    Cp = (C + 31) // 32 * 32
    if out is None:
        out = torch.empty((Cp // 32, R // 16, 3, 64, 8), dtype=torch.bfloat16, device=x.device)
    hip.call("split3_cvt_t", x, x.stride(0), R, C, Cp, out, int(shift), int(T))
    return out


def split3_join(y, rows, K):
    """sfrag pieces -> fp32 [Rp, K] (a1 + a2 + a3): the inverse of split3_cvt for values in the exact range."""
    Rp = y.shape[0] * 32
    x = torch.empty((Rp, K), dtype=torch.float32, device=y.device)
    hip.call("split3_join", y, Rp, K, x, K)
    return x[:rows]


def _split3_cache_of(lstm):
    """The per-layout cache of an LSTM's weight pieces for the CURRENT parameter version (an empty one after any change)."""
    ver = tuple(p._version for p in lstm.parameters()) + tuple(p.data_ptr() for p in lstm.parameters())
    cache = getattr(lstm, "_split3_cache", None)
    if cache is None or cache.get("ver") != ver:
        cache = lstm._split3_cache = {"ver": ver}
    return cache


def lstm_split3_weights(lstm, nu=32):
    """Per layer: (W_ih of both directions stacked [8H, In] as pieces, b_ih + b_hh stacked [8H] fp32, W_hh pieces per direction).
    nu = 32: W_hh rows reordered [32-unit block][gate][32 units] (mmego_split3_step), the projection's columns in
    PyTorch's order; nu = 16: W_hh rows AND the projection's columns (W_ih rows, biases) per direction reordered [16-unit block][gate]
    [16 units] (mmego_split3_step16).  Built once per weight version and layout: keyed like the bf16 copies on every parameter's
    (_version, data_ptr) -- load_state_dict / copy_ in eval mode drop it -- and dropped by weights_changed() for the
    writers that do not bump _version (fused Adam)."""
    cache = _split3_cache_of(lstm)
    if nu in cache:
        return cache[nu]
    H = lstm.hidden_size
    blocked = lambda m: m.view(4, H // nu, nu, -1).permute(1, 0, 2, 3).reshape(4 * H, -1).contiguous()       # rows [block][gate][unit]
    layers = []
    with torch.no_grad():
        for l in range(lstm.num_layers):
            wih, bias, whh = [], [], []
            for d in range(2):
                wi, bi, bh = lstm.w("weight_ih", l, d).detach(), lstm.w("bias_ih", l, d).detach(), lstm.w("bias_hh", l, d).detach()
                if nu == 16:                                   # the projection's columns follow the step kernel's gate-pair blocks
                    wi, bi, bh = blocked(wi), blocked(bi.view(-1, 1)).view(-1), blocked(bh.view(-1, 1)).view(-1)
                wih.append(wi)
                bias.append(bi + bh)
                whh.append(split3_cvt(blocked(lstm.w("weight_hh", l, d).detach())))
            layers.append((split3_cvt(torch.cat(wih, 0).contiguous()), torch.cat(bias).contiguous(), whh[0], whh[1]))
    cache[nu] = layers
    return layers


def lstm_steps_forward_split3(ar, key, lstm, x, Bn, T, xfrag=None, nprod=None):
    """lstm_steps_forward with every product on split operands: x [Bn*T, In] fp32 rows (b*T + t) -- or None with `xfrag` = the
    layer-0 operand already as pieces in time-major rows (split3_buffer(ar, key + ".x", T*Bp, In), e.g. written by
    mmego_split3_fc_relu) -> out [Bn*T, 2H] fp32 of the last layer, rows (b*T + t).  Projections tile-major fp32; a layer's h_t of
    all timesteps and both directions live as pieces in ONE buffer [T*Bp/32][2H/16][3] KB that is the next step's operand (a
    window) and the next layer's projection operand (whole).
    The recurrence of a layer runs as TWO CHAINS of single-direction launches on 16-unit workgroups (mmego_split3_step16; two streams =
    two parallel branches under graph capture) wherever blocks.lstm_recurrence would do so for the fp32 kernels (>= 128 rows, forking
    allowed), else as one both-direction launch per timestep on 32-unit workgroups (mmego_split3_step)."""
    H = lstm.hidden_size
    In = lstm.input_size
    nprod = SPLIT3_NPROD if nprod is None else nprod
    chains = _LSTM_TWO_CHAINS and T > 1 and Bn >= 128 and ops.capture_can_fork() and SPLIT3_TWO_CHAINS
    W = lstm_split3_weights(lstm, 16 if chains else 32)
    Bp = (Bn + 31) // 32 * 32
    cur = split3_buffer(ar, "%s.x" % key, T * Bp, In)
    if xfrag is None:
        split3_cvt(x, out=cur, tm=(Bn, T, Bp))
    elif xfrag.data_ptr() != cur.data_ptr():
        raise ValueError("lstm_steps_forward_split3: xfrag must be split3_buffer(ar, key + '.x', T * Bp, In)")
    K = In
    out = None
    nrb = Bp // 32
    S2 = 2 * H // 16                                   # 16-k steps of a layer-output row
    hrb = S2 * 3                                       # 1-KB blocks between row blocks of the layer output
    for l in range(lstm.num_layers):
        wih, bias, whh0, whh1 = W[l]
        last = l == lstm.num_layers - 1
        xpf = ar.get("%s.s3xpf%d" % (key, l), (T * Bp * 8 * H,))
        hip.call("split3_gemm", cur, wih, xpf, None, 0, bias, T * nrb, 8 * H // 32, K, 0, nprod, SPLIT3_WM)
        O = split3_buffer(ar, "%s.s3h%d" % (key, l), T * Bp, 2 * H)
        if last:
            out = ar.get("%s.out%d" % (key, l), (Bn * T, 2 * H))
        c = ar.get("%s.c" % key, (2, Bn, H))
        o_p = O.data_ptr()
        out_p = out.data_ptr() if last else 0
        os_ = T * 2 * H
        win = lambda t, d: o_p + 2 * ((t * nrb * S2 + d * (H // 16)) * 3 * 512)       # bytes: 1 KB = 512 bf16
        ho = lambda t, d: out_p + 4 * (t * 2 * H + d * H) if last else None
        if chains:
            cur_s = torch.cuda.current_stream()
            side = _side_stream(cur_s)
            # the product-less first timestep of both directions as one launch, in front of the fork
            hip.call("split3_step16", 2, Bn, H, 1, None, None, hrb, whh0, whh1, xpf, 0, (T - 1) * nrb, ho(0, 0), ho(T - 1, 1), os_,
                     win(0, 0), win(T - 1, 1), hrb, c[0], c[1], nprod, 0)
            side.wait_stream(cur_s)
            for s in range(1, T):
                t0, t1 = s, T - 1 - s
                hip.call("split3_step16", 1, Bn, H, 0, win(t0 - 1, 0), None, hrb, whh0, None, xpf, t0 * nrb, 0, ho(t0, 0), None, os_,
                         win(t0, 0), None, hrb, c[0], None, nprod, 0)
                with torch.cuda.stream(side):
                    hip.call("split3_step16", 1, Bn, H, 0, win(t1 + 1, 1), None, hrb, whh1, None, xpf, t1 * nrb, 0, ho(t1, 1), None, os_,
                             win(t1, 1), None, hrb, c[1], None, nprod, 1)
            cur_s.wait_stream(side)
        else:
            for s in range(T):
                t0, t1 = s, T - 1 - s
                hip.call("split3_step", 2, Bn, H, int(s == 0), win(t0 - 1, 0) if s > 0 else None, win(t1 + 1, 1) if s > 0 else None, hrb,
                         whh0, whh1, xpf, t0 * nrb, t1 * nrb, ho(t0, 0), ho(t1, 1), os_,
                         win(t0, 0), win(t1, 1), hrb, c[0], c[1], nprod, 0)
        cur, K = O, 2 * H
    return out


def lstm_split3_proj_weights(lstm):
    """Per layer and direction: W_ih as pieces with rows reordered [32-unit block][gate][32 units] (mmego_split3_proj), and b_ih of both
    directions stacked [8H]."""
    cache = _split3_cache_of(lstm)
    if "proj" in cache:
        return cache["proj"]
    H = lstm.hidden_size
    blocked = lambda m: m.view(4, H // 32, 32, -1).permute(1, 0, 2, 3).reshape(4 * H, -1).contiguous()
    layers = []
    with torch.no_grad():
        for l in range(lstm.num_layers):
            w = [split3_cvt(blocked(lstm.w("weight_ih", l, d).detach())) for d in range(2)]
            layers.append((w[0], w[1], torch.cat((lstm.w("bias_ih", l, 0).detach(), lstm.w("bias_ih", l, 1).detach())).contiguous()))
    cache["proj"] = layers
    return layers


def lstm_steps_forward_split3_proj(ar, key, lstm, x, Bn, T, nprod=None):
    """lstm_steps_forward for a BiLSTM whose recurrence stays on the fp32 kernels (IMU_Net's rnn_slow: 64 rows, a persistent
    weight-stationary launch per layer that is latency-bound, not matrix-bound) with only the INPUT PROJECTIONS on split operands:
    per layer one conversion of the layer input to pieces, one mmego_split3_proj launch (row-major fp32 result, rows b*T + t, W_ih x +
    b_ih of both directions), then blocks.lstm_recurrence as in the fp32 path."""
    H = lstm.hidden_size
    nprod = SPLIT3_NPROD if nprod is None else nprod
    W = lstm_split3_proj_weights(lstm)
    rows = Bn * T
    Rp = (rows + 31) // 32 * 32
    cur = x
    out = None
    for l in range(lstm.num_layers):
        K = cur.shape[1]
        xs = split3_buffer(ar, "%s.s3in%d" % (key, l), Rp, K)
        split3_cvt(cur, out=xs, Rp=Rp)
        xp = ar.get("%s.xp%d" % (key, l), (rows, 8 * H))
        hip.call("split3_proj", xs, W[l][0], W[l][1], W[l][2], xp, xp.stride(0), rows, H, K, nprod)
        out = ar.get("%s.out%d" % (key, l), (rows, 2 * H))
        lstm_recurrence(ar, key, lstm, l, xp, out, Bn, T)
        cur = out
    return out
