"""GPU: the HIP path (through the C ABI) against the CPU oracle and the reference goldens.

Tolerances (fp32 path): joint coordinates 2e-5 m absolute (= 2e-3 cm per coordinate; the north-star bar
of 1e-3 cm is on the error METRIC and is asserted separately), gradients 2e-4 of the largest gradient
(the reference's own fp32 gradients are ~3e-3 away from fp64, see test_oracle_golden), indices bit-exact.
"""
import re

import numpy as np
import pytest
import torch

from conftest import check_pinned, golden, load_weights, set_lstm_dropout
from oracle import geometry as geo
from oracle import metric as om
from oracle import nets as on
from oracle import skeleton as sk
from oracle import train as ot

pytestmark = pytest.mark.gpu

NOISE_GRAD = re.compile(r"(conv[123]\.bias|tcn\.2\.bias|residual\.0\.bias|attn\.bias|to_k\.bias|fusion\.attn\.weight)$")


def T(a):
    return torch.tensor(np.asarray(a))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from mmego_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def test_gemm_variants(dev):
    from mmego_amd import ops
    g = torch.Generator().manual_seed(3)
    cases = [(256, 256, 64), (128, 384, 512), (70, 33, 17), (1000, 8, 6), (5, 87, 128), (513, 257, 100)]
    for M, N, K in cases:
        A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
        ref = A.double() @ W.double().t() + b.double()
        out = torch.empty(M, N, device=dev)
        ops.linear(A.to(dev), W.to(dev), b.to(dev), out)
        assert torch.allclose(out.cpu().double(), ref, rtol=1e-5, atol=1e-4 * K ** 0.5), (M, N, K)
        # dX = dY @ W, dW = dY^T @ X (split over rows), strided output slice
        dY = torch.randn(M, N, generator=g)
        dX = torch.empty(M, K + 3, device=dev)
        ops.grad_input(dY.to(dev), W.to(dev), dX[:, 3:])
        assert torch.allclose(dX[:, 3:].cpu().double(), dY.double() @ W.double(), rtol=1e-5, atol=1e-4 * N ** 0.5)
        dW = torch.empty(N, K, device=dev)
        ops.grad_weight(dY.to(dev), A.to(dev), dW)
        assert torch.allclose(dW.cpu().double(), dY.double().t() @ A.double(), rtol=1e-5, atol=1e-4 * M ** 0.5)
    # large-tile kernels in every operand orientation (NT / NN / TN / TT), column-slice operands, accumulate, relu + bias,
    # split-K slabs, 128x128 persistent + 128x64 half tiles (M x N = 2304 x 2048 -> 288 tiles ... 10240 x 1024 -> 640 units)
    for M, N, K, nsplit in ((512, 256, 192, 1), (1024, 640, 128, 1), (256, 128, 1536, 4), (2048, 1024, 256, 3), (5120, 2048, 128, 1)):
        Ad, Bd = torch.randn(M, K + 8, generator=g).to(dev), torch.randn(K, N + 4, generator=g).to(dev)
        At, Bt = Ad[:, 4:K + 4].t().contiguous(), Bd[:, :N].t().contiguous()        # [K, M], [N, K]
        bias = torch.randn(N, generator=g).to(dev)
        ref = Ad[:, 4:K + 4].double() @ Bd[:, :N].double()
        tol = dict(rtol=1e-5, atol=2e-5 * K ** 0.5)
        for a_kc in (True, False):
            for b_kc in (True, False):
                Aop = Ad[:, 4:K + 4] if a_kc else At.t()            # [M, K] view: k- or m-contiguous
                Bop = Bt.t() if b_kc else Bd[:, :N]                 # [K, N] view: k- or n-contiguous
                C = torch.zeros(M, N + 64, device=dev)
                ops.mm(Aop, Bop, C[:, 64:], nsplit=nsplit)
                assert torch.allclose(C[:, 64:].double(), ref, **tol), (M, N, K, nsplit, a_kc, b_kc)
                assert (C[:, :64] == 0).all()
        C0 = torch.randn(M, N, generator=g).to(dev)
        C = C0.clone()
        ops.mm(Ad[:, 4:K + 4], Bd[:, :N], C, accumulate=True)
        assert torch.allclose(C.double(), C0.double() + ref, **tol), (M, N, K, "accumulate")
        ops.mm(At.t(), Bt.t(), C, bias=bias, relu=True, nsplit=nsplit)
        assert torch.allclose(C.double(), torch.relu(ref + bias.double()), **tol), (M, N, K, "bias+relu")
    # relu + accumulate + batched broadcast
    A, B = torch.randn(7, 15, 15, generator=g), torch.randn(7, 15, 40, generator=g)
    C = torch.randn(7, 15, 40, generator=g)
    Cd = C.to(dev)
    ops.bmm(A[0].t().unsqueeze(0).expand(7, 15, 15).to(dev), B.to(dev), Cd, accumulate=True)
    assert torch.allclose(Cd.cpu(), C + A[0].t() @ B, atol=1e-4)


def test_linear_pair_and_batched_bias(dev):
    """ops.linear_pair (both directions' input projections as one batched product with a bias batch stride) equals two separate
    ops.linear calls bit for bit on every dispatch path: persistent tiles (10240 rows), plain tiles (512 rows), the K-quartered
    small kernel (ragged 200 x 96) and the non-contiguous fallback."""
    from mmego_amd import ops
    g = torch.Generator().manual_seed(5)
    # (10240 x 2048 per direction: 2560 tiles = five whole rounds of the persistent grid -> the phase-shifted walk, gemm_tile.hip)
    for rows, ncol, K in ((10240, 256, 128), (10240, 2048, 128), (512, 2048, 256), (200, 96, 160), (64, 32, 24)):
        x = torch.randn(rows, K, generator=g).to(dev)
        Wb = (torch.randn(2 * ncol + 8, K, generator=g) * 0.1).to(dev)
        bb = torch.randn(2 * ncol + 8, generator=g).to(dev)
        W0, W1, b0, b1 = Wb[:ncol], Wb[ncol + 8:], bb[:ncol], bb[ncol + 8:]        # a gap between the two blocks
        ref = torch.empty(rows, 2 * ncol, device=dev)
        ops.linear(x, W0, b0, ref[:, :ncol])
        ops.linear(x, W1, b1, ref[:, ncol:])
        out = torch.full((rows, 2 * ncol + 4), 3.0, device=dev)
        ops.linear_pair(x, W0, W1, b0, b1, out, ncol)
        assert torch.equal(out[:, :2 * ncol], ref), (rows, ncol, K)
        assert (out[:, 2 * ncol:] == 3.0).all()
        dbl = x.double() @ W1.double().t() + b1.double()
        assert torch.allclose(out[:, ncol:2 * ncol].double(), dbl, rtol=1e-5, atol=2e-5 * K ** 0.5)
    # non-contiguous weight view -> falls back to two products
    Wt = (torch.randn(64, 2 * 48, generator=g) * 0.1).to(dev)
    x = torch.randn(100, 64, generator=g).to(dev)
    out = torch.empty(100, 96, device=dev)
    b = torch.zeros(96, device=dev)
    ops.linear_pair(x, Wt[:, :48].t(), Wt[:, 48:].t(), b[:48], b[48:], out, 48)
    assert torch.allclose(out.double(), x.double() @ Wt.double(), rtol=1e-5, atol=1e-4)


def test_fused_eval_mlp_and_bn_fold(dev):
    """mmego_bn_fold_linear + mmego_mlp3_eval (eval-mode PointNet stages in one kernel) against conv -> BatchNorm(eval) -> ReLU
    in fp64, for the three channel plans of the path, ragged row counts and column-slice inputs / outputs."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(11)
    for rows, dims in ((1000, (28, 32, 48, 64)), (77, (6, 8, 16, 24)), (4099, (6, 16, 32, 61))):
        Cin = dims[0]
        xbuf = torch.randn(rows, Cin + 5, generator=g).to(dev)
        x = xbuf[:, 3:3 + Cin]                                            # column slice (row stride Cin + 5)
        ybuf = torch.full((rows, dims[3] + 7), 7.0, device=dev)
        y = ybuf[:, 2:2 + dims[3]]
        cur = x.double().cpu()
        folded, raw, bn = [], [], []
        for i in range(3):
            K, C = dims[i], dims[i + 1]
            W, b = torch.randn(C, K, generator=g) * 0.4, torch.randn(C, generator=g) * 0.2
            gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
            mean, var = torch.randn(C, generator=g) * 0.3, torch.rand(C, generator=g) + 0.2
            z = cur @ W.double().t() + b.double()
            cur = torch.relu((z - mean.double()) / torch.sqrt(var.double() + 1e-5) * gamma.double() + beta.double())
            wf, bf = torch.empty(C, K, device=dev), torch.empty(C, device=dev)
            dv = [t.to(dev) for t in (W, b, gamma, beta, mean, var)]
            hip.call("bn_fold_linear", dv[0], dv[1], C, K, dv[2], dv[3], dv[4], dv[5], 1e-5, wf, bf)
            folded += [wf, bf, C]
            raw += [dv[0], dv[1], C]
            bn += dv[2:]
        hip.call("mlp3_eval", x, x.stride(0), rows, Cin, *folded, y, y.stride(0), None, 0.0, 0)
        torch.cuda.synchronize()
        err = (y.double().cpu() - cur).abs().max().item()
        assert err < 2e-5 * max(1.0, cur.abs().max().item()), (rows, dims, err)
        assert (ybuf[:, :2] == 7.0).all() and (ybuf[:, 2 + dims[3]:] == 7.0).all(), "writes stay inside the output slice"
        # the kernel folding BatchNorm itself (raw conv parameters + a host array of the 12 BatchNorm vectors): same bits
        y2 = torch.empty(rows, dims[3], device=dev)
        hip.call("mlp3_eval", x, x.stride(0), rows, Cin, *raw, y2, y2.stride(0),
                 torch.tensor([t.data_ptr() for t in bn], dtype=torch.int64), 1e-5, 0)
        assert torch.equal(y2, y)
        # pre = 2: the first two input columns land in front of every row's outputs (r06), nothing else moves
        if Cin >= 2:
            y3 = torch.full((rows, dims[3] + 5), 7.0, device=dev)
            hip.call("mlp3_eval", x, x.stride(0), rows, Cin, *raw, y3[:, 3:], y3.stride(0),
                     torch.tensor([t.data_ptr() for t in bn], dtype=torch.int64), 1e-5, 2)
            assert torch.equal(y3[:, 3:3 + dims[3]], y) and torch.equal(y3[:, 1:3], x[:, :2])
            assert (y3[:, 0] == 7.0).all() and (y3[:, 3 + dims[3]:] == 7.0).all()


def test_bn_backward_pair_equals_two_calls(dev):
    """ops.bn_backward_pair (st_gcn's two BatchNorms behind one ReLU share dY and the mask) gives the bits of two bn_backward calls."""
    from mmego_amd import ops
    g = torch.Generator().manual_seed(9)
    for rows, C in ((7680, 128), (7680, 32), (1000, 64), (333, 24)):           # (C = 24: not a tile multiple -> falls back to two calls)
        ar = ops.Arena(dev)
        dY, Y = torch.randn(rows, C, generator=g).to(dev), torch.randn(rows, C, generator=g).to(dev)
        Xs = [torch.randn(rows, C, generator=g).to(dev) for _ in range(2)]
        sts = []
        for i, X in enumerate(Xs):
            bn = torch.nn.BatchNorm1d(C).to(dev)
            bn.weight.data.uniform_(0.5, 1.5)
            sts.append(ops.bn_stats(ar, "bn%d" % i, X, bn, True))
        # the statistics pair gives the states (and running statistics) of two single calls
        bns = [torch.nn.BatchNorm1d(C).to(dev) for _ in range(4)]
        for b in bns:
            b.weight.data.uniform_(0.5, 1.5)
        for i in (0, 1):
            bns[2 + i].load_state_dict(bns[i].state_dict())
        sa = [ops.bn_stats(ar, "s%d" % i, Xs[i], bns[i], True) for i in (0, 1)]
        sb = ops.bn_stats_pair(ar, "p0", Xs[0], bns[2], "p1", Xs[1], bns[3], True)
        for i in (0, 1):
            assert torch.equal(sa[i].all, sb[i].all), (rows, C)
            assert torch.equal(bns[i].running_mean, bns[2 + i].running_mean) and torch.equal(bns[i].running_var, bns[2 + i].running_var)
        single = [[torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(rows, C, device=dev)] for _ in range(2)]
        for X, st, (dg, db, dX) in zip(Xs, sts, single):
            ops.bn_backward(dY, Y, X, st, dg, db, dX)
        pair = [[torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(rows, C, device=dev)] for _ in range(2)]
        ops.bn_backward_pair(dY, Y, Xs[0], sts[0], *pair[0], Xs[1], sts[1], *pair[1])
        for a, b in zip(single, pair):
            for x, y in zip(a, b):
                assert torch.equal(x, y), (rows, C)


def test_weight_gradient_with_bias_gradient_in_one_launch(dev):
    """ops.grad_weight(dY, X, dW, db): the bias gradient (column sums of dY) comes out of the weight-gradient product itself where
    that runs on the K-quartered small-product kernel (unsplit and split-K), and from a column-sum launch otherwise."""
    from mmego_amd import ops
    g = torch.Generator().manual_seed(5)
    for rows, N, K in ((512, 87, 128), (7680, 64, 32), (7680, 128, 64), (7680, 64, 3), (1000, 42, 64), (32768, 64, 64)):
        dY = torch.randn(rows, N + 3, generator=g).to(dev)[:, 1:N + 1]            # column slice: row stride N + 3
        X = torch.randn(rows, K, generator=g).to(dev)
        dW, db = torch.full((N, K), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
        ops.grad_weight(dY, X, dW, db=db)
        ref_w = (dY.double().t() @ X.double())
        ref_b = dY.double().sum(0)
        assert torch.allclose(dW.double(), ref_w, rtol=1e-5, atol=2e-4 * max(1.0, ref_w.abs().max().item())), (rows, N, K)
        assert torch.allclose(db.double(), ref_b, rtol=1e-5, atol=1e-5 * max(1.0, dY.abs().sum(0).max().item())), (rows, N, K)
        dW2 = torch.empty_like(dW)
        ops.grad_weight(dY, X, dW2)                                               # without db: the same product bits
        assert torch.equal(dW2, dW)


def test_lstm64_fused_dropout_and_bias_pair(dev):
    """nn.LSTM(dropout=0.1)'s inter-layer dropout as applied by the layer kernel while it stores its outputs: masks are 0 or
    1/(1-p) with the right rate, differ per layer, are a function of the seed counter alone and change when the net's
    once-per-forward tick advances it; the layer outputs themselves are those of the dropout-free pass on the same inputs.
    Backward: the mask on the input gradient (product epilogue) and the four bias gradients of a layer in one launch."""
    from mmego_amd import blocks, hip, ops
    torch.manual_seed(3)
    lstm = blocks.LstmParams(64, 64, 3, dropout=0.1).to(dev)
    B, T, p = 64, 8, 0.1
    x = torch.randn(B * T, 64, device=dev)
    seed = torch.tensor([12345], dtype=torch.int64, device=dev)
    ar = ops.Arena(dev)
    blocks.lstm64_forward(ar, "k", lstm, x, B, T, None, None, True, p, seed)
    masks = []
    for l in (0, 1):
        o, d, m = (ar.get("k.%s%d" % (n, l), (B * T, 128)) for n in ("out", "do", "mk"))
        assert torch.equal(d, o * m)
        keep = torch.tensor(1.0 / (1.0 - p), device=dev)
        assert ((m == 0) | (m == keep)).all()
        rate = (m == 0).float().mean().item()
        assert abs(rate - p) < 0.01, rate                       # 65536 draws: sigma = 0.0012
        masks.append(m.clone())
    assert not torch.equal(masks[0], masks[1])
    ar2 = ops.Arena(dev)
    blocks.lstm64_forward(ar2, "k", lstm, x, B, T, None, None, True, p, seed)         # same counter: same masks
    assert torch.equal(ar2.get("k.mk0", (B * T, 128)), masks[0])
    hip.call("inc_i64", None, 0, seed)                                              # the tick of a training forward
    blocks.lstm64_forward(ar2, "k", lstm, x, B, T, None, None, True, p, seed)
    assert not torch.equal(ar2.get("k.mk0", (B * T, 128)), masks[0])
    ar3 = ops.Arena(dev)
    blocks.lstm64_forward(ar3, "k", lstm, x, B, T, None, None, True, 0.0, seed)      # layer 0 sees the same input either way
    assert torch.equal(ar3.get("k.out0", (B * T, 128)), ar.get("k.out0", (B * T, 128)))
    # backward pieces
    dg = torch.randn(B * T, 512, device=dev)
    outs = [torch.zeros(256, device=dev) for _ in range(4)]
    hip.call("colsum_pair", dg, 512, B * T, 256, outs[0], outs[1], outs[2], outs[3], 0)
    ref = dg.double().sum(0).float()
    assert torch.allclose(outs[0], ref[:256], rtol=1e-6, atol=1e-5) and torch.equal(outs[0], outs[1])
    assert torch.allclose(outs[2], ref[256:], rtol=1e-6, atol=1e-5) and torch.equal(outs[2], outs[3])
    W = torch.randn(256, 128, device=dev) * 0.1
    dx, dx2 = torch.randn(B * T, 128, device=dev), None
    dx2 = dx.clone()
    ops.grad_input(dg[:, 256:], W, dx, accumulate=True, cmul=masks[0])
    ops.grad_input(dg[:, 256:], W, dx2, accumulate=True)
    assert torch.equal(dx, dx2 * masks[0])


@pytest.mark.parametrize("B,T,p", [(64, 8, 0.1), (13, 5, 0.0), (16, 3, 0.1)])
def test_lstm64_stacks_side_by_side_are_the_separate_launches(dev, B, T, p):
    """mmego_lstm64_forward_multi / _backward_multi (UpperNetwlocal's global and anchor stacks in one launch per layer, grid z = stack):
    every output, stash, dropout mask, input gradient and weight gradient bit-identical to the stacks run one after the other; the
    last layers may write the halves of a wider tensor (the nets' concatenation)."""
    from mmego_amd import blocks, ops
    torch.manual_seed(11)
    lstms = [blocks.LstmParams(64, 64, 3, dropout=0.1).to(dev) for _ in range(2)]
    xs = [torch.randn(B * T, 64, device=dev) for _ in range(2)]
    h0 = [torch.randn(6, B, 64, device=dev) * 0.3 for _ in range(2)]
    c0 = [torch.randn(6, B, 64, device=dev) * 0.3 for _ in range(2)]
    seed = torch.tensor([777], dtype=torch.int64, device=dev)
    douts = [torch.randn(B * T, 128, device=dev) for _ in range(2)]

    def grads():
        store = {}
        def G(t):
            return store.setdefault(t.data_ptr(), torch.zeros_like(t))
        return store, G

    ar1, ar2 = ops.Arena(dev), ops.Arena(dev)
    sep = [blocks.lstm64_forward(ar1, "s%d" % i, lstms[i], xs[i], B, T, h0[i], c0[i], True, p, seed, salt=i) for i in range(2)]
    cat = torch.zeros(B * T, 256, device=dev)
    both = blocks.lstm64_forward_multi(ar2, [("s%d" % i, lstms[i], xs[i], h0[i], c0[i], p, seed, i) for i in range(2)], B, T, True,
                                       last_out=(cat[:, :128], cat[:, 128:]))
    for i in range(2):
        assert torch.equal(cat[:, 128 * i:128 * i + 128], sep[i][0])
        assert torch.equal(both[i][1], sep[i][1]) and torch.equal(both[i][2], sep[i][2])
        for l in range(3):
            for n, shape in (("g", (2, T, B, 256)), ("c", (2, T, B, 64)), ("hp", (2, B * T, 64))) + ((("mk", (B * T, 128)),) if p > 0 and l < 2 else ()):
                key = "s%d.%s%d" % (i, n, l)
                # (the per-layer buffers the stacks' batched products touch are the halves of one buffer in the paired form)
                got = ar2.get("s0+s1.%s%d" % (n, l), (2,) + shape)[i] if n == "mk" else ar2.get(key, shape)
                assert torch.equal(ar1.get(key, shape), got), key
    st1, G1 = grads()
    st2, G2 = grads()
    dx_sep = [blocks.lstm64_backward(ar1, "s%d" % i, lstms[i], xs[i], B, T, c0[i], douts[i], G1, p, True) for i in range(2)]
    dx_both = blocks.lstm64_backward_multi(ar2, [("s%d" % i, lstms[i], xs[i], c0[i], douts[i], p) for i in range(2)], B, T, G2, True)
    for i in range(2):
        assert torch.equal(dx_sep[i], dx_both[i])
    assert st1.keys() == st2.keys() and len(st1) == 2 * 3 * 2 * 4
    for k in st1:
        assert torch.equal(st1[k], st2[k])
    # eval form (no stashes, no dropout)
    ar3 = ops.Arena(dev)
    ev = blocks.lstm64_forward_multi(ar3, [("s%d" % i, lstms[i], xs[i], None, None, 0.0, None, i) for i in range(2)], B, T, False)
    for i in range(2):
        ref = blocks.lstm64_forward(ops.Arena(dev), "e", lstms[i], xs[i], B, T, None, None, False, 0.0, None)
        assert torch.equal(ev[i][0], ref[0]) and torch.equal(ev[i][1], ref[1])


def _seq(real16, i, device=None):
    x = T(real16["x"][i:i + 1]).clone()
    tgt = T(real16["target"][i:i + 1])
    skl, R = T(real16["skl"][i:i + 1]), T(real16["R"][i:i + 1])
    t = tgt[:, :, 20].contiguous()
    out = [x, skl, R, t, tgt]
    return [o.to(device) for o in out] if device is not None else out


@pytest.fixture(scope="module")
def pretrained(dev):
    from mmego_amd import nets
    wu, wl = golden("w_upper_pretrained.npz"), golden("w_lower_pretrained.npz")
    ou, ol = load_weights(on.UpperNet(), wu).eval(), load_weights(on.LowerNet(64), wl).eval()
    hu, hl = load_weights(nets.UpperNet(), wu).to(dev).eval(), load_weights(nets.LowerNet(64), wl).to(dev).eval()
    return ou, ol, hu, hl


def test_transform_bit_exact(dev, real16):
    from mmego_amd import ops
    x, skl, R, t, _ = _seq(real16, 3)
    ref = x.clone()
    geo.transform_to_head_(ref, R, t)
    xd = x.to(dev)
    ops.transform2h_(xd, R.to(dev), t.to(dev))
    assert torch.equal(xd.cpu(), ref), "Transform2H must be bit-identical to the CPU path (it feeds sort keys)"
    g1 = golden("g1_transforms.npz")
    p = T(g1["pts"]).to(dev)
    ops.transform2h_(p, T(g1["R"]).to(dev), T(g1["t"]).to(dev))
    assert torch.equal(p.cpu(), T(g1["mutated"]))
    # the same launch fed from a source tensor (whole rows), with the two copies Upper_Net's forward takes behind the transform
    src = x.to(dev)
    dst = torch.full_like(src, float("nan"))
    keep = torch.full_like(src, float("nan"))
    feats = torch.full((src.numel() // src.shape[-1], 28), float("nan"), device=dev)
    ops.transform2h_(dst, R.to(dev), t.to(dev), src=src, keep=keep, feats=feats, nfeat=4)
    assert torch.equal(src.cpu(), x), "the source is left alone"
    assert torch.equal(dst.cpu(), ref) and torch.equal(keep.cpu(), ref)
    assert torch.equal(feats[:, :4].cpu(), ref.view(-1, ref.shape[-1])[:, :4]) and torch.isnan(feats[:, 4:]).all()


def test_transforms_through_utils_surface(dev):
    """mmego_amd.utils.Transform2H / Transform2R (the drop-in surface of Util/Universal_Util/Utils.py:274-292) against the
    reference's own outputs (golden G1): the in-place quirk Q1, the returned view, and the head -> world transform."""
    from mmego_amd import utils
    g1 = golden("g1_transforms.npz")
    Rg, tg = T(g1["R"]).to(dev), T(g1["t"]).to(dev)
    B_, T_, P_ = g1["pts"].shape[:3]
    pts = T(g1["pts"]).to(dev)
    out_h = utils.Transform2H(pts, B_, T_, P_, Rg, tg)
    assert out_h.shape == (B_ * T_, P_, 6) and out_h.data_ptr() == pts.data_ptr(), "Transform2H returns a view of its input"
    assert torch.equal(out_h.cpu(), T(g1["out_h"])) and torch.equal(pts.cpu(), T(g1["mutated"]))
    joints = T(g1["joints"]).to(dev)
    keep = joints.clone()
    out_r = utils.Transform2R(joints, B_, T_, joints.shape[2], Rg, tg)
    assert out_r.shape == tuple(g1["out_r"].shape) and torch.equal(joints, keep), "Transform2R returns a new tensor"
    assert torch.allclose(out_r.cpu(), T(g1["out_r"]), rtol=0, atol=1e-6), (out_r.cpu() - T(g1["out_r"])).abs().max()
    assert torch.allclose(out_r.cpu(), geo.transform_to_world(T(g1["joints"]), T(g1["R"]), T(g1["t"])), rtol=0, atol=1e-6)
    with pytest.raises(RuntimeError):
        utils.Transform2R(T(g1["joints"]), B_, T_, joints.shape[2], T(g1["R"]), T(g1["t"]))      # no CPU fallback


def test_eval_forward_pretrained(dev, real16, pretrained):
    ou, ol, hu, hl = pretrained
    g4, g5 = golden("g4_upper_eval.npz"), golden("g5_lower_eval.npz")
    rows_h, rows_o, rows_p = [], [], []
    with torch.no_grad():
        for i in range(16):
            x, skl, R, t, tgt = _seq(real16, i)
            xd, skld, Rd, td, _ = _seq(real16, i, dev)
            h0, c0 = ot.zeros_state(1)
            lo_, qo, wo, hno, cno = ou(x, h0, c0, skl, R, t)
            lh, qh, wh, hnh, cnh = hu(xd, h0.to(dev), c0.to(dev), skld, Rd, td)
            assert torch.equal(xd.cpu(), x), "Q1: the caller's tensor holds the transformed points"
            for name, a, b in (("l", lh, lo_), ("q", qh, qo), ("w", wh, wo), ("hn", hnh, hno), ("cn", cnh, cno)):
                assert torch.allclose(a.cpu(), b, rtol=1e-4, atol=2e-5), (name, i, (a.cpu() - b).abs().max())
            if i < 8:
                assert torch.allclose(lh.cpu(), T(g4["l_%d" % i]), rtol=1e-4, atol=2e-5), "vs the real reference"
            llo, lqo = ol(lo_.clone(), x, h0, c0, h0, c0, skl, R, t)
            llh, lqh = hl(lh.clone(), xd, None, None, None, None, skld, Rd, td)
            assert torch.equal(xd.cpu(), x), "Q1: second in-place transform"
            assert torch.equal(hl.last_select_idx.cpu(), ol.last_select_idx), "top-64 indices bit-exact"
            assert torch.allclose(llh.cpu(), llo, rtol=1e-4, atol=2e-5), (i, (llh.cpu() - llo).abs().max())
            assert torch.allclose(lqh.cpu(), lqo, rtol=1e-4, atol=5e-5), (i, (lqh.cpu() - lqo).abs().max())
            rows_h.append(om.batch_errors(lh.cpu(), llh.cpu(), tgt))
            rows_o.append(om.batch_errors(lo_, llo, tgt))
            # the reference's own (unstable-sort) top-64 choice replayed on the HIP path: Lower_Net vs the REFERENCE's outputs
            xp_ = xd.clone()
            hu_x = T(real16["x"][i:i + 1]).clone().to(dev)
            lh2 = hu(hu_x, h0.to(dev), c0.to(dev), skld, Rd, td)[0]
            llp, lqp = hl(lh2.clone(), hu_x, None, None, None, None, skld, Rd, td, pin_select_idx=T(real16["ref_sel_idx"][i]))
            assert torch.equal(hu_x, xp_), "pinned run sees the same doubly transformed points"
            assert torch.equal(hl.last_select_idx.cpu(), T(real16["ref_sel_idx"][i]))
            if i < 8:
                assert torch.allclose(llp.cpu(), T(g5["l_%d" % i]), rtol=1e-4, atol=2e-5), ("pinned l vs reference", i)
                assert torch.allclose(lqp.cpu(), T(g5["q_%d" % i]), rtol=1e-4, atol=5e-5), ("pinned q vs reference", i)
            rows_p.append(om.batch_errors(lh.cpu(), llp.cpu(), tgt))
    sh, so, sp = om.summarize(rows_h), om.summarize(rows_o), om.summarize(rows_p)
    for k in ("all_cm", "upper_cm", "lower_cm"):
        assert abs(sh[k] - so[k]) < 1e-3, (k, sh[k], so[k])          # north star: error equal within 1e-3 cm
    assert np.allclose(sh["per_joint_cm"], so["per_joint_cm"], atol=1e-3)
    g9 = golden("g9_end2end.npz")
    assert abs(sh["upper_cm"] - float(g9["upper_cm"])) < 1e-3                # vs the real reference (no tie dependence)
    assert abs(sh["lower_cm"] - float(g9["lower_cm"])) < 0.05               # reference tie order differs (see oracle tests)
    # ... and with the reference's tie choices replayed, the HIP path meets the 1e-3 cm bar against the REFERENCE directly
    for k in ("all_cm", "upper_cm", "lower_cm"):
        assert abs(sp[k] - float(g9[k])) < 1e-3, (k, sp[k], float(g9[k]))
    assert np.allclose(sp["per_joint_cm"], g9["per_joint_cm"], atol=1e-3)
    assert abs(sp["rot_deg"] - float(g9["rot_deg"])) < 1e-2


def test_imu_forward(dev):
    from mmego_amd import nets
    g = golden("g7_imu.npz")
    imu = T(g["imu"])
    small = load_weights(nets.IMUNet(15, 9, 32, 2, True, 0.1), g, "small.w.").to(dev).eval()
    with torch.no_grad():
        R, t = small(imu.to(dev))
    assert torch.allclose(R.cpu(), T(g["small.R"]), atol=5e-6) and torch.allclose(t.cpu(), T(g["small.t"]), atol=5e-6)
    torch.manual_seed(703)
    big = nets.IMUNet(15, 9, 512, 2, True, 0.1)
    for k in ("fc1.weight", "rnn_fast.weight_hh_l1_reverse", "rnn_slow.weight_ih_l0"):
        v = big.state_dict()[k].double()
        assert abs(v.sum().item() - g["big.chk." + k][0]) < 1e-9, "seeded init must equal the reference's"
    big = big.to(dev).eval()
    with torch.no_grad():
        R, t = big(imu.to(dev))
    assert torch.allclose(R.cpu(), T(g["big.R"]), atol=2e-5) and torch.allclose(t.cpu(), T(g["big.t"]), atol=2e-5)
    # 64 rows in the fast LSTM (the small-batch step kernel) against the oracle
    torch.manual_seed(9)
    imu2 = torch.randn(8, 8, 20, 15)
    ob = on.IMUNet(15, 9, 512, 2, True, 0.1).eval()
    ob.load_state_dict({k: v.cpu() for k, v in big.state_dict().items()})
    with torch.no_grad():
        Ro, to_ = ob(imu2)
        Rh, th = big(imu2.to(dev))
    assert torch.allclose(Rh.cpu(), Ro, atol=2e-5) and torch.allclose(th.cpu(), to_, atol=2e-5)
    # >= 128 rows: the LDS-DMA step kernel, the persistent / tile GEMMs; 200 rows = 3 full 64-row blocks + a ragged one of 8
    for Bq, Tq in ((25, 8), (16, 8)):
        imu3 = torch.randn(Bq, Tq, 20, 15)
        with torch.no_grad():
            Ro, to_ = ob(imu3)
            Rh, th = big(imu3.to(dev))
        assert torch.allclose(Rh.cpu(), Ro, atol=2e-5) and torch.allclose(th.cpu(), to_, atol=2e-5), (Bq, Tq)


def test_lstm_step_kernel_variants(dev):
    """The recurrent-step kernels (LDS-DMA kernel for >= 128 rows, W_hh-streaming kernel below) give the same h and c as a plain
    torch LSTM cell step in fp64: 200 rows (ragged last block), the first step (h = c = 0), the stashing mode, small batches."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, torch
sys.path.insert(0, %r)
from mmego_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
Bn, H = int(os.environ.get("T_BN", "200")), int(os.environ.get("T_H", "512"))
w = [torch.randn(4 * H, H, device=dev) * 0.04 for _ in range(2)]
b = [torch.randn(4 * H, device=dev) * 0.1 for _ in range(2)]
xp = torch.randn(2, Bn, 4 * H, device=dev)
h0 = torch.randn(2, Bn, H, device=dev) * 0.5
c0 = torch.randn(2, Bn, H, device=dev) * 0.5
for first in (0, 1):
    c = c0.clone() if not first else torch.zeros_like(c0)
    hout = torch.zeros(2, Bn, H, device=dev)
    gst = torch.zeros(2, Bn, 4 * H, device=dev); cst = torch.zeros(2, Bn, H, device=dev)
    hip.call("lstm_step", 2, Bn, H, first, None if first else h0[0], None if first else h0[1], H, w[0], w[1], b[0], b[1],
             xp[0], xp[1], 4 * H, hout[0], hout[1], H, c[0], c[1], gst[0], gst[1], cst[0], cst[1])
    torch.cuda.synchronize()
    for d in range(2):
        hp = torch.zeros(Bn, H, device=dev, dtype=torch.float64) if first else h0[d].double()
        cp = torch.zeros(Bn, H, device=dev, dtype=torch.float64) if first else c0[d].double()
        g = xp[d].double() + b[d].double() + hp @ w[d].double().t()
        i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2*H]), torch.tanh(g[:, 2*H:3*H]), torch.sigmoid(g[:, 3*H:])
        cn = f * cp + i * gg
        hn = o * torch.tanh(cn)
        assert (c[d].double() - cn).abs().max().item() < 2e-5, ("c", first, d)
        assert (hout[d].double() - hn).abs().max().item() < 2e-5, ("h", first, d)
        assert (cst[d].double() - cn).abs().max().item() < 2e-5 and (gst[d][:, :H].double() - i).abs().max().item() < 2e-5
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    # small batches (rnn_slow: Bn < 128 -> lstm_step_small_kernel), ragged rows, H a multiple of 64 and of 32 only
    for bn, h in (("100", "512"), ("64", "256"), ("37", "96")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, T_BN=bn, T_H=h), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (bn, h, r.stdout[-500:], r.stderr[-1500:])


def test_upper_front_eval_fused_equals_chain(dev, monkeypatch):
    """mmego_upper_front_eval (Transform2H + PointNet + concat + GlobalPointNet + attention pooling of an eval-mode Upper_Net in ONE
    launch, front.hip) against the four-launch chain it replaces (transform2h, 2 x mlp3_eval, attn_pool_forward) and against fp64
    torch: the in-place transformed points bit-identical (they are Lower_Net's sort keys), pooled features / attention weights /
    joints to 1e-5; N = 128, 256 and a ragged-in-waves 48; with and without a separate source tensor; non-trivial BatchNorm
    statistics."""
    from mmego_amd import nets
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(78)
    net = nets.UpperNet()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.3)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.3)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.2)
    net = net.to(dev).eval()
    for B, Tn, N in ((3, 4, 128), (2, 3, 256), (5, 2, 48), (70, 8, 128)):
        x0 = torch.randn(B, Tn, N, 6, generator=g)
        x0[torch.rand(B, Tn, N, generator=g) < 0.3] = 0.0
        R = torch.linalg.qr(torch.randn(B, Tn, 3, 3, generator=g))[0].contiguous()
        t = torch.randn(B, Tn, 3, generator=g) * 0.2
        body = torch.randn(B, 20, 3, generator=g) * 0.2
        h0 = torch.zeros(6, B, 64, device=dev)
        outs = {}
        for fused in (False, True):
            monkeypatch.setattr(nets, "_FUSED_FRONT", fused)
            for with_src in (False, True):
                x = x0.clone().to(dev)
                with torch.no_grad():
                    if with_src:
                        xs = x0.clone().to(dev)
                        x.fill_(float("nan"))
                        res = net._forward_impl(x, h0, h0.clone(), body.to(dev), R.to(dev), t.to(dev), stash=False, x_src=xs)
                        assert torch.equal(xs.cpu(), x0), "the source tensor is read only"
                    else:
                        res = net(x, h0, h0.clone(), body.to(dev), R.to(dev), t.to(dev))
                torch.cuda.synchronize()
                outs[(fused, with_src)] = (x.cpu(), res[0].cpu(), res[2].cpu(), net.arena("eval").get("vec", (B * Tn, 64)).cpu().clone())
        ref = outs[(False, False)]
        for key, got in outs.items():
            assert torch.equal(got[0], ref[0]), ("transformed points must be bit-identical", key, N)
            assert torch.allclose(got[1], ref[1], rtol=1e-5, atol=1e-5), (key, N, (got[1] - ref[1]).abs().max())
            assert torch.allclose(got[2], ref[2], rtol=1e-4, atol=1e-7), (key, N, (got[2] - ref[2]).abs().max())
            assert torch.allclose(got[3], ref[3], rtol=1e-5, atol=1e-5), (key, N, (got[3] - ref[3]).abs().max())
        # fp64 torch restatement of the front end on the transformed points
        xt = ref[0].double().view(B * Tn, N, 6)
        sd = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}

        def mlp(z, pre):
            for i in (1, 2, 3):
                z = z @ sd[pre + "conv%d.weight" % i][:, :, 0].t() + sd[pre + "conv%d.bias" % i]
                z = (z - sd[pre + "cb%d.running_mean" % i]) / torch.sqrt(sd[pre + "cb%d.running_var" % i] + 1e-5) \
                    * sd[pre + "cb%d.weight" % i] + sd[pre + "cb%d.bias" % i]
                z = torch.relu(z)
            return z
        feats = torch.cat((xt[..., :4], mlp(xt, "module0.")), dim=-1)
        g3 = mlp(feats, "module1.gpointnet.")
        a = torch.softmax(g3 @ sd["module1.gpointnet.attn.weight"].t() + sd["module1.gpointnet.attn.bias"], dim=1)
        want_vec = (g3 * a).sum(1)
        got = outs[(True, False)]
        assert (got[3].double() - want_vec).abs().max().item() < 2e-5 * max(1.0, want_vec.abs().max().item()), N
        assert (got[2].double().view_as(a) - a).abs().max().item() < 1e-6, N


def test_lstm_recurrence_two_chains_equal_one_launch_per_step(dev, monkeypatch):
    """blocks.lstm_recurrence: the two directions as two chains of single-direction launches on two streams (64 x 16 tiles, the
    default from 128 rows) give bit-identical h_t, final c and backward stashes to both directions in one launch per timestep
    (64 x 32 tiles) -- full and ragged row blocks, eagerly and as a replayed HIP graph -- and match fp64 arithmetic."""
    from mmego_amd import blocks, ops
    H = 512
    torch.manual_seed(41)
    lstm = blocks.LstmParams(H, H, 1).to(dev)
    for Bn, Tn, stash in ((512, 20, False), (200, 7, True), (130, 3, False)):
        xp = torch.randn(Bn * Tn, 8 * H, generator=torch.Generator().manual_seed(100 * Bn)).to(dev)
        res = []
        for two, graph in ((True, False), (False, False), (True, True)):
            monkeypatch.setattr(blocks, "_LSTM_TWO_CHAINS", two)
            ar = ops.Arena(dev)
            out = ar.get("out", (Bn * Tn, 2 * H), zero=True)
            gst = ar.get("gst", (2, Tn, Bn, 4 * H)) if stash else None
            cst = ar.get("cst", (2, Tn, Bn, H)) if stash else None
            run = lambda: blocks.lstm_recurrence(ar, "k", lstm, 0, xp, out, Bn, Tn, gst=gst, cst=cst)
            if graph:
                run()
                torch.cuda.synchronize()
                ops.fill(out, 0.0)
                g = torch.cuda.CUDAGraph()
                with ops.capture(g):
                    run()
                g.replay()
            else:
                run()
            torch.cuda.synchronize()
            res.append((out.clone(), ar.get("k.c", (2, Bn, H)).clone(), None if gst is None else gst.clone(),
                        None if cst is None else cst.clone()))
        for other in res[1:]:
            for a, b in zip(res[0], other):
                if a is not None:
                    assert torch.equal(a, b), (Bn, Tn, stash, (a - b).abs().max().item())
    Bn, Tn = 256, 5
    xp = (torch.randn(Bn * Tn, 8 * H, generator=torch.Generator().manual_seed(7)) * 0.5).to(dev)
    ar = ops.Arena(dev)
    monkeypatch.setattr(blocks, "_LSTM_TWO_CHAINS", True)
    out = ar.get("out", (Bn * Tn, 2 * H))
    blocks.lstm_recurrence(ar, "k", lstm, 0, xp, out, Bn, Tn)
    x3 = xp.double().view(Bn, Tn, 2, 4 * H).cpu()
    want = torch.zeros(Bn, Tn, 2, H, dtype=torch.float64)
    for d in range(2):
        W, b = lstm.w("weight_hh", 0, d).detach().double().cpu(), lstm.w("bias_hh", 0, d).detach().double().cpu()
        h = torch.zeros(Bn, H, dtype=torch.float64)
        c = torch.zeros(Bn, H, dtype=torch.float64)
        for s_ in range(Tn):
            t_ = s_ if d == 0 else Tn - 1 - s_
            g_ = x3[:, t_, d] + b + h @ W.t()
            i_, f_, gg, o_ = g_.split(H, dim=1)
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(gg)
            h = torch.sigmoid(o_) * torch.tanh(c)
            want[:, t_, d] = h
    assert (out.double().cpu().view(Bn, Tn, 2, H) - want).abs().max().item() < 2e-5


def _train_pair(tag, seed, octor, hctor, dev):
    torch.manual_seed(seed)
    o = octor()
    torch.manual_seed(seed)
    h = hctor()
    for (ko, vo), (kh, vh) in zip(o.state_dict().items(), h.state_dict().items()):
        assert ko == kh and torch.equal(vo, vh), "same seed -> same initial weights: " + ko
    set_lstm_dropout(o, 0.0)
    set_lstm_dropout(h, 0.0)
    return o.train(), h.to(dev).train()


def _compare_training(tag, o, h, fwd_o, fwd_h, target, g, dev, later_grad_tol=2e-4, later_out_atol=2e-5, resync=False, near_tie=None,
                      tie_grad_tol=5e-3, tie_params=None):
    """Three train steps on both sides.  ``resync``: after every optimiser step the oracle's parameters and BatchNorm buffers
    are copied into the HIP net, so steps 2 and 3 start from bit-equal states and are held to the step-1 bar -- this separates
    "the two runs drifted apart by sign-of-noise Adam updates" from "the backward is wrong from step 2 on".
    ``near_tie(h)`` (optional): True when this step's forward left a ReLU pre-activation within fp32 rounding of zero in a SMALL
    tensor (one of a few thousand elements): which side of zero it lands on depends on the summation order, the mask element it
    switches carries a whole gradient entry, and the parameters UPSTREAM of that mask (``tie_params``: a regex on their names; the
    gradients of everything downstream do not pass through it and keep the normal bar) are then held to
    ``tie_grad_tol`` instead."""
    from mmego_amd.params import FusedAdam
    opt_o = torch.optim.Adam(o.parameters(), lr=3e-5)
    opt_h = FusedAdam(h.flat(), lr=3e-5)
    tgt_d = target.to(dev)
    for step in (1, 2, 3):
        opt_o.zero_grad()
        lo_ = fwd_o(o)
        loss_o = ot.l1_sum(lo_, target)
        loss_o.backward()
        lh = fwd_h(h)
        tie = bool(near_tie(h)) if near_tie is not None else False
        loss_h = (lh - tgt_d).abs().sum()
        loss_h.backward()
        assert abs(loss_h.item() - loss_o.item()) < 2e-5 * abs(loss_o.item()), (step, loss_h.item(), loss_o.item())
        assert abs(loss_h.item() - float(g["%s.loss%d" % (tag, step)])) < 2e-4 * abs(loss_o.item()), "vs the real reference"
        # (from step 2 on the two runs start from weights that differ by sign-of-noise Adam updates: see later_grad_tol)
        assert torch.allclose(lh.detach().cpu(), lo_.detach(), rtol=1e-4, atol=2e-5 if step == 1 else later_out_atol), \
            (step, (lh.detach().cpu() - lo_.detach()).abs().max())
        po, ph = dict(o.named_parameters()), dict(h.named_parameters())
        scale = max(p.grad.abs().max().item() for p in po.values() if p.grad is not None)
        for k in po:
            go = po[k].grad if po[k].grad is not None else torch.zeros_like(po[k])
            gh = ph[k].grad.cpu()
            err = (gh - go).abs().max().item()
            # after the first Adam step the two runs no longer hold bit-equal weights (sign-of-noise updates), so
            # later steps compare two slightly different points of an ill-conditioned BatchNorm chain
            tol = 2e-4 if step == 1 else later_grad_tol
            wide = tie and (tie_params is None or re.search(tie_params, k) is not None)
            assert err < (max(tol, tie_grad_tol) if wide else tol) * scale, (tag, step, k, err, scale, tie)
        if step == 1:
            grads = [(k, p.grad) for k, p in ph.items()]
            check_pinned(g, "%s.grad." % tag, grads, rtol=5e-3, atol=5e-3 * scale)     # vs the real reference
        if resync:
            # Parameters whose true gradient is exactly zero (a bias in front of a batch-statistics BatchNorm, ...): the oracle's
            # value there is rounding noise, its first Adam steps are +-lr by the SIGN of that noise, and the CPU reductions behind
            # it are multi-threaded -- the oracle's trajectory then differs from run to run (measured: the parameter checksum
            # after step 1 changed in every one of 24 runs, and one trajectory in five lands where this ill-conditioned chain
            # amplifies fp32 rounding beyond the step-1 bar).  Zeroed here, so that every run takes the same trajectory.
            for k in po:
                if NOISE_GRAD.search(k) and po[k].grad is not None:
                    po[k].grad.zero_()
        opt_o.step()
        opt_h.step()
        if resync:
            h.load_state_dict({k: v.to(dev) for k, v in o.state_dict().items()})
            for k, v in o.state_dict().items():      # the next step really starts from the oracle's state
                assert torch.equal(h.state_dict()[k].cpu(), v), ("resync did not take", tag, step, k)
            continue
        n_bad = n_all = 0
        for k in po:
            if NOISE_GRAD.search(k):
                continue
            d = (ph[k].detach().cpu() - po[k].detach()).abs()
            assert d.max().item() <= 6e-5 * step + 2e-6, (tag, step, k, d.max().item())      # at most a +-lr flip per step
            n_bad += int((d > 2e-6).sum())
            n_all += d.numel()
        # elements whose gradient is ~0 take a sign-of-rounding-noise Adam step; counted over the whole net
        assert n_bad < 0.05 * n_all, (tag, step, n_bad, n_all)
        for (ko, bo), (kh, bh) in zip(o.named_buffers(), h.named_buffers()):
            # running stats absorb the (rounding-noise driven, +-lr per step) drift of the zero-gradient conv biases
            assert torch.allclose(bh.cpu().float(), bo.float(), rtol=1e-3, atol=1e-4 * step), (tag, step, ko)


def test_projection_product_on_320x256_tiles_is_the_128x128_walk_bit_for_bit(dev):
    """gemm_tile_big_kernel (gemm_tile.hip, r06: 320 x 256 tiles, operands by LDS-DMA into a swizzled unpadded LDS image; what
    mmego_gemm picks for IMU_Net's batched BiLSTM input projections, Net/IMU_Net.py:58-62) against the 128 x 128 persistent walk:
    the same product with 128 extra rows (10 368 is not a multiple of 320, so the dispatch falls back) must give the first 10 240
    rows BIT FOR BIT -- same k permutation, same accumulation order -- for both layers' K, and both agree with float64."""
    from mmego_amd import ops
    g = torch.Generator().manual_seed(11)
    M, N = 10240, 2048
    for K in (512, 1024):
        A = torch.randn(M + 128, K, generator=g).to(dev)
        W = (torch.randn(2 * N, K, generator=g) * 0.05).to(dev)
        b = torch.randn(2 * N, generator=g).to(dev)
        c_big = torch.full((M, 2 * N), float("nan"), device=dev)
        c_walk = torch.full((M + 128, 2 * N), float("nan"), device=dev)
        ops.linear_pair(A[:M], W[:N], W[N:], b[:N], b[N:], c_big, N)
        ops.linear_pair(A, W[:N], W[N:], b[:N], b[N:], c_walk, N)
        assert torch.isfinite(c_big).all()
        assert torch.equal(c_big, c_walk[:M]), (K, float((c_big - c_walk[:M]).abs().max()))
        rows = torch.tensor([0, 31, 32, 159, 160, 319, 320, 5119, 5120, 10239], device=dev)
        ref = A[rows].double() @ W.double().t() + b.double()
        assert float((c_big[rows].double() - ref).abs().max()) < 2e-5 * (K / 512) ** 0.5, K


@pytest.mark.gpu
@pytest.mark.parametrize("F,aligned", [(1, True), (37, True), (512, True), (37, False)])
def test_cross_attention_forward_against_float64(dev, F, aligned):
    """mmego_cross_attn_forward (Lower_Net.py:95-136's FusionModule: softmax(Q K^T scale) V over 64 queries x 15 keys x 64 channels per
    frame) against float64: the fp32 matrix-pipe kernel (r06; 16-byte aligned operands) and the scalar kernel it falls back to (here:
    K / V / O shifted by one float).  Both keep fp32 accuracy: 2e-6 of the largest output; the probabilities sum to one."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(F)
    Q = torch.randn(F * 64, 64, generator=g).to(dev)
    off = 0 if aligned else 1
    KV = torch.randn(F * 15 * 128 + 4, generator=g).to(dev)
    Kv, Vv = KV[off:], KV[off + 64:]
    Obuf = torch.full((F * 64 * 128 + 4,), float("nan"), device=dev)
    O = Obuf[off + 64:]
    P = torch.zeros(F * 64, 15, device=dev)
    scale = 0.125
    hip.call("cross_attn_forward", Q, Kv, Vv, F, scale, O, 128, P, 128)
    torch.cuda.synchronize()
    Kd = KV[off:off + F * 15 * 128].view(F, 15, 128)[:, :, :64].double().cpu()
    Vd = KV[off:off + F * 15 * 128].view(F, 15, 128)[:, :, 64:].double().cpu()
    Qd = Q.view(F, 64, 64).double().cpu()
    Pw = torch.softmax(Qd @ Kd.transpose(1, 2) * scale, dim=-1)
    Ow = Pw @ Vd
    Og = Obuf[off:off + F * 64 * 128].view(F, 64, 128)[:, :, 64:].double().cpu()
    assert float((P.view(F, 64, 15).double().cpu() - Pw).abs().max()) < 2e-6
    assert float((Og - Ow).abs().max()) < 2e-6 * max(1.0, float(Ow.abs().max()))
    assert float((P.sum(dim=1) - 1).abs().max()) < 1e-5
    # nothing outside the 64 output columns of a row is written
    assert bool(torch.isnan(Obuf[off:off + F * 64 * 128].view(F, 64, 128)[:, :, :64]).all())


@pytest.mark.parametrize("F,N,keep", [(5, 64, 64), (7, 100, 64), (33, 256, 64), (3, 256, 200), (2, 600, 64),            # the rank kernel
                                      (1025, 64, 64), (1027, 100, 64), (1030, 128, 64), (1026, 200, 64), (1100, 256, 64),
                                      (1025, 256, 200), (1029, 300, 64), (1025, 512, 64), (1025, 600, 64)])
def test_topk_rows_is_the_stable_descending_sort(dev, F, N, keep):
    """mmego_topk_rows / mmego_topk_rows2 (Lower_Net.py:216-227: the 64 points with the largest x; r06: an in-register bitonic sort per
    wave for more than 1024 frames of up to 512 points, the rank kernel otherwise) against torch's STABLE descending sort -- ties by index, -0 == +0 -- with many ties in the
    keys: indices, gathered rows and the second output, all bit for bit; rows / columns outside the outputs untouched."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(N * 10 + F)
    C = 6
    x = torch.randn(F, N, C, generator=g)
    x[:, :, 0] = torch.randint(-6, 7, (F, N), generator=g).float() * 0.5          # 13 distinct keys: ties everywhere
    x[0, ::3, 0] = -0.0
    x[0, 1::3, 0] = 0.0
    xd = x.to(dev)
    out = torch.full((F * keep, C), float("nan"), device=dev)
    idx = torch.full((F, keep), -1, dtype=torch.int64, device=dev)
    out2 = torch.full((F * keep, 8), float("nan"), device=dev)
    hip.call("topk_rows2", xd, F, N, C, keep, out, idx, out2, 8, 3)
    out_b = torch.full((F * keep, C), float("nan"), device=dev)
    idx_b = torch.full((F, keep), -1, dtype=torch.int64, device=dev)
    hip.call("topk_rows", xd, F, N, C, keep, out_b, idx_b)
    torch.cuda.synchronize()
    want = torch.sort(x[:, :, 0], dim=1, descending=True, stable=True)[1][:, :keep]
    assert torch.equal(idx.cpu(), want) and torch.equal(idx_b.cpu(), want)
    rows = torch.gather(x, 1, want.unsqueeze(-1).expand(F, keep, C)).reshape(F * keep, C)
    assert torch.equal(out.cpu().view(torch.int32), rows.view(torch.int32)) and torch.equal(out_b.cpu().view(torch.int32), rows.view(torch.int32))
    assert torch.equal(out2.cpu()[:, :3].view(torch.int32), rows[:, :3].contiguous().view(torch.int32))
    assert bool(torch.isnan(out2[:, 3:]).all())


@pytest.mark.parametrize("rows,In,K", [(10240, 512, 4096), (10240, 1024, 4096), (5120, 1024, 2048), (640, 512, 4096)])
def test_input_gradient_product_in_k_slabs_on_320x256_tiles(dev, rows, In, K):
    """ops.grad_input_slabs (r06; stage-1 IMU_Net training's dX = dgates . W_ih in fp32): W^T + gemm_tile_big_kernel with K cut into
    256 / tiles slabs (grid.y; the deferred split-K format) + the streaming slab sum, against float64 -- no worse than the plain product
    of ops.grad_input; the last shape (4 tiles: 64 slabs of 64 k would be too short) must decline and leave dX alone."""
    from mmego_amd import ops
    g = torch.Generator().manual_seed(rows + In)
    dY = (torch.randn(rows, K, generator=g) * 0.1).to(dev)
    W = (torch.randn(K, In, generator=g) * 0.05).to(dev)
    dX = torch.full((rows, In), 7.0, device=dev)
    ar = ops.Arena(dev)
    took = ops.grad_input_slabs(ar, "t", dY, W, dX)
    torch.cuda.synchronize()
    if rows == 640:
        assert not took and bool((dX == 7.0).all())
        return
    assert took
    sel = torch.tensor([0, 1, 159, 160, 319, 320, rows // 2, rows - 1], device=dev)
    ref = dY[sel].double() @ W.double()
    plain = torch.empty(rows, In, device=dev)
    ops.grad_input(dY, W, plain)
    e_slabs = float((dX[sel].double() - ref).abs().max())
    e_plain = float((plain[sel].double() - ref).abs().max())
    assert e_slabs <= max(1.5 * e_plain, 3e-7 * float(ref.abs().max())), (e_slabs, e_plain)
    assert float((dX - plain).abs().max()) < 1e-4 * float(plain.abs().max())


@pytest.mark.parametrize("rows,C,ld,off", [(10240, 2048, 4096, 2048), (4100, 260, 264, 0), (10240, 2048, 4097, 0), (5000, 200, 200, 0)])
def test_colsum_of_long_wide_tensors(dev, rows, C, ld, off):
    """ops.colsum on > 1024 rows: the 128 x 64 16-byte-load partial kernel (r06; aligned tensors of >= 4096 rows and >= 256 columns: the
    first two shapes -- the second with ragged last row block and column tile) and the 16-row-block kernel (unaligned row stride; narrow)
    against float64, with the copy in out2 and accumulation."""
    from mmego_amd import ops
    g = torch.Generator().manual_seed(rows + C)
    buf = torch.randn(rows, ld, generator=g).to(dev)
    X = buf[:, off:off + C]
    out, out2 = torch.full((C,), 3.0, device=dev), torch.zeros(C, device=dev)
    ops.colsum(X, out, out2=out2)
    ref = X.double().sum(0)
    tol = 2e-6 * float(X.abs().double().sum(0).max())
    assert float((out.double() - ref).abs().max()) < tol and torch.equal(out, out2)
    ops.colsum(X, out, accumulate=True)
    assert float((out.double() - 2 * ref).abs().max()) < 2 * tol


@pytest.mark.parametrize("F", [1, 37, 1024])
def test_pooled_cross_attention_is_the_two_launch_form_bit_for_bit(dev, F):
    """mmego_cross_attn_forward_pooled (r06, eval forwards of Lower_Net: the attention output summed over the frame's 64 points inside
    the attention launch, O and P never stored) against mmego_cross_attn_forward + mmego_group_sum2 over its output: the same products
    and the same order of addition -- every bit equal; columns of the output row outside the 64 sums untouched."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(100 + F)
    Q = torch.randn(F * 64, 64, generator=g).to(dev)
    KV = torch.randn(F * 15, 128, generator=g).to(dev)
    both = torch.zeros(F * 64, 128, device=dev)
    P = torch.zeros(F * 64, 15, device=dev)
    hip.call("cross_attn_forward", Q, KV, KV[:, 64:], F, 0.125, both[:, 64:], 128, P, 128)
    ak = torch.full((F, 192), 7.0, device=dev)
    kv = torch.zeros(F * 15, 64, device=dev)
    hip.call("group_sum2", F, both, 64, 128, 1.0, ak, 192, kv, 15, 64, 1.0, ak[:, 128:], 192)
    ak2 = torch.full((F, 192), 7.0, device=dev)
    hip.call("cross_attn_forward_pooled", Q, KV, KV[:, 64:], F, 0.125, ak2[:, 64:], 192, 128)
    torch.cuda.synchronize()
    assert torch.equal(ak2[:, 64:128].view(torch.int32), ak[:, 64:128].view(torch.int32))
    assert bool((ak2[:, :64] == 7.0).all()) and bool((ak2[:, 128:] == 7.0).all())
    # ... and with the query projection inside (mmego_cross_attn_forward_pooled_q): Q = X Wq^T + bq never stored; against float64 and
    # against the stored-Q form (other summation orders in two products: 2e-6 of the largest sum)
    X = torch.randn(F * 64, 64, generator=g).to(dev)
    Wq, bq = (torch.randn(64, 64, generator=g) * 0.2).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
    ak3 = torch.full((F, 192), 7.0, device=dev)
    hip.call("cross_attn_forward_pooled_q", X, 64, Wq, bq, KV, KV[:, 64:], F, 0.125, ak3[:, 64:], 192, 128)
    Qd = X.double().cpu() @ Wq.double().cpu().t() + bq.double().cpu()
    Kd, Vd = KV[:, :64].double().cpu().view(F, 15, 64), KV[:, 64:].double().cpu().view(F, 15, 64)
    want = (torch.softmax(Qd.view(F, 64, 64) @ Kd.transpose(1, 2) * 0.125, dim=-1) @ Vd).sum(dim=1)
    torch.cuda.synchronize()
    assert float((ak3[:, 64:128].double().cpu() - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max())) * 8
    assert bool((ak3[:, :64] == 7.0).all()) and bool((ak3[:, 128:] == 7.0).all())


def test_train_upper(dev):
    from mmego_amd import nets
    g = golden("g6_train.npz")
    x0, body, R, t, target = [T(g[k]) for k in ("x", "body", "R", "t", "target")]
    h0, c0 = ot.zeros_state(4)
    o, h = _train_pair("upper", 601, on.UpperNet, nets.UpperNet, dev)
    d = lambda v: v.to(dev)
    _compare_training("upper", o, h, lambda m: m(x0.clone(), h0, c0, body, R, t)[0],
                      lambda m: m(d(x0.clone()), d(h0), d(c0), d(body), d(R), d(t))[0], target[:, :, list(sk.UPPER_MAP)], g, dev)


def test_train_steps_from_synced_states(dev):
    """Steps 2 and 3 of Upper_Net and Lower_Net training at the STEP-1 tolerances (outputs 2e-5, gradients 2e-4 of the largest
    gradient), each step starting from the oracle's parameters and BatchNorm statistics."""
    from mmego_amd import nets
    g = golden("g6_train.npz")
    x0, body, R, t, target = [T(g[k]) for k in ("x", "body", "R", "t", "target")]
    up_l, x_l = T(g["lower.upper_in"]), T(g["lower.x_in"])
    h0, c0 = ot.zeros_state(4)
    d = lambda v: v.to(dev)
    o, h = _train_pair("upper", 601, on.UpperNet, nets.UpperNet, dev)
    _compare_training("upper", o, h, lambda m: m(x0.clone(), h0, c0, body, R, t)[0],
                      lambda m: m(d(x0.clone()), d(h0), d(c0), d(body), d(R), d(t))[0], target[:, :, list(sk.UPPER_MAP)], g, dev,
                      resync=True)
    o, h = _train_pair("lower", 604, lambda: on.LowerNet(64), lambda: nets.LowerNet(64), dev)
    _compare_training("lower", o, h, lambda m: m(up_l.clone(), x_l.clone(), h0, c0, h0, c0, body, R, t)[0],
                      lambda m: m(d(up_l.clone()), d(x_l.clone()), None, None, None, None, d(body), d(R), d(t))[0],
                      target[:, :, list(sk.LOWER_MAP)], g, dev, resync=True)


def test_train_lower(dev):
    from mmego_amd import nets
    g = golden("g6_train.npz")
    body, R, t, target = [T(g[k]) for k in ("body", "R", "t", "target")]
    up_l, x_l = T(g["lower.upper_in"]), T(g["lower.x_in"])
    h0, c0 = ot.zeros_state(4)
    o, h = _train_pair("lower", 604, lambda: on.LowerNet(64), lambda: nets.LowerNet(64), dev)
    d = lambda v: v.to(dev)
    _compare_training("lower", o, h, lambda m: m(up_l.clone(), x_l.clone(), h0, c0, h0, c0, body, R, t)[0],
                      lambda m: m(d(up_l.clone()), d(x_l.clone()), None, None, None, None, d(body), d(R), d(t))[0],
                      target[:, :, list(sk.LOWER_MAP)], g, dev)


def test_gcn_kernels_against_torch(dev):
    """mmego_graph_mix (einsum 'nkctv,kvw->nctw' with A * edge_importance, and its input gradient) and mmego_tconv (9x1 temporal
    convolution as an implicit GEMM: plain / with BatchNorm+ReLU applied on load / input gradient on reversed taps / weight
    gradient; ragged row and channel counts) against torch in fp64."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(23)
    Fn, V, K, C = 37, 15, 2, 24
    z = torch.randn(Fn, V, K * C, generator=g)
    A, imp = torch.rand(K, V, V, generator=g), torch.randn(K, V, V, generator=g)
    Ae = (A * imp).double()
    want = torch.einsum("fvkc,kvw->fwc", z.double().view(Fn, V, K, C), Ae)
    y = torch.empty(Fn, V, C, device=dev)
    stats = torch.empty(C, Fn, 3, device=dev)
    hip.call("graph_mix", z.to(dev), A.to(dev), imp.to(dev), y, Fn, V, K, C, 0, stats, K * C)
    assert (y.cpu().double() - want).abs().max().item() < 1e-4
    # the BatchNorm partial records it leaves beside the output: (V, mean, M2) per channel and frame
    yc = y.double().cpu()
    assert (stats[:, :, 0] == V).all()
    assert (stats[:, :, 1].double().cpu() - yc.mean(1).t()).abs().max().item() < 1e-5
    assert (stats[:, :, 2].double().cpu() - ((yc - yc.mean(1, keepdim=True)) ** 2).sum(1).t()).abs().max().item() < 1e-3
    dy = torch.randn(Fn, V, C, generator=g)
    want_dz = torch.einsum("fwc,kvw->fvkc", dy.double(), Ae).reshape(Fn, V, K * C)
    dz = torch.empty(Fn, V, K * C, device=dev)
    hip.call("graph_mix", dy.to(dev), A.to(dev), imp.to(dev), dz, Fn, V, K, C, 1, None, C)
    assert (dz.cpu().double() - want_dz).abs().max().item() < 1e-4
    # both gradients of the einsum from mmego_graph_dA: dA (per-workgroup partials, summed here) and the same dz, bit for bit
    nblk = hip.graph_dA_nblk(Fn)
    part, dz2 = torch.empty(nblk, K * V * V, device=dev), torch.empty(Fn, V, K * C, device=dev)
    hip.call("graph_dA", z.to(dev), dy.to(dev), Fn, V, K, C, part, A.to(dev), imp.to(dev), dz2, K * C, K * C)
    assert torch.equal(dz2, dz)
    want_dA = torch.einsum("fvkc,fwc->kvw", z.double().view(Fn, V, K, C), dy.double())
    assert (part.double().sum(0).cpu().view(K, V, V) - want_dA).abs().max().item() < 1e-3
    part2 = torch.empty_like(part)
    hip.call("graph_dA", z.to(dev), dy.to(dev), Fn, V, K, C, part2, None, None, None, K * C, 0)
    assert torch.equal(part2, part)
    for Bq, Tq, Ci, Co in ((3, 8, 32, 32), (2, 5, 20, 70), (5, 7, 36, 72), (70, 16, 128, 128)):
        taps = 9
        x = torch.randn(Bq, Tq, V, Ci, generator=g)
        W = torch.randn(Co, Ci, taps, 1, generator=g) / (Ci * taps) ** 0.5
        b = torch.randn(Co, generator=g)
        conv = lambda inp, w, bias: torch.nn.functional.conv2d(inp.permute(0, 3, 1, 2), w, bias, padding=(taps // 2, 0)).permute(0, 2, 3, 1)
        want = conv(x.double(), W.double(), b.double())
        rows = Bq * Tq * V
        xd, Wd, bd = x.to(dev).view(rows, Ci), W.to(dev), b.to(dev)
        wp = torch.empty(2, W.numel(), device=dev)
        hip.call("tconv_pack", Wd, Co, Ci, taps, 2, wp)
        out2 = torch.empty(rows, Co, device=dev)
        hip.call("tconv", xd, Ci, None, wp[0], bd, out2, Co, None, Bq, Tq, V, Ci, Co, taps)
        assert (out2.cpu().double().view_as(want) - want).abs().max().item() < 2e-4, (Bq, Tq, Ci, Co)
        # BatchNorm + ReLU applied on load: state = mean, invstd, a, b; the activated rows come out beside the product
        st = torch.stack((torch.randn(Ci, generator=g) * 0.2, torch.ones(Ci), torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.1))
        act = torch.relu((x.double() - st[0].double()) * st[2].double() + st[3].double())
        want3 = conv(act, W.double(), b.double())
        out3, act_out = torch.empty(rows, Co, device=dev), torch.full((rows, Ci), float("nan"), device=dev)
        hip.call("tconv", xd, Ci, st.to(dev).contiguous(), wp[0], bd, out3, Co, act_out, Bq, Tq, V, Ci, Co, taps)
        assert (out3.cpu().double().view_as(want3) - want3).abs().max().item() < 2e-4
        assert (act_out.cpu().double().view_as(act) - act).abs().max().item() < 1e-5
        # input gradient: autograd of the convolution against the same kernel on the gradient pack
        xg = x.double().clone().requires_grad_(True)
        Wg = W.double().clone().requires_grad_(True)
        dyc = torch.randn(Bq, Tq, V, Co, generator=g)
        (conv(xg, Wg, None) * dyc.double()).sum().backward()
        dx = torch.empty(rows, Ci, device=dev)
        dyd = dyc.to(dev).view(rows, Co)
        if Co % 4 == 0:
            hip.call("tconv", dyd, Co, None, wp[1], None, dx, Ci, None, Bq, Tq, V, Co, Ci, taps)
            assert (dx.cpu().double().view_as(xg.grad) - xg.grad).abs().max().item() < 2e-4
            # weight gradient, implicit too: dW[co][ci][tap] = sum_r dy[r][co] x[r + (tap-4) V][ci]; plain and accumulating
            nsp = hip.lib().mmego_tconv_wgrad_nsplit(Bq, Tq, V, Ci, Co, taps)
            ws = torch.empty(nsp * W.numel(), device=dev)
            dW = torch.full((Co, Ci, taps, 1), float("nan"), device=dev)
            hip.call("tconv_wgrad", dyd, Co, xd, Ci, ws, dW, 0, Bq, Tq, V, Ci, Co, taps)
            scale = Wg.grad.abs().max().item()
            assert (dW.cpu().double() - Wg.grad).abs().max().item() < 1e-5 * scale + 1e-4, (Bq, Tq, Ci, Co)
            hip.call("tconv_wgrad", dyd, Co, xd, Ci, ws, dW, 1, Bq, Tq, V, Ci, Co, taps)
            assert (dW.cpu().double() - 2 * Wg.grad).abs().max().item() < 2e-5 * scale + 2e-4


def test_gemm_group_equals_separate_launches(dev):
    """hip.gemm_group (mmego_gemm_group: several independent small products in one launch) gives the bits of the separate
    mmego_gemm launches: weight-gradient shaped products (dY^T X, with the bias gradient as row sums of A) of different sizes,
    batched and unbatched; a group the small-tile kernel cannot take falls back to separate launches."""
    from mmego_amd import hip, ops
    g = torch.Generator().manual_seed(41)
    rows = 512
    shapes = [(256, 128), (256, 64), (96, 200), (33, 17)]
    dYs = [torch.randn(rows, m, generator=g).to(dev) for m, _ in shapes]
    Xs = [torch.randn(rows, n, generator=g).to(dev) for _, n in shapes]

    def run(grouped):
        outs = [torch.full((m, n), float("nan"), device=dev) for m, n in shapes]
        dbs = [torch.full((m,), float("nan"), device=dev) for m, _ in shapes]

        def all_():
            for dY, X, W, db in zip(dYs, Xs, outs, dbs):
                ops.grad_weight(dY, X, W, db=db, prefer_fused=True)
        if grouped:
            with hip.gemm_group():
                all_()
        else:
            all_()
        torch.cuda.synchronize()
        return outs, dbs
    o1, b1 = run(False)
    o2, b2 = run(True)
    for a, b, dY, X in zip(o1, o2, dYs, Xs):
        assert torch.equal(a, b)
        assert (a.double() - dY.double().t() @ X.double()).abs().max().item() < 1e-3
    for a, b, dY in zip(b1, b2, dYs):
        assert torch.equal(a, b) and (a.double() - dY.double().sum(0)).abs().max().item() < 1e-3
    # a group with a product the small-tile kernel does not take (k-contiguous operands: the tile kernels' shape) still works
    A, Bm = torch.randn(128, 64, generator=g).to(dev), torch.randn(64, 64, generator=g).to(dev)
    C1, C2 = torch.empty(128, 64, device=dev), torch.empty(128, 64, device=dev)
    ops.mm(A, Bm, C1)
    with hip.gemm_group():
        ops.mm(A, Bm, C2)
        ops.grad_weight(dYs[0], Xs[0], o2[0])
    assert torch.equal(C1, C2)


def test_imu_fc2_head_kernel(dev):
    """mmego_imu_fc2_head (IMU_Net.fc2 as row-wise dot products + the 6-D head in one launch) against torch in fp64 and against
    mmego_imu_head on its own y (same bits); odd row count, row-strided input."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(31)
    for F, K, ld in ((37, 512, 512), (512, 1024, 1028)):
        xw = torch.randn(F, ld, generator=g).to(dev)
        x = xw[:, :K]
        W, b = (torch.randn(9, K, generator=g) / K ** 0.5).to(dev), torch.randn(9, generator=g).to(dev)
        y, R, t = torch.empty(F, 9, device=dev), torch.empty(F, 3, 3, device=dev), torch.empty(F, 3, device=dev)
        hip.call("imu_fc2_head", x, ld, W, b, F, K, y, R, t)
        want = x.double() @ W.double().t() + b.double()
        assert (y.double() - want).abs().max().item() < 1e-5
        R2, t2 = torch.empty_like(R), torch.empty_like(t)
        hip.call("imu_head", y, F, R2, t2)
        assert torch.equal(R, R2) and torch.equal(t, t2)
        R3, t3 = torch.empty_like(R), torch.empty_like(t)
        hip.call("imu_fc2_head", x, ld, W, b, F, K, None, R3, t3)
        assert torch.equal(R, R3) and torch.equal(t, t3)


def test_fused_adam_matches_torch(dev):
    from mmego_amd import hip
    torch.manual_seed(5)
    n = 4096 + 8
    for wd in (0.0, 1e-3):
        p0 = torch.randn(n)
        ref_p = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([ref_p], lr=3e-5, weight_decay=wd)
        p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        state = torch.zeros(3, dtype=torch.float64, device=dev)
        ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        for step in range(1, 6):
            grad = torch.randn(n) * (10.0 ** (step - 3))
            ref_p.grad = grad.clone()
            opt.step()
            hip.call("adam_step", p, grad.to(dev), m, v, n, state, 3e-5, 0.9, 0.999, 1e-8, wd, None, 0, ticket)
            assert torch.allclose(p.cpu(), ref_p.detach(), rtol=0, atol=2e-7), (wd, step)
        assert state[0].item() == 5.0 and ticket.item() == 0


def test_gcn_fused_training_step_equals_the_launch_chain(dev):
    """The fused ST-GCN training kernels (gcn_fused.hip: BatchNorm statistics as partial records finalized in the consumers'
    prologues, einsum on MFMAs, deferred slab sums) against the per-operation launch chain they replace (nets._GCN_FUSED = False), on the
    same Lower_Net and minibatch: outputs, every gradient, BatchNorm running statistics -- and both against the CPU oracle through
    the tests above.  Two shapes: ragged (frames not a multiple of the 4-frame tile) and the bench shape's row count."""
    from mmego_amd import nets
    for Bq, Tq in ((3, 5), (16, 8)):
        g = torch.Generator().manual_seed(41 + Bq)
        x = torch.randn(Bq, Tq, 128, 6, generator=g)
        x[:, :, 90:] = 0.0
        upper_l = torch.randn(Bq, Tq, 15, 3, generator=g) * 0.3
        body = 0.2 * torch.randn(Bq, 20, 3, generator=g)
        R = torch.linalg.qr(torch.randn(Bq, Tq, 3, 3, generator=g))[0].contiguous()
        t = torch.randn(Bq, Tq, 3, generator=g) * 0.1
        target = torch.randn(Bq, Tq, 8, 3, generator=g)
        res = []
        for fused in (False, True):
            torch.manual_seed(77)
            net = nets.LowerNet(64).to(dev).train()
            net.lstm_dropout = 0.0
            was = nets._GCN_FUSED
            nets._GCN_FUSED = fused
            try:
                l = net(upper_l.to(dev), x.clone().to(dev), None, None, None, None, body.to(dev), R.to(dev), t.to(dev))[0]
                (l - target.to(dev)).abs().sum().backward()
                torch.cuda.synchronize()
            finally:
                nets._GCN_FUSED = was
            assert net._gcn_was_fused == fused
            res.append((l.detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()},
                        {k: v.detach().cpu().clone() for k, v in net.named_buffers()}))
        (l0, g0, b0), (l1, g1, b1) = res
        assert (l0 - l1).abs().max().item() < 2e-5, (Bq, Tq)
        scale = max(v.abs().max().item() for v in g0.values())
        for k in g0:
            assert (g0[k] - g1[k]).abs().max().item() < 2e-5 * scale, (k, (g0[k] - g1[k]).abs().max().item(), scale)
        for k in b0:
            assert (b0[k].double() - b1[k].double()).abs().max().item() < 1e-5 * max(1.0, b0[k].double().abs().max().item()), k


def test_fused_pooling_in_the_pointnet_chain_equals_the_separate_launches(dev):
    """pool128_bn_act / pool128_backward (GlobalPointNet's last BatchNorm + ReLU inside the softmax pooling, and the pooling's backward
    with the stage's BatchNorm sums and the attention partials) against mlp_bn_act + attn_pool_forward / attn_pool_backward + column
    sums + mlp_bn_bwd_reduce (blocks._POOL_FUSED = False) on the same Upper_Net: outputs, attention weights, every gradient, running statistics."""
    from mmego_amd import blocks, nets
    g = torch.Generator().manual_seed(13)
    Bq, Tq = 4, 6
    x = torch.randn(Bq, Tq, 128, 6, generator=g) * 0.5
    x[:, :, 80:] = 0.0
    body = 0.2 * torch.randn(Bq, 20, 3, generator=g)
    R = torch.linalg.qr(torch.randn(Bq, Tq, 3, 3, generator=g))[0].contiguous()
    t = torch.randn(Bq, Tq, 3, generator=g) * 0.1
    target = torch.randn(Bq, Tq, 15, 3, generator=g)
    h0 = torch.zeros(6, Bq, 64)
    res = []
    for fused in (True, False):
        torch.manual_seed(5)
        net = nets.UpperNet().to(dev).train()
        net.lstm_dropout = 0.0
        was = blocks._POOL_FUSED
        blocks._POOL_FUSED = fused
        try:
            out = net(x.clone().to(dev), h0.to(dev), h0.to(dev), body.to(dev), R.to(dev), t.to(dev))
            (out[0] - target.to(dev)).abs().sum().backward()
            torch.cuda.synchronize()
        finally:
            blocks._POOL_FUSED = was
        assert net._gpool_fused == fused
        res.append((out[0].detach().cpu(), out[2].detach().cpu(), {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()},
                    {k: v.detach().cpu().clone() for k, v in net.named_buffers()}))
    (l1, a1, g1, b1), (l0, a0, g0, b0) = res
    assert (l1 - l0).abs().max().item() < 2e-5 and (a1 - a0).abs().max().item() < 1e-6
    scale = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        assert (g1[k] - g0[k]).abs().max().item() < 2e-5 * scale, (k, (g1[k] - g0[k]).abs().max().item(), scale)
    for k in b0:
        assert (b1[k].double() - b0[k].double()).abs().max().item() < 1e-5 * max(1.0, b0[k].double().abs().max().item()), k


@pytest.mark.parametrize("Bn,R,C", [(5, 240, 64), (7, 27, 64), (3, 64, 27), (1, 2048, 512), (2, 130, 70), (9000, 3, 5)])
def test_transpose_batched_bit_exact(dev, Bn, R, C):
    """mmego_transpose_batched (64 x 64 tiles through LDS): out[b][c][r] = in[b][r][c], every shape the nets use and ragged ones."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(Bn + R + C)
    x = torch.randn(Bn, R, C, generator=g).to(dev)
    out = torch.full((Bn * R * C + 8,), 7.0, device=dev)
    hip.call("transpose_batched", x, out, Bn, R, C)
    torch.cuda.synchronize()
    assert torch.equal(out[:Bn * R * C].view(Bn, C, R), x.transpose(1, 2).contiguous())
    assert torch.all(out[Bn * R * C:] == 7.0)


@pytest.mark.parametrize("Bn,T", [(64, 8), (37, 5), (16, 2), (5, 16), (64, 1), (65, 3)])
def test_rnn_slow_persistent_recurrence_against_the_step_launches(dev, Bn, T, monkeypatch):
    """mmego_lstm_seq_xcd (lstm_seq.hip: a BiLSTM(512) layer's whole recurrence for <= 64 rows as one persistent launch, weights
    stationary) against the launch-per-timestep form on the same projections: same expressions, the products summed in another
    order -> 2e-6 absolute on |h| < 1 after up to 20 steps and two layers; no bounded spin ran out; the synchronisation words are
    back at zero; a second run gives the same bits (fixed summation order); T = 1 and more than 64 rows take the step path."""
    from mmego_amd import blocks, ops
    torch.manual_seed(Bn * 100 + T)
    H, In = 512, 1024
    lstm = blocks.LstmParams(In, H, 2, bidirectional=True).to(dev)
    x = torch.randn(Bn * T, In, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = {}
    for on in (False, True, True):
        monkeypatch.setattr(blocks, "_LSTM_SEQ_XCD", on)
        ar = ops.Arena(dev)
        o = blocks.lstm_steps_forward(ar, "slow", lstm, x, Bn, T).clone()
        torch.cuda.synchronize()
        if on:
            assert ar.has("slow.seqsync") == (T > 1 and Bn <= 64)
            assert blocks.seq_xcd_errors(ar) == 0
            if T > 1 and Bn <= 64:
                words = ar.get("slow.seqsync", (16,), dtype=torch.int32).cpu()
                assert int(words[8]) == 0 and int(words[9]) == 0 and int(words[10]) == 2        # two layers = two launches
            if True in outs:
                assert torch.equal(outs[True], o)
        outs[on] = o
    err = float((outs[True] - outs[False]).abs().max())
    assert torch.isfinite(outs[True]).all()
    assert err < 2e-6, err
    if T == 1 or Bn > 64:
        assert torch.equal(outs[True], outs[False])


def test_two_persistent_recurrences_side_by_side(dev):
    """Two mmego_lstm_seq_xcd launch chains on two streams at once (what PipelinedStages does with the two frozen IMU_Net forwards):
    each launch's 256 workgroups wait for each other, so both grids must be resident together (two workgroups per CU).  Results equal
    the one-at-a-time runs bit for bit, no bounded spin runs out."""
    from mmego_amd import blocks, ops
    H, In, Bn, T = 512, 1024, 64, 8
    nets_ = []
    for seed in (1, 2):
        torch.manual_seed(seed)
        lstm = blocks.LstmParams(In, H, 2).to(dev)
        x = torch.randn(Bn * T, In, generator=torch.Generator().manual_seed(10 + seed)).to(dev)
        nets_.append((lstm, x, ops.Arena(dev)))
    alone = [blocks.lstm_steps_forward(ar, "slow", lstm, x, Bn, T).clone() for lstm, x, ar in nets_]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for it in range(20):
        for i, (lstm, x, ar) in enumerate(nets_):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                outs[i] = blocks.lstm_steps_forward(ar, "slow", lstm, x, Bn, T)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    assert blocks.seq_xcd_errors() == 0
    for a, o in zip(alone, outs):
        assert torch.equal(a, o)


def test_mlp_dw_reduce_multi_against_torch(dev):
    """mmego_mlp_dw_reduce_multi: several layers' per-workgroup partial records summed by one launch (fp64, fixed order) -- dW tiles
    (records 4096 floats apart, element (m, n) at m * 64 + n) of two row counts, and the pooling kernels' attention-parameter partials
    (records 128 floats apart: weight [64], bias at column 64) -- against float64 sums; neighbouring memory untouched."""
    from mmego_amd import hip
    g = torch.Generator().manual_seed(9)
    descs, want, outs, keep = [], [], [], []
    for rows, Cout, Cin in ((65536, 64, 28), (65536, 8, 6), (3000, 32, 31)):
        nblk = int(hip.lib().mmego_mlp_train_nblk(rows))
        part = torch.randn(nblk, 64, 64, generator=g).to(dev)
        out = torch.full((Cout * Cin + 4,), 7.0, device=dev)
        descs.append(hip.DwRed(hip.ptr(part), hip.ptr(out), Cout, Cin, rows, 0, 0))
        want.append(part[:, :Cout, :Cin].double().sum(0).reshape(-1))
        outs.append(out); keep.append(part)
    awp = torch.randn(200, 128, generator=g).to(dev)
    ow, ob = torch.full((68,), 7.0, device=dev), torch.full((5,), 7.0, device=dev)
    descs.append(hip.DwRed(hip.ptr(awp), hip.ptr(ow), 1, 64, 0, 200, 128))
    descs.append(hip.DwRed(hip.ptr(awp[:, 64:]), hip.ptr(ob), 1, 1, 0, 200, 128))
    want += [awp[:, :64].double().sum(0), awp[:, 64:65].double().sum(0)]
    outs += [ow, ob]
    hip.call("mlp_dw_reduce_multi", len(descs), (hip.DwRed * len(descs))(*descs))
    torch.cuda.synchronize()
    for o, w in zip(outs, want):
        n = w.numel()
        assert torch.equal(o[:n].cpu(), w.float().cpu())          # fp64 sums rounded once: the fp32 nearest to the exact sum
        assert torch.all(o[n:] == 7.0)
