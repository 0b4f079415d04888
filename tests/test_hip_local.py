"""GPU: anchor grouping ("voxel indices"), UpperNetwlocal, the device-side evaluation metric and the
`--infer` pipeline against the oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from conftest import check_pinned, golden, load_weights, set_lstm_dropout
from oracle import geometry as geo
from oracle import metric as om
from oracle import nets as on
from oracle import skeleton as sk
from oracle import train as ot
from test_oracle_golden import assert_indices_equal_modulo_ties
from test_hip_parity import _compare_training, _train_pair, T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from mmego_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def test_anchor_grouping_bit_exact(dev):
    from mmego_amd import hip
    from mmego_amd.nets_local import anchor_grid
    g = golden("g2_grouping.npz")
    xyz, feats = T(g["xyz"]), T(g["feats"])
    Fn, N, D = xyz.shape[0], xyz.shape[1], feats.shape[2]
    assert torch.equal(anchor_grid(), T(g["anchors"]))
    xf = torch.cat((xyz, feats), dim=-1).contiguous().to(dev)
    idx = torch.empty((Fn, 27, 8), dtype=torch.int64, device=dev)
    grouped = torch.empty((Fn * 27 * 8, 6 + D), device=dev)
    dist = torch.empty((Fn, 27, N), device=dev)
    hip.call("anchor_group", xf, 3 + D, Fn, N, D, anchor_grid().to(dev), idx, grouped, dist)
    ref_d = T(g["dist"])
    assert torch.equal(dist.cpu(), ref_d), "distance matrix must be bit-identical to the reference's"
    # vs the reference's own indices: identical wherever its (unstable) sort had no tie to break
    assert_indices_equal_modulo_ties(idx.cpu(), T(g["idx"]), ref_d)
    # vs the oracle (stable, lowest index first): exact, and so is the gathered tensor
    o_grouped, o_idx = geo.anchor_grouping(xyz, feats, 8)
    assert torch.equal(idx.cpu(), o_idx), "group indices bit-exact"
    assert torch.equal(grouped.cpu().view(Fn, 27, 8, 6 + D), o_grouped)
    same = (o_idx == T(g["idx"])).all(dim=-1)
    assert torch.equal(grouped.cpu().view(Fn, 27, 8, 6 + D)[same], T(g["grouped"])[same])
    # backward: scatter-add equals autograd of the gather
    dg = torch.randn(Fn * 27 * 8, 6 + D)
    xf_req = torch.cat((xyz, feats), dim=-1).clone().requires_grad_(True)
    og, _ = geo.anchor_grouping(xf_req[..., :3], xf_req[..., 3:], 8)
    og.backward(dg.view_as(og))
    dxf = torch.zeros((Fn * N, 3 + D), device=dev)
    hip.call("anchor_group_backward", dg.to(dev), idx, Fn, N, D, dxf, 3 + D)
    assert torch.allclose(dxf.cpu().view(Fn, N, 3 + D), xf_req.grad, atol=1e-5)


def test_local_group_l1_bit_exact_and_against_the_launch_chain(dev):
    """local.hip: the wave-parallel grouping kernel (mmego_local_group_l1) -- distances bit-identical to the reference's, int64 group
    indices and gathered rows bit-identical to the oracle / to mmego_anchor_group (golden G2, incl. a frame with < 8 valid points and
    exact-zero rows; a seeded N = 64 / 128 / 256 sweep with duplicated points = ties); its fused first layer + BatchNorm partial sums
    against torch; mmego_anchor_scatter bit-identical to mmego_anchor_group_backward."""
    from mmego_amd import hip
    from mmego_amd.nets_local import anchor_grid
    g = golden("g2_grouping.npz")
    xyz, feats = T(g["xyz"]), T(g["feats"])
    anchors = anchor_grid().to(dev)
    cases = [(xyz, feats)]
    gen = torch.Generator().manual_seed(5)
    for N in (64, 128, 256):
        x = torch.randn(5, N, 3, generator=gen) * 0.4 + torch.tensor([0.3, 0.0, 0.0])
        x[:, N // 2:N // 2 + 9] = x[:, 0:9]                  # exact duplicates: ties
        x[1, 5:] = 0.0                                        # a frame with 5 valid points (the rest at distance +inf)
        x[2, ::3] = 0.0
        cases.append((x, torch.randn(5, N, 25, generator=gen)))
    for xyz_, feats_ in cases:
        Fn, N, D = xyz_.shape[0], xyz_.shape[1], feats_.shape[2]
        if N not in (64, 128, 256):
            continue
        xf = torch.cat((xyz_, feats_), dim=-1).contiguous().to(dev)
        idx0, idx1 = (torch.empty((Fn, 27, 8), dtype=torch.int64, device=dev) for _ in range(2))
        gr0, gr1 = (torch.empty((Fn * 216, 6 + D), device=dev) for _ in range(2))
        d0, d1 = (torch.empty((Fn, 27, N), device=dev) for _ in range(2))
        hip.call("anchor_group", xf, 3 + D, Fn, N, D, anchors, idx0, gr0, d0)
        W1, b1 = torch.randn(32, 6 + D, generator=gen).to(dev) * 0.2, torch.randn(32, generator=gen).to(dev)
        for nwg in (256, 3):                                 # one frame per workgroup / several frames per workgroup
            z1 = torch.full((Fn * 216, 32), float("nan"), device=dev)
            nw = min(nwg, Fn)
            part = torch.empty(nw * 2 * 64, dtype=torch.float64, device=dev)
            hip.call("local_group_l1", xf, 3 + D, Fn, N, D, anchors, idx1, gr1, W1, b1, 32, z1, 32, part, nwg, d1)
            assert torch.equal(d0, d1) and torch.equal(idx0, idx1) and torch.equal(gr0, gr1), (N, nwg)
            want = gr0.double() @ W1.double().t() + b1.double()
            assert (z1.double() - want).abs().max().item() < 1e-4
            ps = part.view(nw, 2, 64).sum(0)
            assert (ps[0, :32] - want.sum(0)).abs().max().item() < 1e-3 and (ps[1, :32] - (want ** 2).sum(0)).abs().max().item() < 1e-2
            assert float(ps[:, 32:].abs().max()) == 0.0
        o_grouped, o_idx = geo.anchor_grouping(xyz_, feats_, 8)
        assert torch.equal(idx1.cpu(), o_idx) and torch.equal(gr1.cpu().view(Fn, 27, 8, 6 + D), o_grouped)
        if xyz_ is xyz:
            assert torch.equal(d1.cpu(), T(g["dist"])), "distance matrix bit-identical to the reference's (golden G2)"
            assert_indices_equal_modulo_ties(idx1.cpu(), T(g["idx"]), T(g["dist"]))
        idx2 = torch.empty_like(idx1)
        hip.call("local_group_l1", xf, 3 + D, Fn, N, D, anchors, idx2, None, None, None, 0, None, 0, None, 256, None)     # grouping only
        assert torch.equal(idx2, idx1)
        dg = torch.randn(Fn * 216, 6 + D, generator=gen).to(dev)
        a0, a1 = (torch.ones((Fn * N, 3 + D), device=dev) for _ in range(2))
        hip.call("anchor_group_backward", dg, idx1, Fn, N, D, a0, 3 + D)
        hip.call("anchor_scatter", dg, idx1, Fn, N, D, a1, 3 + D)
        assert torch.equal(a0, a1), "same summation order as the slot walk"


def test_pool8_kernels_against_torch(dev):
    """mmego_pool8_bn_act (BatchNorm + ReLU + score + 8-way softmax + weighted sum, pooled vectors in [frame][channel][anchor] order)
    and mmego_pool8_backward (row gradients, BatchNorm sums through the ReLU mask, attention parameter partials) against autograd
    in fp64, fed by mmego_mlp_fwd_layer's statistics partials."""
    from mmego_amd import hip
    gen = torch.Generator().manual_seed(9)
    Fn = 7
    rows = Fn * 216
    x = torch.randn(rows, 48, generator=gen)
    W, b = torch.randn(64, 48, generator=gen) * 0.2, torch.randn(64, generator=gen) * 0.1
    gamma, beta = torch.rand(64, generator=gen) + 0.5, torch.randn(64, generator=gen) * 0.1
    aw_w, aw_b = torch.randn(1, 64, generator=gen) * 0.3, torch.randn(1, generator=gen)
    dv = torch.randn(Fn, 64, 27, generator=gen)
    nblk = hip.lib().mmego_mlp_train_nblk(rows)
    assert nblk == hip.lib().mmego_pool8_nblk(rows)
    z = torch.empty(rows, 64, device=dev)
    part = torch.empty(nblk * 2 * 64, dtype=torch.float64, device=dev)
    d = lambda v: v.to(dev).contiguous()
    hip.call("mlp_fwd_layer", d(x), 48, rows, 48, None, None, None, 0.0, None, None, 0.0, None, d(W), d(b), 64, z, 64, part)
    rm, rv, state = torch.zeros(64, device=dev), torch.ones(64, device=dev), torch.empty(4, 64, device=dev)
    voxT, attn = torch.empty(Fn, 64, 27, device=dev), torch.empty(rows, device=dev)
    hip.call("pool8_bn_act", z, 64, rows, part, d(gamma), d(beta), 1e-5, rm, rv, 0.1, state, d(aw_w), d(aw_b), voxT, attn)
    # fp64 reference
    zd = (x.double() @ W.double().t() + b.double()).requires_grad_(True)
    g64, b64, w64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), aw_w.double().requires_grad_(True)
    ab64 = aw_b.double().requires_grad_(True)
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    y = torch.relu((zd - mean) / torch.sqrt(var + 1e-5) * g64 + b64)
    y.retain_grad()
    sc = (y @ w64.t() + ab64).view(-1, 8)
    at = torch.softmax(sc, dim=1)
    vec = (at.unsqueeze(-1) * y.view(-1, 8, 64)).sum(1)                      # [F*27, 64]
    assert (attn.double().cpu().view(-1, 8) - at).abs().max().item() < 1e-5
    assert (voxT.double().cpu() - vec.view(Fn, 27, 64).transpose(1, 2)).abs().max().item() < 1e-4
    assert (rm.double().cpu() - 0.1 * mean.detach()).abs().max().item() < 1e-5
    (vec.view(Fn, 27, 64).transpose(1, 2) * dv.double()).sum().backward()
    dY = torch.empty(rows, 64, device=dev)
    gpart = torch.empty(nblk * 2 * 64, dtype=torch.float64, device=dev)
    awp = torch.empty(nblk, 128, device=dev)
    hip.call("pool8_backward", z, 64, rows, state, attn, d(dv), d(aw_w), dY, 64, gpart, awp)
    assert (dY.double().cpu() - y.grad).abs().max().item() < 1e-4 * max(1.0, y.grad.abs().max().item())
    gmask = y.grad * (y.detach() > 0)
    xhat = ((zd - mean) / torch.sqrt(var + 1e-5)).detach()
    gp = gpart.view(nblk, 2, 64).sum(0).cpu()
    assert (gp[0] - gmask.sum(0)).abs().max().item() < 1e-3 and (gp[1] - (gmask * xhat).sum(0)).abs().max().item() < 1e-3
    assert (awp[:, :64].double().sum(0).cpu() - w64.grad.view(-1)).abs().max().item() < 1e-3 * max(1.0, w64.grad.abs().max().item())
    assert abs(float(awp[:, 64].double().sum()) - float(ab64.grad)) < 1e-3
    assert float(awp[:, 65:].abs().max()) == 0.0


def test_upper_wlocal_eval_forward_fused_against_oracle_and_chain(dev):
    """Eval-mode UpperNetwlocal with the anchor branch as ONE launch (mmego_local_front_eval: grouping + folded conv/BatchNorm/ReLU x3
    + 8-way softmax pooling) against the CPU oracle (all eight outputs) and against the launch chain (nets_local._LOCAL_FUSED = False): group
    indices bit-identical, everything else at the eval-forward bar."""
    from mmego_amd import nets_local
    g = torch.Generator().manual_seed(21)
    Bq, Tq, N = 3, 5, 128
    x = torch.randn(Bq, Tq, N, 6, generator=g) * 0.4
    x[:, :, 100:] = 0.0
    x[1, 2, 4:] = 0.0                                         # a frame with 4 valid points
    body = 0.2 * torch.randn(Bq, 20, 3, generator=g)
    R = torch.linalg.qr(torch.randn(Bq, Tq, 3, 3, generator=g))[0].contiguous()
    t = torch.randn(Bq, Tq, 3, generator=g) * 0.1
    h0, c0 = ot.zeros_state(Bq)
    torch.manual_seed(31)
    o = on.UpperNetwlocal()
    for m in o.modules():                                     # non-trivial running statistics
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.running_mean.normal_(0.0, 0.2, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
    o.eval()
    hnet = nets_local.UpperNetwlocal()
    hnet.load_state_dict(o.state_dict())
    hnet = hnet.to(dev).eval()
    d = lambda v: v.to(dev)
    with torch.no_grad():
        want = o(x.clone(), h0, c0, h0, c0, body, R, t)
        outs = {}
        for fused in (True, False):
            was = nets_local._LOCAL_FUSED
            nets_local._LOCAL_FUSED = fused
            try:
                outs[fused] = [v.cpu() for v in hnet(d(x.clone()), d(h0), d(c0), d(h0), d(c0), d(body), d(R), d(t))]
                idx = hnet.last_group_idx.cpu().clone()
            finally:
                nets_local._LOCAL_FUSED = was
            outs[(fused, "idx")] = idx
    assert torch.equal(outs[(True, "idx")], outs[(False, "idx")])
    for k, (a, b, w) in enumerate(zip(outs[True], outs[False], want)):
        assert (a - b).abs().max().item() < 2e-5, ("fused vs chain, output", k)
        assert (a - w.view_as(a)).abs().max().item() < 5e-5, ("fused vs oracle, output", k)


@pytest.mark.parametrize("rows", [512, 40, 37, 17])
def test_local_voxel_net_kernels_against_fp64_and_the_generic_launches(dev, rows):
    """LocalVoxelNet's training step on its own kernels (vox.hip: 4 forward + 5 backward launches, BatchNorm statistics as partial
    records between them) against a float64 torch autograd reference of Net/Upper_Net.py:180-205 on the same inputs (output, input
    gradient, every parameter gradient, running statistics) and against the generic launch chain it replaces; whole and ragged 16-row
    tiles, down to a last tile of one row.  (Two-row batches are not held to fp64: xhat = +-1 there and the fp32 chain itself is 1e-3 off.)"""
    import torch.nn.functional as Fn
    from mmego_amd import blocks, nets_local, ops
    g = torch.Generator().manual_seed(100 + rows)
    mod = nets_local.LocalVoxelNet()
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.weight.data.uniform_(0.5, 1.5, generator=g)
            m.bias.data.normal_(0.0, 0.2, generator=g)
            m.running_mean.normal_(0.0, 0.2, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
    mod = mod.to(dev).train()
    x = (torch.randn(rows, 1728, generator=g) * 0.7).to(dev)
    dout = torch.randn(rows, 64, generator=g).to(dev)
    layers = blocks._mlp3_layers(mod)
    rstats0 = [(bn.running_mean.clone(), bn.running_var.clone()) for _, bn in layers]

    # float64 reference
    xd = x.double().requires_grad_(True)
    ps, cur, rstats_ref = [], xd, []
    for (conv, bn), (rm, rv) in zip(layers, rstats0):
        W = conv.weight.detach().double().view(conv.weight.shape[0], -1).requires_grad_(True)
        b = conv.bias.detach().double().requires_grad_(True)
        ga, be = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
        rm, rv = rm.double().clone(), rv.double().clone()
        cur = torch.relu(Fn.batch_norm(cur @ W.t() + b, rm, rv, ga, be, True, bn.momentum, bn.eps))
        ps.append((W, ga, be))
        rstats_ref.append((rm, rv))
    cur.backward(dout.double())

    def grads():
        store = {}
        def G(t):
            return store.setdefault(t.data_ptr(), torch.full_like(t, 7.0))      # (assigned, not accumulated)
        return store, G

    def restore():
        for (_, bn), (rm, rv) in zip(layers, rstats0):
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)

    res = {}
    for fused in (True, False):
        restore()
        ar = ops.Arena(dev)
        out = torch.empty(rows, 64, device=dev)
        store, G = grads()
        if fused:
            assert blocks.vox_fusable(mod, x)
            blocks.voxel_forward(ar, "vx", mod, x, out)
            dx = blocks.voxel_backward(ar, "vx", mod, x, out, dout, G)
        else:
            blocks.mlp3_forward(ar, "vx", mod, x, out, True)
            dx = blocks.mlp3_backward(ar, "vx", mod, x, out, dout, G, True)
        torch.cuda.synchronize()
        res[fused] = (out.clone(), dx.clone(), {k: v.clone() for k, v in store.items()},
                      [(bn.running_mean.clone(), bn.running_var.clone()) for _, bn in layers])

    def close(a, ref, tol, what):
        scale = max(1.0, float(ref.abs().max()))
        err = float((a.double() - ref.double()).abs().max())
        assert err < tol * scale, (what, rows, err, scale)

    out_f, dx_f, gr_f, rs_f = res[True]
    close(out_f, cur.detach(), 2e-5, "output")
    close(dx_f, xd.grad, 2e-4, "input gradient")
    for i, ((conv, bn), (W, ga, be)) in enumerate(zip(layers, ps)):
        tol = 2e-4
        close(gr_f[conv.weight.data_ptr()].view(W.shape), W.grad, tol, "dW%d" % (i + 1))
        close(gr_f[bn.weight.data_ptr()], ga.grad, tol, "dgamma%d" % (i + 1))
        close(gr_f[bn.bias.data_ptr()], be.grad, tol, "dbeta%d" % (i + 1))
        assert conv.bias.data_ptr() not in gr_f                                  # no bias gradient is written
        close(rs_f[i][0], rstats_ref[i][0], 1e-5, "running_mean%d" % (i + 1))
        close(rs_f[i][1], rstats_ref[i][1], 1e-5, "running_var%d" % (i + 1))
    out_c, dx_c, gr_c, rs_c = res[False]
    close(out_f, out_c, 2e-5, "output vs chain")
    close(dx_f, dx_c, 2e-4, "input gradient vs chain")
    for k in gr_c:
        if k in gr_f:
            close(gr_f[k], gr_c[k], 2e-4, "gradient vs chain")
    for (m1, v1), (m2, v2) in zip(rs_f, rs_c):
        close(m1, m2, 1e-5, "running_mean vs chain")
        close(v1, v2, 1e-5, "running_var vs chain")


def test_train_upper_wlocal(dev):
    from mmego_amd.nets_local import UpperNetwlocal
    g = golden("g6_train.npz")
    x0, body, R, t, target = [T(g[k]) for k in ("x", "body", "R", "t", "target")]
    h0, c0 = ot.zeros_state(4)
    o, h = _train_pair("wlocal", 602, on.UpperNetwlocal, UpperNetwlocal, dev)
    d = lambda v: v.to(dev)
    holder = {}

    def fwd_h(m):
        out = m(d(x0.clone()), d(h0), d(c0), d(h0), d(c0), d(body), d(R), d(t))
        holder.setdefault("idx", m.last_group_idx.cpu().clone())
        holder.setdefault("n", len(out))
        return out[0]

    def fwd_o(m):
        out = m(x0.clone(), h0, c0, h0, c0, body, R, t)
        holder.setdefault("oidx", m.module2.last_group_idx.clone())
        return out[0]
    # Steps 2-3 of the free-running comparison start from weights that already differ by sign-of-noise Adam updates, and the
    # LocalPointNet BatchNorm chain over 8-point groups amplifies that (2.4e-3..6e-3 of the largest gradient observed run to
    # run): hence the wide later_* bounds HERE.  That this is drift and not a step-2 defect of the anchor-group / LocalPointNet
    # backward is what test_train_upper_wlocal_from_synced_states shows: the same three steps, each started from the oracle's
    # parameters and BatchNorm statistics, hold the step-1 bar (2e-4 of the largest gradient, outputs 2e-5).
    _compare_training("wlocal", o, h, fwd_o, fwd_h, target[:, :, list(sk.UPPER_MAP)], g, dev, later_grad_tol=2e-2, later_out_atol=1e-3)
    assert holder["n"] == 8, "UpperNetwlocal returns the reference's 8-tuple"
    assert torch.equal(holder["idx"], holder["oidx"]), "voxel (group) indices bit-exact vs the oracle"
    with torch.no_grad():
        xh = geo.transform_to_head_(x0.clone(), R, t)[..., :3].contiguous()
        keys = geo.square_distance(geo.anchor_grid().unsqueeze(0).expand(32, -1, -1), xh)
    assert_indices_equal_modulo_ties(holder["idx"], T(g["wlocal.group_idx"]), keys)      # vs the real reference


def test_train_upper_wlocal_from_synced_states(dev):
    """UpperNetwlocal (U10-U12): steps 2 and 3 at the step-1 tolerances when every step starts from the oracle's parameters
    and BatchNorm buffers, plus the eval-mode forward of the trained net against the oracle."""
    from mmego_amd.nets_local import UpperNetwlocal
    g = golden("g6_train.npz")
    x0, body, R, t, target = [T(g[k]) for k in ("x", "body", "R", "t", "target")]
    h0, c0 = ot.zeros_state(4)
    o, h = _train_pair("wlocal", 602, on.UpperNetwlocal, UpperNetwlocal, dev)
    d = lambda v: v.to(dev)
    fwd_h = lambda m: m(d(x0.clone()), d(h0), d(c0), d(h0), d(c0), d(body), d(R), d(t))[0]
    fwd_o = lambda m: m(x0.clone(), h0, c0, h0, c0, body, R, t)[0]

    def near_tie(m):
        # LocalVoxelNet's stages are 32 rows x 64..128 channels here.  On this trajectory step 2 has bn3(z3) = -1.9e-6 at one element
        # (values of order 3: fp32 rounding): the generic product lands below zero, vox.hip's summation order above it, and that one
        # mask element moves the gradients by 2.4e-3 of the largest one -- while both forms are 3e-7 from a float64 run of the oracle
        # at steps 1 and 3 and the fp32 oracle itself is 1e-4 from it (tests/probe_vox_tie.py prints all of this).
        ar = m.arena("train")
        F = 4 * 8
        worst = float("inf")
        for i, C in ((1, 96), (2, 128), (3, 64)):
            z, st = ar.get("vx.z%d" % i, (F, C)), ar.get("vx.bn%d" % i, (4, C))
            worst = min(worst, float(((z - st[0]) * st[2] + st[3]).abs().min()))
        return worst < 1e-5
    _compare_training("wlocal", o, h, fwd_o, fwd_h, target[:, :, list(sk.UPPER_MAP)], g, dev, resync=True, near_tie=near_tie,
                      tie_params=r"^(module0\.|module2\.apointnet\.|module2\.avoxel\.)")      # what lies upstream of LocalVoxelNet's ReLU masks
    o.eval(); h.eval()
    with torch.no_grad():
        out_o = o(x0.clone(), h0, c0, h0, c0, body, R, t)
        out_h = h(d(x0.clone()), d(h0), d(c0), d(h0), d(c0), d(body), d(R), d(t))
    assert len(out_h) == len(out_o) == 8
    for i, (a, b) in enumerate(zip(out_h, out_o)):
        assert torch.allclose(a.cpu().reshape(b.shape), b, rtol=1e-4, atol=2e-5), ("eval-mode 8-tuple element", i)


def test_pose_metric_kernel(dev):
    from mmego_amd import processors
    from mmego_amd.config import ConfigDemo
    g = golden("g8_metric.npz")
    pred, target = T(g["pred"]), T(g["target"])
    base = processors._Base(ConfigDemo, make_dirs=False)
    a, u, l, pj, ang = base.pose_metrics(pred[:, :, list(sk.UPPER_MAP)].contiguous().to(dev),
                                         pred[:, :, list(sk.LOWER_MAP)].contiguous().to(dev), target.to(dev))
    assert abs(a - float(g["accu"])) < 1e-6 and abs(u - float(g["upper"])) < 1e-6 and abs(l - float(g["lower"])) < 1e-6
    assert np.allclose(pj, g["accu_l"], atol=1e-6) and np.allclose(ang, g["angle_l"], atol=2e-3)


def test_infer_pipeline_on_real_sequences(dev, real16):
    """main.py --infer path (Evaluator.evaluate_full) with the recorded head pose on the 16 golden sequences."""
    from mmego_amd import nets, processors
    from mmego_amd.config import ConfigDemo
    ConfigDemo.gt_head_pose = True
    base = processors._Base(ConfigDemo, make_dirs=False)
    up = load_weights(nets.UpperNet(), golden("w_upper_pretrained.npz")).to(dev).eval()
    lo = load_weights(nets.LowerNet(64), golden("w_lower_pretrained.npz")).to(dev).eval()
    from mmego_amd.data import ArraySplit
    split = ArraySplit(real16["x"], real16["target"], real16["skl"], real16["imu"], real16["R"])
    _, s = processors.evaluate_full(base, None, up, lo, split, 1, False)
    g9 = golden("g9_end2end.npz")
    assert abs(s["upper_cm"] - float(g9["upper_cm"])) < 1e-3, (s["upper_cm"], float(g9["upper_cm"]))
    assert abs(s["lower_cm"] - float(g9["lower_cm"])) < 0.05          # reference tie order at the top-64 cut differs
    # and against the oracle with the same (stable) tie rule: within the 1e-3 cm bar
    ou = load_weights(on.UpperNet(), golden("w_upper_pretrained.npz")).eval()
    ol = load_weights(on.LowerNet(64), golden("w_lower_pretrained.npz")).eval()
    rows = []
    with torch.no_grad():
        for i in range(16):
            x = T(real16["x"][i:i + 1]).clone()
            tgt = T(real16["target"][i:i + 1])
            skl, R = T(real16["skl"][i:i + 1]), T(real16["R"][i:i + 1])
            t = tgt[:, :, 20].contiguous()
            h0, c0 = ot.zeros_state(1)
            l = ou(x, h0, c0, skl, R, t)[0]
            ll = ol(l.clone(), x, h0, c0, h0, c0, skl, R, t)[0]
            rows.append(om.batch_errors(l, ll, tgt))
    so = om.summarize(rows)
    for k in ("all_cm", "upper_cm", "lower_cm"):
        assert abs(s[k] - so[k]) < 1e-3, (k, s[k], so[k])
    assert abs(s["rot_deg"] - so["rot_deg"]) < 1e-2


def test_upper_epoch_evaluation_on_device(dev, real16):
    """SURVEY 8-f row f2, the Train_Upper half (reference Processor/Train/Train_Upper.py:189-251): UpperTrainer.eval_model -- device-
    resident test split, per-frame figures from mmego_pose_errors_upper, one host read per epoch -- returns the reference's six
    values (eval_loss, eval_loss_l, eval_accu, dis_l, accu_ll[15], angle_ll[14]) for minibatches of 5, 5, 5 and 1 sequences in
    the shuffled order of the trainer's RNG; pinned against the oracle's forward + the reference's formulas written out in torch."""
    from mmego_amd import nets, processors
    from mmego_amd.config import Config
    from mmego_amd.data import ArraySplit, batch_indices
    Config.gt_head_pose = True
    base = processors._Base(Config, make_dirs=False)
    base.model = load_weights(nets.UpperNet(), golden("w_upper_pretrained.npz")).to(dev)
    base.model_IMU, base.batchsize = None, 5
    base.test_data = ArraySplit(real16["x"], real16["target"], real16["skl"], real16["imu"], real16["R"])
    outs = []
    for _ in range(2):                                     # second epoch: cached split / scratch buffers, another shuffle
        base._rng = np.random.RandomState(77)
        outs.append(processors.UpperTrainer.eval_model(base))
    ou = load_weights(on.UpperNet(), golden("w_upper_pretrained.npz")).eval()
    umap = list(sk.UPPER_MAP)
    bones = [(umap.index(p), umap.index(c)) for p, c in sk.BONES_UPPER]
    loss_l, accu_l, dis_l, pj, ang = [], [], [], [], []
    with torch.no_grad():
        for idx in batch_indices(16, 5, True, np.random.RandomState(77)):
            x, tgt = T(real16["x"][idx]).clone(), T(real16["target"][idx])
            skl, R = T(real16["skl"][idx]), T(real16["R"][idx])
            B, Tn = x.shape[0], x.shape[1]
            h0, c0 = ot.zeros_state(B)
            up = ou(x, h0, c0, skl, R, tgt[:, :, 20].contiguous())[0]
            tu = tgt[:, :, umap]
            loss_l.append((up - tu).abs().sum().item() / B / Tn)
            d = torch.sqrt(torch.sum(torch.square(up - tu), dim=-1))
            accu_l.append(d.mean().item()); pj.append(d.mean(0).mean(0).numpy()); dis_l.append((up - tu).abs().mean().item())
            pv = torch.stack([up[:, :, c] - up[:, :, p] for p, c in bones], 2)
            tv = torch.stack([tu[:, :, c] - tu[:, :, p] for p, c in bones], 2)
            cs = torch.nn.functional.cosine_similarity(pv, tv, dim=-1)
            ang.append((torch.acos(cs.clamp(-1.0, 1.0)) / 3.14159265358 * 180.0).abs().mean(0).mean(0).numpy())
    want = (float(np.mean(loss_l)), np.asarray([np.mean(loss_l) / 15]), float(np.mean(accu_l)), float(np.mean(dis_l)),
            np.mean(pj, axis=0), np.mean(ang, axis=0))
    for got in outs:
        assert len(got) == 6 and got[4].shape == (15,) and got[5].shape == (14,)
        assert abs(got[0] - want[0]) < 2e-5 * abs(want[0]) and abs(got[1][0] - want[1][0]) < 2e-5 * abs(want[1][0])
        assert abs(got[2] - want[2]) * 100 < 1e-3 and abs(got[3] - want[3]) * 100 < 1e-3                 # cm
        assert np.abs(got[4] - want[4]).max() * 100 < 1e-3, np.abs(got[4] - want[4]).max()
        assert np.abs(got[5] - want[5]).max() < 1e-2, np.abs(got[5] - want[5]).max()                     # degrees


def test_trainers_and_eval_passes_raise_on_a_failed_persistent_launch(dev, real16):
    """VERDICT r04 item 2 / ADVICE r04: mmego_lstm_seq_xcd (the frozen IMU_Net's rnn_slow recurrence as one persistent launch) reports
    "my workgroups were not co-resident, results invalid" only through the sticky word 9 of its sync buffer.  The product path must
    read it: every evaluation pass (processors._epoch_eval: UpperTrainer / LowerTrainer.eval_model, Evaluator.eval_model via
    evaluate_full) at its end and the trainers' epoch loop once per epoch (reference loop: Processor/Train/Train_Upper.py:90-132).
    The word is set BY HAND here (a real failure needs a partitioned / shared device) and reset afterwards."""
    from mmego_amd import blocks, hip, nets, processors
    from mmego_amd.config import Config
    from mmego_amd.data import ArraySplit
    if not blocks._LSTM_SEQ_XCD or hip.lib().mmego_lstm_seq_xcd_slots() < 1:
        pytest.skip("persistent rnn_slow launch not in use on this device")
    was = Config.gt_head_pose
    Config.gt_head_pose = False
    try:
        base = processors._Base(Config, make_dirs=False)
    finally:
        Config.gt_head_pose = was
    base.model = load_weights(nets.UpperNet(), golden("w_upper_pretrained.npz")).to(dev)
    torch.manual_seed(3)
    base.model_IMU = nets.IMUNet(15, 9, 512, 2, True, 0.1).to(dev).eval()
    base.batchsize = 8
    base.test_data = ArraySplit(real16["x"], real16["target"], real16["skl"], real16["imu"], real16["R"])
    base._rng = np.random.RandomState(5)
    n0 = len(blocks._seq_sync_bufs)
    out = processors.UpperTrainer.eval_model(base)                   # head pose from IMU_Net: persistent launches, word stays 0
    assert len(out) == 6 and np.isfinite(out[0])
    bufs = blocks._seq_sync_bufs[n0:]
    assert bufs, "the IMU_Net forward of the evaluation pass did not take the persistent launch"
    assert blocks.seq_xcd_errors() == 0
    try:
        bufs[0][9] = 1
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match="mmego_lstm_seq_xcd"):
            processors.UpperTrainer.eval_model(base)
        # the trainers' epoch loop: the check sits directly behind train_once
        class _Loop(processors._StageTrainer):
            def __init__(self):
                pass
        tr = _Loop()
        tr.cfg, tr.start_epoch, tr.num_epochs, tr.pg, tr.rank = Config, 0, 1, None, 0
        ran = []
        tr.train_once = lambda: ran.append(1)
        tr.eval_model = lambda: pytest.fail("the epoch went on after a failed persistent launch")
        tr.model = base.model
        rp = getattr(Config, "resume_path", None)
        Config.resume_path = None
        try:
            with pytest.raises(RuntimeError, match="mmego_lstm_seq_xcd"):
                tr._train_loop(lambda *a: None)
        finally:
            Config.resume_path = rp
        assert ran == [1]
    finally:
        bufs[0][9] = 0
        torch.cuda.synchronize()
    assert blocks.seq_xcd_errors() == 0
    processors.UpperTrainer.eval_model(base)                         # and the path works again once the word is clear


def test_imu_stage1_training(dev):
    """IMU_Net forward + backward (BiLSTM backward through time, attention pool, geodesic + position loss) vs the
    reference's gradients (golden g7, hidden 32) and vs the oracle at hidden 64 with the benchmark's T/S."""
    from mmego_amd import hip, nets
    g = golden("g7_imu.npz")
    imu = T(g["imu"])
    h = load_weights(nets.IMUNet(15, 9, 32, 2, True, 0), g, "train.w.").to(dev).train()
    R, t = h(imu.to(dev))
    Rg, hg = T(g["train.R_gt"]).to(dev), T(g["train.head_gt"]).to(dev)
    loss = torch.zeros(1, device=dev)
    dR, dt = torch.empty_like(R), torch.empty_like(t)
    hip.call("imu_loss", R.detach().contiguous(), t.detach().contiguous(), Rg.contiguous(), hg.contiguous(), 6, 1.0, loss, dR, dt)
    assert abs(loss.item() - float(g["train.loss"])) < 1e-4 * abs(float(g["train.loss"]))
    torch.autograd.backward((R, t), (dR, dt))
    grads = [(k, p.grad) for k, p in h.named_parameters()]
    scale = max(x.abs().max().item() for _, x in grads)
    check_pinned(g, "train.grad.", grads, rtol=2e-3, atol=2e-4 * scale)
    # bigger instance against the oracle, through torch's autograd of an arbitrary scalar of (R, t)
    torch.manual_seed(21)
    o = on.IMUNet(15, 9, 64, 2, True, 0).train()
    torch.manual_seed(21)
    hb = nets.IMUNet(15, 9, 64, 2, True, 0).to(dev).train()
    imu2 = torch.randn(3, 4, 20, 15)
    wR, wt = torch.randn(3, 4, 3, 3), torch.randn(3, 4, 3)
    Ro, to_ = o(imu2)
    ((Ro * wR).sum() + (to_ * wt).sum()).backward()
    Rh, th = hb(imu2.to(dev))
    ((Rh * wR.to(dev)).sum() + (th * wt.to(dev)).sum()).backward()
    assert torch.allclose(Rh.detach().cpu(), Ro.detach(), atol=1e-5) and torch.allclose(th.detach().cpu(), to_.detach(), atol=1e-5)
    po = dict(o.named_parameters())
    scale = max(p.grad.abs().max().item() for p in po.values() if p.grad is not None)
    for k, ph in hb.named_parameters():
        go = po[k].grad if po[k].grad is not None else torch.zeros_like(po[k])
        assert (ph.grad.cpu() - go).abs().max().item() < 2e-4 * scale, k


@pytest.mark.parametrize("train_precision", ["fp32", "split3"])
def test_imu_stage1_gradients_at_full_size(dev, monkeypatch, train_precision):
    """Stage-1 backward at the REAL size: IMUNet(15, 9, 512, 2) with 128 and 512 rnn_fast rows -- the
    dispatch the 8.9 ms/step figure runs on (persistent 128x128 tile products in NN / TN orientation with split-K, the batched
    K-quartered dh launch with the cell backward on its tiles, lstm_step_dma_kernel with gate and cell stashes) -- against the oracle's autograd: every gradient
    element, 2e-4 of the largest gradient.  Then one Adam step with Train_IMU's weight decay: fc3 (never used in forward, Q7)
    must stay untouched, as under torch.optim.Adam (its .grad is None there), everything else must match torch's update."""
    from mmego_amd import nets
    from mmego_amd.params import FusedAdam
    torch.manual_seed(31)
    o = on.IMUNet(15, 9, 512, 2, True, 0).train()
    hb = nets.IMUNet(15, 9, 512, 2, True, 0)
    hb.load_state_dict(o.state_dict())
    hb = hb.to(dev).train()
    # "split3" (r05, opt-in): rnn_fast's input-projection and input-gradient products as fp32-accurate piece products on the bf16
    # matrix pipe (imu_train._s3_worth: the 10 240-row products of the B = 64 case; at B = 16 every product stays on the fp32 kernels)
    hb.train_precision = train_precision
    for Bq, Tq in ((16, 8), (64, 8)):
        gen = torch.Generator().manual_seed(100 + Bq)
        imu = torch.randn(Bq, Tq, 20, 15, generator=gen)
        wR, wt = torch.randn(Bq, Tq, 3, 3, generator=gen), torch.randn(Bq, Tq, 3, generator=gen)
        for p_ in o.parameters():
            p_.grad = None
        Ro, to_ = o(imu)
        ((Ro * wR).sum() + (to_ * wt).sum()).backward()
        Rh, th = hb(imu.to(dev))
        ((Rh * wR.to(dev)).sum() + (th * wt.to(dev)).sum()).backward()
        assert torch.allclose(Rh.detach().cpu(), Ro.detach(), atol=2e-5) and torch.allclose(th.detach().cpu(), to_.detach(), atol=2e-5)
        po = dict(o.named_parameters())
        scale = max(p_.grad.abs().max().item() for p_ in po.values() if p_.grad is not None)
        for k, ph in hb.named_parameters():
            go = po[k].grad if po[k].grad is not None else torch.zeros_like(po[k])
            err = (ph.grad.cpu() - go).abs().max().item()
            assert err < 2e-4 * scale, (Bq, Tq, k, err, scale)
        # (these row counts run the LDS-DMA backward step, lstm_bwd_step.hip.)  The K-quartered small-tile form of the same launch
        # (MMEGO_LSTM_BWD_DMA=0: other row counts) sums K in another order -- same gradients to rounding -- and gives the bits of
        # the two separate launches (product, then cell backward: MMEGO_LSTM_BWD_FUSED=0)
        def grads_with(**env):
            for k_, v_ in env.items():
                monkeypatch.setenv(k_, v_)
            for ph_ in hb.parameters():
                ph_.grad = None
            R2, t2 = hb(imu.to(dev))
            ((R2 * wR.to(dev)).sum() + (t2 * wt.to(dev)).sum()).backward()
            for k_ in env:
                monkeypatch.delenv(k_)
            return {k_: ph_.grad.clone() for k_, ph_ in hb.named_parameters()}
        dma = {k: ph.grad.clone() for k, ph in hb.named_parameters()}
        kq = grads_with(MMEGO_LSTM_BWD_DMA="0")
        sep = grads_with(MMEGO_LSTM_BWD_DMA="0", MMEGO_LSTM_BWD_FUSED="0")
        for k in dma:
            assert torch.equal(kq[k], sep[k]), (Bq, Tq, k)
            assert (dma[k] - kq[k]).abs().max().item() < 2e-5 * scale, (Bq, Tq, k, (dma[k] - kq[k]).abs().max().item(), scale)
        assert any(not torch.equal(dma[k], kq[k]) for k in dma), "the two forms of the backward step should differ in summation order"
    # one optimiser step as Train_IMU.py:71-72 configures it (lr 1e-4 is the CLI's; weight_decay 1e-3)
    before = {k: v.clone() for k, v in o.state_dict().items()}
    opt_o = torch.optim.Adam(o.parameters(), lr=1e-4, weight_decay=0.001)
    opt_o.step()
    FusedAdam(hb.flat(), lr=1e-4, weight_decay=0.001).step()
    sd_h, sd_o = hb.state_dict(), o.state_dict()
    for k in ("fc3.weight", "fc3.bias"):
        assert torch.equal(sd_o[k], before[k]), "torch leaves a grad=None parameter alone"
        assert torch.equal(sd_h[k].cpu(), before[k]), k + " must not decay: it never receives a gradient"
    n_bad = n_all = 0
    for k in sd_o:
        dlt = (sd_h[k].cpu() - sd_o[k]).abs()
        assert dlt.max().item() <= 2.1e-4, (k, dlt.max().item())                  # at most a +-lr flip of a ~0 gradient
        n_bad += int((dlt > 2e-6).sum()); n_all += dlt.numel()
    assert n_bad < 0.02 * n_all, (n_bad, n_all)


@pytest.mark.parametrize("H,Bq,Tq", [(32, 8, 8), (96, 16, 8), (64, 16, 4)])
def test_lstm_bwd_step_kernel_forms_agree_at_small_sizes(dev, monkeypatch, H, Bq, Tq):
    """The LDS-DMA backward step (lstm_bwd_step.hip) at the edges of its shape range -- K = 4H of two, four and six 64-k chunks
    (the ring has three stages), one and two row blocks -- against the K-quartered small-tile form of the same launch
    (MMEGO_LSTM_BWD_DMA=0) and against the oracle's autograd."""
    from mmego_amd import nets
    torch.manual_seed(40 + H)
    o = on.IMUNet(15, 9, H, 2, True, 0).train()
    hb = nets.IMUNet(15, 9, H, 2, True, 0)
    hb.load_state_dict(o.state_dict())
    hb = hb.to(dev).train()
    gen = torch.Generator().manual_seed(H)
    imu = torch.randn(Bq, Tq, 20, 15, generator=gen)
    wR, wt = torch.randn(Bq, Tq, 3, 3, generator=gen), torch.randn(Bq, Tq, 3, generator=gen)
    Ro, to_ = o(imu)
    ((Ro * wR).sum() + (to_ * wt).sum()).backward()
    po = dict(o.named_parameters())
    scale = max(p_.grad.abs().max().item() for p_ in po.values() if p_.grad is not None)

    def grads(dma):
        monkeypatch.setenv("MMEGO_LSTM_BWD_DMA", dma)
        for ph_ in hb.parameters():
            ph_.grad = None
        R2, t2 = hb(imu.to(dev))
        ((R2 * wR.to(dev)).sum() + (t2 * wt.to(dev)).sum()).backward()
        monkeypatch.delenv("MMEGO_LSTM_BWD_DMA")
        return {k_: ph_.grad.clone() for k_, ph_ in hb.named_parameters()}
    g_dma, g_kq = grads("1"), grads("0")
    for k in g_dma:
        go = po[k].grad if po[k].grad is not None else torch.zeros_like(po[k])
        assert (g_dma[k].cpu() - go).abs().max().item() < 2e-4 * scale, ("oracle", k)
        assert (g_dma[k] - g_kq[k]).abs().max().item() < 2e-5 * scale, ("forms", k)


def test_fused_stage_step_equals_autograd_path(dev):
    """train_step.StageStep (fused L1 kernel, weight gradients on a second stream, HIP graph) produces the same
    gradients and the same Adam update as loss.backward() through the autograd bridge on one stream."""
    from mmego_amd import nets
    from mmego_amd.params import FusedAdam
    from mmego_amd.train_step import StageStep
    g = golden("g6_train.npz")
    x0, body, R, target = [T(g[k]).to(dev) for k in ("x", "body", "R", "target")]
    t_gt = target[:, :, 20].contiguous()

    def make(stage):
        torch.manual_seed(77)
        net = (nets.UpperNet() if stage == "upper" else nets.LowerNet(64)).to(dev).train()
        net.lstm_dropout = 0.0
        return net
    for stage in ("upper", "lower"):
        torch.manual_seed(78)
        frozen = nets.UpperNet().to(dev).eval()
        h0 = torch.zeros(6, 4, 64, device=dev)
        a = make(stage)
        xa = x0.clone()
        if stage == "upper":
            la = a(xa, h0, h0.clone(), body, R, t_gt)[0]
            jm = list(sk.UPPER_MAP)
        else:
            with torch.no_grad():
                up = frozen(xa, h0, h0.clone(), body, R, t_gt)[0]
            la = a(up.clone(), xa, None, None, None, None, body, R, t_gt)[0]
            jm = list(sk.LOWER_MAP)
        loss_a = (la - target[:, :, jm]).abs().sum()
        loss_a.backward()
        ga = a.flat().flat_g.clone()
        FusedAdam(a.flat(), lr=3e-5).step()
        scale = ga.abs().max().item()
        for use_graph in (False, True):
            b = make(stage)
            st = StageStep(stage, b, None, upper_frozen=frozen if stage == "lower" else None, lr=3e-5, use_graph=use_graph)
            st.bind(x0.clone(), torch.zeros(4, 8, 20, 15, device=dev), body, target, R_gt=R)
            st.step()
            assert abs(st.loss.item() - loss_a.item()) < 1e-5 * abs(loss_a.item()), (stage, use_graph)
            assert torch.allclose(b.flat().flat_g, ga, rtol=0, atol=1e-5 * scale), (stage, use_graph, (b.flat().flat_g - ga).abs().max().item(), scale)
            assert torch.allclose(b.flat().flat_p, a.flat().flat_p, rtol=0, atol=1e-7), (stage, use_graph)


def test_concurrent_stages_equal_sequential(dev):
    """train_step.ConcurrentStages (Upper and Lower bodies as two branches on two streams / of one HIP graph) gives the
    bit-identical losses, gradients and parameters as running the two StageSteps one after the other: the stages share
    nothing and every reduction in the kernels has a fixed order.  Also: sharing a net between stages is refused."""
    from mmego_amd import nets
    from mmego_amd.train_step import ConcurrentStages, StageStep
    g = golden("g6_train.npz")
    x0, body, R, target = [T(g[k]).to(dev) for k in ("x", "body", "R", "target")]
    imu_in = torch.zeros(4, 8, 20, 15, device=dev)

    def build():
        torch.manual_seed(91)
        up = nets.UpperNet().to(dev).train()
        lo = nets.LowerNet(64).to(dev).train()
        fr = nets.UpperNet().to(dev).eval()
        su = StageStep("upper", up, None, lr=3e-5, use_graph=False)
        sl = StageStep("lower", lo, None, upper_frozen=fr, lr=3e-5, use_graph=False)
        for st in (su, sl):
            st.bind(x0.clone(), imu_in, body, target, R_gt=R)
        return su, sl
    ref_u, ref_l = build()
    for _ in range(3):
        ref_u.step()
        ref_l.step()
    for use_graph in (False, True):
        su, sl = build()
        both = ConcurrentStages([su, sl], use_graph=use_graph)
        for _ in range(3):
            both.step()
        torch.cuda.synchronize()
        # (warm-up bodies before graph capture are side-effect free, so the graph run is comparable bit for bit as well)
        for a, b in ((su, ref_u), (sl, ref_l)):
            assert a.loss.item() == b.loss.item(), (use_graph, a.stage, a.loss.item(), b.loss.item())
            assert torch.equal(a.net.flat().flat_g, b.net.flat().flat_g), (use_graph, a.stage)
            assert torch.equal(a.net.flat().flat_p, b.net.flat().flat_p), (use_graph, a.stage)
            for ba, bb in zip(a.net.buffers(), b.net.buffers()):
                assert torch.equal(ba, bb), (use_graph, a.stage)
    su, sl = build()
    sl.upper_frozen = su.net
    with pytest.raises(ValueError):
        ConcurrentStages([su, sl])


def test_stage_engines_with_two_chain_recurrence_in_graphs(dev):
    """From 128 rows the BiLSTM recurrences of IMU_Net run as two chains on two streams (blocks.lstm_recurrence).  Under graph
    capture a side stream may only be forked from the capture's origin stream (ROCm defect, scripts/repro_nested_capture_fork.py),
    and the engines that overlap an IMU_Net forward with other work switch the two-chain form off (blocks.two_chains).  Here:
    B = 16, T = 8 (128 rows through rnn_fast); per-stage HIP graphs (two captured chains), ConcurrentStages and PipelinedStages
    graphs, against plain eager StageSteps run one after the other with the one-launch-per-timestep recurrence -- bit-identical
    losses, gradients, parameters and BatchNorm buffers."""
    from mmego_amd import blocks, nets
    from mmego_amd.train_step import ConcurrentStages, PipelinedStages, StageStep
    gen = torch.Generator().manual_seed(11)
    Bq, Tq = 16, 8
    x0 = torch.randn(Bq, Tq, 128, 6, generator=gen)
    x0[torch.rand(Bq, Tq, 128, generator=gen) < 0.4] = 0.0
    x0 = x0.to(dev)
    body = (0.2 * torch.randn(Bq, 20, 3, generator=gen)).to(dev)
    target = torch.randn(Bq, Tq, 21, 3, generator=gen).to(dev)
    imus = [torch.randn(Bq, Tq, 20, 15, generator=gen).to(dev) for _ in range(3)]

    def build(own_imu, use_graph=False):
        torch.manual_seed(95)
        imu_u = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
        imu_l = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
        up, lo, fr = nets.UpperNet().to(dev).train(), nets.LowerNet(64).to(dev).train(), nets.UpperNet().to(dev).eval()
        su = StageStep("upper", up, imu_u if own_imu else None, lr=3e-5, use_graph=use_graph)
        sl = StageStep("lower", lo, imu_l if own_imu else None, upper_frozen=fr, lr=3e-5, use_graph=use_graph)
        return su, sl, imu_u, imu_l

    def run(kind):
        was = blocks._LSTM_TWO_CHAINS                 # (restored as found)
        blocks._LSTM_TWO_CHAINS = kind != "reference"
        try:
            su, sl, imu_u, imu_l = build(kind != "pipelined", use_graph=kind == "stage_graphs")
            ib, inext = imus[0].clone(), imus[0].clone()
            if kind in ("reference", "stage_graphs"):
                eng = None                   # (stage_graphs: each StageStep its own HIP graph, recurrences as two captured chains)
            elif kind == "concurrent":
                eng = ConcurrentStages([su, sl], use_graph=True)
            else:
                eng = PipelinedStages([su, sl], [imu_u, imu_l], inext, use_graph=True)
            for st in (su, sl):
                st.bind(x0.clone(), ib, body, target)
            if kind == "pipelined":
                eng.prime()
            losses = []
            for i in range(2):
                ib.copy_(imus[i]); inext.copy_(imus[i + 1])
                if eng is None:
                    su.step(); sl.step()
                else:
                    eng.step()
                losses.append((su.loss.item(), sl.loss.item()))
            torch.cuda.synchronize()
            return losses, su, sl
        finally:
            blocks._LSTM_TWO_CHAINS = was
    ref_losses, ru, rl = run("reference")
    for kind in ("stage_graphs", "concurrent", "pipelined"):
        losses, su, sl = run(kind)
        assert losses == ref_losses, (kind, losses, ref_losses)
        for a, b in ((su, ru), (sl, rl)):
            assert torch.equal(a.net.flat().flat_g, b.net.flat().flat_g), (kind, a.stage)
            assert torch.equal(a.net.flat().flat_p, b.net.flat().flat_p), (kind, a.stage)
            for ba, bb in zip(a.net.buffers(), b.net.buffers()):
                assert torch.equal(ba, bb), (kind, a.stage)


def test_shared_imu_stages_equal_separate_forwards(dev):
    """train_step.SharedImuStages (one IMU_Net forward feeding both stage bodies) == each stage running its own IMU_Net
    forward with the same weights: identical losses, gradients and parameters after two steps."""
    from mmego_amd import nets
    from mmego_amd.train_step import SharedImuStages, StageStep
    g = golden("g6_train.npz")
    x0, body, target = [T(g[k]).to(dev) for k in ("x", "body", "target")]
    torch.manual_seed(5)
    imu_in = torch.randn(4, 8, 20, 15, device=dev)

    def build(own_imu):
        torch.manual_seed(92)
        imu = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
        imu2 = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
        imu2.load_state_dict(imu.state_dict())
        up, lo, fr = nets.UpperNet().to(dev).train(), nets.LowerNet(64).to(dev).train(), nets.UpperNet().to(dev).eval()
        su = StageStep("upper", up, imu if own_imu else None, lr=3e-5, use_graph=False)
        sl = StageStep("lower", lo, imu2 if own_imu else None, upper_frozen=fr, lr=3e-5, use_graph=False)
        shared = None if own_imu else SharedImuStages(imu, [su, sl], imu_in, use_graph=False)
        for st in (su, sl):
            st.bind(x0.clone(), imu_in, body, target)
        return su, sl, shared
    ru, rl, _ = build(True)
    su, sl, shared = build(False)
    for _ in range(2):
        ru.step(); rl.step()
        shared.step()
    torch.cuda.synchronize()
    for a, b in ((su, ru), (sl, rl)):
        assert a.loss.item() == b.loss.item(), a.stage
        assert torch.equal(a.net.flat().flat_g, b.net.flat().flat_g) and torch.equal(a.net.flat().flat_p, b.net.flat().flat_p), a.stage


def test_pipelined_stages_equal_concurrent_stages(dev):
    """train_step.PipelinedStages (the frozen IMU_Net forwards of minibatch i+1 overlap the stage bodies of minibatch i) gives,
    on a sequence of DIFFERENT minibatches, bit-identical losses, gradients and parameters to ConcurrentStages, eagerly and as
    a replayed HIP graph."""
    from mmego_amd import nets
    from mmego_amd.train_step import ConcurrentStages, PipelinedStages, StageStep
    g = golden("g6_train.npz")
    x0, body, target = [T(g[k]).to(dev) for k in ("x", "body", "target")]
    gen = torch.Generator().manual_seed(7)
    nb = 4
    xs = [(x0.cpu() * (1.0 + 0.05 * i)).to(dev) for i in range(nb)]                      # four different minibatches
    tg = [(target.cpu() + 0.01 * torch.randn(target.shape, generator=gen)).to(dev) for _ in range(nb)]
    im = [torch.randn(4, 8, 20, 15, generator=gen).to(dev) for _ in range(nb + 1)]

    def build(pipelined):
        torch.manual_seed(93)
        imu_u = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
        imu_l = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
        up, lo, fr = nets.UpperNet().to(dev).train(), nets.LowerNet(64).to(dev).train(), nets.UpperNet().to(dev).eval()
        su = StageStep("upper", up, None if pipelined else imu_u, lr=3e-5, use_graph=False)
        sl = StageStep("lower", lo, None if pipelined else imu_l, upper_frozen=fr, lr=3e-5, use_graph=False)
        return su, sl, imu_u, imu_l

    def run(pipelined, use_graph):
        su, sl, imu_u, imu_l = build(pipelined)
        xb, ib, tb = xs[0].clone(), im[0].clone(), tg[0].clone()                          # the static minibatch buffers
        inext = im[0].clone()
        if pipelined:
            eng = PipelinedStages([su, sl], [imu_u, imu_l], inext, use_graph=use_graph)
        else:
            eng = ConcurrentStages([su, sl], use_graph=use_graph)
        for st in (su, sl):
            st.bind(xb, ib, body, tb)
        if pipelined:
            eng.prime()
            eng.prepare()
        losses = []
        for i in range(nb):
            xb.copy_(xs[i]); ib.copy_(im[i]); tb.copy_(tg[i]); inext.copy_(im[i + 1])
            eng.step()
            losses.append((su.loss.item(), sl.loss.item()))
        torch.cuda.synchronize()
        return losses, su, sl

    ref_losses, ru, rl = run(False, False)
    assert len(set(ref_losses)) == nb                                                    # the minibatches really differ
    for use_graph in (False, True):
        losses, su, sl = run(True, use_graph)
        assert losses == ref_losses, (use_graph, losses, ref_losses)
        for a, b in ((su, ru), (sl, rl)):
            assert torch.equal(a.net.flat().flat_g, b.net.flat().flat_g), (use_graph, a.stage)
            assert torch.equal(a.net.flat().flat_p, b.net.flat().flat_p), (use_graph, a.stage)
            for ba, bb in zip(a.net.buffers(), b.net.buffers()):
                assert torch.equal(ba, bb), (use_graph, a.stage)


def test_device_resident_minibatches(dev):
    """mmego_gather_rows / data.DeviceArrays: minibatches gathered on the device equal numpy fancy indexing followed by the
    reference's float64 -> float32 conversion (bit-exact), repeated and out-of-range indices included."""
    from mmego_amd import ops
    from mmego_amd.data import DeviceArrays
    rng = np.random.RandomState(0)
    X = torch.randn(50, 37, device=dev)
    idx = torch.tensor([3, 49, 3, 0, 17, 50, -1], dtype=torch.int64, device=dev)
    Y = torch.full((7, 37), 9.0, device=dev)
    ops.gather_rows(X, idx, Y)
    assert torch.equal(Y[:5], X[idx[:5]]) and (Y[5:] == 0).all()

    class FakeSet:
        _items = [rng.randn(31, 4, 16, 6).astype(np.float32), rng.randn(31, 4, 21, 3), rng.randn(31, 20, 3), rng.randn(31, 4, 20, 15),
                  None, None, rng.randn(31, 4, 3, 3), None]

        def __len__(self):
            return 31
    ds = FakeSet()
    da = DeviceArrays(ds, dev)
    for pick in (rng.permutation(31)[:8], rng.permutation(31)[:8], rng.permutation(31)[:5]):
        b = da.gather(pick)
        for name, i in DeviceArrays.FIELDS:
            ref = torch.tensor(ds._items[i][pick], dtype=torch.float32)
            assert b[name].shape == ref.shape and torch.equal(b[name].cpu(), ref), name
    assert da.gather(np.arange(8))["data"].data_ptr() == da.gather(np.arange(8, 16))["data"].data_ptr()   # static per batch size


def test_large_batch_stress_forward_property(dev):
    """BASELINE config 5 shape (B=2048, T=16, N=256; fp32 here): Upper_Net + Lower_Net eval forward on 8.4 M points.
    Size-independent property: in eval mode every sequence is independent, so the first 4 sequences of the big
    batch must equal the same 4 sequences run alone (bit for bit), and all outputs must be finite."""
    from mmego_amd import nets
    torch.manual_seed(5)
    up, lo = nets.UpperNet().to(dev).eval(), nets.LowerNet(64).to(dev).eval()
    B, Tn, N = 2048, 16, 256
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn(B, Tn, N, 6, generator=g)
    x[torch.rand(B, Tn, N, generator=g) < 0.4] = 0.0
    body = 0.2 * torch.randn(1, 20, 3, generator=g).repeat(B, 1, 1)       # identical bodies (Q2 pairs frame n with body n % B)
    ang = torch.rand(B, Tn, generator=g)
    R = torch.zeros(B, Tn, 3, 3)
    R[..., 0, 0], R[..., 0, 1], R[..., 1, 0], R[..., 1, 1], R[..., 2, 2] = ang.cos(), -ang.sin(), ang.sin(), ang.cos(), 1.0
    t = 0.1 * torch.randn(B, Tn, 3, generator=g)
    with torch.no_grad():
        xd = x.to(dev)
        h0 = torch.zeros(6, B, 64, device=dev)
        lu = up(xd, h0, h0.clone(), body.to(dev), R.to(dev), t.to(dev))[0]
        ll = lo(lu.clone(), xd, None, None, None, None, body.to(dev), R.to(dev), t.to(dev))[0]
        assert torch.isfinite(lu).all() and torch.isfinite(ll).all()
        xs = x[:4].clone().to(dev)
        h0s = torch.zeros(6, 4, 64, device=dev)
        lus = up(xs, h0s, h0s.clone(), body[:4].to(dev), R[:4].to(dev), t[:4].to(dev))[0]
        lls = lo(lus.clone(), xs, None, None, None, None, body[:4].to(dev), R[:4].to(dev), t[:4].to(dev))[0]
    assert torch.equal(xd[:4], xs), "in-place transform identical"
    assert torch.allclose(lu[:4], lus, rtol=0, atol=1e-6) and torch.allclose(ll[:4], lls, rtol=0, atol=1e-6)
