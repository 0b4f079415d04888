"""CPU: the oracle restatement equals the real reference on the committed golden vectors.

The goldens were produced by tests/golden/make_golden.py importing /root/reference;
these tests never touch the reference.
"""
import re

import numpy as np
import pytest
import torch

from conftest import check_pinned, golden, load_weights, set_lstm_dropout
from oracle import geometry as geo
from oracle import graph as gr
from oracle import metric as om
from oracle import nets as on
from oracle import train as ot
from oracle import skeleton as sk

# parameters whose true gradient is exactly zero in train mode (bias before a batch-stat BN, bias of a
# softmax-pooled score, the Q6 gate): their measured gradient is rounding noise, so Adam turns it into
# +-lr steps that no two implementations agree on.  They cannot change any output.
NOISE_GRAD = re.compile(r"(conv[123]\.bias|tcn\.2\.bias|residual\.0\.bias|attn\.bias|to_k\.bias|fusion\.attn\.weight)$")


def T(a):
    return torch.tensor(np.asarray(a))


def assert_indices_equal_modulo_ties(idx, ref_idx, keys):
    """Index parity rule.  The reference sorts with torch.sort(stable=False); on exactly equal keys
    (duplicate radar points, +inf dead points, identical padded rows) its order is unspecified and
    differs between torch builds.  So: the selected KEY sequence must be bit-identical position by
    position, and wherever a selected key is unique in its row the INDEX must be identical."""
    k_mine = torch.gather(keys, -1, idx)
    k_ref = torch.gather(keys, -1, ref_idx)
    assert torch.equal(k_mine, k_ref), "selected key sequences differ"
    n_equal = (keys.unsqueeze(-2) == k_ref.unsqueeze(-1)).sum(-1)   # multiplicity of each selected key in its row
    unique = n_equal == 1
    assert torch.equal(idx[unique], ref_idx[unique]), "indices differ on untied keys"
    return unique.float().mean().item()


def test_g1_transforms():
    g = golden("g1_transforms.npz")
    pts = T(g["pts"]).clone()
    out = geo.transform_to_head_(pts, T(g["R"]), T(g["t"]))
    assert torch.allclose(out, T(g["out_h"]), atol=1e-6)
    assert torch.allclose(pts, T(g["mutated"]), atol=1e-6), "Q1: the caller's tensor must be mutated"
    assert out.data_ptr() == pts.data_ptr()
    back = geo.transform_to_world(T(g["joints"]), T(g["R"]), T(g["t"]))
    assert torch.allclose(back, T(g["out_r"]), atol=1e-6)


def test_g2_grouping_indices_bit_exact():
    g = golden("g2_grouping.npz")
    xyz, feats = T(g["xyz"]), T(g["feats"])
    assert torch.equal(geo.anchor_grid(), T(g["anchors"]))
    anchors = geo.anchor_grid().unsqueeze(0).expand(xyz.shape[0], -1, -1)
    d = geo.square_distance(anchors, xyz)
    ref = T(g["dist"])
    assert torch.equal(torch.isinf(d), torch.isinf(ref))
    fin = ~torch.isinf(ref)
    assert torch.allclose(d[fin], ref[fin], rtol=0, atol=1e-6)
    assert torch.equal(d[fin], ref[fin]), "distances must be bit-identical (same formula, same order)"
    grouped, idx = geo.anchor_grouping(xyz, feats, 8)
    assert idx.dtype == torch.int64
    assert_indices_equal_modulo_ties(idx, T(g["idx"]), ref)
    same = (idx == T(g["idx"])).all(dim=-1)                     # whole groups with identical indices
    assert same.float().mean() > 0.5
    assert torch.equal(grouped[same], T(g["grouped"])[same])


def test_g3_adjacency():
    g = golden("g3_adjacency.npz")
    assert np.array_equal(gr.adjacency("distance"), g["distance"])
    assert np.array_equal(gr.adjacency("uniform"), g["uniform"])
    A = gr.adjacency("distance")
    assert A.shape == (2, 15, 15) and np.count_nonzero(A[1]) == 28
    assert abs(A[1][0, 1] - 0.2887) < 1e-4 and abs(A[1][3, 14] - 0.4082) < 1e-4


@pytest.fixture(scope="module")
def pretrained():
    up = load_weights(on.UpperNet(), golden("w_upper_pretrained.npz")).eval()
    lo = load_weights(on.LowerNet(64), golden("w_lower_pretrained.npz")).eval()
    return up, lo


def test_state_dict_keys_match_shipped_checkpoints():
    up, lo = golden("w_upper_pretrained.npz"), golden("w_lower_pretrained.npz")
    assert set(on.UpperNet().state_dict()) == set(up.files) and len(up.files) == 72
    assert set(on.LowerNet(64).state_dict()) == set(lo.files)
    for k, v in on.LowerNet(64).state_dict().items():
        assert tuple(v.shape) == lo[k].shape, k


def test_g4_g5_eval_forward(pretrained, real16):
    up, lo = pretrained
    g4, g5 = golden("g4_upper_eval.npz"), golden("g5_lower_eval.npz")
    with torch.no_grad():
        for i in range(8):
            x = T(real16["x"][i:i + 1]).clone()
            tgt = T(real16["target"][i:i + 1])
            skl, R = T(real16["skl"][i:i + 1]), T(real16["R"][i:i + 1])
            t = tgt[:, :, 20].contiguous()
            h0, c0 = ot.zeros_state(1)
            cap = {}
            hooks = [up.module0.register_forward_hook(lambda m, a, o: cap.__setitem__("feats", o)),
                     up.module1.register_forward_hook(lambda m, a, o: cap.__setitem__("seq", o[0]))]
            l, q, gw, hn, cn = up(x, h0, c0, skl, R, t)
            for h in hooks:
                h.remove()
            for name, got in dict(l=l, q=q, gw=gw, hn=hn, cn=cn, feats=cap["feats"], seq=cap["seq"], x_after=x).items():
                ref = T(g4["%s_%d" % (name, i)])
                assert torch.allclose(got.reshape(ref.shape), ref, rtol=1e-4, atol=2e-5), (name, i)
            ref_sel = T(real16["ref_sel_idx"][i])
            ll, lq = lo(l.clone(), x, h0, c0, h0, c0, skl, R, t, pin_select_idx=ref_sel)
            assert torch.allclose(x, T(g5["x_after_%d" % i]), atol=1e-5), "Q1: second in-place transform"
            assert torch.allclose(ll, T(g5["l_%d" % i]), rtol=1e-4, atol=2e-5), i
            assert torch.allclose(lq, T(g5["q_%d" % i]), rtol=1e-4, atol=2e-5), i
            if i < 2:
                keys = x.view(20, 128, 6)[:, :, 0].unsqueeze(1)            # x after both transforms
                assert_indices_equal_modulo_ties(lo.last_select_idx.unsqueeze(1), ref_sel.unsqueeze(1), keys)
                assert torch.allclose(lo.last_p_vec, T(g5["p_vec_%d" % i]).view(20, 64, 64), rtol=1e-4, atol=2e-5)
                assert torch.allclose(lo.last_k_vec, T(g5["k_vec_%d" % i]).view(20, 15, 64), rtol=1e-4, atol=2e-5), \
                    "Q8: ST-GCN output is re-viewed, not permuted"


def test_g9_end_to_end_error_cm(pretrained, real16):
    up, lo = pretrained
    g9 = golden("g9_end2end.npz")
    rows = []
    with torch.no_grad():
        for i in range(16):
            x = T(real16["x"][i:i + 1]).clone()
            tgt = T(real16["target"][i:i + 1])
            skl, R = T(real16["skl"][i:i + 1]), T(real16["R"][i:i + 1])
            t = tgt[:, :, 20].contiguous()
            h0, c0 = ot.zeros_state(1)
            l = up(x, h0, c0, skl, R, t)[0]
            ll = lo(l.clone(), x, h0, c0, h0, c0, skl, R, t, pin_select_idx=T(real16["ref_sel_idx"][i]))[0]
            rows.append(om.batch_errors(l, ll, tgt))
    s = om.summarize(rows)
    for k in ("all_cm", "upper_cm", "lower_cm"):
        assert abs(s[k] - float(g9[k])) < 1e-3, (k, s[k], float(g9[k]))     # the north-star bar: 1e-3 cm
    assert abs(s["rot_deg"] - float(g9["rot_deg"])) < 1e-3
    assert np.allclose(s["per_joint_cm"], g9["per_joint_cm"], atol=1e-3)


def test_g8_metric():
    g = golden("g8_metric.npz")
    pred, target = T(g["pred"]), T(g["target"])
    a, u, l, pj, ang = om.batch_errors(pred[:, :, list(sk.UPPER_MAP)], pred[:, :, list(sk.LOWER_MAP)], target)
    assert abs(a - float(g["accu"])) < 1e-7 and abs(u - float(g["upper"])) < 1e-7 and abs(l - float(g["lower"])) < 1e-7
    assert np.allclose(pj, g["accu_l"], atol=1e-7) and np.allclose(ang, g["angle_l"], atol=1e-4)


def _train_case(tag, seed, ctor, fwd, tgt_map, g):
    torch.manual_seed(seed)
    model = ctor()
    set_lstm_dropout(model, 0.0)
    model.train()
    for k, v in model.state_dict().items():
        if v.numel() > 1:
            chk = g["%s.init.%s#c" % (tag, k)]
            assert abs(v.double().sum().item() - chk[0]) <= 1e-9 * max(1, abs(chk[0])), \
                "seeded init differs from the reference for " + k
    opt = torch.optim.Adam(model.parameters(), lr=3e-5)
    target = T(g["target"])[:, :, list(tgt_map)]
    for step in (1, 2, 3):
        opt.zero_grad()
        joints = fwd(model)
        loss = ot.l1_sum(joints, target)
        loss.backward()
        assert abs(loss.item() - float(g["%s.loss%d" % (tag, step)])) < 2e-4 * abs(float(g["%s.loss%d" % (tag, step)]))
        if step == 1:
            assert torch.allclose(joints, T(g["%s.l" % tag]), rtol=1e-4, atol=2e-5)
            grads = [(k, p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()]
            scale = max(gr_.abs().max().item() for _, gr_ in grads)
            # the reference's own fp32 gradients sit ~3e-3*scale from the fp64 truth on the BatchNorm chains
            # (measured: tests/test_oracle_golden.py::test_oracle_fp32_gradients_close_to_fp64), so that is the pin
            check_pinned(g, "%s.grad." % tag, grads, rtol=5e-3, atol=5e-3 * scale)
        opt.step()
        if step in (1, 3):
            sd = [(k, v) for k, v in model.state_dict().items()
                  if v.dtype.is_floating_point and not NOISE_GRAD.search(k)]
            check_pinned(g, "%s.step%d." % (tag, step), sd, rtol=1e-5, atol=2e-6, bad_frac=0.10, hard_atol=6e-5 * step + 2e-6)


def test_g6_train_upper():
    g = golden("g6_train.npz")
    x0, body, R, t = T(g["x"]), T(g["body"]), T(g["R"]), T(g["t"])
    h0, c0 = ot.zeros_state(4)
    _train_case("upper", 601, on.UpperNet, lambda m: m(x0.clone(), h0, c0, body, R, t)[0], sk.UPPER_MAP, g)


def test_g6_train_upper_wlocal_and_group_indices():
    g = golden("g6_train.npz")
    x0, body, R, t = T(g["x"]), T(g["body"]), T(g["R"]), T(g["t"])
    h0, c0 = ot.zeros_state(4)
    holder = {}

    def fwd(m):
        out = m(x0.clone(), h0, c0, h0, c0, body, R, t)
        holder.setdefault("idx", m.module2.last_group_idx.clone())
        return out[0]
    _train_case("wlocal", 602, on.UpperNetwlocal, fwd, sk.UPPER_MAP, g)
    with torch.no_grad():
        xh = geo.transform_to_head_(x0.clone(), R, t)[..., :3].contiguous()
        keys = geo.square_distance(geo.anchor_grid().unsqueeze(0).expand(32, -1, -1), xh)
    frac = assert_indices_equal_modulo_ties(holder["idx"], T(g["wlocal.group_idx"]), keys)
    assert frac > 0.9


def test_g6_train_lower():
    g = golden("g6_train.npz")
    body, R, t = T(g["body"]), T(g["R"]), T(g["t"])
    up_l, x_l = T(g["lower.upper_in"]), T(g["lower.x_in"])
    h0, c0 = ot.zeros_state(4)
    # the frozen Upper that produced lower.upper_in is itself reproducible from its seed
    torch.manual_seed(603)
    upf = on.UpperNet().eval()
    with torch.no_grad():
        xx = T(g["x"]).clone()
        assert torch.allclose(upf(xx, h0, c0, body, R, t)[0], up_l, rtol=1e-4, atol=2e-5)
        assert torch.allclose(xx, x_l, atol=1e-5)
    _train_case("lower", 604, lambda: on.LowerNet(64),
                lambda m: m(up_l.clone(), x_l.clone(), h0, c0, h0, c0, body, R, t)[0], sk.LOWER_MAP, g)


def test_g7_imu():
    g = golden("g7_imu.npz")
    imu = T(g["imu"])
    small = load_weights(on.IMUNet(15, 9, 32, 2, True, 0.1), g, "small.w.").eval()
    with torch.no_grad():
        R, t = small(imu)
    assert torch.allclose(R, T(g["small.R"]), atol=2e-6) and torch.allclose(t, T(g["small.t"]), atol=2e-6)
    tr = load_weights(on.IMUNet(15, 9, 32, 2, True, 0), g, "train.w.").train()
    R, t = tr(imu)
    loss = ot.imu_loss(R, t, T(g["train.R_gt"]), T(g["train.head_gt"]))
    assert abs(loss.item() - float(g["train.loss"])) < 1e-4 * abs(float(g["train.loss"]))
    loss.backward()
    grads = [(k, p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in tr.named_parameters()]
    scale = max(x.abs().max().item() for _, x in grads)
    check_pinned(g, "train.grad.", grads, rtol=2e-3, atol=2e-5 * scale)
    # full-size instance from its seed: same construction order => same weights as the reference
    torch.manual_seed(703)
    big = on.IMUNet(15, 9, 512, 2, True, 0.1).eval()
    for k in ("fc1.weight", "rnn_fast.weight_hh_l1_reverse", "rnn_slow.weight_ih_l0"):
        v = big.state_dict()[k].double()
        assert abs(v.sum().item() - g["big.chk." + k][0]) < 1e-9 and abs(v.norm().item() - g["big.chk." + k][1]) < 1e-9
    with torch.no_grad():
        R, t = big(imu)
    assert torch.allclose(R, T(g["big.R"]), atol=5e-6) and torch.allclose(t, T(g["big.t"]), atol=5e-6)


def test_adam_restatement_equals_torch_optim():
    torch.manual_seed(5)
    for wd in (0.0, 1e-3):
        p = torch.randn(257, requires_grad=True)
        ref_p = p.detach().clone().requires_grad_(True)
        opt = torch.optim.Adam([ref_p], lr=3e-5, weight_decay=wd)
        m, v = torch.zeros(257), torch.zeros(257)
        mine = p.detach().clone()
        for step in range(1, 6):
            grad = torch.randn(257) * (10.0 ** (step - 3))
            ref_p.grad = grad.clone()
            opt.step()
            ot.adam_update_(mine, grad, m, v, step, 3e-5, weight_decay=wd)
            assert torch.allclose(mine, ref_p.detach(), rtol=0, atol=1e-7)


def test_oracle_fp32_gradients_close_to_fp64():
    """The oracle's fp32 gradients are ~1e-5*scale from an fp64 run of the same code (the reference's
    fp32 path is ~3e-3 away: tighter pins than that against the reference are meaningless)."""
    g = golden("g6_train.npz")

    def run(dtype):
        torch.manual_seed(601)
        m = on.UpperNet()
        set_lstm_dropout(m, 0.0)
        m = m.train().to(dtype)
        x0, body, R, t = [T(g[k]).to(dtype) for k in ("x", "body", "R", "t")]
        h0 = torch.zeros(6, 4, 64, dtype=dtype)
        joints = m(x0.clone(), h0, h0.clone(), body, R, t)[0]
        ot.l1_sum(joints, T(g["target"]).to(dtype)[:, :, list(sk.UPPER_MAP)]).backward()
        return {k: p.grad.double() for k, p in m.named_parameters()}
    g32, g64 = run(torch.float32), run(torch.float64)
    scale = max(v.abs().max().item() for v in g64.values())
    for k in g64:
        assert (g32[k] - g64[k]).abs().max().item() < 1e-4 * scale, k


def test_g10_oracle_trains_like_the_reference(real16):
    """G10 (tests/golden/make_dropout_band.py): 200 dropout-FREE Upper_Net training steps of the oracle, live, against the REAL
    reference's recorded deterministic run -- final train-set joint error and the loss at steps 50/100/150/200 -- and the
    oracle's recorded dropout-active runs against the reference's band.

    Runs on ONE torch thread (r06).  (a) The intermittent "stall" of the CPU suite was this test under CPU
    contention: its ~40 000 tiny OpenMP regions per 200 steps each end in a spin barrier, and with another busy process on the
    container's 8 CPUs (a hipcc build, a second pytest) every barrier waits for a descheduled worker -- 10 s alone, > 180 s beside
    8 busy processes (profiles/r06_cpu_stall.txt; native stacks of the workers: tests/stall_probe.py), 22 s single-threaded either
    way.  (b) The trajectory is chaotic in the rounding noise: the final error moves by up to 0.2 cm with the thread count alone
    (4.03 / 4.12 / 4.09 / 4.04 / 4.07 / 4.25 / 4.02 / 4.03 cm at 1..8 threads against the reference's 4.04), so the 0.05-cm bar held
    at 8 threads by luck of the partition; one thread fixes the summation order on every machine."""
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        _g10_body(real16)
    finally:
        torch.set_num_threads(nthreads)


def _g10_body(real16):
    band = golden("g10_dropout_band.npz")
    x0, target, body, R = [torch.tensor(real16[k]) for k in ("x", "target", "skl", "R")]
    t = target[:, :, 20].contiguous()
    h0, c0 = ot.zeros_state(x0.shape[0])
    tgt = target[:, :, list(sk.UPPER_MAP)]
    torch.manual_seed(int(band["init_upper"]))
    net = on.UpperNet().train()
    set_lstm_dropout(net, 0.0)
    opt = torch.optim.Adam(net.parameters(), lr=float(band["lr"]))
    curve = []
    for s in range(1, int(band["steps"]) + 1):
        opt.zero_grad()
        loss = ot.l1_sum(net(x0.clone(), h0, c0, body, R, t)[0], tgt)
        loss.backward()
        opt.step()
        if s % 50 == 0:
            curve.append(loss.item())
    net.eval()
    with torch.no_grad():
        err = (net(x0.clone(), h0, c0, body, R, t)[0] - tgt).norm(dim=-1).mean().item() * 100.0
    assert abs(err - float(band["ref.upper.p00.err_cm"][0])) < 0.05, (err, band["ref.upper.p00.err_cm"])
    for i, (a, b) in enumerate(zip(curve, band["ref.upper.p00.loss_curve"][0])):      # (trajectories part ways slowly: 1e-4 at step 50)
        assert abs(a - b) < (1e-2 if i == 0 else 5e-2) * abs(b), (curve, band["ref.upper.p00.loss_curve"][0])
    # the recorded dropout-active oracle runs: trajectories of one seed part ways between the two sides (rounding differences,
    # amplified by 200 Adam steps), so the comparison is of distributions -- the oracle's mean inside the reference's band, every
    # oracle run inside the band widened by its width
    for stage in ("upper", "lower"):
        o, r = band["oracle.%s.p01.err_cm" % stage], band["ref.%s.p01.err_cm" % stage]
        lo, hi = float(r.min()), float(r.max())
        assert lo <= float(o.mean()) <= hi, (stage, o, r)
        assert (o > lo - (hi - lo)).all() and (o < hi + (hi - lo)).all(), (stage, o, r)
