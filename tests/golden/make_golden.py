#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Run once in the build container (the reference lives at /root/reference and
never travels):   python tests/golden/make_golden.py

Only data (inputs / expected outputs / weights as arrays) is written; no
reference source is copied.  Import recipe: SURVEY.md section 8-c.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("MMEGO_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import matplotlib  # noqa: E402

matplotlib.use("Agg")
for _m in ("seaborn", "imageio", "imageio.v2"):      # plot-only deps of Utils.py, not installed
    sys.modules.setdefault(_m, types.ModuleType(_m))

import torch  # noqa: E402

torch.set_num_threads(8)

from Config.config import Config  # noqa: E402
from Net.GCN import Graph  # noqa: E402
from Net.IMU_Net import IMUNet  # noqa: E402
from Net.Lower_Net import LowerNet  # noqa: E402
from Net import Upper_Net as RU  # noqa: E402
from Net.Upper_Net import UpperNet, UpperNetwlocal  # noqa: E402
from Util.Universal_Util.Utils import Transform2H, Transform2R  # noqa: E402


def sd_np(module, prefix="w."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print("%-28s %8.1f kB" % (name, os.path.getsize(path) / 1e3))


def summarize(t, n):
    """Compact pin for a big tensor: sum, l2, stride, then a strided sample of n values (float64)."""
    a = t.detach().double().flatten()
    stride = max(1, a.numel() // n)
    return np.concatenate(([a.sum().item(), a.norm().item(), float(stride)], a[::stride][:n].numpy()))


def pin(d, key, t, full_limit=9000, n=1024):
    if t.numel() <= full_limit:
        d[key] = t.detach().cpu().numpy().copy()      # copy: .numpy() aliases live parameters
    else:
        d[key + "#s"] = summarize(t, n)


# --------------------------------------------------------------------------
def g1_transforms():
    g = torch.Generator().manual_seed(11)
    B, T, N = 3, 4, 5
    pts = torch.randn(B, T, N, 6, generator=g)
    ang = 0.3
    Rz = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], dtype=torch.float32)
    Rx = torch.tensor([[1, 0, 0], [0, np.cos(0.7), -np.sin(0.7)], [0, np.sin(0.7), np.cos(0.7)]], dtype=torch.float32)
    R = torch.stack([Rz @ torch.matrix_power(Rx, k) for k in range(B * T)]).view(B, T, 3, 3).contiguous()
    t = 0.1 * torch.randn(B, T, 3, generator=g)
    inp = pts.clone()
    out_h = Transform2H(inp, B, T, N, R, t)
    joints = torch.randn(B, T, 7, 3, generator=g)
    out_r = Transform2R(joints.clone(), B, T, 7, R, t)
    save("g1_transforms.npz", pts=pts.numpy(), R=R.numpy(), t=t.numpy(), out_h=out_h.numpy(),
         mutated=inp.numpy(), joints=joints.numpy(), out_r=out_r.numpy())


def g2_grouping(real_x):
    g = torch.Generator().manual_seed(12)
    real = torch.tensor(real_x[:4, 0])                               # 4 real frames (128,6)
    syn = torch.randn(4, 128, 3, generator=g) * 0.4 + torch.tensor([0.3, 0.0, 0.0])
    syn[0, 5:] = 0.0                                                   # only 5 live points (<8)
    syn[1, ::3] = 0.0                                                  # exact-zero rows sprinkled
    syn[2, 10] = syn[2, 20]                                            # exact duplicate point
    xyz = torch.cat((real[:, :, :3], syn), 0).contiguous()
    feats = torch.randn(8, 128, 25, generator=g)
    anchors = RU.AnchorInit().cpu().view(1, 27, 3).repeat(8, 1, 1)
    dist = RU.square_distance(anchors, xyz)
    idx = RU.point_ball_set(8, xyz, anchors)
    grouped = RU.AnchorGrouping(anchors, 8, xyz, feats)
    save("g2_grouping.npz", xyz=xyz.numpy(), feats=feats.numpy(), anchors=anchors[0].numpy(),
         dist=dist.numpy(), idx=idx.numpy(), grouped=grouped.numpy())


def g3_adjacency():
    save("g3_adjacency.npz", distance=Graph(layout="kinect_upper", strategy="distance").A,
         uniform=Graph(layout="kinect_upper", strategy="uniform").A)


def load_real():
    from Util.Universal_Util.Dataset_sample import PosePC
    np.random.seed(0)
    ds = PosePC(train=False, vis=True, batch_length=20)
    sel = np.arange(16) * 52
    real = dict(x=ds.data_ti_[sel].astype(np.float32), target=ds.data_key_[sel].astype(np.float32),
                skl=ds.skl_[sel].astype(np.float32), R=ds.R_R0R_[sel].astype(np.float32),
                imu=ds.imu_[sel].astype(np.float32), sel=sel)
    return real, ds


def demo_metrics(upper_l, lower_l, target):
    """The per-batch figures of Demo_test.eval_model (:121-123,:150-163), evaluated with the reference's Config."""
    up_map, lo_map = Config.upper_joint_map, Config.lower_joint_map
    sk_all = torch.tensor(Config.skeleton_all)
    root, leaf = sk_all[:, 0], sk_all[:, 1]
    pred = torch.zeros((upper_l.shape[0], upper_l.shape[1], 21, 3), dtype=torch.float32)
    pred[:, :, up_map, :] = upper_l
    pred[:, :, lo_map, :] = lower_l
    pv = pred[:, :, [l for l in leaf], :] - pred[:, :, [l for l in root], :]
    tv = target[:, :, [l for l in leaf], :] - target[:, :, [l for l in root], :]
    cs = torch.nn.functional.cosine_similarity(pv, tv, dim=-1)
    ang = torch.abs(torch.acos(torch.clamp(cs, min=-1.0, max=1.0)) / 3.14159265358 * 180.0)
    accu_a = torch.sqrt(torch.sum(torch.square(pred - target), dim=-1))
    au = torch.mean(torch.sqrt(torch.sum(torch.square(upper_l - target[:, :, up_map]), dim=-1))).item()
    al = torch.mean(torch.sqrt(torch.sum(torch.square(lower_l - target[:, :, lo_map]), dim=-1))).item()
    return (torch.mean(accu_a).item(), au, al, torch.mean(accu_a, dim=0).mean(dim=0).numpy().tolist(),
            torch.mean(ang, dim=0).mean(dim=0).numpy().tolist())


def g4_g5_g9(real):
    up = UpperNet()
    up.load_state_dict(torch.load(Config.model_upper_path, map_location="cpu"))
    lo = LowerNet(hidden_dim=64)
    lo.load_state_dict(torch.load(Config.model_lower_path, map_location="cpu"))
    up.eval(), lo.eval()
    save("w_upper_pretrained.npz", **sd_np(up, ""))
    save("w_lower_pretrained.npz", **sd_np(lo, ""))

    cap = {}
    up.module0.register_forward_hook(lambda m, i, o: cap.__setitem__("feats", o.detach().clone()))
    up.module1.gpointnet.register_forward_hook(lambda m, i, o: cap.__setitem__("gvec", o[0].detach().clone()))
    up.module1.register_forward_hook(lambda m, i, o: cap.__setitem__("seq", o[0].detach().clone()))
    up.mlpHead.register_forward_hook(lambda m, i, o: cap.__setitem__("head", o[1].detach().clone()))
    lo.pointEncoder.register_forward_hook(lambda m, i, o: cap.__setitem__("p_vec", o.detach().clone()))
    lo.keyEncoder.register_forward_hook(lambda m, i, o: cap.__setitem__("k_vec", o.detach().clone()))

    g4, g5, errs, sel = {}, {}, [], []
    with torch.no_grad():
        for i in range(16):
            x = torch.tensor(real["x"][i:i + 1])
            tgt = torch.tensor(real["target"][i:i + 1])
            skl = torch.tensor(real["skl"][i:i + 1])
            R = torch.tensor(real["R"][i:i + 1])
            t = tgt[:, :, 20].contiguous()
            h0 = torch.zeros(6, 1, 64)
            c0 = torch.zeros(6, 1, 64)
            l, q, gw, hn, cn = up(x, h0, c0, skl, R, t)
            x_after_upper = x.clone()
            up_l = l.clone().detach()
            ll, lq = lo(up_l, x, h0, c0, h0, c0, skl, R, t)
            errs.append(demo_metrics(up_l, ll, tgt))
            if i < 8:
                for k, v in dict(l=l, q=q, gw=gw, hn=hn, cn=cn, feats=cap["feats"], gvec=cap["gvec"],
                                 seq=cap["seq"], head=cap["head"], x_after=x_after_upper).items():
                    g4["%s_%d" % (k, i)] = v.numpy()
                g5["l_%d" % i] = ll.numpy()
                g5["q_%d" % i] = lq.numpy()
                g5["x_after_%d" % i] = x.numpy()
                if i < 2:
                    g5["p_vec_%d" % i] = cap["p_vec"].numpy()
                    g5["k_vec_%d" % i] = cap["k_vec"].numpy()
            # the reference's own top-64 choice (torch.sort, stable=False: tie order is build-specific)
            xs = x_after_upper.view(20, 128, 6).clone()
            xs = Transform2H(xs, 1, 20, 128, R, t)
            sel.append(torch.sort(xs[..., 0], dim=1, descending=True)[1][:, :64].numpy())
    real = dict(real)
    real["ref_sel_idx"] = np.stack(sel)
    ang = np.mean([e[4] for e in errs], axis=0)
    s = dict(all_cm=float(np.mean([e[0] for e in errs]) * 100), upper_cm=float(np.mean([e[1] for e in errs]) * 100),
             lower_cm=float(np.mean([e[2] for e in errs]) * 100), rot_deg=float(sum(ang) / len(ang)),
             per_joint_cm=(np.mean([e[3] for e in errs], axis=0) * 100).tolist())
    save("real16.npz", **real)
    save("g4_upper_eval.npz", **g4)
    save("g5_lower_eval.npz", **g5)
    save("g9_end2end.npz", all_cm=s["all_cm"], upper_cm=s["upper_cm"], lower_cm=s["lower_cm"],
         rot_deg=s["rot_deg"], per_joint_cm=np.asarray(s["per_joint_cm"]),
         per_batch_all=np.asarray([e[0] for e in errs]))
    print("g9: all %.4f upper %.4f lower %.4f cm" % (s["all_cm"], s["upper_cm"], s["lower_cm"]))


def synth_batch(seed, B, T, N, distinct_body, real):
    g = torch.Generator().manual_seed(seed)
    mu = torch.tensor([0.84, 0.05, 0.18])
    sd = torch.tensor([0.41, 0.30, 0.38])
    xyz = torch.randn(B, T, N, 3, generator=g) * sd + mu
    x = torch.zeros(B, T, N, 6)
    x[..., :3] = xyz
    x[..., 3] = xyz.norm(dim=-1)
    x[..., 4] = torch.randn(B, T, N, generator=g) * 0.41
    x[..., 5] = torch.rand(B, T, N, generator=g) * 36 + 10
    dead = torch.rand(B, T, N, generator=g) < 0.4
    x[dead] = 0.0
    body = torch.tensor(real["skl"][0]).unsqueeze(0).repeat(B, 1, 1)
    if distinct_body:
        body = body * (1 + 0.2 * torch.rand(B, 1, 1, generator=g)) + 0.01 * torch.randn(B, 20, 3, generator=g)
    ax = torch.randn(B, T, 3, generator=g)
    ax = ax / ax.norm(dim=-1, keepdim=True)
    ang = 0.5 * torch.randn(B, T, 1, generator=g)
    K = torch.zeros(B, T, 3, 3)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -ax[..., 2], ax[..., 1], ax[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -ax[..., 0], -ax[..., 1], ax[..., 0]
    R = torch.eye(3) + torch.sin(ang)[..., None] * K + (1 - torch.cos(ang))[..., None] * (K @ K)
    t = torch.tensor([0.0, 0.0, 1.5]) + 0.1 * torch.randn(B, T, 3, generator=g)
    target = torch.tensor(real["target"][0, :T]).unsqueeze(0).repeat(B, 1, 1, 1) + 0.05 * torch.randn(B, T, 21, 3, generator=g)
    return x.contiguous(), body.contiguous(), R.contiguous(), t.contiguous(), target.contiguous()


def set_lstm_dropout(model, p=0.0):
    for m in model.modules():
        if isinstance(m, torch.nn.LSTM):
            m.dropout = p


def g6_train(real):
    B, T, N = 4, 8, 128
    x0, body, R, t, target = synth_batch(61, B, T, N, True, real)
    upper_map, lower_map = Config.upper_joint_map, Config.lower_joint_map
    out = dict(x=x0.numpy(), body=body.numpy(), R=R.numpy(), t=t.numpy(), target=target.numpy())
    h0 = torch.zeros(6, B, 64)
    c0 = torch.zeros(6, B, 64)
    loss_fn = torch.nn.L1Loss(reduction="sum")

    def run(tag, model, fwd, tgt):
        set_lstm_dropout(model, 0.0)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=3e-5)
        for k, v in model.state_dict().items():
            if v.numel() > 1:
                out["%s.init.%s#c" % (tag, k)] = np.asarray([v.double().sum().item(), v.double().norm().item()])
        for step in (1, 2, 3):
            opt.zero_grad()
            l = fwd(model)
            loss = loss_fn(l, tgt)
            loss.backward()
            if step == 1:
                out["%s.loss" % tag] = np.asarray(loss.item())
                out["%s.l" % tag] = l.detach().numpy()
                for k, p in model.named_parameters():
                    pin(out, "%s.grad.%s" % (tag, k), p.grad if p.grad is not None else torch.zeros_like(p))
            opt.step()
            if step in (1, 3):
                for k, v in model.state_dict().items():
                    if v.dtype.is_floating_point:
                        pin(out, "%s.step%d.%s" % (tag, step, k), v, full_limit=512, n=256)
            out["%s.loss%d" % (tag, step)] = np.asarray(loss.item())

    torch.manual_seed(601)
    up = UpperNet()
    run("upper", up, lambda m: m(x0.clone(), h0, c0, body, R, t)[0], target[:, :, upper_map])

    torch.manual_seed(602)
    upl = UpperNetwlocal()
    grp = {}

    def fwd_local(m):
        r = m(x0.clone(), h0, c0, h0, c0, body, R, t)
        return r[0]
    run("wlocal", upl, fwd_local, target[:, :, upper_map])
    # group indices of the first forward (initial weights do not matter for them: indices depend on xyz only)
    with torch.no_grad():
        xh = Transform2H(x0.clone(), B, T, N, R, t)
        anchors = RU.AnchorInit().cpu().view(1, 27, 3).repeat(B * T, 1, 1)
        out["wlocal.group_idx"] = RU.point_ball_set(8, xh[..., :3].contiguous(), anchors).numpy()

    # Lower: input upper joints from a frozen seeded UpperNet in eval mode (as Train_Lower does)
    torch.manual_seed(603)
    upf = UpperNet().eval()
    torch.manual_seed(604)
    lo = LowerNet(hidden_dim=64)
    with torch.no_grad():
        x_l = x0.clone()
        up_l = upf(x_l, h0, c0, body, R, t)[0].clone()
    out["lower.upper_in"] = up_l.numpy()
    out["lower.x_in"] = x_l.numpy()                                      # already transformed once (Q1)
    run("lower", lo, lambda m: m(up_l.clone(), x_l.clone(), h0, c0, h0, c0, body, R, t)[0], target[:, :, lower_map])
    save("g6_train.npz", **out)


def g7_imu():
    out = {}
    g = torch.Generator().manual_seed(71)
    imu = torch.randn(2, 3, 20, 15, generator=g)
    out["imu"] = imu.numpy()
    torch.manual_seed(701)
    small = IMUNet(15, 9, 32, 2, True, 0.1).eval()
    for k, v in sd_np(small, "small.w.").items():
        out[k] = v
    with torch.no_grad():
        R, t = small(imu)
    out["small.R"], out["small.t"] = R.numpy(), t.numpy()
    # train-mode (dropout 0) stage-1 loss and grads for the small instance
    torch.manual_seed(702)
    tr = IMUNet(15, 9, 32, 2, True, 0).train()
    for k, v in sd_np(tr, "train.w.").items():
        out[k] = v
    Rg = torch.linalg.qr(torch.randn(2, 3, 3, 3, generator=g))[0].contiguous()
    hg = torch.randn(2, 3, 3, generator=g)
    R, t = tr(imu)
    m = torch.bmm(R.view(-1, 3, 3), Rg.view(-1, 3, 3).transpose(1, 2))
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - 1) / 2
    loss = torch.sum(torch.acos(torch.clamp(cos, -1 + 1e-7, 1 - 1e-7))) / 3.14159265358 * 180 \
        + 100 * torch.sum(torch.sqrt(torch.sum(torch.square(t - hg), dim=-1)))
    loss.backward()
    out["train.R_gt"], out["train.head_gt"], out["train.loss"] = Rg.numpy(), hg.numpy(), np.asarray(loss.item())
    for k, p in tr.named_parameters():
        out["train.grad." + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    # full-size instance: seed + outputs + checksums only (weights are 92 MB)
    torch.manual_seed(703)
    big = IMUNet(15, 9, 512, 2, True, 0.1).eval()
    with torch.no_grad():
        R, t = big(imu)
    out["big.R"], out["big.t"] = R.numpy(), t.numpy()
    for k in ("fc1.weight", "rnn_fast.weight_hh_l1_reverse", "rnn_slow.weight_ih_l0"):
        v = big.state_dict()[k].double()
        out["big.chk." + k] = np.asarray([v.sum().item(), v.norm().item()])
    save("g7_imu.npz", **out)


def g8_metric():
    """Pin the metric restatement against Demo_test's formulas, evaluated here line by line."""
    g = torch.Generator().manual_seed(81)
    pred = torch.randn(1, 20, 21, 3, generator=g)
    target = pred + 0.05 * torch.randn(1, 20, 21, 3, generator=g)
    sk_all = torch.tensor(Config.skeleton_all)
    root, leaf = sk_all[:, 0], sk_all[:, 1]
    pv = pred[:, :, [l for l in leaf], :] - pred[:, :, [l for l in root], :]
    tv = target[:, :, [l for l in leaf], :] - target[:, :, [l for l in root], :]
    cs = torch.nn.functional.cosine_similarity(pv, tv, dim=-1)
    ang = torch.abs(torch.acos(torch.clamp(cs, min=-1.0, max=1.0)) / 3.14159265358 * 180.0)
    accu_a = torch.sqrt(torch.sum(torch.square(pred - target), dim=-1))
    up_map, lo_map = Config.upper_joint_map, Config.lower_joint_map
    save("g8_metric.npz", pred=pred.numpy(), target=target.numpy(),
         accu=np.asarray(torch.mean(accu_a).item()),
         accu_l=np.asarray(torch.mean(accu_a, dim=0).mean(dim=0).numpy()),
         angle_l=np.asarray(torch.mean(ang, dim=0).mean(dim=0).numpy()),
         upper=np.asarray(torch.mean(accu_a[:, :, up_map]).item()),
         lower=np.asarray(torch.mean(accu_a[:, :, lo_map]).item()))


if __name__ == "__main__":
    which = set(sys.argv[1:])
    real = None
    if not which or which & {"g2", "g4", "g6", "real"}:
        if os.path.exists(os.path.join(OUT, "real16.npz")) and "real" not in which and which:
            real = dict(np.load(os.path.join(OUT, "real16.npz")))
        else:
            real, _ = load_real()
    if not which or "g1" in which:
        g1_transforms()
    if not which or "g2" in which:
        g2_grouping(real["x"])
    if not which or "g3" in which:
        g3_adjacency()
    if not which or "g4" in which:
        g4_g5_g9(real)
    if not which or "g6" in which:
        g6_train(real)
    if not which or "g7" in which:
        g7_imu()
    if not which or "g8" in which:
        g8_metric()
