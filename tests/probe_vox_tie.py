"""Not a test (run by hand on the GPU box: python tests/probe_vox_tie.py).  UpperNetwlocal, the three synced-state training steps of
test_hip_local.test_train_upper_wlocal_from_synced_states: is a step that misses the gradient bar a defect or a ReLU near-tie?
Per step, from the same state: gradients of the fp32 oracle / the HIP net with the generic voxel launches / the HIP net with vox.hip
against a FLOAT64 run of the oracle, the voxel chain's intermediates of the two HIP forms against each other, and the ReLU mask
elements that differ between them with their pre-activations.  r05: step 2 has bn3(z3) = -1.9e-6 at one element; steps 1 and 3 have
no pre-activation below 1.9e-5 and both HIP forms are 2-3e-7 from float64 there (the fp32 oracle: 1e-4 / 5e-5 / 3e-7)."""
import copy, os, sys
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))
import torch
import test_hip_parity as tp
from conftest import golden
from oracle import nets as on, train as ot, skeleton as sk
import mmego_amd.nets_local as nl
from mmego_amd.params import FusedAdam
dev = torch.device("cuda:0")
T = tp.T
from oracle import geometry as geo
_ag = geo.anchor_grid
g = golden("g6_train.npz")
x0, body, R, t, target = [T(g[k]) for k in ("x", "body", "R", "t", "target")]
target = target[:, :, list(sk.UPPER_MAP)]
h0, c0 = ot.zeros_state(4)
o, h = tp._train_pair("wlocal", 602, on.UpperNetwlocal, nl.UpperNetwlocal, dev)
d = lambda v: v.to(dev)
opt_o = torch.optim.Adam(o.parameters(), lr=3e-5)
for step in (1, 2, 3):
    # fp64 truth from the same state
    o64 = copy.deepcopy(o).double()
    for p in o64.parameters(): p.grad = None
    geo.anchor_grid = lambda: _ag().double()
    l64 = o64(x0.clone().double(), h0.double(), c0.double(), h0.double(), c0.double(), body.double(), R.double(), t.double())[0]
    (l64 - target.double()).abs().sum().backward()
    geo.anchor_grid = _ag
    g64 = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in o64.named_parameters()}
    scale = max(v.abs().max().item() for v in g64.values())
    opt_o.zero_grad()
    lo_ = o(x0.clone(), h0, c0, h0, c0, body, R, t)[0]
    ot.l1_sum(lo_, target).backward()
    res = {"oracle32": {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in o.named_parameters()}}
    sd = {k: v.clone() for k, v in h.state_dict().items()}
    for name, fused in (("hip chain", False), ("hip vox", True)):
        nl._VOX_FUSED = fused
        h.load_state_dict(sd)
        for p in h.parameters(): p.grad = None
        h.flat().flat_g.zero_()
        lh = h(d(x0.clone()), d(h0), d(c0), d(h0), d(c0), d(body), d(R), d(t))[0]
        (lh - target.to(dev)).abs().sum().backward()
        res[name] = {k: p.grad.detach().cpu().clone() for k, p in h.named_parameters()}
        ar = h.arena("train")
        snap = {}
        for key, shape in (("vx.z1", (32, 96)), ("vx.z2", (32, 128)), ("vx.z3", (32, 64)), ("vx.y1", (32, 96)), ("vx.y2", (32, 128)), ("vvec", (32, 64)),
                           ("vx.dz1", (32, 96)), ("vx.dz2", (32, 128)), ("vx.dz3", (32, 64)), ("arnn.dx0", (32, 64)), ("voxT", (32, 1728)),
                           ("vx.bn1", (4, 96)), ("vx.bn2", (4, 128)), ("vx.bn3", (4, 64))):
            snap[key] = ar.get(key, shape).detach().cpu().clone()
        snap["dx"] = ar.get("vx.dx" if fused else "vx.dy0", (32, 1728)).detach().cpu().clone()
        res[name + ".snap"] = snap
        if fused:
            g3 = ar.get("vx.g3", (32, 64)).cpu(); prt3 = ar.get("vx.prt3", (2, 64, 2)).cpu()
            dv, vv, z3, st3 = snap["arnn.dx0"], snap["vvec"], snap["vx.z3"], snap["vx.bn3"]
            g_ref = dv * (vv > 0).float()
            xhat = (z3 - st3[0]) * st3[1]
            s1, s2 = g_ref.sum(0), (g_ref * xhat).sum(0)
            print("   g3 err %.2e  S1 err %.2e (max %.2e)  S2 err %.2e (max %.2e)" % ((g3 - g_ref).abs().max(), (prt3[:, :, 0].sum(0) - s1).abs().max(), s1.abs().max(),
                  (prt3[:, :, 1].sum(0) - s2).abs().max(), s2.abs().max()))
            dz_ref = st3[2] * (g_ref - s1 / 32 - xhat * s2 / 32)
            print("   dz3 vs formula: vox %.2e" % (snap["vx.dz3"] - dz_ref).abs().max(), " gamma grad (flat) vs S2: %.2e" % (h.module2.avoxel.cb3.weight.grad.cpu() - s2).abs().max(),
                  " n(vvec == 0 & dv != 0) =", int(((vv == 0) & (dv != 0)).sum()), " n(vv<0)=", int((vv < 0).sum()))
    sa, sb = res.pop("hip chain.snap"), res.pop("hip vox.snap")
    print("step %d chain vs vox: " % step + ", ".join("%s %.1e/%.1e" % (k, (sa[k] - sb[k]).abs().max().item(), sa[k].abs().max().item()) for k in sa))
    for key, zk, bk in (("vvec", "vx.z3", "vx.bn3"), ("vx.y2", "vx.z2", "vx.bn2"), ("vx.y1", "vx.z1", "vx.bn1")):
        flip = (sa[key] > 0) != (sb[key] > 0)
        pre = (sa[zk] - sa[bk][0]) * sa[bk][2] + sa[bk][3]
        print("   %s: %d ReLU mask elements differ between the runs; their pre-activations (chain run): %s; smallest |pre-activation| overall %.2e"
              % (key, int(flip.sum()), pre[flip].tolist(), pre.abs().min().item()))
    bad = (sa["dx"] - sb["dx"]).abs()
    if bad.max() > 1e-4 * sa["dx"].abs().max():
        rows_bad = (bad.max(1).values > 1e-4 * sa["dx"].abs().max()).nonzero().flatten().tolist()
        print("   dx rows off:", rows_bad, "dz1 rows off:", ((sa["vx.dz1"] - sb["vx.dz1"]).abs().max(1).values > 1e-5 * sa["vx.dz1"].abs().max()).nonzero().flatten().tolist())
    for name, gr in res.items():
        worst = max(((gr[k].double() - g64[k]).abs().max().item() / scale, k) for k in g64 if not tp.NOISE_GRAD.search(k))
        vox = max(((gr[k].double() - g64[k]).abs().max().item() / scale, k) for k in g64 if "avoxel" in k and not tp.NOISE_GRAD.search(k))
        print("step %d %-10s worst |g - g64| / scale = %.2e (%s); avoxel params: %.2e (%s)" % (step, name, worst[0], worst[1], vox[0], vox[1]))
    for k, p in o.named_parameters():
        if tp.NOISE_GRAD.search(k) and p.grad is not None: p.grad.zero_()
    opt_o.step()
    h.load_state_dict({k: v.to(dev) for k, v in o.state_dict().items()})
