"""r05's standalone reproducer of the co-residency finding (kept as written; r06's tools supersede it: scripts/coexec_variants.py --
victim variants --, coexec_asm_patch.py -- the real kernel's assembly patched --, coexec_fullstep.py -- the whole step --).
mmego_head_fk_loss (which = 1) launched on one stream while a BiLSTM(512) stack runs on another -- as the fp32 step kernels, as the split3
32-unit step kernel, and as the split3 16-unit step kernel in its two-chain form.  With the PRODUCT library it reports 0 differing rounds
everywhere since r06 (no packed-fp32 instruction in the victim); r05's statement below the table "never beside the others" was wrong for
the whole step (22 of 60 engines in the 32-unit arrangement)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, ops  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Bn, S, H = 512, 20, 512
lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
xs = torch.randn(Bn * S, H, device=dev).relu_()
ar = ops.Arena(dev)
sB = torch.cuda.Stream()


_s16 = {}


def step16_stack():
    """The two-layer stack of blocks.lstm_steps_forward_split3 with every recurrence on the 16-unit kernel, both directions per launch
    (512 workgroups, two per CU): the densest way to put its workgroups beside the victim's (the two-chain form reproduces the same
    at a lower rate: ~10 % of ConcurrentStages steps)."""
    if not _s16:
        nrb, S2 = Bn // 32, 2 * H // 16
        _s16.update(W=blocks.lstm_split3_weights(lstm, 16), x=blocks.split3_cvt(xs, tm=(Bn, S, Bn)), xpf=torch.empty(S * Bn * 8 * H, device=dev),
                    O=[blocks.split3_cvt(torch.zeros(S * Bn, 2 * H, device=dev)) for _ in range(2)], out=torch.empty(Bn * S, 2 * H, device=dev),
                    c=torch.zeros(2, Bn, H, device=dev), nrb=nrb, S2=S2)
    d = _s16
    nrb, S2 = d["nrb"], d["S2"]
    cur, K = d["x"], H
    for layer in range(2):
        wih, bias, whh0, whh1 = d["W"][layer]
        hip.call("split3_gemm", cur, wih, d["xpf"], None, 0, bias, S * nrb, 8 * H // 32, K, 0, 6, 0)
        o_p, out_p = d["O"][layer].data_ptr(), d["out"].data_ptr()
        win = lambda tt, dd: o_p + 2 * ((tt * nrb * S2 + dd * (H // 16)) * 3 * 512)
        ho = lambda tt, dd: out_p + 4 * (tt * 2 * H + dd * H) if layer == 1 else None
        for s_ in range(S):
            t0, t1 = s_, S - 1 - s_
            hip.call("split3_step16", 2, Bn, H, int(s_ == 0), win(t0 - 1, 0) if s_ else None, win(t1 + 1, 1) if s_ else None, S2 * 3,
                     whh0, whh1, d["xpf"], t0 * nrb, t1 * nrb, ho(t0, 0), ho(t1, 1), S * 2 * H, win(t0, 0), win(t1, 1), S2 * 3,
                     d["c"][0], d["c"][1], 6, 0)
        cur, K = d["O"][layer], 2 * H


def stress(kind):
    with torch.no_grad(), blocks.two_chains(False):
        if kind == "fp32":
            blocks.lstm_steps_forward(ar, "t", lstm, xs, Bn, S)
        elif kind == "step32":
            blocks.lstm_steps_forward_split3(ar, "t", lstm, xs, Bn, S, nprod=6)
        elif kind == "step16":
            step16_stack()


for kind in ("step16", "step32", "fp32"):
    with torch.cuda.stream(sB):
        stress(kind)
torch.cuda.synchronize()
B, F = 64, 512
g = torch.Generator().manual_seed(3)
R = torch.linalg.qr(torch.randn(F, 3, 3, generator=g))[0].contiguous().to(dev)
t = (torch.randn(F, 3, generator=g) * 0.1).to(dev)
y = torch.randn(F, 42, generator=g).to(dev)
body = (torch.randn(B, 20, 3, generator=g) * 0.2).to(dev)
target = torch.randn(F, 21, 3, generator=g).to(dev)
jmap = torch.tensor([12, 13, 14, 15, 16, 17, 18, 19], dtype=torch.int32, device=dev)
q = torch.zeros(F, 6, 3, 3, device=dev)
jh, l = torch.zeros(F, 8, 3, device=dev), torch.zeros(F, 8, 3, device=dev)
loss2, dy, scr = torch.zeros(2, device=dev), torch.zeros(F, 42, device=dev), torch.zeros(17, dtype=torch.float64, device=dev)
outs = (q, jh, l, dy, loss2)


def victim():
    for _ in range(20):
        hip.call("head_fk_loss", 1, y, body, B, F, q, jh, R, t, l, None, 0, None, target, jmap, 21, 1.0, loss2, dy, scr)


victim()
torch.cuda.synchronize()
ref = [o.clone() for o in outs]
for kind in ("none", "fp32", "step32", "step16"):
    bad = 0
    for it in range(rounds):
        with torch.cuda.stream(sB):
            stress(kind)
        victim()
        torch.cuda.synchronize()
        d = [(a - b).abs().max().item() for a, b in zip(outs, ref)]
        if max(d) > 0:
            bad += 1
            if bad == 1:
                rows = sorted(set((dy != ref[3]).nonzero()[:, 0].tolist()))
                print("   first difference: max |diff| of (q, joints, world, dy, loss) = %s; dy rows %s" % (["%.3g" % v for v in d], rows))
    print("head_fk_loss<1> beside %-7s: %2d of %d rounds differ" % (kind, bad, rounds), flush=True)
