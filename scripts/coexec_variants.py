"""r06: which instruction class of head_fk_loss_kernel<1> miscomputes beside the bf16-MFMA step kernel.

Victim = mmego_head_fk_loss(which = 1) from VARIANT builds of csrc/geom.hip alone (mmego_amd/lib/variants/libgeom_<name>.so, built by
scripts/build_hfk_variants.sh: every one without the 144-KB LDS request), aggressor = the 16-unit split3 step kernel from the
product library, two workgroups per CU on a second stream (scripts/coexec_head_fk.py's densest arrangement).  One pass per variant:
reference outputs on an idle GPU, then ROUNDS rounds of (aggressor stack on stream B, 20 victim launches on the current stream),
counting the rounds whose outputs differ.  The first differing dy of each variant goes to gpurun_out/coexec_variants.npz with the
inputs, for the offline look at WHICH values are wrong (scripts/coexec_analyse.py)."""
import ctypes
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmego_amd import blocks, hip, ops  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 600
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
Bn, S, H = 512, 20, 512
lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
xs = torch.randn(Bn * S, H, device=dev).relu_()
sB = torch.cuda.Stream()
_s16 = {}


def step16_stack():
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


with torch.cuda.stream(sB), torch.no_grad():
    step16_stack()
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
args = hip._protos["mmego_head_fk_loss"]
dump = {"y": y.cpu().numpy(), "body": body.cpu().numpy(), "R": R.cpu().numpy(), "t": t.cpu().numpy(), "target": target.cpu().numpy()}

libs = sorted(glob.glob(os.path.join(ROOT, "mmego_amd", "lib", "variants", "libgeom_*.so")))
for path in libs:
    name = os.path.basename(path)[len("libgeom_"):-3]
    if only and name not in only:
        continue
    fn = ctypes.CDLL(path).mmego_head_fk_loss
    fn.restype, fn.argtypes = ctypes.c_int, [a for a, _ in args]

    def victim():
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(20):
            rc = fn(st, 1, y.data_ptr(), body.data_ptr(), B, F, q.data_ptr(), jh.data_ptr(), R.data_ptr(), t.data_ptr(), l.data_ptr(), None, 0, None,
                    target.data_ptr(), jmap.data_ptr(), 21, 1.0, loss2.data_ptr(), dy.data_ptr(), scr.data_ptr())
            assert rc == 0, rc

    for o in outs:
        o.zero_()
    victim()
    torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    bad, nsaved = 0, 0
    badcols, badlanes = {}, {}
    for it in range(rounds):
        with torch.cuda.stream(sB), torch.no_grad():
            step16_stack()
        victim()
        torch.cuda.synchronize()
        d = [(a - b).abs().max().item() for a, b in zip(outs, ref)]
        if max(d) > 0 or any(torch.isnan(a).any().item() for a in outs):
            bad += 1
            nz = (dy != ref[3]).nonzero()
            for r_, c_ in nz.tolist():
                badcols[c_] = badcols.get(c_, 0) + 1
                badlanes[r_ % 64 // 16] = badlanes.get(r_ % 64 // 16, 0) + 1
            if nsaved < 4:
                dump["%s_bad%d" % (name, nsaved)] = dy.cpu().numpy()
                dump["%s_ref" % name] = ref[3].cpu().numpy()
                nsaved += 1
            if bad == 1:
                print("   first difference: max |diff| of (q, joints, world, dy, loss) = %s" % (["%.3g" % v for v in d],))
    print("variant %-12s: %3d of %d rounds differ; differing dy columns %s; 16-lane group of the wave %s" % (name, bad, rounds, dict(sorted(badcols.items())), dict(sorted(badlanes.items()))), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "coexec_variants.npz"), **dump)
