"""r06: scripts/coexec_pk_probe.hip's pattern kernels beside the bf16-MFMA step kernel (see that file).  Per mode: reference on an idle
GPU, then ROUNDS rounds of (aggressor stack on stream B, 20 probe launches on the current stream); rounds whose output differs."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmego_amd import blocks, hip  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 24
# what runs on the second stream (argv[5], comma-separated; default the 16-unit split3 stack): none | step16 (s3_gemm + 16-unit steps,
# two workgroups per CU) | step32 (the product's split3 stack: s3_gemm + 32-unit steps) | s3gemm (the projection products alone) |
# s3gemm4 / s3gemm10 (the same with the 256 x 128 register-staged / the 320 x 256 LDS-DMA kernel forced) | s3steps (the 16-unit steps alone) | fp32 (the fp32 stack: gemm_tile + LDS-DMA steps) | bf16 (the bf16-mode stack of section 7a)
aggressors = sys.argv[5].split(",") if len(sys.argv) > 5 else ["step16"]
from mmego_amd import ops  # noqa: E402
ar = ops.Arena(dev)
vic = ctypes.CDLL(os.path.join(ROOT, "mmego_amd", "lib", "variants", "libpkprobe.so"))
vic.pk_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
Bn, S, H = 512, 20, 512
lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
xs = torch.randn(Bn * S, H, device=dev).relu_()
sB = torch.cuda.Stream()
_s16 = {}


def step16_stack(gemm=True, steps=True, wm=0):
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
        if gemm:
            hip.call("split3_gemm", cur, wih, d["xpf"], None, 0, bias, S * nrb, 8 * H // 32, K, 0, 6, wm)
        o_p, out_p = d["O"][layer].data_ptr(), d["out"].data_ptr()
        win = lambda tt, dd: o_p + 2 * ((tt * nrb * S2 + dd * (H // 16)) * 3 * 512)
        ho = lambda tt, dd: out_p + 4 * (tt * 2 * H + dd * H) if layer == 1 else None
        for s_ in range(S if steps else 0):
            t0, t1 = s_, S - 1 - s_
            hip.call("split3_step16", 2, Bn, H, int(s_ == 0), win(t0 - 1, 0) if s_ else None, win(t1 + 1, 1) if s_ else None, S2 * 3,
                     whh0, whh1, d["xpf"], t0 * nrb, t1 * nrb, ho(t0, 0), ho(t1, 1), S * 2 * H, win(t0, 0), win(t1, 1), S2 * 3,
                     d["c"][0], d["c"][1], 6, 0)
        cur, K = d["O"][layer], 2 * H


def aggressor(kind):
    with torch.no_grad(), blocks.two_chains(False):
        if kind == "step16":
            step16_stack()
        elif kind in ("s3gemm", "s3gemm4", "s3gemm10"):      # the library's choice (r06: the 320 x 256 LDS-DMA kernel) / 256 x 128 / 320 x 256
            for _ in range(3):
                step16_stack(steps=False, wm={"s3gemm": 0, "s3gemm4": 4, "s3gemm10": 10}[kind])
        elif kind == "s3steps":
            step16_stack(gemm=False)
        elif kind == "step32":
            blocks.lstm_steps_forward_split3(ar, "t3", lstm, xs, Bn, S, nprod=6)
        elif kind == "fp32":
            blocks.lstm_steps_forward(ar, "t", lstm, xs, Bn, S)
        elif kind == "bf16":
            blocks.lstm_steps_forward_bf16(ar, "tb", lstm, xs, Bn, S)
        elif kind != "none":
            raise ValueError(kind)


for kind in aggressors:
    with torch.cuda.stream(sB):
        aggressor(kind)
torch.cuda.synchronize()
NAMES = {0: "mov(hi) -> pk_mul", 1: "mov(lo) -> pk_mul", 2: "mov(hi), s_nop 1, pk_mul", 3: "pk_mul -> add(hi)", 4: "pk ops, no short dependency",
         5: "control: mov -> mul", 6: "cmp, cndmask(hi) -> pk_mul", 7: "mov -> pk_mov op_sel -> pk_add",
         8: "cmp, pk_add, mov, cndmask", 9: "cmp, mov, mov, cndmask (control)", 10: "cmp, pk_add, pk_mul, cndmask", 11: "cmp, pk_mul op_sel, mov, cndmask",
         12: "cmp, pk_add, cndmask (1 state)", 13: "cmp, mov, cndmask (1 state)", 14: "cmp, s_nop 1, cndmask (control)", 15: "cmp, mov, pk_mul, cndmask", 16: "cmp, pk_mul, s_nop 3, mov, cndmask", 17: "cmp, mul, s_nop 3, mov, cndmask (ctl)",
         18: "cmp, s_nop 3, pk_mul, mov, cndmask", 19: "pk_mul, cmp, s_nop 3, mov, cndmask", 20: "pk, nop, cmp, nop, pk, nop, mov, cndmask", 21: "pk_add op_sel:[0,1] op_sel_hi:[1,0]", 22: "pk_mul op_sel:[0,1]",
         23: "pk_mul op_sel_hi:[1,0]", 24: "pk_add, no op_sel (control)", 25: "pk_fma op_sel:[0,1,0]", 26: "pk_mov op_sel:[1,0]", 27: "pk_add op_sel:[1,0]"}
modes = [int(m) for m in sys.argv[4].split(",")] if len(sys.argv) > 4 else list(range(28))
out = torch.zeros(nblk * 64 * 4, device=dev)
for mode in modes:
    def victim():
        for _ in range(20):
            rc = vic.pk_probe_launch(torch.cuda.current_stream().cuda_stream, mode, out.data_ptr(), nblk, iters)
            assert rc == 0, rc
    victim()
    torch.cuda.synchronize()
    ref = out.clone()
    assert torch.isfinite(ref).all()
    for kind in aggressors:
        bad, lanes, cols = 0, {}, {}
        for it in range(rounds):
            with torch.cuda.stream(sB):
                aggressor(kind)
            victim()
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                idx = (out != ref).nonzero().view(-1)
                for i in idx.tolist():
                    lanes[(i // 4) % 64 // 16] = lanes.get((i // 4) % 64 // 16, 0) + 1
                    cols[i % 4] = cols.get(i % 4, 0) + 1
        print("mode %d  %-36s beside %-7s: %3d of %d rounds differ; 16-lane groups %s; outputs (0 = low half sum, 1 = high half sum) %s" % (
            mode, NAMES[mode], kind, bad, rounds, dict(sorted(lanes.items())), dict(sorted(cols.items()))), flush=True)
