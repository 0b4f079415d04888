"""r05 finding, standalone: does a bf16-MFMA kernel (mmego_split3_step16: 220 VGPRs, two workgroups per CU) change the RESULTS of an
unrelated, purely arithmetic kernel (scripts/coexec_victim.hip) that runs beside it on another stream?"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmego_amd import blocks, hip, ops
dev = torch.device("cuda:0")
hip.lib()
vic = ctypes.CDLL(os.path.join(ROOT, "scripts", "exp", "libvictim.so"))
Bn, S, H = 512, 20, 512
lstm = blocks.LstmParams(H, H, 2, dropout=0.0, bidirectional=True).to(dev)
xs = torch.randn(Bn * S, H, device=dev).relu_()
ar = ops.Arena(dev)
sB = torch.cuda.Stream()
def stress(kind):
    os.environ["MMEGO_S3_DBG_V"] = "2" if kind == "step16" else ""
    with torch.no_grad(), blocks.two_chains(False):
        if kind == "fp32":
            blocks.lstm_steps_forward(ar, "t", lstm, xs, Bn, S)
        elif kind != "none":
            blocks.lstm_steps_forward_split3(ar, "t", lstm, xs, Bn, S, nprod=6)
for kind in ("step16", "step32", "fp32"):
    with torch.cuda.stream(sB):
        stress(kind)
torch.cuda.synchronize()
nblk = 64
for nreg, mode, iters in ((200, 9, 3000), (200, 9, 300)):
    out = torch.zeros(nblk * 64 * nreg, device=dev)
    def victim():
        rc = vic.victim_launch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), nreg, mode, ctypes.c_void_p(out.data_ptr()), nblk, iters)
        assert rc == 0, rc
    victim(); torch.cuda.synchronize()
    ref = out.clone()
    for kind in ("none", "step16", "fp32"):
        bad = 0
        worst = 0.0
        for it in range(40):
            out.zero_()
            with torch.cuda.stream(sB):
                stress(kind)
            for _ in range(6):
                victim()
            torch.cuda.synchronize()
            d = (out - ref).abs().max().item()
            if d > 0:
                bad += 1
                worst = max(worst, d)
                if bad == 1:
                    idx = (out != ref).nonzero().view(-1)
                    lanes = sorted(set(((idx // nreg) % 64).tolist()))
                    print("    first bad run: %d elements differ, lanes %s, registers %s" % (idx.numel(), lanes[:20], sorted(set((idx % nreg).tolist()))[:12]))
        print("victim NREG=%3d mode=%d beside %-7s: %2d of 40 runs differ (max %.3g)" % (nreg, mode, kind, bad, worst), flush=True)
