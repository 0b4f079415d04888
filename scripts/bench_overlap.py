"""Can one IMU_Net's input-projection product run BESIDE another IMU_Net's recurrence?  (replayed HIP graphs, MI355X)

A U+L step holds two independent frozen IMU_Net forwards (reference Net/IMU_Net.py:58-62,77,82 inside Train_Upper.py / Train_Lower.py's
bodies), each a strict alternation of a throughput product (input projection, 0.77 of the fp32 MFMA peak) and a latency chain
(20 recurrent steps, 0.55).  If net B's product overlaps net A's recurrence the matrix pipe sees both instruction streams.
What stands in the way is residency: the persistent projection kernel runs 2 workgroups per CU at 204 VGPRs, which leaves no
registers for a recurrent-step wave (120 VGPRs) -- the step's 256 workgroups then queue behind 20-us tiles.  MMEGO_GEMM_SLOTS=256
runs the product with ONE workgroup per CU.

  python scripts/bench_overlap.py            (run once per MMEGO_GEMM_SLOTS setting: the knob is read once per process)
prints, per K in {512, 1024}: product alone, recurrence alone, one after the other, side by side (two streams of one graph).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, ops  # noqa: E402

dev = torch.device("cuda:0")
hip.lib()
Bn, H, T = 512, 512, 20
torch.manual_seed(0)
lstm = blocks.LstmParams(H, H, 1).to(dev)
xproj = torch.randn(Bn * T, 8 * H, device=dev) * 0.1
ar = ops.Arena(dev)
out = ar.get("out", (Bn * T, 2 * H))
side = torch.cuda.Stream()


def rec():
    with blocks.two_chains(False):
        blocks.lstm_recurrence(ar, "k", lstm, 0, xproj, out, Bn, T)


def timed(body, replays=20):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with ops.capture(g, stream=st):
            body()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / replays * 1e3


print("MMEGO_GEMM_SLOTS =", os.environ.get("MMEGO_GEMM_SLOTS", "512 (default)"))
for K in (512, 1024):
    x = torch.randn(Bn * T, K, device=dev) * 0.1
    W = [torch.randn(4 * H, K, device=dev) * 0.03 for _ in range(2)]
    b = [torch.randn(4 * H, device=dev) * 0.03 for _ in range(2)]
    Wb, bb = torch.cat(W), torch.cat(b)
    xp_out = torch.empty(Bn * T, 8 * H, device=dev)

    def gemm():
        ops.linear_pair(x, Wb[:4 * H], Wb[4 * H:], bb[:4 * H], bb[4 * H:], xp_out, 4 * H)

    def seq():
        gemm()
        rec()

    def par():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gemm()
        rec()
        cur.wait_stream(side)
    tg, tr, ts, tp = timed(gemm), timed(rec), timed(seq), timed(par)
    fl = 2.0 * 2 * Bn * T * 4 * H * K + 2.0 * 2 * Bn * 4 * H * H * (T - 1)
    print("K=%4d: product %.0f us, recurrence %.0f us (%.1f per step), one after the other %.0f us, side by side %.0f us "
          "(%.1f TFLOP/s, %.2f of the fp32 MFMA peak)" % (K, tg, tr, tr / T, ts, tp, fl / tp / 1e6, fl / tp / 1e6 / 157.3))
