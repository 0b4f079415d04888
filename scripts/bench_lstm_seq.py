"""Time mmego_lstm_seq_xcd alone (replayed graph of one BiLSTM(512) stack forward, projections excluded by subtraction) and, for the
row-tiled kernel, with phases switched off through MMEGO_LQ_DBG (timing by elimination; results are wrong with a mask):
  1 no MFMA | 2 no operand DMA | 4 no exchange stores | 8 no vmcnt wait at the hand-over | 16 no arrival polling | 32 no arrival signal
usage: python scripts/bench_lstm_seq.py [Bn] [T]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import blocks, hip, ops

Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
H = 512
torch.manual_seed(0)
lstm = blocks.LstmParams(H, H, 1).to(dev)
ar = ops.Arena(dev)
xp = torch.randn(Bn * T, 8 * H, device=dev)
out = torch.empty(Bn * T, 2 * H, device=dev)


def run(n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    blocks.lstm_recurrence(ar, "b", lstm, 0, xp, out, Bn, T)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        blocks.lstm_recurrence(ar, "b", lstm, 0, xp, out, Bn, T)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for seq in (False, True):
    blocks._LSTM_SEQ_XCD = seq
    print("%s: %.1f us per layer (%d rows x %d steps), %.2f us per timestep" % ("persistent launch" if seq else "step launches", run(), Bn, T, run() / T))
blocks._LSTM_SEQ_XCD = True
for mask in (1, 2, 4, 8, 16, 1 | 2, 1 | 2 | 4, 1 | 2 | 4 | 16 | 32, 63):     # (32 alone: consumers would spin to their bound)
    os.environ["MMEGO_LQ_DBG"] = str(mask)
    print("  MMEGO_LQ_DBG=%2d: %.1f us per layer" % (mask, run()))
    ar.get("b.seqsync", (128,), dtype=torch.int32).zero_()
os.environ.pop("MMEGO_LQ_DBG")
print("errors:", blocks.seq_xcd_errors())
