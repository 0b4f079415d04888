"""mmego_upper_front_eval alone (eval-mode Upper_Net front end, front.hip; "bf16": front_bf16.hip's mmego_upper_front_eval_bf16) over
frame counts: us per launch in a replayed graph and TFLOP/s on the algorithmic 2 x 6064 flop per point; a checksum of the pooled features
for a cross-build comparison.  usage: python scripts/bench_front.py [bf16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import nets, hip
dev = torch.device("cuda:0")
entry = "upper_front_eval_bf16" if "bf16" in sys.argv[1:] else "upper_front_eval"
torch.manual_seed(0)
net = nets.UpperNet().to(dev).eval()
for F, N in ((128, 128), (256, 128), (512, 128), (1024, 128), (2048, 128), (8192, 128), (32768, 256)):
    x = torch.randn(F, 1, N, 6, device=dev)
    R = torch.eye(3, device=dev).repeat(F, 1, 1, 1).contiguous(); t = torch.zeros(F, 1, 3, device=dev)
    vec = torch.empty(F, 64, device=dev); attn = torch.empty(F, N, device=dev)
    tab = net._front_table()
    def run():
        hip.call(entry, x, None, R, t, F, N, tab, 1e-5, vec, attn)
    for _ in range(3): run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): run()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2.0 * 6064 * F * N
    print("%s F=%6d N=%d: %.1f us per launch, %.1f TFLOP/s (algorithmic); sum |vec| %.6f" % (entry, F, N, us, fl / us / 1e6, float(vec.double().abs().sum())))
