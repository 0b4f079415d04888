"""mmego_topk_rows2 alone (geom.hip: Lower_Net's selection of the 64 points with the largest x) at the U+L step's shape (512 frames x 128
points) and config 5's (32 768 x 256): us per launch in a replayed graph.  usage: python scripts/bench_topk.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip
dev = torch.device("cuda:0")
hip.lib()
for F, N in ((512, 128), (4096, 256), (32768, 256)):
    x = torch.randn(F, N, 6, generator=torch.Generator().manual_seed(1)).to(dev)
    out = torch.zeros(F * 64, 6, device=dev); idx = torch.zeros(F, 64, dtype=torch.int64, device=dev); both = torch.zeros(F * 64, 128, device=dev)
    plain = "plain" in sys.argv[1:]           # mmego_topk_rows: without the second output (12 bytes into every 512-byte row of another buffer)
    run = (lambda: hip.call("topk_rows", x, F, N, 6, 64, out, idx)) if plain else (lambda: hip.call("topk_rows2", x, F, N, 6, 64, out, idx, both, 128, 3))
    for _ in range(3): run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10): run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("topk_rows2 F=%6d N=%d: %.1f us per launch; sum idx %d" % (F, N, e0.elapsed_time(e1) / 50 * 1e3, int(idx.sum())))
