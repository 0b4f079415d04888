"""What one more kernel node costs in a replayed HIP graph: N dependent one-workgroup kernels (mmego_fill of 64 floats) in one
captured stream, and two such chains side by side on two streams; us per node."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
bufs = [torch.zeros(64, device=dev) for _ in range(2)]
side = torch.cuda.Stream()


def chain(n, buf):
    for _ in range(n):
        hip.call("fill", buf, 64, 0.0)


def timed(body, n_nodes, label):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            body()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("%s: %.1f us per replay = %.2f us per node" % (label, us, us / n_nodes))


N = 400
timed(lambda: chain(N, bufs[0]), N, "%d dependent one-workgroup kernels on one stream" % N)


def two():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(N, bufs[1])
    chain(N, bufs[0])
    cur.wait_stream(side)


timed(two, N, "two such chains side by side (per node of ONE chain)")

# one LONG kernel (a 10240 x 2048 x 1024 fp32 product, ~350 us) on one stream, the chain of tiny kernels on the other
from mmego_amd import ops  # noqa: E402
A = torch.randn(10240, 1024, device=dev)
W = torch.randn(2048, 1024, device=dev)
C = torch.empty(10240, 2048, device=dev)


def long_only():
    ops.linear(A, W, None, C)


def long_and_chain():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(200, bufs[1])
    ops.linear(A, W, None, C)
    cur.wait_stream(side)


timed(long_only, 1, "one long product alone")
timed(long_and_chain, 1, "the long product with a chain of 200 tiny kernels beside it (chain alone: 200 x 1.55 = 310 us)")

# a chain of MEDIUM kernels (256-workgroup fills of 4 MB: ~5 us each) beside the chain of tiny ones
big = torch.zeros(1 << 20, device=dev)


def medium_chain(n):
    for _ in range(n):
        hip.call("fill", big, big.numel(), 0.0)


timed(lambda: medium_chain(200), 200, "200 dependent 4-MB fills on one stream")


def medium_and_tiny():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        chain(200, bufs[1])
    medium_chain(200)
    cur.wait_stream(side)


timed(medium_and_tiny, 200, "the same with 200 tiny kernels on a second stream (per node of ONE chain)")

# r03: the same two chains as TWO GRAPHS (each a plain linear chain) replayed on two streams, instead of two branches of one graph
def capture(body):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            body()
    torch.cuda.synchronize()
    return g


ga, gb = capture(lambda: chain(N, bufs[0])), capture(lambda: chain(N, bufs[1]))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def replay_pair():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ga.replay()
    with torch.cuda.stream(sb):
        gb.replay()
    cur.wait_stream(sa); cur.wait_stream(sb)


for _ in range(3):
    replay_pair()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    replay_pair()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("two chains of %d as TWO graphs on two streams: %.1f us per pair of replays = %.2f us per node of ONE chain" % (N, us, us / N))
# and the medium chain (256-workgroup kernels) beside the tiny chain, as two graphs
gm, gt = capture(lambda: medium_chain(200)), capture(lambda: chain(200, bufs[1]))


def replay_pair2():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        gm.replay()
    with torch.cuda.stream(sb):
        gt.replay()
    cur.wait_stream(sa); cur.wait_stream(sb)


for _ in range(3):
    replay_pair2()
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    replay_pair2()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print("200 4-MB fills and 200 tiny kernels as TWO graphs on two streams: %.1f us per pair = %.2f us per node of ONE chain" % (us, us / 200))
