"""Micro-benchmark of mmego_lstm_step (Bn x H, both directions): us per launch and TFLOP/s.
`first` carries diagnostic bits in this script only: 2 = no cell update math, 4 = no global tile loads, 8 = no MFMA."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmego_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
Bn, H, T = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 512, 20
modes = [int(a) for a in sys.argv[2:] if not a.startswith("--")] or [0]
use_graph = "--graph" in sys.argv      # 20 dependent launches per replayed HIP graph: no host launch cost in the figure
torch.manual_seed(0)
out = torch.randn(Bn, T, 2 * H, device=dev) * 0.1
xp = torch.randn(Bn, T, 8 * H, device=dev) * 0.1
w = [torch.randn(4 * H, H, device=dev) * 0.04 for _ in range(2)]
b = [torch.randn(4 * H, device=dev) * 0.04 for _ in range(2)]
c = torch.zeros(2, Bn, H, device=dev)
xs, os_ = T * 8 * H, T * 2 * H


def launch(mode, s=1):
    t0, t1 = s, T - 1 - s
    hip.call("lstm_step", 2, Bn, H, mode, out.data_ptr() + 4 * ((t0 - 1) * 2 * H), out.data_ptr() + 4 * ((t1 + 1) * 2 * H + H), os_,
             w[0], w[1], b[0], b[1], xp.data_ptr() + 4 * (t0 * 8 * H), xp.data_ptr() + 4 * (t1 * 8 * H + 4 * H), xs,
             out.data_ptr() + 4 * (t0 * 2 * H), out.data_ptr() + 4 * (t1 * 2 * H + H), os_, c[0], c[1], None, None, None, None)


for mode in modes:
    for _ in range(5):
        launch(mode)
    torch.cuda.synchronize()
    if use_graph:
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            launch(mode)
            with torch.cuda.graph(g, stream=st):
                for i in range(20):
                    launch(mode, 1 + (i % (T - 2)))
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 400 * 1e3
        print("Bn=%d mode=%d (graph): %.1f us/launch, %.1f TFLOP/s" % (Bn, mode, us, 2 * 2 * Bn * 4 * H * H / us / 1e6))
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record()
    for i in range(n):
        launch(mode, 1 + (i % (T - 2)))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("Bn=%d mode=%d: %.1f us/launch, %.1f TFLOP/s" % (Bn, mode, us, 2 * 2 * Bn * 4 * H * H / us / 1e6))


if "--chains" in sys.argv:
    # the whole T-step recurrence of one BiLSTM layer: both directions per launch against two chains of single-direction launches
    from mmego_amd import blocks, ops
    lstm = blocks.LstmParams(H, H, 1).to(dev)
    xp2 = torch.randn(Bn * T, 8 * H, device=dev) * 0.1
    for two in (False, True):
        blocks._LSTM_TWO_CHAINS = two
        ar = ops.Arena(dev)
        o2 = ar.get("out", (Bn * T, 2 * H))
        run = lambda: blocks.lstm_recurrence(ar, "k", lstm, 0, xp2, o2, Bn, T)
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            run()
            with ops.capture(g, stream=st):
                run()
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
        fl = 2.0 * 2 * Bn * 4 * H * H * (T - 1)
        print("layer recurrence Bn=%d T=%d, %s: %.1f us per layer = %.2f us per timestep, %.1f TFLOP/s"
              % (Bn, T, "2 chains x %d single-direction launches" % T if two else "%d launches (both directions each)" % T, us, us / T,
                 fl / us / 1e6))


if "--pair" in sys.argv:
    # two INDEPENDENT BiLSTM layers (the two stages' frozen IMU_Net instances): one after the other against side by side on two
    # streams, every launch carrying both directions of its instance (HT = 32, two workgroups of different launches per CU)
    from mmego_amd import blocks, ops
    inst = [blocks.LstmParams(H, H, 1).to(dev) for _ in range(2)]
    xps = [torch.randn(Bn * T, 8 * H, device=dev) * 0.1 for _ in range(2)]
    ars = [ops.Arena(dev) for _ in range(2)]
    outs = [a.get("out", (Bn * T, 2 * H)) for a in ars]
    side = torch.cuda.Stream()

    def one(i):
        blocks.lstm_recurrence(ars[i], "k", inst[i], 0, xps[i], outs[i], Bn, T)

    def serial(two_first):
        with blocks.two_chains(two_first):
            one(0)
        with blocks.two_chains(False):
            one(1)

    def beside():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with blocks.two_chains(False):
            with torch.cuda.stream(side):
                one(1)
            one(0)
        cur.wait_stream(side)

    for name, run in (("one after the other, both one launch per timestep", lambda: serial(False)),
                      ("one after the other, the first as two chains", lambda: serial(True)),
                      ("side by side on two streams, one launch per timestep each", beside)):
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            with ops.capture(g, stream=st):
                run()
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
        print("two instances' layer recurrence Bn=%d T=%d, %s: %.1f us = %.2f us per timestep of both" % (Bn, T, name, us, us / T))
