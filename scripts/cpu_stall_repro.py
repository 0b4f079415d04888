"""r06, the CPU-suite "stall"  reproduced on purpose: the 200-step oracle training loop of
tests/test_oracle_golden.py::test_g10_oracle_trains_like_the_reference with N torch threads beside K busy processes, with the native
backtraces of every thread (tests/stall_probe.py) after 30 s.   python scripts/cpu_stall_repro.py <threads> <hogs> [steps]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from conftest import golden, set_lstm_dropout  # noqa: E402
from oracle import nets as on, skeleton as sk, train as ot  # noqa: E402
import stall_probe  # noqa: E402

n, hogs = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.set_num_threads(n)
procs = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(hogs)]
try:
    band, real16 = golden("g10_dropout_band.npz"), golden("real16.npz")
    x0, target, body, R = [torch.tensor(real16[k]) for k in ("x", "target", "skl", "R")]
    t = target[:, :, 20].contiguous()
    h0, c0 = ot.zeros_state(x0.shape[0])
    tgt = target[:, :, list(sk.UPPER_MAP)]
    torch.manual_seed(int(band["init_upper"]))
    net = on.UpperNet().train()
    set_lstm_dropout(net, 0.0)
    opt = torch.optim.Adam(net.parameters(), lr=float(band["lr"]))
    os.environ["MMEGO_STALL_PROBE"] = "30"
    w = stall_probe.arm("g10 loop, %d threads beside %d busy processes" % (n, hogs))
    t0 = time.time()
    for s in range(1, steps + 1):
        opt.zero_grad()
        loss = ot.l1_sum(net(x0.clone(), h0, c0, body, R, t)[0], tgt)
        loss.backward()
        opt.step()
        if s % 25 == 0:
            print("step %3d at %.1f s" % (s, time.time() - t0), flush=True)
    w.stop()
    print("threads %d, busy processes %d: %d steps in %.1f s" % (n, hogs, steps, time.time() - t0), flush=True)
finally:
    for p in procs:
        p.kill()
