"""CPU-only diagnostic behind the NOISE_GRAD handling of the synced-state tests: is the ORACLE's trajectory the
same from run to run?  Three Adam steps of oracle.UpperNetwlocal on the G6 fixture, repeated N times in one process; prints the
parameter checksum after every step.  The thread count changes the trajectory (8 threads: ...351457758, 1 thread: ...350760054 after
step 1 in the build container), and on a many-core host (128 threads on the GPU boxes) it differed from run to run: multi-threaded
reductions flip the sign of zero-true-gradient noise, Adam turns the sign into a +-lr step.  With one thread it is reproducible.  No GPU.
usage: python scripts/repeat_wlocal_synced.py [N] [threads]"""
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from oracle import nets as on  # noqa: E402
from oracle import skeleton as sk  # noqa: E402
from oracle import train as ot  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
if len(sys.argv) > 2:
    torch.set_num_threads(int(sys.argv[2]))
print("torch threads", torch.get_num_threads())
g = np.load(os.path.join(root, "tests", "golden", "g6_train.npz"))
x0, body, R, t, target = [torch.tensor(g[k]) for k in ("x", "body", "R", "t", "target")]
h0, c0 = ot.zeros_state(x0.shape[0])
tgt = target[:, :, list(sk.UPPER_MAP)]
seen = set()
for i in range(n):
    torch.manual_seed(602)
    net = on.UpperNetwlocal().train()
    for m in net.modules():
        if isinstance(m, torch.nn.LSTM):
            m.dropout = 0.0
    opt = torch.optim.Adam(net.parameters(), lr=3e-5)
    sums = []
    for _ in range(3):
        opt.zero_grad()
        ot.l1_sum(net(x0.clone(), h0, c0, h0, c0, body, R, t)[0], tgt).backward()
        opt.step()
        with torch.no_grad():
            sums.append(sum(float(p.double().abs().sum()) for p in net.parameters()))
    seen.add(tuple(sums))
    print("run %d oracle parameter checksums %s" % (i, ["%.9f" % v for v in sums]))
print("%d distinct trajectories in %d runs" % (len(seen), n))
