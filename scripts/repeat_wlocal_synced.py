"""Diagnostic: run tests/test_hip_local.py::test_train_upper_wlocal_from_synced_states N times in one process (after one pass of
the Upper/Lower synced-state test) and report every failure.  usage: python scripts/repeat_wlocal_synced.py [N]"""
import os
import sys
import traceback

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_hip_local as tl  # noqa: E402
import test_hip_parity as tp  # noqa: E402

# checksum of the ORACLE's parameters after each of its Adam steps (is the CPU side the same from run to run?)
sums = []
_orig_step = torch.optim.Adam.step


def _step(self, *a, **k):
    r = _orig_step(self, *a, **k)
    with torch.no_grad():
        sums.append(sum(float(p.double().abs().sum()) for g in self.param_groups for p in g["params"]))
    return r


torch.optim.Adam.step = _step
if len(sys.argv) > 2:
    torch.set_num_threads(int(sys.argv[2]))
print("torch threads", torch.get_num_threads())
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
tp.test_train_steps_from_synced_states(dev)
bad = 0
for i in range(n):
    del sums[:]
    try:
        tl.test_train_upper_wlocal_from_synced_states(dev)
    except AssertionError as e:
        bad += 1
        print("run %d FAILED: %s" % (i, str(e)[:300]))
        traceback.print_exc(limit=2)
    print("run %d oracle checksums %s" % (i, ["%.9f" % v for v in sums[:3]]))
    if i % 3 == 0:
        tl.test_train_upper_wlocal(dev)       # (the unsynced twin in between, as in the suite)
print("%d / %d runs failed" % (bad, n))
