"""Per-call-site device time of one eager U+L step (event pairs around every C-ABI launch).
Usage (GPU box): python scripts/profile_calls.py [iters] [both|upper|lower]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmego_amd import hip  # noqa: E402
from mmego_amd.train_step import StageStep  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
which = sys.argv[2] if len(sys.argv) > 2 else "both"      # "upper" / "lower": that stage's body alone
dev = torch.device("cuda:0")
imu, upper, lower, upper_frozen = bench.build_hip_models(dev)
x, imu_in, body, target = bench.synth_batch(1234, dev)
su = StageStep("upper", upper, imu, use_graph=False)
sl = StageStep("lower", lower, imu, upper_frozen=upper_frozen, use_graph=False)
su.bind(x, imu_in, body, target)
sl.bind(x, imu_in, body, target)


def step():
    if which in ("both", "upper"):
        su.step()
    if which in ("both", "lower"):
        sl.step()


step()
torch.cuda.synchronize()
rec = []
orig = hip.call


def key_of(name, a):
    if name == "gemm":
        return "gemm M%d N%d K%d nb%d split%d sak%d sbk%d" % (a[10], a[11], a[12], a[13], a[20], a[2], a[4])
    if name == "lstm_step":
        return "lstm_step Bn%d H%d" % (a[1], a[2])
    if name in ("bn_train_stats", "bn_backward", "colsum", "affine_act", "copy2d", "relu_mask"):
        ints = [v for v in a if isinstance(v, int) and not isinstance(v, bool)]
        return name + " " + " ".join(str(v) for v in ints[-3:])
    return name


def timed(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(name, *args)
    e1.record()
    rec.append((key_of(name, args), e0, e1))


hip.call = timed
for _ in range(iters):
    step()
torch.cuda.synchronize()
hip.call = orig
agg = collections.defaultdict(lambda: [0, 0.0])
for k, a, b in rec:
    agg[k][0] += 1
    agg[k][1] += a.elapsed_time(b)
tot = sum(v[1] for v in agg.values())
print("total %.2f ms/step over %d launches/step" % (tot / iters, len(rec) // iters))
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-70s n/step %4d  avg %8.1f us  %6.2f ms/step %5.1f%%" % (k, n // iters, ms / n * 1e3, ms / iters, 100 * ms / tot))
if os.environ.get("MMEGO_PROFILE_ORDER"):
    # the last step's calls in launch order (event-pair time of each)
    n = len(rec) // iters
    print("-- launch order of one step --")
    for i, (k, a, b) in enumerate(rec[-n:]):
        print("%4d %-70s %8.1f us" % (i, k, a.elapsed_time(b) * 1e3))
