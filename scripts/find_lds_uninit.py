"""Which kernel of a stage step reads LDS it never wrote?  Runs one eager Lower (or Upper) training-step body with the LDS of every CU
filled with NaN bit patterns (scripts/lds_poison.hip) in front of EVERY launch, compares the gradient buffer with a clean run, and
-- when they differ -- poisons in front of one launch at a time to name the kernel.  (r05: the Lower stage's gradients changed from
run to run when another branch's kernels with bf16 / partial-sum LDS contents ran beside it.)
build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared scripts/lds_poison.hip -o scripts/exp/liblds_poison.so"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mmego_amd import hip  # noqa: E402
from mmego_amd.train_step import StageStep  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "exp", "liblds_poison.so"))
sink = torch.zeros(4, dtype=torch.int32, device=dev)
stage = sys.argv[1] if len(sys.argv) > 1 else "lower"
pattern = int(sys.argv[2], 16) if len(sys.argv) > 2 else 0x7fc00000


def poison():
    rc = lib.lds_poison(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_uint(pattern), ctypes.c_void_p(sink.data_ptr()))
    assert rc == 0, rc


x, imu_in, body, target = [v.to(dev) for v in bench.synth_batch(1234, "cpu")]


def run(which):
    """one eager body of a freshly seeded stage; which: None (clean), "all", or the index of the one launch to poison in front of"""
    himu, hup, hlo, hfr = bench.build_hip_models(dev)
    bench._lstm_dropout_off(hup, hlo)
    st = (StageStep("upper", hup, himu, lr=3e-5, use_graph=False) if stage == "upper" else
          StageStep("lower", hlo, himu, upper_frozen=hfr, lr=3e-5, use_graph=False))
    st.bind(x, imu_in, body, target)
    st._body()                       # warm-up (arenas, attributes)
    torch.cuda.synchronize()
    names = []
    orig = hip._launch

    def launch(name, *args):
        if which == "all" or which == len(names):
            poison()
        names.append(name)
        orig(name, *args)
    hip._launch = launch
    try:
        st._body()
        torch.cuda.synchronize()
    finally:
        hip._launch = orig
    return st.net.flat().flat_g.detach().cpu().clone(), st.loss.item(), names


g0, l0, names = run(None)
g1, l1, _ = run(None)
print("clean vs clean: grad diff %.3g, loss %.9g %.9g, %d launches" % ((g0 - g1).abs().max().item(), l0, l1, len(names)))
ga, la, _ = run("all")
bad = ~torch.isfinite(ga)
print("poison in front of every launch: grad diff %.3g (non-finite: %d), loss %.9g" % ((ga - g0).nan_to_num(1e30).abs().max().item(), int(bad.sum()), la))
if (ga - g0).nan_to_num(1e30).abs().max().item() > 0 or la != l0:
    for k in range(len(names)):
        gk, lk, _ = run(k)
        d = (gk - g0).nan_to_num(1e30).abs().max().item()
        if d > 0 or lk != l0:
            print("  launch %3d  mmego_%-28s grad diff %.3g loss %.9g" % (k, names[k], d, lk), flush=True)
print("done")
