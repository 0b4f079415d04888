"""r06: the whole U+L step with both IMU_Net forwards in split3 mode as two CONCURRENT graph branches (the arrangement whose Lower
gradients changed in ~10 % of runs in r05: gpurun_out/r05_t_vox.log, r05_split3_tests.log), N fresh engines x M steps each, both stages'
gradient buffers compared bit for bit with the first engine's.  Run once per library build (MMEGO_HIP_LIB=...): the product build (no
packed-fp32 instruction in any kernel) against lib/variants/libmmego_hip_pk.so (the r05 code generation)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mmego_amd import blocks, hip  # noqa: E402
from mmego_amd.train_step import ConcurrentStages, StageStep  # noqa: E402

dev = torch.device("cuda:0")
n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_step = int(sys.argv[2]) if len(sys.argv) > 2 else 4
two = len(sys.argv) > 3 and sys.argv[3] == "two"
unguarded = os.environ.get("MMEGO_COEXEC_UNGUARDED") == "1"
x, imu_in, body, target = [v.to(dev) for v in bench.synth_batch(1234, "cpu")]
ref, bad, worst = None, 0, 0.0
ctx = blocks.split3_two_chains(True) if two and hasattr(blocks, "split3_two_chains") else None
if two and ctx is None:
    blocks.SPLIT3_TWO_CHAINS = True
for it in range(n_eng):
    himu, hup, hlo, hfr = bench.build_hip_models(dev)
    himu_l = bench.clone_imu(himu, dev)
    himu.precision = himu_l.precision = "split3"
    bench._lstm_dropout_off(hup, hlo)
    su = StageStep("upper", hup, himu, lr=3e-5, use_graph=True)
    sl = StageStep("lower", hlo, himu_l, upper_frozen=hfr, lr=3e-5, use_graph=True)
    su.bind(x, imu_in, body, target)
    sl.bind(x, imu_in, body, target)
    kw = {"unguarded": True} if unguarded else {}
    eng = ConcurrentStages([su, sl], use_graph=True, **kw)
    gs = []
    for s in range(n_step):
        eng.step()
        torch.cuda.synchronize()
        gs.append([st.net.flat().flat_g.detach().clone() for st in (su, sl)])
    if ref is None:
        ref = gs
    diff = max(float((a - b).abs().max()) for ga, gb in zip(gs, ref) for a, b in zip(ga, gb))
    if diff > 0:
        bad += 1
        worst = max(worst, diff)
    del su, sl, eng
print("%s  two_chains=%s: %d of %d engines (x %d steps) differ from the first; worst |diff| %.3g; persistent-launch errors %d" % (
    os.path.basename(hip.LIBPATH), two, bad, n_eng, n_step, worst, blocks.seq_xcd_errors()), flush=True)
