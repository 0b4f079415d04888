"""GPU: parity AT THE SHAPE AND ARRANGEMENT THAT IS BENCHMARKED.

The kernel-level and B=4 tests elsewhere run other dispatch branches than the timed step does (split-K thresholds, gemm64 vs
gemm32kq, multi-round mlp_* workgroups, BatchNorm partial-block counts, tconv NG, the persistent projection GEMM and the
LDS-DMA recurrent step at 512 rows).  Here the timed engine itself -- `ConcurrentStages(use_graph=True)` at B=64, T=8, N=128
with IMU_Net(hidden 512) -- takes ONE step and is compared with one oracle Train_Upper + Train_Lower body
(reference Processor/Train/Train_Upper.py:134-187, Train_Lower.py:155-230) on the same seeded weights and minibatch:
losses 2e-5 rel., every element of both gradient buffers 2e-4 of the stage's largest gradient, joints 1e-3 cm, post-Adam
parameters at the step-1 bar of test_hip_parity._compare_training.  bench.py prints the same figures in `parity`.
"""
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    from mmego_amd import hip
    hip.lib()
    return torch.device("cuda:0")


@pytest.mark.parametrize("use_graph", [True, False])
def test_ul_step_at_bench_shape_against_oracle(dev, use_graph):
    import bench
    torch.set_num_threads(bench.host_cores())
    r = bench.ul_step_parity(dev, use_graph=use_graph)
    noise = re.compile(bench.NOISE_GRAD)
    for tag in ("upper", "lower"):
        assert r["loss_rel_err_" + tag] < 2e-5, (tag, r["loss_rel_err_" + tag])
        assert r[tag + "_cm"] < 1e-3, (tag, r[tag + "_cm"])                         # joints: the north-star bar, in cm
        scale = r["tensors"][tag]["scale"]
        for k, (eg, dp) in r["tensors"][tag]["per_param"].items():
            assert eg < 2e-4 * scale, (tag, k, eg, scale)
            if not noise.search(k):
                assert dp <= 6e-5 + 2e-6, (tag, k, dp)                              # at most one +-lr flip of a ~0 gradient's sign
        assert r["param_frac_moved_" + tag] < 0.05, (tag, r["param_frac_moved_" + tag])


def _train_200(stage, p, dseed, dev):
    """The HIP side of tests/golden/make_dropout_band.py: same fixed initial weights, `steps` StageStep.step()s (HIP graph) on
    the 16 real sequences as one minibatch with the recorded head pose, then the eval-mode train-set joint error in cm."""
    from conftest import golden, load_weights
    from mmego_amd import nets
    from mmego_amd.train_step import StageStep
    from oracle import skeleton as sk
    band, real = golden("g10_dropout_band.npz"), golden("real16.npz")
    x0, target, body, R = [torch.tensor(real[k]).to(dev) for k in ("x", "target", "skl", "R")]
    B, T = x0.shape[0], x0.shape[1]
    torch.manual_seed(int(band["init_upper"] if stage == "upper" else band["init_lower"]))
    net = (nets.UpperNet() if stage == "upper" else nets.LowerNet(64)).to(dev).train()
    net.lstm_dropout = p
    frozen = None
    if stage == "lower":
        frozen = load_weights(nets.UpperNet(), golden("w_upper_pretrained.npz")).to(dev).eval()
    st = StageStep(stage, net, None, upper_frozen=frozen, lr=float(band["lr"]), use_graph=True)
    st.bind(x0.clone(), torch.zeros(B, T, 20, 15, device=dev), body, target, R_gt=R)
    net.seed_counter().fill_(dseed)
    curve = []
    for s in range(1, int(band["steps"]) + 1):
        st.step()
        if s % 50 == 0:
            curve.append(st.loss.item())
    net.eval()
    t = target[:, :, 20].contiguous()
    h0 = torch.zeros(6, B, 64, device=dev)
    with torch.no_grad():
        if stage == "upper":
            pred = net(x0.clone(), h0, h0.clone(), body, R, t)[0]
            tgt = target[:, :, list(sk.UPPER_MAP)]
        else:
            xl = x0.clone()
            up = frozen(xl, h0, h0.clone(), body, R, t)[0]
            pred = net(up.clone(), xl, None, None, None, None, body, R, t)[0]
            tgt = target[:, :, list(sk.LOWER_MAP)]
    return (pred - tgt).norm(dim=-1).mean().item() * 100.0, curve


@pytest.mark.parametrize("stage", ["upper", "lower"])
def test_dropout_active_training_lands_in_the_reference_band(dev, stage):
    """VERDICT r02 item 1c.  The benchmarked configuration trains with nn.LSTM(dropout=0.1) live (reference
    Net/Upper_Net.py:333, Net/Lower_Net.py:91-93); the two sides draw their masks from different RNGs, so parity there is
    statistical.  tests/golden/g10_dropout_band.npz holds the REAL reference's final train-set joint error after 200 dropout-active
    steps from fixed initial weights over 6 dropout seeds (the run-to-run band), the dropout-free deterministic run, and runs at
    twice the rate (which leave the band upwards: the statistic separates a wrong rate from the band).  Here:
      * dropout 0.1, six HIP dropout seeds: their mean within three standard errors (two-sample) of the reference's mean, every
        run within the reference band widened by its width on each side;
      * dropout 0: the HIP run's final error and 50/100/150/200-step losses inside the spread of the reference's dropout-free runs
        from 1e-7-perturbed initial weights.
    What the statistic can and cannot see (the fixture's own numbers): for Upper_Net the runs at rate 0.2 (4.37-4.47 cm) leave the
    band [4.08, 4.33] cm upwards, the dropout-free runs ([3.97, 4.23] cm) overlap its lower half; for Lower_Net neither separates
    ([4.90, 5.12] cm against 4.90-5.25 and [4.91, 5.31] cm).  It pins "trains as well as the reference, dropout live", and a
    grossly wrong rate in Upper_Net; the mask rate itself is pinned by test_hip_parity.test_lstm64_fused_dropout_and_bias_pair."""
    from conftest import golden
    band = golden("g10_dropout_band.npz")
    ref = band["ref.%s.p01.err_cm" % stage]
    lo, hi = float(ref.min()), float(ref.max())
    w = hi - lo
    # six HIP dropout seeds against the reference's six: the means within three standard errors of each other (two-sample), every
    # run within the reference band widened by its width (a future change of a kernel's rounding moves these end points around
    # inside their spread -- the criterion is statistical on purpose)
    errs = [_train_200(stage, 0.1, 1000 + k, dev)[0] for k in range(6)]
    import numpy as np
    se = float(np.sqrt(ref.var(ddof=1) / len(ref) + np.var(errs, ddof=1) / len(errs)))
    assert abs(float(np.mean(errs)) - float(ref.mean())) < 3.0 * se + 0.02, (stage, errs, list(ref), se)
    for e in errs:
        assert lo - w <= e <= hi + w, (stage, errs, (lo, hi))
    # Dropout-free: 200 Adam steps at lr 3e-4 amplify ROUNDING-sized differences -- the reference itself, started from weights
    # perturbed by 1e-7 relative, ends anywhere in [3.97, 4.23] cm (Upper) / [4.91, 5.31] cm (Lower), and the CPU oracle's
    # dropout-free run is not reproducible from process to process (multi-threaded reductions: 5.04 and 5.47 cm were both seen for
    # Lower).  So a different-but-correct implementation is held to THAT band (widened by half its width), end point and curve.
    e0, curve0 = _train_200(stage, 0.0, 0, dev)
    pe = list(band["ref.%s.p00_perturbed.err_cm" % stage]) + list(band["ref.%s.p00.err_cm" % stage])
    pc = list(band["ref.%s.p00_perturbed.loss_curve" % stage]) + list(band["ref.%s.p00.loss_curve" % stage])
    plo, phi = float(min(pe)), float(max(pe))
    assert plo - 0.5 * (phi - plo) <= e0 <= phi + 0.5 * (phi - plo), (stage, e0, pe)
    for i, a in enumerate(curve0):
        clo, chi = min(c[i] for c in pc), max(c[i] for c in pc)
        assert clo - (chi - clo) - 1e-3 * clo <= a <= chi + (chi - clo) + 1e-3 * chi, (stage, i, curve0, [c[i] for c in pc])
    print("dropout band %s: reference [%.4f, %.4f] cm, HIP %s; dropout-free: reference (perturbed 1e-7) [%.4f, %.4f] cm, HIP %.4f"
          % (stage, lo, hi, ["%.4f" % e for e in errs], plo, phi, e0))


def test_engines_agree_bit_for_bit_at_bench_shape(dev):
    """The arrangements bench.py reports beside `value` -- stages one after the other (per-stage HIP graphs), the IMU-shared engine
    and the prefetch-pipelined engine -- against the timed one (`ConcurrentStages`, one graph) at B=64, T=8, N=128 with IMU_Net(512),
    dropout live (same seeds): three steps each, losses / gradient buffers / parameters / BatchNorm buffers bit-identical.  (At B=16 the
    same is checked in test_hip_local; here the 512-row recurrences, the persistent projection kernel and the large-grid tails run.)
    Also with the bodies captured through plan.StepPlan (train_step._MULTI_GRAPH: a graph per stream segment + event waits)."""
    import bench
    from mmego_amd.train_step import ConcurrentStages, PipelinedStages, SharedImuStages, StageStep
    x, imu_in, body, target = bench.synth_batch(1234, dev)

    def run(kind):
        himu, hup, hlo, hfr = bench.build_hip_models(dev)
        himu_l = bench.clone_imu(himu, dev)
        own = kind in ("concurrent", "sequential")
        su = StageStep("upper", hup, himu if own else None, lr=3e-5, use_graph=kind == "sequential")
        sl = StageStep("lower", hlo, himu_l if own else None, upper_frozen=hfr, lr=3e-5, use_graph=kind == "sequential")
        if kind == "concurrent":
            eng = ConcurrentStages([su, sl], use_graph=True)
        elif kind == "shared":
            eng = SharedImuStages(himu, [su, sl], imu_in, use_graph=True)
        elif kind == "pipelined":
            eng = PipelinedStages([su, sl], [himu, himu_l], imu_in, use_graph=True)
        else:
            eng = None
        su.bind(x, imu_in, body, target)          # (the engines hand the stages their head-pose buffers: bind comes after)
        sl.bind(x, imu_in, body, target)
        if kind == "pipelined":
            eng.prime()
        for _ in range(3):
            if eng is None:
                su.step(); sl.step()
            else:
                eng.step()
        torch.cuda.synchronize()
        return su, sl
    ref = run("concurrent")
    from mmego_amd import train_step
    kinds = [("sequential", False), ("shared", False), ("pipelined", False), ("concurrent", True), ("pipelined", True), ("sequential", True)]
    for kind, multi in kinds:
        # multi: the body as one linear-chain HIP graph per stream segment (plan.StepPlan) instead of one graph with branches
        was = train_step._MULTI_GRAPH
        train_step._MULTI_GRAPH = multi
        try:
            got = run(kind)
        finally:
            train_step._MULTI_GRAPH = was
        kind = kind + ("/plan" if multi else "")
        for a, b in zip(got, ref):
            assert a.loss.item() == b.loss.item(), (kind, a.stage, a.loss.item(), b.loss.item())
            assert torch.equal(a.net.flat().flat_g, b.net.flat().flat_g), (kind, a.stage)
            assert torch.equal(a.net.flat().flat_p, b.net.flat().flat_p), (kind, a.stage)
            for ba, bb in zip(a.net.buffers(), b.net.buffers()):
                assert torch.equal(ba, bb), (kind, a.stage)


@pytest.mark.gpu
def test_engines_agree_bit_for_bit_with_split3_imu(dev):
    """The same with both frozen IMU_Net forwards in the split3 mode (fp32-accurate piece products): stages one after the other and the
    prefetch-pipelined engine -- what `main.py --train --imu_precision split3` runs -- against ConcurrentStages, three steps each, bit
    for bit."""
    import bench
    from mmego_amd.train_step import ConcurrentStages, PipelinedStages, StageStep
    x, imu_in, body, target = bench.synth_batch(1234, dev)

    def run(kind):
        himu, hup, hlo, hfr = bench.build_hip_models(dev)
        himu_l = bench.clone_imu(himu, dev)
        himu.precision = himu_l.precision = "split3"
        own = kind != "pipelined"
        su = StageStep("upper", hup, himu if own else None, lr=3e-5, use_graph=kind == "sequential")
        sl = StageStep("lower", hlo, himu_l if own else None, upper_frozen=hfr, lr=3e-5, use_graph=kind == "sequential")
        eng = None
        if kind == "concurrent":
            eng = ConcurrentStages([su, sl], use_graph=True)
        elif kind == "pipelined":
            eng = PipelinedStages([su, sl], [himu, himu_l], imu_in, use_graph=True)
        su.bind(x, imu_in, body, target)
        sl.bind(x, imu_in, body, target)
        if kind == "pipelined":
            eng.prime()
        for _ in range(3):
            if eng is None:
                su.step(); sl.step()
            else:
                eng.step()
        torch.cuda.synchronize()
        return su, sl
    ref = run("concurrent")
    for kind in ("sequential", "pipelined"):
        got = run(kind)
        for a, b in zip(got, ref):
            assert a.loss.item() == b.loss.item(), (kind, a.stage, a.loss.item(), b.loss.item())
            assert torch.equal(a.net.flat().flat_g, b.net.flat().flat_g), (kind, a.stage)
            assert torch.equal(a.net.flat().flat_p, b.net.flat().flat_p), (kind, a.stage)
    from mmego_amd import blocks
    assert blocks.seq_xcd_errors() == 0


@pytest.mark.gpu
def test_fused_head_loss_launch_is_bit_identical_to_the_three_launches():
    """mmego_head_fk_loss (kinematics + transform + L1(sum) loss + its gradient + kinematics backward in one launch, what StageStep
    uses) against head_fk_forward -> l1_loss -> head_fk_backward (nets._FUSED_HEAD_LOSS = False): predictions, both loss figures and every
    gradient bit for bit (the loss figures to an fp32 ulp: fixed but different summation order), Upper_Net, Lower_Net and UpperNetwlocal,
    B = 64 (8 workgroups, ticketed fixed-order sum) and B = 5; replayed twice more to check the ticket resets."""
    import os
    from mmego_amd import nets, nets_local
    from mmego_amd.train_step import StageStep
    dev = torch.device("cuda:0")
    for Bq in (5, 64):
        g = torch.Generator().manual_seed(3 + Bq)
        x = torch.randn(Bq, 8, 128, 6, generator=g).to(dev)
        imu = torch.randn(Bq, 8, 20, 15, generator=g).to(dev)
        body = (0.2 * torch.randn(Bq, 20, 3, generator=g)).to(dev)
        target = torch.randn(Bq, 8, 21, 3, generator=g).to(dev)
        Rg = torch.linalg.qr(torch.randn(Bq, 8, 3, 3, generator=g))[0].contiguous().to(dev)
        for kind in ("upper", "lower", "wlocal"):
            res = []
            for fused in (True, False):
                nets._FUSED_HEAD_LOSS = fused
                try:
                    torch.manual_seed(9)
                    if kind == "lower":
                        net, fr = nets.LowerNet(64).to(dev).train(), nets.UpperNet().to(dev).eval()
                        st = StageStep("lower", net, None, upper_frozen=fr, use_graph=False)
                    else:
                        net = (nets.UpperNet() if kind == "upper" else nets_local.UpperNetwlocal()).to(dev).train()
                        st = StageStep("upper", net, None, use_graph=False)
                    net.lstm_dropout = 0.0
                    st.bind(x, imu, body, target, R_gt=Rg)
                    for _ in range(3):                       # (the ticket of the fused launch must come back to 0 every time)
                        st._body()
                    torch.cuda.synchronize()
                    assert getattr(net, "_dy_ready", False) == fused
                    res.append((st.last_pred.clone(), st.loss2.clone(), net.flat().flat_g.clone()))
                finally:
                    nets._FUSED_HEAD_LOSS = True
            (p1, l1, g1), (p0, l0, g0) = res
            assert torch.equal(p1, p0) and torch.equal(g1, g0), (kind, Bq)
            # (the two loss figures: fp64 sums in a different, fixed order -- equal after rounding to fp32 up to one ulp)
            assert torch.allclose(l1, l0, rtol=2e-7, atol=0.0), (kind, Bq, l1, l0)


@pytest.mark.gpu
def test_precision_modes_joint_error_against_the_oracle(dev):
    """bench.mode_joint_errors (`config5.joint_error_vs_oracle` in the bench detail): the eval forward IMU_Net -> Upper_Net ->
    Lower_Net at the bench shape in each precision mode against the CPU oracle's fp32 forward.  The native path and the split3 mode
    are inside the parity bar (1e-3 cm); the bf16 mode (what config 5 times, opt-in, outside the bar) is held to 0.3 cm -- measured
    0.03 cm upper / 0.10 cm lower body on these random-init nets."""
    import bench
    r = bench.mode_joint_errors(dev)
    for mode in ("fp32", "split3"):
        assert r[mode]["upper_cm"] < 1e-3 and r[mode]["lower_cm"] < 1e-3, (mode, r[mode])
    assert r["bf16"]["upper_cm"] < 0.3 and r["bf16"]["lower_cm"] < 0.3, r["bf16"]
    assert r["bf16"]["lower_cm"] > r["split3"]["lower_cm"]          # (the modes are really different arithmetic)
