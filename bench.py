#!/usr/bin/env python3
"""Headline benchmark: training frames/s of the metric unit "U+L step" (SURVEY.md section 8-d).

One step = one Train_Upper.train_once body + one Train_Lower.train_once body on the same synthetic
minibatch (per GPU: B=64 sequences x T=8 frames x N=128 points, 21 joints), each as in the reference:
frozen IMU_Net forward inside both, frozen Upper_Net forward inside the Lower body, L1(sum) loss,
backward, Adam.  fp32 end to end.  Inputs are resident in HBM before the timed region.
The two bodies are independent programs in the reference (the Lower stage loads a frozen, pre-trained
Upper_Net), so by default they run as two concurrent branches of one HIP graph (--sequential runs them
one after the other; `ms_per_step_sequential` is always reported as well).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL gradient all-reduce)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, timed
live with events on the launch stream) and, at N=1, `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 MFMA peak (same guide)
PEAK_HBM_GBPS = 8000.0
B, T, N = 64, 8, 128
# algorithmic FLOPs per frame of the literal U+L step (IMU_Net forward counted in BOTH bodies, as the reference
# runs it): 2 x 444.96 (IMU) + 6.45 + 2.15 (Upper f/b + frozen forward) + 27.79 (Lower f/b) = 926.3 MFLOP
STEP_MFLOP_PER_FRAME = 926.3


def synth_batch(seed, device):
    """Synthetic minibatch with Sample_data's column statistics."""
    g = torch.Generator().manual_seed(seed)
    mu, sd = torch.tensor([0.84, 0.05, 0.18]), torch.tensor([0.41, 0.30, 0.38])
    xyz = torch.randn(B, T, N, 3, generator=g) * sd + mu
    x = torch.zeros(B, T, N, 6)
    x[..., :3] = xyz
    x[..., 3] = xyz.norm(dim=-1)
    x[..., 4] = torch.randn(B, T, N, generator=g) * 0.41
    x[..., 5] = torch.rand(B, T, N, generator=g) * 36 + 10
    dead = torch.rand(B, T, N, generator=g) < 0.40
    dead[:, :, :16] = False                                   # keep >= 16 live points per frame
    x[dead] = 0.0
    imu = torch.zeros(B, T, 20, 15)
    q = torch.linalg.qr(torch.randn(B, T, 20, 3, 3, generator=g))[0]
    imu[..., :9] = q.reshape(B, T, 20, 9)
    imu[..., 9:12] = torch.randn(B, T, 20, 3, generator=g)
    imu[..., 12:] = 0.3 * torch.randn(B, T, 20, 3, generator=g)
    bones = torch.tensor([[0.0, 0.0, 0.18], [0.0, 0.0, 0.2], [0.0, 0.0, 0.15], [0.17, 0.0, 0.03], [-0.17, 0.0, 0.03],
                          [0.27, 0.0, 0.0], [0.25, 0.0, 0.0], [0.08, 0.0, 0.0], [-0.27, 0.0, 0.0], [-0.25, 0.0, 0.0],
                          [-0.08, 0.0, 0.0], [0.0, 0.0, 0.2], [0.09, 0.0, 0.0], [-0.09, 0.0, 0.0], [0.0, 0.0, 0.4],
                          [0.0, 0.0, 0.4], [0.0, 0.12, 0.05], [0.0, 0.0, 0.4], [0.0, 0.0, 0.4], [0.0, 0.12, 0.05]])
    body = bones.unsqueeze(0).repeat(B, 1, 1)                  # identical across the batch, as in the reference
    skel = torch.cumsum(torch.cat((torch.tensor([[0.8, 0.0, 0.9]]), bones)), 0)
    target = skel.view(1, 1, 21, 3) + 0.05 * torch.randn(B, T, 21, 3, generator=g)
    return [v.contiguous().to(device) for v in (x, imu, body, target)]


def build_hip_models(device):
    from mmego_amd import nets
    torch.manual_seed(1234)
    imu = nets.IMUNet(15, 9, 512, 2, True, 0.1)
    upper = nets.UpperNet()
    lower = nets.LowerNet(64)
    upper_frozen = nets.UpperNet()
    upper_frozen.load_state_dict(upper.state_dict())
    return imu.to(device).eval(), upper.to(device).train(), lower.to(device).train(), upper_frozen.to(device).eval()


def clone_imu(imu, device):
    """Second frozen IMU_Net instance with the same weights: each stage program owns its copy (as two reference
    processes would), which is what lets the two stage bodies run concurrently."""
    from mmego_amd import nets
    c = nets.IMUNet(15, 9, 512, 2, True, 0.1)
    c.load_state_dict(imu.state_dict())
    return c.to(device).eval()


def profile_kernels(steps_fn, names, iters=3):
    """Average launch duration (ms) and launch count per step of the named C-ABI entry points, measured with
    event pairs on the launch stream while `steps_fn` runs eagerly (no graph)."""
    from mmego_amd import hip
    rec = {n: [] for n in names}
    orig = hip.call

    def timed_call(name, *args):
        if name in rec:
            # The big projections are launched twice back to back between the event pair (the product is idempotent), so the
            # event / launch latency of an eager launch (5-10 us) is amortised and the figure is the kernel's own duration,
            # comparable with rocprofv3's AverageNs.  The recurrent step updates c in place: launched once.
            rep = 2 if (name == "gemm" and 2.0 * args[10] * args[11] * args[12] * args[13] > 1e10 and not args[18]) else 1
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(rep):
                orig(name, *args)
            e1.record()
            rec[name].append((e0, e1, args, rep))
        else:
            orig(name, *args)
    hip.call = timed_call
    try:
        for _ in range(iters):
            steps_fn()
        torch.cuda.synchronize()
    finally:
        hip.call = orig
    out = {n: [(a.elapsed_time(b) / rep, args) for a, b, args, rep in v] for n, v in rec.items()}
    return out, iters


def recurrence_graph(device, Bn=512, H=512, T=20, replays=20):
    """One rnn_fast layer's recurrence (T timesteps, Bn rows, both directions) as REPLAYED HIP GRAPHS, the way the step runs it
    (an eager event pair around a launch includes 5-10 us of host launch latency): (a) one launch per timestep for both
    directions, (b) two concurrent chains of single-direction launches (what the first IMU_Net forward of the U+L step uses),
    (c) two independent nets' recurrences side by side on two streams (not what the step uses).  Per
    timestep of ONE net: us and fraction of the fp32 MFMA peak over the T-1 product-carrying steps."""
    from mmego_amd import blocks, ops
    nets = [blocks.LstmParams(H, H, 1).to(device) for _ in range(2)]
    xps = [torch.randn(Bn * T, 8 * H, device=device) * 0.1 for _ in range(2)]
    ars = [ops.Arena(device) for _ in range(2)]
    outs = [a.get("out", (Bn * T, 2 * H)) for a in ars]
    side = torch.cuda.Stream()
    one = lambda i: blocks.lstm_recurrence(ars[i], "k", nets[i], 0, xps[i], outs[i], Bn, T)

    def single():
        with blocks.two_chains(False):
            one(0)

    def chains():
        with blocks.two_chains(True):
            one(0)

    def pair():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with blocks.two_chains(False):
            with torch.cuda.stream(side):
                one(1)
            one(0)
        cur.wait_stream(side)
    res = {}
    flops = 2.0 * 2 * Bn * 4 * H * H * (T - 1)
    for name, body, nets_in_body in (("one_launch_per_timestep", single, 1), ("two_chains", chains, 1), ("two_nets_side_by_side", pair, 2)):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            with ops.capture(g, stream=st):
                body()
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / replays * 1e3 / nets_in_body
        res[name] = {"us_per_timestep": us / T, "tflops": flops / us / 1e6, "frac": flops / us / 1e6 / PEAK_FP32_MFMA_TFLOPS}
    res["what"] = ("replayed graphs of one BiLSTM layer's recurrence, Bn=%d H=%d T=%d; per timestep of one net; 2 x 2 x Bn x 4H x H "
                   "flop per product step" % (Bn, H, T))
    return res


def host_cores():
    """CPU threads this process may really use: affinity, capped by the cgroup CPU quota and by the 16-core
    share a one-GPU box grants (oversubscribing a quota'd cgroup makes the baseline pathologically slow)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, int(os.environ.get("MMEGO_CPU_THREADS", "16"))))


def pmc_traffic(kernel_label):
    """(HBM bytes per launch, source) of a kernel from the COMMITTED rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in
    separate runs, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950).  Read from profiles/, not measured in this run
    (counters need rocprofv3 around the process); `traffic_source` says which passes the file holds: from r03 on they are taken
    around THIS program (`bench.py --trace-only --no-graph`, scripts/collect_profiles.sh), the r02 / r01 files around the
    micro-benchmarks scripts/bench_gemm_pair.py / bench_lstm_step.py."""
    key = kernel_label.split(" ")[0]
    for name, what in (("r06_pmc_counters.json", "rocprofv3 --pmc passes of `bench.py --trace-only --no-graph` itself (scripts/collect_r06.sh)"),
                       ("r05_pmc_counters.json", "rocprofv3 --pmc passes of `bench.py --trace-only --no-graph` itself (scripts/collect_r05.sh)"),
                       ("r04_pmc_counters.json", "rocprofv3 --pmc passes of `bench.py --trace-only --no-graph` itself (scripts/collect_r04.sh)"),
                       ("r03_pmc_counters.json", "rocprofv3 --pmc passes of `bench.py --trace-only --no-graph` itself"),
                       ("r02_pmc_counters.json", "rocprofv3 --pmc passes of the micro-benchmarks scripts/bench_gemm_pair.py / bench_lstm_step.py"),
                       ("r01_pmc_counters.json", "rocprofv3 --pmc passes of the micro-benchmarks")):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            d = json.load(open(path))
            e = d.get(key) or d.get(key + "<32>") or d.get(key.split("<")[0]) or {}
            v = e.get("hbm_bytes_per_launch")
            if v is not None:
                return v, "profiles/%s (%s; committed, not measured in this run)" % (name, what)
    return None, None


NOISE_GRAD = r"(conv[123]\.bias|tcn\.2\.bias|residual\.0\.bias|attn\.bias|to_k\.bias|fusion\.attn\.weight)$"


def _lstm_dropout_off(*modules):
    """LSTM inter-layer dropout off (the two sides draw from different RNGs): nn.LSTM.dropout on the oracle's modules,
    `lstm_dropout` on the HIP nets."""
    for m in modules:
        for mod in m.modules():
            if hasattr(mod, "dropout") and isinstance(getattr(mod, "dropout"), float):
                mod.dropout = 0.0
        if hasattr(m, "lstm_dropout"):
            m.lstm_dropout = 0.0


def ul_step_parity(device, use_graph=True, imu_precision="fp32"):
    """The arrangement that is timed, checked against the CPU oracle AT THE TIMED SHAPE: freshly seeded nets on both sides
    (identical weights, IMU_Net hidden 512), LSTM dropout off, the full synthetic minibatch (B=64, T=8, N=128: 512 rows through
    rnn_fast, 65 536 points, 7 680 graph rows -- the dispatch branches of the timed step), ONE
    `ConcurrentStages(use_graph=True).step()` (frozen IMU_Net forward in both bodies, frozen Upper_Net forward in the Lower body,
    L1(sum) loss, backward, fused Adam) against one oracle Train_Upper + Train_Lower body (Processor/Train/Train_Upper.py:134-187,
    Train_Lower.py:155-230).  Returns the figures and the per-parameter tensors the test asserts on."""
    import re
    from copy import deepcopy
    from mmego_amd.train_step import ConcurrentStages, StageStep
    from oracle import nets as on
    from oracle import skeleton as sk
    from oracle import train as ot
    noise = re.compile(NOISE_GRAD)
    torch.manual_seed(1234)
    o_imu = on.IMUNet(15, 9, 512, 2, True, 0.1).eval()
    o_up, o_lo = on.UpperNet().train(), on.LowerNet(64).train()
    himu, hup, hlo, hfr = build_hip_models(device)                          # same seed -> the same initial weights
    himu_l = clone_imu(himu, device)
    himu.precision = himu_l.precision = imu_precision          # ("split3": the fp32-accurate bf16-pipe mode, same bars)
    for (ko, vo), (kh, vh) in zip(list(o_up.state_dict().items()) + list(o_lo.state_dict().items()),
                                  list(hup.state_dict().items()) + list(hlo.state_dict().items())):
        assert ko == kh and torch.equal(vo, vh.cpu()), "same seed -> same initial weights: " + ko
    _lstm_dropout_off(o_up, o_lo, hup, hlo)
    o_fr = deepcopy(o_up).eval()
    x, imu_in, body, target = synth_batch(1234, "cpu")
    # ---- HIP: one step of the timed engine
    xd, imud, bodyd, targetd = [v.to(device) for v in (x, imu_in, body, target)]
    su = StageStep("upper", hup, himu, lr=3e-5, use_graph=use_graph)
    sl = StageStep("lower", hlo, himu_l, upper_frozen=hfr, lr=3e-5, use_graph=use_graph)
    su.bind(xd, imud, bodyd, targetd)
    sl.bind(xd, imud, bodyd, targetd)
    both = ConcurrentStages([su, sl], use_graph=use_graph)
    both.step()
    torch.cuda.synchronize()
    # ---- oracle: one Train_Upper body + one Train_Lower body
    opt_u = torch.optim.Adam(o_up.parameters(), lr=3e-5)
    opt_l = torch.optim.Adam(o_lo.parameters(), lr=3e-5)
    loss_u, joints_u = ot.upper_train_step(o_up, o_imu, opt_u, x.clone(), imu_in, body, target)
    loss_l, joints_l = ot.lower_train_step(o_lo, o_fr, o_imu, opt_l, x.clone(), imu_in, body, target)
    res = {"what": "HIP vs CPU oracle at the timed shape: same seeded weights, LSTM dropout off, all %d sequences of the synthetic "
                   "batch (%d rnn_fast rows), one ConcurrentStages(%s).step() vs one Train_Upper + Train_Lower body" %
                   (B, B * T, "HIP graph" if use_graph else "eager"),
           "tolerance_cm": 1e-3, "tensors": {}}
    for tag, st, o_net, lo_, jo in (("upper", su, o_up, loss_u, joints_u), ("lower", sl, o_lo, loss_l, joints_l)):
        res[tag + "_cm"] = (st.last_pred.detach().cpu().view_as(jo) - jo).norm(dim=-1).max().item() * 100.0
        res["loss_rel_err_" + tag] = abs(st.loss.item() - lo_.item()) / abs(lo_.item())
        po, ph = dict(o_net.named_parameters()), dict(st.net.named_parameters())
        flat = st.net.flat()
        scale = max(p.grad.abs().max().item() for p in po.values() if p.grad is not None)
        worst_g = worst_p = 0.0
        n_bad = n_all = 0
        per = {}
        for k in po:
            go = po[k].grad if po[k].grad is not None else torch.zeros_like(po[k])
            gh = flat.grad(ph[k]).detach().cpu()
            eg = (gh - go).abs().max().item()
            worst_g = max(worst_g, eg)
            dp = (ph[k].detach().cpu() - po[k].detach()).abs()
            per[k] = (eg, dp.max().item())
            if not noise.search(k):
                worst_p = max(worst_p, dp.max().item())
                n_bad += int((dp > 2e-6).sum())
                n_all += dp.numel()
        res["grad_rel_err_" + tag] = worst_g / scale
        res["param_max_abs_diff_" + tag] = worst_p
        res["param_frac_moved_" + tag] = n_bad / max(1, n_all)
        res["tensors"][tag] = {"scale": scale, "per_param": per,
                               "grads": {k: (flat.grad(ph[k]).detach().cpu().clone(), None if po[k].grad is None else po[k].grad.detach().clone())
                                         for k in po}}
    res["max_joint_distance_cm"] = max(res["upper_cm"], res["lower_cm"])
    return res


def cpu_baseline(steps, warmup, device=None):
    """The CPU oracle (a port of the reference path, pinned to it by tests/golden) on the host cores.  While it is at hand it
    also serves as the checker of the metric's "joint-err parity" (`ul_step_parity`: one U+L step of the timed engine at the
    timed shape against one oracle step -- joints, losses, gradients, post-Adam parameters)."""
    import platform
    from oracle import nets as on
    from oracle import train as ot
    ncores = host_cores()
    torch.set_num_threads(ncores)
    parity = None
    if device is not None:
        parity = ul_step_parity(device)
        del parity["tensors"]
    torch.manual_seed(1234)
    imu = on.IMUNet(15, 9, 512, 2, True, 0.1)
    upper, lower = on.UpperNet(), on.LowerNet(64)
    x, imu_in, body, target = synth_batch(1234, "cpu")
    tu, tl = ot.time_ul_step(upper, lower, imu, x, imu_in, body, target, steps=steps, warmup=warmup)
    model = platform.processor() or platform.machine()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"value": B * T / (tu + tl), "unit": "frames/s", "cores": ncores, "kind": "port",
           "cpu_model": model, "torch": torch.__version__,
           "sample": "%d U+L steps (after %d warm-up) of the same B=64,T=8,N=128 batch; t_upper %.1f ms, t_lower %.1f ms"
                     % (steps, warmup, tu * 1e3, tl * 1e3)}
    return out, parity


def _synth_points(Bq, Tq, Nq, seed):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.randn(Bq, Tq, Nq, 3, generator=g) * torch.tensor([0.41, 0.30, 0.38]) + torch.tensor([0.84, 0.05, 0.18])
    x = torch.zeros(Bq, Tq, Nq, 6)
    x[..., :3] = xyz
    x[..., 3] = xyz.norm(dim=-1)
    x[..., 4] = torch.randn(Bq, Tq, Nq, generator=g) * 0.41
    x[..., 5] = torch.rand(Bq, Tq, Nq, generator=g) * 36 + 10
    dead = torch.rand(Bq, Tq, Nq, generator=g) < 0.40
    dead[:, :, :16] = False
    x[dead] = 0.0
    R = torch.linalg.qr(torch.randn(Bq, Tq, 3, 3, generator=g))[0].contiguous()
    t = torch.randn(Bq, Tq, 3, generator=g) * 0.1 + torch.tensor([0.8, 0.0, 0.9])
    body = torch.randn(Bq, 20, 3, generator=g) * 0.2
    imu = torch.randn(Bq, Tq, 20, 15, generator=g)
    return x, R, t, body, imu


def _time_events(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def config2_forward(device):
    """BASELINE config 2: Upper_Net eval forward, synthetic B=64 T=8 N=128, fp32, one GPU.  20 forwards are captured into one
    HIP graph (the trainers replay graphs too), so the figure holds no host launch cost.  SURVEY 8-d: 2.15 MFLOP and 3 330 B
    per frame + 1.2 MB of weights per forward => t_roof ~ 7 us per 512-frame batch, BELOW the latency of one dependent kernel
    launch: this configuration is bound by its chain of dependent launches, and the fractions below say so."""
    from mmego_amd import nets
    torch.manual_seed(2)
    up = nets.UpperNet().to(device).eval()
    x0, R, t, body, _ = [v.to(device) for v in _synth_points(B, T, N, 22)]
    h0 = torch.zeros(6, B, 64, device=device)
    x = torch.empty_like(x0)

    def fwd():
        with torch.no_grad():
            x.copy_(x0)                                           # (Upper_Net transforms its input in place, Q1)
            return up(x, h0, h0, body, R, t)[0]
    fwd()
    torch.cuda.synchronize()
    inner = 20
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fwd()
    ms = _time_events(g.replay, 20, 3) / inner
    flop, byts = 2.15e6 * B * T, 3330.0 * B * T + 1.2e6
    t_roof_us = max(flop / (PEAK_FP32_MFMA_TFLOPS * 1e12), byts / (PEAK_HBM_GBPS * 1e9)) * 1e6
    return {"workload": "Upper_Net eval forward, B=64 T=8 N=128, fp32, HIP-graph replay", "ms_per_forward": ms,
            "frames_per_s": B * T / (ms * 1e-3), "t_roofline_us": t_roof_us, "roofline_frac": t_roof_us / (ms * 1e3),
            "bound": "dependent-launch latency (t_roofline is below one kernel boundary)"}


def call_breakdown(body, iters=3):
    """GPU time per C-ABI entry point of an eagerly launched body (event pairs on the launch stream around EVERY launch):
    -> {name: {"launches": per pass, "us": per pass}} sorted by time.  Eager launches carry 5-10 us of host latency between
    them but the event pair brackets the kernel alone; figures of tiny kernels are upper bounds."""
    from mmego_amd import hip
    rec = {}
    orig = hip.call

    def timed(name, *a):
        if name == "gemm" and hip._gemm_rec is not None:      # deferred into a gemm_group context: the group's launch is what runs (and is counted)
            return orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); orig(name, *a); e1.record()
        rec.setdefault(name, []).append((e0, e1))
    body()
    torch.cuda.synchronize()
    hip.call = timed
    try:
        for _ in range(iters):
            body()
        torch.cuda.synchronize()
    finally:
        hip.call = orig
    out = {n: {"launches": len(v) // iters, "us": sum(a.elapsed_time(b) for a, b in v) / iters * 1e3} for n, v in rec.items()}
    return dict(sorted(out.items(), key=lambda kv: -kv[1]["us"]))


def wlocal_figures(device, cpu_steps=3, with_cpu=True, trace_only=False):
    """The UpperNetwlocal path (Net/Upper_Net.py:406-432: UpperNet plus the anchor / "voxel" branch -- grouping of the 8 nearest
    points around each of 27 anchors, per-group PointNet with softmax pooling, the 3x3x3 Conv3d extractor, a second BiLSTM), which
    no reference trainer constructs: reported separately from the headline.  One training step (forward, L1(sum)
    loss, backward, fused Adam; head pose given, no IMU_Net inside so that the figure is this net's own) as a replayed HIP graph,
    and the eval forward, both at B=64 T=8 N=128 fp32; the CPU oracle's step beside it.
    HBM bytes of the local branch per frame: what its activations move between launches (written once + read once per consumer)."""
    from mmego_amd import nets_local
    from mmego_amd.train_step import StageStep
    torch.manual_seed(77)
    net = nets_local.UpperNetwlocal().to(device).train()
    x, imu_in, body, target = synth_batch(1234, device)
    g = torch.Generator().manual_seed(9)
    Rg = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous().to(device)
    st = StageStep("upper", net, None, lr=3e-5, use_graph=os.environ.get("MMEGO_WLOCAL_EAGER") != "1")   # (eager: PMC passes)
    st.bind(x, imu_in, body, target, R_gt=Rg)
    st.prepare()
    ms_train = _time_events(st.step, 50, 5)
    loss = float(st.loss.item())
    if trace_only:                       # (clean input for rocprofv3: graph replays of the training step only)
        return {"trace_only": True, "train_ms_per_step": ms_train, "train_loss": loss}
    st_e = StageStep("upper", net, None, lr=3e-5, use_graph=False)
    st_e.bind(x, imu_in, body, target, R_gt=Rg)
    calls = call_breakdown(st_e._body)
    n_launch = sum(v["launches"] for v in calls.values())
    # eval forward: 10 forwards in one graph
    net.eval()
    h0 = torch.zeros(6, B, 64, device=device)
    tg = target[:, :, 20].contiguous()
    xw = torch.empty_like(x)

    def fwd():
        with torch.no_grad():
            xw.copy_(x)
            return net(xw, h0, h0, h0, h0, body, Rg, tg)[0]
    fwd()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            fwd()
    ms_eval = _time_events(gr.replay, 20, 3) / 10
    calls_eval = call_breakdown(fwd)
    net.train()
    local = getattr(net, "local_branch_bytes", None)
    res = {"workload": "UpperNetwlocal (Upper_Net + anchor/voxel branch), B=64 T=8 N=128, fp32; train step = forward + L1(sum) + "
                       "backward + fused Adam as one replayed HIP graph (head pose given), eval = forward only",
           "train_ms_per_step": ms_train, "train_frames_per_s": B * T / (ms_train * 1e-3), "train_loss": loss,
           "train_launches_per_step": n_launch,
           "eval_ms_per_forward": ms_eval, "eval_frames_per_s": B * T / (ms_eval * 1e-3),
           "eval_launches_per_forward": sum(v["launches"] for v in calls_eval.values()),
           "train_gpu_us_by_entry_point": {k: v for k, v in list(calls.items())[:14]},
           "eval_gpu_us_by_entry_point": {k: v for k, v in list(calls_eval.items())[:8]}}
    if local is not None:
        res["local_branch_hbm_bytes_per_frame"] = local(B * T, N)
    if with_cpu:
        from oracle import nets as on
        from oracle import train as ot
        from oracle import skeleton as sk
        torch.set_num_threads(host_cores())
        torch.manual_seed(77)
        o = on.UpperNetwlocal().train()
        opt = torch.optim.Adam(o.parameters(), lr=3e-5)
        xc, bc, tc, Rc = x.cpu(), body.cpu(), target.cpu(), Rg.cpu()
        h0c, c0c = ot.zeros_state(B)
        ts = []
        for it in range(cpu_steps + 1):
            t0 = time.perf_counter()
            opt.zero_grad()
            l = o(xc.clone(), h0c, c0c, h0c, c0c, bc, Rc, tc[:, :, 20].contiguous())[0]
            ot.l1_sum(l, tc[:, :, list(sk.UPPER_MAP)]).backward()
            opt.step()
            if it:
                ts.append(time.perf_counter() - t0)
        cpu_ms = sum(ts) / len(ts) * 1e3
        res["cpu_oracle"] = {"train_ms_per_step": cpu_ms, "train_frames_per_s": B * T / (cpu_ms * 1e-3), "cores": host_cores(),
                             "kind": "port", "sample": "%d steps after 1 warm-up, same batch" % cpu_steps}
        res["gpu_over_cpu"] = cpu_ms / ms_train
    del net, st, st_e
    torch.cuda.empty_cache()
    return res


def stage1_step(device):
    """SURVEY 8-f row 3, an extra figure: one stage-1 training step of IMU_Net (Processor/Train/Train_IMU.py:114-149 -- forward,
    geodesic + 100 x position loss, backward through both BiLSTM(512) stacks, Adam with coupled weight decay) at B=64 T=8 as one
    replayed HIP graph.  Algorithmic FLOPs: forward 222.48 MFLOP/frame, backward twice that."""
    from mmego_amd import nets
    from mmego_amd.train_step import ImuStep
    torch.manual_seed(4)
    net = nets.IMUNet(15, 9, 512, 2, True, 0).to(device).train()
    st = ImuStep(net, lr=1e-4, weight_decay=0.001, use_graph=True)
    g = torch.Generator().manual_seed(44)
    imu = torch.randn(B, T, 20, 15, generator=g).to(device)
    Rg = torch.linalg.qr(torch.randn(B, T, 3, 3, generator=g))[0].contiguous().to(device)
    tgt = torch.randn(B, T, 21, 3, generator=g).to(device)
    st.bind(imu, Rg, tgt)
    for _ in range(3):
        st.step()
    ms = _time_events(st.step, 20, 2)
    # the same step with rnn_fast's input-projection and input-gradient products as fp32-accurate piece products on the bf16 matrix
    # pipe (IMUNet.train_precision = "split3", opt-in; test_imu_stage1_gradients_at_full_size holds it to the same bars)
    ms_s3 = None
    try:
        torch.manual_seed(4)
        net3 = nets.IMUNet(15, 9, 512, 2, True, 0).to(device).train()
        net3.train_precision = "split3"
        st3 = ImuStep(net3, lr=1e-4, weight_decay=0.001, use_graph=True)
        st3.bind(imu, Rg, tgt)
        for _ in range(3):
            st3.step()
        ms_s3 = _time_events(st3.step, 20, 2)
        loss_s3 = float(st3.loss.item())
        del net3, st3
    except Exception as e:                                  # (reported, never fatal for the headline)
        ms_s3, loss_s3 = None, repr(e)
    flop = 3.0 * 2.0 * 222.48e6 * B * T
    res = {"workload": "stage-1 IMU_Net training step (forward, geodesic + position loss, backward, Adam with weight decay), "
                       "B=64 T=8 S=20, fp32, one HIP graph per step", "ms_per_step": ms, "frames_per_s": B * T / (ms * 1e-3),
           "algorithmic_tflops": flop / (ms * 1e-3) / 1e12, "frac_of_fp32_mfma_peak": flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
           "loss": float(st.loss.item()),
           "ms_per_step_split3_products": ms_s3, "loss_split3_products": loss_s3}
    del net, st
    torch.cuda.empty_cache()
    return res


def config5_forward(device):
    """BASELINE config 5: B=2048 T=16 N=256, IMU_Net -> Upper_Net -> Lower_Net eval forward with IMU_Net's BiLSTM products in
    bf16-operand / fp32-accumulate mode (everything else fp32).  Reports ms per forward, frames/s, and for the dominant kernel
    (the fused projection + recurrence step, mmego_lstm_step_bf16_fused: lstm_step_bf16_fused256_kernel at this shape) its MFMA and HBM fractions from live event pairs:
      flop per launch   = 2 x ndir x Bn x 4H x (K_in + H)            (K_in + 0 on the first timestep)
      bytes per launch  = ndir x Bn x [2 K_in (x_t bf16) + 2 H (h_t-1 bf16) + 8 H (c read + write) + 2 H (h_t bf16)
                          + 4 H (h_t fp32, last layer only)] + the weights once (2 x 4H x (K_in + H) x 2 B)."""
    from mmego_amd import hip, nets
    Bq, Tq, Nq, H = 2048, 16, 256, 512
    torch.manual_seed(5)
    imu_net = nets.IMUNet(15, 9, H, 2, True, 0.1).to(device).eval()
    imu_net.precision = "bf16"
    up, lo = nets.UpperNet().to(device).eval(), nets.LowerNet(64).to(device).eval()
    up.precision = lo.precision = "bf16"
    x0, _, _, body, imu_in = [v.to(device) for v in _synth_points(Bq, Tq, Nq, 55)]
    h0 = torch.zeros(6, Bq, 64, device=device)

    def fwd():
        with torch.no_grad():
            x = x0.clone()
            R, t = imu_net(imu_in)
            u = up(x, h0, h0, body, R, t)[0]
            return lo(u, x, None, None, None, None, body, R, t)[0]
    ms = _time_events(fwd, 3, 2)
    out = fwd()
    finite = bool(torch.isfinite(out).all())
    rec = []
    orig = hip.call

    def timed(name, *a):
        if name == "lstm_step_bf16_fused":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); orig(name, *a); e1.record()
            rec.append((e0, e1, a))
        else:
            orig(name, *a)
    hip.call = timed
    try:
        fwd()
        torch.cuda.synchronize()
    finally:
        hip.call = orig
    tot_ms = tot_fl = tot_by = 0.0
    for e0, e1, a in rec:
        ndir, Bn, Hh, first, kin = a[0], a[1], a[2], a[3], a[9] + a[14]
        last = a[20] is not None                                  # fp32 output pointer: last layer of a stack only
        tot_ms += e0.elapsed_time(e1)
        tot_fl += 2.0 * ndir * Bn * 4 * Hh * (kin + (0 if first else Hh))
        tot_by += ndir * Bn * (2.0 * kin + (0 if first else 2.0 * Hh) + 8.0 * Hh + 2.0 * Hh + (4.0 * Hh if last else 0.0)) \
            + ndir * 4.0 * Hh * (kin + Hh) * 2.0
    k = {}
    if rec:
        tf, gbps = tot_fl / (tot_ms * 1e-3) / 1e12, tot_by / (tot_ms * 1e-3) / 1e9
        k = {"kernel": "lstm_step_bf16_fused256_kernel (projection folded into the recurrent step; persistent 256x256 tiles)", "launches_per_forward": len(rec),
             "avg_launch_us": tot_ms / len(rec) * 1e3, "share_of_forward": tot_ms / ms,
             "mfma": {"achieved": tf, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_BF16_MFMA_TFLOPS},
             "hbm": {"achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                     "algorithmic_bytes_per_launch_avg": tot_by / len(rec)}}
    # algorithmic FLOPs of the forward (SURVEY 8-d per-frame figures: IMU 444.96, Upper 2.15, Lower 9.26 MFLOP at N=128; the
    # per-point part of Upper/Lower doubles at N=256 and stays < 3 % of the total)
    fl = (444.96e6 + 2.15e6 + 9.26e6) * Bq * Tq
    res = {"workload": "IMU_Net -> Upper_Net -> Lower_Net eval forward, B=2048 T=16 N=256, precision = 'bf16' on all three nets: "
                       "IMU_Net's BiLSTM products, Upper_Net's PointNet / GlobalPointNet stages, Lower_Net's BasePointNet stages and ST-GCN products "
                       "(1x1 and temporal convs) on bf16 operands with fp32 accumulation, everything else fp32", "ms_per_forward": ms, "frames_per_s": Bq * Tq / (ms * 1e-3),
           "algorithmic_tflops": fl / (ms * 1e-3) / 1e12, "frac_of_bf16_mfma_peak": fl / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
           "outputs_finite": finite, "dominant_kernel": k}
    del imu_net, up, lo, x0, imu_in
    torch.cuda.empty_cache()
    return res


def mode_joint_errors(device):
    """A joint-error statement for the precision MODES (config 5's numbers carried none beyond "< 2e-2"): the eval forward
    IMU_Net -> Upper_Net -> Lower_Net at B=64, T=8, N=128 on the HIP path in each mode against the CPU oracle's fp32 forward on the same
    seeded weights and batch: largest joint distance in cm, upper and lower body.  "bf16" = all three nets in the bf16 mode (what
    config 5 times), "split3" = IMU_Net's products as fp32-accurate piece products, "fp32" = the native path (the parity path)."""
    from oracle import nets as on
    from oracle import train as ot
    torch.manual_seed(1234)
    o_imu = on.IMUNet(15, 9, 512, 2, True, 0.1).eval()
    o_up, o_lo = on.UpperNet().eval(), on.LowerNet(64).eval()
    himu, hup, hlo, _ = build_hip_models(device)
    hup.eval(); hlo.eval()
    x, imu_in, body, _ = synth_batch(1234, "cpu")
    h0, c0 = ot.zeros_state(B)
    with torch.no_grad():
        R, t = o_imu(imu_in)
        xo = x.clone()
        up_o = o_up(xo, h0, c0, body, R, t)[0]
        lo_o = o_lo(up_o.clone(), xo, h0, c0, h0, c0, body, R, t)[0]
    res = {"what": "eval forward IMU_Net -> Upper_Net -> Lower_Net, B=64 T=8 N=128, HIP path in each precision mode vs the CPU oracle (fp32), "
                   "same seeded weights: max joint distance in cm (upper body, lower body)"}
    d = lambda v: v.to(device)
    hz = torch.zeros(6, B, 64, device=device)
    for mode in ("fp32", "split3", "bf16"):
        himu.precision = "fp32" if mode == "fp32" else mode
        hup.precision = hlo.precision = "bf16" if mode == "bf16" else "fp32"
        with torch.no_grad():
            xh = d(x.clone())
            Rh, th = himu(d(imu_in))
            up_h = hup(xh, hz, hz, d(body), Rh, th)[0]
            lo_h = hlo(up_h.clone(), xh, None, None, None, None, d(body), Rh, th)[0]
        torch.cuda.synchronize()
        res[mode] = {"upper_cm": (up_h.cpu().view_as(up_o) - up_o).norm(dim=-1).max().item() * 100.0,
                     "lower_cm": (lo_h.cpu().view_as(lo_o) - lo_o).norm(dim=-1).max().item() * 100.0}
    del himu, hup, hlo
    torch.cuda.empty_cache()
    return res


def split3_figures(device, imu, imu_in, out, with_parity=True):
    """The split3 mode's own block of the detail line: the IMU_Net forward in both modes (replayed graphs, one launch per timestep),
    the two kernel families' rooflines from event pairs around an eager forward -- `achieved` = bf16 MFMA flops actually issued
    (6 x the algorithmic product flops) / time against the 2.5 PF dense bf16 peak, `fp32_equivalent_tflops` = algorithmic flops / time
    against nothing (the fp32 pipe's peak is 157.3) -- and the parity of one U+L step at the timed shape with the mode on, at the
    fp32 path's bars (bench.ul_step_parity)."""
    from mmego_amd import blocks
    res = {"what": "IMUNet.precision = 'split3': rnn_fast's products (input projections + recurrent steps, 94 %% of the step's FLOPs) "
                   "on exactly split bf16 operands a = a1 + a2 + a3, %d piece products each, fp32 accumulation (csrc/split3.hip); "
                   "everything else is the fp32 path" % blocks.SPLIT3_NPROD, "nprod": blocks.SPLIT3_NPROD}
    fwd_ms = {}
    was = imu.precision
    try:
        for prec in ("fp32", "split3"):
            imu.precision = prec

            def fwd():
                with torch.no_grad(), blocks.two_chains(False):
                    return imu(imu_in)
            fwd()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(4):
                    fwd()
            fwd_ms[prec] = _time_events(g.replay, 10, 3) / 4
        res["imu_forward_ms"] = {"fp32_one_launch_per_timestep": fwd_ms["fp32"], "split3": fwd_ms["split3"]}
        imu.precision = "split3"

        def eager():
            with torch.no_grad(), blocks.two_chains(False):
                imu(imu_in)
        eager()
        torch.cuda.synchronize()
        rec, iters = profile_kernels(eager, ("split3_gemm", "split3_step"))
    finally:
        imu.precision = was
    npr = blocks.SPLIT3_NPROD
    gem = [(ms_, 2.0 * 32 * a[6] * 32 * a[7] * a[8]) for ms_, a in rec["split3_gemm"]]            # args: A, W, Cf, C, ldc, bias, Mrb, Nrb, K
    stp = [(ms_, 0.0 if a[3] else 2.0 * a[0] * a[1] * 4 * a[2] * a[2]) for ms_, a in rec["split3_step"]]   # args: ndir, Bn, H, first
    fam = {}
    for name, v in (("s3_gemm_big_kernel (rnn_fast input projections, both directions: 10240 x 4096 x {512, 1024}; 320 x 256 tiles, LDS-DMA)", gem),
                    ("s3_step_kernel (rnn_fast recurrent steps, both directions per launch: 2 x 512 x 2048 x 512)", stp)):
        if not v:
            continue
        tot_ms, tot_fl = sum(m for m, _ in v), sum(f for _, f in v)
        fam[name] = {"avg_launch_us": tot_ms / len(v) * 1e3, "launches_per_forward": len(v) // iters, "ms_per_forward": tot_ms / iters,
                     "achieved": npr * tot_fl / (tot_ms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "bound": "mfma",
                     "frac": npr * tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                     "fp32_equivalent_tflops": tot_fl / (tot_ms * 1e-3) / 1e12}
        tr, src = pmc_traffic(name)
        fam[name]["traffic"], fam[name]["traffic_source"] = tr, src
    res["kernels"] = fam
    if fam:
        k0 = max(fam, key=lambda k: fam[k]["ms_per_forward"])
        res["roofline"] = dict(fam[k0], kernel=k0.split(" ")[0],
                               achieved_is="%d x algorithmic product flops (the bf16 MFMAs issued) / event-pair time" % npr)
    if with_parity:
        par = ul_step_parity(device, use_graph=True, imu_precision="split3")
        del par["tensors"]
        res["parity"] = par
    res["summary"] = {"ms_guarded": round(out["ms_split3_guarded"], 4), "ms_sequential": round(out["ms_per_step_split3_sequential"], 4),
                      "imu_fwd_ms": [round(fwd_ms["fp32"], 4), round(fwd_ms["split3"], 4)]}
    if "roofline" in res:
        res["summary"]["roofline"] = {"kernel": res["roofline"]["kernel"], "frac_of_2.5PF": round(res["roofline"]["frac"], 3),
                                      "fp32_equiv_TF": round(res["roofline"]["fp32_equivalent_tflops"], 1)}
    if "parity" in res:
        res["summary"]["parity_cm"] = [float("%.3g" % res["parity"]["upper_cm"]), float("%.3g" % res["parity"]["lower_cm"])]
    return res


def emit(out):
    """Everything measured goes out as ONE comment line (`# detail: {...}`, also written to gpurun_out/bench_detail.json when that
    directory can be written), THEN the contract's JSON line -- kept under 2 KB, so that a reader who only has the tail of stdout
    still holds every headline figure (the sequential figure had fallen off the driver's 8-KB tail)."""
    detail = json.dumps(out)
    print("# detail: " + detail)
    try:
        d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "bench_detail.json"), "w") as f:
            f.write(detail + "\n")
    except OSError:
        pass
    r5 = lambda v: round(v, 5) if isinstance(v, float) else v
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data")
    line = {k: r5(out[k]) for k in keep}
    cfg = out["config"]
    line["config"] = {"workload": "U+L step (Train_Upper + Train_Lower train_once bodies), B=64/GPU T=8 N=128, fp32, dropout live, "
                                  + ("stages one after the other" if not cfg["stages_concurrent"] else "two concurrent HIP-graph branches"),
                      "global_batch": cfg["global_batch"], "seq_len": cfg["seq_len"], "points": cfg["points"],
                      "parallelism": cfg["parallelism"]}
    rf = out["roofline"]
    line["roofline"] = {k: r5(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    line["roofline"]["kernel"] = rf["kernel"].split(" ")[0]
    line["roofline"]["avg_launch_us"] = r5(rf["avg_launch_us"])
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {k: r5(cb[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
        if len(str(line["cpu_baseline"].get("sample", ""))) > 160:
            line["cpu_baseline"]["sample"] = str(line["cpu_baseline"]["sample"])[:157] + "..."
    ex = {"ms_min_median_max": [r5(out["ms_per_step_min"]), r5(out["ms_per_step_median"]), r5(out["ms_per_step_max"])],
          "ms_sequential": r5(out["ms_per_step_sequential"]), "frames_per_s_sequential": r5(out["frames_per_s_sequential"]),
          "t_upper_ms": r5(out["t_upper_ms"]), "t_lower_ms": r5(out["t_lower_ms"]),
          "ms_imu_shared": r5(out["ms_per_step_imu_shared"]), "roofline_step_frac": r5(out["roofline_step"]["frac"])}
    for k_out, k_in in (("ms_pipelined", "ms_per_step_pipelined"), ("ms_bf16_imu", "ms_per_step_bf16_imu"),
                        ("ms_split3_guarded", "ms_split3_guarded"), ("ms_split3_unguarded", "ms_per_step_split3_unguarded"),
                        ("ms_pipelined_split3", "ms_per_step_pipelined_split3")):
        if k_in in out:
            ex[k_out] = r5(out[k_in])
    if "roofline_by_kernel_time" in out:
        ex["by_kernel_time"] = {"kernel": "lstm_step_dma_kernel", "frac": r5(out["roofline_by_kernel_time"]["frac"])}
    if "split3" in out:
        ex["split3"] = out["split3"].get("summary")
    if "parity" in out and isinstance(out["parity"], dict):
        ex["parity_cm"] = {k[:-3]: float("%.3g" % out["parity"][k]) for k in ("upper_cm", "lower_cm", "tolerance_cm") if k in out["parity"]}
    for k, sub in (("config2_ms", ("config2", "ms_per_forward")), ("config5_ms", ("config5", "ms_per_forward")),
                   ("stage1_ms", ("stage1", "ms_per_step")), ("stage1_split3_ms", ("stage1", "ms_per_step_split3_products")), ("wlocal_train_ms", ("wlocal", "train_ms_per_step")),
                   ("wlocal_eval_ms", ("wlocal", "eval_ms_per_forward"))):
        v = out.get(sub[0])
        if isinstance(v, dict) and sub[1] in v:
            ex[k] = r5(v[sub[1]])
    if "gpu_over_cpu" in out:
        ex["gpu_over_cpu"] = round(out["gpu_over_cpu"], 1)
    line["extra"] = ex
    line["detail"] = "the '# detail:' line above / gpurun_out/bench_detail.json"
    text = json.dumps(line)
    if len(text) > 2040:                                       # never let the headline line outgrow a tail
        line.pop("extra")
        text = json.dumps(line)
    sys.stdout.flush()
    print(text)
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed U+L steps (default: > 1 s of timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=10, help="extra timed blocks of --steps steps for ms_per_step_min/median/max")
    ap.add_argument("--no-wlocal", action="store_true", help="skip the UpperNetwlocal figures")
    ap.add_argument("--wlocal-only", action="store_true", help="only the UpperNetwlocal figures (input for rocprofv3 summaries of that path)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying HIP graphs")
    ap.add_argument("--sequential", action="store_true", help="run the Upper and Lower bodies one after the other")
    ap.add_argument("--trace-only", action="store_true", help="warm-up + timed loop only, then exit (clean input for rocprofv3 summaries)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-variant", action="store_true", help="skip the extra bf16-IMU figure")
    ap.add_argument("--no-pipelined-variant", action="store_true", help="skip the extra prefetch-pipelined figure")
    ap.add_argument("--no-split3-variant", action="store_true", help="skip the split3 (fp32-accurate bf16-pipe IMU_Net) figures")
    ap.add_argument("--no-config-extras", action="store_true", help="skip the config-2 / config-5 forward figures")
    ap.add_argument("--cpu-steps", type=int, default=12, help="timed U+L steps of the CPU baseline (~0.85 s each on 16 cores: ~10 s)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    local_rank %= torch.cuda.device_count()         # (rehearsals on a one-GPU box put every rank on cuda:0)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = None
    # MMEGO_FORCE_DIST=1: take the multi-rank code path (RCCL process group, barriers, MAX over ranks, warm-up collectives) with
    # ONE rank as well -- the one-GPU box allows one rank per device, so this is how that path is rehearsed there
    dist_on = world > 1 or os.environ.get("MMEGO_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("MMEGO_DIST_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    from mmego_amd import hip
    from mmego_amd.train_step import ConcurrentStages, StageStep
    hip.lib()
    if args.wlocal_only:
        print(json.dumps({"wlocal": wlocal_figures(device, with_cpu=not args.no_cpu_baseline, trace_only=args.trace_only)}))
        return
    imu, upper, lower, upper_frozen = build_hip_models(device)
    imu_l = clone_imu(imu, device)
    x, imu_in, body, target = synth_batch(1234 + rank, device)      # weak scaling: every rank its own B=64 shard
    su = StageStep("upper", upper, imu, lr=3e-5, process_group=pg, use_graph=not args.no_graph)
    sl = StageStep("lower", lower, imu_l, upper_frozen=upper_frozen, lr=3e-5, process_group=pg, use_graph=not args.no_graph)
    su.bind(x, imu_in, body, target)
    sl.bind(x, imu_in, body, target)
    both = ConcurrentStages([su, sl], use_graph=not args.no_graph)

    def ul_step():
        if args.sequential:
            su.step()
            sl.step()
        else:
            both.step()

    if dist_on:       # bring the RCCL communicator (and its channels for these message sizes) up outside the timed region
        torch.distributed.all_reduce(torch.zeros(1, device=device))
        for net in (upper, lower):      # the gradient buffers are overwritten by every backward: reducing them here is harmless
            torch.distributed.all_reduce(net.flat().flat_g)
        torch.cuda.synchronize()

    def sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.sequential:                 # graph capture happens here, outside the timed region, whatever --warmup is
        su.prepare(); sl.prepare()
    else:
        both.prepare()
    # clock ramp: a GPU that has just been idle (process start-up, graph capture on the host) runs its first steps several per cent
    # slower -- untimed burn-in in front of the W warm-up steps, so that a short --steps block measures the steady state.  (40 steps
    # = 0.2 s were not always enough: in about one run in four the first timed block still came out 3 % above the ten blocks behind
    # it -- 5.16 against 5.01-5.03 ms; 1 s of burn-in.)
    for _ in range(0 if args.trace_only else 200):
        ul_step()
    for _ in range(args.warmup):
        ul_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ul_step()
    sync()
    dt = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = tt.item()
    # spread of the headline: the same K-step block timed REPEATS more times (the contract's block above stays `ms_per_step`)
    rep_dt = [dt]
    if not args.trace_only:
        for _ in range(args.repeats):
            sync()
            t0r = time.perf_counter()
            for _ in range(args.steps):
                ul_step()
            sync()
            rep_dt.append(time.perf_counter() - t0r)
        if dist_on:
            tt = torch.tensor(rep_dt, dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            rep_dt = tt.tolist()
            rep_dt[0] = dt
    rep_ms = sorted(d / args.steps * 1e3 for d in rep_dt)
    loss_u, loss_l = su.loss.item(), sl.loss.item()
    if args.trace_only:
        if rank == 0:
            print(json.dumps({"trace_only": True, "ms_per_step": dt / args.steps * 1e3, "stages_concurrent": not args.sequential}))
        if dist_on:
            torch.distributed.destroy_process_group()
        return

    # per-stage split, stages one after the other (device time, events on the launch stream).  The steps run BACK TO BACK -- events
    # recorded around every stage, one host synchronisation behind the last -- and the medians are reported: with a host
    # synchronisation behind every step (r03: median of 5 isolated steps) the same code measured 5.67 and 5.84 ms in two runs on one
    # box (the GPU idles between isolated steps and its clock follows)
    su.step(); sl.step()                                 # (the first pass captures the per-stage graphs)
    torch.cuda.synchronize()
    evs = []
    for i in range(44):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record(); su.step(); ev[1].record(); sl.step(); ev[2].record()
        evs.append(ev)
    torch.cuda.synchronize()
    tu_s = [ev[0].elapsed_time(ev[1]) for ev in evs[4:]]
    tl_s = [ev[1].elapsed_time(ev[2]) for ev in evs[4:]]
    t_u, t_l = sorted(tu_s)[len(tu_s) // 2], sorted(tl_s)[len(tl_s) // 2]

    # "IMU-shared" variant: one IMU_Net forward per minibatch feeds both bodies.  Extra figure, not `value`.
    from mmego_amd.train_step import SharedImuStages
    su_s = StageStep("upper", upper, None, lr=3e-5, process_group=pg, use_graph=False)
    sl_s = StageStep("lower", lower, None, upper_frozen=upper_frozen, lr=3e-5, process_group=pg, use_graph=False)
    shared = SharedImuStages(imu, [su_s, sl_s], imu_in, use_graph=not args.no_graph)
    su_s.bind(x, imu_in, body, target)
    sl_s.bind(x, imu_in, body, target)
    for _ in range(3):
        shared.step()
    sync()
    t0s = time.perf_counter()
    for _ in range(args.steps):
        shared.step()
    sync()
    dt_shared = time.perf_counter() - t0s
    if dist_on:
        tt = torch.tensor([dt_shared], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt_shared = tt.item()

    # bf16-operand variant of the two frozen IMU_Net forwards (IMUNet.precision = "bf16": BiLSTM products with bf16 operands
    # and fp32 accumulation, everything else fp32).  Extra figure, never `value`: it is outside the 1e-3 cm parity bound.
    bf16_extra = {}
    if not args.no_bf16_variant and world == 1:      # single-GPU extra; the scaling runs keep to the fp32 path
        try:
            imu.precision = imu_l.precision = "bf16"
            su_b = StageStep("upper", upper, imu, lr=3e-5, process_group=pg, use_graph=not args.no_graph)
            sl_b = StageStep("lower", lower, imu_l, upper_frozen=upper_frozen, lr=3e-5, process_group=pg, use_graph=not args.no_graph)
            su_b.bind(x, imu_in, body, target)
            sl_b.bind(x, imu_in, body, target)
            def time_both(unguarded):
                both_b = ConcurrentStages([su_b, sl_b], use_graph=not args.no_graph, unguarded=unguarded)
                both_b.prepare()
                for _ in range(3):
                    both_b.step()
                sync()
                t0b = time.perf_counter()
                for _ in range(args.steps):
                    both_b.step()
                sync()
                return time.perf_counter() - t0b
            # the product engine runs a step with bf16-MFMA kernels as ONE chain (train_step.needs_exclusive);
            # `unguarded` = the two-branch form of r02-r05, kept as a comparison figure
            dt_b, dt_bu = time_both(False), time_both(True)
            bf16_extra = {"ms_per_step_bf16_imu": dt_b / args.steps * 1e3,
                          "frames_per_s_bf16_imu": world * B * T / (dt_b / args.steps),
                          "ms_per_step_bf16_imu_unguarded": dt_bu / args.steps * 1e3}
        finally:
            imu.precision = imu_l.precision = "fp32"

    # split3 variant (IMUNet.precision = "split3", csrc/split3.hip): the SAME fp32 step with rnn_fast's products of both frozen
    # IMU_Net forwards on exactly split bf16 operands (six piece products, fp32 accumulation): fp32-accurate -- it passes the fp32
    # path's parity bars (`split3.parity` below, tests/test_split3_gpu.py) -- but NOT `value` this round.
    split3_extra = {}
    if not args.no_split3_variant and world == 1:
        try:
            imu.precision = imu_l.precision = "split3"
            su_3 = StageStep("upper", upper, imu, lr=3e-5, process_group=pg, use_graph=not args.no_graph)
            sl_3 = StageStep("lower", lower, imu_l, upper_frozen=upper_frozen, lr=3e-5, process_group=pg, use_graph=not args.no_graph)
            su_3.bind(x, imu_in, body, target)
            sl_3.bind(x, imu_in, body, target)
            def time_both3(unguarded):
                both_3 = ConcurrentStages([su_3, sl_3], use_graph=not args.no_graph, unguarded=unguarded)
                both_3.prepare()
                for _ in range(20):
                    both_3.step()
                sync()
                t03 = time.perf_counter()
                for _ in range(args.steps):
                    both_3.step()
                sync()
                return time.perf_counter() - t03
            # guarded (what the engines do since r06): the step as ONE chain, nothing resident beside a bf16-MFMA workgroup;
            # unguarded: the two concurrent branches of r05 (safe by the build -- no packed-fp32 instruction -- but not by construction)
            dt_3, dt_3u = time_both3(False), time_both3(True)
            su_3.step(); sl_3.step()
            torch.cuda.synchronize()
            ev3 = []
            for i in range(24):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(); su_3.step(); ev[1].record(); sl_3.step(); ev[2].record()
                ev3.append(ev)
            torch.cuda.synchronize()
            tu3 = sorted(ev[0].elapsed_time(ev[1]) for ev in ev3[4:])
            tl3 = sorted(ev[1].elapsed_time(ev[2]) for ev in ev3[4:])
            split3_extra = {"ms_split3_guarded": dt_3 / args.steps * 1e3, "frames_per_s_split3_guarded": world * B * T / (dt_3 / args.steps),
                            "ms_per_step_split3": dt_3 / args.steps * 1e3, "frames_per_s_split3": world * B * T / (dt_3 / args.steps),
                            "ms_per_step_split3_unguarded": dt_3u / args.steps * 1e3,
                            "ms_per_step_split3_sequential": tu3[len(tu3) // 2] + tl3[len(tl3) // 2],
                            "t_upper_ms_split3": tu3[len(tu3) // 2], "t_lower_ms_split3": tl3[len(tl3) // 2]}
        finally:
            imu.precision = imu_l.precision = "fp32"

    # Prefetch-pipelined variant (train_step.PipelinedStages): the frozen IMU_Net forwards of minibatch i+1 overlap the trainable
    # bodies of minibatch i; two DIFFERENT synthetic minibatches alternate and are copied into the static buffers inside the timed
    # loop.  Same work per step, bit-identical results (tests/test_hip_local.py); extra figure, never `value`.
    pipe_extra = {}
    if not args.no_pipelined_variant and world == 1:
        from mmego_amd.train_step import PipelinedStages
        x2, imu2, _, target2 = synth_batch(4321 + rank, device)
        sets = [(x, target, imu_in), (x2, target2, imu2)]
        xb, tb, inext = x.clone(), target.clone(), imu_in.clone()
        su_p = StageStep("upper", upper, None, lr=3e-5, process_group=pg, use_graph=False)
        sl_p = StageStep("lower", lower, None, upper_frozen=upper_frozen, lr=3e-5, process_group=pg, use_graph=False)
        pipe = PipelinedStages([su_p, sl_p], [imu, imu_l], inext, use_graph=not args.no_graph)
        su_p.bind(xb, imu_in, body, tb)
        sl_p.bind(xb, imu_in, body, tb)
        pipe.prime()
        pipe.prepare()

        def pipe_step(i):
            xs_, ts_, _ = sets[i % 2]
            xb.copy_(xs_); tb.copy_(ts_); inext.copy_(sets[(i + 1) % 2][2])
            pipe.step()
        for i in range(4):
            pipe_step(i)
        sync()
        t0p = time.perf_counter()
        for i in range(args.steps):
            pipe_step(i)
        sync()
        dt_p = time.perf_counter() - t0p
        pipe_extra = {"ms_per_step_pipelined": dt_p / args.steps * 1e3, "frames_per_s_pipelined": world * B * T / (dt_p / args.steps)}
        if not args.no_split3_variant:
            # the same engine with both frozen forwards in the split3 mode (`main.py --train --imu_precision split3`)
            try:
                imu.precision = imu_l.precision = "split3"
                def time_pipe3(unguarded):
                    pipe3 = PipelinedStages([su_p, sl_p], [imu, imu_l], inext, use_graph=not args.no_graph, unguarded=unguarded)
                    su_p.bind(xb, imu_in, body, tb)
                    sl_p.bind(xb, imu_in, body, tb)
                    pipe3.prime()
                    pipe3.prepare()

                    def pipe3_step(i):
                        xs_, ts_, _ = sets[i % 2]
                        xb.copy_(xs_); tb.copy_(ts_); inext.copy_(sets[(i + 1) % 2][2])
                        pipe3.step()
                    for i in range(4):
                        pipe3_step(i)
                    sync()
                    t0p = time.perf_counter()
                    for i in range(args.steps):
                        pipe3_step(i)
                    sync()
                    return time.perf_counter() - t0p
                dt_p3, dt_p3u = time_pipe3(False), time_pipe3(True)
                pipe_extra["ms_per_step_pipelined_split3"] = dt_p3 / args.steps * 1e3          # (guarded: one chain)
                pipe_extra["frames_per_s_pipelined_split3"] = world * B * T / (dt_p3 / args.steps)
                pipe_extra["ms_per_step_pipelined_split3_unguarded"] = dt_p3u / args.steps * 1e3
            finally:
                imu.precision = imu_l.precision = "fp32"

    out = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {"metric": "train frames/sec (Upper+Lower U+L step, B=64/GPU, T=8, 128 pts)", "value": world * B * T / (dt / args.steps),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
               "ms_per_step_min": rep_ms[0], "ms_per_step_median": rep_ms[len(rep_ms) // 2], "ms_per_step_max": rep_ms[-1],
               "timed_blocks": len(rep_ms),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "U+L step = Train_Upper.train_once body + Train_Lower.train_once body "
                                      "(IMU_Net fwd in both, frozen Upper fwd in the Lower body), per-GPU B=64 T=8 N=128, "
                                      "21 joints, Adam lr 3e-5, LSTM dropout 0.1 active; the two bodies (independent programs "
                                      "in the reference) run " + ("one after the other" if args.sequential else
                                                                  "as concurrent branches of one HIP graph"),
                          "global_batch": world * B, "seq_len": T, "points": N, "parallelism": "dp%d" % world,
                          "hip_graph": not args.no_graph, "stages_concurrent": not args.sequential},
               "value_is": ("B*T / ms_per_step of the arrangement named in config.workload (%s); SURVEY 8-d's formula "
                            "B*T / (t_upper + t_lower) is `frames_per_s_sequential`" %
                            ("stages one after the other" if args.sequential else "the two bodies as concurrent branches of one HIP graph")),
               "ms_per_step_imu_shared": dt_shared / args.steps * 1e3,
               "frames_per_s_imu_shared": world * B * T / (dt_shared / args.steps),
               "ms_per_step_sequential": t_u + t_l, "frames_per_s_sequential": world * B * T / ((t_u + t_l) * 1e-3),
               "t_upper_ms": t_u, "t_lower_ms": t_l, "loss_upper": loss_u, "loss_lower": loss_l}
        out.update(bf16_extra)
        out.update(split3_extra)
        out.update(pipe_extra)

    # ---- roofline of the dominant kernel: eager replay with event pairs around every launch ----------------
    if rank == 0:
        su_e = StageStep("upper", upper, imu, lr=3e-5, use_graph=False)
        sl_e = StageStep("lower", lower, imu, upper_frozen=upper_frozen, lr=3e-5, use_graph=False)
        su_e.bind(x, imu_in, body, target)
        sl_e.bind(x, imu_in, body, target)

        from mmego_amd import blocks as _blocks

        def eager():
            # the same recurrence forms as the timed arrangement (train_step.ConcurrentStages): both IMU_Net forwards run their
            # rnn_fast recurrences as two single-direction chains.
            with _blocks.two_chains(True):
                su_e._body()
            with _blocks.two_chains(True):
                sl_e._body()
        eager()
        torch.cuda.synchronize()
        rec, iters = profile_kernels(eager, ("lstm_step", "gemm", "lstm_seq_xcd"))
        # lstm_step: args = (ndir, Bn, H, first, ...).  Algorithmic flops per launch = 2 * ndir * Bn * 4H * H (0 for the
        # first timestep of a sequence, whose h_{t-1} = 0 product is skipped).  ALL launches of the kernel family are
        # averaged, so avg_launch_us is directly comparable with rocprofv3's per-kernel AverageNs.
        def step_flops(a):
            return 0.0 if a[3] else 2.0 * a[0] * a[1] * 4 * a[2] * a[2]
        big = [(ms_, step_flops(a)) for ms_, a in rec["lstm_step"] if a[1] >= 128 and a[0] == 2]
        big16 = [(ms_, step_flops(a)) for ms_, a in rec["lstm_step"] if a[1] >= 128 and a[0] == 1]
        small = [(ms_, step_flops(a)) for ms_, a in rec["lstm_step"] if a[1] < 128]

        def is_nt_aligned(a):   # the dispatch rule of mmego_gemm (gemm.hip / gemm_tile.hip); a[13] = nbatch
            return (a[20] == 1 and not a[18] and a[2] == 1 and a[4] == 1 and a[8] == 1 and a[10] % 64 == 0
                    and a[11] % 64 == 0 and a[12] % 64 == 0)

        def is_tile128(a):
            return is_nt_aligned(a) and a[10] % 128 == 0 and a[11] % 128 == 0 and (a[10] // 128) * (a[11] // 128) * a[13] >= 192
        g128 = [(ms_, 2.0 * a[10] * a[11] * a[12] * a[13]) for ms_, a in rec["gemm"] if is_tile128(a)]
        g64 = [(ms_, 2.0 * a[10] * a[11] * a[12] * a[13]) for ms_, a in rec["gemm"] if is_nt_aligned(a) and not is_tile128(a)]
        cands = {}
        if big:
            cands["lstm_step_dma_kernel<32> (IMU_Net rnn_fast recurrent steps, both directions per launch: 2 x 512 rows x 2048 gates x "
                  "K=512)"] = big
        if big16:
            cands["lstm_step_dma_kernel<16> (IMU_Net rnn_fast recurrent steps, ONE direction per launch: 512 rows x 2048 gates x K=512; "
                  "the two directions' launches run concurrently as two chains, so a launch's duration is NOT its share of the "
                  "step: see recurrence_graph)"] = big16
        if small:
            cands["lstm_step_small_kernel (IMU_Net rnn_slow recurrent steps: 2 dirs x 64 rows x 2048 gates x K=512)"] = small
        # rnn_slow's recurrence as one persistent launch per layer (lstm_seq.hip): args = (..., Bn, H, T) at 10..12; T - 1 product steps
        seqx = [(ms_, 2.0 * 2 * a[10] * 4 * a[11] * a[11] * (a[12] - 1)) for ms_, a in rec.get("lstm_seq_xcd", [])]
        if seqx:
            cands["lstm_seq_xcd_kernel (IMU_Net rnn_slow: a layer's whole recurrence per launch, 2 dirs x 64 rows x 2048 gates x K=512 x "
                  "(T-1) steps, weights stationary)"] = seqx
        def is_tile_big(a):      # gemm_tile.hip's dispatch: whole rounds of 320 x 256 tiles (r06); MMEGO_GEMM_BIG=0 keeps the 128 x 128 walk
            return (os.environ.get("MMEGO_GEMM_BIG", "1") != "0" and is_nt_aligned(a) and a[10] % 320 == 0 and a[11] % 256 == 0
                    and ((a[10] // 320) * (a[11] // 256) * a[13]) % 256 == 0 and not a[17])
        gbig = [(ms_, fl) for (ms_, fl), (_, a) in zip(g128, [r_ for r_ in rec["gemm"] if is_tile128(r_[1])]) if is_tile_big(a)]
        if gbig and len(gbig) == len(g128):
            cands["gemm_tile_big_kernel (320x256 tiles, operands by LDS-DMA, one workgroup per CU; IMU_Net LSTM input projections, both directions per launch: "
                  "2 x 10240 x 2048 x {512,1024} = 512 tiles = two rounds)"] = g128
        elif g128:
            cands["gemm_tile_persistent_kernel (128x128 tiles; IMU_Net LSTM input projections, both directions per launch: 2 x 10240 x 2048 x {512,1024})"] = g128
        if g64:
            cands["gemm_tile_kernel<64,64> (smaller 64-aligned products)"] = g64
        # dominant kernel = largest CRITICAL-PATH time over the eager replay; no tie rule.  A family's critical-path time is the
        # sum of its launch durations, except that the single-direction recurrent steps (<16>) run as two concurrent chains: two
        # of their launches share one slot of the dependent chain, so they count half.  The rnn_fast recurrent steps enter the
        # contest as ONE family (<32> launches + <16> launches).  Every candidate is also reported on its own in `roofline_kernels`.
        crit = {k: sum(m for m, _ in v) * (0.5 if k.startswith("lstm_step_dma_kernel<16>") else 1.0) for k, v in cands.items()}
        fam_step = [k for k in cands if k.startswith("lstm_step_dma_kernel")]
        contest = {k: c for k, c in crit.items() if k not in fam_step}
        if fam_step:
            contest["lstm_step_dma_kernel (IMU_Net rnn_fast recurrent steps, 512 rows x 2048 gates x K=512 per direction; <32>: both "
                    "directions per launch, <16>: one direction per launch as two concurrent chains, counted at half their duration)"] = \
                sum(crit[k] for k in fam_step)
        best_k = max(contest, key=contest.get)
        rgraph = recurrence_graph(device)
        fam_two = rgraph["two_chains"] if big16 else rgraph["one_launch_per_timestep"]
        fam_what = ("replayed-graph recurrence (recurrence_graph.%s): flop of a layer's T-1 product steps / their time -- the two "
                    "directions' launches overlap, so the family's rate is taken from the pair, not from halved eager durations"
                    % ("two_chains" if big16 else "one_launch_per_timestep"))
        if best_k in cands:
            best = (best_k, cands[best_k])
        else:       # the recurrent-step family: launches as measured (ranking used half durations for the concurrent chains)
            best = (best_k, [(m, f) for k in fam_step for m, f in cands[k]])
        tot_ms = sum(m for m, _ in best[1])
        tot_fl = sum(f for _, f in best[1])
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        if best_k not in cands:
            ach = fam_two["tflops"]                  # (not FLOPs over halved event-pair time)
        traffic, traffic_src = pmc_traffic(best[0])
        out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": best[0],
                           "avg_launch_us": tot_ms / len(best[1]) * 1e3, "launches_per_step": len(best[1]) // iters,
                           "flop_per_launch_avg": tot_fl / len(best[1]),
                           "share_of_step": (tot_ms / iters) / (t_u + t_l),
                           "contest_ms_per_step": {k.split(" ")[0]: c / iters for k, c in contest.items()},
                           "contest": "largest critical-path time per step over the eager replay (see roofline_kernels for every family)"}
        # the WHOLE step against the same roofline: algorithmic FLOPs of the literal U+L step / measured step time / peak
        step_flop = STEP_MFLOP_PER_FRAME * 1e6 * B * T
        out["roofline_step"] = {"bound": "mfma", "algorithmic_gflop_per_step": step_flop / 1e9,
                                "achieved": step_flop / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "frac": step_flop / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                "frac_sequential": step_flop / ((t_u + t_l) * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                "what": "SURVEY 8-d: 926.3 MFLOP/frame (IMU_Net forward in both bodies) x %d frames / ms_per_step / "
                                        "157.3 TFLOP/s; frac_sequential uses t_upper + t_lower" % (B * T)}
        # The step against the HBM roofline (SURVEY 8-d figures): 129.1 MB of parameter / gradient / optimiser traffic per
        # U+L step + 4 830 B of streaming I/O per frame.  The path is matrix-pipe and latency bound, not HBM bound.
        alg_bytes = 129.1e6 + 4830.0 * B * T
        out["hbm_roofline"] = {"algorithmic_bytes_per_step": alg_bytes, "peak_GBps": PEAK_HBM_GBPS,
                               "achieved_GBps": alg_bytes / (out["ms_per_step"] * 1e-3) / 1e9,
                               "frac": alg_bytes / (out["ms_per_step"] * 1e-3) / 8e12}
        out["roofline_kernels"] = {k: {"avg_us": sum(m for m, _ in v) / len(v) * 1e3, "launches_per_step": len(v) // iters,
                                       "tflops": sum(f for _, f in v) / (sum(m for m, _ in v) * 1e-3) / 1e12,
                                       "frac": sum(f for _, f in v) / (sum(m for m, _ in v) * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                       "ms_per_step": sum(m for m, _ in v) / iters,
                                       "share_of_step": (sum(m for m, _ in v) / iters) / (t_u + t_l)} for k, v in cands.items()}
        out["recurrence_graph"] = rgraph
        # by summed KERNEL time (what a rocprofv3 --stats table ranks by) the recurrent-step family leads, not the projection
        # product: its effective rate is the pair's (two direction chains overlap), from the replayed-graph measurement
        if fam_step:
            fam_launch = [(m, f) for k in fam_step for m, f in cands[k]]
            fam_ms = sum(m for m, _ in fam_launch)
            by_time = {k: sum(m for m, _ in v) for k, v in cands.items() if k not in fam_step}
            by_time["lstm_step_dma_kernel family"] = fam_ms
            tr_f, tr_src = pmc_traffic("lstm_step_dma_kernel<16>")
            out["roofline_by_kernel_time"] = {
                "bound": "mfma", "kernel": "lstm_step_dma_kernel<16>/<32> (IMU_Net rnn_fast recurrent steps)",
                "achieved": fam_two["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": fam_two["frac"],
                "achieved_is": fam_what, "us_per_timestep_pair": fam_two["us_per_timestep"],
                "per_launch_alone": {"avg_launch_us": fam_ms / len(fam_launch) * 1e3,
                                     "frac": sum(f for _, f in fam_launch) / (fam_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
                "launches_per_step": len(fam_launch) // iters, "kernel_ms_per_step": fam_ms / iters,
                "traffic": tr_f, "traffic_source": tr_src,
                "kernel_ms_per_step_by_family": {k.split(" ")[0]: v / iters for k, v in by_time.items()}}
        # the persistent rnn_slow launches (lstm_seq.hip) bound their spins: a launch that could not make progress flags it
        from mmego_amd import blocks as _blocks
        out["persistent_launch_errors"] = _blocks.seq_xcd_errors()
        if out["persistent_launch_errors"]:
            raise RuntimeError("mmego_lstm_seq_xcd: %d launch(es) ran out of their bounded spin -- results invalid" % out["persistent_launch_errors"])
        sys.stderr.write("[bench] gpu part done: %.1f frames/s; timing the CPU oracle on %d threads\n" % (out["value"], host_cores()))
        sys.stderr.flush()
        if world == 1 and split3_extra:
            out["split3"] = split3_figures(device, imu, imu_in, out, with_parity=not args.no_cpu_baseline)
        if world == 1 and not args.no_config_extras:
            out["config2"] = config2_forward(device)
            out["config5"] = config5_forward(device)
            if not args.no_cpu_baseline:           # (uses the oracle as the checker, like `parity`)
                out["config5"]["joint_error_vs_oracle"] = mode_joint_errors(device)
            out["stage1"] = stage1_step(device)
        if world == 1 and not args.no_wlocal:
            out["wlocal"] = wlocal_figures(device, with_cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity"] = cpu_baseline(args.cpu_steps, 1, device)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        emit(out)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
