"""Fused train_once bodies of stage 2 (Upper_Net) and stage 3 (Lower_Net) on the HIP path.

Mirrors reference Processor/Train/Train_Upper.py:134-187 and Train_Lower.py:155-230 per minibatch:
frozen IMU_Net forward (eval) -> [frozen Upper_Net forward (eval)] -> trained net forward -> L1(sum) loss
-> backward -> Adam.  Differences, all host-side: no DataLoader/numpy round trip, the loss stays on the
device (no .item() sync per step), gradients live in one flat buffer (one all-reduce for data parallel,
one fused Adam launch).  Each body is capturable into a HIP graph.
"""
import contextlib
import os

import torch

from . import hip, ops
from .params import FusedAdam
from .plan import StepPlan

# A step's HIP graphs: False (the product) = the whole body as ONE graph whose parallel branches are the body's side streams; True (tests flip it) = one
# linear-chain graph per stream segment, replayed on the body's own streams with event waits in between (plan.StepPlan).  Measured
# in r03: a graph with parallel branches dispatches TINY nodes in 5-6 us against 1.5-1.8 us for linear chains in separate graphs
# (scripts/bench_launch_gap.py), but with the step's real kernels the ~15 graph boundaries cost more than that saves: 5.35 against
# 5.25 ms per U+L step, 5.11 against 5.04 ms pipelined (bit-identical results; host enqueue time 0.6 against 2.1 ms per step).
_MULTI_GRAPH = False


def _capture_body(body, stream=None):
    """-> an object with .replay(): the body as HIP graphs (no state is changed: recording / capturing executes nothing).
    ``stream``: capture under this stream (per-stream resources -- ops.scratch -- are keyed by the stream a launch was captured
    on: a graph that will run BESIDE another one needs a capture stream of its own)."""
    if _MULTI_GRAPH:
        if stream is None:
            return StepPlan().record(body).build()
        with torch.cuda.stream(stream):
            return StepPlan().record(body).build()
    g = torch.cuda.CUDAGraph()
    if stream is None:
        with ops.capture(g):
            body()
    else:
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            with ops.capture(g, stream=stream):
                body()
    return g

from .skeleton import LOWER_MAP, UPPER_MAP


def allreduce_grads(flat, process_group):
    """SUM the flat gradient buffer over the ranks with ONE collective (RCCL over xGMI on the GPUs, gloo in the
    CPU tests).  The loss is a SUM over the batch, so the global-batch gradient is the SUM of the shard
    gradients -- not the mean."""
    if process_group is not None and torch.distributed.get_world_size(process_group) > 1:
        torch.distributed.all_reduce(flat.flat_g, op=torch.distributed.ReduceOp.SUM, group=process_group)


class GradBucket:
    """The flat gradient buffers of several nets back to back in ONE buffer: one all-reduce per step for all of them (the
    Upper and Lower stage of a U+L step: 1.2 + 2.5 MB -- latency-bound messages, so one collective instead of two is one
    launch + one ring latency saved per step)."""

    def __init__(self, nets):
        flats = [n.flat() for n in nets]
        sizes = [f.flat_g.numel() for f in flats]
        self.members = [id(f) for f in flats]
        self.buf = torch.zeros(sum(sizes), dtype=torch.float32, device=flats[0].flat_g.device)
        off = 0
        for f, n in zip(flats, sizes):
            f.rebind_grad(self.buf[off:off + n])
            f._bucket = self
            off += n

    @staticmethod
    def of(nets):
        """The bucket these nets already share (same nets, same order), else a new one.  Re-bucketing moves the gradient
        buffers, which would strand HIP graphs captured on the old addresses, so an existing bucket is always reused."""
        flats = [n.flat() for n in nets]
        b = getattr(flats[0], "_bucket", None)
        if b is not None and b.members == [id(f) for f in flats] and all(getattr(f, "_bucket", None) is b for f in flats):
            return b
        return GradBucket(nets)

    def allreduce(self, process_group):
        if process_group is not None and torch.distributed.get_world_size(process_group) > 1:
            torch.distributed.all_reduce(self.buf, op=torch.distributed.ReduceOp.SUM, group=process_group)


def shard_of(rank, world):
    """Slice of a global minibatch owned by ``rank`` (interleaved: balances a short last batch)."""
    return slice(rank, None, world)


def sync_replicas(net, process_group, src=0, params=True):
    """Make every rank's replica of ``net`` identical to rank ``src``'s: ONE broadcast of the flat parameter buffer (when
    ``params``) and one per registered buffer (BatchNorm running statistics, step counters, graph adjacency).  Called once when a
    data-parallel trainer starts (the nets are initialised from each rank's own RNG unless --seed is given) and once per epoch
    for the buffers alone (BatchNorm statistics are per shard, so the running averages drift apart; rank 0's are the ones
    that get saved and evaluated)."""
    if process_group is None or torch.distributed.get_world_size(process_group) < 2:
        return
    flat = net.flat()
    if params:
        torch.distributed.broadcast(flat.flat_p, src=src, group=process_group)
    seen = set()
    for b in net.buffers():
        if b.data_ptr() in seen:
            continue
        seen.add(b.data_ptr())
        torch.distributed.broadcast(b if b.dim() else b.view(1), src=src, group=process_group)


def empty_step(net, opt, process_group):
    """A rank whose shard of a short last global minibatch is empty still takes part in the step: it contributes a zero
    gradient to the all-reduce and applies the same Adam update as everybody else (otherwise the other ranks would wait
    for it forever and the replicas would part ways)."""
    flat = net.flat()
    ops.fill(flat.flat_g, 0.0)
    allreduce_grads(flat, process_group)
    opt.step()


def broadcast_flag(value, device, process_group, src=0):
    """Rank ``src``'s boolean decision (early stopping), agreed on by every rank."""
    if process_group is None or torch.distributed.get_world_size(process_group) < 2:
        return bool(value)
    t = torch.tensor([1.0 if value else 0.0], device=device)
    torch.distributed.broadcast(t, src=src, group=process_group)
    return bool(t.item() != 0.0)


def needs_exclusive(nets):
    """True when one of the nets runs bf16-MFMA kernels (IMUNet.precision / train_precision, UpperNet / LowerNet.precision other than
    "fp32").  r06: a kernel that shares a CU with a bf16-MFMA workgroup of another kernel can compute wrong
    results; the cause found there (packed-fp32 instructions with op_sel) is compiled out of the library, and INDEPENDENTLY of that the
    engines keep every launch of such a step ordered against its bf16-MFMA kernels (those on one chain; only all-fp32 bodies fork, behind
    it): nothing is resident beside a bf16-MFMA workgroup but its own kernel."""
    return any(getattr(m, a, "fp32") != "fp32" for m in nets if m is not None for a in ("precision", "train_precision"))


class StageStep:
    """One training stage's per-minibatch body with static buffers (graph friendly).

    ``imu_net=None`` takes the head pose from the recording (R_gt, ground-truth head joint) instead of the
    frozen IMU_Net -- the shipped reference snapshot has no IMU_Net checkpoint."""

    def __init__(self, stage, net, imu_net, upper_frozen=None, lr=3e-5, weight_decay=0.0, process_group=None,
                 use_graph=True, pose=None):
        assert stage in ("upper", "lower")
        self.stage, self.net, self.imu, self.upper_frozen = stage, net, imu_net, upper_frozen
        self.pose = pose              # (R, t) device buffers filled by somebody else (the "IMU-shared" arrangement)
        self.opt = FusedAdam(net.flat(), lr=lr, weight_decay=weight_decay)
        self.pg = process_group
        self.use_graph = use_graph
        self.graph = None
        self.static = None
        dev = next(net.parameters()).device
        self.jmap = torch.tensor(UPPER_MAP if stage == "upper" else LOWER_MAP, dtype=torch.int32, device=dev)
        self.loss2 = torch.zeros(2, dtype=torch.float32, device=dev)   # [L1 sum, sum of per-joint Euclidean distances]
        self.loss = self.loss2[:1]
        self.last_pred = None

    def _body(self):
        self._body_forward()
        self._body_backward()

    def _body_forward(self):
        """Fresh batch, head pose, forward of the trained net (and of the frozen Upper_Net in the Lower stage), loss."""
        s = self.static
        B, T = s["x"].shape[0], s["x"].shape[1]
        from .nets import UpperNet
        first_net = self.net if self.stage == "upper" else self.upper_frozen
        # fresh batch (x is transformed in place): Upper_Net's transform launch reads it from x_src; other nets get a copy first
        from .nets_local import UpperNetwlocal
        via_transform = (type(first_net) is UpperNet or (self.stage == "upper" and type(first_net) is UpperNetwlocal)) and s["x"].shape[-1] <= 8
        # the trained net's kinematics launch takes the loss, its gradient and the first backward step along (nets._head_fk)
        self.net.loss_hook = (s["target"], self.jmap, self.loss2, 1.0)
        try:
            self._body_forward_inner(s, B, T, first_net, via_transform)
        finally:
            self.net.loss_hook = None

    def _body_forward_inner(self, s, B, T, first_net, via_transform):
        from .nets import UpperNet
        if not via_transform:
            ops.copy2d(s["x_src"].view(B * T, -1), s["x"].view(B * T, -1))
        x_src = s["x_src"] if via_transform else None
        with torch.no_grad():
            if self.pose is not None:
                R, t = self.pose
            elif self.imu is not None:
                R, t = self.imu(s["imu"])
            else:
                R, t = s["R_gt"], s["t_gt"]
                ops.copy2d(s["target"].view(B * T, 63)[:, 60:63], t.view(B * T, 3))
            if self.stage == "upper":
                from .nets_local import UpperNetwlocal
                if isinstance(self.net, UpperNetwlocal):     # (Net/Upper_Net.py:406-432: a second state pair for the anchor branch)
                    l = self.net._forward_impl(s["x"], s["h0"], s["c0"], s["h0"], s["c0"], s["body"], R, t, stash=True, x_src=x_src)[0]
                elif via_transform:
                    l = self.net._forward_impl(s["x"], s["h0"], s["c0"], s["body"], R, t, stash=True, x_src=x_src)[0]
                else:
                    l = self.net._forward_impl(s["x"], s["h0"], s["c0"], s["body"], R, t, stash=True)[0]
                nsel = 15
            else:
                if via_transform and not self.upper_frozen.training:
                    up = self.upper_frozen._forward_impl(s["x"], s["h0"], s["c0"], s["body"], R, t, stash=False, x_src=x_src)[0]
                else:
                    if via_transform:
                        ops.copy2d(s["x_src"].view(B * T, -1), s["x"].view(B * T, -1))
                    up = self.upper_frozen(s["x"], s["h0"], s["c0"], s["body"], R, t)[0]
                l = self.net._forward_impl(up, s["x"], s["body"], R, t, stash=True)[0]
                nsel = 8
            if not getattr(self.net, "_dy_ready", False):
                hip.call("l1_loss", l, s["target"], self.jmap, nsel, 21, B * T, 1.0, self.loss2, s["dl"])
        self.last_pred = l

    def _body_backward(self):
        with torch.no_grad():
            self.net._backward_impl(self.static["dl"])

    def bind(self, x, imu, body, target, R_gt=None):
        """Register the (device-resident) minibatch buffers; contents may be overwritten between steps."""
        dev = x.device
        B, T = x.shape[0], x.shape[1]
        nsel = 15 if self.stage == "upper" else 8
        if self.imu is None and R_gt is None and self.pose is None:
            raise ValueError("StageStep without an IMU_Net needs the recorded head rotations (R_gt)")
        self.static = dict(x_src=x, x=torch.empty_like(x), imu=imu, body=body, target=target,
                           h0=torch.zeros(6, B, 64, device=dev), c0=torch.zeros(6, B, 64, device=dev),
                           dl=torch.empty(B, T, nsel, 3, device=dev), R_gt=R_gt,
                           t_gt=torch.empty(B, T, 3, device=dev))
        self.graph = None

    def _mutable_state(self):
        """What a body changes besides gradients/activations: BatchNorm running statistics + step counters and the dropout
        counter of the trained net (the frozen nets run in eval mode)."""
        return list(self.net.buffers()) + [self.net.seed_counter()]

    def warm_up(self):
        """Run the body once WITHOUT side effects (sizes the arenas, sets kernel attributes before graph capture): the
        mutable state is put back afterwards, so graph and eager runs -- and a resumed run -- see identical states."""
        keep = [t.clone() for t in self._mutable_state()]
        self._body()
        torch.cuda.synchronize()
        for t, k in zip(self._mutable_state(), keep):
            t.copy_(k)
        torch.cuda.synchronize()

    def prepare(self):
        """Size the arenas and capture the HIP graph without changing any state (so that the first step() costs what
        every step costs)."""
        if self.use_graph and self.graph is None:
            self.warm_up()
            self.graph = _capture_body(self._body)

    def step(self):
        if self.use_graph:
            self.prepare()
            self.graph.replay()
        else:
            self._body()
        allreduce_grads(self.net._flat, self.pg)
        self.opt.step()
        return self.loss


class ImuStep:
    """Stage-1 per-minibatch body (reference Processor/Train/Train_IMU.py:114-149): IMU_Net forward, geodesic + 100 x
    position loss (sum), backward through the two BiLSTM(512) stacks, Adam with coupled weight decay -- with static
    buffers, capturable into one HIP graph (the eager body is ~650 launches and CPU-launch bound)."""

    def __init__(self, net, lr=1e-4, weight_decay=0.001, process_group=None, use_graph=True):
        self.net = net
        self.opt = FusedAdam(net.flat(), lr=lr, weight_decay=weight_decay)
        self.pg = process_group
        self.use_graph = use_graph
        self.graph = None
        self.static = None
        dev = next(net.parameters()).device
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)

    def bind(self, imu, R_gt, target):
        dev = imu.device
        B, T = imu.shape[0], imu.shape[1]
        self.static = dict(imu=imu, R_gt=R_gt, target=target, head=torch.empty(B, T, 3, device=dev),
                           dR=torch.empty(B, T, 3, 3, device=dev), dt=torch.empty(B, T, 3, device=dev))
        self.graph = None

    def _body(self):
        from . import imu_train
        s = self.static
        B, T = s["imu"].shape[0], s["imu"].shape[1]
        with torch.no_grad():
            ops.copy2d(s["target"].view(B * T, 63)[:, 60:63], s["head"].view(B * T, 3))     # head joint = joint 20
            R, t = imu_train.forward_train(self.net, s["imu"])
            hip.call("imu_loss", R, t, s["R_gt"], s["head"], B * T, 1.0, self.loss, s["dR"], s["dt"])
            imu_train.backward(self.net, s["dR"], s["dt"])

    def step(self):
        if self.use_graph:
            if self.graph is None:
                keep = self.net.seed_counter().clone()
                self._body()                                   # warm-up (side-effect free: IMU_Net has no BatchNorm)
                torch.cuda.synchronize()
                self.net.seed_counter().copy_(keep)
                self.graph = _capture_body(self._body)
            self.graph.replay()
        else:
            self._body()
        allreduce_grads(self.net._flat, self.pg)
        self.opt.step()
        return self.loss


class SharedImuStages:
    """The "IMU-shared" arrangement of SURVEY 8-d: ONE frozen IMU_Net forward per minibatch feeds both stage bodies (they
    run as two concurrent branches after it).  Not what the reference does (each stage program runs its own IMU_Net
    forward); reported beside the literal U+L step."""

    def __init__(self, imu_net, stages, imu_in, use_graph=True):
        self.imu, self.stages, self.imu_in = imu_net, list(stages), imu_in
        B, T = imu_in.shape[0], imu_in.shape[1]
        dev = imu_in.device
        self.R, self.t = torch.empty(B, T, 3, 3, device=dev), torch.empty(B, T, 3, device=dev)
        for st in self.stages:
            st.pose = (self.R, self.t)
        self.pair = ConcurrentStages(self.stages, use_graph=False)
        self.pair.extra_nets = [self.imu]
        self.use_graph, self.graph = use_graph, None

    def _body(self):
        with torch.no_grad(), (ops.no_fork() if self.pair.exclusive() else contextlib.nullcontext()):
            R, t = self.imu(self.imu_in)
            ops.copy2d(R.view(-1, 9), self.R.view(-1, 9))
            ops.copy2d(t.view(-1, 3), self.t.view(-1, 3))
        self.pair._bodies()

    def step(self):
        if self.use_graph:
            if self.graph is None:
                keep = [[t.clone() for t in st._mutable_state()] for st in self.stages]
                for _ in range(2):
                    self._body()
                    torch.cuda.synchronize()
                for st, ks in zip(self.stages, keep):
                    for t, k in zip(st._mutable_state(), ks):
                        t.copy_(k)
                torch.cuda.synchronize()
                self.graph = _capture_body(self._body)
            self.graph.replay()
        else:
            self._body()
        self.pair.allreduce()
        for st in self.stages:
            st.opt.step()


class ConcurrentStages:
    """Several INDEPENDENT stage bodies per minibatch as concurrent branches of one HIP graph.

    In the reference the Upper and Lower stages are separate programs (Train_Lower.py:129-137 loads a frozen, already
    trained Upper_Net; nothing flows from one body to the other), so their per-minibatch bodies may run side by side.
    On the GPU that matters: half of a body is IMU_Net's compute-bound products, the other half is hundreds of small
    latency-bound kernels that leave most CUs idle; interleaving two bodies fills those CUs (measured: 8.2 -> 7.1 ms
    per U+L step).  Each stage must own its buffers: in particular each needs its OWN frozen IMU_Net instance (the
    nets keep their activations in per-instance arenas).  Results are bit-identical to running the stages one after
    the other (every reduction in the kernels has a fixed order)."""

    def __init__(self, stages, use_graph=True, unguarded=False):
        self.stages = list(stages)
        # unguarded: keep the concurrent branches even when a net runs bf16-MFMA kernels (bench.py's comparison figure and the
        # reproducers of scripts/ only; see needs_exclusive)
        self.unguarded = unguarded
        nets_used = [id(m) for st in self.stages for m in (st.net, st.imu, st.upper_frozen) if m is not None]
        if len(set(nets_used)) != len(nets_used):
            raise ValueError("ConcurrentStages: stages share a network instance (each stage needs its own IMU_Net / "
                             "frozen Upper_Net copy: their activation arenas would be written concurrently)")
        self.use_graph = use_graph
        self.graph = None
        self.extra_nets = []                                  # nets that run beside the stages without being theirs (the owner's IMU_Nets)
        self.side = [torch.cuda.Stream() for _ in self.stages[1:]]
        # data parallel: the stages' gradients share one buffer, so one collective per step serves all of them
        self.bucket = None
        pgs = {id(st.pg) for st in self.stages}
        pg = self.stages[0].pg
        if len(pgs) == 1 and pg is not None and torch.distributed.get_world_size(pg) > 1 and len(self.stages) > 1:
            self.bucket = GradBucket.of([st.net for st in self.stages])

    def allreduce(self):
        if self.bucket is not None:
            self.bucket.allreduce(self.stages[0].pg)
        else:
            for st in self.stages:
                allreduce_grads(st.net._flat, st.pg)

    def exclusive(self):
        """One chain instead of branches: a net of a stage (or one handed in by PipelinedStages / SharedImuStages) runs bf16 MFMAs."""
        return not self.unguarded and needs_exclusive([m for st in self.stages for m in (st.net, st.imu, st.upper_frozen)] + list(self.extra_nets))

    def _bodies(self):
        if not self.exclusive():
            return self._bodies_concurrent()
        # A net runs bf16 MFMAs (needs_exclusive): everything that may launch such a kernel goes onto ONE chain -- the frozen IMU_Net
        # forwards first, one after the other, every fork site inside them closed (ops.no_fork) -- and only bodies whose own nets are
        # all fp32 fork off behind it (one stage's tail beside the other's: neither holds a bf16-MFMA kernel).  Bodies with such a net
        # (a frozen Upper_Net in bf16 precision) stay on the chain as well.  The structural test reads exactly this off the recorded
        # launch graph (tests/test_split3_gpu.py::test_no_kernel_can_run_beside_a_bf16_mfma_kernel).
        main = torch.cuda.current_stream()
        stages = self.stages
        keep = [(st.imu, st.pose) for st in stages]
        try:
            with ops.no_fork(), torch.no_grad():
                for st in reversed(stages):
                    if st.imu is not None and st.pose is None:
                        st.pose = st.imu(st.static["imu"])
            hot_bodies = needs_exclusive([m for st in stages for m in (st.net, st.upper_frozen)])
            if hot_bodies or len(stages) == 1 or not ops.capture_can_fork():
                with ops.no_fork():
                    for st in reversed(stages):               # (the order of the branch form; the results do not depend on it)
                        st._body()
            else:
                for i in range(len(stages) - 1, 0, -1):
                    self.side[i - 1].wait_stream(main)
                    with torch.cuda.stream(self.side[i - 1]):
                        stages[i]._body()
                stages[0]._body()
                for side in self.side:
                    main.wait_stream(side)
        finally:
            for st, (imu, pose) in zip(stages, keep):
                st.imu, st.pose = imu, pose

    def _bodies_concurrent(self):
        """Branch order: the LAST stage (longest tail: the Lower body also runs the frozen Upper_Net) gets its IMU_Net forward
        first, alone on the GPU, with its recurrences as two single-direction chains (blocks.two_chains); each earlier stage's
        IMU_Net forward follows on the launching stream when that one has finished, with the previous stage's small-kernel tail
        beside it; each stage's remaining body forks off right after its own forward and takes the head pose from a per-stage
        buffer.  The IMU_Net forwards are compute-bound and gain nothing from running side by side, whereas the small-kernel
        tail of one stage overlaps with the IMU_Net forward of another (measured: 6.88 -> 6.54 ms per U+L step).  Other
        arrangements that were measured and dropped: NOTES.md."""
        main = torch.cuda.current_stream()
        stages, streams = self.stages, [main] + self.side
        keep = [(st.imu, st.pose) for st in stages]
        try:
            for i in range(len(stages) - 1, -1, -1):
                st = stages[i]
                if st.imu is not None and st.pose is None:
                    from . import blocks
                    # the FIRST forward (the last stage's) has the GPU to itself: its recurrences run as two chains; the
                    # later ones have another stage's tail beside them, which fills the same gaps (blocks.two_chains)
                    # r03: with the shorter tails the later forwards gain from the two-chain form as well (5.30 -> 5.24 ms per
                    # U+L step), so all of them take it
                    with torch.no_grad(), blocks.two_chains(True):
                        # (the forward's own output tensors serve as the stage's head pose: they stay referenced -- and, under
                        # capture, reserved in the graph's pool -- until every branch has been enqueued; no copies)
                        st.pose = st.imu(st.static["imu"])
                if i > 0:
                    streams[i].wait_stream(main)
                    with torch.cuda.stream(streams[i]):
                        st._body()
                else:
                    st._body()
        finally:
            for st, (imu, pose) in zip(stages, keep):
                st.imu, st.pose = imu, pose
        for side in self.side:
            main.wait_stream(side)

    def prepare(self):
        """Warm-up (side-effect free) and graph capture, so that the first step() costs what every step costs."""
        if self.use_graph and self.graph is None:
            for st in self.stages:                          # warm-up one by one: sizes arenas, sets kernel attributes
                st.warm_up()
            keep = [[t.clone() for t in st._mutable_state()] for st in self.stages]
            self._bodies()                                  # per-stream scratch buffers of the side streams
            torch.cuda.synchronize()
            for st, ks in zip(self.stages, keep):
                for t, k in zip(st._mutable_state(), ks):
                    t.copy_(k)
            torch.cuda.synchronize()
            self.graph = _capture_body(self._bodies)

    def step(self):
        if self.use_graph:
            self.prepare()
            self.graph.replay()
        else:
            self._bodies()
        self.allreduce()
        for st in self.stages:
            st.opt.step()
        return [st.loss for st in self.stages]


class PipelinedStages:
    """ConcurrentStages with the frozen IMU_Net forwards moved one minibatch ahead (a prefetch pipeline).

    The IMU_Net forwards of a stage body depend on nothing the step changes (frozen weights, the minibatch's IMU samples), so
    the forwards for minibatch i+1 can run while the trainable bodies work on minibatch i: the compute-bound half of a body
    overlaps the latency-bound half of the previous one.  One replay = [head poses of minibatch i move from the "next" to the
    "current" buffers] -> {Upper body(i) | Lower body(i) | IMU_Net_L(i+1) | IMU_Net_U(i+1)} as four concurrent branches: the two
    forwards run SIDE BY SIDE (each on a stream of its own, forked from the capture's origin), so that one net's recurrent step
    fills the launch gap / prologue / cell update of the other's -- 19.3 us per timestep and net instead of 23.4 (bench.py
    `recurrence_graph`), 5.18 ms per U+L step instead of 5.48 with the two forwards one after the other in ONE branch
    (side_by_side = False).  Within one minibatch the same pairing loses (both tails then start together, NOTES.md); across minibatches nothing waits for the forwards.
    Every step still runs both IMU_Net forwards in full; results are bit-identical to ConcurrentStages on the same sequence of
    minibatches (tests/test_hip_local.py).  `imu_next` is the static buffer the caller fills with minibatch i+1's IMU samples
    before step i; `prime()` runs the forwards for the first minibatch."""

    def __init__(self, stages, imu_nets, imu_next, use_graph=True, unguarded=False):
        self.stages, self.imus, self.imu_next = list(stages), list(imu_nets), imu_next
        if len(self.stages) != len(self.imus) or any(st.imu is not None for st in self.stages):
            raise ValueError("PipelinedStages: one IMU_Net per stage, and the stages themselves must be built with imu_net=None")
        if len({id(m) for m in self.imus}) != len(self.imus):
            raise ValueError("PipelinedStages: every stage needs its own IMU_Net instance")
        B, T = imu_next.shape[0], imu_next.shape[1]
        dev = imu_next.device
        mk = lambda: (torch.empty(B, T, 3, 3, device=dev), torch.empty(B, T, 3, device=dev))
        self.cur = [mk() for _ in self.stages]
        self.nxt = [mk() for _ in self.stages]
        for st, pose in zip(self.stages, self.cur):
            st.pose = pose
        self.pair = ConcurrentStages(self.stages, use_graph=False, unguarded=unguarded)
        self.pair.extra_nets = self.imus
        self.side = torch.cuda.Stream()
        self.sides = [self.side] + [torch.cuda.Stream() for _ in self.imus[1:]]
        self.side_by_side = True
        self.use_graph, self.graph = use_graph, None

    def _imu_forward(self, k, persistent=True):
        from . import blocks
        net, (Rn, tn) = self.imus[k], self.nxt[k]
        # (the stage bodies run beside these forwards: see blocks.two_chains; `persistent`: blocks.seq_xcd)
        with torch.no_grad(), blocks.two_chains(False), blocks.seq_xcd(persistent):
            R, t = net(self.imu_next)
            ops.copy2d(R.view(-1, 9), Rn.view(-1, 9))
            ops.copy2d(t.view(-1, 3), tn.view(-1, 3))

    def _imu_forwards(self):
        for k in reversed(range(len(self.imus))):             # (Lower's first, as in ConcurrentStages)
            self._imu_forward(k)

    def prime(self):
        """Head poses of the first minibatch (its IMU samples are in `imu_next`)."""
        self._imu_forwards()
        torch.cuda.synchronize()

    def _body(self):
        main = torch.cuda.current_stream()
        for (Rc, tc), (Rn, tn) in zip(self.cur, self.nxt):
            ops.copy2d(Rn.view(-1, 9), Rc.view(-1, 9))
            ops.copy2d(tn.view(-1, 3), tc.view(-1, 3))
        if self.pair.exclusive():
            # a net runs bf16 MFMAs: the prefetched forwards and the bodies as ONE chain (needs_exclusive)
            with ops.no_fork():
                self._imu_forwards()
            self.pair._bodies()                               # (forks only all-fp32 bodies, behind the forwards)
            return
        if self.side_by_side:
            # every forward that runs at once needs its own co-resident set of 256 workgroups for the persistent rnn_slow launch:
            # the device holds `slots` of them (2 on a whole MI355X); the others take the launch-per-timestep form
            slots = hip.lib().mmego_lstm_seq_xcd_slots()
            for n, k in enumerate(reversed(range(len(self.imus)))):
                self.sides[k].wait_stream(main)
                with torch.cuda.stream(self.sides[k]):
                    self._imu_forward(k, persistent=n < slots)
            self.pair._bodies()
            for sd in self.sides:
                main.wait_stream(sd)
            return
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self._imu_forwards()
        self.pair._bodies()
        main.wait_stream(self.side)

    def prepare(self):
        if self.use_graph and self.graph is None:
            keep = [[t.clone() for t in st._mutable_state()] for st in self.stages]
            keep_pose = [[t.clone() for pr in (self.cur, self.nxt) for pose in pr for t in pose]]
            for _ in range(2):
                self._body()
                torch.cuda.synchronize()
            for st, ks in zip(self.stages, keep):
                for t, k in zip(st._mutable_state(), ks):
                    t.copy_(k)
            for t, k in zip([t for pr in (self.cur, self.nxt) for pose in pr for t in pose], keep_pose[0]):
                t.copy_(k)
            torch.cuda.synchronize()
            self.graph = _capture_body(self._body)

    def step(self):
        if self.use_graph:
            self.prepare()
            self.graph.replay()
        else:
            self._body()
        self.pair.allreduce()
        for st in self.stages:
            st.opt.step()
        return [st.loss for st in self.stages]
