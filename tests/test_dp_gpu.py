"""GPU, two ranks on ONE MI355X (gloo between them): the data-parallel U+L step end to end through the real StageStep /
ConcurrentStages bodies (reference Train_Upper.py:154-182, Train_Lower.py:186-224 per rank).

Each rank owns a shard of the minibatch, builds its nets from its OWN seed (as an unseeded `main.py --train` would), takes rank
0's weights through train_step.sync_replicas, runs the two stage bodies as one HIP graph, sums the gradients of BOTH stages
with one all-reduce (GradBucket) and applies Adam.  Checked against a single process that runs the two shards one after the
other: the all-reduced gradient must equal g(shard 0) + g(shard 1) bit for bit, the parameters after Adam must be bit-equal on
both ranks and equal to the single-process result, and each rank's BatchNorm running statistics must be its own shard's."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import golden

pytestmark = pytest.mark.gpu


def _batch(dev):
    g = golden("g6_train.npz")
    x, body, target = [torch.tensor(np.asarray(g[k])).to(dev) for k in ("x", "body", "target")]
    imu = torch.randn(4, 8, 20, 15, generator=torch.Generator().manual_seed(3)).to(dev)
    return x, imu, body, target


def _build(dev, seed):
    from mmego_amd import nets
    torch.manual_seed(seed)
    up, lo = nets.UpperNet().to(dev).train(), nets.LowerNet(64).to(dev).train()
    torch.manual_seed(50)                                        # frozen nets come from checkpoints: identical everywhere
    imu_u = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
    imu_l = nets.IMUNet(15, 9, 64, 2, True, 0.1).to(dev).eval()
    imu_l.load_state_dict(imu_u.state_dict())
    fr = nets.UpperNet().to(dev).eval()
    return up, lo, imu_u, imu_l, fr


def _stages(nets_, shard, pg, use_graph):
    from mmego_amd.train_step import StageStep
    up, lo, imu_u, imu_l, fr = nets_
    x, imu, body, target = shard
    su = StageStep("upper", up, imu_u, lr=3e-5, process_group=pg, use_graph=use_graph)
    sl = StageStep("lower", lo, imu_l, upper_frozen=fr, lr=3e-5, process_group=pg, use_graph=use_graph)
    for st in (su, sl):
        st.bind(x.clone(), imu, body, target)
    return su, sl


def _worker(rank, world, store, q):
    # rendezvous through a file:// store in tmp_path (no port picked by bind-close and handed over: no reuse race)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)
    try:
        from mmego_amd.train_step import ConcurrentStages, shard_of, sync_replicas
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        pg = dist.group.WORLD
        nets_ = _build(dev, seed=1000 + rank)                    # DIFFERENT trainable weights per rank before the sync
        for net in nets_[:2]:
            sync_replicas(net, pg)
        sh = shard_of(rank, world)
        shard = tuple(v[sh].contiguous() for v in _batch(dev))
        su, sl = _stages(nets_, shard, pg, use_graph=True)
        both = ConcurrentStages([su, sl], use_graph=True)
        assert both.bucket is not None, "one gradient bucket for both stages"
        out = {}
        for step in range(2):
            both.step()
            torch.cuda.synchronize()
            # (numpy arrays: pickled by value -- torch tensors would travel as file descriptors of a process that is gone)
            out["g%d" % step] = [st.net.flat().flat_g.cpu().numpy().copy() for st in (su, sl)]
            out["p%d" % step] = [st.net.flat().flat_p.cpu().numpy().copy() for st in (su, sl)]
        out["buffers"] = [[b.cpu().numpy().copy() for b in st.net.buffers()] for st in (su, sl)]
        out["loss"] = [st.loss.item() for st in (su, sl)]
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_ul_step_equals_sum_of_shard_gradients(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker, args=(r, 2, store, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in res:
        for k, v in res[r].items():
            if k != "loss":
                res[r][k] = [[torch.from_numpy(a) for a in e] if isinstance(e, list) else torch.from_numpy(e) for e in v]
    # single-process reference: rank 0's initial weights (seed 1000), the two shards one after the other, eager, no process group
    from mmego_amd.train_step import shard_of
    dev = torch.device("cuda:0")
    full = _batch(dev)
    reps = []
    for r in range(2):
        nets_ = _build(dev, seed=1000)
        shard = tuple(v[shard_of(r, 2)].contiguous() for v in full)
        reps.append(_stages(nets_, shard, None, use_graph=False))
    for step in range(2):
        for su, sl in reps:
            su._body(); sl._body()
        torch.cuda.synchronize()
        for k in range(2):                                               # k = 0: Upper stage, 1: Lower stage
            g0, g1 = reps[0][k].net.flat().flat_g, reps[1][k].net.flat().flat_g
            gsum = g0 + g1
            for r in range(2):
                assert torch.equal(res[r]["g%d" % step][k], gsum.cpu()), ("all-reduced gradient == g0 + g1", step, k, r)
            for rep in reps:                                             # every replica applies the same summed gradient
                rep[k].net.flat().flat_g.copy_(gsum)
                rep[k].opt.step()
            torch.cuda.synchronize()
            pref = reps[0][k].net.flat().flat_p.cpu()
            assert torch.equal(reps[1][k].net.flat().flat_p.cpu(), pref)
            assert torch.equal(res[0]["p%d" % step][k], res[1]["p%d" % step][k]), ("parameters bit-equal on both ranks", step, k)
            assert torch.equal(res[0]["p%d" % step][k], pref), ("... and equal to the single-process result", step, k)
    for r in range(2):                                                   # BatchNorm statistics are per shard (local BN)
        for k in range(2):
            for a, b in zip(res[r]["buffers"][k], reps[r][k].net.buffers()):
                assert torch.equal(a, b.cpu()), ("rank's BatchNorm buffers are its own shard's", r, k)
            assert res[r]["loss"][k] == reps[r][k].loss.item()
    assert not torch.equal(res[0]["buffers"][0][0], res[1]["buffers"][0][0]), "the shards really differ"


def _rccl_worker(store, q):
    os.environ.update(RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1, device_id=dev)   # "nccl" IS RCCL on ROCm
    try:
        from mmego_amd import train_step
        from mmego_amd.train_step import ConcurrentStages, broadcast_flag, sync_replicas
        pg = dist.group.WORLD
        real_ws = dist.get_world_size
        # the one-GPU box allows one rank per device: make the trainers' "world > 1" tests true so that every collective the
        # data-parallel path issues really goes through the (one-rank) RCCL communicator
        dist.get_world_size = lambda group=None: 2
        nets_ = _build(dev, seed=1000)
        for net in nets_[:2]:
            sync_replicas(net, pg)                                   # broadcasts
        shard = _batch(dev)
        su, sl = _stages(nets_, shard, pg, use_graph=True)
        both = ConcurrentStages([su, sl], use_graph=True)
        assert both.bucket is not None
        for _ in range(2):
            both.step()                                              # graph replay -> RCCL all-reduce of the bucket -> fused Adam
        torch.cuda.synchronize()
        assert broadcast_flag(True, dev, pg) is True
        dist.get_world_size = real_ws
        q.put(dict(g=[st.net.flat().flat_g.cpu().numpy().copy() for st in (su, sl)],
                   p=[st.net.flat().flat_p.cpu().numpy().copy() for st in (su, sl)],
                   backend=dist.get_backend(pg)))
    finally:
        dist.destroy_process_group()


def test_single_rank_rccl_collectives_in_the_training_step(tmp_path):
    """The collectives of the data-parallel step on the REAL backend (RCCL, `init_process_group("nccl", device_id=...)` as bench.py
    and the trainers do), with one rank -- the one-GPU box refuses two ranks on a device: replica sync broadcasts, the gradient
    bucket's all-reduce between the graph replay and the fused Adam, the early-stopping broadcast.  A sum over one rank must leave
    the step's results exactly those of a run without a process group."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(str(tmp_path / "rendezvous"), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res["backend"] == "nccl"
    dev = torch.device("cuda:0")
    from mmego_amd.train_step import ConcurrentStages
    nets_ = _build(dev, seed=1000)
    su, sl = _stages(nets_, _batch(dev), None, use_graph=True)
    both = ConcurrentStages([su, sl], use_graph=True)
    for _ in range(2):
        both.step()
    torch.cuda.synchronize()
    for k, st in enumerate((su, sl)):
        assert np.array_equal(res["g"][k], st.net.flat().flat_g.cpu().numpy())
        assert np.array_equal(res["p"][k], st.net.flat().flat_p.cpu().numpy())
