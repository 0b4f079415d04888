"""CPU, world_size 2 over gloo: the data-parallel plumbing (flat gradient buffer, SUM all-reduce, sharding)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _init(rank, world, store):
    """Rendezvous through a file:// store in the test's tmp_path: no TCP port is picked and handed over, so there is no
    bind-close-reuse race between choosing a port and the ranks' store (a lost race ends in init_process_group's own timeout)."""
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)


def _worker(rank, world, store, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    _init(rank, world, store)
    from mmego_amd import nets
    from mmego_amd.params import FlatParams
    from mmego_amd.train_step import allreduce_grads, shard_of
    torch.manual_seed(0)
    net = nets.UpperNet()                                  # parameter container only: no kernel runs on CPU
    flat = FlatParams(net).ensure()
    # parameters are views of one flat buffer, 16-byte aligned, state_dict untouched
    assert all(p.data_ptr() % 16 == 0 for p in net.parameters())
    assert flat.flat_p.numel() >= sum(p.numel() for p in net.parameters())
    for p in net.parameters():
        flat.grad(p).fill_(float(rank + 1))
    allreduce_grads(flat, dist.group.WORLD)
    flat.bind_grads()
    ok = all(torch.all(p.grad == 3.0).item() for p in net.parameters())      # 1 + 2: SUM, not mean
    alias = all(p.grad.data_ptr() == flat.grad(p).data_ptr() for p in net.parameters())
    idx = list(range(10))[shard_of(rank, world)]
    # replicas built from different seeds agree after sync_replicas (weights AND buffers); rank 0 decides early stopping;
    # two nets' gradients in one bucket are summed by ONE collective
    from mmego_amd.train_step import GradBucket, broadcast_flag, sync_replicas
    torch.manual_seed(100 + rank)
    a, b = nets.UpperNet(), nets.LowerNet(64)
    for m in (a, b):
        for buf in m.buffers():
            if buf.dtype.is_floating_point:
                buf.add_(float(rank))
        sync_replicas(m, dist.group.WORLD)
    torch.manual_seed(100)
    a0, b0 = nets.UpperNet(), nets.LowerNet(64)
    synced = all(torch.equal(v, w) for m, m0 in ((a, a0), (b, b0)) for v, w in zip(m.state_dict().values(), m0.state_dict().values()))
    bucket = GradBucket([a, b])
    same_bucket = GradBucket.of([a, b]) is bucket
    a.flat().flat_g.fill_(float(rank + 1))
    b.flat().flat_g.fill_(10.0 * (rank + 1))
    bucket.allreduce(dist.group.WORLD)
    a.flat().bind_grads(); b.flat().bind_grads()
    bucket_ok = (all(torch.all(p.grad == 3.0).item() for p in a.parameters()) and all(torch.all(p.grad == 30.0).item() for p in b.parameters())
                 and a.flat().flat_g.data_ptr() == bucket.buf.data_ptr() and same_bucket)
    flag = broadcast_flag(rank == 0, torch.device("cpu"), dist.group.WORLD)       # rank 1 says False, rank 0's True wins
    q.put((rank, ok, alias, idx, synced, bucket_ok, flag))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_sum_and_sharding(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker, args=(r, 2, store, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)
    assert all(r[4] for r in res), "sync_replicas: every rank holds rank 0's parameters and buffers"
    assert all(r[5] for r in res), "GradBucket: both nets' gradients summed by one collective"
    assert all(r[6] is True for r in res), "broadcast_flag: rank 0's decision on every rank"
    assert res[0][3] == [0, 2, 4, 6, 8] and res[1][3] == [1, 3, 5, 7, 9]      # disjoint, exhaustive shards


class _CpuAdam:
    """Test stand-in for params.FusedAdam (the product optimiser is a HIP kernel): the same update rule in torch ops on the flat
    buffers, so that the data-parallel CONTROL FLOW of a step can run on CPU ranks."""

    def __init__(self, flat, lr=3e-5):
        self.flat, self.lr, self.t = flat, lr, 0
        self.m, self.v = torch.zeros_like(flat.flat_p), torch.zeros_like(flat.flat_p)

    def step(self):
        self.t += 1
        g = self.flat.flat_g
        self.m.mul_(0.9).add_(g, alpha=0.1)
        self.v.mul_(0.999).addcmul_(g, g, value=0.001)
        mh, vh = self.m / (1 - 0.9 ** self.t), self.v / (1 - 0.999 ** self.t)
        self.flat.flat_p.sub_(self.lr * mh / (vh.sqrt() + 1e-8))


def _worker_empty_shard(rank, world, store, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world))
    _init(rank, world, store)
    from mmego_amd import nets, ops, train_step
    from mmego_amd.params import FlatParams
    ops.fill = lambda t, v: t.fill_(v)                     # (CPU stand-in for the device fill kernel: test-side only)
    torch.manual_seed(5)
    net = nets.UpperNet()
    flat = net.flat()                                      # (the net's own flat buffers: what empty_step works on)
    opt = _CpuAdam(flat)
    gen = torch.Generator().manual_seed(77)
    steps = []
    for it in range(3):
        g_full = torch.randn(flat.flat_g.shape, generator=gen)      # the same stream on both ranks
        # a short last global minibatch: only rank 0 owns sequences in step 1 (rank 1's shard is empty); in steps 0 and 2 both
        # ranks hold half of the gradient
        if it == 1:
            if rank == 0:
                flat.flat_g.copy_(g_full)
                train_step.allreduce_grads(flat, dist.group.WORLD)
                opt.step()
            else:
                flat.flat_g.fill_(123.0)                   # stale content of the last backward: empty_step must zero it
                train_step.empty_step(net, opt, dist.group.WORLD)
        else:
            flat.flat_g.copy_(g_full * (0.25 if rank == 0 else 0.75))
            train_step.allreduce_grads(flat, dist.group.WORLD)
            opt.step()
        steps.append(flat.flat_p.clone())
    # single-process reference: the full gradient every step
    torch.manual_seed(5)
    ref_net = nets.UpperNet()
    ref_flat = FlatParams(ref_net).ensure()
    ref_opt = _CpuAdam(ref_flat)
    gen = torch.Generator().manual_seed(77)
    for it in range(3):
        g_full = torch.randn(ref_flat.flat_g.shape, generator=gen)
        ref_flat.flat_g.copy_(g_full * 0.25 + g_full * 0.75 if it != 1 else g_full)
        ref_opt.step()
    q.put((rank, [s.numpy().tobytes() for s in steps], torch.equal(steps[-1], ref_flat.flat_p), opt.t))
    dist.destroy_process_group()


def test_rank_with_an_empty_last_shard_stays_bit_equal(tmp_path):
    """train_step.empty_step (a rank whose shard of a short last global minibatch is empty): zero gradient into the SUM, the same
    optimiser update -- both ranks' parameters stay bit-equal through and after that step, and equal a one-process run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker_empty_shard, args=(r, 2, store, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1], "replicas parted ways around the empty-shard step"
    assert res[0][2] and res[1][2], "data-parallel run != single-process run on the full gradient"
    assert res[0][3] == res[1][3] == 3, "every rank took every optimiser step"
