"""CPU, world_size 2 over gloo: the data-parallel plumbing (flat gradient buffer, SUM all-reduce, sharding)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    q.put((rank, ok, alias, idx))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_sum_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res)
    assert res[0][3] == [0, 2, 4, 6, 8] and res[1][3] == [1, 3, 5, 7, 9]      # disjoint, exhaustive shards
