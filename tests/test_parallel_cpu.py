"""CPU (gloo, world_size 2): the data-parallel plumbing — in-place all-reduce of a gradient arena in chunks, partial
(per-network) asynchronous launches, 1/world folding into the optimizer's grad_scale, and the rank-ordered
all-gather used for the ClusterMemory update.  The kernels themselves are GPU-only; this covers the N>1 logic."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rg_hip.optim import Arena
from rg_hip.parallel import GradReducer, all_gather_rows, world_size


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeOptimizer(object):
    def __init__(self, arena):
        self._arena = arena
        self.grad_scale = 1.0


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                                  # identical replicas
        net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 11))
        extra = torch.nn.Linear(5, 3)                         # a second "network" in the same arena
        params = list(net.parameters()) + list(extra.parameters())
        arena = Arena(params)
        opt = _FakeOptimizer(arena)
        red = GradReducer(opt, bucket_mb=0.001)               # ~262 elements per chunk -> many chunks
        assert red.active() and opt.grad_scale == 1.0 / world

        g = torch.Generator().manual_seed(1)
        x = torch.randn(8 * world, 37, generator=g)
        y = torch.randn(8 * world, 11, generator=g)
        xe = torch.randn(8 * world, 5, generator=g)
        sl = slice(8 * rank, 8 * (rank + 1))                  # this rank's shard
        loss = (net(x[sl]) - y[sl]).pow(2).mean() + extra(xe[sl]).pow(2).mean()
        loss.backward()
        for p in params:                                      # what the tape runtime does on the GPU
            p._rg_grad.copy_(p.grad)
            p.grad = p._rg_grad
        # overlap pattern of the FD-GAN generator update: launch one network early, the rest at the end
        red.reduce_async(list(extra.parameters()))
        red.reduce()
        got = [(p.grad * opt.grad_scale).clone() for p in params]

        # single-process reference on the global batch
        torch.manual_seed(0)
        net2 = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 11))
        extra2 = torch.nn.Linear(5, 3)
        ((net2(x) - y).pow(2).mean() + extra2(xe).pow(2).mean()).backward()
        ref = [p.grad for p in list(net2.parameters()) + list(extra2.parameters())]
        err = max((a - b).abs().max().item() for a, b in zip(got, ref))

        # a second round must start from a clean slate (ranges are per round)
        for p in params:
            p._rg_grad.fill_(float(rank + 1))
        red.reduce()
        round2 = float(arena.flat_grad[arena.offsets[0]])

        rows = torch.full((3, 4), float(rank))
        gathered = all_gather_rows(rows)
        ok_gather = gathered.shape == (3 * world, 4) and all(
            torch.all(gathered[3 * r:3 * (r + 1)] == float(r)).item() for r in range(world))
        q.put((rank, err, round2, ok_gather, world_size()))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_and_gather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, round2, ok_gather, ws in res:
        assert err < 1e-6, (rank, err)                # mean over the global batch == sum of shard means / world
        assert round2 == 3.0, round2                  # 1 + 2 summed over the two ranks
        assert ok_gather
        assert ws == world


def test_reducer_is_a_noop_without_process_group():
    net = torch.nn.Linear(4, 4)
    arena = Arena(list(net.parameters()))
    opt = _FakeOptimizer(arena)
    red = GradReducer(opt)
    assert not red.active() and opt.grad_scale == 1.0
    red.reduce()                                      # must not raise
    assert world_size() == 1
    x = torch.ones(2, 3)
    assert all_gather_rows(x) is x


def test_arena_views_alias_parameters():
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    before = [p.detach().clone() for p in net.parameters()]
    arena = Arena(list(net.parameters()))
    for p, b, o in zip(net.parameters(), before, arena.offsets):
        assert torch.equal(p.detach(), b)
        assert p.data_ptr() == arena.flat[o:].data_ptr() and o % 64 == 0
        assert p._rg_grad.data_ptr() == arena.flat_grad[o:].data_ptr()
    arena.flat.mul_(2.0)                              # updating the arena updates the module
    for p, b in zip(net.parameters(), before):
        assert torch.equal(p.detach(), 2 * b)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_state_dict(sd)                           # in-place copy keeps the views
    assert next(net.parameters()).data_ptr() == arena.flat.data_ptr()
