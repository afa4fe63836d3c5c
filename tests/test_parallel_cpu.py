"""CPU (gloo, world_size 2): the data-parallel plumbing — in-place all-reduce of a gradient arena in chunks, partial
(per-network) asynchronous launches, 1/world folding into the optimizer's grad_scale, and the rank-ordered
all-gather used for the ClusterMemory update.  The kernels themselves are GPU-only; this covers the N>1 logic."""
import atexit
import datetime
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rg_hip.optim import Arena
from rg_hip.parallel import GradReducer, all_gather_rows, world_size


def _free_port():
    """rendezvous token of one test: the path of a fresh file for gloo's FileStore (no TCP port to collide on or to find in
    TIME_WAIT; the name is kept from the first version, which probed for a free port)"""
    import tempfile
    fd, path = tempfile.mkstemp(prefix="rg_gloo_")
    os.close(fd)
    os.unlink(path)                       # the store creates it
    atexit.register(lambda: os.path.exists(path) and os.unlink(path))
    return path


class _FakeOptimizer(object):
    def __init__(self, arena):
        self._arena = arena
        self.grad_scale = 1.0


def _worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=90))
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
        if p.is_alive():
            p.terminate()            # this exact child (a rank stuck in a collective must not outlive the test)
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


# ---- replicas start identical: rank 0's parameters + buffers are broadcast when the reducer becomes active --------
def _bcast_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=90))
    try:
        torch.manual_seed(100 + rank)                         # DIFFERENT replicas per rank, as a user script may build them
        net = torch.nn.Sequential(torch.nn.Linear(9, 6), torch.nn.BatchNorm1d(6), torch.nn.Linear(6, 2))
        net[1].running_mean.fill_(float(rank) + 0.5)
        net[1].num_batches_tracked.fill_(7 + rank)
        arena = Arena(list(net.parameters()))
        opt = _FakeOptimizer(arena)
        GradReducer(opt, bucket_mb=0.0001, modules=[net])     # tiny chunks: several broadcasts
        flat = arena.flat.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        same_params = all(torch.equal(g, gathered[0]) for g in gathered)
        q.put((rank, same_params, float(net[1].running_mean[0]), int(net[1].num_batches_tracked),
               float(next(net.parameters()).reshape(-1)[0]), float(arena.flat[0])))
    finally:
        dist.destroy_process_group()


def test_broadcast_makes_replicas_identical_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()            # this exact child (a rank stuck in a collective must not outlive the test)
        assert p.exitcode == 0
    for rank, same_params, rm, nbt, p0, a0 in res:
        assert same_params
        assert rm == 0.5 and nbt == 7                 # rank 0's buffers everywhere
        assert p0 == a0                               # module parameters are still views of the arena
    assert res[0][4] == res[1][4]


def _outside_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=90))
    try:
        net = torch.nn.Linear(4, 4)
        arena = Arena(list(net.parameters()))
        red = GradReducer(_FakeOptimizer(arena))
        net(torch.ones(2, 4)).sum().backward()        # stock autograd: .grad is NOT the arena view
        try:
            red.reduce()
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_reducer_raises_on_gradient_outside_arena_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_outside_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()            # this exact child (a rank stuck in a collective must not outlive the test)
        assert p.exitcode == 0
    for _, msg in res:
        assert "outside the optimizer's gradient arena" in msg, msg


# ---- ClusterMemory under data parallelism: gather in forward, identical ordered update on every rank -------------
def _cpu_cm_update(inputs, targets, features, momentum, hard=False, normalize_eps=False):
    """CPU stand-in for rg_cm_update with the reference's loop (CC/clustercontrast/models/cm.py:29-31)."""
    assert not hard
    for x, y in zip(inputs, targets):
        features[y] = momentum * features[y] + (1.0 - momentum) * x
        features[y] /= features[y].norm()
    return features


def _cm_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=90))
    try:
        from rg_hip import ops
        import clustercontrast.models.cm as CMM
        ops.linear_fwd = lambda x, w, bias=None: x @ w.t()          # CPU stand-ins for the HIP kernels (test only)
        ops.linear_dgrad = lambda dy, w: dy @ w
        ops.cm_update = _cpu_cm_update
        g = torch.Generator().manual_seed(3)
        B, D, K = 6, 16, 5
        bank0 = torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1)
        x_all = torch.nn.functional.normalize(torch.randn(B * world, D, generator=g), dim=1)
        y_all = torch.tensor([0, 3, 3, 1, 0, 3, 2, 2, 3, 0, 1, 3])[:B * world]       # repeated labels across ranks
        bank = bank0.clone()
        xs = x_all[B * rank:B * (rank + 1)].clone().requires_grad_(True)
        ys = y_all[B * rank:B * (rank + 1)]
        out = CMM.cm(xs, ys, bank, 0.2)
        out.sum().backward()
        # single-process reference on the global batch in rank order
        ref = _cpu_cm_update(x_all, y_all, bank0.clone(), 0.2)
        grad_ref = torch.ones(B, K) @ bank0
        q.put((rank, (bank - ref).abs().max().item(), (xs.grad - grad_ref).abs().max().item(), bank.numpy().tobytes()))   # bytes, not a tensor:
        # a tensor travels as a shared-memory handle the parent must fetch while this process is still alive
    finally:
        dist.destroy_process_group()


def test_cluster_memory_gather_and_ordered_update_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()            # this exact child (a rank stuck in a collective must not outlive the test)
        assert p.exitcode == 0
    for rank, bank_err, grad_err, _ in res:
        assert bank_err == 0.0, (rank, bank_err)      # every replica applied the same updates in the same order
        assert grad_err < 1e-6, (rank, grad_err)      # the input gradient used the PRE-update bank and stays local
    assert res[0][3] == res[1][3]


# ---- stage-bucketed all-reduce launched from inside the trunk's backward program (rg_hip.resnet_trunk.trunk_tb) ----------------
class _FakeBlock(object):
    """stands in for a bottleneck block on the CPU: its backward writes a rank- and tensor-dependent gradient straight into the
    arena views and records them on the tape, as the HIP weight-gradient kernels do"""

    def __init__(self, shapes, seed):
        g = torch.Generator().manual_seed(seed)
        self.ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
        self.seed = seed

    def parameters(self):
        return list(self.ps)

    def fill(self, tape, rank):
        for j, p in enumerate(self.ps):
            g = torch.Generator().manual_seed(1000 * self.seed + 10 * j + rank)
            p._rg_grad.copy_(torch.randn(p.shape, generator=g))
            tape.add_grad(p, p._rg_grad)

    def tb(self, tape, dy, dy_masked=False, mask_input=False):
        self.fill(tape, _FakeBlock.rank)
        return dy


class _FakeStem(_FakeBlock):
    @property
    def weight(self):
        return self.ps[0]


class _FakePool(object):
    def tb(self, tape, dy):
        return dy


def _stage_worker(rank, world, port, q):
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=90))
    try:
        from rg_hip import resnet_trunk
        from rg_hip.tape import Tape
        _FakeBlock.rank = rank
        conv1, bn1 = _FakeStem([(8, 3, 7, 7)], 1), _FakeBlock([(8,), (8,)], 2)
        layers = [[_FakeBlock([(8, 8, 1, 1), (8,), (8,), (8, 8, 3, 3)], 10 * li + b) for b in range(nb)]
                  for li, nb in ((1, 3), (2, 4), (3, 6), (4, 3))]
        mods = [conv1, bn1, None, _FakePool()] + layers
        params = conv1.parameters() + bn1.parameters() + [p for layer in layers for blk in layer for p in blk.parameters()]

        def run(bucketed):
            arena = Arena(params)
            for p in params:
                p.grad = None
            red = GradReducer(_FakeOptimizer(arena), bucket_mb=0.001)
            if bucketed:
                conv1.__dict__["_rg_stage_hook"] = red.reduce_stage
            else:
                conv1.__dict__.pop("_rg_stage_hook", None)
            tape = Tape(param_grad=True)
            # the stem's backward (conv_bn_tb) is a HIP program: a CPU stand-in with the same side effects
            old = resnet_trunk.rnn.conv_bn_tb
            resnet_trunk.rnn.conv_bn_tb = lambda tape, c, b, dy, need_dx=True: (c.fill(tape, rank), b.fill(tape, rank), dy)[2]
            try:
                resnet_trunk.trunk_tb(tape, mods, torch.zeros(1), need_dx=False)
            finally:
                resnet_trunk.rnn.conv_bn_tb = old
            in_flight = red.in_flight()
            for p in params:                                  # what tape._NetFn.backward does after the program returned
                p.grad = p._rg_grad
            red.reduce()
            return in_flight, arena.flat_grad.clone()

        n_plain, g_plain = run(False)
        n_stage, g_stage = run(True)
        q.put((rank, n_plain, n_stage, bool(torch.equal(g_plain, g_stage)), float(g_stage.abs().sum())))
    finally:
        dist.destroy_process_group()


def test_stage_bucketed_allreduce_inside_trunk_backward_world2():
    """bucketed (one launch per finished stage, from inside trunk_tb) == unbucketed (one reduce() after the backward) bit for bit,
    and the collectives of all five stages are in flight when trunk_tb returns"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stage_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0
    sums = set()
    for rank, n_plain, n_stage, same, s in res:
        assert n_plain == 0, n_plain                          # nothing is launched before the end without the hook
        assert n_stage >= 5, n_stage                          # layer4, layer3, layer2, layer1, stem (>= 3 asked for)
        assert same, rank
        sums.add(round(s, 3))
    assert len(sums) == 1                                     # both ranks hold the same reduced gradients
