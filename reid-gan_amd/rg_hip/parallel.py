"""Data parallelism, MI355X style: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference's only multi-GPU mechanism is single-process torch.nn.DataParallel (FD/fdgan/model.py:67-70,
CC/examples/cluster_contrast_gan_train_usl_infomap.py:197), which re-broadcasts every parameter on every
forward call and reduces gradients onto GPU 0.  Here every rank owns a full replica; after (and
overlapped with) each backward pass the gradient ARENA of an optimizer (rg_hip.optim.Arena.flat_grad) is
all-reduced in place — no bucket copy-in/copy-out because weight gradients were written contiguously —
in chunks of `bucket_mb` on the process group's side stream, and the 1/world averaging is folded into
the fused optimizer kernel (`grad_scale`).

`DataParallel` is an API shim: it keeps the `.module` attribute and the `module.` state_dict prefix the
reference's scripts and checkpoints rely on (FD/train.py:57, FD/fdgan/networks.py:51-55).
"""
from __future__ import absolute_import

import os

import torch
import torch.distributed as dist
from torch import nn


class DataParallel(nn.Module):
    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        super(DataParallel, self).__init__()
        self.module = module

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)


def init_process_group(rank, world, local_rank, **kw):
    """`dist.init_process_group("nccl", ...)` for one process per GPU.  The collectives' stream stays in the DEFAULT-priority pool, on
    purpose: HIP streams of one priority share `GPU_MAX_HW_QUEUES` (4) hardware queues and the step already drives four streams
    (main, weight-gradient side stream, FD-GAN's auxiliary stream and its side stream); a FIFTH active hardware queue — a
    high-priority collective stream, or GPU_MAX_HW_QUEUES=8 — makes the queue scheduler time-slice them and every cross-stream
    dependency then waits for a slice: measured on one rank, config 2, 35.4 -> 46.9 ms (high-priority stream) and 56-66 ms
    (8 queues) per step (`tools/debug/rccl_ab.sh`, `profiles/r04_hw_queues.txt`).  RG_NCCL_HIGH_PRIO=1 asks for the high-priority
    stream anyway (A/B)."""
    opts = None
    if os.environ.get("RG_NCCL_HIGH_PRIO", "0") == "1":
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    return dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                   pg_options=opts, **kw)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class GradReducer(object):
    """All-reduce (sum) of an optimizer's gradient arena; averaging happens in the optimizer kernel."""

    def __init__(self, optimizer, bucket_mb=64, group=None, modules=None, broadcast=True):
        """`modules`: the networks whose parameters live in the optimizer's arena; their buffers (BatchNorm running
        statistics, spectral-norm u / v) are broadcast from rank 0 together with the parameter arena when the reducer
        becomes active, so replicas start identical whatever seed each rank built its networks with (the reference's
        DataParallel re-broadcasts rank 0's parameters and buffers on every forward call)."""
        self.optimizer = optimizer
        self.arena = getattr(optimizer, "_arena", None)
        self.group = group
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self._pending = []
        self._done = []          # element ranges already launched in this round
        if self.active():
            optimizer.grad_scale = 1.0 / world_size()
            if broadcast:
                self.broadcast_state(modules)

    def broadcast_state(self, modules=None):
        """rank 0's parameter arena (one collective per `bucket_mb` chunk) and the given modules' float / integer
        buffers to every rank."""
        if not self.active():
            return
        flat = self.arena.flat
        pos = 0
        while pos < flat.numel():
            end = min(flat.numel(), pos + self.bucket_elems)
            dist.broadcast(flat[pos:end], src=0, group=self.group)
            pos = end
        if modules is None:
            return
        if isinstance(modules, nn.Module):
            modules = [modules]
        seen = set()
        for m in modules:
            for b in m.buffers():
                if b is None or id(b) in seen or b.numel() == 0:
                    continue
                seen.add(id(b))
                dist.broadcast(b.data, src=0, group=self.group)

    def active(self):
        # RG_FORCE_REDUCE=1 exercises the collective path on a single rank (RCCL init, side stream, waits)
        forced = os.environ.get("RG_FORCE_REDUCE") == "1" and dist.is_available() and dist.is_initialized()
        return self.arena is not None and (world_size() > 1 or forced)

    def _ranges_for(self, params):
        a = self.arena
        if params is None:
            idx = range(len(a.params))
        else:
            ids = {id(p) for p in params}
            idx = [i for i, p in enumerate(a.params) if id(p) in ids]
        out, cur = [], None
        for i in idx:
            p, o = a.params[i], a.offsets[i]
            if p.grad is None:
                cur = None
                continue
            if p.grad.data_ptr() != p._rg_grad.data_ptr():
                # a gradient that is not the arena view would silently miss the all-reduce: replicas would diverge
                raise RuntimeError("GradReducer: parameter %d (shape %s) has a gradient outside the optimizer's "
                                   "gradient arena; write it through p._rg_grad (rg_hip.optim._have() does this "
                                   "for stock-autograd gradients before the step)" % (i, tuple(p.shape)))
            end = o + p.numel()
            if any(s <= o < e for s, e in self._done):
                cur = None
                continue
            if cur is not None and o - cur[1] < 64 + 1 and o >= cur[1]:
                cur[1] = end
            else:
                cur = [o, end]
                out.append(cur)
        return [(s, e) for s, e in out]

    def reduce_async(self, params=None):
        """Launch the all-reduce of the gradients of `params` (default: everything not launched yet).
        Returns immediately; the collective runs on the process group's stream after the work already
        queued on the current stream."""
        if not self.active():
            return
        for s, e in self._ranges_for(params):
            self._done.append((s, e))
            pos = s
            while pos < e:
                end = min(e, pos + self.bucket_elems)
                work = dist.all_reduce(self.arena.flat_grad[pos:end], op=dist.ReduceOp.SUM, group=self.group,
                                       async_op=True)
                self._pending.append(work)
                pos = end

    def reduce_stage(self, tape, params):
        """Launch the all-reduce of one finished STAGE of a backward program that is still running (rg_hip.resnet_trunk.trunk_tb
        calls this through `_rg_stage_hook`).  `p.grad` is not assigned yet at that point: a parameter takes part when the gradient
        the program recorded for it (`tape.grads`) IS its arena view and is its first contribution of the step — exactly the
        gradients that are final; anything else (accumulated contributions, gradients outside the arena) is left to the
        `reduce()` / `reduce_async()` at the end of the pass, which skips the ranges launched here.
        The collective is ordered behind the weight-gradient side stream of the running backward session as well as behind the
        calling stream (the stage's filter gradients run there), without making the calling stream wait for either."""
        if not self.active():
            return 0
        a = self.arena
        index = getattr(a, "_rg_index", None)
        if index is None:
            index = a._rg_index = {id(p): i for i, p in enumerate(a.params)}
        offs = []
        for p in params:
            i = index.get(id(p))
            g = tape.grads.get(id(p)) if i is not None else None
            v = getattr(p, "_rg_grad", None)
            if g is None or v is None or p.grad is not None or g.data_ptr() != v.data_ptr():
                continue
            offs.append((a.offsets[i], a.offsets[i] + p.numel()))
        offs.sort()
        ranges = []
        for s, e in offs:
            if any(ds <= s < de for ds, de in self._done):
                continue
            if ranges and 0 <= s - ranges[-1][1] <= 64:
                ranges[-1][1] = e
            else:
                ranges.append([s, e])
        if not ranges:
            return 0
        launched = 0
        with _behind_side_stream(a.flat_grad):
            for s, e in ranges:
                self._done.append((s, e))
                pos = s
                while pos < e:
                    end = min(e, pos + self.bucket_elems)
                    self._pending.append(dist.all_reduce(a.flat_grad[pos:end], op=dist.ReduceOp.SUM, group=self.group,
                                                         async_op=True))
                    launched += 1
                    pos = end
        return launched

    def in_flight(self):
        return len(self._pending)

    def wait(self):
        for w in self._pending:
            w.wait()
        self._pending = []
        self._done = []

    def reduce(self):
        self.reduce_async(None)
        self.wait()


class _behind_side_stream(object):
    """Context: collectives launched inside are ordered behind BOTH the current stream and the weight-gradient side stream of the
    backward session running on it (rg_hip.ops.side_call), by issuing them from the side stream after it has been made to wait for
    an event of the current stream.  CPU tensors / no open session: a no-op."""

    def __init__(self, tensor):
        self.ctx = None
        if tensor.is_cuda:
            from . import ops
            main, sess = ops._side_session(create=False)
            if sess is not None and sess.depth > 0 and sess.used:
                ev = torch.cuda.Event()
                ev.record(main)
                sess.stream.wait_event(ev)
                self.ctx = torch.cuda.stream(sess.stream)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def attach_stage_hooks(reducer, *modules):
    """Give every ResNet trunk under `modules` the stage hook of `reducer`: its gradient all-reduce is then launched stage by stage
    from inside the trunk's backward program (layer4 first) instead of after it.  Returns the number of trunks found.  Harmless on
    one rank (the hook returns at once while the reducer is inactive)."""
    from .resnet_trunk import TVResNet
    n = 0
    if os.environ.get("RG_STAGE_BUCKETS", "1") == "0":       # A/B switch: one reduction per arena after the backward pass (round 3)
        return 0
    for root in modules:
        root = getattr(root, "module", root)
        for m in root.modules():
            mods = None
            if isinstance(m, TVResNet):
                mods = m.trunk_modules()
            elif hasattr(m, "base") and isinstance(getattr(m, "base"), nn.Sequential) and len(m.base) >= 8:
                mods = list(m.base)                   # clustercontrast ResNet: `base` is the trunk as a Sequential
            if mods is not None and hasattr(mods[0], "weight"):
                mods[0].__dict__["_rg_stage_hook"] = reducer.reduce_stage
                n += 1
    return n


class GatherHandle(object):
    """In-flight all_gather_rows_async: `wait()` returns the gathered tensor (rank order)."""

    def __init__(self, out, work):
        self.out, self.work = out, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out


def all_gather_rows_async(x, group=None):
    """Issue the gather now (it runs on the process group's stream behind the work already queued on the current
    stream) and collect it later: ClusterMemory issues it in forward and waits right before the centroid update in
    backward, so the collective overlaps the loss and the whole encoder backward set-up."""
    w = world_size()
    if w == 1:
        return GatherHandle(x, None)
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    work = dist.all_gather_into_tensor(out, x.contiguous(), group=group, async_op=True)
    return GatherHandle(out, work)


def all_gather_rows(x, group=None):
    """[B_local, D] -> [world * B_local, D] in rank order (features / labels for the ClusterMemory update,
    so that every rank applies the identical sequential centroid update)."""
    w = world_size()
    if w == 1:
        return x
    out = torch.empty((w * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out
