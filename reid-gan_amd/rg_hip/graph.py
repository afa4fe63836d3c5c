"""hipGraph capture of a whole training step.

The step drivers enqueue 1 000 - 1 600 kernel launches per step from Python (tape programs -> ctypes -> hipLaunchKernel):
25 - 45 ms of host time, which is the whole step for the small-kernel configurations.  `CapturedStep(fn)` runs `fn` eagerly
for a few warm-up calls (allocator pools, workspaces, caches, first-step branches of the optimizers settle), then captures
ONE call into a hipGraph (torch.cuda.graph: stream capture in global mode, so the launches the autograd worker threads and
the side / auxiliary streams make are captured too, with their event dependencies) and replays it from then on: one
hipGraphLaunch per step.

What makes a step capturable here (kernel arguments are frozen at capture time):
  * per-step scalars live in device memory: Adam's bias corrections (`rg_adam_advance` / `rg_adam_step_dev`), the dropout
    clock (`rg_dropout_clocked`, advanced by this module once per replay), fp8 scaling states;
  * no host<->device synchronisation and no pageable host->device copy inside `fn` (loss weights are cached device tensors,
    losses are read by the caller after the replay);
  * inputs are read from the SAME device tensors on every call (the caller copies new data into them);
  * torch's CUDA generator is graph-aware (noise `z`), Python's `random` is not: FD-GAN's label smoothing / flip draws are
    frozen at capture, so `CapturedStep` refuses a model with `smooth_label` set.
Host-side bookkeeping that the captured code advanced once (optimizer step counts, arena epochs that invalidate cached filter
layouts, BatchNorm `num_batches_tracked`) is advanced by the same amounts after every replay, so eager code that runs between
replays (evaluation, checkpointing) sees consistent state.

Results: a replayed step launches exactly the kernels of the eager step in the same order with the same arguments — the
outputs are bit-identical (tests/test_graph_gpu.py).
"""
from __future__ import absolute_import

import torch

from . import ops


def _counters():
    """{key: (get, set)} accessors of every host-side integer the step code advances; the key identifies the counter's owner, so
    two snapshots can be compared even when unrelated objects were garbage-collected in between"""
    from . import nn as rnn
    from . import optim as roptim
    acc = {}

    def attr(obj, name):
        acc[(id(obj), name)] = (lambda: getattr(obj, name), lambda v: setattr(obj, name, v))

    def item(seq, i):
        acc[(id(seq), i)] = (lambda: seq[i], lambda v: seq.__setitem__(i, v))

    for a in list(roptim.ARENAS):
        attr(a, "epoch")
    for o in list(roptim.OPTIMIZERS):
        steps = getattr(o, "_steps", None)
        if steps is not None:
            for i in range(len(steps)):
                item(steps, i)
        for c in getattr(o, "_clocks", {}).values():
            item(c, 1)
            item(c, 2)
    for bn in list(rnn.BN_LAYERS):
        d = bn.__dict__
        d.setdefault("_nbt_pending", 0)
        item(d, "_nbt_pending")
    item(rnn.WEIGHT_EPOCH, 0)
    return acc


class CapturedStep(object):
    """callable: eager for `warmup` calls, then one capture + replay per call.  `fn` takes no arguments and returns tensors
    (or nothing); the returned tensors are the graph's output buffers, refreshed by every replay."""

    def __init__(self, fn, warmup=3, device=None, pool=None):
        self.fn, self.warmup, self.calls = fn, int(warmup), 0
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.graph, self.out, self.deltas, self.pool = None, None, None, pool
        self.replays = 0

    @property
    def captured(self):
        return self.graph is not None

    def __call__(self):
        if self.graph is None:
            if self.calls < self.warmup:
                self.calls += 1
                return self.fn()
            self._capture()
            self.graph.replay()              # the capture itself executed nothing; the bookkeeping it advanced is this step's
        else:
            ops.check_not_profiling()
            self.graph.replay()
            for (get, set_), d in self.deltas:
                set_(get() + d)
        self.replays += 1
        return self.out

    def _capture(self):
        ops.check_not_profiling()
        torch.cuda.synchronize(self.device)
        acc = _counters()
        before = {k: g() for k, (g, _) in acc.items()}
        self.graph = torch.cuda.CUDAGraph()
        kw = {} if self.pool is None else {"pool": self.pool}
        with torch.cuda.graph(self.graph, **kw):
            ops.advance_step_clock(self.device)
            self.out = self.fn()
        # a counter that did not exist before the capture (a lazily built clock, an optimizer or BatchNorm layer constructed inside
        # the step) has no known per-step increment
        if any(k not in acc for k in _counters()):
            raise RuntimeError("CapturedStep: host-side counters appeared during the capture (an optimizer or BatchNorm layer was "
                               "built inside the step, or the warm-up was too short for a lazily created clock); raise `warmup`")
        self.deltas = []
        for k, (g, s_) in acc.items():
            d = g() - before[k]
            if d:
                self.deltas.append(((g, s_), d))
        torch.cuda.synchronize(self.device)
