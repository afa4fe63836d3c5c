"""Tape runtime: every network is ONE torch.autograd node whose forward/backward run a hand-written
program of HIP kernel launches (no per-op autograd dispatch, no tracing compiler).

An RGModule is a torch.nn.Module (parameters / buffers / state_dict / train()/eval() behave as in the
reference) that also implements
    tf(tape, *inputs)            -> outputs      forward program; pushes what backward needs
    tb(tape, *grad_outputs, need_dx=True) -> input grads; pops in reverse order, records parameter
                                                 gradients with tape.add_grad(param, g)
Calling the module (forward) wraps the whole program in `_NetFn`, so `loss.backward()`,
`optimizer.step()` and `.grad` work exactly like with the reference's modules.
"""
from __future__ import absolute_import

import contextlib

import torch
from torch import nn

from . import ops


class Tape(object):
    __slots__ = ("stack", "grads", "param_grad", "record", "needs_input")

    def __init__(self, param_grad=True, record=True, needs_input=None):
        self.stack = []
        self.grads = {}
        self.param_grad = param_grad
        self.record = record
        self.needs_input = needs_input        # per network input: does anything upstream want its gradient?

    def push(self, item):
        if self.record:
            self.stack.append(item)

    def pop(self):
        return self.stack.pop()

    def wants(self, p):
        return self.param_grad and p is not None and p.requires_grad

    def grad_out(self, p):
        """Arena view the FIRST gradient contribution of `p` may be written into directly (or None)."""
        v = getattr(p, "_rg_grad", None)
        if v is None or p.grad is not None or id(p) in self.grads:
            return None
        return v

    def add_grad(self, p, g):
        k = id(p)
        if g.shape != p.shape:
            g = g.view(p.shape)
        prev = self.grads.get(k)
        if prev is None:
            v = getattr(p, "_rg_grad", None)
            if v is not None and p.grad is None and g.data_ptr() != v.data_ptr():
                g = ops.axpby(g, None, 1.0, 0.0, out=v)           # move into the arena
            self.grads[k] = g
        else:
            ops.side_sync()      # `prev` / `g` may still be in flight on the side stream
            self.grads[k] = ops.axpby(prev, g, 1.0, 1.0, out=prev)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, n_in, *tensors):
        xs, params = tensors[:n_in], tensors[n_in:]
        tape = Tape(param_grad=len(params) > 0,
                    needs_input=[isinstance(x, torch.Tensor) and x.requires_grad for x in xs])
        ctx.set_materialize_grads(False)
        outs = net.tf(tape, *[x.detach() if isinstance(x, torch.Tensor) else x for x in xs])
        ctx.tape, ctx.net, ctx.params, ctx.n_in = tape, net, params, n_in
        return outs

    @staticmethod
    def backward(ctx, *dys):
        tape, n_in = ctx.tape, ctx.n_in
        if tape is None:
            raise RuntimeError("rg_hip: backward through a network a second time (the tape is freed after backward)")
        need = ctx.needs_input_grad[2:2 + n_in]
        ops.side_begin()
        try:
            dxs = ctx.net.tb(tape, *dys, need_dx=any(need))
        finally:
            ops.side_join()          # weight gradients launched on the side stream are complete from here on
        if not isinstance(dxs, (tuple, list)):
            dxs = (dxs,)
        dxs = tuple(dxs) + (None,) * (n_in - len(dxs))
        grads = []
        for p in ctx.params:
            g = tape.grads.get(id(p))
            v = getattr(p, "_rg_grad", None)
            if g is None or v is None:
                grads.append(g)
            elif g.data_ptr() == v.data_ptr():                    # written straight into the gradient arena
                p.grad = v
                grads.append(None)
            elif p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                ops.axpby(v, g, 1.0, 1.0, out=v)                  # accumulate across backward calls
                grads.append(None)
            else:
                grads.append(g)
        ctx.tape = None
        hook = getattr(ctx.net, "_rg_after_backward", None)
        if hook is not None and grads:      # e.g. start this network's gradient all-reduce while upstream runs
            hook()
        return (None, None) + tuple(d if n else None for d, n in zip(dxs, need)) + tuple(grads)


_AUTOGRAD_MT = __import__("os").environ.get("RG_AUTOGRAD_MT") == "1"      # A/B switch: leave the engine's worker thread on


def backward(loss, **kw):
    """loss.backward() with the network backward programs run on the CALLING thread: the autograd engine otherwise hands every
    node to its per-device worker thread, and the hand-off (plus the GIL ping-pong between the two threads while one of them
    enqueues kernels) costs ~1 ms per step on the small-kernel steps (tools/debug/host_cost.py: DPTN step 15.7 -> 14.8 ms).
    Streams behave as before: the engine restores each node's forward stream either way."""
    if _AUTOGRAD_MT:
        loss.backward(**kw)
        return
    with torch.autograd.set_multithreading_enabled(False):
        loss.backward(**kw)


def _param_list(net):
    """list(net.parameters()), built once per network: walking the module tree on every call costs milliseconds per
    step on the small-kernel networks, whose steps are host-bound.  The tree of a built network does not change; call
    invalidate_param_cache(net) after surgery on its sub-modules."""
    plist = net.__dict__.get("_rg_plist")
    if plist is None:
        plist = list(net.parameters())
        net.__dict__["_rg_plist"] = plist
    return plist


def invalidate_param_cache(net):
    for m in net.modules():
        m.__dict__.pop("_rg_plist", None)


def run(net, *xs):
    """Execute `net` as one autograd node (or plainly when no gradient is required)."""
    grad_on = torch.is_grad_enabled()
    frozen = getattr(net, "_rg_frozen", False)
    params = [] if (frozen or not grad_on) else [p for p in _param_list(net) if p.requires_grad]
    need_graph = grad_on and (len(params) > 0 or any(isinstance(x, torch.Tensor) and x.requires_grad for x in xs))
    if not need_graph:
        return net.tf(Tape(param_grad=False, record=False), *xs)
    if "_krsc_grouped" not in net.__dict__:
        from . import nn as _rnn
        _rnn.group_krsc(net)              # one re-layout launch per optimizer step for all filters of the network
    if net.__dict__.get("_rg_graph", False):
        # small-kernel networks: forward / backward programs replayed as two single-stream hipGraphs (rg_hip.netgraph)
        from . import netgraph
        if netgraph.ENABLED:
            return netgraph.call(net, xs, params, lambda: _NetFn.apply(net, len(xs), *xs, *params))
    return _NetFn.apply(net, len(xs), *xs, *params)


class RGModule(nn.Module):
    def tf(self, tape, *xs):
        raise NotImplementedError

    def tb(self, tape, *dys, **kw):
        raise NotImplementedError

    def forward(self, *xs):
        return run(self, *xs)


@contextlib.contextmanager
def no_param_grad(*nets):
    """Within the block the given networks propagate input gradients only (their parameters are
    treated as constants) — e.g. the discriminators inside the generator update."""
    old = [getattr(n, "_rg_frozen", False) for n in nets]
    for n in nets:
        n._rg_frozen = True
    try:
        yield
    finally:
        for n, o in zip(nets, old):
            n._rg_frozen = o
