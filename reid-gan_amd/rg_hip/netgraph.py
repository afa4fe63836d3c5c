"""Per-network launch programs: the forward and backward programs of ONE network, captured once into two single-stream hipGraphs
and replayed with one `hipGraphLaunch` each (0.4 us of host time per kernel node instead of 12-14 us per Python-issued launch).

Why per network and single-stream: round 2 captured the WHOLE step, multi-stream, and found the replay no cheaper than eager
launches and one capture (the two-stream FD-GAN step) faulting inside the runtime; `tools/debug/graph_probe.py` shows that a
single-stream capture of one network forward replays with 65 us of host time for ~150 nodes.  So the unit here is the network
program (`net.tf` / `net.tb` of rg_hip.tape), the capture runs with the weight-gradient side stream off, and everything between
networks (losses, optimizers, collectives, ATen glue) stays eager Python.

A network opts in with `net._rg_graph = True` (the small-kernel dual_gan generators / discriminators: their steps are host-bound).
`tape.run` sends its calls here; a call is replayed when a record for its key exists and is free, captured after `WARMUP` eager calls
with that key, and run eagerly (the ordinary `_NetFn`) otherwise.  The key is everything the recorded launch sequence depends on:
training flag, input shapes and requires_grad pattern, whether parameter gradients are produced and whether they are the first
contribution of the step.  What a replay does NOT re-run is the Python of the programs, so its host-side effects are either forced to
be unconditional during capture (filter re-layout / fp8 re-quantisation caches refresh every time: `ops.CAPTURING`), re-applied per
replay (BatchNorm `num_batches_tracked` bookkeeping), or a reason not to graph the network (active Dropout draws a seed from the host
generator per call; an active launch profiler needs the eager launches).

Memory: a record's inputs are copied into static buffers, its intermediates / outputs / saved activations live in the record's private
graph pool (torch.cuda.graph), exactly as they would stay resident for the backward in eager mode; a network called k times before
its backward runs (a discriminator on real and fake batches) holds k records.
"""
from __future__ import absolute_import

import os

import torch

from . import ops
from .tape import Tape

ENABLED = os.environ.get("RG_NET_GRAPHS", "1") != "0"
WARMUP = 2          # eager calls per key before capture (first-use calibration of fp8 scales, lazily built caches)
MAX_RECORDS = 4     # records per key (calls of one network between its backward passes)


class _Record(object):
    __slots__ = ("pool", "fwd", "bwd", "static_in", "outs", "tape", "bn_deltas", "busy", "dead", "static_dys", "dy_none",
                 "dxs", "assign", "extra_grads", "need", "n_out", "gen", "bwd_first")

    def __init__(self):
        self.fwd = self.bwd = None
        self.busy = self.dead = False
        self.gen = 0


def _bn_layers(net):
    from .nn import _BatchNorm
    return [m for m in net.modules() if isinstance(m, _BatchNorm)]


# hipStreamCaptureModeThreadLocal: only the CAPTURING thread is held to the capture rules.  With the default ("global") a HIP call
# from any other thread while a capture is open invalidates the capture — and the process group's watchdog thread polls the events
# of in-flight collectives (hipEventQuery) every few milliseconds: a gradient all-reduce still in flight when a network's program is
# captured (the joint step launches the encoder's and the discriminators' reductions before the generator's backward) then aborts
# the capture and, through the watchdog's own exception, the process.
_CAPTURE_MODE = "thread_local"


def graphable(net):
    """no active Dropout (it draws a seed from the host generator per call; a replay would repeat the captured one)"""
    from .nn import Dropout
    for m in net.modules():
        if isinstance(m, Dropout) and m.training and m.p > 0.0:
            return False
    return True


def _key(net, xs, params):
    sig = []
    for x in xs:
        if isinstance(x, torch.Tensor):
            sig.append((tuple(x.shape), x.dtype, bool(x.requires_grad)))
        else:
            sig.append(("const", x))
    first = params[0].grad is None if params else None
    extra = net._rg_key() if hasattr(net, "_rg_key") else None
    # first / last parameter addresses: a relocated parameter set (a new optimizer arena, .to()) must not meet old records
    where = (params[0].data_ptr(), params[-1].data_ptr()) if params else None
    return (net.training, tuple(sig), len(params), first, extra, where)


class _capture_mode(object):
    """single stream, unconditional cache refreshes"""

    def __enter__(self):
        self.side = ops._SIDE["on"]
        if not ops._GRAPH_SIDE:
            ops._SIDE["on"] = False
        ops.CAPTURING[0] += 1
        ops.CAPTURE_GEN[0] += 1

    def __exit__(self, *a):
        ops.CAPTURING[0] -= 1
        ops._SIDE["on"] = self.side


def _capture_forward(net, xs, params):
    rec = _Record()
    rec.static_in = [x.detach().clone() if isinstance(x, torch.Tensor) else x for x in xs]
    rec.need = [isinstance(x, torch.Tensor) and x.requires_grad for x in xs]
    rec.tape = Tape(param_grad=len(params) > 0, needs_input=list(rec.need))
    bns = _bn_layers(net)
    before = [b.__dict__.get("_nbt_pending", 0) for b in bns]
    g = torch.cuda.CUDAGraph()
    try:
        with _capture_mode():
            with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                outs = net.tf(rec.tape, *rec.static_in)
    except Exception:
        # the aborted program's Python side effects must not survive it: none of its kernels ran.  Cache keys are not stamped while
        # capturing (nn._KrscCache / KrscGroup / lowp.F8Layer); the BatchNorm step counters are put back here.
        for b, n0 in zip(bns, before):
            b.__dict__["_nbt_pending"] = n0
        raise
    rec.pool = g.pool()
    rec.bn_deltas = [(b, b.__dict__.get("_nbt_pending", 0) - n0) for b, n0 in zip(bns, before)
                     if b.__dict__.get("_nbt_pending", 0) != n0]
    rec.fwd = g
    rec.outs = outs
    rec.n_out = len(outs) if isinstance(outs, (tuple, list)) else 1
    g.replay()                      # the capture executed nothing: this produces the outputs of THIS call
    return rec


def _replay_forward(rec, xs):
    for s, x in zip(rec.static_in, xs):
        if isinstance(s, torch.Tensor):
            s.copy_(x)
    rec.fwd.replay()
    for b, d in rec.bn_deltas:
        b.__dict__["_nbt_pending"] = b.__dict__.get("_nbt_pending", 0) + d


def _tb_and_grads(net, tape, dys, need, params):
    """the body of tape._NetFn.backward: run the backward program and settle the parameter gradients (arena views are assigned,
    everything else is returned); -> (dxs, grads per parameter, [(param, arena view)] assignments made)"""
    ops.side_begin()                  # no-op unless RG_GRAPH_SIDE keeps the weight-gradient side stream inside the capture
    try:
        dxs = net.tb(tape, *dys, need_dx=any(need))
    finally:
        ops.side_join()               # fork / join nodes of the graph: weight gradients are complete from here on
    if not isinstance(dxs, (tuple, list)):
        dxs = (dxs,)
    dxs = tuple(dxs) + (None,) * (len(need) - len(dxs))
    grads, assign = [], []
    for p in params:
        g = tape.grads.get(id(p))
        v = getattr(p, "_rg_grad", None)
        if g is None or v is None:
            grads.append(g)
        elif g.data_ptr() == v.data_ptr():
            p.grad = v
            assign.append((p, v))
            grads.append(None)
        elif p.grad is not None and p.grad.data_ptr() == v.data_ptr():
            ops.axpby(v, g, 1.0, 1.0, out=v)
            grads.append(None)
        else:
            grads.append(g)
    return dxs, grads, assign


def _grad_pattern(params):
    """which parameters already hold a gradient (assign-or-accumulate state of the whole set, not of params[0] alone)"""
    return tuple(p.grad is None for p in params) if params else None


class _Sentinel(object):
    """frees the record when the autograd node of ITS use dies without a backward (outputs dropped, graph never differentiated);
    a node of an earlier use that is collected late (its outputs were still referenced) must not free the current use"""
    __slots__ = ("rec", "gen")

    def __init__(self, rec):
        self.rec, self.gen = rec, rec.gen

    def __del__(self):
        if self.rec.gen == self.gen:
            self.rec.busy = False


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, rec, n_in, *tensors):
        ctx.set_materialize_grads(False)
        ctx.net, ctx.rec, ctx.n_in, ctx.params = net, rec, n_in, tensors[n_in:]
        ctx.sentinel = _Sentinel(rec)
        outs = rec.outs
        if isinstance(outs, (tuple, list)):
            return tuple(o.detach() if isinstance(o, torch.Tensor) else o for o in outs)
        return outs.detach()

    @staticmethod
    def backward(ctx, *dys):
        net, rec, n_in, params = ctx.net, ctx.rec, ctx.n_in, ctx.params
        need = ctx.needs_input_grad[3:3 + n_in]
        none_pat = tuple(d is None for d in dys)
        try:
            if rec.bwd is None:
                rec.static_dys = [None if d is None else d.detach().clone() for d in dys]
                rec.dy_none = none_pat
                rec.bwd_first = _grad_pattern(params)
                g = torch.cuda.CUDAGraph()
                stack0 = list(rec.tape.stack)             # an aborted capture has popped (and recorded) without executing anything
                try:
                    with _capture_mode():
                        with torch.cuda.graph(g, pool=rec.pool, capture_error_mode=_CAPTURE_MODE):
                            dxs, grads, assign = _tb_and_grads(net, rec.tape, rec.static_dys, need, params)
                except Exception as e:
                    # a backward program that cannot be captured: run THIS backward eagerly over the record's tape (the forward
                    # replay produced its saved activations for real) and retire the record; the key runs eagerly from now on
                    import warnings
                    warnings.warn("rg_hip.netgraph: the backward program of %s is not capturable (%s: %s); running it eagerly"
                                  % (type(net).__name__, type(e).__name__, str(e)[:200]))
                    torch.cuda.synchronize()
                    rec.dead = True
                    for ent in (net.__dict__.get("_rg_graphs") or {}).values():
                        if rec in ent["records"]:
                            ent["bad"] = True
                    rec.tape.stack[:] = stack0
                    rec.tape.grads = {}
                    dxs, grads, assign = _tb_and_grads(net, rec.tape, list(dys), need, params)
                    rec.tape = None
                    hook = getattr(net, "_rg_after_backward", None)
                    if hook is not None and params:
                        hook()
                    return (None, None, None) + tuple(d if (d is not None and n) else None for d, n in zip(dxs, need)) + tuple(grads)
                rec.bwd, rec.dxs, rec.extra_grads, rec.assign = g, dxs, grads, assign
                rec.tape = None
                g.replay()
            else:
                if none_pat != rec.dy_none or _grad_pattern(params) != rec.bwd_first:
                    raise RuntimeError("rg_hip.netgraph: the pattern of output gradients / the accumulate-or-assign state of the "
                                       "parameter gradients changed between steps; set net._rg_graph = False for this network")
                for s, d in zip(rec.static_dys, dys):
                    if s is not None:
                        s.copy_(d)
                rec.bwd.replay()
                for p, v in rec.assign:
                    p.grad = v
        finally:
            rec.busy = False
        hook = getattr(net, "_rg_after_backward", None)
        if hook is not None and params:
            hook()
        # gradients of leaf inputs and of parameters outside an optimizer arena are handed to autograd, which may keep the tensor
        # (AccumulateGrad steals it): they must not alias the graph's static buffers, which the next replay overwrites
        dxs = tuple(d.detach().clone() if (d is not None and n) else None for d, n in zip(rec.dxs, need))
        grads = tuple(None if g is None else g.detach().clone() for g in rec.extra_grads)
        return (None, None, None) + dxs + grads


def call(net, xs, params, eager):
    """replay / capture / eager dispatch for one network call that needs a graph (tape.run decides that)"""
    st = net.__dict__.get("_rg_graphs")
    if st is None:
        st = net.__dict__["_rg_graphs"] = {}
    key = _key(net, xs, params)
    ent = st.get(key)
    if ent is None:
        ent = st[key] = {"calls": 0, "records": [], "bad": not graphable(net)}
    ent["calls"] += 1
    if ent["bad"] or ent["calls"] <= WARMUP or ops._PROFILING[0]:
        return eager()
    rec = None
    for r in ent["records"]:
        if not r.busy and not r.dead:
            rec = r
            break
    if rec is None:
        if len(ent["records"]) >= MAX_RECORDS:
            return eager()
        try:
            rec = _capture_forward(net, xs, params)
        except Exception as e:          # a program that cannot be captured (host read-back, ...) stays eager, loudly once
            ent["bad"] = True
            import warnings
            warnings.warn("rg_hip.netgraph: %s is not capturable (%s: %s); running it eagerly"
                          % (type(net).__name__, type(e).__name__, str(e)[:200]))
            torch.cuda.synchronize()
            return eager()
        ent["records"].append(rec)
    else:
        _replay_forward(rec, xs)
    rec.busy = True
    rec.gen += 1
    return _GraphedFn.apply(net, rec, len(xs), *xs, *params)


def release(net):
    """drop every record of `net` (after surgery on its modules or parameters)"""
    net.__dict__.pop("_rg_graphs", None)


def stats(net):
    """{key: (calls, records, capturable)} — what the tests and bench.py report"""
    st = net.__dict__.get("_rg_graphs") or {}
    return {k: (e["calls"], len(e["records"]), not e["bad"]) for k, e in st.items()}
