"""Fused Adam / SGD over flat parameter arenas.

`Adam` / `SGD` take the same arguments as torch.optim.Adam / torch.optim.SGD (the reference builds those:
FD/fdgan/model.py:100-125, CC/examples/cluster_contrast_gan_train_usl_infomap.py:281-284) and are
torch.optim.Optimizer subclasses, so schedulers, param_groups, zero_grad() and state_dict() behave the same.

At construction every parameter group is moved into ONE contiguous fp32 arena (p.data becomes a view)
with a matching gradient arena; the tape runtime writes weight gradients straight into those views
(`p._rg_grad`), so a step is one kernel launch per run of consecutive parameters that received a
gradient (normally one per group) instead of ~10 element-wise launches per tensor, and the
data-parallel reducer can all-reduce the same flat buffer without bucket copies.
"""
from __future__ import absolute_import

import torch
from torch.optim import Optimizer

from . import ops

import struct
import weakref

_ALIGN = 64          # elements; keeps every view 256-byte aligned (float4 kernels, RCCL)
ARENAS = weakref.WeakSet()          # every live arena / optimizer: rg_hip.graph advances their host-side counters per replay
OPTIMIZERS = weakref.WeakSet()


class Arena(object):
    """Flat storage for a list of parameters (values + gradients)."""

    def __init__(self, params):
        params = [p for p in params]
        if not params:
            raise ValueError("Arena: empty parameter list")
        dev = params[0].device
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("Arena: all parameters must be fp32 on one device")
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.size = off
        ARENAS.add(self)
        self.epoch = 0           # bumped by every optimizer step: invalidates cached filter re-layouts
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(params, self.offsets):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p._rg_grad = self.flat_grad[o:o + n].view(p.shape)
            p._rg_arena = self
            p._rg_offset = o

    def runs(self):
        """[(start, end)] element ranges covering consecutive parameters whose .grad is the arena view."""
        out, cur = [], None
        for p, o in zip(self.params, self.offsets):
            ok = p.grad is not None and p.grad.data_ptr() == p._rg_grad.data_ptr()
            end = o + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if ok:
                cur = [o, end] if cur is None else [cur[0], end]
            elif cur is not None:
                out.append(tuple(cur))
                cur = None
        if cur is not None:
            out.append(tuple(cur))
        return out


def flatten_params(params):
    return Arena(list(params))


class _FlatOptimizer(Optimizer):
    """One arena for ALL parameter groups (the ReID optimizer of the reference has one group per tensor,
    SURVEY §9.12); launches are split only where hyper-parameters or step counts differ."""

    def _setup(self):
        ps = [p for g in self.param_groups for p in g["params"]]
        owned = [getattr(p, "_rg_arena", None) for p in ps]
        if any(a is not None for a in owned):
            a = owned[0]
            if a is None or any(o is not a for o in owned) or len(a.params) != len(ps) or \
                    any(x is not y for x, y in zip(a.params, ps)):
                raise ValueError("rg_hip.optim: parameters already belong to a different arena")
            self._arena = a
        else:
            self._arena = Arena(ps)
        self._group_of = [gi for gi, g in enumerate(self.param_groups) for _ in g["params"]]

    def _have(self):
        """indices of the parameters that received a gradient; gradients set by hand / by stock autograd are copied into
        the arena first.  (`p.grad is view` is an identity test on the Python wrapper the tape assigned — no data_ptr calls:
        this loop runs while the GPU queue is nearly empty, so its host time is GPU idle time.)"""
        a = self._arena
        views = getattr(a, "_views", None)
        if views is None:
            views = a._views = [p._rg_grad for p in a.params]
        have = []
        for i, p in enumerate(a.params):
            g = p.grad
            if g is None:
                continue
            if g is not views[i]:
                if g.data_ptr() != views[i].data_ptr():
                    views[i].copy_(g)
                p.grad = views[i]
            have.append(i)
        return have

    def _segments(self, key_of):
        """maximal runs of consecutive parameters that have a gradient and share key_of(i): [a0, a1, [i...], key].
        The run structure is cached: it only depends on WHICH parameters have a gradient and on the per-group
        hyper-parameters, both normally identical from step to step."""
        a = self._arena
        have = self._have()
        gkeys = tuple(key_of(None, gi) for gi in range(len(self.param_groups)))
        sig = (tuple(have), gkeys)
        cache = getattr(self, "_seg_cache", None)
        if cache is not None and cache[0] == sig and self._uniform_state(cache[1]):
            return cache[1]
        ends = getattr(a, "_ends", None)
        if ends is None:
            ends = a._ends = [o + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN for p, o in zip(a.params, a.offsets)]
        segs, cur, last = [], None, -2
        for i in have:
            k = key_of(i, self._group_of[i])
            if cur is not None and last == i - 1 and cur[3] == k:
                cur[1] = ends[i]
                cur[2].append(i)
            else:
                cur = [a.offsets[i], ends[i], [i], k]
                segs.append(cur)
            last = i
        self._seg_cache = (sig, segs)
        return segs

    def zero_grad(self, set_to_none=True):
        for p in self._arena.params:
            p.grad = None

    # ---- checkpointing: torch.optim's per-parameter layout built from the flat moment buffers ------------------
    def state_dict(self):
        """{'state': {index: {...per-parameter tensors...}}, 'param_groups': [...]} as torch.optim returns it (the flat
        moment buffers are sliced per parameter), so a resume restores moments / momentum and step counts."""
        a = self._arena
        packed = super(_FlatOptimizer, self).state_dict()
        state = {}
        for i, (p, o) in enumerate(zip(a.params, a.offsets)):
            entry = self._export_param_state(i, o, p)
            if entry is not None:
                state[i] = entry
        packed["state"] = state
        return packed

    def load_state_dict(self, state_dict):
        a = self._arena
        state = state_dict.get("state", {})
        super(_FlatOptimizer, self).load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        self._reset_state()                  # parameters absent from the loaded state start from zero, as in torch.optim
        for k, entry in state.items():
            i = int(k)
            if not 0 <= i < len(a.params):
                raise ValueError("rg_hip.optim: optimizer state refers to parameter %d of %d" % (i, len(a.params)))
            self._import_param_state(i, a.offsets[i], a.params[i], entry)
        self._seg_cache = None


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super(Adam, self).__init__(params, defaults)
        self.grad_scale = grad_scale
        self._setup()
        self._m = torch.zeros_like(self._arena.flat)
        self._v = torch.zeros_like(self._arena.flat)
        self._steps = [0] * len(self._arena.params)
        # device-side clocks {step, beta1^step, beta2^step}, one per beta pair: the bias corrections of the fused kernel are
        # read from device memory (no per-step value in the kernel arguments: the launch record of a step is replayable)
        self._clocks = {}
        OPTIMIZERS.add(self)

    def _clock(self, b1, b2, step):
        """the device clock for (beta1, beta2) showing `step` completed steps, or None when no clock is at that step (a
        parameter that joined late): the by-value kernel then serves that range"""
        c = self._clocks.get((b1, b2))
        if c is None:
            # seeded at the CURRENT step count (0 for a fresh optimizer, the resumed count after load_state_dict), in float64
            # like the clock kernel's own products, so a resumed optimizer stays on the device-clock kernel
            f1, f2 = struct.unpack("ff", struct.pack("ff", b1, b2))     # the kernel receives the betas as C floats
            p1 = p2 = 1.0
            for _ in range(step):                                       # the clock kernel's own sequence of double products
                p1 *= f1
                p2 *= f2
            st = torch.tensor([float(step), p1, p2, 0.0], dtype=torch.float64).to(self._arena.flat.device)
            c = self._clocks[(b1, b2)] = [st, step, -1]       # [device state, host mirror of its step, epoch of last advance]
        return c if c[1] == step or (c[1] == step + 1 and c[2] == self._arena.epoch) else None

    def _uniform_state(self, segs):
        """cached runs stay valid while every member of a run still has the run's step count"""
        st = self._steps
        return all(st[m[0]] == st[m[-1]] for _, _, m, _ in segs)

    def _export_param_state(self, i, o, p):
        if self._steps[i] == 0:
            return None
        n = p.numel()
        return {"step": torch.tensor(float(self._steps[i])), "exp_avg": self._m[o:o + n].view(p.shape).clone(),
                "exp_avg_sq": self._v[o:o + n].view(p.shape).clone()}

    def _reset_state(self):
        self._m.zero_()
        self._v.zero_()
        self._steps = [0] * len(self._arena.params)
        self._clocks = {}                    # re-seeded from the resumed step counts at the next step()

    def _import_param_state(self, i, o, p, entry):
        n = p.numel()
        self._steps[i] = int(entry["step"])
        self._m[o:o + n].copy_(entry["exp_avg"].reshape(-1))
        self._v[o:o + n].copy_(entry["exp_avg_sq"].reshape(-1))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        a, groups = self._arena, self.param_groups

        def key_of(i, gi):
            g = groups[gi]
            base = (g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
            return base if i is None else base + (self._steps[i],)
        for seg in self._segments(key_of):
            a0, a1, members, key = seg
            lr, b1, b2, eps, wd = key[:5]
            st = self._steps[members[0]]
            clock = self._clock(b1, b2, st)
            if clock is not None:
                if clock[1] == st:                       # first range of this step with these betas: tick the clock
                    ops.adam_advance(clock[0], b1, b2)
                    clock[1], clock[2] = st + 1, a.epoch
                ops.adam_step_dev(a.flat[a0:a1], a.flat_grad[a0:a1], self._m[a0:a1], self._v[a0:a1], lr, b1, b2, eps, wd,
                                  clock[0], self.grad_scale)
            else:
                ops.adam_step(a.flat[a0:a1], a.flat_grad[a0:a1], self._m[a0:a1], self._v[a0:a1], lr, b1, b2, eps, wd,
                              st + 1, self.grad_scale)
            for i in members:
                self._steps[i] = st + 1
            seg[3] = key[:5] + (st + 1,)
        a.epoch += 1                  # invalidates cached filter re-layouts of THIS arena's layers
        return loss


class SGD(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, grad_scale=1.0):
        if dampening != 0 or nesterov:
            raise NotImplementedError("rg_hip.optim.SGD: dampening / nesterov are not used by the reference")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov)
        super(SGD, self).__init__(params, defaults)
        self.grad_scale = grad_scale
        self._setup()
        self._buf = torch.zeros_like(self._arena.flat)
        self._started = [False] * len(self._arena.params)
        OPTIMIZERS.add(self)

    def _uniform_state(self, segs):
        st = self._started
        return all(st[m[0]] == st[m[-1]] for _, _, m, _ in segs)

    def _export_param_state(self, i, o, p):
        if not self._started[i]:
            return None
        n = p.numel()
        return {"momentum_buffer": self._buf[o:o + n].view(p.shape).clone()}

    def _reset_state(self):
        self._buf.zero_()
        self._started = [False] * len(self._arena.params)

    def _import_param_state(self, i, o, p, entry):
        buf = entry.get("momentum_buffer")
        if buf is None:
            self._started[i] = False
            return
        self._started[i] = True
        self._buf[o:o + p.numel()].copy_(buf.reshape(-1))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        a, groups = self._arena, self.param_groups

        def key_of(i, gi):
            g = groups[gi]
            base = (g["lr"], g["momentum"], g["weight_decay"])
            return base if i is None else base + (self._started[i],)
        for seg in self._segments(key_of):
            a0, a1, members, key = seg
            lr, mom, wd = key[:3]
            started = self._started[members[0]]
            ops.sgd_step(a.flat[a0:a1], a.flat_grad[a0:a1], self._buf[a0:a1], lr, mom, wd, not started, self.grad_scale)
            if not started:
                for i in members:
                    self._started[i] = True
            seg[3] = key[:3] + (True,)
        a.epoch += 1
        return loss
