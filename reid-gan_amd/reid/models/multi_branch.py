"""SiameseNet — restates FD-GAN-master/reid/models/multi_branch.py:6-15.

The reference runs `base_model` once per branch.  When every BatchNorm of the base normalises with its
running statistics (FD-GAN: `set_bn_fix`, FD/fdgan/model.py:72-85) samples are independent, so both
branches go through the trunk as ONE batch — identical results, half the launches, twice the GEMM N.
`forward_shared(x1, [x2a, x2b, ...])` extends that to several second branches sharing x1 (the
discriminator update evaluates D_id(origin, target) and D_id(origin, fake), model.py:176-177): the
shared branch is computed once and receives the sum of its gradients.
"""
from __future__ import absolute_import

import torch

from rg_hip import ops
from rg_hip.resnet_trunk import bn_all_eval
from rg_hip.tape import RGModule, Tape, run


class SiameseNet(RGModule):
    def __init__(self, base_model, embed_model):
        super(SiameseNet, self).__init__()
        self.base_model = base_model
        self.embed_model = embed_model

    # ---- standard two-branch call: returns (f1, f2[, embed(f1, f2)]) -----------------------------
    def tf(self, tape, x1, x2):
        eval_bn = bn_all_eval(self.base_model)
        ni = tape.needs_input
        if (eval_bn and tape.record and not tape.param_grad and ni is not None and not ni[0] and ni[1]
                and self.embed_model is not None):
            # generator update: only the second branch carries a gradient and the weights are constants
            # (FD/fdgan/model.py:196) — run the first branch without recording anything
            f1 = self.base_model.tf(Tape(param_grad=False, record=False), x1)
            f2 = self.base_model.tf(tape, x2)
            tape.push(("x2only", x1.shape[0]))
            return f1, f2, self.embed_model.tf(tape, f1, f2)
        batched = eval_bn and x1.shape[1:] == x2.shape[1:]
        if batched:
            b = x1.shape[0]
            f = self.base_model.tf(tape, torch.cat([x1, x2], 0))
            f1, f2 = f[:b], f[b:]
        else:
            f1 = self.base_model.tf(tape, x1)
            f2 = self.base_model.tf(tape, x2)
        tape.push((batched, x1.shape[0]))
        if self.embed_model is None:
            return f1, f2
        return f1, f2, self.embed_model.tf(tape, f1, f2)

    def tb(self, tape, d1, d2, ds=None, need_dx=True):
        if self.embed_model is not None:
            if ds is None:
                raise RuntimeError("SiameseNet: backward needs a gradient for the embedding output")
            e1, e2 = self.embed_model.tb(tape, ds)
            d1 = e1 if d1 is None else ops.add(d1, e1)
            d2 = e2 if d2 is None else ops.add(d2, e2)
        batched, b = tape.pop()
        if batched == "x2only":
            return None, self.base_model.tb(tape, d2, need_dx=need_dx)
        if batched:
            dx = self.base_model.tb(tape, torch.cat([d1, d2], 0), need_dx=need_dx)
            return (dx[:b], dx[b:]) if need_dx else (None, None)
        dx2 = self.base_model.tb(tape, d2, need_dx=need_dx)
        dx1 = self.base_model.tb(tape, d1, need_dx=need_dx)
        return dx1, dx2

    # ---- one shared first branch, several second branches: returns [embed(f1, f2_i)] -------------
    def forward_shared(self, x1, x2_list):
        if self.embed_model is None or not bn_all_eval(self.base_model):
            return [self(x1, x2)[-1] for x2 in x2_list]
        return list(run(_SharedProgram(self), x1, *x2_list))


class _SharedProgram(object):
    """tape program over SiameseNet's own submodules (not an nn.Module: it owns no parameters)."""

    def __init__(self, net):
        self.net = net

    def parameters(self):
        return self.net.parameters()

    @property
    def _rg_frozen(self):
        return getattr(self.net, "_rg_frozen", False)

    def tf(self, tape, x1, *x2s):
        net, b = self.net, x1.shape[0]
        f = net.base_model.tf(tape, torch.cat((x1,) + tuple(x2s), 0))
        tape.push(b)                                  # marker between the trunk and the embed records
        f1 = f[:b]
        return tuple(net.embed_model.tf(tape, f1, f[b * (i + 1):b * (i + 2)]) for i in range(len(x2s)))

    def tb(self, tape, *ds, **kw):
        need_dx = kw.get("need_dx", True)
        net, n2 = self.net, len(ds)
        d1, d2s = None, [None] * n2
        for i in range(n2 - 1, -1, -1):               # reverse of the forward order (tape is LIFO)
            if ds[i] is None:
                raise RuntimeError("SiameseNet.forward_shared: every score must take part in the loss")
            e1, d2s[i] = net.embed_model.tb(tape, ds[i])
            d1 = e1 if d1 is None else ops.add(d1, e1)
        b = tape.pop()
        dx = net.base_model.tb(tape, torch.cat([d1] + d2s, 0), need_dx=need_dx)
        if not need_dx:
            return (None,) * (n2 + 1)
        return tuple(dx[b * i:b * (i + 1)] for i in range(n2 + 1))
