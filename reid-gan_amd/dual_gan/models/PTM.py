"""Pose Transformer Module on channel-major token maps.

Mirror of CC/dual_gan/models/PTM.py (PCTM :6-58, PTM :60-112, CABs :115-137, TTBs :140-161, CAB :162-199,
TTB :202-247): same classes, constructor arguments and parameter names.  The reference permutes between [L, B, C]
(attention, linears) and [B, C, L] (InstanceNorm1d) around every sub-layer; here every sub-layer works on [B, C, L]:
linears are 1x1 convolutions with the LeakyReLU / residual add in the MFMA epilogue, attention is
rg_hip.attention.MultiheadAttention, norms are the instance-norm kernels.  No transpose is ever materialised.
"""
from __future__ import absolute_import

import copy

import torch
from torch import nn

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.attention import MultiheadAttention, _conv1x1, _conv1x1_dgrad, _conv1x1_wgrad
from rg_hip.tape import RGModule
from .base_function import get_nonlinearity_layer, _slope


def _norm1d(norm, d_model, affine):
    if norm == 'batch':
        return rnn.BatchNorm1d(d_model, affine=affine)
    return rnn.InstanceNorm1d(d_model, affine=affine)


class _FFN(object):
    """linear2(act(linear1(x))) + x on [B, C, L] (the residual add in linear2's epilogue)."""

    @staticmethod
    def tf(tape, lin1, lin2, act, slope, x):
        h = _conv1x1(x, lin1.weight.detach(), lin1.bias.detach(), act=act, slope=slope)
        y = _conv1x1(h, lin2.weight.detach(), lin2.bias.detach(), residual=x)
        tape.push((x, h))
        return y

    @staticmethod
    def tb(tape, lin1, lin2, act, slope, dy):
        """gradient w.r.t. x INCLUDING the residual branch"""
        x, h = tape.pop()
        if tape.wants(lin2.weight):
            tape.add_grad(lin2.weight, _conv1x1_wgrad(h, dy, out=tape.grad_out(lin2.weight)))
        if tape.wants(lin2.bias):
            tape.add_grad(lin2.bias, ops.channel_sum(dy.unsqueeze(-1), out=tape.grad_out(lin2.bias)))
        dh = ops.act_bwd(_conv1x1_dgrad(dy, lin2.weight.detach()), h, act, slope)
        if tape.wants(lin1.weight):
            tape.add_grad(lin1.weight, _conv1x1_wgrad(x, dh, out=tape.grad_out(lin1.weight)))
        if tape.wants(lin1.bias):
            tape.add_grad(lin1.bias, ops.channel_sum(dh.unsqueeze(-1), out=tape.grad_out(lin1.bias)))
        w1 = lin1.weight.detach()
        dx = ops.conv2d_dgrad(dh.unsqueeze(-1), w1.view(w1.shape[0], w1.shape[1], 1, 1), (dh.shape[2], 1), 1, 0,
                              residual=dy.unsqueeze(-1))
        return dx.squeeze(-1)


def _norm_tf(tape, norm, x):
    if isinstance(norm, rnn.InstanceNorm1d):
        return norm.tf(tape, x)
    return norm.tf(tape, x.unsqueeze(-1)).squeeze(-1)          # BatchNorm1d on [B, C, L]


def _norm_tb(tape, norm, dy):
    if isinstance(norm, rnn.InstanceNorm1d):
        return norm.tb(tape, dy)
    return norm.tb(tape, dy.unsqueeze(-1)).squeeze(-1)


def _to_bcl(x):
    """[L, B, C] -> [B, C, L] for the public per-block calls (the networks never take this path)."""
    return x.permute(1, 2, 0).contiguous()


class _BlockFn(torch.autograd.Function):
    """Runs one block's tape program as an autograd node for stand-alone [L, B, C] calls."""

    @staticmethod
    def forward(ctx, mod, n_in, *tensors):
        from rg_hip.tape import Tape
        xs = tensors[:n_in]
        tape = Tape(param_grad=True)
        ctx.mod, ctx.tape, ctx.n_in, ctx.params = mod, tape, n_in, tensors[n_in:]
        return mod.tf(tape, *[x.detach() for x in xs])

    @staticmethod
    def backward(ctx, g):
        dxs = ctx.mod.tb(ctx.tape, g.contiguous())
        if not isinstance(dxs, (tuple, list)):
            dxs = (dxs,)
        return (None, None) + tuple(dxs)[:ctx.n_in] + tuple(ctx.tape.grads.get(id(p)) for p in ctx.params)


def _call_block(mod, *xs):
    params = [p for p in mod.parameters() if p.requires_grad]
    return _BlockFn.apply(mod, len(xs), *xs, *params)


class CAB(RGModule):
    """Context Augment Block (PTM.py:162-199)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, activation="LeakyReLU", affine=True, norm='instance'):
        super(CAB, self).__init__()
        self.self_attn = MultiheadAttention(d_model, nhead)
        self.linear1 = rnn.Linear(d_model, dim_feedforward)
        self.linear2 = rnn.Linear(dim_feedforward, d_model)
        self.norm1 = _norm1d(norm, d_model, affine)
        self.norm2 = _norm1d(norm, d_model, affine)
        self.activation = get_nonlinearity_layer(activation)

    def with_pos_embed(self, tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward(self, src, pos=None):
        if pos is not None:
            raise NotImplementedError("positional embeddings are never passed by the reference generators")
        return _call_block(self, _to_bcl(src)).permute(2, 0, 1)

    def tf(self, tape, src):
        act, slope = _slope(self.activation)
        x = self.self_attn.tf(tape, src, src, src, residual=src)
        x = _norm_tf(tape, self.norm1, x)
        x = _FFN.tf(tape, self.linear1, self.linear2, act, slope, x)
        return _norm_tf(tape, self.norm2, x)

    def tb(self, tape, dy, need_dx=True):
        act, slope = _slope(self.activation)
        d = _norm_tb(tape, self.norm2, dy)
        d = _FFN.tb(tape, self.linear1, self.linear2, act, slope, d)
        d = _norm_tb(tape, self.norm1, d)
        dx, _, _ = self.self_attn.tb(tape, d)
        return ops.axpby(dx, d, 1.0, 1.0, out=dx)


class TTB(RGModule):
    """Texture Transfer Block (PTM.py:202-247)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, activation="LeakyReLU", affine=True, norm='instance'):
        super(TTB, self).__init__()
        self.self_attn = MultiheadAttention(d_model, nhead)
        self.multihead_attn = MultiheadAttention(d_model, nhead)
        self.linear1 = rnn.Linear(d_model, dim_feedforward)
        self.linear2 = rnn.Linear(dim_feedforward, d_model)
        self.norm1 = _norm1d(norm, d_model, affine)
        self.norm2 = _norm1d(norm, d_model, affine)
        self.norm3 = _norm1d(norm, d_model, affine)
        self.activation = get_nonlinearity_layer(activation)

    def with_pos_embed(self, tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward(self, tgt, memory, val, pos=None):
        if pos is not None:
            raise NotImplementedError("positional embeddings are never passed by the reference generators")
        m = _to_bcl(memory)
        v = m if val is memory else _to_bcl(val)
        return _call_block(self, _to_bcl(tgt), m, v).permute(2, 0, 1)

    def tf(self, tape, tgt, memory, val):
        act, slope = _slope(self.activation)
        x = self.self_attn.tf(tape, tgt, tgt, tgt, residual=tgt)
        x = _norm_tf(tape, self.norm1, x)
        x = self.multihead_attn.tf(tape, x, memory, val, residual=x)
        x = _norm_tf(tape, self.norm2, x)
        x = _FFN.tf(tape, self.linear1, self.linear2, act, slope, x)
        return _norm_tf(tape, self.norm3, x)

    def tb(self, tape, dy, need_dx=True):
        """Returns (d_tgt, d_memory, d_val); d_val is None when val and memory were the same tensor (its gradient is
        inside d_memory)."""
        act, slope = _slope(self.activation)
        d = _norm_tb(tape, self.norm3, dy)
        d = _FFN.tb(tape, self.linear1, self.linear2, act, slope, d)
        d = _norm_tb(tape, self.norm2, d)
        dq, dmem, dval = self.multihead_attn.tb(tape, d)
        d = ops.axpby(dq, d, 1.0, 1.0, out=dq)
        d = _norm_tb(tape, self.norm1, d)
        dx, _, _ = self.self_attn.tb(tape, d)
        return ops.axpby(dx, d, 1.0, 1.0, out=dx), dmem, dval


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for i in range(N)])


class CABs(RGModule):
    """Context Augment Blocks (PTM.py:115-137)."""

    def __init__(self, encoder_layer, num_CABs, norm=None):
        super(CABs, self).__init__()
        self.layers = _get_clones(encoder_layer, num_CABs)
        self.norm = norm

    def forward(self, src, pos=None):
        return _call_block(self, _to_bcl(src)).permute(2, 0, 1)

    def tf(self, tape, src):
        x = src
        for layer in self.layers:
            x = layer.tf(tape, x)
        if self.norm is not None:
            x = _norm_tf(tape, self.norm, x)
        return x

    def tb(self, tape, dy, need_dx=True):
        d = dy
        if self.norm is not None:
            d = _norm_tb(tape, self.norm, d)
        for layer in reversed(list(self.layers)):
            d = layer.tb(tape, d)
        return d


class TTBs(RGModule):
    """Texture Transfer Blocks (PTM.py:140-161); the final norm leaves the result in [B, C, L] (the reference does
    not permute back either, :158-160)."""

    def __init__(self, decoder_layer, num_TTBs, norm=None):
        super(TTBs, self).__init__()
        self.layers = _get_clones(decoder_layer, num_TTBs)
        self.norm = norm

    def forward(self, tgt, memory, val, pos=None):
        m = _to_bcl(memory)
        v = m if val is memory else _to_bcl(val)
        out = _call_block(self, _to_bcl(tgt), m, v)
        return out if self.norm is not None else out.permute(2, 0, 1)

    def tf(self, tape, tgt, memory, val):
        x = tgt
        for layer in self.layers:
            x = layer.tf(tape, x, memory, val)
        if self.norm is not None:
            x = _norm_tf(tape, self.norm, x)
        tape.push(val is memory)
        return x

    def tb(self, tape, dy, need_dx=True):
        shared = tape.pop()
        d = dy
        if self.norm is not None:
            d = _norm_tb(tape, self.norm, d)
        dmem = dval = None
        for layer in reversed(list(self.layers)):
            d, dm, dv = layer.tb(tape, d)
            dmem = dm if dmem is None else ops.axpby(dmem, dm, 1.0, 1.0, out=dmem)
            if dv is not None:
                dval = dv if dval is None else ops.axpby(dval, dv, 1.0, 1.0, out=dval)
        return d, dmem, dval


class _PTMBase(RGModule):
    def __init__(self, d_model=512, nhead=8, num_CABs=6, num_TTBs=6, dim_feedforward=2048, activation="LeakyReLU",
                 affine=True, norm='instance'):
        super(_PTMBase, self).__init__()
        encoder_layer = CAB(d_model, nhead, dim_feedforward, activation, affine, norm)
        encoder_norm = None
        decoder_norm = _norm1d(norm, d_model, affine)
        self.encoder = CABs(encoder_layer, num_CABs, encoder_norm)
        decoder_layer = TTB(d_model, nhead, dim_feedforward, activation, affine, norm)
        self.decoder = TTBs(decoder_layer, num_TTBs, decoder_norm)
        self._reset_parameters()
        self.d_model = d_model
        self.nhead = nhead

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class PCTM(_PTMBase):
    """Pose Transformer Module of PoseGenerator1 (PTM.py:6-58): value -> CABs; query attends to it through the TTBs."""

    def forward(self, query, value, pos_embed=None):
        if pos_embed is not None:
            raise NotImplementedError("positional embeddings are never passed by the reference generators")
        return super(PCTM, self).forward(query, value)

    def tf(self, tape, query, value):
        bs, c, h, w = query.shape
        tape.push((query.shape, value.shape))
        mem = self.encoder.tf(tape, value.reshape(value.shape[0], value.shape[1], -1))
        hs = self.decoder.tf(tape, query.reshape(bs, c, h * w), mem, mem)
        return hs.view(bs, c, h, w)

    def tb(self, tape, dy, need_dx=True):
        d = dy.reshape(dy.shape[0], dy.shape[1], -1)
        dq, dmem, _ = self.decoder.tb(tape, d)
        dval = self.encoder.tb(tape, dmem)
        qshape, vshape = tape.pop()
        return dq.view(qshape), dval.view(vshape)


class PTM(_PTMBase):
    """Pose Transformer Module of DPTNGenerator (PTM.py:60-112): src -> CABs = memory; tgt attends (key memory, value val)."""

    def forward(self, src, tgt, val, pos_embed=None):
        if pos_embed is not None:
            raise NotImplementedError("positional embeddings are never passed by the reference generators")
        return super(PTM, self).forward(src, tgt, val)

    def tf(self, tape, src, tgt, val):
        bs, c, h, w = src.shape
        tape.push((src.shape, tgt.shape, val.shape))
        mem = self.encoder.tf(tape, src.reshape(bs, c, -1))
        hs = self.decoder.tf(tape, tgt.reshape(tgt.shape[0], tgt.shape[1], -1), mem, val.reshape(val.shape[0], val.shape[1], -1))
        return hs.view(bs, c, h, w)

    def tb(self, tape, dy, need_dx=True):
        d = dy.reshape(dy.shape[0], dy.shape[1], -1)
        dt, dmem, dval = self.decoder.tb(tape, d)
        dsrc = self.encoder.tb(tape, dmem)
        sshape, tshape, vshape = tape.pop()
        return dsrc.view(sshape), dt.view(tshape), dval.view(vshape)
