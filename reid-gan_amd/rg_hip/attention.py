"""nn.MultiheadAttention on channel-major token maps.

The reference's PTM blocks (CC/dual_gan/models/PTM.py:162-247) flatten [B, C, H, W] feature maps to [L, B, C],
run `nn.MultiheadAttention(d_model, nhead)` and permute back around every InstanceNorm1d.  Here tokens stay in the
[B, C, L] layout of the surrounding convolutions for the whole module: the in/out projections are 1x1 convolutions on
the MFMA conv kernels, head h is the channel range [h*D, (h+1)*D), and Q^T K, P V and the four backward products are
stride choices of one batched MFMA GEMM (`rg_bgemm`) — no permute or copy exists anywhere.

Parameter names and shapes equal torch's (`in_proj_weight [3E, E]`, `in_proj_bias [3E]`, `out_proj.weight [E, E]`,
`out_proj.bias [E]`), so the reference's state_dicts load unchanged.  Dropout is 0 in the reference; attention masks
and `need_weights` outputs are not used by it and not provided.
"""
from __future__ import absolute_import

import math

import torch
from torch import nn
from torch.nn import init

from . import ops
from .tape import RGModule


class _OutProj(nn.Module):
    """Parameter holder named like torch's NonDynamicallyQuantizableLinear (class name contains 'Linear' so the
    reference's `init_weights` treats it the same way)."""

    def __init__(self, e):
        super(_OutProj, self).__init__()
        self.weight = nn.Parameter(torch.empty(e, e))
        self.bias = nn.Parameter(torch.zeros(e))


_OutProj.__name__ = "NonDynamicallyQuantizableLinear"


def _conv1x1(x, w, b, residual=None, act=ops.ACT_NONE, slope=0.0):
    """[B, C, L] -> [B, K, L] with w [K, C]."""
    y = ops.conv2d_fwd(x.unsqueeze(-1), w.view(w.shape[0], w.shape[1], 1, 1), 1, 0, shift=b,
                       residual=None if residual is None else residual.unsqueeze(-1), act=act, slope=slope)
    return y.squeeze(-1)


def _conv1x1_dgrad(dy, w):
    dx = ops.conv2d_dgrad(dy.unsqueeze(-1), w.view(w.shape[0], w.shape[1], 1, 1), (dy.shape[2], 1), 1, 0)
    return dx.squeeze(-1)


def _conv1x1_wgrad(x, dy, out=None):
    K, C = dy.shape[1], x.shape[1]
    g = ops.conv2d_wgrad(x.unsqueeze(-1), dy.unsqueeze(-1), (K, C, 1, 1), 1, 0,
                         out=None if out is None else out.view(K, C, 1, 1))
    return g.view(K, C)


class MultiheadAttention(RGModule):
    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True):
        super(MultiheadAttention, self).__init__()
        if embed_dim % num_heads != 0:
            raise AssertionError("embed_dim must be divisible by num_heads")
        if dropout != 0.0 or not bias:
            raise NotImplementedError("MultiheadAttention: the hot path uses dropout = 0 and bias = True")
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = _OutProj(embed_dim)
        self._reset_parameters()

    def _reset_parameters(self):          # torch.nn.MultiheadAttention._reset_parameters
        init.xavier_uniform_(self.in_proj_weight)
        init.constant_(self.in_proj_bias, 0.0)
        init.kaiming_uniform_(self.out_proj.weight, a=math.sqrt(5))
        init.constant_(self.out_proj.bias, 0.0)

    # ---- public call in torch's [L, B, E] convention (API parity; the networks call tf/tb directly) ------------
    def forward(self, query, key, value):
        same_qk, same_kv = key is query, value is key
        q = query.permute(1, 2, 0).contiguous()
        k = q if same_qk else key.permute(1, 2, 0).contiguous()
        v = k if same_kv else value.permute(1, 2, 0).contiguous()
        out = _MHAFn.apply(self, q, k, v, self.in_proj_weight, self.in_proj_bias, self.out_proj.weight, self.out_proj.bias)
        return out.permute(2, 0, 1), None

    # ---- tape program on [B, E, L] maps ----------------------------------------------------------------------
    def _attend(self, q, k, v, B, L, S):
        """q/k/v: views [B, E, L|S] (channel slices of projection buffers).  Returns probabilities and output."""
        E, H, D = self.embed_dim, self.num_heads, self.head_dim
        p = torch.empty((B, H, L, S), dtype=torch.float32, device=q.device)
        ops.bgemm(q, k, p, L, S, D, (1, L), (S, 1), (S, 1), (B, H), (q.stride(0), D * L), (k.stride(0), D * S),
                  (H * L * S, L * S))
        ops.softmax_rows_fwd(p, 1.0 / math.sqrt(D), out=p)
        o = torch.empty((B, E, L), dtype=torch.float32, device=q.device)
        ops.bgemm(v, p, o, D, L, S, (S, 1), (1, S), (L, 1), (B, H), (v.stride(0), D * S), (H * L * S, L * S), (E * L, D * L))
        return p, o

    def tf(self, tape, q_in, k_in, v_in, residual=None):
        E = self.embed_dim
        W, b = self.in_proj_weight.detach(), self.in_proj_bias.detach()
        B, _, L = q_in.shape
        S = k_in.shape[2]
        if q_in is k_in and k_in is v_in:
            qkv = _conv1x1(q_in, W, b)
            q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
            mode = 0
        elif k_in is v_in:
            q = _conv1x1(q_in, W[:E], b[:E])
            kv = _conv1x1(k_in, W[E:], b[E:])
            k, v = kv[:, :E], kv[:, E:]
            mode = 1
        else:
            q = _conv1x1(q_in, W[:E], b[:E])
            k = _conv1x1(k_in, W[E:2 * E], b[E:2 * E])
            v = _conv1x1(v_in, W[2 * E:], b[2 * E:])
            mode = 2
        p, o = self._attend(q, k, v, B, L, S)
        out = _conv1x1(o, self.out_proj.weight.detach(), self.out_proj.bias.detach(), residual=residual)
        tape.push((mode, q_in, k_in, v_in, q, k, v, p, o))
        return out

    def tb(self, tape, d_out, need_dx=True):
        """Returns (dq_in, dk_in, dv_in); with shared inputs the shared gradient is returned once and the others are
        None (mode 0: (dx, None, None); mode 1: (dq, dkv, None))."""
        mode, q_in, k_in, v_in, q, k, v, p, o = tape.pop()
        E, H, D = self.embed_dim, self.num_heads, self.head_dim
        B, _, L = q_in.shape
        S = k_in.shape[2]
        W = self.in_proj_weight.detach()
        Wo = self.out_proj.weight
        if tape.wants(Wo):
            tape.add_grad(Wo, _conv1x1_wgrad(o, d_out, out=tape.grad_out(Wo)))
        if tape.wants(self.out_proj.bias):
            tape.add_grad(self.out_proj.bias, ops.channel_sum(d_out.unsqueeze(-1), out=tape.grad_out(self.out_proj.bias)))
        d_o = _conv1x1_dgrad(d_out, Wo.detach())
        dev = d_o.device
        # gradient buffers laid out like the forward projections so the in-projection backward is one conv per buffer
        if mode == 0:
            dqkv = torch.empty((B, 3 * E, L), dtype=torch.float32, device=dev)
            dq, dk, dv = dqkv[:, :E], dqkv[:, E:2 * E], dqkv[:, 2 * E:]
        elif mode == 1:
            dq = torch.empty((B, E, L), dtype=torch.float32, device=dev)
            dkv = torch.empty((B, 2 * E, S), dtype=torch.float32, device=dev)
            dk, dv = dkv[:, :E], dkv[:, E:]
        else:
            dq = torch.empty((B, E, L), dtype=torch.float32, device=dev)
            dk = torch.empty((B, E, S), dtype=torch.float32, device=dev)
            dv = torch.empty((B, E, S), dtype=torch.float32, device=dev)
        pb = (H * L * S, L * S)
        # dV[d, s] = sum_l dO[d, l] P[l, s]
        ops.bgemm(d_o, p, dv, D, S, L, (L, 1), (S, 1), (S, 1), (B, H), (E * L, D * L), pb, (dv.stride(0), D * S))
        # dP[l, s] = sum_d dO[d, l] V[d, s];  dS = softmax backward (in place)
        dp = torch.empty_like(p)
        ops.bgemm(d_o, v, dp, L, S, D, (1, L), (S, 1), (S, 1), (B, H), (E * L, D * L), (v.stride(0), D * S), pb)
        ops.softmax_rows_bwd(p, dp, 1.0 / math.sqrt(D), out=dp)
        # dQ[d, l] = sum_s K[d, s] dS[l, s];  dK[d, s] = sum_l Q[d, l] dS[l, s]
        ops.bgemm(k, dp, dq, D, L, S, (S, 1), (1, S), (L, 1), (B, H), (k.stride(0), D * S), pb, (dq.stride(0), D * L))
        ops.bgemm(q, dp, dk, D, S, L, (L, 1), (S, 1), (S, 1), (B, H), (q.stride(0), D * L), pb, (dk.stride(0), D * S))

        wants_w, wants_b = tape.wants(self.in_proj_weight), tape.wants(self.in_proj_bias)
        gw = gb = None
        if wants_w:
            gw = tape.grad_out(self.in_proj_weight)
            if gw is None:
                gw = torch.empty_like(W)
        if wants_b:
            gb = tape.grad_out(self.in_proj_bias)
            if gb is None:
                gb = torch.empty(3 * E, dtype=torch.float32, device=dev)
        if mode == 0:
            parts = [(q_in, dqkv, 0, 3 * E)]
        elif mode == 1:
            parts = [(q_in, dq, 0, E), (k_in, dkv, E, 3 * E)]
        else:
            parts = [(q_in, dq, 0, E), (k_in, dk, E, 2 * E), (v_in, dv, 2 * E, 3 * E)]
        dxs = []
        for x, dy, r0, r1 in parts:
            if wants_w:
                _conv1x1_wgrad(x, dy, out=gw[r0:r1])
            if wants_b:
                ops.channel_sum(dy.unsqueeze(-1), out=gb[r0:r1])
            dxs.append(_conv1x1_dgrad(dy, W[r0:r1]) if need_dx else None)
        if wants_w:
            tape.add_grad(self.in_proj_weight, gw)
        if wants_b:
            tape.add_grad(self.in_proj_bias, gb)
        return tuple(dxs) + (None,) * (3 - len(dxs))


class _MHAFn(torch.autograd.Function):
    """Stand-alone autograd wrapper for the public [L, B, E] call."""

    @staticmethod
    def forward(ctx, mod, q, k, v, *params):
        from .tape import Tape
        tape = Tape(param_grad=True)
        ctx.mod, ctx.tape = mod, tape
        ctx.shared = (k is q, v is k)
        return mod.tf(tape, q, q if k is q else k, (q if k is q else k) if v is k else v)

    @staticmethod
    def backward(ctx, g):
        mod, tape = ctx.mod, ctx.tape
        dq, dk, dv = mod.tb(tape, g.contiguous())
        grads = [tape.grads.get(id(p)) for p in (mod.in_proj_weight, mod.in_proj_bias, mod.out_proj.weight, mod.out_proj.bias)]
        same_qk, same_kv = ctx.shared
        if same_qk and same_kv:
            dk = dv = None
        elif same_kv:
            dv = None
        return (None, dq, dk, dv) + tuple(grads)
