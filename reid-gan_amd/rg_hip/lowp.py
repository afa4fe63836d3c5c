"""FP8 convolution path (BASELINE config 5): per-layer scaling states, operand quantisation and the three fp8 MFMA
GEMMs of csrc/conv_f8.hip behind rg_hip.nn's Conv2d / ConvTranspose2d / SNConv2d.

`set_conv_dtype(net, 'fp8')` gives every convolution of `net` an `F8Layer`; their tf / tb programs then run

    forward   x  --quantise e4m3--> [N][HW][Cp] (+ [C][HW][Np] kept for the weight gradient)   W --e4m3--> [K][RS][Cp], [C][RS][Kp]
              y  = act(sx*sw * conv(xq, wq) + bias + residual)                                  fp32 out, fused epilogue
    backward  dy --quantise e5m2--> [N][PQ][Kp] and [K][PQ][Np]
              dx = sdy*sw * dgrad(dyq, wq^T) (+ residual),  dw = sx*sdy * wgrad(xq, dyq)          fp32 out

with fp32 accumulation; normalisation layers, activations, losses, attention and the optimizers stay fp32.  One scale
per tensor: `policy='jit'` measures max|x| right before quantising (3 small launches more per tensor; exact, what the
parity tests use), `policy='delayed'` quantises with the scale of the previous step while collecting the next one
(rg_f8_roll_scales once per step via `F8States.roll()`; the first use of a tensor is calibrated just in time).
Everything here raises on non-GPU tensors; there is no fallback path.
"""
from __future__ import absolute_import

import torch

from . import ops
from .lib import lib
from .ops import _chk, _p, _pair, _stream, workspace

E4M3, E5M2 = 0, 1
FMAX = (448.0, 57344.0)


def pad16(v):
    return (int(v) + 15) // 16 * 16


class F8States(object):
    """Arena of scaling states ([count, 4] fp32 on the device) for one network; slots are handed out at construction."""

    def __init__(self, device, capacity=1024):
        self.buf = torch.zeros(capacity, 4, dtype=torch.float32, device=device)
        self.used = 0
        self.policy = "jit"
        self._fmax_host = []

    def new(self, fmt):
        if self.used >= self.buf.shape[0]:
            raise RuntimeError("F8States: capacity %d exhausted" % self.buf.shape[0])
        i = self.used
        self.used += 1
        self._fmax_host.append(FMAX[fmt])
        return F8State(self, i, fmt)

    def finalize(self):
        """write the format maxima (one host->device copy for the whole arena)"""
        col = torch.tensor(self._fmax_host + [1.0] * (self.buf.shape[0] - self.used), dtype=torch.float32)
        self.buf[:, 3].copy_(col.to(self.buf.device))
        self.buf[:, 2].fill_(1.0)

    def roll(self):
        """delayed scaling: make the collected maxima current (one launch for the whole network)"""
        if self.used:
            lib.rg_f8_roll_scales(_p(self.buf), self.used, _stream())


class F8State(object):
    __slots__ = ("arena", "index", "fmt", "calibrated")

    def __init__(self, arena, index, fmt):
        self.arena, self.index, self.fmt, self.calibrated = arena, index, fmt, False

    @property
    def ptr(self):
        return self.arena.buf.data_ptr() + 16 * self.index

    def view(self):
        return self.arena.buf[self.index]

    def prepare(self, x, force_jit=False):
        """make state[0] / state[2] valid for quantising `x` now"""
        if force_jit or self.arena.policy == "jit" or not self.calibrated:
            lib.rg_f8_amax(_p(x), x.numel(), self.ptr, _stream())
            lib.rg_f8_roll_scales(self.ptr, 1, _stream())
            self.calibrated = True


class QTensor(object):
    """A quantised operand: bytes + the one-float device dequantisation scale the quantiser used + the fp8 format."""
    __slots__ = ("buf", "scale", "fmt")

    def __init__(self, buf, scale, fmt):
        self.buf, self.scale, self.fmt = buf, scale, fmt

    @property
    def ptr(self):
        return self.scale.data_ptr()


def quantize(x, state, layout, scale=None):
    """fp32 [N, C, H, W] (or [K, C, KH, KW] filters) -> QTensor.
    layout 'nhwc': [N][HW][Cp];  'chwn': [C][HW][Np];  (filters) 'krsc' == 'nhwc', 'crsk' == 'chwn'."""
    x = _chk(x, "x")
    N, C = x.shape[0], x.shape[1]
    L = x.numel() // (N * C)
    if scale is None:
        scale = torch.empty(1, dtype=torch.float32, device=x.device)
    if layout in ("nhwc", "krsc"):
        out = torch.empty((N, L, pad16(C)), dtype=torch.uint8, device=x.device)
        lib.rg_f8_quantize(_p(x), _p(out), state.ptr, _p(scale), state.fmt, N, C, L, C * L, L, _stream())
    elif layout in ("chwn", "crsk"):
        out = torch.empty((C, L, pad16(N)), dtype=torch.uint8, device=x.device)
        lib.rg_f8_quantize(_p(x), _p(out), state.ptr, _p(scale), state.fmt, C, N, L, L, C * L, _stream())
    else:
        raise ValueError("quantize: unknown layout %r" % (layout,))
    return QTensor(out, scale, state.fmt)


def conv_fwd(xq, wq, geom, shift=None, residual=None, act=ops.ACT_NONE, slope=0.0):
    """geom = (N, C, H, W, K, KH, KW, sh, sw, ph, pw); QTensors xq [N][HW][Cp], wq [K][RS][Cp] (e4m3) -> y fp32 NCHW"""
    N, C, H, W, K, KH, KW, sh, sw_, ph, pw = geom
    P, Q = ops.conv_out_size(H, W, KH, KW, (sh, sw_), (ph, pw))
    y = torch.empty((N, K, P, Q), dtype=torch.float32, device=xq.buf.device)
    shift, residual = _chk(shift, "shift"), _chk(residual, "residual")
    if residual is not None and residual.shape != y.shape:
        raise ValueError("conv_fwd(fp8): residual shape mismatch")
    lib.rg_conv2d_f8_fwd(_p(xq.buf), _p(wq.buf), xq.ptr, wq.ptr, xq.fmt, _p(y), N, C, H, W, K, KH, KW, sh, sw_, ph, pw, P, Q,
                         _p(shift), _p(residual), act, slope, _stream())
    return y


def conv_dgrad(dyq, wq_t, geom, x_hw, shift=None, residual=None, act=ops.ACT_NONE, slope=0.0):
    """geom as conv_fwd (of the convolution whose data gradient this is); QTensors dyq [N][PQ][Kp], wq_t [C][RS][Kp] -> dx"""
    N, C, _, _, K, KH, KW, sh, sw_, ph, pw = geom
    H, W = x_hw
    P, Q = geom_pq(geom, x_hw)
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dyq.buf.device)
    shift, residual = _chk(shift, "shift"), _chk(residual, "residual")
    if residual is not None and residual.shape != dx.shape:
        raise ValueError("conv_dgrad(fp8): residual shape mismatch")
    lib.rg_conv2d_f8_dgrad(_p(dyq.buf), _p(wq_t.buf), dyq.ptr, wq_t.ptr, dyq.fmt, _p(dx), N, C, H, W, K, KH, KW, sh, sw_, ph, pw, P, Q,
                           _p(shift), _p(residual), act, slope, _stream())
    return dx


def geom_pq(geom, x_hw):
    _, _, _, _, _, KH, KW, sh, sw_, ph, pw = geom
    return ops.conv_out_size(x_hw[0], x_hw[1], KH, KW, (sh, sw_), (ph, pw))


def conv_wgrad(xq_chwn, dyq_chwn, geom, out=None):
    """QTensors xq_chwn [C][HW][Np], dyq_chwn [K][PQ][Np] -> dw fp32 [K][C][KH][KW]"""
    N, C, H, W, K, KH, KW, sh, sw_, ph, pw = geom
    P, Q = ops.conv_out_size(H, W, KH, KW, (sh, sw_), (ph, pw))
    dw = out if out is not None else torch.empty((K, C, KH, KW), dtype=torch.float32, device=xq_chwn.buf.device)
    nbytes = ops._ws_query("rg_conv2d_f8_wgrad_workspace", N, C, K, KH, KW, P, Q)
    ws = workspace(nbytes, xq_chwn.buf.device)
    lib.rg_conv2d_f8_wgrad(_p(xq_chwn.buf), _p(dyq_chwn.buf), xq_chwn.ptr, dyq_chwn.ptr, xq_chwn.fmt, dyq_chwn.fmt, _p(dw), N, C, H, W, K, KH, KW, sh,
                           sw_, ph, pw, P, Q, _p(ws), ws.numel(), _stream())
    return dw


def quantize_dual(x, state, want_a=True, want_b=True):
    """(QTensor 'nhwc' [N][L][Cp] or None, QTensor 'chwn' [C][L][Np] or None) from ONE pass over the fp32 tensor"""
    x = _chk(x, "x")
    N, C = x.shape[0], x.shape[1]
    L = x.numel() // (N * C)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    a = torch.empty((N, L, pad16(C)), dtype=torch.uint8, device=x.device) if want_a else None
    b = torch.empty((C, L, pad16(N)), dtype=torch.uint8, device=x.device) if want_b else None
    lib.rg_f8_quantize_dual(_p(x), _p(a), _p(b), state.ptr, _p(scale), state.fmt, N, C, L, _stream())
    return (QTensor(a, scale, state.fmt) if want_a else None, QTensor(b, scale, state.fmt) if want_b else None)


def quantize_grad_dual(dy, state, want_a=True, want_b=True, y=None, act=ops.ACT_NONE, slope=0.0, want_sum=False):
    """the output gradient of a convolution, ONE pass: g = dy * act'(y) -> (QTensor [N][L][Kp] or None, QTensor [K][L][Np] or None,
    per-tile channel sums of g [tiles][K] or None).  g is not materialised; sum the partials with ops.rows_sum_pair."""
    dy, y = _chk(dy, "dy"), _chk(y, "y")
    if act != ops.ACT_NONE and (y is None or y.shape != dy.shape):
        raise ValueError("quantize_grad_dual: the activation backward needs the forward output of the same shape")
    N, C = dy.shape[0], dy.shape[1]
    L = dy.numel() // (N * C)
    scale = torch.empty(1, dtype=torch.float32, device=dy.device)
    a = torch.empty((N, L, pad16(C)), dtype=torch.uint8, device=dy.device) if want_a else None
    b = torch.empty((C, L, pad16(N)), dtype=torch.uint8, device=dy.device) if want_b else None
    part = torch.empty((lib.rg_f8_grad_tiles(N, L), C), dtype=torch.float32, device=dy.device) if want_sum else None
    lib.rg_f8_quantize_grad(_p(dy), _p(y) if act != ops.ACT_NONE else None, act, slope, _p(a), _p(b), _p(part), state.ptr, _p(scale),
                            state.fmt, N, C, L, _stream())
    return (QTensor(a, scale, state.fmt) if want_a else None, QTensor(b, scale, state.fmt) if want_b else None, part)


class F8Layer(object):
    """Scaling states and the cached quantised filters of one convolution layer."""

    def __init__(self, states):
        self.states = states
        self.sx = states.new(E4M3)          # input activations
        self.sw = states.new(E4M3)          # filters
        self.sdy = states.new(E5M2)         # output gradients
        self._wkey = None
        self._wq = None

    def quant_act(self, x, layout):
        self.sx.prepare(x)
        return quantize(x, self.sx, layout)

    def quant_act_both(self, x, want_chwn):
        self.sx.prepare(x)
        return quantize_dual(x, self.sx, True, want_chwn)

    def quant_grad_fused(self, dy, y, act, slope, want_nhwc, want_chwn, want_sum):
        """(dyq, dyq_t, bias gradient or None) of g = dy * act'(y).  With a calibrated delayed scale everything is one launch
        (+ the column sum of the tile partials for the bias); a scale that has to be measured first needs g itself, so that case
        runs the activation backward separately and measures it (policy 'jit', first use of the layer)."""
        measured = self.states.policy == "jit" or not self.sdy.calibrated
        if measured and act != ops.ACT_NONE:
            dy = ops.act_bwd(dy, y, act, slope)
            act = ops.ACT_NONE
        if measured:
            self.sdy.prepare(dy)
        if not (want_nhwc or want_chwn or want_sum):
            return None, None, None
        a, b, part = quantize_grad_dual(dy, self.sdy, want_nhwc, want_chwn, y=y, act=act, slope=slope, want_sum=want_sum)
        return a, b, part

    def weights(self, w, key=None):
        """(wq [K][RS][Cp], wq_t [C][RS][Kp]) of the filter tensor, re-quantised (always with a fresh amax: filters are
        small) when `key` changes; key=None: every call (spectral-normed filters change every forward)."""
        if key is None or key != self._wkey or ops.CAPTURING[0]:
            # filters change by one optimizer step between uses: the previous step's maximum scales them (delayed policy);
            # 'jit' measures first
            self.sw.prepare(w)
            self._wq = quantize_dual(w, self.sw, True, True)
            self._wkey = None if ops.CAPTURING[0] else key      # recorded, not executed, inside a capture: stays stale for eager code
        return self._wq


def set_conv_dtype(net, dtype, policy="delayed", keep_fp32=()):
    """Switch every rg_hip.nn convolution of `net` to `dtype` ('fp8' or 'fp32').  `keep_fp32`: sub-module names (as in
    named_modules) that stay fp32.  Returns the F8States arena (None for fp32); call `.roll()` once per training step."""
    from . import nn as rnn
    if dtype not in ("fp8", "fp32"):
        raise ValueError("set_conv_dtype: %r" % (dtype,))
    convs = [(n, m) for n, m in net.named_modules() if isinstance(m, (rnn.Conv2d, rnn.ConvTranspose2d, rnn.SNConv2d))]
    if dtype == "fp32":
        for _, m in convs:
            m.__dict__.pop("_rg_f8", None)
        net.__dict__.pop("_rg_f8_states", None)
        return None
    dev = next(net.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("set_conv_dtype('fp8'): the network must live on the GPU (the fp8 path has no CPU fallback)")
    states = F8States(dev, capacity=max(16, 3 * len(convs)))
    states.policy = policy
    for name, m in convs:
        if any(name == k or name == "module." + k for k in keep_fp32):
            continue
        m.__dict__["_rg_f8"] = F8Layer(states)
    states.finalize()
    net.__dict__["_rg_f8_states"] = states
    return states


def states_of(net):
    return net.__dict__.get("_rg_f8_states") or getattr(getattr(net, "module", None), "__dict__", {}).get("_rg_f8_states")
