"""Layer set of the hot path as RGModules (same constructor arguments, parameter names, default
initialisation and state_dict layout as the torch.nn layers the reference composes).

Fusion available to the network programs (all exact, no re-association beyond fp32 rounding):
  Conv2d / ConvTranspose2d : + bias, + activation in the MFMA epilogue
  BatchNorm               : normalise + affine (+ residual add) (+ activation) in one pass
"""
from __future__ import absolute_import

import math

import torch
from torch import nn
from torch.nn import init

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH
from .tape import RGModule


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


import weakref

WEIGHT_EPOCH = [0]      # bumped by rg_hip.optim after every step (its kernels bypass torch's version counters)
BN_LAYERS = weakref.WeakSet()      # rg_hip.graph advances their host-side `num_batches_tracked` bookkeeping per replay


def _geom_gflop(geom, dy):
    """algorithmic GFLOP of one conv pass with geometry (N, C, H, W, K, KH, KW, ...) and output gradient dy"""
    return 2e-9 * dy.numel() * geom[1] * geom[5] * geom[6]


def _bias_grad(tape, bias, dy):
    """sum of dy over N and the pixels.  When dy is the tensor an InstanceNorm backward just produced, that kernel already
    summed it per (n, c): one tiny launch over [N][C] instead of another pass over the activation."""
    part = getattr(dy, "_rg_inst_sums", None)
    N, C = dy.shape[0], dy.shape[1]
    if part is not None and part.numel() == N * C:
        db, _ = ops.rows_sum_pair(part, None, N, C, tape.grad_out(bias), None)
    else:
        db = ops.channel_sum(dy, out=tape.grad_out(bias))
    tape.add_grad(bias, db)


def _f8_backward_operands(tape, f8, bias, dy, y, act, slope, need_dx, want_w):
    """fp8 backward head shared by the convolution classes: activation backward, e5m2 quantisation in both layouts and the bias
    gradient from ONE pass over dy (lowp.F8Layer.quant_grad_fused).  An InstanceNorm backward that produced dy has already summed
    it per (n, c): that shortcut (see _bias_grad) is kept when there is no activation in between."""
    want_b = tape.wants(bias)
    inst = getattr(dy, "_rg_inst_sums", None) if act == ACT_NONE else None
    use_inst = want_b and inst is not None and inst.numel() == dy.shape[0] * dy.shape[1]
    dyq, dyq_t, part = f8.quant_grad_fused(dy, y, act, slope, need_dx, want_w, want_b and not use_inst)
    if want_b:
        if use_inst:
            db, _ = ops.rows_sum_pair(inst, None, dy.shape[0], dy.shape[1], tape.grad_out(bias), None)
        else:
            db, _ = ops.rows_sum_pair(part, None, part.shape[0], part.shape[1], tape.grad_out(bias), None)
        tape.add_grad(bias, db)
    return dyq, dyq_t


def sn_prepare(convs, training):
    """power iteration + W / sigma of every spectral-normed convolution a network forward is about to run, in two launches instead
    of two per layer; each SNConv2d.tf picks its result up (same u / v updates, same values as the per-layer launches)"""
    convs = [m for m in convs if isinstance(m, SNConv2d)]
    if not convs:
        return
    eps = convs[0].eps
    res = ops.spectral_norm_fwd_multi([(m.weight_orig.detach(), m.weight_u, m.weight_v) for m in convs], training, eps,
                                      save_uv=True)
    for m, r in zip(convs, res):
        m.__dict__["_sn_pre"] = r


class _KrscCache(object):
    """[K][KH*KW][C] copy of a filter tensor for the (r,s)-major kernels, rebuilt only when the weights changed.  Convolutions of one
    network share a KrscGroup (group_krsc): the first stale member re-lays ALL of them out in one launch."""

    def _krsc_wanted(self):
        w = self.weight
        return not (w.shape[2] * w.shape[3] == 1 or w.shape[1] % 4 != 0)

    def _krsc_key(self):
        w = self.weight
        arena = getattr(w, "_rg_arena", None)          # the fused optimizers bump their own arena's epoch
        return (arena.epoch if arena is not None else WEIGHT_EPOCH[0], w._version, w.data_ptr())

    def _krsc(self):
        if not self._krsc_wanted():
            return None
        grp = self.__dict__.get("_krsc_group")
        if grp is not None:
            return grp.get(self)
        key = self._krsc_key()
        if getattr(self, "_wk_key", None) != key or ops.CAPTURING[0]:
            self._wk = ops.weights_to_krsc(self.weight.detach())
            # inside a capture nothing executes (and the copy lives in the capture's pool): the key stays stale, so eager code that
            # runs after the capture — including the fallback of a capture that FAILED — rebuilds the copy for real
            self._wk_key = None if ops.CAPTURING[0] else key
        return self._wk


class KrscGroup(object):
    """The (r,s)-major filter copies of every plain convolution of one network, refreshed together by ONE launch
    (`rg_weights_to_krsc_multi`) when the first member finds its copy stale — after an optimizer step every member is.  The copies
    are persistent buffers; the device table is rebuilt when a parameter moved (load_state_dict, .to())."""

    def __init__(self, candidates):
        self.members = []                 # the candidates that actually asked for their copy (folded convolutions never do)
        self.ptrs = None
        self.table = None
        self.blocks = 0
        self.cap_gen = -1
        for m in candidates:
            m.__dict__["_krsc_group"] = self

    def _build(self):
        chunk = ops.lib.rg_krsc_chunk()
        rows, first = [], 0
        for m in self.members:
            w = m.weight
            K, C, RS = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            wk = m.__dict__.get("_wk")
            if wk is None or tuple(wk.shape) != (K, RS, C) or wk.device != w.device:
                wk = m._wk = torch.empty((K, RS, C), dtype=torch.float32, device=w.device)
            rows.append([w.data_ptr(), wk.data_ptr(), K, C, RS, first])
            first += (K * C * RS + chunk - 1) // chunk
        dev = self.members[0].weight.device
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.blocks = first
        self.ptrs = tuple(m.weight.data_ptr() for m in self.members)

    def refresh(self):
        if self.ptrs != tuple(m.weight.data_ptr() for m in self.members):
            if ops.CAPTURING[0]:
                raise RuntimeError("KrscGroup: a parameter moved while a network program is being captured")
            self._build()
        ops.lib.rg_weights_to_krsc_multi(ops._p(self.table), len(self.members), self.blocks, ops._stream())
        for m in self.members:           # (recorded, not executed, inside a capture: keys stay stale there — see _KrscCache._krsc)
            m._wk_key = None if ops.CAPTURING[0] else m._krsc_key()

    def get(self, m):
        if not m.__dict__.get("_krsc_member"):
            if ops.CAPTURING[0]:
                raise RuntimeError("KrscGroup: a convolution asked for its (r,s)-major filters for the first time inside a capture")
            m.__dict__["_krsc_member"] = True
            self.members.append(m)
            self.ptrs = None              # table rebuilt by the refresh below (the new member's key is stale)
        if ops.CAPTURING[0]:
            if self.cap_gen != ops.CAPTURE_GEN[0]:       # once per captured program: the refresh is one of its nodes
                self.refresh()
                self.cap_gen = ops.CAPTURE_GEN[0]
        elif m.__dict__.get("_wk_key") != m._krsc_key():
            self.refresh()
        return m._wk


_KRSC_GROUPS = __import__("os").environ.get("RG_KRSC_GROUP", "1") != "0"      # A/B switch: 0 = one re-layout launch per filter


def group_krsc(net):
    """give the plain convolutions of `net` one shared KrscGroup (idempotent; called on a network's first run)"""
    if net.__dict__.get("_krsc_grouped"):
        return
    net.__dict__["_krsc_grouped"] = True
    if not _KRSC_GROUPS:
        return
    root = net if isinstance(net, nn.Module) else getattr(net, "net", None)     # tape programs over another module's layers
    if not isinstance(root, nn.Module):
        return
    members = [m for m in root.modules() if isinstance(m, _KrscCache) and isinstance(getattr(m, "weight", None), torch.Tensor)
               and m.weight.is_cuda and m._krsc_wanted() and "_krsc_group" not in m.__dict__]
    if len(members) >= 2:
        KrscGroup(members)


class Conv2d(RGModule, _KrscCache):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super(Conv2d, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):          # identical to torch.nn.Conv2d.reset_parameters
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return "%d, %d, kernel_size=%s, stride=%s, padding=%s, bias=%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.bias is not None)

    def _geom(self, x):
        return (x.shape[0], x.shape[1], x.shape[2], x.shape[3], self.weight.shape[0], self.kernel_size[0], self.kernel_size[1],
                self.stride[0], self.stride[1], self.padding[0], self.padding[1])

    def _wkey(self):
        w = self.weight
        arena = getattr(w, "_rg_arena", None)
        return (arena.epoch if arena is not None else WEIGHT_EPOCH[0], w._version, w.data_ptr())

    def tf(self, tape, x, act=ACT_NONE, slope=0.0, residual=None):
        """y = act(conv(x) + bias + residual): bias, residual add and activation run in the MFMA epilogue."""
        f8 = self.__dict__.get("_rg_f8")
        if f8 is not None:
            # fp8 MFMA family (rg_hip.lowp): e4m3 operands, fp32 accumulate, same fused epilogue
            from . import lowp
            want_w = tape.record and tape.wants(self.weight)
            xq, xq_t = f8.quant_act_both(x, want_w)
            wq, _ = f8.weights(self.weight.detach(), self._wkey())
            geom = self._geom(x)
            y = lowp.conv_fwd(xq, wq, geom, shift=self.bias, residual=residual, act=act, slope=slope)
            tape.push((xq_t, y if act != ACT_NONE else None, act, slope, geom))
            return y
        if self._full_extent(x):
            # the filter covers the whole (unpadded) map — the generator's 512 -> 128 (8,4) bottleneck,
            # FD/fdgan/networks.py:96-100: a plain GEMM x[N][C*H*W] . w[K][C*KH*KW]^T, run with 1x1 geometry
            K = self.weight.shape[0]
            y = ops.conv2d_fwd(x.view(x.shape[0], -1, 1, 1), self.weight.view(K, -1, 1, 1), 1, 0, shift=self.bias,
                               residual=residual, act=act, slope=slope)
        else:
            y = ops.conv2d_fwd(x, self.weight, self.stride, self.padding, shift=self.bias, residual=residual, act=act,
                               slope=slope, w_krsc=self._krsc())
        tape.push((x, y if act != ACT_NONE else None, act, slope))
        return y

    def _full_extent(self, x):
        return (tuple(x.shape[2:]) == self.kernel_size and self.padding == (0, 0) and self.kernel_size != (1, 1)
                and x.is_contiguous())

    def tb(self, tape, dy, need_dx=True, residual=None, mask_input=False, dx_channels=None, want_rowsum=False):
        """mask_input: the conv's input x is the ReLU output of the layer below — its backward (zero where x <= 0) is
        applied to dx (+ residual) in the dgrad epilogue.  dx_channels=(c0, c1): only that channel range of the input
        receives a gradient (the rest of a concatenated input is constant) -> dx has c1 - c0 channels."""
        f8 = self.__dict__.get("_rg_f8")
        if f8 is not None:
            from . import lowp
            if mask_input or dx_channels is not None or want_rowsum:
                raise NotImplementedError("Conv2d(fp8): mask_input / dx_channels / want_rowsum belong to the fp32 ResNet programs")
            xq_t, y, act, slope, geom = tape.pop()
            want_w = tape.wants(self.weight)
            dyq, dyq_t = _f8_backward_operands(tape, f8, self.bias, dy, y, act, slope, need_dx, want_w)
            if want_w:
                gout = tape.grad_out(self.weight)
                tape.add_grad(self.weight, ops.side_call(lambda: lowp.conv_wgrad(xq_t, dyq_t, geom, out=gout), xq_t, dyq_t, gout,
                                                           worth=ops.side_worth(_geom_gflop(geom, dy), fp8=True)))
            if not need_dx:
                return None
            _, wq_t = f8.weights(self.weight.detach(), self._wkey())
            return lowp.conv_dgrad(dyq, wq_t, geom, (geom[2], geom[3]), residual=residual)
        x, y, act, slope = tape.pop()
        if act != ACT_NONE:
            dy = ops.act_bwd(dy, y, act, slope)
        if tape.wants(self.weight):
            tape.add_grad(self.weight, ops.conv2d_wgrad(x, dy, self.weight.shape, self.stride, self.padding,
                                                        out=tape.grad_out(self.weight), side=True))
        if tape.wants(self.bias):
            _bias_grad(tape, self.bias, dy)
        if not need_dx:
            return None
        if self._full_extent(x) and dx_channels is None and residual is None and not mask_input and not want_rowsum:
            K = self.weight.shape[0]
            return ops.conv2d_dgrad(dy, self.weight.view(K, -1, 1, 1), (1, 1), 1, 0).view(x.shape)
        if dx_channels is not None:
            c0, c1 = dx_channels
            return ops.conv2d_dgrad(dy, self.weight.detach()[:, c0:c1].contiguous(), x.shape[2:], self.stride, self.padding)
        return ops.conv2d_dgrad(dy, self.weight, x.shape[2:], self.stride, self.padding, residual=residual,
                                w_krsc=self._krsc(), relu_mask=x if mask_input else None, want_rowsum=want_rowsum)


class ConvTranspose2d(RGModule, _KrscCache):
    """weight [in][out][kh][kw] as torch; forward is the dgrad kernel, backward-data the forward kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        super(ConvTranspose2d, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.output_padding = _pair(output_padding)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):          # identical to torch.nn.ConvTranspose2d (fan_in from dim 1)
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)

    def out_hw(self, H, W):
        return ((H - 1) * self.stride[0] - 2 * self.padding[0] + self.kernel_size[0] + self.output_padding[0],
                (W - 1) * self.stride[1] - 2 * self.padding[1] + self.kernel_size[1] + self.output_padding[1])

    def _geom(self, hw):
        """geometry of the convolution whose data gradient is this layer's forward: (N filled by the caller, C = out_channels,
        H, W = output size, K = in_channels, ...)"""
        return (self.out_channels, hw[0], hw[1], self.in_channels, self.kernel_size[0], self.kernel_size[1], self.stride[0],
                self.stride[1], self.padding[0], self.padding[1])

    def _wkey(self):
        w = self.weight
        arena = getattr(w, "_rg_arena", None)
        return (arena.epoch if arena is not None else WEIGHT_EPOCH[0], w._version, w.data_ptr())

    def tf(self, tape, x, act=ACT_NONE, slope=0.0, residual=None):
        hw = self.out_hw(x.shape[2], x.shape[3])
        f8 = self.__dict__.get("_rg_f8")
        if f8 is not None:
            # forward == data gradient of the convolution with this weight tensor: x plays dy (e4m3 here), filters [C][RS][Kp]
            from . import lowp
            want_w = tape.record and tape.wants(self.weight)
            xq, xq_t = f8.quant_act_both(x, want_w)
            _, wq_t = f8.weights(self.weight.detach(), self._wkey())
            geom = (x.shape[0],) + self._geom(hw)
            y = lowp.conv_dgrad(xq, wq_t, geom, hw, shift=self.bias, residual=residual, act=act, slope=slope)
            tape.push((xq_t, y if act != ACT_NONE else None, act, slope, geom))
            return y
        if (x.shape[2] == 1 and x.shape[3] == 1 and self.padding == (0, 0) and self.output_padding == (0, 0)
                and self.bias is None and act == ACT_NONE and residual is None):
            # a 1x1 input makes the transposed conv a plain GEMM x[N][K] . w[K][C*KH*KW] (the generator's
            # 2432 -> 512 (8,4) layer, FD/fdgan/networks.py:105-109): run it with 1x1 geometry so no
            # MFMA work is spent on taps that fall outside the single input pixel
            K, C, KH, KW = self.weight.shape
            y = ops.conv2d_dgrad(x, self.weight.view(K, C * KH * KW, 1, 1), (1, 1), 1, 0).view(x.shape[0], C, KH, KW)
            tape.push((x, None, act, slope))
            return y
        y = ops.conv2d_dgrad(x, self.weight, hw, self.stride, self.padding, shift=self.bias, residual=residual, act=act,
                             slope=slope, w_krsc=self._krsc())
        tape.push((x, y if act != ACT_NONE else None, act, slope))
        return y

    def tb(self, tape, dy, need_dx=True, residual=None):
        f8 = self.__dict__.get("_rg_f8")
        if f8 is not None:
            from . import lowp
            xq_t, y, act, slope, geom = tape.pop()
            want_w = tape.wants(self.weight)
            dyq, dyq_t = _f8_backward_operands(tape, f8, self.bias, dy, y, act, slope, need_dx, want_w)
            if want_w:
                # filter gradient with the roles swapped: the "input" is dy (e5m2), the "output gradient" is x (e4m3)
                gout = tape.grad_out(self.weight)
                tape.add_grad(self.weight, ops.side_call(lambda: lowp.conv_wgrad(dyq_t, xq_t, geom, out=gout), xq_t, dyq_t, gout,
                                                           worth=ops.side_worth(_geom_gflop(geom, dy), fp8=True)))
            if not need_dx:
                return None
            wq, _ = f8.weights(self.weight.detach(), self._wkey())
            return lowp.conv_fwd(dyq, wq, geom, residual=residual)
        x, y, act, slope = tape.pop()
        if act != ACT_NONE:
            dy = ops.act_bwd(dy, y, act, slope)
        if tape.wants(self.weight):
            tape.add_grad(self.weight, ops.conv2d_wgrad(dy, x, self.weight.shape, self.stride, self.padding,
                                                        out=tape.grad_out(self.weight), side=True))
        if tape.wants(self.bias):
            _bias_grad(tape, self.bias, dy)
        if not need_dx:
            return None
        if (x.shape[2] == 1 and x.shape[3] == 1 and self.padding == (0, 0) and self.output_padding == (0, 0)
                and tuple(dy.shape[2:]) == self.kernel_size and dy.is_contiguous()):
            # 1x1 input (see tf): dx[N][K] = dy[N][C*KH*KW] . w[K][C*KH*KW]^T, no (r,s)-major filter copy needed
            K = self.weight.shape[0]
            return ops.conv2d_fwd(dy.view(dy.shape[0], -1, 1, 1), self.weight.view(K, -1, 1, 1), 1, 0, residual=residual)
        return ops.conv2d_fwd(dy, self.weight, self.stride, self.padding, residual=residual, w_krsc=self._krsc())


class Linear(RGModule):
    def __init__(self, in_features, out_features, bias=True):
        super(Linear, self).__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(self.in_features)
            init.uniform_(self.bias, -bound, bound)

    def tf(self, tape, x):
        tape.push(x)
        return ops.linear_fwd(x, self.weight, self.bias)

    def tb(self, tape, dy, need_dx=True):
        x = tape.pop()
        if tape.wants(self.weight):
            tape.add_grad(self.weight, ops.linear_wgrad(x, dy, out=tape.grad_out(self.weight)))
        if tape.wants(self.bias):
            _bias_grad(tape, self.bias, dy)
        return ops.linear_dgrad(dy, self.weight) if need_dx else None


class _BatchNorm(RGModule):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super(_BatchNorm, self).__init__()
        BN_LAYERS.add(self)
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        if track_running_stats:
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        else:
            self.register_buffer("running_mean", None)
            self.register_buffer("running_var", None)
            self.register_buffer("num_batches_tracked", None)

    def extra_repr(self):
        return "%d, eps=%g, momentum=%g, affine=%s" % (self.num_features, self.eps, self.momentum, self.affine)

    def _flush_nbt(self):
        n = self.__dict__.get("_nbt_pending", 0)
        if n:
            self.__dict__["_nbt_pending"] = 0
            buf = self._buffers.get("num_batches_tracked")
            if buf is not None:
                buf += n

    def __getattr__(self, name):
        if name == "num_batches_tracked":
            self._flush_nbt()
        return super(_BatchNorm, self).__getattr__(name)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_nbt()
        super(_BatchNorm, self)._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        self.__dict__["_nbt_pending"] = 0          # the loaded counter replaces whatever was pending
        super(_BatchNorm, self)._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def tf(self, tape, x, residual=None, act=ACT_NONE, slope=0.0):
        batch_stats = self.training or not self.track_running_stats
        if batch_stats:
            rm = self.running_mean if self.track_running_stats else None
            rv = self.running_var if self.track_running_stats else None
            if self.track_running_stats:
                # bookkeeping counter (int64 buffer): counted on the host, written to the buffer when someone looks
                # (attribute access, state_dict) instead of one tiny device launch per layer and step
                self.__dict__["_nbt_pending"] = self.__dict__.get("_nbt_pending", 0) + 1
            if ops.bn_train_fused_ok(x):
                # small per-channel extent: statistics, running-statistics update and the normalisation in ONE launch
                y, mean, stat = ops.bn_train_fwd_fused(x, self.weight, self.bias, residual, rm, rv, self.eps, self.momentum, act,
                                                       slope)
                tape.push((x, y if act != ACT_NONE else None, mean, stat, False, True, act, slope, residual is not None))
                return y
            mean, stat = ops.bn_stats(x, rm, rv, self.eps, self.momentum)
            is_var = False
        else:
            mean, stat, is_var = self.running_mean, self.running_var, True
        y = ops.bn_apply_fwd(x, mean, stat, self.weight, self.bias, residual, is_var, self.eps, act, slope)
        tape.push((x, y if act != ACT_NONE else None, mean, stat, is_var, batch_stats, act, slope, residual is not None))
        return y

    def tb(self, tape, dy, need_dx=True, dy_masked=False):
        """dy_masked: dy already went through this layer's activation backward (the consumer's dgrad epilogue applied the
        ReLU mask, see Conv2d.tb(mask_input=True)), so the forward output is not read again."""
        x, y, mean, stat, is_var, train, act, slope, has_res = tape.pop()
        if dy_masked:
            act, y = ACT_NONE, None
        need_affine = tape.wants(self.weight) or tape.wants(self.bias)
        if not train and is_var:
            # running statistics: dx does not depend on the channel sums -> one fused pass
            o1 = tape.grad_out(self.bias) if tape.wants(self.bias) else None
            o2 = tape.grad_out(self.weight) if tape.wants(self.weight) else None
            dx, dres, s1, s2 = ops.bn_eval_bwd(x, dy, y, mean, stat, self.weight, self.eps, act, slope,
                                               need_dx=need_dx or not has_res, need_dres=has_res,
                                               need_sums=need_affine, out_sum_dy=o1, out_sum_dy_xhat=o2)
            if tape.wants(self.weight):
                tape.add_grad(self.weight, s2)
            if tape.wants(self.bias):
                tape.add_grad(self.bias, s1)
            return (dx, dres) if has_res else dx
        s1 = s2 = None
        if train and not is_var and ops.bn_train_fused_ok(x):
            o1 = tape.grad_out(self.bias) if tape.wants(self.bias) else None
            o2 = tape.grad_out(self.weight) if tape.wants(self.weight) else None
            dx, dres, s1, s2 = ops.bn_train_bwd_fused(x, dy, y, mean, stat, self.weight, act, slope,
                                                      need_dx=need_dx or not has_res, need_dres=has_res,
                                                      out_sum_dy=o1, out_sum_dy_xhat=o2)
            if tape.wants(self.weight):
                tape.add_grad(self.weight, s2)
            if tape.wants(self.bias):
                tape.add_grad(self.bias, s1)
            return (dx, dres) if has_res else dx
        if need_affine or (train and need_dx):
            o1 = tape.grad_out(self.bias) if tape.wants(self.bias) else None
            o2 = tape.grad_out(self.weight) if tape.wants(self.weight) else None
            s1, s2 = ops.bn_bwd_reduce(x, dy, y, mean, stat, is_var, self.eps, act, slope, o1, o2)
            if tape.wants(self.weight):
                tape.add_grad(self.weight, s2)
            if tape.wants(self.bias):
                tape.add_grad(self.bias, s1)
        dx, dres = ops.bn_bwd_apply(x, dy, y, mean, stat, self.weight, s1, s2, train, is_var, self.eps, act, slope,
                                    need_dx=need_dx or not has_res, need_dres=has_res)
        return (dx, dres) if has_res else dx


# ---- convolution + BatchNorm as one layer ------------------------------------------------------------------------
# With running statistics (FD-GAN's E and D_id: `set_bn_fix`, every eval-mode network) the normalisation is a per-channel
# affine map of the conv output: it runs in the conv epilogue together with the residual add and the ReLU, and the
# pre-normalisation tensor is never written or read again (see csrc/norm.hip, "conv + frozen-statistics BatchNorm").
# Train-mode BatchNorm keeps the separate statistics / apply passes.
class _Fold(object):
    __slots__ = ("key", "scale", "shift", "invstd", "w_scaled", "w_scaled_krsc")


def _foldable(conv, bn):
    return (isinstance(bn, _BatchNorm) and not bn.training and bn.track_running_stats and bn.affine and conv.bias is None
            and isinstance(conv, Conv2d))


def _pair_key(conv, bn):
    w, g = conv.weight, bn.weight
    aw, ag = getattr(w, "_rg_arena", None), getattr(g, "_rg_arena", None)
    return (aw.epoch if aw is not None else WEIGHT_EPOCH[0], ag.epoch if ag is not None else WEIGHT_EPOCH[0],
            w._version, g._version, bn.bias._version, bn.running_mean._version, bn.running_var._version, w.data_ptr(),
            g.data_ptr())


def _fold_of(conv, bn):
    """Per-pair fold (one small launch group); networks fold all their pairs at once through FoldGroup."""
    key = _pair_key(conv, bn)
    f = getattr(conv, "_rg_fold", None)
    if f is None or f.key != key:
        f = _Fold()
        f.key = key
        f.scale, f.shift, f.invstd = ops.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        ws = f.w_scaled = ops.scale_rows(conv.weight.detach(), f.scale)
        f.w_scaled_krsc = ops.weights_to_krsc(ws) if (ws.shape[2] * ws.shape[3] > 1 and ws.shape[1] % 4 == 0) else None
        conv._rg_fold = f
    return f


class FoldGroup(object):
    """All (conv, BatchNorm) pairs of one network, folded by ONE kernel launch per weight version
    (`rg_fold_filters_multi`): scale / shift / invstd and the scaled filters of every pair live in two flat buffers."""

    def __init__(self, pairs):
        self.pairs = list(pairs)
        self.key = None
        self.table = None

    def _build(self, device):
        chunk = lib_fold_chunk()
        n_w = n_k = n_c = 0
        for conv, bn in self.pairs:
            w = conv.weight
            n_w += (w.numel() + 63) // 64 * 64
            if w.shape[2] * w.shape[3] > 1 and w.shape[1] % 4 == 0:
                n_k += (w.numel() + 63) // 64 * 64
            n_c += (w.shape[0] + 63) // 64 * 64
        self.buf_w = torch.empty(n_w + n_k, dtype=torch.float32, device=device)
        self.buf_c = torch.empty(3 * n_c, dtype=torch.float32, device=device)
        rows, ow, oc, blocks = [], 0, 0, 0
        self.folds = []
        for conv, bn in self.pairs:
            w = conv.weight
            K, C, RS = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            f = _Fold()
            f.w_scaled = self.buf_w[ow:ow + w.numel()].view(w.shape)
            ow += (w.numel() + 63) // 64 * 64
            if RS > 1 and C % 4 == 0:
                f.w_scaled_krsc = self.buf_w[ow:ow + w.numel()].view(K, RS, C)
                ow += (w.numel() + 63) // 64 * 64
            else:
                f.w_scaled_krsc = None
            kc = (K + 63) // 64 * 64
            f.scale, f.shift, f.invstd = (self.buf_c[oc:oc + K], self.buf_c[oc + kc:oc + kc + K],
                                          self.buf_c[oc + 2 * kc:oc + 2 * kc + K])
            oc += 3 * kc
            eps_bits = int(torch.tensor(bn.eps, dtype=torch.float32).view(torch.int32).item())
            rows.append([w.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                         bn.running_var.data_ptr(), f.w_scaled.data_ptr(),
                         f.w_scaled_krsc.data_ptr() if f.w_scaled_krsc is not None else 0, f.scale.data_ptr(),
                         f.shift.data_ptr(), f.invstd.data_ptr(), K, C, RS, eps_bits, blocks, 0])
            blocks += (w.numel() + chunk - 1) // chunk
            self.folds.append(f)
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.blocks = blocks
        self.ptrs = self._ptrs()

    def _ptrs(self):
        return tuple(conv.weight.data_ptr() for conv, _ in self.pairs) + tuple(bn.weight.data_ptr() for _, bn in self.pairs)

    def usable(self):
        return all(_foldable(conv, bn) for conv, bn in self.pairs)

    def prepare(self):
        """Fold every pair if any weight changed since the last call; afterwards conv._rg_fold is current."""
        conv0, bn0 = self.pairs[0]
        ptrs = self._ptrs()
        if self.table is None or ptrs != self.ptrs:
            self._build(conv0.weight.device)
        key = tuple(_pair_key(conv, bn)[:7] for conv, bn in self.pairs)
        if key == self.key:
            return
        ops.fold_filters_multi(self.table, len(self.pairs), self.blocks)
        self.key = key
        for (conv, bn), f in zip(self.pairs, self.folds):
            f.key = _pair_key(conv, bn)
            conv._rg_fold = f


def lib_fold_chunk():
    from .lib import lib
    return lib.rg_fold_chunk()


def conv_bn_tf(tape, conv, bn, x, residual=None, act=ACT_NONE):
    """act(bn(conv(x)) [+ residual])"""
    if not _foldable(conv, bn):
        y = bn.tf(tape, conv.tf(tape, x), residual=residual, act=act)
        tape.push(None)
        return y
    f = _fold_of(conv, bn)               # a hit when the network's FoldGroup ran for this weight version
    y = ops.conv2d_fwd(x, f.w_scaled, conv.stride, conv.padding, shift=f.shift, residual=residual, act=act,
                       w_krsc=f.w_scaled_krsc)
    tape.push((x, y if act != ACT_NONE else None, f, act, residual is not None))
    return y


def conv_bn_tb(tape, conv, bn, dy, need_dx=True, residual=None, dy_masked=False, mask_input=False, want_rowsum=None):
    """-> dx, or (dx, d_residual) when the forward had a residual input; `residual` here is added to dx.
    dy_masked: dy already carries this layer's ReLU backward (the consumer's dgrad applied it, see mask_input);
    mask_input: apply the ReLU backward of the layer BELOW (whose output is this conv's input) to dx + residual."""
    rec = tape.pop()
    if rec is None:
        r = bn.tb(tape, dy, dy_masked=dy_masked)
        d, dres = r if isinstance(r, tuple) else (r, None)
        dx = conv.tb(tape, d, need_dx=need_dx, residual=residual, mask_input=mask_input)
        return (dx, dres) if isinstance(r, tuple) else dx
    if want_rowsum is None:
        want_rowsum = mask_input and tape.param_grad      # the layer below is a fold that will want the channel sums
    x, y, f, act, has_res = rec
    want_w, want_g, want_b = tape.wants(conv.weight), tape.wants(bn.weight), tape.wants(bn.bias)
    need_sum = want_g or want_b
    ob = tape.grad_out(bn.bias) if want_b else None
    fused_sums = need_sum and (want_w or want_g)          # the wgrad finishing pass adds the slice partials up
    sg = part = None
    if act != ACT_NONE and not dy_masked:
        if fused_sums:
            g, part = ops.act_bwd_partial(dy, y, act, 0.0, need_g=True)
        else:
            g, sg = ops.act_bwd_sum(dy, y, act, 0.0, need_g=True, need_sum=need_sum, out_sum=ob)
    else:
        g = dy
        if fused_sums:
            part = getattr(dy, "_rg_rowsum", None) if dy_masked else None      # written by the dgrad epilogue that made dy
            if part is None:
                part = ops.act_bwd_partial(dy, None, ACT_NONE, need_g=False)[1]
        elif need_sum:
            sg = ops.act_bwd_sum(dy, None, ACT_NONE, need_g=False, need_sum=True, out_sum=ob)[1]
    dbeta = None
    dbeta_late = False
    if want_b:
        dbeta = sg if sg is not None else (ob if ob is not None else torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device))
        # with fused sums dbeta is WRITTEN by finish() on the side stream: register it only after that launch is
        # enqueued, so an accumulating add_grad (second backward of the pair in one tape) reads finished values
        dbeta_late = part is not None and (want_w or want_g)
        if not dbeta_late:
            tape.add_grad(bn.bias, dbeta)
    if want_w or want_g:
        w = conv.weight.detach()
        og = tape.grad_out(bn.weight) if want_g else None
        dgamma = (og if og is not None else torch.empty_like(f.scale)) if want_g else None

        # G = wgrad(x, g) and the fold's finish (dgamma, dbeta, dW = scale G) in ONE call: with split-K the finish runs inside the
        # reduction launch (rg_conv2d_wgrad_fold)
        dw = ops.conv2d_wgrad(x, g, conv.weight.shape, conv.stride, conv.padding,
                              out=tape.grad_out(conv.weight) if want_w else None, side=True,
                              fold=(w, f.scale, f.invstd, bn.running_mean, sg, part, dbeta if part is not None else None, dgamma))
        if want_w:
            tape.add_grad(conv.weight, dw)
        if want_g:
            tape.add_grad(bn.weight, dgamma)
        if dbeta_late:
            tape.add_grad(bn.bias, dbeta)
    dx = None
    if need_dx:
        dx = ops.conv2d_dgrad(g, f.w_scaled, x.shape[2:], conv.stride, conv.padding, residual=residual,
                              w_krsc=f.w_scaled_krsc, relu_mask=x if mask_input else None, want_rowsum=want_rowsum)
    return (dx, g) if has_res else dx


class BatchNorm2d(_BatchNorm):
    pass


class BatchNorm1d(_BatchNorm):
    pass


class _Act(RGModule):
    ACT, SLOPE = ACT_NONE, 0.0

    def __init__(self, *args, **kw):
        super(_Act, self).__init__()

    def tf(self, tape, x):
        y = ops.act_fwd(x, self.ACT, self.SLOPE)
        tape.push(y)
        return y

    def tb(self, tape, dy, need_dx=True):
        y = tape.pop()
        return ops.act_bwd(dy, y, self.ACT, self.SLOPE) if need_dx else None


class ReLU(_Act):
    ACT = ACT_RELU

    def __init__(self, inplace=False):
        super(ReLU, self).__init__()
        self.inplace = inplace


class LeakyReLU(_Act):
    ACT = ACT_LEAKY

    def __init__(self, negative_slope=0.01, inplace=False):
        super(LeakyReLU, self).__init__()
        self.negative_slope = negative_slope
        self.SLOPE = negative_slope
        self.inplace = inplace


class Tanh(_Act):
    ACT = ACT_TANH


class Dropout(RGModule):
    """Inverted dropout with a counter-based mask (seed drawn from torch's CPU generator per call, so
    torch.manual_seed makes runs reproducible); identity in eval mode or for p == 0."""

    def __init__(self, p=0.5, inplace=False):
        super(Dropout, self).__init__()
        self.p = float(p)

    def tf(self, tape, x):
        if not self.training or self.p == 0.0:
            tape.push(None)
            return x
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        tape.push(seed)
        return ops.dropout(x, self.p, seed)

    def tb(self, tape, dy, need_dx=True):
        seed = tape.pop()
        if seed is None or not need_dx:
            return dy if need_dx else None
        return ops.dropout(dy, self.p, seed)


class MaxPool2d(RGModule):
    def __init__(self, kernel_size, stride=None, padding=0):
        super(MaxPool2d, self).__init__()
        self.kernel_size, self.padding = kernel_size, padding
        self.stride = stride if stride is not None else kernel_size

    def tf(self, tape, x):
        y, arg = ops.maxpool2d_fwd(x, self.kernel_size, self.stride, self.padding)
        tape.push((arg, x.shape))
        return y

    def tb(self, tape, dy, need_dx=True):
        arg, shape = tape.pop()
        return ops.maxpool2d_bwd(dy, arg, shape, self.kernel_size, self.stride, self.padding) if need_dx else None


class Sequential(RGModule):
    def __init__(self, *mods):
        super(Sequential, self).__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return list(self._modules.values())[i]

    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def tf(self, tape, x):
        for m in self._modules.values():
            x = m.tf(tape, x)
        return x

    def tb(self, tape, dy, need_dx=True):
        mods = list(self._modules.values())
        for i in range(len(mods) - 1, -1, -1):
            dy = mods[i].tb(tape, dy, need_dx=(need_dx or i > 0))
        return dy


class InstanceNorm2d(RGModule):
    """One launch per direction (`rg_instnorm_fwd` / `rg_instnorm_bwd`): statistics, affine map, residual add and activation in the
    forward; activation gradient, both per-instance sums, dx and the residual gradient in the backward.  affine=False is the
    FD-GAN setting (FD/fdgan/networks.py:30); affine=True dual_gan's (CC/dual_gan/models/base_function.py:38-48)."""

    def __init__(self, num_features, eps=1e-5, affine=False):
        super(InstanceNorm2d, self).__init__()
        self.num_features, self.eps, self.affine = num_features, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    def tf(self, tape, x, residual=None, act=ACT_NONE, slope=0.0):
        g = self.weight.detach() if self.affine else None
        b = self.bias.detach() if self.affine else None
        y, mean, invstd = ops.instnorm_fwd(x, g, b, residual, self.eps, act, slope)
        tape.push((x, y if act != ACT_NONE else None, mean, invstd, g, act, slope, residual is not None))
        return y

    def tb(self, tape, dy, need_dx=True):
        x, y, mean, invstd, g, act, slope, has_res = tape.pop()
        N, C = x.shape[0], x.shape[1]
        dx, dres, s1, s2, s3 = ops.instnorm_bwd(x, dy, y, mean, invstd, g, act, slope, need_dx=True, need_dres=has_res)
        if self.affine and tape.wants(self.weight):
            dg, db = ops.rows_sum_pair(s2, s1, N, C, tape.grad_out(self.weight), tape.grad_out(self.bias))
            tape.add_grad(self.weight, dg)
            tape.add_grad(self.bias, db)
        dx._rg_inst_sums = s3           # per-(n, c) sums of dx: the bias gradient of a convolution in front (see _bias_grad)
        return (dx, dres) if has_res else dx


class InstanceNorm1d(InstanceNorm2d):
    """[B, C, L] token maps (the PTM blocks, CC/dual_gan/models/PTM.py:176-181): the same kernels on a [B, C, L, 1] view."""

    def tf(self, tape, x, residual=None, act=ACT_NONE, slope=0.0):
        y = super(InstanceNorm1d, self).tf(tape, x.unsqueeze(-1), None if residual is None else residual.unsqueeze(-1),
                                           act, slope)
        return y.squeeze(-1)

    def tb(self, tape, dy, need_dx=True):
        out = super(InstanceNorm1d, self).tb(tape, dy.unsqueeze(-1), need_dx)
        if isinstance(out, tuple):
            return tuple(o.squeeze(-1) for o in out)
        return out.squeeze(-1)


class AvgPool2d(RGModule):
    """nn.AvgPool2d(kernel_size=k, stride=k) (the only form the dual_gan blocks use)."""

    def __init__(self, kernel_size, stride=None, padding=0):
        super(AvgPool2d, self).__init__()
        stride = kernel_size if stride is None else stride
        if _pair(kernel_size) != _pair(stride) or _pair(kernel_size)[0] != _pair(kernel_size)[1] or _pair(padding) != (0, 0):
            raise NotImplementedError("AvgPool2d: only square kernel == stride, padding 0")
        self.kernel_size = _pair(kernel_size)[0]

    def tf(self, tape, x):
        tape.push(x.shape)
        return ops.avgpool2d_fwd(x, self.kernel_size)

    def tb(self, tape, dy, need_dx=True):
        shape = tape.pop()
        return ops.avgpool2d_bwd(dy, shape, self.kernel_size) if need_dx else None


class ReflectionPad2d(RGModule):
    def __init__(self, padding):
        super(ReflectionPad2d, self).__init__()
        self.padding = int(padding)

    def tf(self, tape, x):
        return ops.reflection_pad2d_fwd(x, self.padding)

    def tb(self, tape, dy, need_dx=True):
        return ops.reflection_pad2d_bwd(dy, self.padding) if need_dx else None


class SNConv2d(RGModule):
    """nn.Conv2d under torch.nn.utils.spectral_norm (CC/dual_gan/models/base_function.py:121-126): parameters
    `weight_orig`, `bias`; buffers `weight_u`, `weight_v` (same names / shapes as the hook-based original, so its
    checkpoints load).  Every training-mode forward runs one power iteration in place on u, v and convolves with
    W / sigma (`rg_spectral_norm_fwd`); the backward maps the filter gradient through the division."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, eps=1e-12):
        super(SNConv2d, self).__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.eps = eps
        self.weight_orig = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        init.kaiming_uniform_(self.weight_orig, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight_orig)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)
        h, w = out_channels, self.weight_orig.numel() // out_channels
        self.register_buffer("weight_u", torch.nn.functional.normalize(torch.randn(h), dim=0, eps=eps))
        self.register_buffer("weight_v", torch.nn.functional.normalize(torch.randn(w), dim=0, eps=eps))
        self.weight = None          # W / sigma of the latest forward (plain tensor, as the hook leaves it)

    @classmethod
    def from_conv(cls, conv):
        m = cls(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.bias is not None)
        with torch.no_grad():
            m.weight_orig.copy_(conv.weight)
            if conv.bias is not None:
                m.bias.copy_(conv.bias)
        return m

    def tf(self, tape, x, act=ACT_NONE, slope=0.0, residual=None):
        keep_uv = tape.record and tape.wants(self.weight_orig)
        u = v = None
        pre = self.__dict__.pop("_sn_pre", None)      # (w_sn, sigma, u, v) from the network's batched launch (sn_prepare)
        if pre is not None:
            w_sn, sigma, u, v = pre
        elif keep_uv:
            w_sn, sigma, u, v = ops.spectral_norm_fwd(self.weight_orig.detach(), self.weight_u, self.weight_v, self.training,
                                                      self.eps, save_uv=True)
        else:
            w_sn, sigma = ops.spectral_norm_fwd(self.weight_orig.detach(), self.weight_u, self.weight_v, self.training, self.eps)
        self.weight = w_sn
        f8 = self.__dict__.get("_rg_f8")
        if f8 is not None:
            from . import lowp
            xq, xq_t = f8.quant_act_both(x, keep_uv)
            wq, wq_t = f8.weights(w_sn, None)                      # W / sigma changes with every power iteration
            geom = (x.shape[0], x.shape[1], x.shape[2], x.shape[3], w_sn.shape[0], self.kernel_size[0], self.kernel_size[1],
                    self.stride[0], self.stride[1], self.padding[0], self.padding[1])
            y = lowp.conv_fwd(xq, wq, geom, shift=self.bias, residual=residual, act=act, slope=slope)
            tape.push((xq_t, y if act != ACT_NONE else None, act, slope, w_sn, (wq_t, geom), sigma, u, v))
            return y
        wk = ops.weights_to_krsc(w_sn) if (w_sn.shape[2] * w_sn.shape[3] > 1 and w_sn.shape[1] % 4 == 0) else None
        y = ops.conv2d_fwd(x, w_sn, self.stride, self.padding, shift=self.bias, residual=residual, act=act, slope=slope,
                           w_krsc=wk)
        tape.push((x, y if act != ACT_NONE else None, act, slope, w_sn, wk, sigma, u, v))
        return y

    def tb(self, tape, dy, need_dx=True, residual=None):
        x, y, act, slope, w_sn, wk, sigma, u, v = tape.pop()
        f8 = self.__dict__.get("_rg_f8")
        if f8 is not None:
            from . import lowp
            wq_t, geom = wk
            want_w = tape.wants(self.weight_orig)
            dyq, dyq_t = _f8_backward_operands(tape, f8, self.bias, dy, y, act, slope, need_dx, want_w)
            if want_w:
                gout = tape.grad_out(self.weight_orig)
                tape.add_grad(self.weight_orig, ops.side_call(
                    lambda: ops.spectral_norm_bwd(lowp.conv_wgrad(x, dyq_t, geom), w_sn, u, v, sigma, out=gout),
                    x, dyq_t, w_sn, u, v, sigma, gout, worth=ops.side_worth(_geom_gflop(geom, dy), fp8=True)))
            if not need_dx:
                return None
            return lowp.conv_dgrad(dyq, wq_t, geom, (geom[2], geom[3]), residual=residual)
        if act != ACT_NONE:
            dy = ops.act_bwd(dy, y, act, slope)
        if tape.wants(self.weight_orig):
            dw_sn = ops.conv2d_wgrad(x, dy, w_sn.shape, self.stride, self.padding)
            tape.add_grad(self.weight_orig, ops.spectral_norm_bwd(dw_sn, w_sn, u, v, sigma,
                                                                  out=tape.grad_out(self.weight_orig)))
        if tape.wants(self.bias):
            _bias_grad(tape, self.bias, dy)
        if not need_dx:
            return None
        return ops.conv2d_dgrad(dy, w_sn, x.shape[2:], self.stride, self.padding, residual=residual, w_krsc=wk)
