"""Tensor-level wrappers over the C ABI: shape logic, output allocation (torch caching allocator),
stream hand-off.  No autograd here and no arithmetic in torch — every number is produced by a HIP
kernel of libreidgan_hip.so.  All functions require contiguous fp32 CUDA(HIP) tensors and raise
otherwise (there is deliberately no CPU path).
"""
from __future__ import absolute_import

import ctypes
import os

import torch

from .lib import lib

ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_TANH = 0, 1, 2, 3


_DEV = [None]
CAPTURE_GEN = [0]    # one number per capture (rg_hip.netgraph): "already refreshed inside THIS capture" for grouped caches
CAPTURING = [0]      # > 0 while rg_hip.netgraph captures a network program: caches refresh unconditionally (their launches are recorded)


def _stream():
    """raw hipStream_t of torch's current stream on this process's device (fast path: no Stream object is built; one process
    drives one GPU, so the device index is looked up once)"""
    d = _DEV[0]
    if d is None:
        d = _DEV[0] = torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(d)


def _bind_device(t):
    """the process's GPU is the device of the FIRST tensor an op sees (not whatever torch's current device happened to be when the
    first op ran: a rank that touches an op before torch.cuda.set_device(local_rank) must not launch on device 0's stream handle
    forever after); tensors of another GPU are refused — one process drives one GPU"""
    i = t.device.index
    if _DEV[0] is None:
        _DEV[0] = i
    elif i != _DEV[0]:
        raise RuntimeError("rg_hip: tensor on cuda:%d, but this process has been launching on cuda:%d (one process drives one GPU; "
                           "call torch.cuda.set_device(local_rank) before building the networks)" % (i, _DEV[0]))


def _chk(t, name="tensor", dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("rg_hip: %s must live on the GPU (got %s); the HIP path has no CPU fallback"
                           % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("rg_hip: %s must be %s, got %s" % (name, dtype, t.dtype))
    if t.device.index != _DEV[0]:
        _bind_device(t)
    if not t.is_contiguous() or (t.data_ptr() & 15):
        t = t.contiguous()
        if t.data_ptr() & 15:
            t = t.clone()
    return t


def _p(t):
    return None if t is None else t.data_ptr()


def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


_SPLITK_INKERNEL = os.environ.get("RG_SPLITK_INKERNEL", "0") not in ("", "0")


class _Workspace(object):
    """One growing scratch buffer per (device, stream); kernels on a stream are ordered, so reuse is safe."""

    ARRIVALS = 4096      # arrival counters per stream (split-K launches have at most 1 536 workgroups: <= 768 tiles)

    def __init__(self):
        self.bufs = {}
        self.pinned = {}
        self.arrivals = {}

    def get(self, nbytes, device):
        key = (device.index, _stream())
        if _SPLITK_INKERNEL and key not in self.arrivals and not CAPTURING[0]:
            # RG_SPLITK_INKERNEL=1: split-K convolutions on this stream finish inside the kernel (rg_conv_splitk_arrivals) — measured
            # equal to the finishing kernel, so off by default; zeroed once, the launches leave the counters at zero; never freed,
            # never written from here
            cnt = self.arrivals[key] = torch.zeros(self.ARRIVALS, dtype=torch.int32, device=device)
            lib.rg_conv_splitk_arrivals(cnt.data_ptr(), self.ARRIVALS, key[1])
        buf = self.bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.bufs[key] = buf
        if CAPTURING[0]:
            # a captured launch keeps this ADDRESS: the buffer must outlive the graph even if a later, larger request replaces it
            # in `bufs` (the allocator would otherwise hand the freed block to a live tensor that the replays then overwrite)
            self.pinned[buf.data_ptr()] = buf
        return buf


_ws = _Workspace()


def splitk_inkernel(enable):
    """(tests, A/B runs) register or remove the current stream's split-K arrival counters: with them the split forward /
    unit-stride data-gradient launches finish inside the convolution kernel, without them a finishing kernel follows — the same
    bits either way.  Returns whether they were registered before."""
    d = _DEV[0]
    if d is None:
        d = _DEV[0] = torch._C._cuda_getDevice()
    key = (d, _stream())
    cnt = _ws.arrivals.get(key)
    was = cnt is not None and cnt is not False
    if enable:
        if not was:
            cnt = _ws.arrivals[key] = torch.zeros(_Workspace.ARRIVALS, dtype=torch.int32, device=torch.device("cuda", d))
            lib.rg_conv_splitk_arrivals(cnt.data_ptr(), _Workspace.ARRIVALS, key[1])
    else:
        _ws.arrivals[key] = False              # keeps _Workspace.get from registering again
        lib.rg_conv_splitk_arrivals(None, 0, key[1])
    return was


_ws_sizes = {}


def _ws_query(fn, *dims):
    """workspace size queries are pure functions of the geometry: ask the library once per shape (the small-kernel
    steps are host-bound; this removes a third of their ctypes calls).  The planner force knob changes the answer, so
    rg_conv_set_force users call _ws_sizes.clear()."""
    key = (fn,) + dims
    v = _ws_sizes.get(key)
    if v is None:
        v = _ws_sizes[key] = getattr(lib, fn)(*dims)
    return v


def workspace(nbytes, device):
    return _ws.get(nbytes, device)


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
def conv_out_size(H, W, KH, KW, stride, padding):
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    return (H + 2 * ph - KH) // sh + 1, (W + 2 * pw - KW) // sw + 1


def weights_to_krsc(w):
    """[K][C][KH][KW] -> [K][KH*KW][C] copy for the (r,s)-major kernels (1x1 filters need none)."""
    w = _chk(w, "w")
    K, C, KH, KW = w.shape
    wt = torch.empty((K, KH * KW, C), dtype=torch.float32, device=w.device)
    lib.rg_weights_to_krsc(_p(w), _p(wt), K, C, KH, KW, _stream())
    return wt


def conv2d_fwd(x, w, stride=1, padding=0, scale=None, shift=None, residual=None, act=ACT_NONE, slope=0.0,
               w_krsc=None):
    x, w = _chk(x, "x"), _chk(w, "w")
    N, C, H, W = x.shape
    K, Cw, KH, KW = w.shape
    if C != Cw:
        raise ValueError("conv2d_fwd: input has %d channels, weight expects %d" % (C, Cw))
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    P, Q = conv_out_size(H, W, KH, KW, (sh, sw), (ph, pw))
    y = torch.empty((N, K, P, Q), dtype=torch.float32, device=x.device)
    scale, shift, residual = _chk(scale, "scale"), _chk(shift, "shift"), _chk(residual, "residual")
    if residual is not None and residual.shape != y.shape:
        raise ValueError("conv2d_fwd: residual shape %s != output shape %s" % (tuple(residual.shape), tuple(y.shape)))
    if w_krsc is None and KH * KW > 1 and C % 16 == 0:
        w_krsc = weights_to_krsc(w)
    nbytes = _ws_query("rg_conv2d_fwd_workspace", N, C, K, KH, KW, P, Q)
    ws = workspace(nbytes, x.device) if nbytes else None
    lib.rg_conv2d_fwd(_p(x), _p(w), _p(w_krsc), _p(y), N, C, H, W, K, KH, KW, sh, sw, ph, pw, P, Q, _p(scale), _p(shift),
                      _p(residual), act, slope, _p(ws), ws.numel() if ws is not None else 0, _stream())
    return y


def conv2d_dgrad(dy, w, x_hw, stride=1, padding=0, scale=None, shift=None, residual=None, act=ACT_NONE, slope=0.0,
                 w_krsc=None, relu_mask=None, want_rowsum=False):
    """dx[N][C][H][W] from dy[N][K][P][Q] and w[K][C][KH][KW]; x_hw = (H, W) of the conv input.
    Also the forward of ConvTranspose2d (weight [in=K][out=C][KH][KW], output size x_hw)."""
    dy, w = _chk(dy, "dy"), _chk(w, "w")
    N, K, P, Q = dy.shape
    Kw, C, KH, KW = w.shape
    if K != Kw:
        raise ValueError("conv2d_dgrad: dy has %d channels, weight expects %d" % (K, Kw))
    H, W = x_hw
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    scale, shift, residual = _chk(scale, "scale"), _chk(shift, "shift"), _chk(residual, "residual")
    if residual is not None and residual.shape != dx.shape:
        raise ValueError("conv2d_dgrad: residual shape mismatch")
    if w_krsc is None and KH * KW > 1 and C % 4 == 0:
        w_krsc = weights_to_krsc(w)
    nbytes = _ws_query("rg_conv2d_dgrad_workspace", N, C, H, W, K, KH, KW, sh, sw)
    ws = workspace(nbytes, dy.device) if nbytes else None
    relu_mask = _chk(relu_mask, "relu_mask")
    if relu_mask is not None and relu_mask.shape != dx.shape:
        raise ValueError("conv2d_dgrad: relu_mask shape mismatch")
    rowsum, cols = None, 0
    if want_rowsum:
        # per-channel sums of the final dx in column blocks, written by the epilogue (0 columns: split-K / small-C launch)
        cols = _ws_query("rg_conv2d_dgrad_rowsum_cols", N, C, H, W, K, KH, KW, sh, sw, ph, pw, P, Q)
        if cols > 0:
            rowsum = torch.empty((C, cols), dtype=torch.float32, device=dy.device)
    lib.rg_conv2d_dgrad(_p(dy), _p(w), _p(w_krsc), _p(dx), N, C, H, W, K, KH, KW, sh, sw, ph, pw, P, Q, _p(scale),
                        _p(shift), _p(residual), act, slope, _p(relu_mask), _p(rowsum), cols, _p(ws),
                        ws.numel() if ws is not None else 0, _stream())
    if rowsum is not None:
        dx._rg_rowsum = rowsum          # rides with the gradient to the BatchNorm fold of the layer below (nn.conv_bn_tb)
    return dx


# ---- weight gradients on a second stream ------------------------------------------------------------------------
# Inside a network's backward program the filter gradient of a layer is a leaf: nothing downstream of it runs before
# the optimizer.  It is therefore launched on a side stream (ordered after the producer of dy by an event) and runs
# concurrently with the data-gradient chain on the main stream, filling the tail of each small grid.  The session is
# opened / joined by rg_hip.tape._NetFn.backward; operands stay referenced until the join, so the caching allocator cannot
# hand their memory to later main-stream kernels while the side stream still reads it.  RG_WGRAD_STREAM=0 disables it.
_SIDE = {"on": os.environ.get("RG_WGRAD_STREAM", "1") != "0", "sessions": {}}


# ---- streams that really run side by side --------------------------------------------------------------------------------
# HIP streams of one priority are mapped onto GPU_MAX_HW_QUEUES (default 4) hardware queues, least-used-first, and a hardware queue
# runs its packets in order.  Which queue a new stream lands on depends on everything created before it (torch's pool of 32 streams,
# RCCL's internal streams, ...): measured on config 2, the weight-gradient side stream sharing the main stream's queue costs the
# whole overlap (38.2 instead of 34.7 ms per step at GPU_MAX_HW_QUEUES=2; 39.0 instead of 34.8 with a process group and 3 queues:
# tools/debug/rccl_ab.sh).  So a second stream is PROBED, once, when it is created: an idle 150 us wave on each of two streams
# takes ~150 us when they run side by side and ~300 us when they share a queue; the first of up to 8 pool streams that runs beside
# every stream in `avoid` (earlier ones matter more) is taken.  RG_STREAM_PROBE=0: the first pool stream, unprobed.
_HANDED = {}                # device index -> streams handed out by concurrent_stream (kept apart from each other too)
_PROBE_US = 150


def _share_a_queue(a, b):
    import time
    best = None
    for _ in range(3):
        a.synchronize()
        b.synchronize()
        t0 = time.perf_counter()
        lib.rg_spin_us(_PROBE_US, a.cuda_stream)
        lib.rg_spin_us(_PROBE_US, b.cuda_stream)
        a.synchronize()
        b.synchronize()
        dt = 1e6 * (time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    if os.environ.get("RG_STREAM_PROBE_DEBUG") == "1":
        import sys
        sys.stderr.write("[stream probe] %#x vs %#x: %.0f us\n" % (a.cuda_stream, b.cuda_stream, best))
    return best > 1.6 * _PROBE_US


def concurrent_stream(device, avoid=()):
    """A stream of `device` whose kernels run beside those of the streams in `avoid` and of the streams handed out earlier."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    first = torch.cuda.Stream(device=device)
    handed = _HANDED.setdefault(device.index, [])
    # the streams to stay apart from, most important first: the caller's, then the most recently handed out ones (four hardware queues)
    apart = [s for s in list(avoid) + handed[::-1][:4] if s is not None]
    if os.environ.get("RG_STREAM_PROBE", "1") == "0" or CAPTURING[0] or not apart:
        handed.append(first)
        del handed[:-8]
        return first
    with torch.cuda.device(device):
        best, best_score, cand = first, None, first
        for i in range(8):
            score = tuple(not _share_a_queue(other, cand) for other in apart)      # compared lexicographically: earlier entries decide
            if best_score is None or score > best_score:
                best, best_score = cand, score
            if all(score):
                break
            cand = torch.cuda.Stream(device=device)
    handed.append(best)
    del handed[:-8]
    return best


_INLINE_SIDE = set()


def side_inline(stream):
    """Networks run on `stream` launch their weight gradients on `stream` itself instead of a side stream of their own: the
    number of ACTIVE hardware queues stays at four with a collective stream in the picture (main, its side stream, `stream`,
    RCCL) — a fifth makes the queue scheduler time-slice (rg_hip.parallel.init_process_group)."""
    _INLINE_SIDE.add(stream.cuda_stream)


class _SideSession(object):
    """Side stream + in-flight operand references of ONE main stream (networks running concurrently on different
    streams each get their own)."""
    __slots__ = ("stream", "depth", "refs", "used")

    def __init__(self, device, main):
        # a main stream registered with side_inline() keeps its weight gradients to itself (session stream = the stream itself)
        self.stream = main if main.cuda_stream in _INLINE_SIDE else concurrent_stream(device, avoid=(main,))
        self.depth, self.refs, self.used = 0, [], False


def _side_session(create=True):
    main = torch.cuda.current_stream()
    key = (main.device.index, main.cuda_stream)
    sess = _SIDE["sessions"].get(key)
    if sess is None and create:
        sess = _SideSession(main.device, main)
        _SIDE["sessions"][key] = sess
    return main, sess


def side_enable(on):
    _SIDE["on"] = bool(on)


def side_begin():
    if _SIDE["on"]:
        _side_session()[1].depth += 1


def side_sync():
    """the current stream waits for everything its side stream was given so far (session stays open)"""
    main, sess = _side_session(create=False)
    if sess is not None and sess.used:
        ev = torch.cuda.Event()
        ev.record(sess.stream)
        main.wait_event(ev)


def side_join():
    main, sess = _side_session(create=False)
    if sess is None:
        return
    sess.depth = max(0, sess.depth - 1)
    if sess.depth == 0 and sess.used:
        side_sync()
        sess.refs = []
        sess.used = False


# Moving a launch to the side stream costs the host ~60 us (event record / wait, stream switch: profiles/r03_host_cost.txt shows
# 3.0 ms per step for the 48 side launches of the DPTN step, whose kernels take 5-20 us each): only work that runs long enough for
# the overlap to buy something goes there.  Estimated from the algorithmic FLOPs at the rates the kernel families reach.
_SIDE_MIN_US = float(os.environ.get("RG_SIDE_MIN_US", "25"))


_GRAPH_SIDE = os.environ.get("RG_GRAPH_SIDE", "0") == "1"          # captured backward programs keep the side stream (fork / join nodes)
_GRAPH_SIDE_MIN_US = float(os.environ.get("RG_GRAPH_SIDE_MIN_US", "4"))


def side_worth(gflop, fp8=False):
    us = gflop / (0.30 if fp8 else 0.08)
    if CAPTURING[0]:
        return us >= _GRAPH_SIDE_MIN_US      # the hand-off is recorded once: only the GPU-side concurrency counts
    return us >= _SIDE_MIN_US


def side_call(fn, *operands, worth=True):
    """run `fn()` — a leaf of the backward program (weight-gradient work) — on the session's side stream, ordered behind
    everything the main stream has launched so far; `operands` (and fn's result) stay referenced until the join.  Inline when
    no session is open, the side stream is switched off, or the work is too short to be worth the hand-off (`worth`)."""
    if _SIDE["on"] and worth:
        main, sess = _side_session(create=False)
        if sess is not None and sess.depth > 0:
            ev = torch.cuda.Event()
            ev.record(main)
            sess.stream.wait_event(ev)
            with torch.cuda.stream(sess.stream):
                out = fn()
            sess.refs.append((operands, out, fn))
            sess.used = True
            return out
    return fn()


def conv2d_wgrad(x, dy, w_shape, stride=1, padding=0, out=None, side=False, after=None, fold=None):
    """dw[K][C][KH][KW]; side=True launches on the backward session's side stream; `after(dw)` is enqueued right
    behind the wgrad kernels on the same stream; `fold` = (w, scale, invstd, running_mean, sum_g, partials, dbeta, dgamma): the
    folded-BatchNorm finish of rg_conv2d_wgrad_fold (dw = scale G, dgamma, dbeta) runs inside the same call."""
    if side and _SIDE["on"] and side_worth(2e-9 * dy.numel() * w_shape[1] * w_shape[2] * w_shape[3]):
        main, sess = _side_session(create=False)
        if sess is not None and sess.depth > 0:
            x, dy = _chk(x, "x"), _chk(dy, "dy")
            dw = out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
            ev = torch.cuda.Event()
            ev.record(main)
            sess.stream.wait_event(ev)
            with torch.cuda.stream(sess.stream):
                conv2d_wgrad(x, dy, w_shape, stride, padding, out=dw, after=after, fold=fold)
            sess.refs.append((x, dy, dw, after, fold))   # the hook's closure / the fold's operands stay alive too
            sess.used = True
            return dw
    x, dy = _chk(x, "x"), _chk(dy, "dy")
    N, C, H, W = x.shape
    K, Cw, KH, KW = w_shape
    Nd, Kd, P, Q = dy.shape
    if (N, K, C) != (Nd, Kd, Cw):
        raise ValueError("conv2d_wgrad: inconsistent shapes x=%s dy=%s w=%s" % (tuple(x.shape), tuple(dy.shape), w_shape))
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dw = out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device)
    nbytes = _ws_query("rg_conv2d_wgrad_workspace", N, C, K, KH, KW, P, Q)
    ws = workspace(nbytes, x.device)
    if fold is not None:
        fw, fscale, finvstd, fmean, fsum, fpart, fdbeta, fdgamma = fold
        lib.rg_conv2d_wgrad_fold(_p(x), _p(dy), _p(dw), N, C, H, W, K, KH, KW, sh, sw, ph, pw, P, Q, _p(fw), _p(fscale),
                                 _p(finvstd), _p(fmean), _p(fsum), _p(fpart), fpart.shape[1] if fpart is not None else 0,
                                 _p(fdbeta), _p(fdgamma), _p(ws), ws.numel(), _stream())
    else:
        lib.rg_conv2d_wgrad(_p(x), _p(dy), _p(dw), N, C, H, W, K, KH, KW, sh, sw, ph, pw, P, Q, _p(ws), ws.numel(),
                            _stream())
    if after is not None:
        after(dw)
    return dw


def linear_fwd(x, w, bias=None):
    """y[B][out] = x[B][in] @ w[out][in]^T + bias, as a 1x1 conv on a 1x1 map."""
    B, Cin = x.shape
    y = conv2d_fwd(x.view(B, Cin, 1, 1), w.view(w.shape[0], Cin, 1, 1), shift=bias)
    return y.view(B, w.shape[0])


def linear_dgrad(dy, w):
    B, K = dy.shape
    dx = conv2d_dgrad(dy.view(B, K, 1, 1), w.view(K, w.shape[1], 1, 1), (1, 1))
    return dx.view(B, w.shape[1])


def linear_wgrad(x, dy, out=None):
    B, Cin = x.shape
    K = dy.shape[1]
    return conv2d_wgrad(x.view(B, Cin, 1, 1), dy.view(B, K, 1, 1), (K, Cin, 1, 1), out=out).view(K, Cin)


# ------------------------------------------------------------------------------------------------
# batch norm on [N][C][HW]
# ------------------------------------------------------------------------------------------------
def _nchw(x):
    if x.dim() == 2:
        return x.shape[0], x.shape[1], 1
    N, C = x.shape[0], x.shape[1]
    return N, C, x.numel() // (N * C)


def _same_size(what, ref, **others):
    """element-wise operands must cover the same elements: the C ABI takes one length, so a mismatch (e.g. a residual branch with
    another stride, as in the reference's ResNet-18 wrapper) would read out of bounds instead of failing like torch does"""
    for name, t in others.items():
        if t is not None and t.numel() != ref.numel():
            raise RuntimeError("%s: the size of tensor %s %s must match %s" % (what, name, tuple(t.shape), tuple(ref.shape)))


def bn_stats(x, running_mean=None, running_var=None, eps=1e-5, momentum=0.1):
    x = _chk(x, "x")
    N, C, HW = _nchw(x)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(_ws_query("rg_bn_workspace", N, C, HW), x.device)
    lib.rg_bn_stats(_p(x), _p(mean), _p(invstd), _p(running_mean), _p(running_var), N, C, HW, eps, momentum, _p(ws),
                    ws.numel(), _stream())
    return mean, invstd


def bn_apply_fwd(x, mean, stat, gamma, beta, residual=None, stat_is_var=False, eps=1e-5, act=ACT_NONE, slope=0.0):
    x = _chk(x, "x")
    residual = _chk(residual, "residual")
    _same_size("bn_apply_fwd", x, residual=residual)
    N, C, HW = _nchw(x)
    y = torch.empty_like(x)
    lib.rg_bn_apply_fwd(_p(x), _p(mean), _p(stat), _p(gamma), _p(beta), _p(residual), _p(y), N, C, HW,
                        int(stat_is_var), eps, act, slope, _stream())
    return y


def bn_bwd_reduce(x, dy, y_act, mean, stat, stat_is_var=False, eps=1e-5, act=ACT_NONE, slope=0.0,
                  out_sum_dy=None, out_sum_dy_xhat=None):
    x, dy, y_act = _chk(x, "x"), _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("bn_bwd_reduce", x, dy=dy, y=y_act)
    N, C, HW = _nchw(x)
    sum_dy = out_sum_dy if out_sum_dy is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    sum_dy_xhat = out_sum_dy_xhat if out_sum_dy_xhat is not None else \
        torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(_ws_query("rg_bn_workspace", N, C, HW), x.device)
    lib.rg_bn_bwd_reduce(_p(x), _p(dy), _p(y_act), _p(mean), _p(stat), _p(sum_dy), _p(sum_dy_xhat), N, C, HW,
                         int(stat_is_var), eps, act, slope, _p(ws), ws.numel(), _stream())
    return sum_dy, sum_dy_xhat


_BN_FUSED = os.environ.get("RG_BN_FUSED", "1") != "0"


def bn_train_fused_ok(x):
    """does this activation qualify for the one-launch train-mode BatchNorm kernels (small per-channel extent, many channels)?"""
    if not _BN_FUSED:
        return False
    N, C, HW = _nchw(x)
    return bool(_ws_query("rg_bn_train_fused_ok", N, C, HW))


def bn_train_fwd_fused(x, gamma, beta, residual, running_mean, running_var, eps, momentum, act=ACT_NONE, slope=0.0):
    """batch statistics + running-statistics update + normalise + affine + residual + activation -> (y, mean[C], invstd[C])"""
    x, residual = _chk(x, "x"), _chk(residual, "residual")
    _same_size("bn_train_fwd_fused", x, residual=residual)
    N, C, HW = _nchw(x)
    y = torch.empty_like(x)
    stats = torch.empty(2, C, dtype=torch.float32, device=x.device)
    lib.rg_bn_train_fwd_fused(_p(x), _p(gamma), _p(beta), _p(residual), _p(y), _p(stats[0]), _p(stats[1]), _p(running_mean),
                              _p(running_var), N, C, HW, eps, momentum, act, slope, _stream())
    return y, stats[0], stats[1]


def bn_train_bwd_fused(x, dy, y_act, mean, invstd, gamma, act=ACT_NONE, slope=0.0, need_dx=True, need_dres=False,
                       out_sum_dy=None, out_sum_dy_xhat=None):
    """-> (dx, dres, sum_g[C], sum_g_xhat[C]) in one launch"""
    x, dy, y_act = _chk(x, "x"), _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("bn_train_bwd_fused", dy, x=x, y=y_act)
    N, C, HW = _nchw(x)
    dx = torch.empty_like(dy) if need_dx else None
    dres = torch.empty_like(dy) if need_dres else None
    s1 = out_sum_dy if out_sum_dy is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    s2 = out_sum_dy_xhat if out_sum_dy_xhat is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    lib.rg_bn_train_bwd_fused(_p(x), _p(dy), _p(y_act), _p(mean), _p(invstd), _p(gamma), _p(dx), _p(dres), _p(s1), _p(s2), N, C, HW,
                              act, slope, _stream())
    return dx, dres, s1, s2


def instnorm_fwd(x, gamma=None, beta=None, residual=None, eps=1e-5, act=ACT_NONE, slope=0.0):
    """InstanceNorm over the trailing dims of x[N][C][...] in one launch -> (y, mean[N*C], invstd[N*C])."""
    x, residual = _chk(x, "x"), _chk(residual, "residual")
    _same_size("instnorm_fwd", x, residual=residual)
    N, C, HW = _nchw(x)
    y = torch.empty_like(x)
    stats = torch.empty(2, N * C, dtype=torch.float32, device=x.device)
    lib.rg_instnorm_fwd(_p(x), _p(gamma), _p(beta), _p(residual), _p(y), _p(stats[0]), _p(stats[1]), N, C, HW, eps, act, slope,
                        _stream())
    return y, stats[0], stats[1]


def instnorm_bwd(x, dy, y_act, mean, invstd, gamma=None, act=ACT_NONE, slope=0.0, need_dx=True, need_dres=False):
    """-> (dx, dres, sum_g[N*C], sum_g_xhat[N*C], sum_dx[N*C]) in one launch."""
    x, dy, y_act = _chk(x, "x"), _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("instnorm_bwd", dy, x=x, y=y_act)
    N, C, HW = _nchw(x)
    dx = torch.empty_like(dy) if need_dx else None
    dres = torch.empty_like(dy) if need_dres else None
    sums = torch.empty(3, N * C, dtype=torch.float32, device=x.device)
    lib.rg_instnorm_bwd(_p(x), _p(dy), _p(y_act), _p(mean), _p(invstd), _p(gamma), _p(dx), _p(dres), _p(sums[0]), _p(sums[1]),
                        _p(sums[2]) if need_dx else None, N, C, HW, act, slope, _stream())
    return dx, dres, sums[0], sums[1], (sums[2] if need_dx else None)


def rows_sum_pair(a, b, N, C, out_a=None, out_b=None):
    """column sums of two [N][C] matrices in one launch (InstanceNorm affine gradients)."""
    dev = (a if a is not None else b).device
    if a is not None and out_a is None:
        out_a = torch.empty(C, dtype=torch.float32, device=dev)
    if b is not None and out_b is None:
        out_b = torch.empty(C, dtype=torch.float32, device=dev)
    lib.rg_rows_sum_pair(_p(a), _p(b), _p(out_a), _p(out_b), N, C, _stream())
    return out_a, out_b


def bn_bwd_apply(x, dy, y_act, mean, stat, gamma, sum_dy, sum_dy_xhat, train, stat_is_var=False, eps=1e-5,
                 act=ACT_NONE, slope=0.0, need_dx=True, need_dres=False):
    x, dy, y_act = _chk(x, "x"), _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("bn_bwd_apply", dy, x=x, y=y_act)
    N, C, HW = _nchw(dy)
    dx = torch.empty_like(dy) if need_dx else None
    dres = torch.empty_like(dy) if need_dres else None
    lib.rg_bn_bwd_apply(_p(x), _p(dy), _p(y_act), _p(mean), _p(stat), _p(gamma), _p(sum_dy), _p(sum_dy_xhat), _p(dx),
                        _p(dres), N, C, HW, int(train), int(stat_is_var), eps, act, slope, _stream())
    return dx, dres


def bn_eval_bwd(x, dy, y_act, running_mean, running_var, gamma, eps=1e-5, act=ACT_NONE, slope=0.0, need_dx=True,
                need_dres=False, need_sums=True, out_sum_dy=None, out_sum_dy_xhat=None):
    """Eval-mode BatchNorm backward in one pass -> (dx, dres, sum_dy, sum_dy_xhat)."""
    x, dy, y_act = _chk(x, "x"), _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("bn_eval_bwd", dy, x=x, y=y_act)
    N, C, HW = _nchw(dy)
    dx = torch.empty_like(dy) if need_dx else None
    dres = torch.empty_like(dy) if need_dres else None
    s1 = s2 = None
    if need_sums:
        s1 = out_sum_dy if out_sum_dy is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
        s2 = out_sum_dy_xhat if out_sum_dy_xhat is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
    ws = workspace(_ws_query("rg_bn_workspace", N, C, HW), dy.device)
    lib.rg_bn_eval_bwd(_p(x), _p(dy), _p(y_act), _p(running_mean), _p(running_var), _p(gamma), _p(dx), _p(dres), _p(s1),
                       _p(s2), N, C, HW, eps, act, slope, _p(ws), ws.numel(), _stream())
    return dx, dres, s1, s2


def bn_fold(gamma, beta, running_mean, running_var, eps=1e-5):
    """(scale, shift, invstd) of a frozen-statistics BatchNorm: y = z * scale + shift."""
    C = running_mean.numel()
    buf = torch.empty(3, C, dtype=torch.float32, device=running_mean.device)
    lib.rg_bn_fold(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(buf[0]), _p(buf[1]), _p(buf[2]), C,
                   _stream())
    return buf[0], buf[1], buf[2]


def act_bwd_sum(dy, y_act, act, slope=0.0, need_g=True, need_sum=True, out_sum=None):
    """g = dy * act'(y) and its per-channel sums over N, HW -> (g, sum_g)."""
    dy, y_act = _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("act_bwd_sum", dy, y=y_act)
    N, C, HW = _nchw(dy)
    g = torch.empty_like(dy) if need_g else None
    sg = None
    if need_sum:
        sg = out_sum if out_sum is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
    ws = workspace(_ws_query("rg_bn_workspace", N, C, HW), dy.device)
    lib.rg_act_bwd_sum(_p(dy), _p(y_act), _p(g), _p(sg), N, C, HW, act, slope, _p(ws), ws.numel(), _stream())
    return g, sg


def bn_fold_wgrad(w, g, scale, invstd, running_mean, sum_g, dgamma=None, partials=None, dbeta=None):
    """in place: g (= wgrad of the un-normalised output gradient) *= scale per filter; dgamma from sum(w * g); the channel
    sums come as sum_g[K] or as slice partials [K, S] (then dbeta is written here too)."""
    K = w.shape[0]
    M = w.numel() // K
    S = partials.shape[1] if partials is not None else 0
    lib.rg_bn_fold_wgrad(_p(w), _p(g), _p(scale), _p(invstd), _p(running_mean), _p(sum_g), _p(partials), S, _p(dbeta),
                         _p(dgamma), K, M, _stream())
    return g


def act_bwd_partial(dy, y_act, act, slope=0.0, need_g=True):
    """g = dy * act'(y) (or None) and the slice partials [C, S] of its channel sums (no finalize launch)."""
    dy, y_act = _chk(dy, "dy"), _chk(y_act, "y")
    _same_size("act_bwd_partial", dy, y=y_act)
    N, C, HW = _nchw(dy)
    g = torch.empty_like(dy) if need_g else None
    S = _ws_query("rg_bn_slices", N, C, HW)
    part = torch.empty((C, S), dtype=torch.float32, device=dy.device)
    lib.rg_act_bwd_partial(_p(dy), _p(y_act), _p(g), _p(part), N, C, HW, act, slope, _stream())
    return g, part


def fold_filters_multi(table, n_pairs, blocks):
    lib.rg_fold_filters_multi(_p(table), n_pairs, blocks, _stream())


def scale_rows(w, scale):
    w = _chk(w, "w")
    out = torch.empty_like(w)
    K = w.shape[0]
    lib.rg_scale_rows(_p(w), _p(scale), _p(out), K, w.numel() // K, _stream())
    return out


def channel_sum(dy, out=None):
    """sum over N and HW of dy[N][C][HW] (bias gradient)."""
    dy = _chk(dy, "dy")
    N, C, HW = _nchw(dy)
    if _ws_query("rg_channel_sum_ok", N, C, HW):
        s = out if out is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
        lib.rg_channel_sum(_p(dy), _p(s), N, C, HW, _stream())
        return s
    s, _ = bn_bwd_reduce(dy, dy, None, const_fill(C, 0.0, dy.device), const_fill(C, 1.0, dy.device), out_sum_dy=out)
    return s


_CONST_FILL = {}


def const_fill(n, v, device):
    """a cached read-only device vector of n copies of v (identity mean / invstd, unit cotangents): filled once, never on the
    step path afterwards"""
    key = (n, float(v), device.index)
    t = _CONST_FILL.get(key)
    if t is None:
        t = _CONST_FILL[key] = fill_(torch.empty(n, dtype=torch.float32, device=device), v)
    return t


# ------------------------------------------------------------------------------------------------
# element-wise
# ------------------------------------------------------------------------------------------------
def act_fwd(x, act, slope=0.0, out=None):
    x = _chk(x, "x")
    y = out if out is not None else torch.empty_like(x)
    lib.rg_act_fwd(_p(x), _p(y), x.numel(), act, slope, _stream())
    return y


def act_bwd(dy, y, act, slope=0.0):
    dy, y = _chk(dy, "dy"), _chk(y, "y")
    _same_size("act_bwd", dy, y=y)
    dx = torch.empty_like(dy)
    lib.rg_act_bwd(_p(dy), _p(y), _p(dx), dy.numel(), act, slope, _stream())
    return dx


def axpby(a, b, alpha=1.0, beta=1.0, out=None):
    a, b = _chk(a, "a"), _chk(b, "b")
    _same_size("axpby", a, b=b, out=out)
    y = out if out is not None else torch.empty_like(a)
    if out is not None and getattr(out, "_rg_inst_sums", None) is not None:
        out._rg_inst_sums = None        # accumulating into an InstanceNorm's dx invalidates the sums that kernel attached to it
    lib.rg_axpby(_p(a), _p(b), _p(y), a.numel(), alpha, beta, _stream())
    return y


def add(a, b):
    return axpby(a, b, 1.0, 1.0)


def scale(a, alpha):
    return axpby(a, None, alpha, 0.0)


def pair_cat(a, b, take_a=None):
    """[2B, ...]: rows 0..B-1 = a, rows B.. = a[i] where take_a[i] != 0 else b[i] (take_a None: b) — one launch."""
    a, b = _chk(a, "a"), _chk(b, "b")
    _same_size("pair_cat", a, b=b)
    take_a = _chk(take_a, "take_a", torch.int64)
    B = a.shape[0]
    out = torch.empty((2 * B,) + tuple(a.shape[1:]), dtype=torch.float32, device=a.device)
    lib.rg_pair_cat(_p(a), _p(b), _p(take_a), _p(out), B, a.numel() // B, _stream())
    return out


def fill_(t, v):
    lib.rg_fill(_p(t), t.numel(), float(v), _stream())
    return t


def zeros(shape, device):
    return fill_(torch.empty(shape, dtype=torch.float32, device=device), 0.0)


def sub_square_fwd(a, b):
    a, b = _chk(a, "a"), _chk(b, "b")
    y = torch.empty_like(a)
    lib.rg_sub_square_fwd(_p(a), _p(b), _p(y), a.numel(), _stream())
    return y


def sub_square_bwd(a, b, dy, need_a=True, need_b=True):
    a, b, dy = _chk(a, "a"), _chk(b, "b"), _chk(dy, "dy")
    da = torch.empty_like(a) if need_a else None
    db = torch.empty_like(a) if need_b else None
    lib.rg_sub_square_bwd(_p(a), _p(b), _p(dy), _p(da), _p(db), a.numel(), _stream())
    return da, db


_CLOCK = {}


def step_clock(device):
    """device-side step counter (uint64 as int64 storage) mixed into dropout seeds; advanced by rg_hip.graph per captured step"""
    c = _CLOCK.get(device.index)
    if c is None:
        c = _CLOCK[device.index] = torch.zeros(1, dtype=torch.int64, device=device)
    return c


def advance_step_clock(device):
    lib.rg_u64_add(_p(step_clock(device)), 1, _stream())


def dropout(x, p, seed):
    """keep iff hash(seed + K * clock, element) >= p 2^32; the clock is 0 unless a captured step advances it (rg_hip.graph), so
    eager runs reproduce the (seed, index) -> bit function of the oracle exactly"""
    x = _chk(x, "x")
    y = torch.empty_like(x)
    lib.rg_dropout_clocked(_p(x), _p(y), x.numel(), p, seed, _p(step_clock(x.device)), _stream())
    return y


def l2norm_rows_fwd(x, eps=1e-12):
    x = _chk(x, "x")
    rows, D = x.shape
    y = torch.empty_like(x)
    norm = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib.rg_l2norm_rows_fwd(_p(x), _p(y), _p(norm), rows, D, eps, _stream())
    return y, norm


def l2norm_rows_bwd(y, dy, norm, eps=1e-12):
    y, dy = _chk(y, "y"), _chk(dy, "dy")
    rows, D = y.shape
    dx = torch.empty_like(y)
    lib.rg_l2norm_rows_bwd(_p(y), _p(dy), _p(norm), _p(dx), rows, D, eps, _stream())
    return dx


def l2norm_channels_fwd(x, eps=1e-12):
    """F.normalize(x, dim=1) of [N, C, H, W] -> (y, norm [N, H*W]) without permuted copies"""
    x = _chk(x, "x")
    N, C, HW = _nchw(x)
    y = torch.empty_like(x)
    norm = torch.empty((N, HW), dtype=torch.float32, device=x.device)
    lib.rg_l2norm_channels_fwd(_p(x), _p(y), _p(norm), N, C, HW, eps, _stream())
    return y, norm


def l2norm_channels_bwd(y, dy, norm, eps=1e-12):
    y, dy = _chk(y, "y"), _chk(dy, "dy")
    _same_size("l2norm_channels_bwd", y, dy=dy)
    N, C, HW = _nchw(y)
    dx = torch.empty_like(y)
    lib.rg_l2norm_channels_bwd(_p(y), _p(dy), _p(norm), _p(dx), N, C, HW, eps, _stream())
    return dx


def copy_channels(src, dst, c_count, src_c0, dst_c0, accumulate=False):
    src, dstc = _chk(src, "src"), dst
    if not dst.is_contiguous():
        raise ValueError("copy_channels: destination must be contiguous")
    N, Cs, HW = _nchw(src)
    Nd, Cd, HWd = _nchw(dst)
    if N != Nd or HW != HWd:
        raise ValueError("copy_channels: batch / spatial mismatch")
    lib.rg_copy_channels(_p(src), _p(dstc), N, c_count, HW, Cs, src_c0, Cd, dst_c0, int(accumulate), _stream())
    return dst


def bicubic_normalize_fwd(x, size, mean=None, std=None):
    """[N,C,H,W] -> [N,C,OH,OW] bicubic (align_corners False) then (v - mean[c]) / std[c]; mean/std device [C] or None."""
    x = _chk(x, "x")
    if x.dim() != 4:
        raise ValueError("bicubic_normalize: expected NCHW")
    N, C, H, W = x.shape
    OH, OW = int(size[0]), int(size[1])
    y = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device)
    lib.rg_bicubic_normalize_fwd(_p(x), _p(y), N, C, H, W, OH, OW, _p(mean) if mean is not None else None,
                                 _p(std) if std is not None else None, _stream())
    return y


def bicubic_normalize_bwd(dy, in_hw, std=None):
    dy = _chk(dy, "dy")
    N, C, OH, OW = dy.shape
    H, W = int(in_hw[0]), int(in_hw[1])
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    lib.rg_bicubic_normalize_bwd(_p(dy), _p(dx), N, C, H, W, OH, OW, _p(std) if std is not None else None, _stream())
    return dx


def avgpool2d_fwd(x, k=2):
    x = _chk(x, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H // k, W // k), dtype=torch.float32, device=x.device)
    lib.rg_avgpool2d_fwd(_p(x), _p(y), N, C, H, W, k, _stream())
    return y


def avgpool2d_bwd(dy, x_shape, k=2):
    dy = _chk(dy, "dy")
    N, C, H, W = x_shape
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    lib.rg_avgpool2d_bwd(_p(dy), _p(dx), N, C, H, W, k, _stream())
    return dx


def reflection_pad_fusable(x, pad, act, slope):
    """can rg_reflection_pad2d_* fold this activation in? (ReLU / LeakyReLU with a positive slope, float4 rows, pad <= 3)"""
    return (act == ACT_RELU or (act == ACT_LEAKY and slope > 0.0)) and x.shape[3] % 4 == 0 and x.shape[3] >= 8 and pad <= 3


def reflection_pad2d_fwd(x, pad, act=ACT_NONE, slope=0.0):
    """pad(act(x)) in one pass"""
    x = _chk(x, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, C, H + 2 * pad, W + 2 * pad), dtype=torch.float32, device=x.device)
    lib.rg_reflection_pad2d_fwd(_p(x), _p(y), N, C, H, W, pad, act, slope, _stream())
    return y


def reflection_pad2d_bwd(dy, pad, x_act=None, act=ACT_NONE, slope=0.0):
    """act'(x_act) * pad^T(dy) in one pass (x_act: what the forward padded; only with act)"""
    dy, x_act = _chk(dy, "dy"), _chk(x_act, "x_act")
    N, C, OH, OW = dy.shape
    H, W = OH - 2 * pad, OW - 2 * pad
    if act != ACT_NONE and (x_act is None or tuple(x_act.shape) != (N, C, H, W)):
        raise ValueError("reflection_pad2d_bwd: the fused activation backward needs the forward input [N, C, H, W]")
    dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device)
    lib.rg_reflection_pad2d_bwd(_p(dy), _p(x_act) if act != ACT_NONE else None, _p(dx), N, C, H, W, pad, act, slope, _stream())
    return dx


def spectral_norm_fwd(w, u, v, training=True, eps=1e-12, save_uv=False):
    """One power iteration in place on u, v (training) and W / sigma; returns (w_sn, sigma[2] = (sigma, 1/sigma)) and, with
    `save_uv`, private copies of the u, v this forward used (written by the same launch)."""
    w = _chk(w, "w")
    K = w.shape[0]
    M = w.numel() // K
    if u.numel() != K or v.numel() != M or not (u.is_contiguous() and v.is_contiguous()):
        raise ValueError("spectral_norm: u / v do not match the weight matrix %d x %d" % (K, M))
    w_sn = torch.empty_like(w)
    sigma = torch.empty(2, dtype=torch.float32, device=w.device)
    saved = torch.empty(K + M, dtype=torch.float32, device=w.device) if save_uv else None
    lib.rg_spectral_norm_fwd(_p(w), _p(u), _p(v), _p(w_sn), _p(sigma), _p(saved), K, M, int(bool(training)), eps, _stream())
    if save_uv:
        return w_sn, sigma, saved[:K], saved[K:]
    return w_sn, sigma


class _SNDesc(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("w_sn", ctypes.c_void_p),
                ("sigma", ctypes.c_void_p), ("uv_saved", ctypes.c_void_p), ("K", ctypes.c_int), ("M", ctypes.c_int)]


def spectral_norm_fwd_multi(items, training=True, eps=1e-12, save_uv=False):
    """`items`: [(w, u, v)] of one network forward -> [(w_sn, sigma, u_saved, v_saved)] in two launches (rg_spectral_norm_fwd_multi);
    outputs are views of one allocation."""
    n = len(items)
    dev = items[0][0].device
    sizes = []
    total = 0
    for w, u, v in items:
        w = _chk(w, "w")
        K = w.shape[0]
        M = w.numel() // K
        if u.numel() != K or v.numel() != M or not (u.is_contiguous() and v.is_contiguous()):
            raise ValueError("spectral_norm: u / v do not match the weight matrix %d x %d" % (K, M))
        sizes.append((K, M, total))
        total += (K * M + 3) // 4 * 4 + 4 + ((K + M + 3) // 4 * 4 if save_uv else 0)
    buf = torch.empty(total, dtype=torch.float32, device=dev)
    base = buf.data_ptr()
    descs = (_SNDesc * n)()
    out = []
    for i, ((w, u, v), (K, M, off)) in enumerate(zip(items, sizes)):
        o_sig = off + (K * M + 3) // 4 * 4
        o_uv = o_sig + 4
        d = descs[i]
        d.w, d.u, d.v = w.data_ptr(), u.data_ptr(), v.data_ptr()
        d.w_sn, d.sigma = base + 4 * off, base + 4 * o_sig
        d.uv_saved = base + 4 * o_uv if save_uv else None
        d.K, d.M = K, M
        out.append((buf[off:off + K * M].view(w.shape), buf[o_sig:o_sig + 2],
                    buf[o_uv:o_uv + K] if save_uv else None, buf[o_uv + K:o_uv + K + M] if save_uv else None))
    lib.rg_spectral_norm_fwd_multi(ctypes.addressof(descs), n, int(bool(training)), eps, _stream())
    return out


def spectral_norm_bwd(dw_sn, w_sn, u, v, sigma, out=None, accumulate=False):
    dw_sn = _chk(dw_sn, "dw_sn")
    K = w_sn.shape[0]
    M = w_sn.numel() // K
    dw = out if out is not None else torch.empty_like(w_sn)
    need = _ws_query("rg_spectral_norm_bwd_workspace", K, M)
    ws = workspace(need, w_sn.device) if need else None
    lib.rg_spectral_norm_bwd(_p(dw_sn), _p(w_sn), _p(u), _p(v), _p(sigma), _p(dw), K, M, int(accumulate), _p(ws),
                             ws.numel() if ws is not None else 0, _stream())
    return dw


def bgemm(A, B, C, M, N, K, a_str, b_str, c_str, batch, a_b, b_b, c_b, alpha=1.0, beta=0.0):
    """C[b0][b1] = alpha A[b0][b1] . B[b0][b1] + beta C; a_str = (row, k) strides of A, b_str = (k, col) of B,
    c_str = (row, col) of C, batch = (n0, n1), *_b = (stride b0, stride b1); all in elements.  Tensors are base buffers."""
    lib.rg_bgemm(_p(A), _p(B), _p(C), M, N, K, a_str[0], a_str[1], b_str[0], b_str[1], c_str[0], c_str[1], batch[0], batch[1],
                 a_b[0], a_b[1], b_b[0], b_b[1], c_b[0], c_b[1], alpha, beta, _stream())
    return C


def softmax_rows_fwd(x, scale=1.0, out=None):
    x = _chk(x, "x")
    cols = x.shape[-1]
    y = out if out is not None else torch.empty_like(x)
    lib.rg_softmax_rows_fwd(_p(x), _p(y), x.numel() // cols, cols, scale, _stream())
    return y


def softmax_rows_bwd(p, dp, scale=1.0, out=None):
    p, dp = _chk(p, "p"), _chk(dp, "dp")
    cols = p.shape[-1]
    ds = out if out is not None else torch.empty_like(p)
    lib.rg_softmax_rows_bwd(_p(p), _p(dp), _p(ds), p.numel() // cols, cols, scale, _stream())
    return ds


def mix_rows_fwd(src, ia, ib, lam):
    src = _chk(src, "src")
    ia, ib = _chk(ia, "idx_a", torch.int64), _chk(ib, "idx_b", torch.int64)
    R, n = src.shape[0], ia.numel()
    length = src.numel() // R
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    lib.rg_mix_rows_fwd(_p(src), _p(ia), _p(ib), lam, _p(out), R, n, length, _stream())
    return out


def mix_rows_bwd(g, ia, ib, lam, src_shape):
    g = _chk(g, "g")
    R, n = src_shape[0], ia.numel()
    dsrc = torch.empty(tuple(src_shape), dtype=torch.float32, device=g.device)
    lib.rg_mix_rows_bwd(_p(g), _p(ia), _p(ib), lam, _p(dsrc), R, n, dsrc.numel() // R, _stream())
    return dsrc


def cat_channels(tensors):
    """torch.cat(tensors, dim=1) for NCHW (or NC) tensors."""
    N = tensors[0].shape[0]
    Ct = sum(t.shape[1] for t in tensors)
    out = torch.empty((N, Ct) + tuple(tensors[0].shape[2:]), dtype=torch.float32, device=tensors[0].device)
    c0 = 0
    for t in tensors:
        copy_channels(t, out, t.shape[1], 0, c0)
        c0 += t.shape[1]
    return out


def slice_channels(src, c0, c1):
    out = torch.empty((src.shape[0], c1 - c0) + tuple(src.shape[2:]), dtype=torch.float32, device=src.device)
    return copy_channels(src, out, c1 - c0, c0, 0)


# ------------------------------------------------------------------------------------------------
# pooling
# ------------------------------------------------------------------------------------------------
def maxpool2d_fwd(x, kernel=3, stride=2, padding=1):
    x = _chk(x, "x")
    N, C, H, W = x.shape
    kh, kw = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    P, Q = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    y = torch.empty((N, C, P, Q), dtype=torch.float32, device=x.device)
    arg = torch.empty((N, C, P, Q), dtype=torch.uint8, device=x.device)
    lib.rg_maxpool2d_fwd(_p(x), _p(y), _p(arg), N, C, H, W, kh, kw, sh, sw, ph, pw, P, Q, _stream())
    return y, arg


def maxpool2d_bwd(dy, arg, x_shape, kernel=3, stride=2, padding=1):
    dy = _chk(dy, "dy")
    N, C, H, W = x_shape
    kh, kw = _pair(kernel)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    P, Q = dy.shape[2], dy.shape[3]
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    lib.rg_maxpool2d_bwd(_p(dy), _p(arg), _p(dx), N, C, H, W, kh, kw, sh, sw, ph, pw, P, Q, _stream())
    return dx


def global_avgpool_fwd(x):
    x = _chk(x, "x")
    N, C, HW = _nchw(x)
    y = torch.empty((N, C), dtype=torch.float32, device=x.device)
    lib.rg_global_avgpool_fwd(_p(x), _p(y), N, C, HW, _stream())
    return y


def global_avgpool_bwd(dy, x_shape):
    dy = _chk(dy, "dy")
    N, C = x_shape[0], x_shape[1]
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    lib.rg_global_avgpool_bwd(_p(dy), _p(dx), N, C, dx.numel() // (N * C), _stream())
    return dx


def gem_pool_fwd(x, p, eps=1e-6):
    x, p = _chk(x, "x"), _chk(p, "p")
    N, C, HW = _nchw(x)
    y = torch.empty((N, C), dtype=torch.float32, device=x.device)
    lib.rg_gem_pool_fwd(_p(x), _p(p), _p(y), N, C, HW, eps, _stream())
    return y


def gem_pool_bwd(x, p, y, dy, eps=1e-6, need_dp=True):
    x, p, y, dy = _chk(x, "x"), _chk(p, "p"), _chk(y, "y"), _chk(dy, "dy")
    N, C, HW = _nchw(x)
    dx = torch.empty_like(x)
    dp = torch.empty(1, dtype=torch.float32, device=x.device) if need_dp else None
    ws = workspace(N * C * 4, x.device)
    lib.rg_gem_pool_bwd(_p(x), _p(p), _p(y), _p(dy), _p(dx), _p(dp), N, C, HW, eps, _p(ws), ws.numel(), _stream())
    return dx, dp


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
def _loss_ws(device):
    return workspace(lib.rg_loss_workspace(), device)


def sigmoid_bce_fwd(x, target):
    x = _chk(x, "x")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = _loss_ws(x.device)
    lib.rg_sigmoid_bce_fwd(_p(x), _p(out), x.numel(), float(target), _p(ws), ws.numel(), _stream())
    return out


def sigmoid_bce_bwd(x, grad_out, target, grad_scale=1.0):
    x = _chk(x, "x")
    dx = torch.empty_like(x)
    lib.rg_sigmoid_bce_bwd(_p(x), _p(grad_out), _p(dx), x.numel(), float(target), grad_scale, _stream())
    return dx


def mse_const_fwd(x, target):
    x = _chk(x, "x")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    ws = _loss_ws(x.device)
    lib.rg_mse_const_fwd(_p(x), _p(out), x.numel(), float(target), _p(ws), ws.numel(), _stream())
    return out


def mse_const_bwd(x, grad_out, target, grad_scale=1.0):
    x = _chk(x, "x")
    dx = torch.empty_like(x)
    lib.rg_mse_const_bwd(_p(x), _p(grad_out), _p(dx), x.numel(), float(target), grad_scale, _stream())
    return dx


def affine_relu_mean_fwd(x, a, b, clamp):
    x = _chk(x, "x")
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    ws = _loss_ws(x.device)
    lib.rg_affine_relu_mean_fwd(_p(x), _p(loss), x.numel(), a, b, int(clamp), _p(ws), ws.numel(), _stream())
    return loss


def affine_relu_mean_bwd(x, grad_out, a, b, clamp, grad_scale=1.0):
    x = _chk(x, "x")
    dx = torch.empty_like(x)
    lib.rg_affine_relu_mean_bwd(_p(x), _p(grad_out), _p(dx), x.numel(), a, b, int(clamp), grad_scale, _stream())
    return dx


def grad_penalty_rows(g, constant, scale):
    """(pen[rows], v[rows, D]): pen[r] = scale * (|g_r + 1e-16| - constant)^2 and its gradient with respect to g_r."""
    g = _chk(g, "g")
    rows, D = g.shape
    pen = torch.empty(rows, dtype=torch.float32, device=g.device)
    v = torch.empty_like(g)
    lib.rg_grad_penalty_rows(_p(g), _p(pen), _p(v), rows, D, float(constant), float(scale), _stream())
    return pen, v


def l1_fwd(a, b, row_labels=None):
    a, b = _chk(a, "a"), _chk(b, "b")
    row_labels = _chk(row_labels, "row_labels", torch.int64)
    rows = a.shape[0]
    inner = a.numel() // rows
    out2 = torch.empty(2, dtype=torch.float32, device=a.device)
    ws = _loss_ws(a.device)
    lib.rg_l1_fwd(_p(a), _p(b), _p(row_labels), _p(out2), rows, inner, _p(ws), ws.numel(), _stream())
    return out2


def l1_bwd(a, b, row_labels, grad_out, out2, need_a=True, need_b=True, grad_scale=1.0):
    a, b = _chk(a, "a"), _chk(b, "b")
    rows = a.shape[0]
    inner = a.numel() // rows
    da = torch.empty_like(a) if need_a else None
    db = torch.empty_like(a) if need_b else None
    lib.rg_l1_bwd(_p(a), _p(b), _p(row_labels), _p(grad_out), _p(out2), _p(da), _p(db), rows, inner, grad_scale,
                  _stream())
    return da, db


def l1_rows_fwd(a, b):
    a, b = _chk(a, "a"), _chk(b, "b")
    _same_size("l1_rows_fwd", a, b=b)
    rows = a.shape[0]
    out = torch.empty(rows, dtype=torch.float32, device=a.device)
    lib.rg_l1_rows_fwd(_p(a), _p(b), _p(out), rows, a.numel() // rows, _stream())
    return out


def l1_rows_bwd(a, b, grad_rows, need_a=True, need_b=True):
    a, b, grad_rows = _chk(a, "a"), _chk(b, "b"), _chk(grad_rows, "grad_rows")
    rows = a.shape[0]
    da = torch.empty_like(a) if need_a else None
    db = torch.empty_like(a) if need_b else None
    lib.rg_l1_rows_bwd(_p(a), _p(b), _p(grad_rows), _p(da), _p(db), rows, a.numel() // rows, _stream())
    return da, db


def mse_const_rows_fwd(x, c):
    x = _chk(x, "x")
    rows = x.shape[0]
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib.rg_mse_const_rows_fwd(_p(x), float(c), _p(out), rows, x.numel() // rows, _stream())
    return out


def mse_const_rows_bwd(x, c, grad_rows):
    x, grad_rows = _chk(x, "x"), _chk(grad_rows, "grad_rows")
    rows = x.shape[0]
    dx = torch.empty_like(x)
    lib.rg_mse_const_rows_bwd(_p(x), float(c), _p(grad_rows), _p(dx), rows, x.numel() // rows, _stream())
    return dx


def softmax_ce_fwd(logits, labels, scale=1.0):
    logits = _chk(logits, "logits")
    labels = _chk(labels, "labels", torch.int64)
    B, K = logits.shape
    loss = torch.empty(B, dtype=torch.float32, device=logits.device)
    lse = torch.empty(B, dtype=torch.float32, device=logits.device)
    lib.rg_softmax_ce_fwd(_p(logits), _p(labels), _p(loss), _p(lse), B, K, scale, _stream())
    return loss, lse


def softmax_ce_bwd(logits, labels, lse, grad_rows, scale=1.0, grad_scale=1.0):
    logits = _chk(logits, "logits")
    grad_rows = _chk(grad_rows, "grad_rows")
    B, K = logits.shape
    dz = torch.empty_like(logits)
    lib.rg_softmax_ce_bwd(_p(logits), _p(labels), _p(lse), _p(grad_rows), _p(dz), B, K, scale, grad_scale, _stream())
    return dz


def weighted_sum_fwd(x, w=None, scale=1.0):
    x, w = _chk(x, "x"), _chk(w, "w")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    lib.rg_weighted_sum_fwd(_p(x), _p(w), _p(out), x.numel(), scale, _stream())
    return out


def weighted_sum_bwd(grad_out, w, n, scale, device):
    dx = torch.empty(n, dtype=torch.float32, device=device)
    lib.rg_weighted_sum_bwd(_p(grad_out), _p(w), _p(dx), n, scale, _stream())
    return dx


# ------------------------------------------------------------------------------------------------
# cluster memory
# ------------------------------------------------------------------------------------------------
def cm_update(inputs, targets, features, momentum, hard=False, normalize_eps=False):
    inputs = _chk(inputs, "inputs")
    targets = _chk(targets, "targets", torch.int64)
    if not (features.is_cuda and features.is_contiguous() and features.dtype == torch.float32):
        raise RuntimeError("cm_update: the memory bank must be a contiguous fp32 GPU tensor (updated in place)")
    B, D = inputs.shape
    K = features.shape[0]
    if hard:
        lib.rg_cm_update_hard(_p(inputs), _p(targets), _p(features), B, D, K, float(momentum), _stream())
    else:
        lib.rg_cm_update(_p(inputs), _p(targets), _p(features), B, D, K, float(momentum), int(normalize_eps), _stream())
    return features


# ------------------------------------------------------------------------------------------------
# optimizers
# ------------------------------------------------------------------------------------------------
def normalize_listed_rows(g, ids, eps=1e-16):
    """in place: g[id] /= |g[id]| + eps for the listed (int64, device) row ids"""
    g = _chk(g, "g")
    ids = _chk(ids, "ids", torch.int64)
    lib.rg_normalize_listed_rows(_p(g), _p(ids), ids.numel(), g.shape[0], g.shape[1], eps, _stream())
    return g


def topk_rows(s, k):
    """(idx int32 [rows, k], val [rows, k]) of the k largest entries per row, ties by lower index"""
    s = _chk(s, "s")
    rows, cols = s.shape
    idx = torch.empty((rows, k), dtype=torch.int32, device=s.device)
    val = torch.empty((rows, k), dtype=torch.float32, device=s.device)
    lib.rg_topk_rows(_p(s), rows, cols, k, _p(idx), _p(val), _stream())
    return idx, val


def row_sqsum(x):
    x = _chk(x, "x")
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    lib.rg_row_sqsum(_p(x), _p(out), x.shape[0], x.shape[1], _stream())
    return out


def add_outer_terms(m, rowv=None, colv=None, alpha=1.0, a=1.0, b=1.0):
    m = _chk(m, "m")
    lib.rg_add_outer_terms(_p(m), _p(rowv), _p(colv), alpha, a, b, m.shape[0], m.shape[1], _stream())
    return m


def segment_mean(x, order, offsets):
    x = _chk(x, "x")
    order, offsets = _chk(order, "order", torch.int64), _chk(offsets, "offsets", torch.int64)
    out = torch.empty((offsets.numel() - 1, x.shape[1]), dtype=torch.float32, device=x.device)
    lib.rg_segment_mean(_p(x), _p(order), _p(offsets), _p(out), offsets.numel() - 1, x.shape[1], _stream())
    return out


def adam_advance(state, beta1, beta2):
    lib.rg_adam_advance(_p(state), beta1, beta2, _stream())


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, weight_decay, state, grad_scale=1.0):
    lib.rg_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, _p(state), grad_scale,
                         _stream())


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    lib.rg_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                     _stream())


def sgd_step(p, g, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    lib.rg_sgd_step(_p(p), _p(g), _p(buf), p.numel(), lr, momentum, weight_decay, int(first_step), grad_scale, _stream())


# ------------------------------------------------------------------------------------------------
# profiler
# ------------------------------------------------------------------------------------------------
_PROFILING = [False]


def profile_enable(on=True):
    _PROFILING[0] = bool(on)
    lib.rg_profile_enable(int(on))


def check_not_profiling():
    """the per-launch event profiler records timing events on the launch stream: not capturable / meaningless under replay"""
    if _PROFILING[0]:
        raise RuntimeError("rg_hip: the launch profiler is on; switch it off (ops.profile_enable(False)) before capturing or "
                           "replaying a step graph")


def profile_reset():
    lib.rg_profile_reset()


def profile_collect():
    import ctypes
    from .lib import FAMILIES
    n = lib.rg_family_count()
    ms = (ctypes.c_double * n)()
    fl = (ctypes.c_double * n)()
    by = (ctypes.c_double * n)()
    calls = (ctypes.c_longlong * n)()
    lib.rg_profile_collect(ctypes.addressof(ms), ctypes.addressof(fl), ctypes.addressof(by), ctypes.addressof(calls))
    return {FAMILIES[i]: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "calls": calls[i]} for i in range(n)}


# ---- on-device input synthesis --------------------------------------------------------------
def pose_maps(centers, sigma, height, width, mode=0):
    """[N, J, H, W] heat maps from int32 joint centres [N, J, 2] = (row, col) (negative: missing joint) and one sigma
    per sample.  mode 0: FD-GAN's filtered impulse / max (preprocessor.py:114-131); mode 1: plain Gaussian
    (pose_utils.py:51-70)."""
    if centers.dtype != torch.int32 or centers.dim() != 3 or centers.shape[2] != 2 or not centers.is_cuda:
        raise ValueError("pose_maps: centers must be an int32 device tensor [N, J, 2]")
    centers = centers.contiguous()
    sigma = _chk(sigma, "sigma")
    N, J = centers.shape[0], centers.shape[1]
    if sigma.numel() != N:
        raise ValueError("pose_maps: one sigma per sample expected")
    out = torch.empty((N, J, height, width), dtype=torch.float32, device=centers.device)
    lib.rg_pose_maps(_p(centers), _p(sigma), _p(out), N, J, height, width, mode, _stream())
    return out


def flip_pad_crop(x, params, out_hw, pad=0, pad_value=None):
    """out[n, c, y, x] = padded(x)[n, c, top + y, left + (flip ? W-1-x : x)], params int32 [N, 3] = (flip, top, left)."""
    x = _chk(x, "x")
    if params.dtype != torch.int32 or tuple(params.shape) != (x.shape[0], 3) or not params.is_cuda:
        raise ValueError("flip_pad_crop: params must be an int32 device tensor [N, 3]")
    N, C, Hs, Ws = x.shape
    H, W = out_hw
    pad_value = _chk(pad_value, "pad_value")
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    lib.rg_flip_pad_crop(_p(x), _p(params.contiguous()), _p(pad_value), _p(out), N, C, Hs, Ws, H, W, pad, _stream())
    return out


def erase_rects_(x, rects, fill):
    """in place: x[n, c, r0:r0+h, c0:c0+w] = fill[c] for rects int32 [N, 4] = (r0, c0, h, w); h = 0: untouched."""
    x = _chk(x, "x")
    if rects.dtype != torch.int32 or tuple(rects.shape) != (x.shape[0], 4) or not rects.is_cuda:
        raise ValueError("erase_rects_: rects must be an int32 device tensor [N, 4]")
    fill = _chk(fill, "fill")
    if fill.numel() != x.shape[1]:
        raise ValueError("erase_rects_: one fill value per channel expected")
    N, C, H, W = x.shape
    lib.rg_erase_rects(_p(x), _p(rects.contiguous()), _p(fill), N, C, H, W, _stream())
    return x
