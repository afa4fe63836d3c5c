"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under reid-gan_amd/ imports it.

CPU emulation of the fp8 convolution family of csrc/conv_f8.hip (BASELINE config 5, "fp8 MFMA convs").  The reference has
no reduced-precision path (its dual_gan layers are fp32 cuDNN convolutions, CC/dual_gan/models/base_function.py:236-443), so
this file is the SPECIFICATION of the build's own fp8 arithmetic rather than a restatement:

  scale      q = fmax / amax (fp32 division), dequantisation scale d = amax / fmax (fp32), fmax = 448 (e4m3fn) / 57344 (e5m2)
  quantise   clamp(x * q, -fmax, fmax) in fp32, round-to-nearest-even to OCP e4m3fn / e5m2 (torch.float8_* casts)
  layouts    'nhwc' [N][HW][Cp], 'chwn' [C][HW][Np] with Cp / Np = the count padded to a multiple of 16 with zero bytes
  GEMMs      exact products of the decoded fp8 values, accumulated (here in fp64; on the GPU in fp32 inside the MFMA),
             times d_a * d_b (fp32 product), then the fp32 epilogue (+ bias, + residual, activation)

Parity of the HIP kernels against this emulation is checked at fp32-accumulation tolerance (the operands are bit-identical);
the DECLARED tolerance of the fp8 family against the fp32 reference arithmetic lives in tests/test_f8_gpu.py.
"""
from __future__ import absolute_import

import torch
import torch.nn.functional as F

FMAX = (448.0, 57344.0)
DTYPES = (torch.float8_e4m3fn, torch.float8_e5m2)


def pad16(v):
    return (int(v) + 15) // 16 * 16


def scales(amax, fmt):
    """(q, d) as fp32 tensors: quantisation multiplier and dequantisation scale for a tensor whose max|x| is `amax`"""
    amax = torch.as_tensor(amax, dtype=torch.float32)
    fmax = torch.tensor(FMAX[fmt], dtype=torch.float32)
    if float(amax) <= 0.0:
        return torch.tensor(1.0), torch.tensor(1.0)
    return fmax / amax, amax / fmax


def quantize(x, amax, fmt):
    """fp32 tensor -> (float8 tensor of the same shape, d)"""
    q, d = scales(amax, fmt)
    v = torch.clamp(x.float() * q, -FMAX[fmt], FMAX[fmt])
    return v.to(DTYPES[fmt]), d


def to_layout(xq, layout):
    """float8 [N, C, ...] -> uint8 bytes in the kernel's operand layout (zero padding)"""
    N, C = xq.shape[0], xq.shape[1]
    b = xq.reshape(N, C, -1).view(torch.uint8)
    L = b.shape[2]
    if layout in ("nhwc", "krsc"):
        out = torch.zeros(N, L, pad16(C), dtype=torch.uint8)
        out[:, :, :C] = b.permute(0, 2, 1)
    elif layout in ("chwn", "crsk"):
        out = torch.zeros(C, L, pad16(N), dtype=torch.uint8)
        out[:, :, :N] = b.permute(1, 2, 0)
    else:
        raise ValueError(layout)
    return out


def conv_fwd(x, w, fmt_x=0, bias=None, residual=None, stride=1, padding=0, act=None, slope=0.0, amax_x=None, amax_w=None):
    """emulated rg_conv2d_f8_fwd: operands quantised with their own amax (just-in-time scaling) unless given"""
    xq, dx = quantize(x, x.abs().max() if amax_x is None else amax_x, fmt_x)
    wq, dw = quantize(w, w.abs().max() if amax_w is None else amax_w, 0)
    y = F.conv2d(xq.double(), wq.double(), None, stride, padding) * float(dx * dw)
    return _epilogue(y, bias, residual, act, slope)


def conv_dgrad(dy, w, x_hw, fmt_dy=1, bias=None, residual=None, stride=1, padding=0, act=None, slope=0.0):
    """emulated rg_conv2d_f8_dgrad (also ConvTranspose2d forward with fmt_dy = 0): w is [K][C][KH][KW]"""
    dyq, ddy = quantize(dy, dy.abs().max(), fmt_dy)
    wq, dw = quantize(w, w.abs().max(), 0)
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    KH, KW = w.shape[2], w.shape[3]
    oph = x_hw[0] - ((dy.shape[2] - 1) * sh - 2 * ph + KH)
    opw = x_hw[1] - ((dy.shape[3] - 1) * sw - 2 * pw + KW)
    dx = F.conv_transpose2d(dyq.double(), wq.double(), None, (sh, sw), (ph, pw), (oph, opw)) * float(ddy * dw)
    return _epilogue(dx, bias, residual, act, slope)


def conv_wgrad(x, dy, w_shape, fmt_x=0, fmt_dy=1, stride=1, padding=0):
    """emulated rg_conv2d_f8_wgrad -> dw [K][C][KH][KW]"""
    xq, dx = quantize(x, x.abs().max(), fmt_x)
    dyq, ddy = quantize(dy, dy.abs().max(), fmt_dy)
    xd = xq.double().requires_grad_(False)
    w0 = torch.zeros(w_shape, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xd, w0, None, stride, padding)
    (g,) = torch.autograd.grad(y, w0, dyq.double())
    return (g * float(dx * ddy)).float()


def _epilogue(y, bias, residual, act, slope):
    y = y.float()
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    if act == "relu":
        y = F.relu(y)
    elif act == "leaky":
        y = F.leaky_relu(y, slope)
    elif act == "tanh":
        y = torch.tanh(y)
    return y
