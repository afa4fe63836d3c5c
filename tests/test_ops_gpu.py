"""Op-level parity of every HIP kernel family against plain PyTorch CPU ops (fp64 reference).

These run on the MI355X only (-m gpu) and call through the C ABI (rg_hip.ops -> libreidgan_hip.so).
Tolerance: fp32 accumulation over K terms; we require max|err| <= 2e-5 * (sum|a||b| scale) which is
far inside the 1e-3 relative budget of the north star.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from rg_hip import ops
    return ops


def _close(got, ref, tol=2e-5, name=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (rel %.3e)" % (name, err, scale, err / scale)


# (N, C, H, W, K, KH, KW, stride, pad)   — every geometry of the reference networks at reduced batch/width
CONV_CASES = [
    (2, 3, 64, 32, 64, 7, 7, 2, 3),      # ResNet stem (Kg = 147, scalar weight path)
    (2, 64, 16, 8, 64, 1, 1, 1, 0),      # bottleneck 1x1
    (2, 64, 16, 8, 256, 1, 1, 1, 0),
    (3, 32, 16, 8, 32, 3, 3, 1, 1),      # bottleneck 3x3
    (2, 128, 16, 8, 128, 3, 3, 2, 1),    # stride-2 3x3
    (2, 256, 16, 8, 512, 1, 1, 2, 0),    # downsample 1x1/2
    (2, 18, 64, 32, 64, 4, 4, 2, 1),     # G en_conv1
    (2, 64, 32, 16, 128, 4, 4, 2, 1),    # G/D 4x4/2
    (2, 21, 64, 32, 64, 4, 4, 2, 1),     # D_pd first conv
    (2, 64, 9, 7, 96, 4, 4, 1, 1),       # D_pd stride-1 4x4 (odd sizes)
    (2, 96, 8, 6, 1, 4, 4, 1, 1),        # D_pd 1-channel head (M = 1)
    (4, 512, 8, 4, 128, (8, 4), None, 1, 0),   # G en_avg (8,4) valid
    (1, 5, 7, 5, 3, 3, 3, 1, 1),         # tiny, ragged everything
    (5, 2048, 1, 1, 2, 1, 1, 1, 0),      # Linear 2048 -> 2 as 1x1
    (64, 512, 1, 1, 640, 1, 1, 1, 0),    # CM-style GEMM
    (3, 16, 17, 9, 24, 3, 3, 2, 1),      # odd spatial, stride 2
    (8, 512, 8, 4, 512, 3, 3, 1, 1),     # layer4 3x3 on an 8x4 map: split-K, KRSC dgrad weights
    (8, 2048, 8, 4, 512, 1, 1, 1, 0),    # layer4 1x1: float4 pixel loads + split-K
    (8, 512, 8, 4, 2048, 1, 1, 1, 0),
    (4, 512, 16, 8, 512, 4, 4, 2, 1),    # G en_conv5 (strided dgrad, 4 parity classes, KRSC weights)
    (32, 512, 31, 15, 1, 4, 4, 1, 1),    # D_pd head at full batch: M = 1, deep split-K
    (2, 256, 16, 8, 1024, 1, 1, 2, 0),   # downsample 1x1 stride 2 (empty parity classes in dgrad)
    (2, 20, 12, 6, 36, 3, 3, 1, 1),      # C % 4 == 0 but tiny: KRSC path with ragged tiles
    (2, 3, 19, 13, 8, 7, 7, 2, 3),       # RGB data gradient, odd sizes: ragged pixel groups of the small-C kernel
    (1, 3, 9, 7, 5, 3, 3, 1, 1),         # small-C, stride 1 (one class, 3 taps per axis)
    (2, 2, 10, 6, 6, 4, 4, 2, 1),        # small-C with 2 channels
    (1, 3, 12, 9, 4, 5, 5, 1, 2),        # small-C, 5 taps per axis: the one-pixel-per-thread kernel
    (2, 4, 10, 6, 6, 3, 3, 2, 1),        # C = 4: one-pixel-per-thread kernel
    (3, 40, 9, 7, 1, 3, 3, 2, 1),        # one output channel, 3x3 / 2 (direct kernels, channel / pixel slices)
    (5, 130, 20, 12, 1, 4, 4, 1, 1),     # one output channel, ragged channel slices
    (3, 64, 18, 10, 3, 3, 3, 1, 0),      # dual_gan Output conv: 64 -> 3 on the reflection-padded map (direct kernels, 3 channels)
    (2, 37, 11, 9, 2, 3, 3, 2, 1),       # 2 output channels, stride 2, ragged slices
    (2, 130, 12, 8, 4, 3, 3, 1, 1),      # 4 output channels
    (2, 33, 16, 12, 2, 3, 3, 1, 1),      # four-pixel thin kernels (Q % 4 == 0): 2 output channels, pad 1, ragged channel slices
    (1, 64, 34, 18, 3, 3, 3, 1, 0),      # Output conv on a 32 x 16 map (valid convolution of the padded 34 x 18 input)
    (3, 8, 6, 8, 1, 3, 3, 1, 1),         # one output channel, 8 channels, fewer pixel quads than one workgroup
    (5, 20, 10, 20, 3, 3, 3, 1, 1),      # pixel count not a multiple of the wgrad slice
    # 3x3 / 1 / 1 with channels % 16 == 0 and power-of-two widths: the tap-reuse kernels (one halo tile for all nine taps)
    (2, 64, 64, 32, 64, 3, 3, 1, 1),     # layer1-like: 4 rows of 32 per tile, 64-row tile
    (2, 128, 32, 16, 128, 3, 3, 1, 1),   # layer2-like: 8 rows of 16
    (1, 64, 8, 64, 96, 3, 3, 1, 1),      # width 64: 2 rows per tile, large halo (264 positions), ragged rows (96 of 128)
    (3, 64, 8, 8, 80, 3, 3, 1, 1),       # 2 images per tile, 192 pixels: a half-empty second tile; dgrad with a 64-row tile
    (5, 32, 4, 4, 160, 3, 3, 1, 1),      # 8 images per tile (288 halo positions), 5 images only; two row tiles (fwd only: C < 64)
    (3, 32, 16, 8, 48, 3, 3, 1, 1),      # fewer than 64 output channels: stays on the generic kernels
]


def _geom(case):
    N, C, H, W, K, KH, KW, s, p = case
    if isinstance(KH, tuple):
        KH, KW = KH
    return N, C, H, W, K, KH, KW, s, p


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(dev, case):
    ops = _ops()
    N, C, H, W, K, KH, KW, s, p = _geom(case)
    g = torch.Generator().manual_seed(1234 + N * 7 + C)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, KH, KW, generator=g) / math.sqrt(C * KH * KW)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())

    y = ops.conv2d_fwd(x.to(dev), w.to(dev), s, p)
    _close(y, y_ref, name="fwd")
    dx = ops.conv2d_dgrad(dy.to(dev), w.to(dev), (H, W), s, p)
    _close(dx, xd.grad, name="dgrad")
    dw = ops.conv2d_wgrad(x.to(dev), dy.to(dev), (K, C, KH, KW), s, p)
    _close(dw, wd.grad, tol=5e-5, name="wgrad")


def test_conv_fused_epilogue(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 40, 12, 10, generator=g)
    w = torch.randn(72, 40, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(72, generator=g) + 0.5, torch.randn(72, generator=g)
    res = torch.randn(3, 72, 12, 10, generator=g)
    ref = F.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double()
    for act, fn in ((ops.ACT_RELU, F.relu), (ops.ACT_LEAKY, lambda t: F.leaky_relu(t, 0.2)), (ops.ACT_TANH, torch.tanh)):
        y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 1, scale=sc.to(dev), shift=sh.to(dev), residual=res.to(dev), act=act,
                           slope=0.2)
        _close(y, fn(ref), name="epilogue act=%d" % act)


def test_tap_reuse_conv_fused_epilogue(dev):
    """3x3 / 1 / 1 with 32 input channels takes the tap-reuse kernel: same fused epilogue (scale, shift, residual, activation),
    with and without split-K (9 images -> 9 tiles -> split reduction; 40 images -> no split)"""
    ops = _ops()
    g = torch.Generator().manual_seed(17)
    for n in (9, 40):
        x = torch.randn(n, 32, 16, 8, generator=g)
        w = torch.randn(72, 32, 3, 3, generator=g) * 0.05
        sc, sh = torch.rand(72, generator=g) + 0.5, torch.randn(72, generator=g)
        res = torch.randn(n, 72, 16, 8, generator=g)
        ref = F.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double()
        for act, fn in ((ops.ACT_NONE, lambda t: t), (ops.ACT_RELU, F.relu), (ops.ACT_LEAKY, lambda t: F.leaky_relu(t, 0.2))):
            y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 1, scale=sc.to(dev), shift=sh.to(dev), residual=res.to(dev), act=act, slope=0.2)
            _close(y, fn(ref), name="tap-reuse epilogue n=%d act=%d" % (n, act))


def test_thin_conv_fused_epilogue(dev):
    """<= 4 output channels: the direct kernels' slice partials go through the split-K finishing kernel, which owns the epilogue"""
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 64, 14, 10, generator=g)
    w = torch.randn(3, 64, 3, 3, generator=g) * 0.05
    b = torch.randn(3, generator=g)
    ref = torch.tanh(F.conv2d(x.double(), w.double(), b.double()))
    y = ops.conv2d_fwd(x.to(dev), w.to(dev), 1, 0, shift=b.to(dev), act=ops.ACT_TANH)
    _close(y, ref, name="thin conv + bias + tanh")


@pytest.mark.parametrize("case", [
    (2, 2432, 1, 1, 64, (8, 4), 1, 0, 0),    # G de_avg ConvTranspose (8,4) on a 1x1 map
    (2, 64, 8, 4, 64, (4, 4), 2, 1, 0),      # G decoder 4x4/2
    (3, 32, 16, 8, 3, (4, 4), 2, 1, 0),      # final 64->3 (tiny M)
    (2, 24, 8, 4, 16, (3, 3), 2, 1, 1),      # dual_gan 3x3/2 with output_padding 1
])
def test_conv_transpose(dev, case):
    ops = _ops()
    N, Cin, H, W, Cout, (KH, KW), s, p, op = case
    g = torch.Generator().manual_seed(99)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, KH, KW, generator=g) / math.sqrt(Cin)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = F.conv_transpose2d(xd, wd, stride=s, padding=p, output_padding=op)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    Ho, Wo = y_ref.shape[2:]
    y = ops.conv2d_dgrad(x.to(dev), w.to(dev), (Ho, Wo), s, p)          # convT forward == dgrad
    _close(y, y_ref, name="convT fwd")
    dx = ops.conv2d_fwd(dy.to(dev), w.to(dev), s, p)                    # convT dgrad == conv fwd
    _close(dx, xd.grad, name="convT dgrad")
    dw = ops.conv2d_wgrad(dy.to(dev), x.to(dev), (Cin, Cout, KH, KW), s, p)   # roles of x / dy swapped
    _close(dw, wd.grad, tol=5e-5, name="convT wgrad")


@pytest.mark.parametrize("shape", [(4, 64, 16, 8), (3, 10, 7, 5), (16, 2048), (2, 512, 31, 15)])
@pytest.mark.parametrize("train", [True, False])
def test_batchnorm(dev, shape, train):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    res = torch.randn(shape, generator=g)
    xd = x.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    resd = res.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    y_ref = F.relu(F.batch_norm(xd, rm_ref, rv_ref, gd, bd, training=train, momentum=0.1, eps=1e-5) + resd)
    dy = torch.randn(shape, generator=g)
    y_ref.backward(dy.double())

    xg, rmg, rvg = x.to(dev), rm.to(dev), rv.to(dev)
    if train:
        mean, stat = ops.bn_stats(xg, rmg, rvg, 1e-5, 0.1)
        is_var = False
        _close(rmg, rm_ref, name="running_mean")
        _close(rvg, rv_ref, name="running_var")
    else:
        mean, stat, is_var = rmg, rvg, True
    y = ops.bn_apply_fwd(xg, mean, stat, gamma.to(dev), beta.to(dev), res.to(dev), is_var, 1e-5, ops.ACT_RELU)
    _close(y, y_ref, name="bn fwd")
    s1, s2 = ops.bn_bwd_reduce(xg, dy.to(dev), y, mean, stat, is_var, 1e-5, ops.ACT_RELU)
    _close(s1, bd.grad, tol=5e-5, name="dbeta")
    _close(s2, gd.grad, tol=5e-5, name="dgamma")
    dx, dres = ops.bn_bwd_apply(xg, dy.to(dev), y, mean, stat, gamma.to(dev), s1, s2, train, is_var, 1e-5,
                                ops.ACT_RELU, need_dx=True, need_dres=True)
    _close(dx, xd.grad, tol=5e-5, name="bn dx")
    _close(dres, resd.grad, name="bn dres")
    if not train:       # the fused one-pass eval backward must give the same four results
        fdx, fdres, f1, f2 = ops.bn_eval_bwd(xg, dy.to(dev), y, rmg, rvg, gamma.to(dev), 1e-5, ops.ACT_RELU,
                                              need_dx=True, need_dres=True, need_sums=True)
        _close(fdx, xd.grad, tol=5e-5, name="fused eval dx")
        _close(fdres, resd.grad, name="fused eval dres")
        _close(f1, bd.grad, tol=5e-5, name="fused eval dbeta")
        _close(f2, gd.grad, tol=5e-5, name="fused eval dgamma")


def test_activations_and_misc(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 7, 5, 3, generator=g)
    dy = torch.randn(3, 7, 5, 3, generator=g)
    for act, fn in ((ops.ACT_RELU, F.relu), (ops.ACT_LEAKY, lambda t: F.leaky_relu(t, 0.2)), (ops.ACT_TANH, torch.tanh)):
        xd = x.double().requires_grad_(True)
        yr = fn(xd)
        yr.backward(dy.double())
        y = ops.act_fwd(x.to(dev), act, 0.2)
        _close(y, yr, name="act fwd")
        _close(ops.act_bwd(dy.to(dev), y, act, 0.2), xd.grad, name="act bwd")
    a, b = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    _close(ops.axpby(a.to(dev), b.to(dev), 0.5, -2.0), 0.5 * a.double() - 2.0 * b.double(), name="axpby")
    ad, bd_ = a.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = (ad - bd_).pow(2)
    d = torch.randn(1000, generator=g)
    yr.backward(d.double())
    _close(ops.sub_square_fwd(a.to(dev), b.to(dev)), yr, name="subsq")
    da, db = ops.sub_square_bwd(a.to(dev), b.to(dev), d.to(dev))
    _close(da, ad.grad, name="subsq da")
    _close(db, bd_.grad, name="subsq db")
    # l2 normalize rows
    m = torch.randn(9, 2048, generator=g)
    md = m.double().requires_grad_(True)
    yr = F.normalize(md, dim=1)
    gy = torch.randn(9, 2048, generator=g)
    yr.backward(gy.double())
    y, nrm = ops.l2norm_rows_fwd(m.to(dev))
    _close(y, yr, name="l2norm")
    _close(ops.l2norm_rows_bwd(y, gy.to(dev), nrm), md.grad, name="l2norm bwd")
    # channel cat / slice
    t1, t2 = torch.randn(2, 3, 4, 5, generator=g), torch.randn(2, 18, 4, 5, generator=g)
    cat = ops.cat_channels([t1.to(dev), t2.to(dev)])
    _close(cat, torch.cat([t1, t2], 1), tol=0, name="cat")
    _close(ops.slice_channels(cat, 3, 21), t2, tol=0, name="slice")
    # dropout: scaling and keep-rate
    xx = torch.ones(1 << 16)
    yy = ops.dropout(xx.to(dev), 0.2, 12345).cpu()
    keep = (yy != 0).float().mean().item()
    assert abs(keep - 0.8) < 0.01
    assert torch.allclose(yy[yy != 0], torch.tensor(1.25))
    assert torch.equal(yy, ops.dropout(xx.to(dev), 0.2, 12345).cpu())


def test_pooling(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 17, 9, generator=g).relu()      # zeros -> ties, like post-ReLU stem output
    xd = x.double().requires_grad_(True)
    yr = F.max_pool2d(xd, 3, 2, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    y, arg = ops.maxpool2d_fwd(x.to(dev), 3, 2, 1)
    _close(y, yr, tol=0, name="maxpool")
    _close(ops.maxpool2d_bwd(dy.to(dev), arg, x.shape, 3, 2, 1), xd.grad, name="maxpool bwd")
    # W % 4 == 0: the 4-pixels-per-thread kernel of the 3x3 / 2 stem pool (odd and even heights, ties from the ReLU)
    for shp in ((2, 5, 18, 12), (1, 3, 7, 8), (3, 2, 32, 16)):
        x4 = torch.randn(*shp, generator=g).relu()
        x4d = x4.double().requires_grad_(True)
        y4r = F.max_pool2d(x4d, 3, 2, 1)
        dy4 = torch.randn(y4r.shape, generator=g)
        y4r.backward(dy4.double())
        y4, arg4 = ops.maxpool2d_fwd(x4.to(dev), 3, 2, 1)
        _close(y4, y4r, tol=0, name="maxpool %s" % (shp,))
        _close(ops.maxpool2d_bwd(dy4.to(dev), arg4, x4.shape, 3, 2, 1), x4d.grad, tol=1e-6, name="maxpool bwd %s" % (shp,))
    # global average
    x = torch.randn(3, 40, 8, 4, generator=g)
    xd = x.double().requires_grad_(True)
    yr = F.avg_pool2d(xd, xd.shape[2:]).flatten(1)
    dy = torch.randn(3, 40, generator=g)
    yr.backward(dy.double())
    _close(ops.global_avgpool_fwd(x.to(dev)), yr, name="gap")
    _close(ops.global_avgpool_bwd(dy.to(dev), x.shape), xd.grad, name="gap bwd")
    # GeM
    x = torch.randn(3, 40, 16, 8, generator=g)
    p = torch.tensor([3.0])
    xd, pd_ = x.double().requires_grad_(True), p.double().requires_grad_(True)
    yr = F.adaptive_avg_pool2d(xd.clamp(min=1e-6).pow(pd_), 1).pow(1.0 / pd_).flatten(1)
    yr.backward(dy.double())
    y = ops.gem_pool_fwd(x.to(dev), p.to(dev))
    _close(y, yr, tol=1e-4, name="gem")
    dx, dp = ops.gem_pool_bwd(x.to(dev), p.to(dev), y, dy.to(dev))
    _close(dx, xd.grad, tol=1e-4, name="gem dx")
    _close(dp, pd_.grad, tol=1e-3, name="gem dp")


def test_losses(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 1, 30, 14, generator=g) * 3
    for target in (1.0, 0.0, 0.83):
        xd = x.double().requires_grad_(True)
        lr = F.binary_cross_entropy(torch.sigmoid(xd), torch.full_like(xd, target))
        (lr * 0.5).backward()
        _close(ops.sigmoid_bce_fwd(x.to(dev), target), lr, name="bce")
        gout = torch.tensor(0.5, device=dev)
        _close(ops.sigmoid_bce_bwd(x.to(dev), gout, target), xd.grad, name="bce bwd")
        xd = x.double().requires_grad_(True)
        lr = F.mse_loss(xd, torch.full_like(xd, target))
        lr.backward()
        _close(ops.mse_const_fwd(x.to(dev), target), lr, name="mse")
        _close(ops.mse_const_bwd(x.to(dev), None, target), xd.grad, name="mse bwd")
    a, b = torch.randn(6, 3, 16, 8, generator=g), torch.randn(6, 3, 16, 8, generator=g)
    ad, bd_ = a.double().requires_grad_(True), b.double().requires_grad_(True)
    lr = F.l1_loss(ad, bd_)
    lr.backward()
    out2 = ops.l1_fwd(a.to(dev), b.to(dev))
    _close(out2[0], lr, name="l1")
    da, db = ops.l1_bwd(a.to(dev), b.to(dev), None, None, out2)
    _close(da, ad.grad, name="l1 da")
    _close(db, bd_.grad, name="l1 db")
    labels = torch.tensor([1, 0, 0, 1, 0, 1])
    ad, bd_ = a.double().requires_grad_(True), b.double().requires_grad_(True)
    mask = labels.view(-1, 1, 1, 1).expand_as(ad) == 1
    lr = F.l1_loss(ad[mask], bd_[mask])
    (lr * 10).backward()
    out2 = ops.l1_fwd(a.to(dev), b.to(dev), labels.to(dev))
    _close(out2[0], lr, name="masked l1")
    da, db = ops.l1_bwd(a.to(dev), b.to(dev), labels.to(dev), torch.tensor(10.0, device=dev), out2)
    _close(da, ad.grad, name="masked l1 da")
    # nothing selected -> NaN like torch
    assert torch.isnan(ops.l1_fwd(a.to(dev), b.to(dev), torch.zeros(6, dtype=torch.long, device=dev))[0]).item()
    # softmax CE with temperature
    z = torch.randn(8, 2048, generator=g)
    y = torch.randint(0, 2048, (8,), generator=g)
    zd = z.double().requires_grad_(True)
    lrows = F.cross_entropy(zd / 0.05, y, reduction="none")
    wts = torch.rand(8, generator=g)
    (lrows * wts.double()).sum().backward()
    loss, lse = ops.softmax_ce_fwd(z.to(dev), y.to(dev), 1 / 0.05)
    _close(loss, lrows, name="ce")
    _close(ops.softmax_ce_bwd(z.to(dev), y.to(dev), lse, wts.to(dev), 1 / 0.05), zd.grad, name="ce bwd")
    _close(ops.weighted_sum_fwd(loss, wts.to(dev), 0.125), (lrows * wts.double()).sum() * 0.125, name="wsum")


def test_cm_update(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    K, D, B = 50, 2048, 24
    feats = F.normalize(torch.randn(K, D, generator=g), dim=1)
    x = F.normalize(torch.randn(B, D, generator=g), dim=1)
    y = torch.tensor([3, 3, 7, 3, 9, 7, 7, 3] * 3)
    ref = feats.double().clone()
    m = 0.2
    for xi, yi in zip(x.double(), y):
        ref[yi] = m * ref[yi] + (1 - m) * xi
        ref[yi] /= ref[yi].norm()
    fg = feats.to(dev).clone()
    ops.cm_update(x.to(dev), y.to(dev), fg, m)
    _close(fg, ref, name="cm update")
    # hard variant: per label, the sample with the smallest similarity
    ref = feats.double().clone()
    for lab in y.unique().tolist():
        idx = (y == lab).nonzero().flatten()
        sims = x.double()[idx] @ ref[lab]
        j = idx[sims.argmin()]
        ref[lab] = ref[lab] * m + (1 - m) * x.double()[j]
        ref[lab] /= ref[lab].norm()
    fg = feats.to(dev).clone()
    ops.cm_update(x.to(dev), y.to(dev), fg, m, hard=True)
    _close(fg, ref, name="cm hard update")


def test_optimizers(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    n = 1003
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) for _ in range(3)]
    pr = p0.double().clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=2e-3, betas=(0.5, 0.999), weight_decay=5e-4)
    pg = torch.zeros(1008, device=dev)[:n]
    pg.copy_(p0)
    mg, vg = torch.zeros(1008, device=dev)[:n], torch.zeros(1008, device=dev)[:n]
    for t, gr in enumerate(grads, 1):
        pr.grad = gr.double()
        opt.step()
        gg = torch.zeros(1008, device=dev)[:n]
        gg.copy_(gr)
        ops.adam_step(pg, gg, mg, vg, 2e-3, 0.5, 0.999, 1e-8, 5e-4, t)
    _close(pg, pr, name="adam")
    pr = p0.double().clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pg = p0.to(dev).clone()
    buf = torch.zeros(n, device=dev)
    for t, gr in enumerate(grads):
        pr.grad = gr.double()
        opt.step()
        ops.sgd_step(pg, gr.to(dev), buf, 0.01, 0.9, 1e-4, t == 0)
    _close(pg, pr, name="sgd")


@pytest.mark.parametrize("case", [
    (2, 3, 128, 64, 256, 128, True),     # my_transform of the joint step
    (2, 3, 128, 64, 256, 128, False),    # my_resize
    (1, 3, 16, 8, 16, 8, True),          # my_normalize only
    (2, 2, 9, 7, 20, 23, False),         # odd, non-integer scale
    (1, 3, 32, 16, 12, 10, True),        # down-sampling without antialias (same formula)
])
def test_bicubic_normalize(dev, case):
    """diff_augs.my_resize / my_normalize / my_transform vs F.interpolate(bicubic, align_corners=False) + affine."""
    ops = _ops()
    N, C, H, W, OH, OW, norm = case
    g = torch.Generator().manual_seed(5)
    x = torch.rand(N, C, H, W, generator=g)
    dy = torch.randn(N, C, OH, OW, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406][:C])
    std = torch.tensor([0.229, 0.224, 0.225][:C])
    xr = x.double().requires_grad_(True)
    yr = F.interpolate(xr, size=(OH, OW), mode="bicubic", align_corners=False) if (OH, OW) != (H, W) else xr * 1.0
    if norm:
        yr = (yr - mean.double().view(1, C, 1, 1)) / std.double().view(1, C, 1, 1)
    yr.backward(dy.double())
    m, s = (mean.to(dev), std.to(dev)) if norm else (None, None)
    y = ops.bicubic_normalize_fwd(x.to(dev), (OH, OW), m, s)
    dx = ops.bicubic_normalize_bwd(dy.to(dev), (H, W), s)
    _close(y, yr, name="bicubic fwd")
    _close(dx, xr.grad, name="bicubic bwd")


def test_diff_augs_api(dev):
    from clustercontrast.utils.data.diff_augs import my_normalize, my_resize, my_transform
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 128, 64, generator=g)
    xd = x.to(dev).requires_grad_(True)
    out = my_transform(xd)
    assert out.shape == (2, 3, 256, 128)
    ref = my_normalize(my_resize(x.to(dev)))
    _close(out, ref, name="my_transform == my_normalize(my_resize)")
    out.sum().backward()
    xr = x.double().requires_grad_(True)
    std = torch.tensor([0.229, 0.224, 0.225]).double().view(1, 3, 1, 1)
    (F.interpolate(xr, size=(256, 128), mode="bicubic", align_corners=False) / std).sum().backward()
    _close(xd.grad, xr.grad, name="my_transform grad")
    assert my_resize(x[0].to(dev), (64, 32)).shape == (3, 64, 32)
    with pytest.raises(ValueError):
        my_normalize(torch.rand(1, 4, 8, 8).to(dev))


def test_avgpool_reflection_pad(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 5, 12, 10, generator=g)
    xr = x.double().requires_grad_(True)
    yr = F.avg_pool2d(xr, 2, 2)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    _close(ops.avgpool2d_fwd(x.to(dev), 2), yr, name="avgpool fwd")
    _close(ops.avgpool2d_bwd(dy.to(dev), x.shape, 2), xr.grad, name="avgpool bwd")
    x2 = torch.randn(2, 3, 7, 9, generator=g)          # odd sizes: the last row / column is dropped
    _close(ops.avgpool2d_fwd(x2.to(dev), 2), F.avg_pool2d(x2.double(), 2, 2), name="avgpool odd")
    for pad in (1, 2):
        xr = x.double().requires_grad_(True)
        yr = F.pad(xr, (pad,) * 4, mode="reflect")
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy.double())
        _close(ops.reflection_pad2d_fwd(x.to(dev), pad), yr, name="reflect fwd")
        _close(ops.reflection_pad2d_bwd(dy.to(dev), pad), xr.grad, name="reflect bwd")
    x3 = torch.randn(2, 3, 37, 70, generator=g)         # rows wider than one 64-lane pass, row count not a multiple of 32
    xr = x3.double().requires_grad_(True)
    yr = F.pad(xr, (1,) * 4, mode="reflect")
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    _close(ops.reflection_pad2d_fwd(x3.to(dev), 1), yr, name="reflect fwd wide")
    _close(ops.reflection_pad2d_bwd(dy.to(dev), 1), xr.grad, name="reflect bwd wide")


@pytest.mark.parametrize("shape", [(64, 256, 128, 1), (5, 7, 3, 3), (32, 64, 32, 16), (3, 130, 1, 1), (64, 48, 32, 16), (32, 64, 64, 32)])
def test_channel_sum_one_launch_and_two_stage(dev, shape):
    """bias gradients: sum over n and the pixels; one launch up to 32768 values per channel, the slice reduction above that"""
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape))
    dy = torch.randn(shape, generator=g)
    ref = dy.double().sum((0, 2, 3))
    out = ops.channel_sum(dy.to(dev))
    scale = dy.double().abs().sum((0, 2, 3)).max().item()
    assert (out.double().cpu() - ref).abs().max().item() <= 2e-6 * scale
    buf = torch.full((shape[1],), 7.0).to(dev)
    assert ops.channel_sum(dy.to(dev), out=buf) is buf and torch.equal(buf, out)


@pytest.mark.parametrize("shape", [(4, 2048, 16, 8), (3, 37, 5, 7), (2, 8, 1, 1), (5, 130, 9, 9)])
def test_l2norm_over_channels_of_a_map(dev, shape):
    """F.normalize(x, dim=1) of [N, C, H, W] and its backward on the NCHW map itself (resnet.py:100-107 `gan_x`)"""
    ops = _ops()
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g)
    x[0, :, 0, 0] = 0.0                                 # a zero vector: norm clamps to eps, output 0, gradient dy / eps
    xr = x.double().requires_grad_(True)
    yr = F.normalize(xr, dim=1)
    dy = torch.randn(shape, generator=g)
    dy[0, :, 0, 0] = 0.0
    yr.backward(dy.double())
    y, nrm = ops.l2norm_channels_fwd(x.to(dev))
    _close(y, yr, name="l2norm channels fwd")
    _close(nrm.view(shape[0], shape[2], shape[3]), x.double().norm(dim=1), name="norms")
    dx = ops.l2norm_channels_bwd(y, dy.to(dev), nrm)
    _close(dx, xr.grad, tol=5e-5, name="l2norm channels bwd")


@pytest.mark.parametrize("shape,pad", [((2, 3, 8, 8), 1), ((3, 5, 6, 12), 2), ((2, 4, 5, 16), 3), ((4, 64, 32, 16), 1), ((1, 2, 4, 8), 3)])
@pytest.mark.parametrize("act", ["none", "relu", "leaky"])
def test_reflection_pad_float4_rows_with_the_activation_folded_in(dev, shape, pad, act):
    """pad(act(x)) and act'(x) * pad^T(dy) in one pass each (the Output block's nonlinearity -> ReflectionPad2d, base_function.py:
    436-441) against torch: values are copies / exact products, so equality is exact for the forward; the backward sums up to nine
    gradient entries per element in the scalar kernel's order."""
    from rg_hip import ops
    g = torch.Generator().manual_seed(sum(shape) + pad)
    x = torch.randn(shape, generator=g)
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "leaky": ops.ACT_LEAKY}[act]
    slope = 0.2
    xr = x.clone().requires_grad_(True)
    a = xr if act == "none" else (torch.relu(xr) if act == "relu" else F.leaky_relu(xr, slope))
    yr = F.pad(a, (pad,) * 4, mode="reflect")
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    assert ops.reflection_pad_fusable(x, pad, code, slope) == (act != "none")
    y = ops.reflection_pad2d_fwd(x.to(dev), pad, code, slope)
    assert torch.equal(y.cpu(), yr.detach()), "forward differs"
    dx = ops.reflection_pad2d_bwd(dy.to(dev), pad, x.to(dev) if act != "none" else None, code, slope)
    _close(dx, xr.grad, tol=1e-6, name="reflect bwd (fused act)")
    # the scalar fallback of the narrow maps still refuses a fused activation loudly
    if act != "none":
        with pytest.raises(RuntimeError):
            ops.reflection_pad2d_fwd(torch.randn(1, 1, 4, 6).to(dev), 1, code, slope)


@pytest.mark.parametrize("shape", [(32, 3, 3, 3), (64, 32, 4, 4), (1, 128, 1, 1), (128, 128, 4, 4)])
def test_spectral_norm(dev, shape):
    """torch.nn.utils.spectral_norm (one power iteration per training forward) — weight, u, v and the gradient."""
    ops = _ops()
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(shape[1], shape[0], shape[2:], bias=False).double()
    ref = torch.nn.utils.spectral_norm(conv)
    w0 = ref.weight_orig.detach().clone()
    u = ref.weight_u.detach().clone().float().to(dev)
    v = ref.weight_v.detach().clone().float().to(dev)
    wd = w0.float().to(dev)
    for it in range(3):                                   # successive forwards keep iterating u, v
        ref.train()
        x = torch.randn(2, shape[1], 8, 8, dtype=torch.float64)
        y = ref(x)
        w_sn, sigma = ops.spectral_norm_fwd(wd, u, v, True)
        _close(w_sn, ref.weight, tol=1e-5, name="w_sn it%d" % it)
        _close(u, ref.weight_u, tol=1e-5, name="u")
        _close(v, ref.weight_v, tol=1e-5, name="v")
    gw = torch.randn(shape, dtype=torch.float64)
    ref.zero_grad()
    (ref.weight * gw).sum().backward()
    dw = ops.spectral_norm_bwd(gw.float().to(dev), w_sn, u, v, sigma)
    _close(dw, ref.weight_orig.grad, tol=1e-5, name="dw")
    ref.eval()                                            # eval: no iteration, same u / v
    ref(x)
    u2, v2 = u.clone(), v.clone()
    w_sn2, _, us, vs = ops.spectral_norm_fwd(wd, u2, v2, False, save_uv=True)
    _close(w_sn2, ref.weight, tol=1e-5, name="w_sn eval")
    assert torch.equal(u2, u) and torch.equal(v2, v)
    assert torch.equal(us, u) and torch.equal(vs, v)       # the copies the backward keeps
    # a training forward saves the u, v it produced; accumulate adds into an existing gradient
    u3, v3 = u.clone(), v.clone()
    w_sn3, sigma3, us3, vs3 = ops.spectral_norm_fwd(wd, u3, v3, True, save_uv=True)
    assert torch.equal(us3, u3) and torch.equal(vs3, v3)
    acc = dw.clone()
    ops.spectral_norm_bwd(gw.float().to(dev), w_sn, u, v, sigma, out=acc, accumulate=True)
    _close(acc, 2 * ref.weight_orig.grad, tol=1e-5, name="dw accumulate")


def test_bgemm_and_softmax(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, H, D, L, S = 3, 2, 40, 50, 70
    q = torch.randn(B, H * D, L, generator=g)
    k = torch.randn(B, H * D, S, generator=g)
    v = torch.randn(B, H * D, S, generator=g)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    # scores[b,h,l,s] = sum_d q[b,hD+d,l] k[b,hD+d,s]
    sc = torch.empty(B, H, L, S, device=dev)
    ops.bgemm(qd, kd, sc, L, S, D, (1, L), (S, 1), (S, 1), (B, H), (H * D * L, D * L), (H * D * S, D * S), (H * L * S, L * S))
    ref = torch.einsum("bhdl,bhds->bhls", q.double().view(B, H, D, L), k.double().view(B, H, D, S))
    _close(sc, ref, name="QtK")
    scale = 1.0 / math.sqrt(D)
    p = ops.softmax_rows_fwd(sc, scale)
    pref = torch.softmax(ref * scale, dim=-1)
    _close(p, pref, name="softmax")
    # out[b,hD+d,l] = sum_s v[b,hD+d,s] p[b,h,l,s]
    o = torch.empty(B, H * D, L, device=dev)
    ops.bgemm(vd, p, o, D, L, S, (S, 1), (1, S), (L, 1), (B, H), (H * D * S, D * S), (H * L * S, L * S), (H * D * L, D * L))
    oref = torch.einsum("bhds,bhls->bhdl", v.double().view(B, H, D, S), pref).reshape(B, H * D, L)
    _close(o, oref, name="PV")
    # alpha / beta accumulate
    o2 = o.clone()
    ops.bgemm(vd, p, o2, D, L, S, (S, 1), (1, S), (L, 1), (B, H), (H * D * S, D * S), (H * L * S, L * S), (H * D * L, D * L),
              alpha=0.5, beta=2.0)
    _close(o2, 2.5 * oref, name="alpha beta")
    dp = torch.randn(B, H, L, S, generator=g)
    ds = ops.softmax_rows_bwd(p, dp.to(dev), scale)
    xr = (ref * 1.0).requires_grad_(True)
    torch.softmax(xr * scale, dim=-1).backward(dp.double())
    _close(ds, xr.grad, name="softmax bwd")


@pytest.mark.parametrize("case", [(3, 24, 12, 10, 40, 3, 1, 1), (2, 16, 17, 9, 24, 3, 2, 1), (2, 32, 16, 8, 64, 1, 2, 0),
                                  (4, 64, 8, 4, 32, 4, 2, 1), (2, 3, 20, 12, 8, 7, 2, 3),
                                  (9, 64, 16, 8, 48, 3, 1, 1)])          # the tap-reuse data-gradient kernel
def test_dgrad_fused_epilogue(dev, case):
    """dgrad epilogue: dx = mask(act(dgrad * scale[c] + shift[c] + residual)) on the stride-1 path, the strided parity-class
    path, the 1x1/stride-2 path with empty classes and the small-C direct kernel."""
    ops = _ops()
    N, C, H, W, K, k, s, p = case
    g = torch.Generator().manual_seed(21 + C)
    w = torch.randn(K, C, k, k, generator=g) / math.sqrt(C * k * k)
    P, Q = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(N, K, P, Q, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    mask = torch.randn(N, C, H, W, generator=g)
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    base = torch.nn.grad.conv2d_input((N, C, H, W), w.double(), dy.double(), stride=s, padding=p)
    ref = base * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double()
    got = ops.conv2d_dgrad(dy.to(dev), w.to(dev), (H, W), s, p, scale=sc.to(dev), shift=sh.to(dev), residual=res.to(dev))
    _close(got, ref, name="dgrad affine+res")
    got = ops.conv2d_dgrad(dy.to(dev), w.to(dev), (H, W), s, p, residual=res.to(dev), relu_mask=mask.to(dev))
    _close(got, (base + res.double()) * (mask.double() > 0), name="dgrad res+mask")
    got = ops.conv2d_dgrad(dy.to(dev), w.to(dev), (H, W), s, p, shift=sh.to(dev), act=ops.ACT_LEAKY, slope=0.1,
                           relu_mask=mask.to(dev))
    _close(got, F.leaky_relu(base + sh.double().view(1, -1, 1, 1), 0.1) * (mask.double() > 0), name="dgrad act+mask")


def test_bn_fold_ops(dev):
    """conv + frozen BatchNorm fold: scale/shift/invstd, scaled filters (single pair and the multi-pair launch), and the
    backward identities dgamma = invstd (sum W.G - mean sum g), dW = scale G against autograd through conv -> BN(eval) -> ReLU."""
    ops = _ops()
    from rg_hip import nn as rnn
    g = torch.Generator().manual_seed(31)
    N, C, H, W, K = 4, 24, 10, 6, 40
    conv = torch.nn.Conv2d(C, K, 3, padding=1, bias=False).double()
    bn = torch.nn.BatchNorm2d(K).double().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(N, C, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    y = F.relu(bn(conv(xr)))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    rconv, rbn = rnn.Conv2d(C, K, 3, padding=1, bias=False), rnn.BatchNorm2d(K)
    rconv.load_state_dict({k: v.float() for k, v in conv.state_dict().items()})
    rbn.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    rconv.to(dev), rbn.to(dev).eval()
    grp = rnn.FoldGroup([(rconv, rbn)])
    assert grp.usable()
    grp.prepare()
    f = rconv._rg_fold
    inv = 1.0 / torch.sqrt(bn.running_var + bn.eps)
    _close(f.scale, bn.weight * inv, name="scale")
    _close(f.shift, bn.bias - bn.running_mean * bn.weight * inv, name="shift")
    _close(f.w_scaled, conv.weight * (bn.weight * inv).view(-1, 1, 1, 1), name="scaled filters")
    _close(f.w_scaled_krsc, (conv.weight * (bn.weight * inv).view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(K, 9, C),
           name="scaled KRSC filters")
    from rg_hip.tape import Tape
    tape = Tape(param_grad=True)
    yd = rnn.conv_bn_tf(tape, rconv, rbn, x.to(dev), act=ops.ACT_RELU)
    _close(yd, y, name="fused forward")
    dx = rnn.conv_bn_tb(tape, rconv, rbn, dy.to(dev))
    _close(dx, xr.grad, name="dx")
    _close(tape.grads[id(rconv.weight)], conv.weight.grad, tol=5e-5, name="dW")
    _close(tape.grads[id(rbn.weight)], bn.weight.grad, tol=5e-5, name="dgamma")
    _close(tape.grads[id(rbn.bias)], bn.bias.grad, tol=5e-5, name="dbeta")


def test_error_paths(dev):
    """the C ABI refuses instead of mis-computing: oversize tensors, bad geometry, k > cols, mismatched shapes"""
    ops = _ops()
    x = torch.zeros(1, 4, 8, 8, device=dev)
    w = torch.zeros(4, 5, 3, 3, device=dev)
    with pytest.raises(ValueError):
        ops.conv2d_fwd(x, w, 1, 1)                               # channel mismatch
    with pytest.raises(RuntimeError) as e:
        ops.conv2d_dgrad(torch.zeros(1, 4, 3, 3, device=dev), torch.zeros(4, 4, 3, 3, device=dev), (8, 8), 3, 1)
    assert "stride" in str(e.value)
    with pytest.raises(RuntimeError) as e:                      # geometry that no convolution produces
        ops.conv2d_dgrad(torch.zeros(1, 4, 4, 4, device=dev), torch.zeros(4, 4, 3, 3, device=dev), (8, 8), 3, 1)
    assert "larger than input" in str(e.value)
    with pytest.raises(RuntimeError) as e:
        ops.topk_rows(torch.zeros(2, 5, device=dev), 6)
    assert "k must be <= cols" in str(e.value)
    with pytest.raises(TypeError):
        ops.conv2d_fwd(x.double(), torch.zeros(4, 4, 3, 3, device=dev), 1, 1)
    i, v = ops.topk_rows(torch.tensor([[1.0, 3.0, 3.0, 2.0, -1.0]], device=dev), 5)          # k == cols, ties by lower index
    assert i.cpu().tolist() == [[1, 2, 3, 0, 4]] and v.cpu().tolist() == [[3.0, 3.0, 2.0, 1.0, -1.0]]


@pytest.mark.parametrize("case", [(48, 64, 32, 16, 64, 3, 1, 1), (32, 32, 33, 17, 48, 3, 2, 1), (32, 64, 32, 16, 128, 1, 2, 0),
                                  (32, 40, 24, 20, 24, 1, 1, 0), (2, 64, 8, 4, 64, 3, 1, 1),
                                  (32, 64, 64, 32, 64, 3, 1, 1)])        # tap-reuse kernel, 512 tiles: no split, row sums written
def test_dgrad_rowsum(dev, case):
    """the row sums a dgrad launch writes next to dx add up to dx.sum over (n, h, w) per channel — including residual + mask"""
    ops = _ops()
    N, C, H, W, K, k, s, p = case
    g = torch.Generator().manual_seed(41 + C)
    w = torch.randn(K, C, k, k, generator=g) / math.sqrt(C * k * k)
    P, Q = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(N, K, P, Q, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    mask = torch.randn(N, C, H, W, generator=g)
    wk = ops.weights_to_krsc(w.to(dev)) if (k > 1 and C % 4 == 0) else None
    dx = ops.conv2d_dgrad(dy.to(dev), w.to(dev), (H, W), s, p, residual=res.to(dev), relu_mask=mask.to(dev), w_krsc=wk,
                          want_rowsum=True)
    part = getattr(dx, "_rg_rowsum", None)
    from rg_hip.lib import lib
    cols = lib.rg_conv2d_dgrad_rowsum_cols(N, C, H, W, K, k, k, s, s, p, p, P, Q)
    if cols == 0:                       # split-K launch: no fused row sums, the consumer falls back to its own pass
        assert part is None
        return
    assert part is not None and tuple(part.shape) == (C, cols)
    _close(part.sum(1), dx.double().sum((0, 2, 3)), tol=2e-5, name="row sums")


@pytest.mark.parametrize("shape,affine,act", [((3, 5, 7, 9), True, "leaky"),        # 63 elements: one wave per instance, scalar loads
                                              ((2, 16, 32, 16), True, "none"),      # 512: one wave, float4
                                              ((2, 6, 64, 48), True, "relu"),       # 3072: one workgroup per instance, float4
                                              ((2, 3, 51, 43), False, "none"),      # 2193: one workgroup, scalar loads
                                              ((4, 8, 1, 50), False, "leaky"),      # token maps [B, C, L]
                                              ((3, 7, 16, 8), True, "leaky"),       # 128 elements: 16 lanes per instance, 4 per wave
                                              ((5, 9, 3, 3), True, "none"),         # 9 elements, scalar loads, ragged instance count
                                              ((2, 4, 64, 32), True, "leaky"),      # 2048: 64 lanes x 8 float4 in registers
                                              ((1, 3, 128, 64), True, "relu"),      # 8192: one workgroup x 8 float4 per thread
                                              ((3, 5, 8, 8), False, "none"),        # 64: 16 lanes x 1
                                              ((2, 4, 16, 16), True, "leaky"),      # 256: 32 lanes x 2
                                              ((2, 2, 256, 128), True, "none")])    # 32768: beyond the register kernels (loop kernel)
def test_instance_norm_single_launch(dev, shape, affine, act):
    """rg_instnorm_fwd / rg_instnorm_bwd against torch's instance_norm in fp64: output, statistics, dx, residual gradient, the
    per-instance sums behind dgamma / dbeta, and the per-instance sums of dx (bias gradient of the convolution in front)."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    N, C = shape[:2]
    x = torch.randn(shape, generator=g) * 1.7 + 0.3
    res = torch.randn(shape, generator=g)
    gam = (torch.rand(C, generator=g) + 0.5) if affine else None
    bet = torch.randn(C, generator=g) if affine else None
    a, slope = {"none": (ops.ACT_NONE, 0.0), "relu": (ops.ACT_RELU, 0.0), "leaky": (ops.ACT_LEAKY, 0.1)}[act]
    fn = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.1)}[act]
    xd, rd = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gd = gam.double().requires_grad_(True) if affine else None
    bd = bet.double().requires_grad_(True) if affine else None
    yref = fn(F.instance_norm(xd, weight=gd, bias=bd, eps=1e-5) + rd)
    dy = torch.randn(shape, generator=g)
    yref.backward(dy.double())
    y, mean, invstd = ops.instnorm_fwd(x.to(dev), gam.to(dev) if affine else None, bet.to(dev) if affine else None, res.to(dev), 1e-5,
                                       a, slope)
    _close(y, yref, name="y")
    _close(mean.view(N, C), x.double().flatten(2).mean(2), name="mean")
    dx, dres, s1, s2, s3 = ops.instnorm_bwd(x.to(dev), dy.to(dev), y, mean, invstd, gam.to(dev) if affine else None, a, slope,
                                            need_dx=True, need_dres=True)
    _close(dx, xd.grad, tol=5e-5, name="dx")
    _close(dres, rd.grad, name="dres")
    if affine:
        dg, db = ops.rows_sum_pair(s2, s1, N, C)
        _close(dg, gd.grad, tol=5e-5, name="dgamma")
        _close(db, bd.grad, tol=5e-5, name="dbeta")
    # sums of dx per instance: compare with the sums of the kernel's own dx (mathematically ~0, numerically noise)
    ref3 = dx.double().flatten(2).sum(2).flatten()
    assert (s3.double() - ref3).abs().max().item() <= 1e-4 * max(dx.abs().max().item(), 1e-6) * dx[0, 0].numel() ** 0.5


def test_pair_cat(dev):
    """FDGANModel.set_input's pair batch: cat([x1, x1*mask + x2*(1-mask)]) with a 0/1 mask is a per-sample selection"""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    for shape in ((5, 3, 6, 4), (4, 7), (3, 18, 5, 3)):
        a, b = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
        lab = torch.randint(0, 2, (shape[0],), generator=g)
        m = lab.view(-1, *([1] * (len(shape) - 1))).float()
        ref = torch.cat([a, a * m + b * (1 - m)])
        assert torch.equal(ops.pair_cat(a.to(dev), b.to(dev), lab.to(dev)).cpu(), ref)
        assert torch.equal(ops.pair_cat(a.to(dev), b.to(dev)).cpu(), torch.cat([a, b]))


def test_spectral_norm_multi_equals_per_layer(dev):
    """rg_spectral_norm_fwd_multi (all filters of a discriminator forward in two launches) == the per-layer entry point bit for bit:
    W / sigma, sigma, the in-place u / v update and the saved copies"""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    shapes = [(32, 3, 3, 3), (64, 32, 3, 3), (1, 128, 1, 1), (128, 64, 4, 4), (5, 7, 1, 1)]
    ws = [torch.randn(s, generator=g).to(dev) for s in shapes]
    us = [F.normalize(torch.randn(s[0], generator=g), dim=0).to(dev) for s in shapes]
    vs = [F.normalize(torch.randn(s[1] * s[2] * s[3], generator=g), dim=0).to(dev) for s in shapes]
    for training in (True, False):
        u1, v1 = [u.clone() for u in us], [v.clone() for v in vs]
        u2, v2 = [u.clone() for u in us], [v.clone() for v in vs]
        single = [ops.spectral_norm_fwd(w, u, v, training, 1e-12, save_uv=True) for w, u, v in zip(ws, u1, v1)]
        multi = ops.spectral_norm_fwd_multi(list(zip(ws, u2, v2)), training, 1e-12, save_uv=True)
        for i, (a, b) in enumerate(zip(single, multi)):
            for x, y, what in zip(a, b, ("w_sn", "sigma", "u saved", "v saved")):
                assert torch.equal(x, y), (i, what, training)
            assert torch.equal(u1[i], u2[i]) and torch.equal(v1[i], v2[i]), (i, training)


@pytest.mark.parametrize("shape,act", [((8, 256, 16, 8), "relu"),       # float4 rows, layer3-like
                                       ((5, 130, 7, 3), "leaky"),       # scalar rows, ragged channel count
                                       ((64, 2048), "none"),            # BatchNorm1d on [N, C] features
                                       ((16, 128, 16, 8), "leaky"),     # register kernels: 2 float4 per thread
                                       ((32, 128, 16, 8), "relu"),      # 4
                                       ((16, 128, 32, 16), "none"),     # 8
                                       ((32, 128, 32, 16), "relu"),     # 16 (the largest one-launch geometry)
                                       ((7, 128, 12, 20), "leaky")])    # ragged: 420 float4 per channel
def test_batchnorm_train_single_launch(dev, shape, act):
    """rg_bn_train_fwd_fused / rg_bn_train_bwd_fused against torch.nn.functional.batch_norm (training) in fp64: output, batch
    statistics, running-statistics update, dx, residual gradient, dgamma / dbeta — and bit-compatible tape records with the
    slice-parallel kernels (same mean / invstd layout)."""
    ops = _ops()
    assert ops.bn_train_fused_ok(torch.empty(shape, device=dev))
    g = torch.Generator().manual_seed(31)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 1.3 + 0.2
    res = torch.randn(shape, generator=g)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    a, slope = {"none": (ops.ACT_NONE, 0.0), "relu": (ops.ACT_RELU, 0.0), "leaky": (ops.ACT_LEAKY, 0.2)}[act]
    fn = {"none": lambda t: t, "relu": F.relu, "leaky": lambda t: F.leaky_relu(t, 0.2)}[act]
    xd, rd = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gd, bd = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    yref = fn(F.batch_norm(xd, rm, rv, gd, bd, True, 0.1, 1e-5) + rd)
    dy = torch.randn(shape, generator=g)
    yref.backward(dy.double())
    rmd, rvd = rm0.to(dev), rv0.to(dev)
    y, mean, invstd = ops.bn_train_fwd_fused(x.to(dev), gam.to(dev), bet.to(dev), res.to(dev), rmd, rvd, 1e-5, 0.1, a, slope)
    _close(y, yref, name="y")
    _close(rmd, rm, name="running_mean")
    _close(rvd, rv, name="running_var")
    red = [d for d in range(len(shape)) if d != 1]
    _close(mean, x.double().mean(red), name="mean")
    dx, dres, s1, s2 = ops.bn_train_bwd_fused(x.to(dev), dy.to(dev), y, mean, invstd, gam.to(dev), a, slope, need_dx=True,
                                              need_dres=True)
    _close(dx, xd.grad, tol=5e-5, name="dx")
    _close(dres, rd.grad, name="dres")
    _close(s2, gd.grad, tol=5e-5, name="dgamma")
    _close(s1, bd.grad, tol=5e-5, name="dbeta")
    # the slice-parallel kernels on the same input agree (the module picks either by geometry)
    m2, i2 = ops.bn_stats(x.to(dev), None, None, 1e-5, 0.1)
    _close(m2, mean, tol=1e-6, name="mean vs bn_stats")
    _close(i2, invstd, tol=1e-5, name="invstd vs bn_stats")


PLANES_CASES = [c for c in CONV_CASES if c[4] > 4 and c[1] > 4][:28] + [
    (16, 128, 16, 8, 256, 1, 1, 1, 0),    # 128 x 128 tiles: 1x1 (float4 both operands)
    (16, 64, 32, 16, 128, 4, 4, 2, 1),    # 128 x 128 tiles: gather loaders, strided data gradient
    (8, 128, 16, 16, 128, 3, 3, 2, 1),    # 3x3 / 2
    (6, 160, 12, 10, 136, 1, 1, 1, 0),    # ragged rows / columns on the big tile
]


@pytest.mark.parametrize("mask", [7, 15])          # 15: the eight-wave form where the plan uses the 128 x 128 tile
@pytest.mark.parametrize("case", PLANES_CASES)
def test_conv_planes_kernels(dev, case, mask):
    """the bf16-plane operand path (csrc/conv_planes.h: operands split once on their way into LDS, fragments by ds_read_b128 /
    ds_read_b64_tr_b16) — not the default (it measured no faster) but kept correct: same bounds as the default
    kernels, every loader variant (float4 along k / along the rows, blocked scalar gathers, ragged tiles, split-K)"""
    ops = _ops()
    from rg_hip import lib as rglib
    N, C, H, W, K, KH, KW, s, p = _geom(case)
    g = torch.Generator().manual_seed(4321 + N * 7 + C)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, KH, KW, generator=g) / math.sqrt(C * KH * KW)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    old = rglib.lib.rg_conv_set_planes(mask)
    try:
        _close(ops.conv2d_fwd(x.to(dev), w.to(dev), s, p), y_ref, name="fwd (planes)")
        _close(ops.conv2d_dgrad(dy.to(dev), w.to(dev), (H, W), s, p), xd.grad, name="dgrad (planes)")
        _close(ops.conv2d_wgrad(x.to(dev), dy.to(dev), (K, C, KH, KW), s, p), wd.grad, tol=5e-5, name="wgrad (planes)")
    finally:
        rglib.lib.rg_conv_set_planes(old)


# (N, C, H, W, K, k, stride, pad, forced tile (0: 128x128, 1: 64x128, 2: 64x64, 3: 32x256), forced splits); N*P*Q % 4 == 0
SPLITK_CASES = [
    (8, 256, 8, 4, 208, 1, 1, 0, 2, 4),     # 1x1, float4 pixel operand, ragged rows on 64 x 64 tiles
    (4, 64, 16, 6, 96, 3, 1, 1, 1, 3),      # 3x3 (r,s)-major gather, 64 x 128 tiles, three splits (width 6: not a tap-reuse geometry)
    (8, 160, 8, 6, 136, 3, 1, 1, 0, 5),     # 128 x 128 tiles (eight-wave candidate), five splits: the 4 + 1 summation tail
    (2, 20, 12, 8, 40, 3, 1, 1, 3, 2),      # generic (c, r, s) gather, 32 x 256 tile, ragged columns
    (32, 1024, 8, 4, 256, 1, 1, 0, 2, 8),   # a layer3-size 1x1 at the bench's 32 crops: many tiles race for the last arrival
    (16, 128, 16, 8, 128, 3, 1, 1, 1, 4),   # tap-reuse geometry: conv3x3_halo_kernel plans its own splits and competes by measurement
]


@pytest.mark.parametrize("mask", [None, 7, 15])     # measured kernel choice / plane path, four waves / eight waves
@pytest.mark.parametrize("case", SPLITK_CASES)
def test_conv_splitk_finishes_in_kernel_with_the_finishing_kernels_bits(dev, case, mask):
    """split-K without the finishing launch (rg_conv_splitk_arrivals): the last split of a tile to arrive sums the tile's partials in
    split order and applies the epilogue — the same bits as the finishing kernel, whichever workgroup came last, with every epilogue
    term (scale, shift, residual, activation / ReLU mask), repeatedly (the counters clean themselves), and right against fp64"""
    ops = _ops()
    from rg_hip import lib as rglib
    N, C, H, W, K, k, s, p, tile, splits = case
    g = torch.Generator().manual_seed(777 + C + K)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, k, k, generator=g) / math.sqrt(C * k * k)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    y_ref = F.conv2d(x.double(), w.double(), stride=s, padding=p)
    res = torch.randn(y_ref.shape, generator=g)
    dy = torch.randn(y_ref.shape, generator=g)
    dres = torch.randn(x.shape, generator=g)
    xd = x.double().requires_grad_(True)
    F.conv2d(xd, w.double(), stride=s, padding=p).backward(dy.double())
    yf_ref = F.leaky_relu(y_ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + res.double(), 0.2)
    dx_ref = torch.where(x.double() > 0, xd.grad + dres.double(), torch.zeros_like(xd.grad))
    X, Wt, SC, SH, RES, DY, DRES = (t.to(dev) for t in (x, w, sc, sh, res, dy, dres))

    def run():
        return (ops.conv2d_fwd(X, Wt, s, p), ops.conv2d_fwd(X, Wt, s, p, scale=SC, shift=SH, residual=RES, act=ops.ACT_LEAKY, slope=0.2),
                ops.conv2d_dgrad(DY, Wt, (H, W), s, p), ops.conv2d_dgrad(DY, Wt, (H, W), s, p, residual=DRES, relu_mask=X))

    old = rglib.lib.rg_conv_set_planes(mask) if mask is not None else None
    rglib.lib.rg_conv_set_force(tile, splits)
    ops._ws_sizes.clear()
    was = ops.splitk_inkernel(True)
    try:
        c0 = rglib.lib.rg_conv_splitk_inkernel_count()
        a = run()
        c1 = rglib.lib.rg_conv_splitk_inkernel_count()
        if k == 1 or W & (W - 1):                           # (tap-reuse geometries may run unsplit on the kernel's own plan)
            assert c1 >= c0 + 4, (c0, c1)                   # all four launches finished in the kernel
        b = run()                                           # the counters are clean again
        ops.splitk_inkernel(False)
        c2 = rglib.lib.rg_conv_splitk_inkernel_count()
        f = run()                                           # the finishing kernel
        assert rglib.lib.rg_conv_splitk_inkernel_count() == c2
        for i, name in enumerate(("fwd", "fwd + epilogue", "dgrad", "dgrad + residual + mask")):
            assert torch.equal(a[i], f[i]), name + ": in-kernel finish differs from the finishing kernel"
            assert torch.equal(a[i], b[i]), name + ": second in-kernel run differs (stale counters?)"
        _close(a[0], y_ref, name="fwd (split-K in kernel)")
        _close(a[1], yf_ref, name="fwd + epilogue (split-K in kernel)")
        _close(a[2], xd.grad, name="dgrad (split-K in kernel)")
        _close(a[3], dx_ref, name="dgrad + residual + mask (split-K in kernel)")
    finally:
        ops.splitk_inkernel(was)
        rglib.lib.rg_conv_set_force(-1, -1)
        ops._ws_sizes.clear()
        if old is not None:
            rglib.lib.rg_conv_set_planes(old)


def test_conv_kernel_choice_is_measured_once_and_results_do_not_change(dev):
    """the first call of a geometry times the round-3 kernel and the plane path and keeps the faster; the output of that first call
    already comes from the kept kernel, so repeating the call gives the same bits (fwd, dgrad, wgrad)"""
    import ctypes
    ops = _ops()
    from rg_hip import lib as rglib
    g = torch.Generator().manual_seed(99)
    x = torch.randn(6, 48, 20, 12, generator=g).to(dev)             # a geometry no other test uses
    w = (torch.randn(80, 48, 3, 3, generator=g) * 0.05).to(dev)
    before = rglib.lib.rg_conv_tune_stats(None)
    y1 = ops.conv2d_fwd(x, w, 2, 1)
    dy = torch.randn(y1.shape, generator=g).to(dev)
    dx1 = ops.conv2d_dgrad(dy, w, (20, 12), 2, 1)
    dw1 = ops.conv2d_wgrad(x, dy, (80, 48, 3, 3), 2, 1)
    cnt = (ctypes.c_int * 2)()
    after = rglib.lib.rg_conv_tune_stats(ctypes.addressof(cnt))
    # at least one measured choice per call (the forward also measures its tile / split plan, one kernel choice per candidate plan)
    assert after >= before + 3 and 3 <= cnt[0] + cnt[1] <= after, (before, after, list(cnt))
    for _ in range(2):
        assert torch.equal(ops.conv2d_fwd(x, w, 2, 1), y1)
        assert torch.equal(ops.conv2d_dgrad(dy, w, (20, 12), 2, 1), dx1)
        assert torch.equal(ops.conv2d_wgrad(x, dy, (80, 48, 3, 3), 2, 1), dw1)
    assert rglib.lib.rg_conv_tune_stats(None) == after
    _close(y1, F.conv2d(x.double().cpu(), w.double().cpu(), stride=2, padding=1), name="fwd (chosen kernel)")
    # a geometry above 1 GFLOP also measures its tile / split-K PLAN (forward, data gradient, weight-gradient split depth): same
    # contract — decided in the first call, the same bits afterwards, right against the fp64 reference
    x = torch.randn(16, 128, 34, 18, generator=g).to(dev)
    w = (torch.randn(256, 128, 3, 3, generator=g) * 0.03).to(dev)
    before = rglib.lib.rg_conv_tune_stats(None)
    y1 = ops.conv2d_fwd(x, w, 1, 1)
    dy = torch.randn(y1.shape, generator=g).to(dev)
    dx1 = ops.conv2d_dgrad(dy, w, (34, 18), 1, 1)
    dw1 = ops.conv2d_wgrad(x, dy, (256, 128, 3, 3), 1, 1)
    after = rglib.lib.rg_conv_tune_stats(None)
    assert after >= before + 6, (before, after)            # three plan choices + at least one kernel choice each
    for _ in range(2):
        assert torch.equal(ops.conv2d_fwd(x, w, 1, 1), y1)
        assert torch.equal(ops.conv2d_dgrad(dy, w, (34, 18), 1, 1), dx1)
        assert torch.equal(ops.conv2d_wgrad(x, dy, (256, 128, 3, 3), 1, 1), dw1)
    assert rglib.lib.rg_conv_tune_stats(None) == after
    xd, wd, dyd = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True), dy.double().cpu()
    yd = F.conv2d(xd, wd, stride=1, padding=1)
    yd.backward(dyd)
    _close(y1, yd.detach(), name="fwd (chosen plan)")
    _close(dx1, xd.grad, name="dgrad (chosen plan)")
    _close(dw1, wd.grad, name="wgrad (chosen plan)")


def test_concurrent_stream_runs_beside_the_streams_it_avoids(dev):
    """rg_hip.ops.concurrent_stream: the stream it hands out does not share a hardware queue with the stream it was asked to avoid
    (an idle wave on each takes one wave's time, not two), nor with the one handed out before it; rg_spin_us refuses nonsense."""
    import time
    from rg_hip import ops
    from rg_hip.lib import lib
    main = torch.cuda.current_stream()
    a = ops.concurrent_stream(dev, avoid=(main,))
    b = ops.concurrent_stream(dev, avoid=(main,))
    assert a.cuda_stream != main.cuda_stream and b.cuda_stream not in (a.cuda_stream, main.cuda_stream)

    def both(x, y, us=300):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            lib.rg_spin_us(us, x.cuda_stream)
            lib.rg_spin_us(us, y.cuda_stream)
            x.synchronize()
            y.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t0))
        return best
    one = both(main, main)
    assert one >= 2 * 300 * 0.95, one                      # same stream: strictly one after the other (the wave really idles 300 us)
    for x, y in ((main, a), (main, b), (a, b)):
        t = both(x, y)
        assert t < 0.8 * one, (t, one)
    with pytest.raises(RuntimeError, match="rg_spin_us"):
        lib.rg_spin_us(0, None)
