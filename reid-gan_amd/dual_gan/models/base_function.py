"""Building blocks of the dual_gan generators / discriminator on the HIP kernels.

Mirror of CC/dual_gan/models/base_function.py (same names, constructor arguments, sub-module layout and therefore
state_dict keys: `model.0.weight`, `shortcut.1.weight_orig`, ...).  Each block is an RGModule whose tf/tb are
hand-written kernel programs: the LeakyReLU(0.1) after a conv / InstanceNorm runs in the producer's epilogue, the
`model(x) + shortcut(x)` add of the residual blocks runs in the epilogue of the last conv, and the two input
gradients of a residual block meet in a dgrad epilogue.
"""
from __future__ import absolute_import

import functools

import torch
from torch import nn
from torch.nn import init
from torch.optim import lr_scheduler

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.ops import ACT_LEAKY, ACT_TANH
from rg_hip.parallel import DataParallel
from rg_hip.tape import RGModule, _param_list


######################################################################################
# base function for network structure   (base_function.py:13-126)
######################################################################################
def init_weights(net, init_type='normal', gain=0.02):
    """Get different initial method for the network weights (base_function.py:13-35).  A spectral-normed conv keeps
    its default `weight_orig` — the reference initialises the hook's derived `weight` attribute there — and gets a
    zero bias."""
    def init_func(m):
        classname = m.__class__.__name__
        if isinstance(m, rnn.SNConv2d):
            if m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif getattr(m, 'weight', None) is not None and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                # drawn and factorised on the host (same distribution; the device QR of hipSOLVER is the only library call
                # the construction would make, and it does not survive rocprofv3's counter collection)
                w = torch.empty(m.weight.shape, dtype=torch.float32)
                init.orthogonal_(w, gain=gain)
                m.weight.data.copy_(w)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if getattr(m, 'bias', None) is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, 0.02)
            init.constant_(m.bias.data, 0.0)

    print('initialize network with %s' % init_type)
    net.apply(init_func)


def get_norm_layer(norm_type='batch'):
    """Get the normalization layer for the networks"""
    if norm_type == 'batch':
        norm_layer = functools.partial(rnn.BatchNorm2d, momentum=0.1, affine=True)
    elif norm_type == 'instance':
        norm_layer = functools.partial(rnn.InstanceNorm2d, affine=True)
    elif norm_type == 'none':
        norm_layer = None
    else:
        raise NotImplementedError('normalization layer [%s] is not found' % norm_type)
    return norm_layer


def get_nonlinearity_layer(activation_type='PReLU'):
    """Get the activation layer for the networks (SELU / PReLU of the reference are not on the hot path)."""
    if activation_type == 'ReLU':
        nonlinearity_layer = rnn.ReLU()
    elif activation_type == 'LeakyReLU':
        nonlinearity_layer = rnn.LeakyReLU(0.1)
    elif activation_type in ('SELU', 'PReLU'):
        raise NotImplementedError('activation layer [%s] has no HIP kernel (the training scripts use LeakyReLU)'
                                  % activation_type)
    else:
        raise NotImplementedError('activation layer [%s] is not found' % activation_type)
    return nonlinearity_layer


def get_scheduler(optimizer, opt):
    """Get the training learning rate for different epoch"""
    if opt.gan_lr_policy == 'lambda':
        def lambda_rule(epoch):
            lr_l = 1.0 - max(0, epoch + opt.iter_start - opt.niter) / float(opt.niter_decay + 1)
            return lr_l
        scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    elif opt.gan_lr_policy == 'step':
        scheduler = lr_scheduler.StepLR(optimizer, step_size=opt.lr_step_size, gamma=0.1)
    elif opt.gan_lr_policy == 'exponent':
        scheduler = lr_scheduler.ExponentialLR(optimizer, gamma=0.95)
    elif opt.gan_lr_policy == 'cosine':
        scheduler = lr_scheduler.CosineAnnealingLR(optimizer, T_max=32, eta_min=0)
    else:
        raise NotImplementedError('learning rate policy [%s] is not implemented', opt.gan_lr_policy)
    return scheduler


def print_network(net):
    """print the network"""
    num_params = 0
    for param in net.parameters():
        num_params += param.numel()
    print('%s: total number of parameters: %.3f M' % (net.__class__.__name__, num_params / 1e6))


def init_net(net, init_type='normal'):
    """print the network structure and initial the network; `.module` is kept through the DataParallel shim
    (one process per GPU here)."""
    print_network(net)
    if torch.cuda.is_available():
        net.cuda()
        net = DataParallel(net)
    init_weights(net, init_type)
    return net


def _freeze(*args):
    """freeze the network for forward process"""
    for module in args:
        if module:
            for p in _param_list(module):
                p.requires_grad = False


def _unfreeze(*args):
    """ unfreeze the network for parameter update"""
    for module in args:
        if module:
            for p in _param_list(module):
                p.requires_grad = True


def spectral_norm(module, use_spect=True):
    """use spectral normal layer to stable the training process"""
    if use_spect:
        if not isinstance(module, rnn.Conv2d):
            raise NotImplementedError('spectral norm is built for nn.Conv2d (the discriminator); got %s'
                                      % module.__class__.__name__)
        return rnn.SNConv2d.from_conv(module)
    else:
        return module


SpectralNorm = spectral_norm


def coord_conv(input_nc, output_nc, use_spect=False, use_coord=False, with_r=False, **kwargs):
    """use coord convolution layer to add position information"""
    if use_coord:
        raise NotImplementedError('CoordConv is not on the hot path (the reference prints ERROR banners for it)')
    return spectral_norm(rnn.Conv2d(input_nc, output_nc, **kwargs), use_spect)


def _slope(nonlinearity):
    if isinstance(nonlinearity, rnn.LeakyReLU):
        return ACT_LEAKY, nonlinearity.negative_slope
    if isinstance(nonlinearity, rnn.ReLU):
        return ops.ACT_RELU, 0.0
    raise NotImplementedError('nonlinearity %s' % nonlinearity.__class__.__name__)


######################################################################################
# Network basic function
######################################################################################
class _NormActConv(object):
    """Shared program pieces: [norm ->] act -> conv, with the activation fused into the norm's apply pass when there
    is a norm and run as one element-wise pass when there is none."""

    @staticmethod
    def pre_tf(tape, norm, act, slope, x):
        if norm is not None:
            return norm.tf(tape, x, act=act, slope=slope)
        y = ops.act_fwd(x, act, slope)
        tape.push(y)
        return y

    @staticmethod
    def pre_tb(tape, norm, act, slope, dy):
        if norm is not None:
            return norm.tb(tape, dy)
        y = tape.pop()
        return ops.act_bwd(dy, y, act, slope)


class ResBlock(RGModule):
    """Residual block (base_function.py:193-233), sample_type 'none' (the only form the generators use):
    out = conv3x3(act(norm(conv3x3(act(norm(x)))))) + conv1x1(x)."""

    def __init__(self, input_nc, output_nc, hidden_nc=None, norm_layer=rnn.BatchNorm2d, nonlinearity=None,
                 sample_type='none', use_spect=False, use_coord=False):
        super(ResBlock, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        hidden_nc = output_nc if hidden_nc is None else hidden_nc
        self.sample = True
        if sample_type == 'none':
            self.sample = False
        elif sample_type in ('up', 'down'):
            raise NotImplementedError("ResBlock sample_type '%s' is not used by the generators of the training scripts"
                                      % sample_type)
        else:
            raise NotImplementedError('sample type [%s] is not found' % sample_type)
        kwargs = {'kernel_size': 3, 'stride': 1, 'padding': 1}
        kwargs_short = {'kernel_size': 1, 'stride': 1, 'padding': 0}
        self.conv1 = coord_conv(input_nc, hidden_nc, use_spect, use_coord, **kwargs)
        self.conv2 = coord_conv(hidden_nc, output_nc, use_spect, use_coord, **kwargs)
        self.bypass = coord_conv(input_nc, output_nc, use_spect, use_coord, **kwargs_short)
        self._act = _slope(nonlinearity)
        if norm_layer is None:
            self.model = nn.Sequential(nonlinearity, self.conv1, nonlinearity, self.conv2,)
            self._ix = (None, None)
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, self.conv1, norm_layer(hidden_nc), nonlinearity,
                                       self.conv2,)
            self._ix = (0, 3)
        self.shortcut = nn.Sequential(self.bypass,)

    def tf(self, tape, x):
        act, slope = self._act
        n1, n2 = self._ix
        m = self.model
        sc = self.bypass.tf(tape, x)
        a = _NormActConv.pre_tf(tape, m[n1] if n1 is not None else None, act, slope, x)
        if n2 is None:
            h = self.conv1.tf(tape, a, act=act, slope=slope)
        else:
            h = m[n2].tf(tape, self.conv1.tf(tape, a), act=act, slope=slope)
        return self.conv2.tf(tape, h, residual=sc)

    def tb(self, tape, dy, need_dx=True):
        act, slope = self._act
        n1, n2 = self._ix
        m = self.model
        d = self.conv2.tb(tape, dy)
        if n2 is not None:
            d = m[n2].tb(tape, d)
        d = self.conv1.tb(tape, d)
        d = _NormActConv.pre_tb(tape, m[n1] if n1 is not None else None, act, slope, d)
        return self.bypass.tb(tape, dy, residual=d)


class EncoderBlockOptimized(RGModule):
    """Encoder block for the first layer of the generator (base_function.py:236-257):
    conv 4x4/2 -> [norm] -> act -> conv 3x3."""

    def __init__(self, input_nc, output_nc, norm_layer=rnn.BatchNorm2d, nonlinearity=None, use_spect=False, use_coord=False):
        super(EncoderBlockOptimized, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        kwargs_down = {'kernel_size': 4, 'stride': 2, 'padding': 1}
        kwargs_fine = {'kernel_size': 3, 'stride': 1, 'padding': 1}
        conv1 = coord_conv(input_nc, output_nc, use_spect, use_coord, **kwargs_down)
        conv2 = coord_conv(output_nc, output_nc, use_spect, use_coord, **kwargs_fine)
        self._act = _slope(nonlinearity)
        if norm_layer is None:
            self.model = nn.Sequential(conv1, nonlinearity, conv2)
            self._ix = (0, None, 2)
        else:
            self.model = nn.Sequential(conv1, norm_layer(output_nc), nonlinearity, conv2)
            self._ix = (0, 1, 3)

    def tf(self, tape, x):
        act, slope = self._act
        c1, n, c2 = self._ix
        if n is None:
            h = self.model[c1].tf(tape, x, act=act, slope=slope)
        else:
            h = self.model[n].tf(tape, self.model[c1].tf(tape, x), act=act, slope=slope)
        return self.model[c2].tf(tape, h)

    def tb(self, tape, dy, need_dx=True):
        c1, n, c2 = self._ix
        d = self.model[c2].tb(tape, dy)
        if n is not None:
            d = self.model[n].tb(tape, d)
        return self.model[c1].tb(tape, d, need_dx=need_dx)


class FeatureAdaptBlock1(RGModule):
    """Encoder block for the input reid features (base_function.py:274-287): conv 1x1 -> norm -> act."""

    def __init__(self, input_nc, output_nc, norm_layer=rnn.BatchNorm2d, nonlinearity=None):
        super(FeatureAdaptBlock1, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        conv1 = rnn.Conv2d(input_nc, output_nc, kernel_size=1)
        self._act = _slope(nonlinearity)
        self.model = nn.Sequential(conv1, norm_layer(output_nc), nonlinearity)

    def tf(self, tape, x):
        act, slope = self._act
        return self.model[1].tf(tape, self.model[0].tf(tape, x), act=act, slope=slope)

    def tb(self, tape, dy, need_dx=True):
        return self.model[0].tb(tape, self.model[1].tb(tape, dy), need_dx=need_dx)


class EncoderBlock(RGModule):
    """Encoder block for the medium layers of the generator (base_function.py:290-312):
    [norm ->] act -> conv 4x4/2 -> [norm ->] act -> conv 3x3   (without norm: conv, act, conv, act)."""

    def __init__(self, input_nc, output_nc, norm_layer=rnn.BatchNorm2d, nonlinearity=None, use_spect=False, use_coord=False):
        super(EncoderBlock, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        kwargs_down = {'kernel_size': 4, 'stride': 2, 'padding': 1}
        kwargs_fine = {'kernel_size': 3, 'stride': 1, 'padding': 1}
        conv1 = coord_conv(input_nc, output_nc, use_spect, use_coord, **kwargs_down)
        conv2 = coord_conv(output_nc, output_nc, use_spect, use_coord, **kwargs_fine)
        self._act = _slope(nonlinearity)
        self._normed = norm_layer is not None
        if norm_layer is None:
            self.model = nn.Sequential(conv1, nonlinearity, conv2, nonlinearity)
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, conv1,
                                       norm_layer(output_nc), nonlinearity, conv2)

    def tf(self, tape, x):
        act, slope = self._act
        m = self.model
        if not self._normed:
            return m[2].tf(tape, m[0].tf(tape, x, act=act, slope=slope), act=act, slope=slope)
        h = m[2].tf(tape, m[0].tf(tape, x, act=act, slope=slope))
        return m[5].tf(tape, m[3].tf(tape, h, act=act, slope=slope))

    def tb(self, tape, dy, need_dx=True):
        m = self.model
        if not self._normed:
            return m[0].tb(tape, m[2].tb(tape, dy), need_dx=need_dx)
        d = m[3].tb(tape, m[5].tb(tape, dy))
        return m[0].tb(tape, m[2].tb(tape, d))


class ResBlockDecoder(RGModule):
    """Decoder block (base_function.py:315-339): out = convT3x3/2(act(norm(conv3x3(act(norm(x)))))) + convT3x3/2(x).
    `extra` (the generator's skip feature, networks.py:726-729 `F_g += skip_list.pop()`) rides in the same epilogue."""

    def __init__(self, input_nc, output_nc, hidden_nc=None, norm_layer=rnn.BatchNorm2d, nonlinearity=None,
                 use_spect=False, use_coord=False):
        super(ResBlockDecoder, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        hidden_nc = output_nc if hidden_nc is None else hidden_nc
        if use_spect:
            raise NotImplementedError('spectral norm in the generator is never enabled by the reference options')
        conv1 = rnn.Conv2d(input_nc, hidden_nc, kernel_size=3, stride=1, padding=1)
        conv2 = rnn.ConvTranspose2d(hidden_nc, output_nc, kernel_size=3, stride=2, padding=1, output_padding=1)
        bypass = rnn.ConvTranspose2d(input_nc, output_nc, kernel_size=3, stride=2, padding=1, output_padding=1)
        self._act = _slope(nonlinearity)
        self._normed = norm_layer is not None
        if norm_layer is None:
            self.model = nn.Sequential(nonlinearity, conv1, nonlinearity, conv2,)
            self._ix = (None, 1, None, 3)
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, conv1, norm_layer(hidden_nc), nonlinearity, conv2,)
            self._ix = (0, 2, 3, 5)
        self.shortcut = nn.Sequential(bypass)

    def tf(self, tape, x, extra=None):
        act, slope = self._act
        n1, c1, n2, c2 = self._ix
        m = self.model
        sc = self.shortcut[0].tf(tape, x, residual=extra)
        a = _NormActConv.pre_tf(tape, m[n1] if n1 is not None else None, act, slope, x)
        if n2 is None:
            h = m[c1].tf(tape, a, act=act, slope=slope)
        else:
            h = m[n2].tf(tape, m[c1].tf(tape, a), act=act, slope=slope)
        tape.push(extra is not None)
        return m[c2].tf(tape, h, residual=sc)

    def tb(self, tape, dy, need_dx=True):
        """Returns dx, or (dx, d_extra) when the forward had an `extra` input (d_extra is dy itself)."""
        act, slope = self._act
        n1, c1, n2, c2 = self._ix
        m = self.model
        d = m[c2].tb(tape, dy)
        has_extra = tape.pop()
        if n2 is not None:
            d = m[n2].tb(tape, d)
        d = m[c1].tb(tape, d)
        d = _NormActConv.pre_tb(tape, m[n1] if n1 is not None else None, act, slope, d)
        dx = self.shortcut[0].tb(tape, dy, residual=d)
        return (dx, dy) if has_extra else dx


class _ResEncoderBase(RGModule):
    """out = model(x) + bypass(avgpool(x)) of the discriminator blocks (base_function.py:372-420)."""

    def _shortcut_tf(self, tape, x, residual):
        xp = self.shortcut[0].tf(tape, x)
        return self.shortcut[1].tf(tape, xp, residual=residual)

    def _shortcut_tb(self, tape, dy, need_dx=True):
        d = self.shortcut[1].tb(tape, dy, need_dx=need_dx)
        return self.shortcut[0].tb(tape, d, need_dx=need_dx)          # always pops the pool's tape entry


class ResBlockEncoderOptimized(_ResEncoderBase):
    """First discriminator block: conv3x3 -> [norm] -> act -> conv4x4/2, shortcut avgpool -> conv1x1."""

    def __init__(self, input_nc, output_nc, hidden_nc=None, norm_layer=rnn.BatchNorm2d, nonlinearity=None,
                 use_spect=False, use_coord=False):
        super(ResBlockEncoderOptimized, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        hidden_nc = input_nc if hidden_nc is None else hidden_nc
        conv1 = spectral_norm(rnn.Conv2d(input_nc, hidden_nc, kernel_size=3, stride=1, padding=1), use_spect)
        conv2 = spectral_norm(rnn.Conv2d(hidden_nc, output_nc, kernel_size=4, stride=2, padding=1), use_spect)
        bypass = spectral_norm(rnn.Conv2d(input_nc, output_nc, kernel_size=1, stride=1, padding=0), use_spect)
        self._act = _slope(nonlinearity)
        if norm_layer is None:
            self.model = nn.Sequential(conv1, nonlinearity, conv2,)
            self._ix = (0, None, 2)
        else:
            self.model = nn.Sequential(conv1, norm_layer(hidden_nc), nonlinearity, conv2,)
            self._ix = (0, 1, 3)
        self.shortcut = nn.Sequential(rnn.AvgPool2d(kernel_size=2, stride=2), bypass)

    def tf(self, tape, x):
        act, slope = self._act
        c1, n, c2 = self._ix
        m = self.model
        if n is None:
            h = m[c1].tf(tape, x, act=act, slope=slope)
        else:
            h = m[n].tf(tape, m[c1].tf(tape, x), act=act, slope=slope)
        return self._shortcut_tf(tape, x, m[c2].tf(tape, h))

    def tb(self, tape, dy, need_dx=True):
        c1, n, c2 = self._ix
        m = self.model
        dxs = self._shortcut_tb(tape, dy, need_dx)
        d = m[c2].tb(tape, dy)
        if n is not None:
            d = m[n].tb(tape, d)
        return m[c1].tb(tape, d, need_dx=need_dx, residual=dxs)


class ResBlockEncoder(_ResEncoderBase):
    """Medium discriminator block: [norm ->] act -> conv3x3 -> [norm ->] act -> conv4x4/2, shortcut avgpool -> conv1x1."""

    def __init__(self, input_nc, output_nc, hidden_nc=None, norm_layer=rnn.BatchNorm2d, nonlinearity=None,
                 use_spect=False, use_coord=False):
        super(ResBlockEncoder, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        hidden_nc = input_nc if hidden_nc is None else hidden_nc
        conv1 = spectral_norm(rnn.Conv2d(input_nc, hidden_nc, kernel_size=3, stride=1, padding=1), use_spect)
        conv2 = spectral_norm(rnn.Conv2d(hidden_nc, output_nc, kernel_size=4, stride=2, padding=1), use_spect)
        bypass = spectral_norm(rnn.Conv2d(input_nc, output_nc, kernel_size=1, stride=1, padding=0), use_spect)
        self._act = _slope(nonlinearity)
        if norm_layer is None:
            self.model = nn.Sequential(nonlinearity, conv1, nonlinearity, conv2,)
            self._ix = (None, 1, None, 3)
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, conv1,
                                       norm_layer(hidden_nc), nonlinearity, conv2,)
            self._ix = (0, 2, 3, 5)
        self.shortcut = nn.Sequential(rnn.AvgPool2d(kernel_size=2, stride=2), bypass)

    def tf(self, tape, x):
        act, slope = self._act
        n1, c1, n2, c2 = self._ix
        m = self.model
        a = _NormActConv.pre_tf(tape, m[n1] if n1 is not None else None, act, slope, x)
        if n2 is None:
            h = m[c1].tf(tape, a, act=act, slope=slope)
        else:
            h = m[n2].tf(tape, m[c1].tf(tape, a), act=act, slope=slope)
        return self._shortcut_tf(tape, x, m[c2].tf(tape, h))

    def tb(self, tape, dy, need_dx=True):
        act, slope = self._act
        n1, c1, n2, c2 = self._ix
        m = self.model
        dxs = self._shortcut_tb(tape, dy, True)
        d = m[c2].tb(tape, dy)
        if n2 is not None:
            d = m[n2].tb(tape, d)
        d = m[c1].tb(tape, d)
        d = _NormActConv.pre_tb(tape, m[n1] if n1 is not None else None, act, slope, d)
        return ops.axpby(d, dxs, 1.0, 1.0, out=d)


class Output(RGModule):
    """Output layer (base_function.py:423-443): [norm ->] act -> ReflectionPad2d(k // 2) -> conv kxk -> tanh."""

    def __init__(self, input_nc, output_nc, kernel_size=3, norm_layer=rnn.BatchNorm2d, nonlinearity=None,
                 use_spect=False, use_coord=False):
        super(Output, self).__init__()
        nonlinearity = rnn.LeakyReLU() if nonlinearity is None else nonlinearity
        kwargs = {'kernel_size': kernel_size, 'padding': 0, 'bias': True}
        self.conv1 = coord_conv(input_nc, output_nc, use_spect, use_coord, **kwargs)
        self._act = _slope(nonlinearity)
        if norm_layer is None:
            self.model = nn.Sequential(nonlinearity, rnn.ReflectionPad2d(int(kernel_size / 2)), self.conv1, rnn.Tanh())
            self._ix = (None, 1, 2)
        else:
            self.model = nn.Sequential(norm_layer(input_nc), nonlinearity, rnn.ReflectionPad2d(int(kernel_size / 2)),
                                       self.conv1, rnn.Tanh())
            self._ix = (0, 2, 3)

    def tf(self, tape, x):
        act, slope = self._act
        n, pad, conv = self._ix
        m = self.model
        fused = n is None and ops.reflection_pad_fusable(x, m[pad].padding, act, slope)
        if fused:
            # act -> pad as ONE pass over the 64-channel full-resolution map (the separate activation pass and its saved copy go)
            y = m[conv].tf(tape, ops.reflection_pad2d_fwd(x, m[pad].padding, act, slope), act=ACT_TANH)
        else:
            a = _NormActConv.pre_tf(tape, m[n] if n is not None else None, act, slope, x)
            y = m[conv].tf(tape, m[pad].tf(tape, a), act=ACT_TANH)
        tape.push((fused, x if fused else None))
        return y

    def tb(self, tape, dy, need_dx=True):
        act, slope = self._act
        n, pad, conv = self._ix
        m = self.model
        fused, x = tape.pop()
        d = m[conv].tb(tape, dy)
        if fused:
            return ops.reflection_pad2d_bwd(d, m[pad].padding, x, act, slope)
        d = m[pad].tb(tape, d)
        return _NormActConv.pre_tb(tape, m[n] if n is not None else None, act, slope, d)
