"""dual_gan generators / discriminator on the HIP kernels.

Mirror of CC/dual_gan/models/networks.py: `define_G` (:14-33), `define_D` (:36-38), `PoseGenerator1` (:639-738),
`ResDiscriminator` (:917-955) — same constructor arguments, attribute names (`block0`, `encoder{i}`, `feature_block`,
`PCTM`, `decoder{i}`, `outconv`, `conv`) and state_dict keys.  Every network is ONE autograd node (rg_hip.tape) running
the block programs of base_function.py / PTM.py.
"""
from __future__ import absolute_import

import functools

import torch

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.ops import ACT_RELU, ACT_TANH
from rg_hip.tape import RGModule
from .base_function import (EncoderBlock, EncoderBlockOptimized, FeatureAdaptBlock1, Output, ResBlock, ResBlockDecoder,
                            ResBlockEncoder, ResBlockEncoderOptimized, SpectralNorm, get_nonlinearity_layer,
                            get_norm_layer, init_net, _slope)
from .PTM import PCTM, PTM  # noqa: F401


###############################################################################
# Functions
###############################################################################
def define_G(opt, image_nc, pose_nc, ngf=64, img_f=1024, encoder_layer=3, norm='batch', activation='ReLU',
             use_spect=True, use_coord=False, output_nc=3, num_blocks=3, affine=True, nhead=2, num_CABs=2, num_TTBs=2):
    print(opt.model_gen)
    if opt.model_gen == 'Pose':
        netG = PoseGenerator1(ngf, pose_nc, img_f, encoder_layer, norm, activation, use_spect, use_coord, output_nc,
                              affine, nhead, num_CABs, num_TTBs)
    elif opt.model_gen == 'AE':
        netG = AEGenerator(image_nc, ngf, img_f, encoder_layer, norm, activation, use_spect, use_coord, output_nc, num_blocks)
    elif opt.model_gen == 'DPTN':
        netG = DPTNGenerator(image_nc, pose_nc, ngf, img_f, encoder_layer, norm, activation, use_spect, use_coord, output_nc,
                             num_blocks, affine, nhead, num_CABs, num_TTBs)
    elif opt.model_gen == 'DEC':
        netG = DECGenerator1(ngf, img_f, encoder_layer, norm, activation, use_spect, use_coord, output_nc, num_blocks)
    elif opt.model_gen == 'PoseAE':
        # the reference's PoseAEGenerator cannot run: its two-argument forward_enc calls itself with one argument
        # (networks.py:811-813, TypeError on the first forward), so there is no behaviour to reproduce
        raise NotImplementedError("generator 'PoseAE' fails on its first forward in the reference (networks.py:811-813); "
                                  "it is not built on the HIP path")
    elif opt.model_gen == 'FD':
        # (every caller in the reference passes ONE argument to net_G — AE_model.py:205-210 — so this 'add'-mode generator dies on
        # `noise.view` of None there, networks.py:526-527; built for the class / state_dict API and for callers that pass the noise)
        netG = FDGenerator(img_f, ngf, output_nc=3, noise_nc=512, fuse_mode='add')
    else:
        raise TypeError('generator not implemented!')          # the reference's `raise('...')` is a TypeError too
    return init_net(netG, opt.init_type)


def define_D(opt, input_nc=3, ndf=64, img_f=1024, layers=3, norm='none', activation='LeakyReLU', use_spect=True,):
    netD = ResDiscriminator(input_nc, ndf, img_f, layers, norm, activation, use_spect)
    return init_net(netD, opt.init_type)


def print_network(net):
    if isinstance(net, list):
        net = net[0]
    num_params = 0
    for param in net.parameters():
        num_params += param.numel()
    print('Total number of parameters: %d' % num_params)


##############################################################################
# Generator
##############################################################################
class _EncFn(torch.autograd.Function):
    """One autograd node for a sub-program of a generator (forward_enc / forward_dec called on their own)."""

    @staticmethod
    def forward(ctx, net, which, x, *params):
        from rg_hip.tape import Tape
        tape = Tape(param_grad=len(params) > 0, needs_input=[x.requires_grad])
        ctx.net, ctx.which, ctx.tape, ctx.params = net, which, tape, params
        return getattr(net, which + "_tf")(tape, x.detach())

    @staticmethod
    def backward(ctx, g):
        from rg_hip import ops as _ops
        _ops.side_begin()
        try:
            dx = getattr(ctx.net, ctx.which + "_tb")(ctx.tape, g.contiguous(), need_dx=ctx.needs_input_grad[2])
        finally:
            _ops.side_join()
        grads = []
        for p in ctx.params:
            gp = ctx.tape.grads.get(id(p))
            v = getattr(p, "_rg_grad", None)
            if gp is not None and v is not None and gp.data_ptr() == v.data_ptr():
                p.grad = v
                gp = None
            grads.append(gp)
        return (None, None, dx) + tuple(grads)


class AEGenerator(RGModule):
    """Auto-encoder generator (networks.py:278-355): image encoder -> ResBlocks -> residual decoder -> Output."""

    def __init__(self, image_nc, ngf=64, img_f=256, layers=3, norm='batch', activation='ReLU', use_spect=True,
                 use_coord=False, output_nc=3, num_blocks=3):
        super(AEGenerator, self).__init__()
        self.layers = layers
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self.block0 = EncoderBlockOptimized(image_nc, ngf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(self.layers - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder' + str(i), EncoderBlock(ngf * mult_prev, ngf * mult, norm_layer, nonlinearity, use_spect,
                                                           use_coord))
        self.num_blocks = num_blocks
        for i in range(num_blocks):
            setattr(self, 'mblock' + str(i), ResBlock(ngf * mult, ngf * mult, norm_layer=norm_layer, nonlinearity=nonlinearity,
                                                      use_spect=use_spect, use_coord=use_coord))
        for i in range(self.layers):
            mult_prev = mult
            mult = min(2 ** (self.layers - i - 2), img_f // ngf) if i != self.layers - 1 else 1
            setattr(self, 'decoder' + str(i), ResBlockDecoder(ngf * mult_prev, ngf * mult, ngf * mult, norm_layer, nonlinearity,
                                                              use_spect, use_coord))
        self.outconv = Output(ngf, output_nc, 3, None, nonlinearity, use_spect, use_coord)

    def _enc(self):
        return [self.block0] + [getattr(self, 'encoder' + str(i)) for i in range(self.layers - 1)]

    def _dec(self):
        return ([getattr(self, 'mblock' + str(i)) for i in range(self.num_blocks)] +
                [getattr(self, 'decoder' + str(i)) for i in range(self.layers)] + [self.outconv])

    def enc_tf(self, tape, x):
        for m in self._enc():
            x = m.tf(tape, x)
        return x

    def enc_tb(self, tape, dy, need_dx=True):
        mods = self._enc()
        for i in range(len(mods) - 1, -1, -1):
            dy = mods[i].tb(tape, dy, need_dx=(need_dx or i > 0))
        return dy

    def dec_tf(self, tape, x):
        for m in self._dec():
            x = m.tf(tape, x)
        return x

    def dec_tb(self, tape, dy, need_dx=True):
        for m in reversed(self._dec()):
            dy = m.tb(tape, dy)
        return dy

    def tf(self, tape, x):
        return self.dec_tf(tape, self.enc_tf(tape, x))

    def tb(self, tape, dy, need_dx=True):
        return self.enc_tb(tape, self.dec_tb(tape, dy), need_dx=need_dx)

    def _sub(self, which, x, mods):
        params = [p for m in mods for p in m.parameters() if p.requires_grad] if torch.is_grad_enabled() else []
        if not params and not (torch.is_grad_enabled() and x.requires_grad):
            from rg_hip.tape import Tape
            return getattr(self, which + "_tf")(Tape(param_grad=False, record=False), x)
        return _EncFn.apply(self, which, x, *params)

    def forward_enc(self, source):
        return self._sub("enc", source, self._enc())

    def forward_dec(self, feature):
        return self._sub("dec", feature, self._dec())


class SourceEncoder(RGModule):
    """Source Image Encoder En_s (networks.py:54-101)."""

    def __init__(self, image_nc, ngf=64, img_f=1024, encoder_layer=3, norm='batch', activation='ReLU', use_spect=True,
                 use_coord=False):
        super(SourceEncoder, self).__init__()
        self.encoder_layer = encoder_layer
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self.block0 = EncoderBlockOptimized(image_nc, ngf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(encoder_layer - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder' + str(i), EncoderBlock(ngf * mult_prev, ngf * mult, norm_layer, nonlinearity, use_spect,
                                                           use_coord))

    def _mods(self):
        return [self.block0] + [getattr(self, 'encoder' + str(i)) for i in range(self.encoder_layer - 1)]

    def tf(self, tape, x):
        for m in self._mods():
            x = m.tf(tape, x)
        return x

    def tb(self, tape, dy, need_dx=True):
        mods = self._mods()
        for i in range(len(mods) - 1, -1, -1):
            dy = mods[i].tb(tape, dy, need_dx=(need_dx or i > 0))
        return dy


class Resize_ReID(RGModule):
    """Adaptor from synthesised 128x64 images to ReID inputs (networks.py:140-162): bicubic resize to 256x128
    (`my_resize`, rg_bicubic_normalize without the normalisation) followed by three residual blocks added back onto the
    resized image.  Built by `--use_adp`; no method of the reference's training loops calls it."""

    def __init__(self, image_nc, ngf=64, norm='batch', activation='ReLU', use_spect=True, use_coord=False):
        super(Resize_ReID, self).__init__()
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self.resblock1 = ResBlock(image_nc, ngf, norm_layer=norm_layer, nonlinearity=nonlinearity, sample_type='none',
                                  use_spect=use_spect, use_coord=use_coord)
        self.resblock2 = ResBlock(ngf, ngf, norm_layer=norm_layer, nonlinearity=nonlinearity, sample_type='none',
                                  use_spect=use_spect, use_coord=use_coord)
        self.resblock3 = ResBlock(ngf, image_nc, norm_layer=norm_layer, nonlinearity=nonlinearity, sample_type='none',
                                  use_spect=use_spect, use_coord=use_coord)

    def tf(self, tape, inputs):
        x = ops.bicubic_normalize_fwd(inputs, (256, 128))
        tape.push(tuple(inputs.shape[2:]))
        out = self.resblock1.tf(tape, x)
        out = self.resblock2.tf(tape, out)
        out = self.resblock3.tf(tape, out)
        return ops.add(x, out)

    def tb(self, tape, dy, need_dx=True):
        dy = dy.contiguous()
        d = self.resblock3.tb(tape, dy)
        d = self.resblock2.tb(tape, d)
        d = self.resblock1.tb(tape, d, need_dx=need_dx)
        hw = tape.pop()
        if not need_dx:
            return None
        return ops.bicubic_normalize_bwd(ops.add(d, dy), hw)


class DPTNGenerator(RGModule):
    """Dual-task Pose Transformer Network generator (networks.py:165-275): a source->source and a source->target branch
    through SHARED encoder / decoder weights, coupled by the PTM.  Both branches run as ONE 2B batch through the shared
    blocks (instance norm and convolutions are per-sample, so this equals the reference's two sequential passes)."""

    def __init__(self, image_nc, pose_nc, ngf=64, img_f=256, layers=3, norm='batch', activation='ReLU', use_spect=True,
                 use_coord=False, output_nc=3, num_blocks=3, affine=True, nhead=2, num_CABs=2, num_TTBs=2):
        super(DPTNGenerator, self).__init__()
        if norm != 'instance':
            raise NotImplementedError("DPTNGenerator batches its two branches, which needs per-sample normalisation "
                                      "(--norm instance, the training default)")
        self.layers = layers
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        input_nc = 2 * pose_nc + image_nc
        self.block0 = EncoderBlockOptimized(input_nc, ngf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(self.layers - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ngf)
            setattr(self, 'encoder' + str(i), EncoderBlock(ngf * mult_prev, ngf * mult, norm_layer, nonlinearity, use_spect,
                                                           use_coord))
        self.num_blocks = num_blocks
        for i in range(num_blocks):
            setattr(self, 'mblock' + str(i), ResBlock(ngf * mult, ngf * mult, norm_layer=norm_layer, nonlinearity=nonlinearity,
                                                      use_spect=use_spect, use_coord=use_coord))
        self.PTM = PTM(d_model=ngf * mult, nhead=nhead, num_CABs=num_CABs, num_TTBs=num_TTBs, dim_feedforward=ngf * mult,
                       activation="LeakyReLU", affine=affine, norm=norm)
        self.source_encoder = SourceEncoder(image_nc, ngf, img_f, layers, norm, activation, use_spect, use_coord)
        for i in range(self.layers):
            mult_prev = mult
            mult = min(2 ** (self.layers - i - 2), img_f // ngf) if i != self.layers - 1 else 1
            setattr(self, 'decoder' + str(i), ResBlockDecoder(ngf * mult_prev, ngf * mult, ngf * mult, norm_layer, nonlinearity,
                                                              use_spect, use_coord))
        self.outconv = Output(ngf, output_nc, 3, None, nonlinearity, use_spect, use_coord)

    def forward(self, source, source_B, target_B, is_train=True):
        self._is_train = bool(is_train)
        return super(DPTNGenerator, self).forward(source, source_B, target_B)

    def _rg_key(self):
        """state outside the tensor arguments that selects the program (rg_hip.netgraph keys its records on it)"""
        return getattr(self, "_is_train", True)

    def _enc(self):
        return ([self.block0] + [getattr(self, 'encoder' + str(i)) for i in range(self.layers - 1)] +
                [getattr(self, 'mblock' + str(i)) for i in range(self.num_blocks)])

    def _dec(self):
        return [getattr(self, 'decoder' + str(i)) for i in range(self.layers)] + [self.outconv]

    def tf(self, tape, source, source_B, target_B):
        is_train = getattr(self, "_is_train", True)
        B = source.shape[0]
        x = torch.cat((ops.cat_channels([source, source_B, source_B]), ops.cat_channels([source, source_B, target_B])), 0)
        for m in self._enc():
            x = m.tf(tape, x)
        F_s_s, F_s_t = x[:B], x[B:]
        F_s = self.source_encoder.tf(tape, source)
        F_s_t = self.PTM.tf(tape, F_s_s, F_s_t, F_s)
        y = torch.cat((F_s_s, F_s_t), 0) if is_train else F_s_t
        for m in self._dec():
            y = m.tf(tape, y)
        tape.push((B, is_train))
        if is_train:
            return y[B:], y[:B]                       # (out_image_t, out_image_s)
        return y, None

    def tb(self, tape, d_t, d_s=None, need_dx=True):
        B, is_train = tape.pop()
        dev = (d_t if d_t is not None else d_s).device
        if is_train:
            zeros = None
            if d_t is None or d_s is None:
                ref = d_t if d_t is not None else d_s
                zeros = ops.fill_(torch.empty_like(ref), 0.0)
            d = torch.cat((d_s if d_s is not None else zeros, d_t if d_t is not None else zeros), 0)
        else:
            d = d_t
        for m in reversed(self._dec()):
            d = m.tb(tape, d)
        if is_train:
            d_ss, d_st = d[:B].contiguous(), d[B:].contiguous()
        else:
            d_ss, d_st = None, d
        d_src, d_tgt, d_val = self.PTM.tb(tape, d_st)
        d_ss = d_src if d_ss is None else ops.axpby(d_ss, d_src, 1.0, 1.0, out=d_ss)
        ni = tape.needs_input
        need_img = need_dx and (ni is None or bool(ni[0]))
        d_source = self.source_encoder.tb(tape, d_val, need_dx=need_img)
        d = torch.cat((d_ss, d_tgt), 0)
        mods = self._enc()
        need_any = need_dx and (ni is None or any(bool(v) for v in ni))
        for i in range(len(mods) - 1, -1, -1):
            d = mods[i].tb(tape, d, need_dx=(need_any or i > 0))
        if not need_any:
            return None, None, None
        # channels of the stacked inputs: [source | source_B | source_B] and [source | source_B | target_B]
        return _dptn_input_grads(d, B, d_source)


def _dptn_input_grads(d, B, d_source):
    """split the gradient of the two stacked DPTN inputs back onto (source, source_B, target_B)"""
    c_img = 3 if d_source is None else d_source.shape[1]
    c_pose = (d.shape[1] - c_img) // 2
    top, bot = d[:B], d[B:]
    g_src = ops.add(ops.slice_channels(top, 0, c_img), ops.slice_channels(bot, 0, c_img))
    if d_source is not None:
        g_src = ops.add(g_src, d_source)
    g_sb = ops.add(ops.add(ops.slice_channels(top, c_img, c_img + c_pose), ops.slice_channels(top, c_img + c_pose, c_img + 2 * c_pose)),
                   ops.slice_channels(bot, c_img, c_img + c_pose))
    g_tb = ops.slice_channels(bot, c_img + c_pose, c_img + 2 * c_pose)
    return g_src, g_sb, g_tb


class DECGenerator1(RGModule):
    """Decoder-only generator of `--model_gen DEC` (networks.py:401-444): ReID feature map [B, 2048, h, w] -> FeatureAdaptBlock1
    (2048 -> 4*ngf) -> `num_blocks` ResBlocks -> `layers` residual up-sampling decoders -> Output; image [B, 3, h*2^layers, w*2^layers].
    AEModel.synthesize(features) drives it (AE_model.py:209-210)."""

    def __init__(self, ngf=64, img_f=256, layers=3, norm='batch', activation='ReLU', use_spect=True, use_coord=False,
                 output_nc=3, num_blocks=3):
        super(DECGenerator1, self).__init__()
        print("DECGenerator1 init")
        self.layers = layers
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        mult = 4
        self.feature_block = FeatureAdaptBlock1(2048, ngf * mult, norm_layer, nonlinearity)
        self.num_blocks = num_blocks
        for i in range(num_blocks):
            setattr(self, 'mblock' + str(i), ResBlock(ngf * mult, ngf * mult, norm_layer=norm_layer, nonlinearity=nonlinearity,
                                                      use_spect=use_spect, use_coord=use_coord))
        for i in range(self.layers):
            mult_prev = mult
            mult = min(2 ** (self.layers - i - 2), img_f // ngf) if i != self.layers - 1 else 1
            setattr(self, 'decoder' + str(i), ResBlockDecoder(ngf * mult_prev, ngf * mult, ngf * mult, norm_layer, nonlinearity,
                                                              use_spect, use_coord))
        self.outconv = Output(ngf, output_nc, 3, None, nonlinearity, use_spect, use_coord)

    def _chain(self):
        return ([self.feature_block] + [getattr(self, 'mblock' + str(i)) for i in range(self.num_blocks)] +
                [getattr(self, 'decoder' + str(i)) for i in range(self.layers)] + [self.outconv])

    def tf(self, tape, feature):
        for m in self._chain():
            feature = m.tf(tape, feature)
        return feature

    def tb(self, tape, dy, need_dx=True):
        mods = self._chain()
        for i in range(len(mods) - 1, -1, -1):
            dy = mods[i].tb(tape, dy, need_dx=(need_dx or i > 0))
        return dy


class DECGenerator(RGModule):
    """The earlier decoder-only generator (networks.py:356-398; `define_G` keeps it commented out, networks.py:23): one ResBlock
    img_f -> 4*ngf straight on the feature map, then the same decoders and Output as DECGenerator1."""

    def __init__(self, ngf=64, img_f=2048, layers=3, norm='batch', activation='ReLU', use_spect=True, use_coord=False,
                 output_nc=3, num_blocks=3):
        super(DECGenerator, self).__init__()
        self.layers = layers
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        mult = 4
        self.resblock = ResBlock(img_f, ngf * mult, norm_layer=norm_layer, nonlinearity=nonlinearity, use_spect=use_spect,
                                 use_coord=use_coord)
        for i in range(self.layers):
            mult_prev = mult
            mult = min(2 ** (self.layers - i - 2), img_f // ngf) if i != self.layers - 1 else 1
            setattr(self, 'decoder' + str(i), ResBlockDecoder(ngf * mult_prev, ngf * mult, ngf * mult, norm_layer, nonlinearity,
                                                              use_spect, use_coord))
        self.outconv = Output(ngf, output_nc, 3, None, nonlinearity, use_spect, use_coord)

    def _chain(self):
        return [self.resblock] + [getattr(self, 'decoder' + str(i)) for i in range(self.layers)] + [self.outconv]

    def tf(self, tape, inputs):
        for m in self._chain():
            inputs = m.tf(tape, inputs)
        return inputs

    def tb(self, tape, dy, need_dx=True):
        mods = self._chain()
        for i in range(len(mods) - 1, -1, -1):
            dy = mods[i].tb(tape, dy, need_dx=(need_dx or i > 0))
        return dy


class FDGenerator(RGModule):
    """The FD-GAN decoder under the dual_gan options (networks.py:449-538; `--model_gen FD` builds it with fuse_mode='add',
    noise_nc=512): ReID feature vector (+ noise) -> W_reid / W_noise -> ReLU -> ConvT (8, 4) -> norm -> four ConvT 4x4 / 2 blocks ->
    ConvT -> tanh, [B, 3, 256, 128].  forward(reid_feature, noise=None): the 'add' / 'cat' modes need the noise (the reference's own
    callers pass none and fail, see define_G)."""

    def __init__(self, reid_feature_nc, ngf=64, noise_nc=3, pose_nc=18, output_nc=3, dropout=0.0, norm_layer=rnn.BatchNorm2d,
                 fuse_mode='none'):
        super(FDGenerator, self).__init__()
        self.fuse_mode = fuse_mode
        self.norm_layer = norm_layer
        self.dropout = dropout
        if type(norm_layer) == functools.partial:
            self.use_bias = norm_layer.func == rnn.InstanceNorm2d
        else:
            self.use_bias = norm_layer == rnn.InstanceNorm2d
        input_channel = [8, 8, 4, 2, 1]
        if fuse_mode == 'cat':
            nc = reid_feature_nc + noise_nc
        elif fuse_mode == 'add':
            nc = max(reid_feature_nc, noise_nc)
            self.W_reid = rnn.Linear(reid_feature_nc, nc, bias=False)
            self.W_noise = rnn.Linear(noise_nc, nc, bias=False)
        elif fuse_mode == 'none':
            nc = reid_feature_nc
            self.W_reid = rnn.Linear(reid_feature_nc, nc, bias=False)
        else:
            raise TypeError('Wrong fuse mode, please select from [cat|add]')       # the reference's `raise (str)` is a TypeError too
        self.de_avg = rnn.Sequential(rnn.ReLU(True), rnn.ConvTranspose2d(nc, ngf * 8, kernel_size=(8, 4), bias=self.use_bias),
                                     norm_layer(ngf * 8), rnn.Dropout(dropout))
        self.de_conv5 = self._make_layer_decode(ngf * input_channel[0], ngf * 8)
        self.de_conv4 = self._make_layer_decode(ngf * input_channel[1], ngf * 4)
        self.de_conv3 = self._make_layer_decode(ngf * input_channel[2], ngf * 2)
        self.de_conv2 = self._make_layer_decode(ngf * input_channel[3], ngf)
        self.de_conv1 = rnn.Sequential(rnn.ReLU(True),
                                       rnn.ConvTranspose2d(ngf * input_channel[4], output_nc, kernel_size=4, stride=2, padding=1,
                                                           bias=self.use_bias),
                                       rnn.Tanh())

    def _make_layer_decode(self, in_nc, out_nc):
        return rnn.Sequential(rnn.ReLU(True),
                              rnn.ConvTranspose2d(in_nc, out_nc, kernel_size=4, stride=2, padding=1, bias=self.use_bias),
                              self.norm_layer(out_nc), rnn.Dropout(self.dropout))

    def forward(self, reid_feature, noise=None):
        if noise is None:
            if self.fuse_mode != 'none':
                raise AttributeError("FDGenerator(fuse_mode=%r) needs the noise input ('NoneType' object has no attribute 'view' in "
                                     "the reference, networks.py:522-527)" % self.fuse_mode)
            return super(FDGenerator, self).forward(reid_feature)
        return super(FDGenerator, self).forward(reid_feature, noise)

    def tf(self, tape, reid_feature, noise=None):
        B = reid_feature.shape[0]
        if self.fuse_mode == 'cat':
            feature = ops.cat_channels([reid_feature.reshape(B, -1, 1, 1), noise.reshape(B, -1, 1, 1)])
        else:
            f = self.W_reid.tf(tape, reid_feature.reshape(B, -1))
            if self.fuse_mode == 'add':
                f = ops.add(f, self.W_noise.tf(tape, noise.reshape(B, -1)))
            feature = f.view(B, -1, 1, 1)
        r = ops.act_fwd(feature, ACT_RELU)
        tape.push((r, tuple(reid_feature.shape), None if noise is None else tuple(noise.shape)))
        x = self.de_avg[1].tf(tape, r)
        norm, drop = self.de_avg[2], self.de_avg[3]
        blocks = (self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2, self.de_conv1)
        for i, blk in enumerate(blocks):
            x = drop.tf(tape, norm.tf(tape, x, act=ACT_RELU))       # the norm of the block before + this block's ReLU in one pass
            if i < 4:
                x = blk[1].tf(tape, x)
                norm, drop = blk[2], blk[3]
            else:
                x = blk[1].tf(tape, x, act=ACT_TANH)
        return x

    def tb(self, tape, dy, need_dx=True):
        blocks = (self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2, self.de_conv1)
        norms = (self.de_avg, self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2)
        d = dy
        for i in range(4, -1, -1):
            d = blocks[i][1].tb(tape, d)
            d = norms[i][2].tb(tape, norms[i][3].tb(tape, d))
        d = self.de_avg[1].tb(tape, d)
        r, reid_shape, noise_shape = tape.pop()
        d = ops.act_bwd(d, r, ACT_RELU)
        B = d.shape[0]
        if self.fuse_mode == 'cat':
            c_reid = 1
            for v in reid_shape[1:]:
                c_reid *= v
            d_reid = ops.slice_channels(d, 0, c_reid).reshape(reid_shape)
            d_noise = ops.slice_channels(d, c_reid, d.shape[1]).reshape(noise_shape)
            return d_reid, d_noise
        d2 = d.reshape(B, -1)
        d_noise = None
        if self.fuse_mode == 'add':
            d_noise = self.W_noise.tb(tape, d2).reshape(noise_shape)
        d_reid = self.W_reid.tb(tape, d2).reshape(reid_shape)
        return d_reid, d_noise


class PoseGenerator1(RGModule):
    """Pose encoder -> PCTM(pose tokens attend to the adapted ReID feature map) -> residual decoder with skips
    (networks.py:639-738).  forward(reid_f [B, 2048, h, w], source_pose [B, pose_nc, H, W]) -> image [B, 3, H, W]."""

    def __init__(self, ngf=64, pose_nc=18, img_f=256, layers=3, norm='batch', activation='ReLU', use_spect=True,
                 use_coord=False, output_nc=3, affine=True, nhead=2, num_CABs=2, num_TTBs=2):
        super(PoseGenerator1, self).__init__()
        self.layers = layers
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        input_nc = pose_nc

        self.block0 = EncoderBlockOptimized(input_nc, ngf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(self.layers - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ngf)
            block = EncoderBlock(ngf * mult_prev, ngf * mult, norm_layer, nonlinearity, use_spect, use_coord)
            setattr(self, 'encoder' + str(i), block)

        self.feature_block = FeatureAdaptBlock1(2048, ngf * mult, norm_layer, nonlinearity)

        self.PCTM = PCTM(d_model=ngf * mult, nhead=nhead, num_CABs=num_CABs, num_TTBs=num_TTBs,
                         dim_feedforward=ngf * mult, activation="LeakyReLU", affine=affine, norm=norm)

        for i in range(self.layers):
            mult_prev = mult
            mult = min(2 ** (self.layers - i - 2), img_f // ngf) if i != self.layers - 1 else 1
            up = ResBlockDecoder(ngf * mult_prev, ngf * mult, ngf * mult, norm_layer, nonlinearity, use_spect, use_coord)
            setattr(self, 'decoder' + str(i), up)

        self.outconv = Output(ngf, output_nc, 3, None, nonlinearity, use_spect, use_coord)

    def tf(self, tape, reid_f, source_pose):
        F_p = self.block0.tf(tape, source_pose)
        skips = []
        for i in range(self.layers - 1):
            skips.append(F_p)
            F_p = getattr(self, 'encoder' + str(i)).tf(tape, F_p)
        F_id = self.feature_block.tf(tape, reid_f)
        F_g = self.PCTM.tf(tape, F_p, F_id)
        for i in range(self.layers):
            extra = skips.pop() if i < self.layers - 1 else None
            F_g = getattr(self, 'decoder' + str(i)).tf(tape, F_g, extra=extra)
        return self.outconv.tf(tape, F_g)

    def tb(self, tape, dy, need_dx=True):
        d = self.outconv.tb(tape, dy)
        dskips = []
        for i in range(self.layers - 1, -1, -1):
            out = getattr(self, 'decoder' + str(i)).tb(tape, d)
            if isinstance(out, tuple):
                d, ds = out
                dskips.append(ds)            # gradient of the skip popped at decoder i (= skips[layers - 2 - i])
            else:
                d = out
        d_fp, d_fid = self.PCTM.tb(tape, d)
        need_f = tape.needs_input is None or bool(tape.needs_input[0])
        d_reid = self.feature_block.tb(tape, d_fid, need_dx=need_f)
        # dskips was filled from the last decoder backwards: dskips[j] belongs to skips[j] (skip j = input of encoder j)
        for i in range(self.layers - 2, -1, -1):
            d_fp = getattr(self, 'encoder' + str(i)).tb(tape, d_fp)
            d_fp = ops.axpby(d_fp, dskips[i], 1.0, 1.0, out=d_fp)
        need_p = tape.needs_input is not None and len(tape.needs_input) > 1 and bool(tape.needs_input[1])
        d_pose = self.block0.tb(tape, d_fp, need_dx=need_p)
        return d_reid, d_pose


##############################################################################
# Discriminator
##############################################################################
class ResDiscriminator(RGModule):
    """ResNet discriminator with spectral norm on every conv (networks.py:917-955)."""

    def __init__(self, input_nc=3, ndf=64, img_f=1024, layers=3, norm='none', activation='LeakyReLU', use_spect=True,
                 use_coord=False):
        super(ResDiscriminator, self).__init__()
        self.layers = layers
        norm_layer = get_norm_layer(norm_type=norm)
        nonlinearity = get_nonlinearity_layer(activation_type=activation)
        self.nonlinearity = nonlinearity

        self.block0 = ResBlockEncoderOptimized(input_nc, ndf, ndf, norm_layer, nonlinearity, use_spect, use_coord)
        mult = 1
        for i in range(layers - 1):
            mult_prev = mult
            mult = min(2 ** (i + 1), img_f // ndf)
            block = ResBlockEncoder(ndf * mult_prev, ndf * mult, ndf * mult_prev, norm_layer, nonlinearity, use_spect, use_coord)
            setattr(self, 'encoder' + str(i), block)
        self.conv = SpectralNorm(rnn.Conv2d(ndf * mult, 1, 1))

    def _sn_convs(self):
        c = self.__dict__.get("_sn_list")
        if c is None:
            c = self.__dict__["_sn_list"] = [m for m in self.modules() if isinstance(m, rnn.SNConv2d)]
        return c

    def tf(self, tape, x):
        act, slope = _slope(self.nonlinearity)
        rnn.sn_prepare(self._sn_convs(), self.training)     # every filter of this forward: one power-iteration launch
        out = self.block0.tf(tape, x)
        for i in range(self.layers - 1):
            out = getattr(self, 'encoder' + str(i)).tf(tape, out)
        a = ops.act_fwd(out, act, slope)
        tape.push(a)
        return self.conv.tf(tape, a)

    def tb(self, tape, dy, need_dx=True):
        act, slope = _slope(self.nonlinearity)
        d = self.conv.tb(tape, dy)
        a = tape.pop()
        d = ops.act_bwd(d, a, act, slope)
        for i in range(self.layers - 2, -1, -1):
            d = getattr(self, 'encoder' + str(i)).tb(tape, d)
        need = tape.needs_input is None or bool(tape.needs_input[0])
        return self.block0.tb(tape, d, need_dx=need)

    # ---- WGAN-GP: input gradient, tangent pass and second-order weight gradients (external_function.cal_gradient_penalty) ----
    def _gp_blocks(self):
        if any(isinstance(m, (rnn.BatchNorm2d, rnn.InstanceNorm2d)) for m in self.modules()):
            raise NotImplementedError("gradient penalty: built for the norm='none' discriminator define_D constructs")
        blocks = [(None, self.block0.model[0], self.block0.model[2], self.block0.shortcut[1])]
        for i in range(self.layers - 1):
            b = getattr(self, 'encoder' + str(i))
            blocks.append((True, b.model[1], b.model[3], b.shortcut[1]))
        return blocks

    @staticmethod
    def _gp_weight(conv, training):
        """effective filter of one convolution for this forward: (w, w_krsc, spectral-norm record or None)"""
        if isinstance(conv, rnn.SNConv2d):
            w_sn, sigma, u, v = ops.spectral_norm_fwd(conv.weight_orig.detach(), conv.weight_u, conv.weight_v, training, conv.eps,
                                                      save_uv=True)
            conv.weight = w_sn
            rec = (sigma, u, v)
            w = w_sn
        else:
            w, rec = conv.weight.detach(), None
        wk = ops.weights_to_krsc(w) if (w.shape[2] * w.shape[3] > 1 and w.shape[1] % 4 == 0) else None
        return w, wk, rec

    def gp_forward(self, tape, x):
        """D(x) (one more power iteration in training mode, as every reference forward), then the adjoint pass with ones at
        the output: returns g = d sum(D(x)) / dx and, for the second-order step, every convolution's effective filter,
        LeakyReLU output and output adjoint."""
        act, slope = _slope(self.nonlinearity)
        recs, h = [], x
        for pre_act, c1, c2, cb in self._gp_blocks():
            W1, W2, Wb = (self._gp_weight(c, self.training) for c in (c1, c2, cb))
            a = ops.act_fwd(h, act, slope) if pre_act else None
            h1 = ops.conv2d_fwd(a if pre_act else h, W1[0], c1.stride, c1.padding, shift=c1.bias, act=act, slope=slope, w_krsc=W1[1])
            m = ops.conv2d_fwd(h1, W2[0], c2.stride, c2.padding, shift=c2.bias, w_krsc=W2[1])
            out = ops.conv2d_fwd(ops.avgpool2d_fwd(h, 2), Wb[0], cb.stride, cb.padding, shift=cb.bias, residual=m, w_krsc=Wb[1])
            recs.append(dict(c1=c1, c2=c2, cb=cb, W1=W1, W2=W2, Wb=Wb, a=a, h1=h1, x_shape=h.shape))
            h = out
        Wc = self._gp_weight(self.conv, self.training)
        a_f = ops.act_fwd(h, act, slope)
        y = ops.conv2d_fwd(a_f, Wc[0], self.conv.stride, self.conv.padding, shift=self.conv.bias, w_krsc=Wc[1])
        # adjoint pass
        ones = ops.fill_(torch.empty_like(y), 1.0)
        d = ops.act_bwd(ops.conv2d_dgrad(ones, Wc[0], a_f.shape[2:], self.conv.stride, self.conv.padding, w_krsc=Wc[1]), a_f, act, slope)
        for r in reversed(recs):
            c1, c2, cb = r['c1'], r['c2'], r['cb']
            r['d_out'] = d
            hw = r['x_shape'][2:]
            dxs = ops.avgpool2d_bwd(ops.conv2d_dgrad(d, r['Wb'][0], (hw[0] // 2, hw[1] // 2), cb.stride, cb.padding, w_krsc=r['Wb'][1]),
                                    tuple(r['x_shape']), 2)
            d1 = ops.act_bwd(ops.conv2d_dgrad(d, r['W2'][0], r['h1'].shape[2:], c2.stride, c2.padding, w_krsc=r['W2'][1]),
                             r['h1'], act, slope)
            r['d_1'] = d1
            din = ops.conv2d_dgrad(d1, r['W1'][0], hw, c1.stride, c1.padding, w_krsc=r['W1'][1])
            if r['a'] is not None:
                din = ops.act_bwd(din, r['a'], act, slope)
            d = ops.axpby(din, dxs, 1.0, 1.0, out=din)
        return d, dict(recs=recs, Wc=Wc, a_f=a_f, ones=ones)

    def gp_backward(self, tape, adj, v, g_pen):
        """{id(param): gradient} of g_pen * <v, g(W)>: tangent pass of v through the masked-linear network, then
        wgrad(input tangent, output adjoint) per convolution, mapped through W / sigma for spectral-normed filters."""
        act, slope = _slope(self.nonlinearity)
        t = ops.weighted_sum_bwd(g_pen, v.reshape(-1).contiguous(), v.numel(), 1.0, v.device).view(v.shape)
        grads = {}

        def wgrad(conv, W, t_in, d_out):
            dw = ops.conv2d_wgrad(t_in.contiguous(), d_out.contiguous(), W[0].shape, conv.stride, conv.padding)
            if W[2] is not None:
                sigma, u, vv = W[2]
                p = conv.weight_orig
                dw = ops.spectral_norm_bwd(dw, W[0], u, vv, sigma)
            else:
                p = conv.weight
            if p.requires_grad:
                grads[id(p)] = dw

        for r in adj['recs']:
            c1, c2, cb = r['c1'], r['c2'], r['cb']
            ta = ops.act_bwd(t, r['a'], act, slope) if r['a'] is not None else t
            wgrad(c1, r['W1'], ta, r['d_1'])
            t1 = ops.act_bwd(ops.conv2d_fwd(ta, r['W1'][0], c1.stride, c1.padding, w_krsc=r['W1'][1]), r['h1'], act, slope)
            wgrad(c2, r['W2'], t1, r['d_out'])
            tm = ops.conv2d_fwd(t1, r['W2'][0], c2.stride, c2.padding, w_krsc=r['W2'][1])
            tp = ops.avgpool2d_fwd(t, 2)
            wgrad(cb, r['Wb'], tp, r['d_out'])
            t = ops.conv2d_fwd(tp, r['Wb'][0], cb.stride, cb.padding, residual=tm, w_krsc=r['Wb'][1])
        wgrad(self.conv, adj['Wc'], ops.act_bwd(t, adj['a_f'], act, slope), adj['ones'])
        return grads
