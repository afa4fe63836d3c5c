"""FD-GAN networks — restates FD-GAN-master/fdgan/networks.py on the HIP tape runtime.

Same public names and signatures (weights_init_normal, init_weights, get_norm_layer, get_scheduler,
print_network, remove_module_key, set_bn_fix, CustomPoseGenerator, NLayerDiscriminator) and the same
sub-module names / indices, so the reference's checkpoints (`en_conv2.1.weight`, `de_avg.1.weight`,
`model.0.weight`, ...) load unchanged.

Program-level fusion (exact): the reference's in-place LeakyReLU/ReLU at the head of every block
(networks.py:141-156) rewrites its input tensor, so every consumer sees the ACTIVATED tensor; the
programs below therefore apply that activation in the epilogue of the producing conv / norm kernel.
ReLU and Dropout commute exactly (mask and 1/(1-p) are non-negative), which lets the decoder's
norm -> dropout -> relu run as norm+relu -> dropout.
"""
from __future__ import absolute_import

import functools

from torch.nn import init
from torch.optim import lr_scheduler

from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip.ops import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_TANH
from rg_hip.tape import RGModule


def weights_init_normal(m):
    classname = m.__class__.__name__
    if classname.find('Conv') != -1:
        init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('Linear') != -1:
        init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find('BatchNorm2d') != -1:
        init.normal_(m.weight.data, 1.0, 0.02)
        init.constant_(m.bias.data, 0.0)


def init_weights(net):
    net.apply(weights_init_normal)


def get_norm_layer(norm_type='batch'):
    if norm_type == 'batch':
        norm_layer = functools.partial(rnn.BatchNorm2d, affine=True)
    elif norm_type == 'instance':
        norm_layer = functools.partial(rnn.InstanceNorm2d, affine=False)
    elif norm_type == 'none':
        norm_layer = None
    else:
        raise NotImplementedError('normalization layer [%s] is not found' % norm_type)
    return norm_layer


def get_scheduler(optimizer, opt):
    def lambda_rule(epoch):
        lr_l = 1.0 - max(0, epoch + 2 - opt.niter) / float(opt.niter_decay + 1)
        return lr_l
    scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    return scheduler


def print_network(net):
    num_params = 0
    for param in net.parameters():
        num_params += param.numel()
    print(net)
    print('Total number of parameters: %d' % num_params)


def remove_module_key(state_dict):
    for key in list(state_dict.keys()):
        if 'module' in key:
            state_dict[key.replace('module.', '')] = state_dict.pop(key)
    return state_dict


def set_bn_fix(m):
    classname = m.__class__.__name__
    if classname.find('BatchNorm') != -1:
        m.eval()


def _is_instance_norm(norm_layer):
    if type(norm_layer) == functools.partial:
        return norm_layer.func == rnn.InstanceNorm2d
    return norm_layer == rnn.InstanceNorm2d


class CustomPoseGenerator(RGModule):
    def __init__(self, pose_feature_nc, reid_feature_nc, noise_nc, pose_nc=18, output_nc=3,
                 dropout=0.0, norm_layer=rnn.BatchNorm2d, fuse_mode='cat', connect_layers=0):
        super(CustomPoseGenerator, self).__init__()
        assert (connect_layers >= 0 and connect_layers <= 5)
        ngf = 64
        self.connect_layers = connect_layers
        self.fuse_mode = fuse_mode
        self.norm_layer = norm_layer
        self.dropout = dropout
        self.use_bias = _is_instance_norm(norm_layer)

        input_channel = [[8, 8, 4, 2, 1],
                         [16, 8, 4, 2, 1],
                         [16, 16, 4, 2, 1],
                         [16, 16, 8, 2, 1],
                         [16, 16, 8, 4, 1],
                         [16, 16, 8, 4, 2]]

        # ---------------- encoder ----------------
        self.en_conv1 = rnn.Conv2d(pose_nc, ngf, kernel_size=4, stride=2, padding=1, bias=self.use_bias)
        self.en_conv2 = self._make_layer_encode(ngf, ngf * 2)
        self.en_conv3 = self._make_layer_encode(ngf * 2, ngf * 4)
        self.en_conv4 = self._make_layer_encode(ngf * 4, ngf * 8)
        self.en_conv5 = self._make_layer_encode(ngf * 8, ngf * 8)
        self.en_avg = rnn.Sequential(rnn.LeakyReLU(0.2, True),
                                     rnn.Conv2d(ngf * 8, pose_feature_nc, kernel_size=(8, 4), bias=self.use_bias),
                                     norm_layer(pose_feature_nc))
        # ---------------- decoder ----------------
        if fuse_mode == 'cat':
            de_in = pose_feature_nc + reid_feature_nc + noise_nc
        elif fuse_mode == 'add':
            nc = max(pose_feature_nc, reid_feature_nc, noise_nc)
            self.W_pose = rnn.Linear(pose_feature_nc, nc, bias=False)
            self.W_reid = rnn.Linear(reid_feature_nc, nc, bias=False)
            self.W_noise = rnn.Linear(noise_nc, nc, bias=False)
            de_in = nc
        else:
            raise ('Wrong fuse mode, please select from [cat|add]')
        self.de_avg = rnn.Sequential(rnn.ReLU(True),
                                     rnn.ConvTranspose2d(de_in, ngf * 8, kernel_size=(8, 4), bias=self.use_bias),
                                     norm_layer(ngf * 8),
                                     rnn.Dropout(dropout))
        self.de_conv5 = self._make_layer_decode(ngf * input_channel[connect_layers][0], ngf * 8)
        self.de_conv4 = self._make_layer_decode(ngf * input_channel[connect_layers][1], ngf * 4)
        self.de_conv3 = self._make_layer_decode(ngf * input_channel[connect_layers][2], ngf * 2)
        self.de_conv2 = self._make_layer_decode(ngf * input_channel[connect_layers][3], ngf)
        self.de_conv1 = rnn.Sequential(rnn.ReLU(True),
                                       rnn.ConvTranspose2d(ngf * input_channel[connect_layers][4], output_nc,
                                                           kernel_size=4, stride=2, padding=1, bias=self.use_bias),
                                       rnn.Tanh())

    def _make_layer_encode(self, in_nc, out_nc):
        return rnn.Sequential(rnn.LeakyReLU(0.2, True),
                              rnn.Conv2d(in_nc, out_nc, kernel_size=4, stride=2, padding=1, bias=self.use_bias),
                              self.norm_layer(out_nc))

    def _make_layer_decode(self, in_nc, out_nc):
        return rnn.Sequential(rnn.ReLU(True),
                              rnn.ConvTranspose2d(in_nc, out_nc, kernel_size=4, stride=2, padding=1, bias=self.use_bias),
                              self.norm_layer(out_nc),
                              rnn.Dropout(self.dropout))

    # ---- tape program (reference forward: networks.py:164-192) -------------------------------------
    def tf(self, tape, posemap, reid_feature, noise):
        B = posemap.shape[0]
        a = [None] * 6                       # a[k] = LeakyReLU(pose_feature_k), what every consumer sees
        a[1] = self.en_conv1.tf(tape, posemap, act=ACT_LEAKY, slope=0.2)
        for k, blk in ((2, self.en_conv2), (3, self.en_conv3), (4, self.en_conv4), (5, self.en_conv5)):
            a[k] = blk[2].tf(tape, blk[1].tf(tape, a[k - 1]), act=ACT_LEAKY, slope=0.2)
        pose_feature = self.en_avg[2].tf(tape, self.en_avg[1].tf(tape, a[5]))

        reid_feature = reid_feature.reshape(B, -1, 1, 1)
        noise = noise.reshape(B, -1, 1, 1)
        if self.fuse_mode == 'cat':
            feature = ops.cat_channels([reid_feature, pose_feature, noise])
            tape.push((reid_feature.shape[1], pose_feature.shape[1], noise.shape[1]))
        else:
            f = self.W_reid.tf(tape, reid_feature.reshape(B, -1))
            f = ops.add(f, self.W_pose.tf(tape, pose_feature.reshape(B, -1)))
            f = ops.add(f, self.W_noise.tf(tape, noise.reshape(B, -1)))
            feature = f.view(B, -1, 1, 1)
        r = ops.act_fwd(feature, ACT_RELU)                      # de_avg[0], in place in the reference
        tape.push(r)
        x = self.de_avg[1].tf(tape, r)
        cn = self.connect_layers
        skips = (a[5], a[4], a[3], a[2], a[1])
        blocks = (self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2, self.de_conv1)
        norm, drop = self.de_avg[2], self.de_avg[3]
        for i, blk in enumerate(blocks):
            # finish the previous block: norm (+ReLU of this block when nothing is concatenated) -> dropout
            if cn > 0:
                x = drop.tf(tape, norm.tf(tape, x))
                c = ops.cat_channels([x, skips[i]])
                x = ops.act_fwd(c, ACT_RELU)
                tape.push((x, c.shape[1] - skips[i].shape[1]))
                cn -= 1
                tape.push(True)
            else:
                x = drop.tf(tape, norm.tf(tape, x, act=ACT_RELU))
                tape.push(False)
            if i < 4:
                x = blk[1].tf(tape, x)
                norm, drop = blk[2], blk[3]
            else:
                x = blk[1].tf(tape, x, act=ACT_TANH)            # de_conv1: ConvT -> Tanh
        tape.push(a)
        return x

    def tb(self, tape, dy, need_dx=True):
        a = tape.pop()
        da = [None] * 6                                         # gradients flowing into the skip tensors
        blocks = (self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2, self.de_conv1)
        norms = (self.de_avg, self.de_conv5, self.de_conv4, self.de_conv3, self.de_conv2)
        d = dy
        for i in range(4, -1, -1):
            d = blocks[i][1].tb(tape, d)
            had_cat = tape.pop()
            nrm, drp = norms[i][2], norms[i][3]
            if had_cat:
                xr, c_main = tape.pop()
                dc = ops.act_bwd(d, xr, ACT_RELU)
                k = 5 - i
                da[k] = ops.slice_channels(dc, c_main, dc.shape[1])
                d = ops.slice_channels(dc, 0, c_main)
            d = nrm.tb(tape, drp.tb(tape, d))
        d = self.de_avg[1].tb(tape, d)
        r = tape.pop()
        d = ops.act_bwd(d, r, ACT_RELU)
        B = d.shape[0]
        if self.fuse_mode == 'cat':
            c_reid, c_pose, c_noise = tape.pop()
            d_reid = ops.slice_channels(d, 0, c_reid)
            d_pose = ops.slice_channels(d, c_reid, c_reid + c_pose)
            d_noise = ops.slice_channels(d, c_reid + c_pose, c_reid + c_pose + c_noise)
        else:
            d2 = d.reshape(B, -1)
            d_noise = self.W_noise.tb(tape, d2).view(B, -1, 1, 1)
            d_pose = self.W_pose.tb(tape, d2).view(B, -1, 1, 1)
            d_reid = self.W_reid.tb(tape, d2).view(B, -1, 1, 1)
        g = self.en_avg[1].tb(tape, self.en_avg[2].tb(tape, d_pose))
        for k, blk in ((5, self.en_conv5), (4, self.en_conv4), (3, self.en_conv3), (2, self.en_conv2)):
            if da[k] is not None:
                g = ops.add(g, da[k])
            g = blk[1].tb(tape, blk[2].tb(tape, g))
        if da[1] is not None:
            g = ops.add(g, da[1])
        # the pose map is data in every step of the reference (FD/fdgan/model.py:110-112): its gradient — the data gradient of
        # the 18 -> 64 first layer at full resolution, 0.32 ms per step at 32 crops — is only formed when somebody asks for it
        ni = tape.needs_input
        need_pose = need_dx and (ni is None or bool(ni[0]))
        d_posemap = self.en_conv1.tb(tape, g, need_dx=need_pose)
        return d_posemap, d_reid, d_noise


class NLayerDiscriminator(RGModule):
    def __init__(self, input_nc, norm_layer=rnn.BatchNorm2d):
        super(NLayerDiscriminator, self).__init__()
        ndf = 64
        n_layers = 3
        use_bias = _is_instance_norm(norm_layer)
        kw = 4
        padw = 1
        sequence = [rnn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw),
                    rnn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev = nf_mult
            nf_mult = min(2 ** n, 8)
            sequence += [rnn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw,
                                    bias=use_bias),
                         norm_layer(ndf * nf_mult),
                         rnn.LeakyReLU(0.2, True)]
        nf_mult_prev = nf_mult
        nf_mult = min(2 ** n_layers, 8)
        sequence += [rnn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw,
                                bias=use_bias),
                     norm_layer(ndf * nf_mult),
                     rnn.LeakyReLU(0.2, True)]
        sequence += [rnn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = rnn.Sequential(*sequence)

    def _groups(self):
        """(conv, norm|None, act_code, slope) groups of the Sequential (conv [-> norm] [-> LeakyReLU])."""
        mods = list(self.model)
        groups, i = [], 0
        while i < len(mods):
            conv = mods[i]
            i += 1
            norm = None
            if i < len(mods) and isinstance(mods[i], (rnn._BatchNorm, rnn.InstanceNorm2d)):
                norm = mods[i]
                i += 1
            act, slope = ACT_NONE, 0.0
            if i < len(mods) and isinstance(mods[i], rnn._Act):
                act, slope = mods[i].ACT, mods[i].SLOPE
                i += 1
            groups.append((conv, norm, act, slope))
        return groups

    def tf(self, tape, x, image=None):
        """forward(x) as the reference; forward(posemap, image) concatenates the two along channels inside the program
        (FD/fdgan/model.py:160-161,197) so that the backward can stop at the image channels."""
        split = None
        if image is not None:
            split = (x.shape[1], x.shape[1] + image.shape[1])
            x = ops.cat_channels([x, image])
        tape.push(split)
        for conv, norm, act, slope in self._groups():
            if norm is None:
                x = conv.tf(tape, x, act=act, slope=slope)
            else:
                x = norm.tf(tape, conv.tf(tape, x), act=act, slope=slope)
        return x

    def tb(self, tape, dy, need_dx=True):
        groups = self._groups()
        for gi in range(len(groups) - 1, 0, -1):
            conv, norm, _, _ = groups[gi]
            if norm is not None:
                dy = norm.tb(tape, dy)
            dy = conv.tb(tape, dy, need_dx=True)
        conv, norm, _, _ = groups[0]
        if norm is not None:
            dy = norm.tb(tape, dy)
        split = tape.stack[-2] if len(tape.stack) >= 2 else None          # pushed before the first conv's record
        if split is None:
            dx = conv.tb(tape, dy, need_dx=need_dx)
            tape.pop()
            return dx
        ni = tape.needs_input
        need_pose = ni is None or bool(ni[0])
        need_img = ni is None or len(ni) < 2 or bool(ni[1])
        c0, c1 = split
        if need_pose or not need_img:
            dx = conv.tb(tape, dy, need_dx=need_dx and (need_pose or need_img))
            tape.pop()
            if dx is None:
                return None, None
            return (ops.slice_channels(dx, 0, c0) if need_pose else None,
                    ops.slice_channels(dx, c0, c1) if need_img else None)
        # only the image receives a gradient (the generator update): data gradient of the image channels alone
        d_img = conv.tb(tape, dy, need_dx=need_dx, dx_channels=(c0, c1))
        tape.pop()
        return None, d_img
