"""DPTNModel — the dual-task pose transformer GAN object (BASELINE config 5) on the HIP runtime.

Restates the training side of CC/dual_gan/models/DPTN_model.py:13-240: same constructor (`DPTNModel(opt)`), options
(`modify_options` :18-42), attributes (net_G / net_D, optimizer_G / optimizer_D, loss_* names :48, visual names :50) and
methods (set_input :117-133, forward :135-137, synthesize :139-144, synthesize_pair :146-155, backward_D_basic :159-174,
backward_D :176-182, backward_G_basic :184-201, backward_G :203-214, optimize_parameters :216-225,
optimize_parameters_generated :227-234).  `DPTNGenerator` runs its source->source and source->target branches as one
stacked batch through the shared encoder / decoder (dual_gan/models/networks.py); `net_D` sees the target branch only.

What differs from the reference is scheduling and three defects that are not reproduced:
  * the generator update treats net_D as a constant (its weight gradients, which the reference computes and discards at
    the next zero_grad, are not computed);
  * `--gan_mode lsgan` makes `loss_ad_gen_t` a per-pixel map in the reference (external_function.py:53-57 returns the
    un-reduced MSE for the generator form) and `G_loss.backward()` (:213) then raises "grad can be implicitly created
    only for scalar outputs": this class raises the same RuntimeError at the same call, before touching any gradient;
  * `VGGLoss()` downloads ImageNet weights in the reference's constructor (:77); here it is built only when the
    perceptual terms are on (`--no_vgg_loss` absent) and takes `--vgg_weights <local torchvision vgg19 state_dict>`;
    with `--no_vgg_loss` the content / style losses are reported as 0, as the run configuration of BASELINE config 5;
  * test-time image dumping (`test`, `save_results`), the `net_A` adaptor (`--use_adp`, needs torchvision's resize module
    graph) are outside the training step and not rebuilt.

`--conv_dtype fp8` switches every convolution of net_G and net_D to the fp8 MFMA family (csrc/conv_f8.hip: e4m3
activations and filters, e5m2 gradients, per-tensor scales, fp32 accumulation); the default is the fp32 family.
"""
from __future__ import absolute_import

import itertools
import os

import torch
from rg_hip.tape import backward as _backward

from rg_hip import functional as RF
from rg_hip import optim as roptim
from rg_hip.parallel import GradReducer
from rg_hip.tape import no_param_grad

from . import base_function, external_function, networks
from .AE_model import _weighted, _wrap_step
from .base_model import BaseModel


class DPTNModel(BaseModel):
    def name(self):
        return 'DPTNModel'

    # command-line surface of the reference object (DPTN_model.py:18-42): flag -> (argparse keywords); same names, types and
    # defaults, so option files and scripts written for the reference parse unchanged.  The last three are this build's.
    _FLAGS = {
        'init_type': dict(type=str, default='orthogonal', help='initial type'),
        'use_spect_g': dict(action='store_false', help='use spectual normalization in generator'),
        'use_spect_d': dict(action='store_false', help='use spectual normalization in generator'),
        'use_coord': dict(action='store_true', help='use coordconv'),
        'lambda_style': dict(type=float, default=500, help='weight for the VGG19 style loss'),
        'lambda_content': dict(type=float, default=0.5, help='weight for the VGG19 content loss'),
        'layers_g': dict(type=int, default=3, help='number of layers in G'),
        'save_input': dict(action='store_true', help='whether save the input images when testing'),
        'num_blocks': dict(type=int, default=3, help='number of resblocks'),
        'affine': dict(action='store_true', default=True, help='affine in PTM'),
        'nhead': dict(type=int, default=2, help='number of heads in PTM'),
        'num_CABs': dict(type=int, default=2, help='number of CABs in PTM'),
        'num_TTBs': dict(type=int, default=2, help='number of CABs in PTM'),
        'ratio_g2d': dict(type=float, default=0.1, help='learning rate ratio G to D'),
        'lambda_rec': dict(type=float, default=2.0, help='weight for image reconstruction loss'),
        'lambda_g': dict(type=float, default=5.0, help='weight for generation loss'),
        't_s_ratio': dict(type=float, default=0.8, help='loss ratio between dual tasks'),
        'dis_layers': dict(type=int, default=3, help='number of layers in D'),
        'vgg_weights': dict(type=str, default='', help='local torchvision vgg19 state_dict (.pth)'),
        'conv_dtype': dict(type=str, default='fp32', help='fp32 | fp8: MFMA family of the convolutions'),
        'f8_scaling': dict(type=str, default='delayed', help='delayed | jit: per-tensor fp8 scale policy'),
    }
    _DEFAULT_OVERRIDES = dict(use_spect_g=False, use_spect_d=True)

    @staticmethod
    def modify_options(parser, is_train=True):
        for flag, kw in DPTNModel._FLAGS.items():
            parser.add_argument('--' + flag, **kw)
        parser.set_defaults(**DPTNModel._DEFAULT_OVERRIDES)
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.old_size = getattr(opt, 'old_size', None)
        self.t_s_ratio = opt.t_s_ratio
        self.loss_names = ['app_gen_s', 'content_gen_s', 'style_gen_s', 'app_gen_t', 'ad_gen_t', 'dis_img_gen_t', 'content_gen_t',
                           'style_gen_t']
        self.model_names = ['G']
        self.visual_names = ['source_image', 'source_pose', 'target_image', 'target_pose', 'fake_image_s', 'fake_image_t']
        self.device = torch.device("cuda", torch.cuda.current_device())

        self.net_G = networks.define_G(opt, image_nc=opt.image_nc, pose_nc=opt.pose_nc, ngf=64, img_f=512, encoder_layer=3,
                                       norm=opt.norm, activation='LeakyReLU', use_spect=opt.use_spect_g,
                                       use_coord=opt.use_coord, output_nc=3, num_blocks=3, affine=True, nhead=opt.nhead,
                                       num_CABs=opt.num_CABs, num_TTBs=opt.num_TTBs)
        # --use_adp (DPTN_model.py:56-59): the adaptor from synthesised 128x64 images to 256x128 ReID inputs; synthesize() /
        # synthesize_pair() pass their images through it, and without --gan_train it is the only network that trains (:91-105)
        self.use_adp = bool(getattr(opt, 'use_adp', False))
        if self.use_adp:
            self.model_names = ['G', 'A']
            self.net_A = networks.Resize_ReID(image_nc=opt.image_nc).to(self.device)

        if self.gan_train:
            self.model_names = ['G', 'D']
            self.net_D = networks.define_D(opt, ndf=32, img_f=128, layers=opt.dis_layers, use_spect=opt.use_spect_d)
        self.conv_dtype = getattr(opt, 'conv_dtype', 'fp32')
        if self.conv_dtype not in ('fp32', 'fp8'):
            raise ValueError("--conv_dtype %r: expected fp32 or fp8" % (self.conv_dtype,))
        self._f8_states = []
        if self.conv_dtype == 'fp8':
            from rg_hip import lowp
            policy = getattr(opt, 'f8_scaling', 'delayed')
            # the layers that touch raw images / pose maps and the 3-channel output layers stay fp32 (usual fp8 practice: they
            # hold < 2 % of the FLOPs and their quantisation noise goes straight into the image / the discriminator's verdict)
            keep_G = ("block0.model.0", "source_encoder.block0.model.0", "outconv.conv1")
            keep_D = ("block0.model.0", "block0.shortcut.1", "conv")
            self._f8_states.append(lowp.set_conv_dtype(self.net_G, 'fp8', policy=policy, keep_fp32=keep_G))
            if self.gan_train:
                self._f8_states.append(lowp.set_conv_dtype(self.net_D, 'fp8', policy=policy, keep_fp32=keep_D))

        # the small-kernel GAN networks run their forward / backward programs as captured single-stream hipGraphs (rg_hip.netgraph):
        # their steps are otherwise bound by the host's launch rate
        for _n in ('net_G', 'net_D'):
            _m = getattr(self, _n, None)
            _m = getattr(_m, "module", _m)             # through the DataParallel shim
            if _m is not None:
                _m.__dict__["_rg_graph"] = True
        if getattr(self.opt, 'verbose', False):
            print('---------- Networks initialized -------------')
        if self.gan_train:
            self.old_lr = opt.gan_lr
            self.GANloss = external_function.GANLoss(opt.gan_mode).to(self.device)
            self.no_vgg = bool(getattr(opt, 'no_vgg_loss', False))
            if not self.no_vgg:
                self.Vggloss = external_function.VGGLoss(vgg_weights=getattr(opt, 'vgg_weights', '') or None).to(self.device)
            self.optimizer_G = roptim.Adam(itertools.chain(filter(lambda p: p.requires_grad, self.net_G.parameters())),
                                           lr=opt.gan_lr, betas=(opt.beta1, 0.999))
            self.optimizers = [self.optimizer_G]
            self.optimizer_D = roptim.Adam(itertools.chain(filter(lambda p: p.requires_grad, self.net_D.parameters())),
                                           lr=opt.gan_lr * opt.ratio_g2d, betas=(opt.beta1, 0.999))
            self.optimizers.append(self.optimizer_D)
            self.schedulers = [base_function.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
            self._red_G = GradReducer(self.optimizer_G, modules=[self.net_G])
            self._red_D = GradReducer(self.optimizer_D, modules=[self.net_D])
            _wrap_step(self.optimizer_G, self._red_G)
            _wrap_step(self.optimizer_D, self._red_D)
            if self._f8_states:
                # delayed fp8 scaling: the maxima collected during a step become current at the step boundary that ALWAYS runs —
                # the generator's optimizer step, last in the reference's order (DPTN_model.py:216-225) — so a caller that drives
                # backward_D / backward_G / optimizer_X.step() itself (the joint trainers do) rolls the scales too
                inner_G = self.optimizer_G.step

                def step_and_roll(*a, **kw):
                    out = inner_G(*a, **kw)
                    for st in self._f8_states:
                        st.roll()
                    return out
                self.optimizer_G.step = step_and_roll
        elif self.use_adp:
            print("use adaptor")
            self.net_G.eval()                                   # only the adaptor trains
            self.old_lr = opt.gan_lr
            self.optimizer_A = roptim.Adam(itertools.chain(filter(lambda p: p.requires_grad, self.net_A.parameters())),
                                           lr=opt.gan_lr, betas=(opt.beta1, 0.999))
            self.schedulers = base_function.get_scheduler(self.optimizer_A, opt)
            self._red_A = GradReducer(self.optimizer_A, modules=[self.net_A])
            _wrap_step(self.optimizer_A, self._red_A)
        else:
            self.net_G.eval()

        if self.load_pretrain != "" or getattr(opt, 'continue_train', False):
            print('model loaded from pretrained')
            self.load_networks(opt.which_epoch)

    # ---- inputs / synthesis (DPTN_model.py:117-155) ---------------------------------------------------------
    def set_input(self, input, b_id=None):
        self.input = input
        if b_id is not None:
            source_image, source_pose = torch.index_select(input['Xs'], 0, b_id), torch.index_select(input['Ps'], 0, b_id)
            target_image, target_pose = torch.index_select(input['Xt'], 0, b_id), torch.index_select(input['Pt'], 0, b_id)
        else:
            source_image, source_pose = input['Xs'], input['Ps']
            target_image, target_pose = input['Xt'], input['Pt']
        dev = self.device
        self.source_image = source_image.to(dev, non_blocking=True).contiguous()
        self.source_pose = source_pose.to(dev, non_blocking=True).contiguous()
        self.target_image = target_image.to(dev, non_blocking=True).contiguous()
        self.target_pose = target_pose.to(dev, non_blocking=True).contiguous()
        self.image_paths = []
        if 'Xs_path' in input and 'Xt_path' in input:
            for i in range(self.source_image.size(0)):
                self.image_paths.append(os.path.splitext(input['Xs_path'][i])[0] + '_2_' + input['Xt_path'][i])

    def forward(self):
        self.fake_image_t, self.fake_image_s = self.net_G(self.source_image, self.source_pose, self.target_pose)

    def synthesize(self, is_tain=False):
        self.fake_image_t, self.fake_image_s = self.net_G(self.source_image, self.source_pose, self.target_pose, is_tain)
        if self.use_adp:
            # (with is_tain False the generator returns None for the source branch; the reference would fail inside my_resize)
            self.fake_image_t = self.net_A(self.fake_image_t)
            self.fake_image_s = self.net_A(self.fake_image_s) if self.fake_image_s is not None else None
        return self.fake_image_t, self.fake_image_s

    def synthesize_pair(self):
        self.fake_image_n, _ = self.net_G(torch.flip(self.source_image, dims=[0]).contiguous(),
                                          torch.flip(self.source_pose, dims=[0]).contiguous(), self.target_pose, False)
        if self.use_adp:
            self.fake_image_n = self.net_A(self.fake_image_n)
        return self.fake_image_n

    # ---- discriminator update (DPTN_model.py:159-182) --------------------------------------------------------
    def backward_D_basic(self, netD, real, fake):
        D_real = netD(real)
        D_real_loss = self.GANloss(D_real, True, True)
        D_fake = netD(fake.detach())
        D_fake_loss = self.GANloss(D_fake, False, True)
        terms = [(D_real_loss, 0.5), (D_fake_loss, 0.5)]
        if self.opt.gan_mode == 'wgangp':
            gradient_penalty, _ = external_function.cal_gradient_penalty(netD, real, fake.detach(),
                                                                         alpha=getattr(self, 'gp_alpha', None))
            terms.append((gradient_penalty, 1.0))
        return _weighted(terms)

    def backward_D(self):
        base_function._unfreeze(self.net_D)
        self.loss_dis_img_gen_t = self.backward_D_basic(self.net_D, self.target_image, self.fake_image_t)
        D_loss = self.loss_dis_img_gen_t
        _backward(D_loss)
        self.loss_dis_img_gen_t = D_loss.detach()

    # ---- generator update (DPTN_model.py:184-214) ----------------------------------------------------------------
    def backward_G_basic(self, fake_image, target_image, use_d):
        """(lambda_rec * L1, lambda_g * GAN loss or None, lambda_style * style, lambda_content * content), all attached 0-dim
        device tensors; with the perceptual terms off the last two are detached zeros."""
        loss_app_gen = _weighted([(RF.l1_loss(fake_image, target_image), self.opt.lambda_rec)])
        loss_ad_gen = None
        if use_d:
            with no_param_grad(self.net_D.module if hasattr(self.net_D, "module") else self.net_D):
                D_fake = self.net_D(fake_image)
            loss_ad_gen = self.GANloss(D_fake, True, False)
            if loss_ad_gen.dim() == 0:
                loss_ad_gen = _weighted([(loss_ad_gen, self.opt.lambda_g)])
        if self.no_vgg:
            zero = torch.zeros((), device=fake_image.device)
            return loss_app_gen, loss_ad_gen, zero, zero
        loss_content_gen, loss_style_gen = self.Vggloss(fake_image, target_image)
        loss_style_gen = _weighted([(loss_style_gen, self.opt.lambda_style)])
        loss_content_gen = _weighted([(loss_content_gen, self.opt.lambda_content)])
        return loss_app_gen, loss_ad_gen, loss_style_gen, loss_content_gen

    def backward_G(self, retain_graph=False):
        base_function._unfreeze(self.net_D)
        self.loss_app_gen_t, self.loss_ad_gen_t, self.loss_style_gen_t, self.loss_content_gen_t = \
            self.backward_G_basic(self.fake_image_t, self.target_image, use_d=True)
        self.loss_app_gen_s, self.loss_ad_gen_s, self.loss_style_gen_s, self.loss_content_gen_s = \
            self.backward_G_basic(self.fake_image_s, self.source_image, use_d=False)
        if self.loss_ad_gen_t.dim() != 0:
            # lsgan: the un-reduced generator form makes G_loss a map; the reference's G_loss.backward() (:213) raises this
            raise RuntimeError("grad can be implicitly created only for scalar outputs")
        r = self.t_s_ratio
        terms = [(self.loss_app_gen_t, r), (self.loss_app_gen_s, 1 - r), (self.loss_ad_gen_t, 1.0)]
        if not self.no_vgg:
            terms += [(self.loss_style_gen_t, r), (self.loss_content_gen_t, r), (self.loss_style_gen_s, 1 - r),
                      (self.loss_content_gen_s, 1 - r)]
        G_loss = _weighted(terms)
        _backward(G_loss, retain_graph=retain_graph)
        for n in ('app_gen_t', 'ad_gen_t', 'style_gen_t', 'content_gen_t', 'app_gen_s', 'style_gen_s', 'content_gen_s'):
            setattr(self, 'loss_' + n, getattr(self, 'loss_' + n).detach())

    def optimize_parameters(self):
        self.forward()
        self.optimize_parameters_generated()

    def optimize_parameters_generated(self):
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()

        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.step()             # (fp8: also rolls the delayed scales, see __init__)
