"""AEModel — the GAN object of the joint ReID + GAN training step, on the HIP kernels.

Mirror of CC/dual_gan/models/AE_model.py:15-410: same constructor (`opt` namespace), attributes (`net_G`, `net_D`,
`optimizer_G`, `optimizer_D`, `loss_G`, `loss_D`, `fake_image`, `source_image`, `source_pose`) and methods used by the
trainers (`set_input`, `synthesize_p`, `get_loss_G`, `backward_D`, `backward_G`, `get_L1_loss`, `optimize_parameters`,
`optimize_generated`, `get_current_errors`, `save_networks` / `load_networks`, `update_learning_rate`).

Scheduling differences that do not change results: the loss terms `mean(|fake - src|) * lambda_rec` and
`mean((D(fake) - 1)^2) * lambda_g` are single fused reductions instead of element-wise maps followed by `.mean()`;
optimizers are the fused arena Adam (rg_hip.optim) and their gradient arenas are all-reduced over RCCL when
torch.distributed is initialised (the reference wraps the nets in nn.DataParallel, base_function.py:93-102).
Every D forward runs its own spectral-norm power iteration, in the reference's order (G-loss pass, then real, then fake).
Generators: 'Pose' (joint step) and 'AE' (GAN warm-up / feature mixing); 'DPTN' etc., `--bipath_gan`, `--use_adp` and the
VGG loss are not built yet.
"""
from __future__ import absolute_import

import itertools

import torch
from rg_hip.tape import backward as _backward

from rg_hip import functional as RF
from rg_hip import nn as rnn
from rg_hip import ops
from rg_hip import optim as roptim
from rg_hip.parallel import GradReducer
from . import base_function, external_function, networks
from .base_model import BaseModel


def _weighted_rows(terms):
    """sum_i w_i * v_i for per-sample vectors (two axpby launches at most; autograd through torch's own add / mul on [B] vectors)"""
    out = None
    for v, w in terms:
        t = v * float(w)
        out = t if out is None else out + t
    return out


def _weighted(terms):
    """sum_i w_i * loss_i of 0-dim device losses in one kernel (keeps the autograd graph)."""
    vals = torch.stack([t for t, _ in terms])
    return RF._WeightedSum.apply(vals, RF.const_vector(tuple(float(w) for _, w in terms), vals.device), 1.0)


class AEModel(BaseModel):
    def name(self):
        return 'AEModel'

    @staticmethod
    def modify_options(parser, is_train=True):
        """Add new options and rewrite default values for existing options (AE_model.py:20-56)"""
        parser.add_argument('--init_type', type=str, default='orthogonal', help='initial type')
        parser.add_argument('--use_spect_g', action='store_false', help='use spectual normalization in generator')
        parser.add_argument('--use_spect_d', action='store_false', help='use spectual normalization in generator')
        parser.add_argument('--use_coord', action='store_true', help='use coordconv')
        parser.add_argument('--lambda_style', type=float, default=500, help='weight for the VGG19 style loss')
        parser.add_argument('--lambda_content', type=float, default=0.5, help='weight for the VGG19 content loss')
        parser.add_argument('--layers_g', type=int, default=3, help='number of layers in G')
        parser.add_argument('--num_feats', type=int, default=256, help='number of layers in G')
        parser.add_argument('--save_input', action='store_true', help="whether save the input images when testing")
        parser.add_argument('--num_blocks', type=int, default=3, help="number of resblocks")
        parser.add_argument('--affine', action='store_true', default=True, help="affine in PTM")
        parser.add_argument('--nhead', type=int, default=2, help="number of heads in PTM")
        parser.add_argument('--num_CABs', type=int, default=2, help="number of CABs in PTM")
        parser.add_argument('--num_TTBs', type=int, default=2, help="number of CABs in PTM")
        parser.add_argument('--bipath_gan', action='store_true', help='bipath gan')
        parser.add_argument('--ratio_g2d', type=float, default=0.1, help='learning rate ratio G to D')
        parser.add_argument('--lambda_rec', type=float, default=2.0, help='weight for image reconstruction loss')
        parser.add_argument('--lambda_g', type=float, default=5.0, help='weight for generation loss')
        parser.add_argument('--lambda_fus', type=float, default=0.8, help='fusion ratio between samples')
        parser.add_argument('--dis_layers', type=int, default=3, help='number of layers in D')
        parser.set_defaults(use_spect_g=False)
        parser.set_defaults(use_spect_d=True)
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.loss_names = ['G', 'D']
        self.model_names = ['G']
        self.visual_names = ['source_image', 'source_pose', 'target_image', 'target_pose', 'fake_image', 'mixed_image']
        self.model_gen = opt.model_gen
        num_feats = opt.num_feats
        G_layer = opt.layers_g
        self.device = torch.device("cuda", torch.cuda.current_device())

        self.feat_bn = rnn.BatchNorm1d(num_feats).to(self.device)
        g_kw = dict(image_nc=opt.image_nc, pose_nc=opt.pose_nc, ngf=64, img_f=num_feats, encoder_layer=G_layer, norm=opt.norm,
                    activation='LeakyReLU', use_spect=opt.use_spect_g, use_coord=opt.use_coord, output_nc=3, affine=True,
                    nhead=opt.nhead, num_CABs=opt.num_CABs, num_TTBs=opt.num_TTBs)
        self.net_G = networks.define_G(opt, num_blocks=opt.num_blocks, **g_kw)
        # bipath (AE_model.py:78-88): a second generator that takes part in the optimizers and checkpoints; no method of the
        # reference ever calls it, so it is constructed for API / state_dict parity and never run
        self.bipath_gan = bool(getattr(opt, 'bipath_gan', False))
        if self.bipath_gan:
            self.model_names.append('Gb')
            self.net_Gb = networks.define_G(opt, num_blocks=opt.num_CABs, **g_kw)
        self.use_adp = bool(getattr(opt, 'use_adp', False))
        if self.use_adp:
            self.model_names.append('A')
            self.net_A = networks.Resize_ReID(image_nc=opt.image_nc).to(self.device)
        if self.gan_train:
            self.model_names.append('D')
            self.net_D = networks.define_D(opt, ndf=32, img_f=128, layers=opt.dis_layers, use_spect=opt.use_spect_d)
            if self.bipath_gan:
                self.model_names.append('Db')
                self.net_Db = networks.define_D(opt, ndf=32, img_f=128, layers=opt.dis_layers, use_spect=opt.use_spect_d)

        # the small-kernel GAN networks run their forward / backward programs as captured single-stream hipGraphs (rg_hip.netgraph):
        # their steps are otherwise bound by the host's launch rate
        for _n in ('net_G', 'net_Gb', 'net_D', 'net_Db'):
            _m = getattr(self, _n, None)
            _m = getattr(_m, "module", _m)             # through the DataParallel shim
            if _m is not None:
                _m.__dict__["_rg_graph"] = True
        if getattr(self.opt, 'verbose', False):
            print('---------- Networks initialized -------------')

        if self.gan_train:
            self.old_lr = opt.gan_lr
            self.GANloss = external_function.GANLoss(opt.gan_mode).to(self.device)
            if not opt.no_vgg_loss:
                self.Vggloss = external_function.VGGLoss(vgg_weights=getattr(opt, 'vgg_weights', '') or None).to(self.device)

            def trainable(net):
                return itertools.chain(filter(lambda p: p.requires_grad, net.parameters()))
            if self.bipath_gan:            # two parameter groups per optimizer, as AE_model.py:130-156
                g_groups = [{"params": trainable(self.net_G)}, {"params": trainable(self.net_Gb)}]
                d_groups = [{"params": trainable(self.net_D)}, {"params": trainable(self.net_Db)}]
            else:
                g_groups, d_groups = trainable(self.net_G), trainable(self.net_D)
            self.optimizer_G = roptim.Adam(g_groups, lr=opt.gan_lr, betas=(opt.beta1, 0.999))
            self.optimizers = [self.optimizer_G]
            self.optimizer_D = roptim.Adam(d_groups, lr=opt.gan_lr * opt.ratio_g2d, betas=(opt.beta1, 0.999))
            self.optimizers.append(self.optimizer_D)
            self.schedulers = [base_function.get_scheduler(optimizer, opt) for optimizer in self.optimizers]
            self._red_G = GradReducer(self.optimizer_G, modules=[self.net_G] + ([self.net_Gb] if self.bipath_gan else []))
            self._red_D = GradReducer(self.optimizer_D, modules=[self.net_D] + ([self.net_Db] if self.bipath_gan else []))
            _wrap_step(self.optimizer_G, self._red_G)
            _wrap_step(self.optimizer_D, self._red_D)
        else:
            self.net_G.eval()

        if self.load_pretrain != "" or getattr(opt, 'continue_train', False):
            print('model loaded from pretrained')
            self.load_networks(opt.which_epoch)

    # ---- inputs / synthesis (AE_model.py:187-214) -----------------------------------------------------------
    def set_input(self, inputs, b_id=None):
        self.input = inputs
        if b_id is not None:
            source_image = torch.index_select(inputs['Xs'], 0, b_id)
            source_pose = torch.index_select(inputs['Ps'], 0, b_id)
        else:
            source_image = inputs['Xs']
            source_pose = inputs['Ps'] if self.opt.model_gen == 'Pose' else None
        self.source_image = source_image.to(self.device, non_blocking=True).contiguous()
        if self.opt.model_gen == 'Pose':
            self.source_pose = source_pose.to(self.device, non_blocking=True).contiguous()

    def forward(self):
        if self.opt.model_gen == 'Pose':
            raise TypeError("AEModel.forward() calls net_G(source_image); the 'Pose' generator takes (features, pose) — "
                            "use synthesize_p(features) (the reference fails the same way for model_gen='Pose')")
        self.fake_image = self.net_G(self.source_image)

    def synthesize(self, features):
        self.fake_image = self.net_G(features)

    def synthesize_fgan(self):
        """detached encoder features of the source images (AE_model.py:251-254)"""
        with torch.no_grad():
            return self.net_G.module.forward_enc(self.source_image)

    def synthesize_fc(self, reid_f, group_size=16):
        """decode a hard mix of the source images' encoder features (AE_model.py:256-272)"""
        F_s = self.net_G.module.forward_enc(self.source_image)
        self.fake_image = self.net_G.module.forward_dec(self.hard_mix(F_s, reid_f, group_size))
        return self.fake_image

    def hard_mix(self, F_s, reid_f, group_size):
        """lambda_fus * F_s[hardest in-identity sample] + (1 - lambda_fus) * F_s[nearest other-identity sample] per
        identity group (AE_model.py:274-292).  The similarity GEMM and the mix run on HIP kernels; the two index
        selections (exp / argmin / argmax over an [identities, batch] matrix) are selection logic, not arithmetic that
        reaches the result."""
        _, fdim = reid_f.shape
        reid_f = reid_f.detach()
        anchor = RF.normalize_rows(_group_mean(reid_f, group_size))
        inst = RF.normalize_rows(reid_f.contiguous())
        sim = torch.exp(ops.linear_fwd(anchor, inst))
        id_mask = torch.eye(anchor.shape[0], device=sim.device).repeat_interleave(group_size, dim=1)
        in_id = torch.argmin(id_mask * sim + (1 - id_mask) * torch.max(sim), dim=1)
        out_id = torch.argmax((1 - id_mask) * sim, dim=1)
        return _MixRows.apply(F_s, in_id, out_id, float(self.opt.lambda_fus))

    def synthesize_p(self, features):
        self.fake_image = self.net_G(features, self.source_pose)
        return self.fake_image

    # ---- discriminator update (AE_model.py:294-314) ------------------------------------------------------
    def backward_D_basic(self, netD, real, fake):
        D_real = netD(real)
        D_real_loss = self.GANloss(D_real, True, True)
        D_fake = netD(fake.detach())
        D_fake_loss = self.GANloss(D_fake, False, True)
        terms = [(D_real_loss, 0.5), (D_fake_loss, 0.5)]
        if self.opt.gan_mode == 'wgangp':                      # AE_model.py:304-306
            gradient_penalty, _ = external_function.cal_gradient_penalty(netD, real, fake.detach(),
                                                                         alpha=getattr(self, 'gp_alpha', None))
            terms.append((gradient_penalty, 1.0))
        return _weighted(terms)

    def backward_D(self):
        base_function._unfreeze(self.net_D)
        self.loss_dis_img_gen = self.backward_D_basic(self.net_D, self.source_image, self.fake_image)
        self.loss_D = self.loss_dis_img_gen
        _backward(self.loss_D)
        self.loss_D = self.loss_D.detach()

    # ---- generator losses (AE_model.py:316-376) ------------------------------------------------------------
    def backward_G_basic(self, fake_image, target_image, use_d):
        """Returns the MEANS of the reference's element-wise maps, already weighted: (lambda_rec * mean|fake - target|,
        lambda_g * mean((D(fake) - 1)^2), None, None)."""
        loss_app_gen = RF.l1_loss(fake_image, target_image)
        loss_ad_gen = None
        if use_d:
            base_function._freeze(self.net_D)
            D_fake = self.net_D(fake_image)
            if self.opt.gan_mode == 'lsgan':
                loss_ad_gen = RF.mse_const(D_fake, self.GANloss.label(True))
            else:
                loss_ad_gen = self.GANloss(D_fake, True, False)
        if self.opt.no_vgg_loss:
            return loss_app_gen, loss_ad_gen, None, None
        # the reference computes the perceptual terms here (:330-335) but none of its generator objectives adds them
        # (:346, :371, :374): they are reported, detached values
        with torch.no_grad():
            loss_content_gen, loss_style_gen = self.Vggloss(fake_image.detach(), target_image)
            loss_style_gen = loss_style_gen * self.opt.lambda_style
            loss_content_gen = loss_content_gen * self.opt.lambda_content
        return loss_app_gen, loss_ad_gen, loss_style_gen, loss_content_gen

    def _loss_G_mean(self):
        base_function._unfreeze(self.net_D)
        self.loss_app_gen, self.loss_ad_gen, self.loss_style_gen, self.loss_content_gen = \
            self.backward_G_basic(self.fake_image, self.source_image, use_d=True)
        # (app.flatten(1).mean(-1) + ad.flatten(1).mean(-1)).mean() == app.mean() + ad.mean(): equal sample counts
        return _weighted([(self.loss_app_gen, self.opt.lambda_rec), (self.loss_ad_gen, self.opt.lambda_g)])

    def backward_G(self, loss_nl=None, group_size=16):
        self.loss_G = self._loss_G_mean()
        if loss_nl is not None:
            self.loss_G = _weighted([(self.loss_G, 1.0), (loss_nl, 1.0)])
        _backward(self.loss_G)
        self.loss_G = self.loss_G.detach()

    def get_loss_G(self, group_size=None, cf_temp=0.2, need_cm=True, cluster_features=None):
        if need_cm and cluster_features is None:
            raise TypeError("get_loss_G(need_cm=True) needs cluster_features (the reference fails inside net_G without them)")
        self.loss_G = self._loss_G_mean()
        if need_cm:
            # synthesize from the cluster features and report its per-sample reconstruction error (AE_model.py:361-372); the
            # generator objective itself is the same mean as without it
            cluster_image = self.net_G(cluster_features, self.source_pose)
            loss_rec = RF.l1_loss_rows(cluster_image, self.source_image)
            return self.loss_G, loss_rec
        return self.loss_G

    def get_L1_loss(self, with_dis=False):
        """per-sample losses of the current fake (AE_model.py:378-390): mean |fake - source| per sample, or with the discriminator
        lambda_rec * that + lambda_g * the per-sample mean of the GAN map"""
        if not with_dis:
            return RF.l1_loss_rows(self.fake_image, self.source_image)
        if self.opt.gan_mode != 'lsgan':
            raise NotImplementedError("get_L1_loss(with_dis=True) per-sample GAN term: only --gan_mode lsgan (the default) is built")
        base_function._unfreeze(self.net_D)
        base_function._freeze(self.net_D)                     # backward_G_basic(use_d=True) order, AE_model.py:323-326
        loss_rec = RF.l1_loss_rows(self.fake_image, self.source_image)
        loss_dis = RF.mse_const_rows(self.net_D(self.fake_image), self.GANloss.label(True))
        return _weighted_rows([(loss_rec, self.opt.lambda_rec), (loss_dis, self.opt.lambda_g)])

    # ---- stand-alone GAN step (AE_model.py:392-410) ------------------------------------------------------
    def optimize_parameters(self):
        self.forward()
        self.optimize_generated()

    def optimize_generated(self):
        self.optimizer_D.zero_grad()
        self.backward_D()
        self.optimizer_D.step()

        self.optimizer_G.zero_grad()
        self.backward_G()
        self.optimizer_G.step()


def _group_mean(x, group_size):
    """mean over consecutive groups of rows: [n * g, D] -> [n, D] (a GEMM with a constant averaging matrix)"""
    n = x.shape[0] // group_size
    avg = torch.zeros(n, x.shape[0], device=x.device)
    avg[torch.arange(x.shape[0], device=x.device) // group_size, torch.arange(x.shape[0], device=x.device)] = 1.0 / group_size
    return ops.linear_fwd(avg, x.t().contiguous())


class _MixRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, ia, ib, lam):
        ctx.save_for_backward(ia, ib)
        ctx.lam, ctx.shape = lam, src.shape
        return ops.mix_rows_fwd(src, ia, ib, lam)

    @staticmethod
    def backward(ctx, g):
        ia, ib = ctx.saved_tensors
        return ops.mix_rows_bwd(g.contiguous(), ia, ib, ctx.lam, ctx.shape), None, None, None


def _wrap_step(optimizer, reducer):
    """All-reduce the optimizer's gradient arena right before its step when running under torch.distributed (a no-op
    on one rank).  The trainers call `gan.optimizer_X.step()` directly (trainers_b.py:744-753), so the collective lives
    behind that call rather than in the trainer."""
    if not reducer.active():
        return
    inner = optimizer.step

    def step(*a, **kw):
        reducer.reduce()
        return inner(*a, **kw)
    optimizer.step = step
