"""FDGANModel — restates the step driver FD-GAN-master/fdgan/model.py:21-264 on the HIP runtime.

Same constructor (`FDGANModel(opt)`), public attributes (net_E / net_G / net_Di / net_Dp, each exposing
`.module`; optimizer_G / optimizer_Di / optimizer_Dp; schedulers) and methods (set_input, forward,
backward_Dp, backward_Di, backward_G, optimize_parameters, get_current_errors, get_current_visuals,
save, update_learning_rate, reset_model_status).  The arithmetic of every method is the reference's;
what differs is how it is scheduled on the MI355X:

  * backward_Di evaluates D_id(origin, target) and D_id(origin, fake) as ONE trunk pass over
    [origin | target | fake] (eval-mode BatchNorm makes samples independent; the shared origin branch
    is computed once and receives both gradients)                                     (reference :176-177)
  * backward_Dp runs D_pd on [real_pose | fake_pose] separately (train-mode BN: batches must stay
    separate to keep the reference's statistics)                                          (:160-163)
  * backward_G treats D_id / D_pd as constants: their weight gradients, which the reference computes
    and then throws away at the next zero_grad(), are not computed                         (:196-204)
  * one process drives one GPU; `.module` wrappers are kept for API compatibility, gradients are
    all-reduced over RCCL by rg_hip.parallel when torch.distributed is initialised.

Defects of the reference that are NOT reproduced: `.data[0]` on 0-dim tensors (-> .item()),
the D_id classifier slice that loses a dimension (:56-57, kept as [1,2048]).
"""
from __future__ import absolute_import

import os
import random
from collections import OrderedDict

import torch
from rg_hip.tape import backward as _backward

from fdgan.losses import GANLoss
from fdgan.networks import (CustomPoseGenerator, NLayerDiscriminator, get_norm_layer, get_scheduler, init_weights,
                            print_network, remove_module_key, set_bn_fix)
from reid.models import create
from reid.models.embedding import EltwiseSubEmbed
from reid.models.multi_branch import SiameseNet
from rg_hip import functional as RF
from rg_hip import ops
from rg_hip import optim as roptim
from rg_hip.parallel import DataParallel, GradReducer, attach_stage_hooks
from rg_hip.tape import no_param_grad


def _weighted(terms):
    """sum_i w_i * loss_i of 0-dim device losses in one kernel (keeps the autograd graph)."""
    vals = torch.stack([t for t, _ in terms])
    return RF._WeightedSum.apply(vals, RF.const_vector(tuple(float(w) for _, w in terms), vals.device), 1.0)


class FDGANModel(object):

    def __init__(self, opt):
        self.opt = opt
        self.save_dir = os.path.join(opt.checkpoints, opt.name)
        self.norm_layer = get_norm_layer(norm_type=opt.norm)
        self.device = torch.device("cuda", torch.cuda.current_device())

        self._init_models()
        self._init_losses()
        self._init_optimizers()

        if not getattr(opt, "quiet", False):
            print('---------- Networks initialized -------------')
            print_network(self.net_E)
            print_network(self.net_G)
            print_network(self.net_Di)
            print_network(self.net_Dp)
            print('-----------------------------------------------')

    # ---- construction -----------------------------------------------------------------------------------------------
    # What is built is the reference's (FD/fdgan/model.py:39-125): E = Siamese(ResNet trunk, 2-way embedding), G, D_id =
    # Siamese(ResNet trunk, 1-way embedding), D_pd = PatchGAN on image + 18 pose channels; stage 1 trains G and the two
    # discriminators on a frozen E, stage 2 everything.  It is organised by what each stage needs rather than by network.
    _STAGE = {
        #        trainable BatchNorm-frozen nets            checkpoint attribute per network (None: initialised, not loaded)
        1: dict(bn_fixed=("Di",), eval_nets=("E",), ckpt=dict(E="netE_pretrain", Di="netE_pretrain", G=None, Dp=None),
                lr_scale=dict(G=0.1, Di=0.01, Dp=1.0)),
        2: dict(bn_fixed=("E", "Di"), eval_nets=(), ckpt=dict(E="netE_pretrain", G="netG_pretrain", Di="netDi_pretrain",
                                                                Dp="netDp_pretrain"),
                lr_scale=dict(G=0.1, Di=1.0, Dp=1.0)),
    }

    def _siamese(self, num_classes):
        pretrained = bool(getattr(self.opt, "imagenet_pretrained", False))
        trunk = create(self.opt.arch, cut_at_pooling=True, pretrained=pretrained)
        return SiameseNet(trunk, EltwiseSubEmbed(use_batch_norm=True, use_classifier=True, num_features=2048,
                                                 num_classes=num_classes))

    def _init_models(self):
        opt = self.opt
        if opt.stage not in self._STAGE:
            raise ValueError("unknown training stage %r" % (opt.stage,))
        nets = dict(
            G=CustomPoseGenerator(opt.pose_feature_size, 2048, opt.noise_feature_size, dropout=opt.drop,
                                  norm_layer=self.norm_layer, fuse_mode=opt.fuse_mode, connect_layers=opt.connect_layers),
            E=self._siamese(2), Di=self._siamese(1), Dp=NLayerDiscriminator(3 + 18, norm_layer=self.norm_layer))
        random_init = bool(getattr(opt, "random_init", False))      # synthetic benchmarks / tests: no checkpoints
        for name, attr in self._STAGE[opt.stage]["ckpt"].items():
            if attr is None or (random_init and name in ("G", "Dp")):
                init_weights(nets[name])
            elif not random_init:
                state = remove_module_key(torch.load(getattr(opt, attr), map_location="cpu"))
                if opt.stage == 1 and name == "Di":
                    # D_id starts from E with the 'same identity' row of its 2-way classifier (:56-57; kept [1, 2048])
                    state = dict(state)
                    for key in ("embed_model.classifier.weight", "embed_model.classifier.bias"):
                        state[key] = state[key][1:2]
                nets[name].load_state_dict(state)
        for name, net in nets.items():
            setattr(self, "net_" + name, DataParallel(net).to(self.device))

    def reset_model_status(self):
        plan = self._STAGE[self.opt.stage]
        for name in ("E", "G", "Di", "Dp"):
            net = getattr(self, "net_" + name)
            net.eval() if name in plan["eval_nets"] else net.train()
        for name in plan["bn_fixed"]:
            getattr(self, "net_" + name).apply(set_bn_fix)

    def _load_state_dict(self, net, path):
        net.load_state_dict(remove_module_key(torch.load(path, map_location="cpu")))

    def _init_losses(self):
        smooth = bool(self.opt.smooth_label)
        self.criterionGAN_D = GANLoss(smooth=smooth)
        self.criterionGAN_G = GANLoss(smooth=False)
        # label flip for the discriminators: one draw in 10001 when smoothing, never otherwise (:90-98)
        self.rand_list = [True] + [False] * 10000 if smooth else [False]

    def _init_optimizers(self):
        opt = self.opt
        scale = self._STAGE[opt.stage]["lr_scale"]
        if opt.stage == 2:
            # `lr_mult` is carried in the groups but never applied, exactly as in the reference (:109-113)
            g_params = [{'params': self.net_E.module.base_model.parameters(), 'lr_mult': 0.1},
                        {'params': self.net_E.module.embed_model.parameters(), 'lr_mult': 1.0},
                        {'params': self.net_G.parameters(), 'lr_mult': 0.1}]
            g_nets = [self.net_G, self.net_E]
        else:
            g_params, g_nets = self.net_G.parameters(), [self.net_G]
        self.optimizer_G = roptim.Adam(g_params, lr=opt.lr * scale["G"], betas=(0.5, 0.999))
        self.optimizer_Di = roptim.SGD(self.net_Di.parameters(), lr=opt.lr * scale["Di"], momentum=0.9, weight_decay=1e-4)
        self.optimizer_Dp = roptim.SGD(self.net_Dp.parameters(), lr=opt.lr * scale["Dp"], momentum=0.9, weight_decay=1e-4)
        self.optimizers = [self.optimizer_G, self.optimizer_Di, self.optimizer_Dp]
        self.schedulers = [get_scheduler(o, opt) for o in self.optimizers]
        # data-parallel gradient reduction (no-op unless torch.distributed is initialised); rank 0's parameters and
        # buffers are broadcast once here, as DataParallel's replicate() does on every call
        self.reducers = [GradReducer(self.optimizer_G, modules=g_nets),
                         GradReducer(self.optimizer_Di, modules=[self.net_Di]),
                         GradReducer(self.optimizer_Dp, modules=[self.net_Dp])]
        if self.reducers[0].active():
            g_list = list(self.net_G.parameters())
            self.net_G.module._rg_after_backward = lambda: self.reducers[0].reduce_async(g_list)
        # the two ResNet-50 trunks launch their ranges stage by stage from inside their backward programs (layer4 first): only the
        # stem + layer1 bytes of E are left exposed at the tail of backward_G
        if opt.stage == 2:
            attach_stage_hooks(self.reducers[0], self.net_E)
        attach_stage_hooks(self.reducers[1], self.net_Di)

    def _aux_stream(self):
        """Second HIP stream for work that is independent of the main chain (RG_AUX_STREAM=0 disables it).  With label
        smoothing the two discriminator passes draw from Python's `random` (GANLoss, label flip): they then run in the
        reference's order on one stream so that a seeded run consumes the generator exactly like the reference."""
        if os.environ.get("RG_AUX_STREAM", "1") == "0" or self.opt.smooth_label:
            return None
        s = getattr(self, "_aux", None)
        if s is None:
            s = self._aux = ops.concurrent_stream(self.device, avoid=(torch.cuda.current_stream(),))
            if os.environ.get("RG_AUX_SIDE", "0") != "1":
                # D_pd's four weight gradients stay on the auxiliary stream: three compute streams + the collectives' stream are
                # the four hardware queues the device schedules without time-slicing
                ops.side_inline(s)
        return s

    def set_input(self, input):
        input1, input2 = input
        labels = (input1['pid'] == input2['pid']).long()
        noise = input1.get('noise') if isinstance(input1, dict) else None     # tests feed z explicitly
        if noise is None:
            noise = torch.randn(labels.size(0), self.opt.noise_feature_size, device=labels.device)

        dev = self.device
        if input1['origin'].is_cuda and input1['origin'].dtype == torch.float32:
            # batch already on the GPU (device-side input pipeline): the pair tensors in one launch each — the 0 / 1 blend
            # x1*mask + x2*(1 - mask) of the reference is a per-sample selection
            self.labels = labels.to(dev, non_blocking=True).contiguous()
            z = noise.to(dev, non_blocking=True).float()
            self.origin = ops.pair_cat(input1['origin'], input2['origin'])
            self.target = ops.pair_cat(input1['target'], input2['target'], self.labels)
            self.posemap = ops.pair_cat(input1['posemap'], input2['posemap'], self.labels)
            self.noise = ops.pair_cat(z, z)
            return

        # host batch (the reference's data loader): same expressions as the reference, then one copy per tensor
        # keep the same pose map for persons with the same identity
        mask = labels.view(-1, 1, 1, 1).expand_as(input1['posemap'])
        posemap2 = input1['posemap'] * mask.float() + input2['posemap'] * (1 - mask.float())
        mask = labels.view(-1, 1, 1, 1).expand_as(input1['target'])
        target2 = input1['target'] * mask.float() + input2['target'] * (1 - mask.float())

        origin = torch.cat([input1['origin'], input2['origin']])
        target = torch.cat([input1['target'], target2])
        posemap = torch.cat([input1['posemap'], posemap2])
        noise = torch.cat((noise, noise))

        self.origin = origin.to(dev, non_blocking=True).contiguous()
        self.target = target.to(dev, non_blocking=True).contiguous()
        self.posemap = posemap.to(dev, non_blocking=True).contiguous()
        self.labels = labels.to(dev, non_blocking=True).contiguous()
        self.noise = noise.to(dev, non_blocking=True).contiguous()

    def forward(self):
        A = self.origin
        B_map = self.posemap
        z = self.noise
        bs = A.size(0)

        A_id1, A_id2, self.id_score = self.net_E(A[:bs // 2], A[bs // 2:])
        A_id = torch.cat((A_id1, A_id2))
        self.fake = self.net_G(B_map, A_id.view(A_id.size(0), A_id.size(1), 1, 1), z.view(z.size(0), z.size(1), 1, 1))

    def backward_Dp(self):
        real_pose = ops.cat_channels([self.posemap, self.target])
        fake_pose = ops.cat_channels([self.posemap, self.fake.detach()])
        pred_real = self.net_Dp(real_pose)
        pred_fake = self.net_Dp(fake_pose)

        if random.choice(self.rand_list):
            loss_D_real = self.criterionGAN_D(pred_fake, True)
            loss_D_fake = self.criterionGAN_D(pred_real, False)
        else:
            loss_D_real = self.criterionGAN_D(pred_real, True)
            loss_D_fake = self.criterionGAN_D(pred_fake, False)
        loss_D = _weighted([(loss_D_real, 0.5), (loss_D_fake, 0.5)])
        _backward(loss_D)
        self.loss_Dp = loss_D.detach()

    def backward_Di(self):
        pred_real, pred_fake = self.net_Di.module.forward_shared(self.origin, [self.target, self.fake.detach()])
        if random.choice(self.rand_list):
            loss_D_real = self.criterionGAN_D(pred_fake, True)
            loss_D_fake = self.criterionGAN_D(pred_real, False)
        else:
            loss_D_real = self.criterionGAN_D(pred_real, True)
            loss_D_fake = self.criterionGAN_D(pred_fake, False)
        loss_D = _weighted([(loss_D_real, 0.5), (loss_D_fake, 0.5)])
        _backward(loss_D)
        self.loss_Di = loss_D.detach()

    def build_loss_G(self):
        """The generator objective of backward_G (:188-214) as an attached 0-dim device tensor (no backward yet);
        the discriminators take part as constants."""
        loss_v = RF.cross_entropy(self.id_score, self.labels.view(-1))
        loss_r = RF.l1_loss(self.fake, self.target)
        half = self.fake.size(0) // 2
        fake_1 = self.fake[:half]
        fake_2 = self.fake[half:]
        loss_sp = RF.l1_loss(fake_1, fake_2, self.labels.view(-1))       # rows with label == 1 (:191-194)

        aux = self._aux_stream()
        with no_param_grad(self.net_Di.module, self.net_Dp.module):
            if aux is not None:
                # D_pd's pass (and, through autograd's stream bookkeeping, its backward) runs next to D_id's
                main = torch.cuda.current_stream()
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    pred_fake_Dp = self.net_Dp(self.posemap, self.fake)
                    loss_G_GAN_Dp = self.criterionGAN_G(pred_fake_Dp, True)
                _, _, pred_fake_Di = self.net_Di(self.origin, self.fake)
                loss_G_GAN_Di = self.criterionGAN_G(pred_fake_Di, True)
                main.wait_stream(aux)
            else:
                _, _, pred_fake_Di = self.net_Di(self.origin, self.fake)
                pred_fake_Dp = self.net_Dp(self.posemap, self.fake)
                loss_G_GAN_Di = self.criterionGAN_G(pred_fake_Di, True)
                loss_G_GAN_Dp = self.criterionGAN_G(pred_fake_Dp, True)

        loss_G = _weighted([(loss_G_GAN_Di, 1.0), (loss_G_GAN_Dp, 1.0),
                            (loss_r, self.opt.lambda_recon),
                            (loss_v, self.opt.lambda_veri),
                            (loss_sp, self.opt.lambda_sp)])
        self.loss_G = loss_G.detach()
        self.loss_v = loss_v.detach()
        self.loss_sp = loss_sp.detach()
        self.loss_r = loss_r.detach()
        self.loss_G_GAN_Di = loss_G_GAN_Di.detach()
        self.loss_G_GAN_Dp = loss_G_GAN_Dp.detach()
        return loss_G

    def backward_G(self):
        _backward(self.build_loss_G())
        del self.id_score
        self.fake = self.fake.detach()

    def optimize_parameters(self):
        """Reference order (:216-229): forward; D_id update; D_pd update; G(+E) update.  The D_pd backward does not
        read D_id's weights, so D_id's optimizer step is issued after it: D_id's gradient all-reduce (RCCL, side
        stream) then overlaps the D_pd backward; in the generator update G's gradients are reduced while the
        encoder's backward still runs (`_rg_after_backward`).  Results are those of the reference order."""
        self.forward()

        aux = self._aux_stream()
        self.optimizer_Di.zero_grad()
        self.optimizer_Dp.zero_grad()
        if aux is not None:
            # the two discriminator updates are independent: D_pd's runs on a second stream next to D_id's
            main = torch.cuda.current_stream()
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                self.backward_Dp()
            self.backward_Di()
            self.reducers[1].reduce_async()
            main.wait_stream(aux)
            self.reducers[2].reduce_async()
        else:
            self.backward_Di()
            self.reducers[1].reduce_async()
            self.backward_Dp()
            self.reducers[2].reduce_async()

        self.reducers[1].wait()
        self.optimizer_Di.step()
        self.reducers[2].wait()
        self.optimizer_Dp.step()

        self.optimizer_G.zero_grad()
        self.backward_G()
        self.reducers[0].reduce()
        self.optimizer_G.step()

    def get_current_errors(self):
        # one device->host transfer for all seven scalars (the reference synchronises seven times)
        vals = torch.stack([self.loss_v, self.loss_r, self.loss_sp, self.loss_G_GAN_Di, self.loss_G_GAN_Dp,
                            self.loss_Di, self.loss_Dp]).tolist()
        return OrderedDict(zip(['G_v', 'G_r', 'G_sp', 'G_gan_Di', 'G_gan_Dp', 'D_i', 'D_p'], vals))

    def get_current_visuals(self):
        import fdgan.utils.util as util            # visualisation helpers are outside the hot path
        input = util.tensor2im(self.origin)
        target = util.tensor2im(self.target)
        fake = util.tensor2im(self.fake)
        map = self.posemap.sum(1)
        map[map > 1] = 1
        map = util.tensor2im(torch.unsqueeze(map, 1))
        return OrderedDict([('input', input), ('posemap', map), ('fake', fake), ('target', target)])

    def save(self, epoch):
        self.save_network(self.net_E, 'E', epoch)
        self.save_network(self.net_G, 'G', epoch)
        self.save_network(self.net_Di, 'Di', epoch)
        self.save_network(self.net_Dp, 'Dp', epoch)

    def save_network(self, network, network_label, epoch_label):
        save_filename = '%s_net_%s.pth' % (epoch_label, network_label)
        os.makedirs(self.save_dir, exist_ok=True)
        save_path = os.path.join(self.save_dir, save_filename)
        torch.save({k: v.detach().cpu().clone() for k, v in network.state_dict().items()}, save_path)

    def update_learning_rate(self):
        for scheduler in self.schedulers:
            scheduler.step()
        lr = self.optimizers[0].param_groups[0]['lr']
        return lr


class _CatPose(torch.autograd.Function):
    """torch.cat((posemap, fake), dim=1) with the gradient sliced back to `fake` (reference :197)."""

    @staticmethod
    def forward(ctx, posemap, fake):
        ctx.c0, ctx.c1 = posemap.shape[1], posemap.shape[1] + fake.shape[1]
        return ops.cat_channels([posemap, fake])

    @staticmethod
    def backward(ctx, g):
        return None, ops.slice_channels(g, ctx.c0, ctx.c1)
