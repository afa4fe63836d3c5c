"""Step drivers — restates CC/clustercontrast/trainers.py (ClusterContrastTrainer :213-270,
ClusterContrastWithGANTrainer :15-210, GANTrainer :273-335) and the joint ReID + GAN step that only exists in
CC/clustercontrast/trainers_b.py:617-814 (`train_all`), on the HIP runtime.

Same constructors, method names and arguments.  Differences that are scheduling only:
  * host->device copies are non-blocking; `loss.item()` (a device sync per step in the reference, :247)
    is deferred to the print interval unless `sync_every_step=True`;
  * the optimizer's gradient arena is all-reduced over RCCL (rg_hip.parallel.GradReducer) when
    torch.distributed is initialised — the reference used single-process DataParallel;
  * wandb / tensorboard writers are optional (not installed on the target machines).
The encoder's train-mode tuple output is accepted everywhere.
"""
from __future__ import print_function, absolute_import

import time

import torch
from rg_hip.tape import backward as _backward
import torch.nn as nn

from rg_hip import functional as RF
from rg_hip.parallel import GradReducer, attach_stage_hooks

from .utils.meters import AverageMeter

try:                                    # optional observability, never on the hot path
    import wandb as _wandb
except Exception:                       # pragma: no cover
    _wandb = None


def _device():
    return torch.device("cuda", torch.cuda.current_device())


def _first(out):
    return out[0] if isinstance(out, (tuple, list)) else out


class _ReducerCache(object):
    """one GradReducer per optimizer object (created at the top of the first step — the trainers receive the optimizer per call,
    not at construction — so rank 0's broadcast precedes the first forward; no-op without torch.distributed)."""

    def __init__(self):
        self._r = {}

    def get(self, optimizer, modules=None):
        """`modules`: the networks this optimizer trains — rank 0's parameters and buffers are broadcast once when the
        reducer is created, so user-built replicas need not share a seed (rg_hip.parallel.GradReducer)."""
        r = self._r.get(id(optimizer))
        if r is None:
            r = self._r[id(optimizer)] = GradReducer(optimizer, modules=modules)
            if modules is not None:
                # the encoder's arena is reduced stage by stage from inside its backward program (layer4 first), not after it
                attach_stage_hooks(r, *(modules if isinstance(modules, (list, tuple)) else [modules]))
        return r


class ClusterContrastTrainer(object):
    def __init__(self, encoder, memory=None):
        super(ClusterContrastTrainer, self).__init__()
        self.encoder = encoder
        self.memory = memory
        self.sync_every_step = False
        self._reducers = _ReducerCache()

    def train(self, epoch, data_loader, optimizer, print_freq=10, train_iters=400, acc_iters=0):
        self.encoder.train()

        batch_time = AverageMeter()
        data_time = AverageMeter()
        losses = AverageMeter()
        pending = []

        end = time.time()
        for i in range(train_iters):
            inputs = data_loader.next()
            data_time.update(time.time() - end)

            inputs, labels, indexes = self._parse_data(inputs)
            loss = self.step(inputs, labels, optimizer)

            pending.append(loss)
            if self.sync_every_step or (i + 1) % print_freq == 0 or i + 1 == train_iters:
                for v in torch.stack(pending).tolist():     # one device->host transfer for the interval
                    losses.update(v)
                pending = []

            batch_time.update(time.time() - end)
            end = time.time()

            if (i + 1) % print_freq == 0:
                print('Epoch: [{}][{}/{}]\t'
                      'Time {:.3f} ({:.3f})\t'
                      'Data {:.3f} ({:.3f})\t'
                      'Loss {:.3f} ({:.3f})'
                      .format(epoch, i + 1, len(data_loader),
                              batch_time.val, batch_time.avg,
                              data_time.val, data_time.avg,
                              losses.val, losses.avg))

    def step(self, inputs, labels, optimizer):
        """One training step (reference loop body :229-244); returns the detached device loss."""
        self._reducers.get(optimizer, self.encoder)       # replicas are synchronised (rank-0 broadcast) BEFORE the first forward
        f_out = _first(self._forward(inputs))
        loss = RF.weighted_mean(self.memory(f_out, labels))
        optimizer.zero_grad()
        _backward(loss)
        self._reducers.get(optimizer, self.encoder).reduce()
        optimizer.step()
        return loss.detach()

    def _parse_data(self, inputs):
        imgs, _, pids, _, indexes = inputs
        dev = _device()
        return imgs.to(dev, non_blocking=True), pids.to(dev, non_blocking=True), indexes.to(dev, non_blocking=True)

    def _forward(self, inputs):
        return self.encoder(inputs)


class ClusterContrastWithGANTrainer(object):
    def __init__(self, encoder, GAN=None, writer=None, memory=None, opt=None):
        super(ClusterContrastWithGANTrainer, self).__init__()
        self.encoder = encoder
        if GAN is None:
            raise TypeError('GAN not implemented!')       # the reference's `raise('...')` is a TypeError too
        self.gan = GAN
        self.memory = memory
        self.memoryb = memory
        self.writer = writer
        self.f_metric = nn.L1Loss()
        self.sync_every_step = False
        self._reducers = _ReducerCache()
        if opt is not None:
            self.opt = opt
            self.T = opt.cl_temp

    # ---- ReID-only step with the GAN's inputs staged (trainers.py:129-187) -----------------------
    def train_all(self, epoch, data_loader, optimizer, print_freq=10, train_iters=400, acc_iters=0, conf_weight=None,
                  joint=None):
        """`joint=None` follows trainers.py:129-187 when the GAN has no generator loss API and the joint step of
        trainers_b.py:617-814 when it has (`synthesize_p`, `get_loss_G`, `backward_D`)."""
        print("train both gan and reid")
        self.encoder.train()
        if joint is None:
            joint = all(hasattr(self.gan, a) for a in ("synthesize_p", "get_loss_G", "backward_D", "optimizer_G",
                                                       "optimizer_D"))
        batch_time = AverageMeter()
        data_time = AverageMeter()
        losses = AverageMeter()
        pending = []
        end = time.time()

        for i in range(train_iters):
            inputs = data_loader.next()
            data_time.update(time.time() - end)

            reid_inputs, labels, indexes = self._parse_data(inputs[0])
            self.gan.set_input(inputs[1])
            if joint:
                loss = self.joint_step(reid_inputs, labels, indexes, optimizer, conf_weight)
            else:
                self._reducers.get(optimizer, self.encoder)
                out = self._forward(reid_inputs)
                f_out = _first(out)
                f_gan = out[1] if isinstance(out, (tuple, list)) else None
                loss_ori = self.memory(f_out, labels, gan_inputs=None if f_gan is None else f_gan.detach())
                loss = RF.weighted_mean(loss_ori)
                optimizer.zero_grad()
                _backward(loss)
                self._reducers.get(optimizer, self.encoder).reduce()
                optimizer.step()
                loss = loss.detach()

            pending.append(loss)
            if self.sync_every_step or (i + 1) % print_freq == 0 or i + 1 == train_iters:
                for v in torch.stack(pending).tolist():
                    losses.update(v)
                    if _wandb is not None and getattr(_wandb, "run", None) is not None:
                        _wandb.log({"total_loss": v})
                pending = []

            batch_time.update(time.time() - end)
            end = time.time()

            if (i + 1) % print_freq == 0:
                print('Epoch: [{}][{}/{}]\t'
                      'Time {:.3f} ({:.3f})\t'
                      'Data {:.3f} ({:.3f})\t'
                      'Loss {:.3f} ({:.3f})\n'
                      .format(epoch, i + 1, len(data_loader),
                              batch_time.val, batch_time.avg,
                              data_time.val, data_time.avg,
                              losses.val, losses.avg))

    def joint_step(self, reid_inputs, labels, indexes, optimizer, conf_weight=None):
        """Joint ReID + GAN step, live lines of trainers_b.py:657-774: encode -> synthesize from the detached
        feature -> generator loss (D frozen) + confidence-weighted cluster-contrast loss -> D step -> one backward
        through G and the encoder -> both optimizers step."""
        gan = self.gan
        self._reducers.get(optimizer, self.encoder)       # rank-0 broadcast before the first forward, not after its backward
        out = self._forward(reid_inputs)
        f_out = _first(out)
        # train-mode encoders return (bn_x, normalize(feature map)) (CC/clustercontrast/models/resnet.py:107); the
        # committed loop passes that tuple straight to `.detach()`.  The generator's input is the map.
        f_gan = out[1] if isinstance(out, (tuple, list)) and len(out) > 1 and out[1] is not None else f_out
        gan.synthesize_p(f_gan.detach())
        loss_G = gan.get_loss_G(need_cm=False)
        if conf_weight is not None:
            conf_mask = conf_weight[indexes]
        else:
            conf_mask = None
        loss_cl = RF.weighted_mean(self.memory(f_out, labels), conf_mask)
        loss = RF._WeightedSum.apply(torch.stack([loss_cl, loss_G]), None, 1.0)

        gan.optimizer_D.zero_grad()
        gan.backward_D()
        gan.optimizer_D.step()

        gan.optimizer_G.zero_grad()
        optimizer.zero_grad()
        _backward(loss)
        self._reducers.get(optimizer, self.encoder).reduce()
        gan.optimizer_G.step()
        optimizer.step()
        return loss.detach()

    def train(self, epoch, data_loader, optimizer, print_freq=10, train_iters=400, acc_iters=0):
        """ReID training with the GAN's inputs staged (the committed `train`, trainers.py:34-127, calls
        `memory(f_out, labels, ex_f=...)`, valid only for ClusterMemory_Gradient, and `synthesize_fc` with
        inconsistent arity — SURVEY §9.5); the well-defined part, the cluster-contrast update, is what runs."""
        return self.train_all(epoch, data_loader, optimizer, print_freq, train_iters, acc_iters, joint=False)

    def _parse_data(self, inputs):
        imgs, _, pids, _, indexes = inputs
        dev = _device()
        return imgs.to(dev, non_blocking=True), pids.to(dev, non_blocking=True), indexes.to(dev, non_blocking=True)

    def _forward(self, inputs, fuse=True):
        if fuse:
            return self.encoder(inputs)
        return self.encoder(inputs, fuse=fuse)

    def intra_cl(self, q, k, group_size=16):
        """group contrast of trainers.py:200-210: logits[n, g] = sum over the g-th group of k of <q_n, k_m> / T."""
        from clustercontrast.models.cm import _MatmulT
        q = RF.normalize_rows(q)
        k = RF.normalize_rows(k)
        logits = _MatmulT.apply(q, k)
        qs, ks = logits.shape
        ones = torch.zeros(ks // group_size, ks, device=q.device)
        ones[torch.arange(ks, device=q.device) // group_size, torch.arange(ks, device=q.device)] = 1.0
        grouped = _MatmulT.apply(logits, ones)                         # sums each group of `group_size` keys
        targets = torch.arange(group_size, dtype=torch.long, device=q.device).repeat_interleave(group_size)
        return RF.cross_entropy_rows(grouped, targets, 1.0 / self.T)


class GANTrainer(object):
    def __init__(self, GAN=None, encoder=None, writer=None, opt=None):
        super(GANTrainer, self).__init__()
        if GAN is None:
            raise TypeError('GAN not implemented!')
        self.gan = GAN
        self.encoder = encoder
        self.writer = writer
        if opt is not None:
            self.opt = opt
            self.T = opt.cl_temp

    def train_gan(self, epoch, data_loader, print_freq=10, train_iters=400, acc_iters=0):
        print("train gan")
        batch_time = AverageMeter()
        data_time = AverageMeter()
        end = time.time()
        gan_losses = {'G': 0.0, 'D': 0.0}
        for i in range(train_iters):
            inputs = data_loader.next()
            data_time.update(time.time() - end)
            self.gan.set_input(inputs)
            self.gan.optimize_parameters()

            if self.writer is not None or (i + 1) % print_freq == 0:
                gan_losses = self.gan.get_current_errors()
            if self.writer is not None:
                total_steps = acc_iters + i
                self.writer.add_scalar('Loss/G_loss', gan_losses['G'], total_steps)
                self.writer.add_scalar('Loss/D_loss', gan_losses['D'], total_steps)

            batch_time.update(time.time() - end)
            end = time.time()
            if (i + 1) % print_freq == 0:
                if _wandb is not None and getattr(_wandb, "run", None) is not None:
                    _wandb.log({"GANLoss_G": gan_losses['G'], "GANLoss_D": gan_losses['D']})
                print('Epoch: [{}][{}/{}]\t'
                      'Time {:.3f} ({:.3f})\t'
                      'Data {:.3f} ({:.3f})\t'
                      'GANLoss: G:{:.3f} D:{:.3f}\n'
                      .format(epoch, i + 1, len(data_loader),
                              batch_time.val, batch_time.avg,
                              data_time.val, data_time.avg,
                              gan_losses['G'], gan_losses['D']))
