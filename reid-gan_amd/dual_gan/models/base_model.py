"""BaseModel of the dual_gan GAN objects — the training-side part of CC/dual_gan/models/base_model.py:12-150
(errors, save / load of networks, learning-rate schedule).  Visualisation and result dumping (:39-66, :153-207: cv2,
pose drawing) are outside the training step and not rebuilt."""
from __future__ import absolute_import

import os
from collections import OrderedDict

import torch


class BaseModel(object):
    def name(self):
        return 'BaseModel'

    def __init__(self, opt):
        self.opt = opt
        self.gan_train = opt.gan_train
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.load_pretrain = opt.load_pretrain

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def test(self):
        pass

    def optimize_parameters(self):
        pass

    def get_current_errors(self):
        """Return training loss (one device->host read per entry)"""
        errors_ret = OrderedDict()
        for name in self.loss_names:
            if isinstance(name, str):
                errors_ret[name] = getattr(self, 'loss_' + name).item()
        return errors_ret

    def save(self, label):
        pass

    def save_networks(self, which_epoch):
        """Save all the networks to the disk (same file names / key layout as the reference)"""
        os.makedirs(self.save_dir, exist_ok=True)
        for name in self.model_names:
            if isinstance(name, str):
                save_filename = '%s_net_%s.pth' % (which_epoch, name)
                save_path = os.path.join(self.save_dir, save_filename)
                net = getattr(self, 'net_' + name)
                torch.save(OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items()), save_path)

    def load_networks(self, which_epoch):
        """Load all the networks from the disk; tolerant of a missing / extra `module.` prefix like the reference
        (:97-141)."""
        for name in self.model_names:
            if not isinstance(name, str):
                continue
            filename = '%s_net_%s.pth' % (which_epoch, name)
            path = os.path.join(self.save_dir if self.load_pretrain == "" else self.load_pretrain, filename)
            net = getattr(self, 'net_' + name)
            if not os.path.exists(path):
                print('do not find checkpoint for network %s' % name)
                continue
            pretrained = torch.load(path, map_location="cpu")
            model_dict = net.state_dict()
            for variant in (lambda k: k, lambda k: k.replace('module.', '', 1), lambda k: 'module.' + k):
                picked = {variant(k): v for k, v in pretrained.items() if variant(k) in model_dict
                          and v.shape == model_dict[variant(k)].shape}
                if picked:
                    break
            missing = sorted({k.split('.')[0] for k in model_dict if k not in picked})
            if missing:
                print('Pretrained network %s has fewer layers; The following are not initialized:' % name)
                print(missing)
            model_dict.update(picked)
            net.load_state_dict(model_dict)
            print('load %s from %s' % (name, path))

    def update_learning_rate(self, epoch=None):
        """Update learning rate"""
        for scheduler in self.schedulers:
            if epoch is None:
                scheduler.step()
            else:
                scheduler.step(epoch)

    def get_current_learning_rate(self):
        lr_G = self.optimizers[0].param_groups[0]['lr']
        lr_D = self.optimizers[1].param_groups[0]['lr']
        return lr_G, lr_D
