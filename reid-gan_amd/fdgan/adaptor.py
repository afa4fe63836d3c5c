"""FDGANModel behind the GAN duck-type of the joint ReID + GAN trainer (BASELINE config 4b).

The committed joint step (CC/clustercontrast/trainers_b.py:617-774) drives a GAN object through
`set_input / synthesize_p / get_loss_G / optimizer_D.zero_grad / backward_D / optimizer_D.step / optimizer_G.zero_grad /
(loss_cl + loss_G).backward() / optimizer_G.step`; the FD-GAN wiring of the example script is commented out
(examples/cluster_contrast_gan_train_usl_infomap.py:28-31,243-246).  This adaptor supplies that surface for
`fdgan.model.FDGANModel`, so the north star's "joint FD-GAN + cluster-contrast" step runs through the same trainer:

  set_input(pair_batch)   -> FDGANModel.set_input (two dicts: pid / origin / target / posemap)
  synthesize_p(features)  -> FDGANModel.forward(): E encodes `origin`, G renders the target pose (the ReID encoder's
                             features are not consumed: the FD-GAN generator is conditioned on its own Siamese encoder E)
  get_loss_G()            -> the backward_G objective as an attached scalar (D_id, D_pd as constants)
  backward_D()            -> backward_Di() + backward_Dp()
  optimizer_D / _G        -> D_id + D_pd SGD pair / the G(+E) Adam

One semantic point the trainer's call order forces: `optimizer_D.step()` is called BEFORE the generator backward that
still needs the discriminators' weights of the forward pass.  The discriminator steps are therefore recorded and applied
right after `optimizer_G.step()` — D and G are both updated from the same pre-update weights, exactly what the reference's
own joint step computes for AEModel, whose spectral-normed convolutions keep `W / sigma` of the forward pass in the graph.
"""
from __future__ import absolute_import

from collections import OrderedDict



class _DeferredD(object):
    def __init__(self, owner):
        self.owner = owner
        self.pending = False

    @property
    def param_groups(self):
        m = self.owner.model
        return m.optimizer_Di.param_groups + m.optimizer_Dp.param_groups

    def zero_grad(self, set_to_none=True):
        m = self.owner.model
        m.optimizer_Di.zero_grad()
        m.optimizer_Dp.zero_grad()

    def step(self):
        m = self.owner.model
        m.reducers[1].reduce_async()          # the collectives overlap the generator backward that follows
        m.reducers[2].reduce_async()
        self.pending = True

    def flush(self):
        if not self.pending:
            return
        m = self.owner.model
        m.reducers[1].wait()
        m.optimizer_Di.step()
        m.reducers[2].wait()
        m.optimizer_Dp.step()
        self.pending = False


class _GThenD(object):
    def __init__(self, owner):
        self.owner = owner

    @property
    def param_groups(self):
        return self.owner.model.optimizer_G.param_groups

    def zero_grad(self, set_to_none=True):
        self.owner.model.optimizer_G.zero_grad()

    def step(self):
        m = self.owner.model
        m.reducers[0].reduce()
        m.optimizer_G.step()
        self.owner.optimizer_D.flush()
        if hasattr(m, "id_score"):
            del m.id_score
        m.fake = m.fake.detach()


class FDGANAdaptor(object):
    def __init__(self, model):
        self.model = model
        self.optimizer_D = _DeferredD(self)
        self.optimizer_G = _GThenD(self)
        self.loss_names = ['G', 'D']

    def set_input(self, inputs):
        self.model.set_input(inputs)

    def synthesize_p(self, features=None):
        self.model.forward()
        self.fake_image = self.model.fake
        return self.fake_image

    def get_loss_G(self, group_size=None, cf_temp=0.2, need_cm=False, cluster_features=None):
        if need_cm:
            raise NotImplementedError("get_loss_G(need_cm=True) is used only by commented-out trainer variants")
        self.loss_G = self.model.build_loss_G()
        return self.loss_G

    def backward_D(self):
        self.model.backward_Di()
        self.model.backward_Dp()

    def optimize_parameters(self):
        self.model.optimize_parameters()

    def get_current_errors(self):
        errs = self.model.get_current_errors()
        out = OrderedDict([('G', float(self.model.loss_G)), ('D', errs['D_i'] + errs['D_p'])])
        out.update(errs)
        return out
