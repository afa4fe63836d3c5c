"""Cluster-contrast path on the MI355X vs the oracle: encoder (layer4 stride 1, GeM, feat_bn), ClusterMemory
(logits, per-sample loss, input gradient with the PRE-update bank, ordered momentum update with repeated labels,
hard variant) and the ClusterContrastTrainer step."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_modules_gpu import _check, _check_grads_anchored, _check_l2

pytestmark = pytest.mark.gpu


def test_cluster_memory(dev):
    from oracle import ref_torch as O
    from clustercontrast.models.cm import ClusterMemory
    g = torch.Generator().manual_seed(1)
    K, D, B = 2048, 2048, 64
    bank = F.normalize(torch.randn(K, D, generator=g), dim=1)
    x = torch.randn(B, D, generator=g) * 2
    ids = torch.randint(0, K, (4,), generator=g)
    labels = ids.repeat_interleave(16)[torch.randperm(B, generator=g)]        # 4 identities x 16 instances
    for hard in (False, True):
        om = O.OClusterMemory(D, K, temp=0.05, momentum=0.2, use_hard=hard)
        om.features = bank.clone()
        rm = ClusterMemory(D, K, temp=0.05, momentum=0.2, use_hard=hard).to(dev)
        rm.features = bank.clone().to(dev)                    # callers re-assign the buffer every epoch
        xo = x.clone().requires_grad_(True)
        xr = x.clone().to(dev).requires_grad_(True)
        lo, lr = om(xo, labels), rm(xr, labels.to(dev))
        assert lr.shape == (B,)
        _check(lr, lo, 1e-4, "per-sample loss hard=%s" % hard)
        lo.mean().backward()
        lr.mean().backward()
        _check(xr.grad, xo.grad, 1e-4, "grad_inputs hard=%s" % hard)
        _check(rm.features, om.features, 1e-5, "bank after update hard=%s" % hard)
        assert not torch.equal(rm.features.cpu(), bank)


def test_cc_functions_api(dev):
    """cm()/cm_hard() keep the reference signature, including a tensor-valued momentum."""
    from oracle import ref_torch as O
    from clustercontrast.models import cm as M
    from tests.golden import cases as C
    bank, feats, labels, gout = C.cm_case()
    for fn, ofn in ((M.cm, O.OCM), (M.cm_hard, O.OCMHard)):
        bo, br = bank.clone(), bank.clone().to(dev)
        xo, xr = feats.clone().requires_grad_(True), feats.clone().to(dev).requires_grad_(True)
        yo = ofn.apply(xo, labels, bo, torch.Tensor([0.2]))
        yr = fn(xr, labels.to(dev), br, torch.Tensor([0.2]).to(dev))
        yo.backward(gout)
        yr.backward(gout.to(dev))
        _check(yr, yo, 1e-5, "logits")
        _check(xr.grad, xo.grad, 1e-5, "grad")
        _check(br, bo, 1e-5, "bank")


def _cc_pair(O, dev, pooling):
    import clustercontrast.models as M
    torch.manual_seed(3)
    o = O.OCCResNet(50, pooling_type=pooling)
    r = M.create('resnet50', pretrained=False, pooling_type=pooling)
    r.load_state_dict(o.state_dict())
    return o, r.to(dev)


@pytest.mark.parametrize("pooling", ["gem", "avg"])
def test_cc_encoder(dev, pooling):
    from oracle import ref_torch as O
    o, r = _cc_pair(O, dev, pooling)
    x = O.synth_images(8, 128, 64, seed=4)
    # eval: L2-normalised embedding
    o.eval()
    r.eval()
    with torch.no_grad():
        _check(r(x.to(dev)), o(x), 1e-3, "eval embedding")
    # train: tuple (bn_x, normalised map), gradients anchored on fp64 (train-mode BN, tiny batch)
    o.train()
    r.train()
    o64 = O.OCCResNet(50, pooling_type=pooling)
    o64.load_state_dict(o.state_dict())
    o64 = o64.double().train()
    bo, go = o(x)
    br, gr = r(x.to(dev))
    b64, _ = o64(x.double())
    _check(br, bo, 1e-3, "bn_x")
    _check(gr, go, 1e-3, "normalised map")
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(bo.shape, generator=gen)
    bo.backward(g)
    br.backward(g.to(dev))
    b64.backward(g.double())
    _check_grads_anchored(r, o, o64, "cc encoder " + pooling)
    assert r.feat_bn.bias.grad is None                      # frozen bias (resnet.py:61)


def _trainer_pair(O, dev, depth):
    import clustercontrast.models as M
    import reid.models as RM
    from clustercontrast.models.cm import ClusterMemory
    torch.manual_seed(3)
    if depth >= 50:
        o = O.OCCResNet(depth, pooling_type="gem")
        r = M.create('resnet%d' % depth, pretrained=False, pooling_type="gem")
    else:
        # the cluster-contrast wrapper's stride edit (resnet.py:34-35) only fits Bottleneck depths — with a
        # BasicBlock trunk the reference itself fails with a shape mismatch — so the shallow encoder for the
        # multi-step trainer test is the FD-GAN ReID wrapper (any encoder works with the trainer)
        o = O.OReidResNet(depth)
        r = RM.create('resnet%d' % depth, pretrained=False)
    r.load_state_dict(o.state_dict())
    r.to(dev).train()
    o.train()
    D = o.num_features
    K = 64
    g = torch.Generator().manual_seed(6)
    bank = F.normalize(torch.randn(K, D, generator=g), dim=1)
    om = O.OClusterMemory(D, K, temp=0.05, momentum=0.1)
    om.features = bank.clone()
    rm = ClusterMemory(D, K, temp=0.05, momentum=0.1).to(dev)
    rm.features = bank.clone().to(dev)
    return o, r, om, rm, g, K


def _per_tensor_adam(net, cls):
    # the reference builds one Adam group per tensor (examples/cluster_contrast_gan_train_usl_infomap.py:281-284)
    return cls([{"params": [p]} for p in net.parameters() if p.requires_grad], lr=3.5e-4, weight_decay=5e-4)


def test_cc_trainer_step_resnet50(dev):
    """First ClusterContrastTrainer step at the real depth: loss and updated bank within 1e-3.
    (Later steps of a randomly initialised train-mode-BN ResNet-50 at batch 16 are chaotic inside the reference
    itself: its own fp32 CPU run moves by 0.3 % in the step-2 loss when only the thread count changes and by 1.1 %
    in the step-1 loss under a 1e-6 input perturbation, so multi-step trajectories are checked on the
    well-conditioned ResNet-18 below.)"""
    from oracle import ref_torch as O
    from clustercontrast.trainers import ClusterContrastTrainer
    from rg_hip import optim as roptim
    o, r, om, rm, g, K = _trainer_pair(O, dev, 50)
    oopt, ropt = _per_tensor_adam(o, torch.optim.Adam), _per_tensor_adam(r, roptim.Adam)
    trainer = ClusterContrastTrainer(r, rm)
    x = O.synth_images(16, 128, 64, seed=10)
    labels = torch.randint(0, K, (4,), generator=g).repeat_interleave(4)
    lo = O.o_cc_step(o, om, oopt, x, labels)
    lr = trainer.step(x.to(dev), labels.to(dev), ropt).item()
    assert abs(lr - lo) <= 1e-3 * abs(lo), (lr, lo)
    _check_l2(rm.features, om.features, 1e-3, "bank after step 0")


def test_cc_trainer_three_steps(dev):
    """Three consecutive steps (forward, CM backward + ordered bank update, per-tensor-group Adam with weight
    decay) on ResNet-18, whose gradients are well conditioned in fp32: per-step loss within 1e-3 of the fp32 oracle.
    The bank rows written after step >= 1 are features of weights that went through Adam steps — a g / (|g| + eps) update
    that amplifies rounding differences of the gradients — so two fp32 implementations are not compared with each other
    there: both are judged against an fp64 run of the oracle, and the HIP path must stay as close to it as the oracle's own
    fp32 CPU run does (factor 2 for the run-to-run spread; measured: the two distances are within 20 % of each other)."""
    import copy
    from oracle import ref_torch as O
    from clustercontrast.trainers import ClusterContrastTrainer
    from rg_hip import optim as roptim
    o, r, om, rm, g, K = _trainer_pair(O, dev, 18)
    o64, om64 = copy.deepcopy(o).double(), copy.deepcopy(om)
    om64.features = om.features.double()
    oopt, ropt = _per_tensor_adam(o, torch.optim.Adam), _per_tensor_adam(r, roptim.Adam)
    oopt64 = _per_tensor_adam(o64, torch.optim.Adam)
    trainer = ClusterContrastTrainer(r, rm)

    def rel(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).norm() / b.norm()).item()
    for it in range(3):
        x = O.synth_images(16, 128, 64, seed=10 + it)
        labels = torch.randint(0, K, (4,), generator=g).repeat_interleave(4)
        lo = O.o_cc_step(o, om, oopt, x, labels)
        lo64 = O.o_cc_step(o64, om64, oopt64, x.double(), labels)
        lr = trainer.step(x.to(dev), labels.to(dev), ropt).item()
        assert abs(lr - lo) <= 1e-3 * abs(lo), "step %d loss %r vs %r" % (it, lr, lo)
        assert abs(lr - lo64) <= 1e-3 * abs(lo64), "step %d loss %r vs fp64 %r" % (it, lr, lo64)
        d_hip, d_cpu = rel(rm.features, om64.features), rel(om.features, om64.features)
        print("bank after step %d: HIP vs fp64 %.3e, fp32 CPU oracle vs fp64 %.3e" % (it, d_hip, d_cpu))
        if it == 0:
            _check_l2(rm.features, om.features, 1e-3, "bank after step 0")
        assert d_hip <= max(1e-3, 2.0 * d_cpu), "bank after step %d: HIP %.3e from the fp64 trajectory, the fp32 oracle %.3e" % (it, d_hip, d_cpu)


def test_cm_gan(dev):
    """cm_gan(): logits, input gradient and BOTH banks (ReID features, GAN features) after the ordered update."""
    import torch.nn.functional as F
    from oracle import ref_torch as O
    from clustercontrast.models import cm as M
    from tests.golden import cases as C
    bank, feats, labels, gout = C.cm_case()
    gbank = F.normalize(bank.flip(0) + 0.1, dim=1)
    gfeat = feats.flip(1) * 3.0 + 0.2
    bo, go = bank.clone(), gbank.clone()
    br, gr = bank.clone().to(dev), gbank.clone().to(dev)
    xo, xr = feats.clone().requires_grad_(True), feats.clone().to(dev).requires_grad_(True)
    yo = O.OCMGan.apply(xo, gfeat, labels, bo, go, torch.Tensor([0.2]))
    yr = M.cm_gan(xr, gfeat.to(dev), labels.to(dev), br, gr, 0.2)
    yo.backward(gout)
    yr.backward(gout.to(dev))
    _check(yr, yo, 1e-5, "logits")
    _check(xr.grad, xo.grad, 1e-5, "grad")
    _check(br, bo, 1e-5, "bank")
    _check(gr, go, 1e-5, "gan bank")


def test_cluster_memory_gradient(dev):
    """ClusterMemory_Gradient: loss (with and without the extra negatives), gradient to the inputs, and the centroid
    update (listed rows of the gradient normalised, SGD step, re-normalised copy)."""
    import torch.nn.functional as F
    from oracle import ref_torch as O
    from clustercontrast.models.cm import ClusterMemory_Gradient
    g = torch.Generator().manual_seed(9)
    K, D, B = 12, 64, 16
    clusters = torch.randn(K, D, generator=g)
    om = O.OClusterMemoryGradient(temp=0.05)
    om.set_clusters(clusters, 0.1)
    rm = ClusterMemory_Gradient(D, K, temp=0.05)
    rm.set_clusters(clusters.to(dev), 0.1)
    labels = torch.randint(0, K, (B,), generator=g)
    x = torch.randn(B, D, generator=g)
    ex = torch.randn(4, D, generator=g)                     # 4 extra negatives, group size 4
    for ex_f in (None, ex):
        xo, xr = x.clone().requires_grad_(True), x.clone().to(dev).requires_grad_(True)
        lo = om.forward(xo, labels, ex_f)
        lr = rm(xr, labels.to(dev), None if ex_f is None else ex_f.to(dev))
        assert abs(lr.item() - lo.item()) <= 1e-4 * abs(lo.item()), (lr.item(), lo.item())
        lo.backward()
        lr.backward()
        _check(xr.grad, xo.grad, 1e-4, "d inputs")
    # centroid update from an externally produced gradient (the reference fills .grad through the GAN loss)
    gc = torch.randn(K, D, generator=g)
    om.trainable_clusters.grad = gc.clone()
    rm.trainable_clusters.grad = gc.clone().to(dev)
    ids = [3, 7, 3, 0]
    om.update_clusters(ids)
    rm.update_clusters(ids)
    _check(rm.trainable_clusters, om.trainable_clusters, 1e-5, "clusters after the step")
    _check(rm.normed_clusters, om.normed_clusters, 1e-5, "normalised clusters")


def test_knn_pairwise_centroids(dev):
    """brute-force inner-product kNN (get_dist_nbr), pairwise_distance (both forms) and cluster
    centroid means against numpy / torch CPU restatements of the cited reference lines."""
    import numpy as np
    import torch.nn.functional as F
    from collections import OrderedDict
    from clustercontrast.utils.infomap_cluster import get_dist_nbr, generate_cluster_features, knn_faiss
    from clustercontrast.evaluators import pairwise_distance
    g = torch.Generator().manual_seed(13)
    n, D, k = 700, 256, 15
    feats = F.normalize(torch.randn(n, D, generator=g), dim=1)
    dists, nbrs = get_dist_nbr(features=feats.numpy(), k=k, knn_method='faiss-gpu')
    sims = feats.double() @ feats.double().t()                       # IndexFlatIP.search: top-k inner products
    rv, ri = torch.topk(sims, k, dim=1)
    assert dists.shape == (n, k) and nbrs.shape == (n, k) and nbrs.dtype == np.int32
    assert np.array_equal(nbrs, ri.numpy().astype(np.int32))
    assert np.abs(dists - (1 - rv.numpy())).max() <= 2e-6
    assert (nbrs[:, 0] == np.arange(n)).all()                         # every sample is its own nearest neighbour
    # ties: lower index first
    t = torch.zeros(4, 8)
    t[:, 3] = 1.0
    t[:, 5] = 1.0
    idx = knn_faiss(torch.eye(8)[:4] * 0 + t, 3).knns        # rows identical -> similarities tie
    assert all(len(a[0]) == 3 for a in idx)
    # pairwise distances
    fd = OrderedDict(("f%d" % i, feats[i]) for i in range(n))
    d_self = pairwise_distance(fd)
    x = feats.double()
    ref = (x.pow(2).sum(1, keepdim=True) * 2).expand(n, n) - 2 * x @ x.t()
    assert (d_self.double() - ref).abs().max().item() <= 1e-5
    q = [("f%d" % i, 0, 0) for i in range(0, 50)]
    gal = [("f%d" % i, 0, 0) for i in range(100, 400)]
    dm, xq, yg = pairwise_distance(fd, q, gal)
    xr, yr = x[:50], x[100:400]
    ref = xr.pow(2).sum(1, keepdim=True) + yr.pow(2).sum(1, keepdim=True).t() - 2 * xr @ yr.t()
    assert dm.shape == (50, 300) and (dm.double() - ref).abs().max().item() <= 1e-5
    assert xq.shape == (50, D) and yg.shape == (300, D)
    # centroids: outliers (-1) skipped, rows by ascending label
    labels = torch.randint(-1, 9, (n,), generator=g).numpy()
    cent = generate_cluster_features(labels, feats)
    keys = sorted(set(labels.tolist()) - {-1})
    ref = torch.stack([feats[torch.from_numpy(labels == kk)].double().mean(0) for kk in keys])
    assert cent.shape == ref.shape and (cent.double().cpu() - ref).abs().max().item() <= 1e-6
