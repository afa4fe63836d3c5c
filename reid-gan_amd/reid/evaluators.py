"""Evaluation distances and the cascade re-ranking stage of FD-GAN's evaluators on the MI355X.

Mirror of the numeric part of FD-GAN-master/reid/evaluators.py: `pairwise_distance` (:76-98), `extract_embeddings`
(:19-43) and `CascadeEvaluator` (:183-227, what `baseline.py:103-104` — BASELINE config 1 — calls).  The loader loop
`extract_features`, `evaluate_all` and the CMC / mAP metrics are host code and stay the reference's (inherited at the bottom of
this file when its tree sits behind this one on sys.path).

MI355X-first restatement of the second stage: the reference scores one query at a time (one embedding-network call on a
[1, 2048] probe against its top-k gallery rows, 3 368 calls for Market-1501).  Here the first-stage ranking is a device
top-k of the distance matrix (`rg_topk_rows` on the negated distances, ties by lower index like a stable argsort), the
top-k gallery rows of ALL queries are gathered into one [Q * k, 2048] operand and the embedding network (eltwise
(x1 - x2)^2 -> BatchNorm1d -> Linear) runs once.  The merge of the two stages (:215-225) is unchanged arithmetic.
"""
from __future__ import print_function, absolute_import

import numpy as np
import torch

from clustercontrast.evaluators import _dist_block
from rg_hip import ops


def _device():
    return torch.device("cuda", torch.cuda.current_device())


def pairwise_distance(features, query=None, gallery=None, metric=None):
    if metric is not None:
        raise NotImplementedError("pairwise_distance: learned metrics (metric.transform) are not on the GAN path")
    dev = _device()
    if query is None and gallery is None:
        n = len(features)
        x = torch.cat(list(features.values())).view(n, -1).float().to(dev).contiguous()
        return _dist_block(x, x, 2.0, False).cpu()
    x = torch.cat([features[f].unsqueeze(0) for f, _, _ in query], 0)
    y = torch.cat([features[f].unsqueeze(0) for f, _, _ in gallery], 0)
    xd = x.view(x.size(0), -1).float().to(dev).contiguous()
    yd = y.view(y.size(0), -1).float().to(dev).contiguous()
    return _dist_block(xd, yd, 1.0, True).cpu()


def rank_topk(distmat, k):
    """indices [Q, k] of the k smallest distances per row, ascending, ties by lower column (device top-k of -dist)"""
    d = torch.as_tensor(distmat, dtype=torch.float32).to(_device()).contiguous()
    idx, _ = ops.topk_rows(ops.scale(d, -1.0), int(k))
    return idx.long()


def extract_embeddings(model, features, alpha, query=None, topk_gallery=None, rerank_topk=0, print_freq=500,
                       topk_indices=None, gallery=None):
    """pairwise scores [Q * rerank_topk, C] of every query against its top-k gallery entries.  Reference signature
    (`topk_gallery`: per query, the list of gallery (fname, pid, cam) triples); `topk_indices` [Q, k] + `gallery` is the
    device form used by CascadeEvaluator (no per-query Python lists)."""
    model.eval()
    dev = _device()
    probe = torch.cat([features[f].unsqueeze(0) for f, _, _ in query], 0).float()
    Q = probe.shape[0]
    if topk_indices is not None:
        gal = torch.cat([features[f].unsqueeze(0) for f, _, _ in gallery], 0).float().to(dev)
        idx = topk_indices.to(dev).reshape(-1)
        x2 = gal.index_select(0, idx)
    else:
        x2 = torch.cat([features[f].unsqueeze(0) for i in range(Q) for f, _, _ in topk_gallery[i]], 0).float().to(dev)
    probe = probe.to(dev)
    x1 = probe.view(Q, 1, -1).expand(Q, rerank_topk, probe.shape[1]).reshape(Q * rerank_topk, -1).contiguous()
    with torch.no_grad():
        score = model(x1, x2.contiguous())
    return score.view(Q * rerank_topk, -1)


class CascadeEvaluator(object):
    def __init__(self, base_model, embed_model, embed_dist_fn=None):
        super(CascadeEvaluator, self).__init__()
        self.base_model = base_model
        self.embed_model = embed_model
        self.embed_dist_fn = embed_dist_fn

    def second_stage(self, distmat, features, query, gallery, alpha=0, rerank_topk=75):
        """first-stage distances [Q, G] -> merged two-stage distances (numpy, as the reference leaves them), :202-225"""
        distmat = np.array(torch.as_tensor(distmat).cpu().numpy(), dtype=np.float32)
        Q, G = distmat.shape
        k = min(int(rerank_topk), G)
        top = rank_topk(distmat, k)
        embeddings = extract_embeddings(self.embed_model, features, alpha, query=query, rerank_topk=k, topk_indices=top,
                                        gallery=gallery)
        if self.embed_dist_fn is not None:
            embeddings = self.embed_dist_fn(embeddings.data)
        emb = torch.as_tensor(embeddings).detach().float().cpu().numpy().reshape(Q, k)
        top_np = top.cpu().numpy()
        rows = np.arange(Q)[:, None]
        if k < G:
            # the first gallery entry OUTSIDE the top-k (:222 `indices[rerank_topk]`): smallest first-stage distance of the rest
            rest = distmat.copy()
            rest[rows, top_np] = np.inf
            nxt = rest.min(axis=1)
        distmat[rows, top_np] = emb
        if k < G:
            bar = emb.max(axis=1)
            gap = np.maximum(bar + 1.0 - nxt, 0.0).astype(np.float32)
            outside = np.ones_like(distmat, dtype=bool)
            outside[rows, top_np] = False
            distmat += outside * gap[:, None]
        return distmat

    def evaluate(self, data_loader, query, gallery, alpha=0, cache_file=None, rerank_topk=75, second_stage=True, dataset=None,
                 top1=True):
        features, _ = extract_features(self.base_model, data_loader)          # noqa: F821  (reference host loop, inherited)
        distmat = pairwise_distance(features, query, gallery)
        print("First stage evaluation:")
        if second_stage:
            evaluate_all(distmat, query=query, gallery=gallery, dataset=dataset, top1=top1)      # noqa: F821
            distmat = self.second_stage(distmat, features, query, gallery, alpha, rerank_topk)
            print("Second stage evaluation:")
        return evaluate_all(distmat, query, gallery, dataset=dataset, top1=top1)                # noqa: F821


# `extract_features`, `evaluate_all`, `Evaluator`, CMC / mAP are the reference's own host code (rg_hip/overlay.py)
from rg_hip.overlay import inherit as _rg_inherit  # noqa: E402
_rg_inherit(globals())
