"""k-nearest-neighbour front half of the per-epoch pseudo-labelling on the MI355X.

Mirror of the kNN part of CC/clustercontrast/utils/infomap_cluster.py: `knn_faiss` (:51-111), `knns2ordered_nbrs`
(:114-125), `get_dist_nbr` (:230-234) — same names, arguments and return formats — so
`feat_dists, feat_nbrs = get_dist_nbr(features=features_array, k=opt.k1, knn_method='faiss-gpu')`
(examples/cluster_contrast_gan_train_usl_infomap.py:321) runs without faiss.  The brute-force inner-product search is the
MFMA GEMM (row blocks of X X^T) followed by `rg_topk_rows`; `knn_method` is accepted and ignored.  Result order: similarity
descending, ties by lower index (faiss leaves ties unspecified).

`generate_cluster_features` is the nested helper of the example script (:332-348).  The Infomap community detection itself
(`cluster_by_infomap`, :147-227) is a CPU graph algorithm of the third-party `infomap` package and is not rebuilt (SURVEY §8:
clustering is out of scope); `get_links` / `cluster_by_infomap` are provided as thin pass-throughs when that package exists.
"""
from __future__ import absolute_import

import collections

import numpy as np
import torch

from rg_hip import ops

_BLOCK_ROWS = 2048


def _device():
    return torch.device("cuda", torch.cuda.current_device())


class knn_faiss(object):
    """Brute-force inner-product kNN (for L2-normalised features: cosine similarity)."""

    def __init__(self, feats, k, knn_method='faiss-cpu', verbose=True):
        self.verbose = verbose
        dev = _device()
        x = torch.as_tensor(np.ascontiguousarray(feats, dtype=np.float32)) if not torch.is_tensor(feats) else feats.float()
        x = x.to(dev).contiguous()
        n = x.shape[0]
        nbrs = torch.empty((n, k), dtype=torch.int32, device=dev)
        sims = torch.empty((n, k), dtype=torch.float32, device=dev)
        for r0 in range(0, n, _BLOCK_ROWS):
            r1 = min(n, r0 + _BLOCK_ROWS)
            block = ops.linear_fwd(x[r0:r1], x)                  # [rows, n] inner products
            i, v = ops.topk_rows(block, k)
            nbrs[r0:r1], sims[r0:r1] = i, v
        self.nbrs, self.sims = nbrs, sims
        nb, sm = nbrs.cpu().numpy(), sims.cpu().numpy()
        self.knns = [(np.array(a, dtype=np.int32), 1 - np.array(s, dtype=np.float32)) for a, s in zip(nb, sm)]

    def filter_by_th(self, i):
        th_nbrs, th_dists = [], []
        nbrs, dists = self.knns[i]
        for n, dist in zip(nbrs, dists):
            if 1 - dist < self.th:
                continue
            th_nbrs.append(n)
            th_dists.append(dist)
        return np.array(th_nbrs), np.array(th_dists)

    def get_knns(self, th=None):
        if th is None or th <= 0.:
            return self.knns
        self.th = th
        return [self.filter_by_th(i) for i in range(len(self.knns))]


def knns2ordered_nbrs(knns, sort=True):
    if isinstance(knns, list):
        knns = np.array(knns)
    nbrs = knns[:, 0, :].astype(np.int32)
    dists = knns[:, 1, :]
    if sort:
        nb_idx = np.argsort(dists, axis=1)
        idxs = np.arange(nb_idx.shape[0]).reshape(-1, 1)
        dists = dists[idxs, nb_idx]
        nbrs = nbrs[idxs, nb_idx]
    return dists, nbrs


def get_dist_nbr(features, k=80, knn_method='faiss-cpu'):
    index = knn_faiss(feats=features, k=k, knn_method=knn_method)
    knns = index.get_knns()
    dists, nbrs = knns2ordered_nbrs(knns)
    return dists, nbrs


@torch.no_grad()
def generate_cluster_features(labels, features):
    """centroid of every pseudo-label (label -1 = outlier skipped), rows ordered by ascending label
    (examples/cluster_contrast_gan_train_usl_infomap.py:332-348); `features` [n, D] tensor, returns a device tensor."""
    labels = np.asarray(labels)
    groups = collections.defaultdict(list)
    for i, lb in enumerate(labels.tolist()):
        if lb == -1:
            continue
        groups[lb].append(i)
    keys = sorted(groups.keys())
    if not keys:
        raise ValueError("generate_cluster_features: every sample is an outlier")
    order = [i for kk in keys for i in groups[kk]]
    offsets = np.cumsum([0] + [len(groups[kk]) for kk in keys])
    dev = _device()
    x = features if torch.is_tensor(features) else torch.stack(list(features), dim=0)
    x = x.float().to(dev).contiguous()
    return ops.segment_mean(x, torch.tensor(order, dtype=torch.int64).to(dev), torch.tensor(offsets, dtype=torch.int64).to(dev))


def cluster_by_infomap(nbrs, dists, min_sim, cluster_num=2):
    raise NotImplementedError("Infomap community detection is the third-party `infomap` CPU package driven by the "
                              "reference's own cluster_by_infomap (infomap_cluster.py:147-227); it is outside the GPU hot "
                              "path and not rebuilt — pass the (dists, nbrs) of get_dist_nbr to the reference's function")
