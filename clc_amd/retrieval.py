"""Reference retrieval — the numeric core of the reference's dataset pipeline on the MI355X.

/root/reference/dataloader_ref_cluster.py:89-146, 241-252 (variant dataloader_CLC.py:110-209, 250-294): every reference image
has a feature vector (ResNet50 pool features there: 2048-d; the pretrained weights are not reachable here, so the extractor is a
pluggable callable), the feature bank is optionally thinned to `n_clusters` representatives (MiniBatchKMeans, then for each
centre the member closest to it) and a query's `n_refs` nearest representatives (Euclidean, exact: sklearn ball_tree) name the
reference frames handed to CLC.forward().  In the reference this runs per sample inside `Dataset.__getitem__` on the host.

Here the bank lives in HBM and a whole batch of queries is answered by one GEMM on the f32-MFMA convolution kernel
(scores = 2 q.r - |r|^2, the 1x1-convolution path of libclc_hip.so) plus the tie-stable top-k kernel of the patch-matching
module (clc_pm_topk: equal scores -> lowest index, like a stable sort of the distances).  Clustering itself stays the
reference's own sklearn call (same estimator, same random_state) so that the representatives are the reference's; the
member-closest-to-centre step runs on the device.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib
from . import ops
from .ops import _L, _stream


class ReferenceIndex:
    def __init__(self, ref_features, feature_to_key: Optional[Dict[int, str]] = None, n_clusters: Optional[int] = None, n_refs: int = 1,
                 device="cuda", extractor: Optional[Callable] = None):
        """ref_features [N, D] (numpy / tensor), feature_to_key {row -> key} as the reference's pickle holds them
        (dataloader_ref_cluster.py:89-104)."""
        if not 1 <= n_refs <= 8:
            raise ValueError("n_refs must be in 1..8")
        self.device = torch.device(device)
        self.n_refs = int(n_refs)
        self.extractor = extractor
        feats = torch.as_tensor(np.asarray(ref_features), dtype=torch.float32)
        if feats.dim() != 2 or feats.shape[1] % 4:
            raise ValueError("ref_features must be [N, D] with D a multiple of 4")
        self.feature_to_key = dict(feature_to_key) if feature_to_key is not None else {i: i for i in range(feats.shape[0])}
        self._set_bank(feats.to(self.device))
        if n_clusters:
            self.cluster_features(int(n_clusters))

    def _set_bank(self, feats):
        self.ref_features = feats.contiguous()
        self.sqnorm = (self.ref_features.double() ** 2).sum(1).float()     # |r|^2 once per bank (double: exact to fp32 rounding)

    # ---- thinning: MiniBatchKMeans representatives (dataloader_ref_cluster.py:106-144)
    def cluster_features(self, n_clusters: int, labels=None, centers=None):
        """labels / centers: precomputed clustering (tests); default: the reference's estimator on the host."""
        if labels is None:
            from sklearn.cluster import MiniBatchKMeans

            km = MiniBatchKMeans(n_clusters=n_clusters, random_state=42, batch_size=1000)
            labels = km.fit_predict(self.ref_features.cpu().numpy())
            centers = km.cluster_centers_
        labels_d = torch.as_tensor(np.asarray(labels), device=self.device, dtype=torch.long)
        centers_d = torch.as_tensor(np.asarray(centers), dtype=torch.float32, device=self.device)
        # distance of every member to ITS centre: |f|^2 - 2 f.c + |c|^2 with the cross term from the MFMA GEMM
        cross = self._scores(self.ref_features, centers_d, (centers_d.double() ** 2).sum(1).float())   # [N, K] = 2 f.c - |c|^2
        d = self.sqnorm - cross.gather(1, labels_d[:, None])[:, 0]
        reps, keys = [], {}
        order = torch.argsort(labels_d, stable=True)
        for i in range(n_clusters):                       # (host loop over clusters as in the reference; the arithmetic is done)
            members = order[labels_d[order] == i]
            if members.numel() == 0:
                continue
            j = int(members[torch.argmin(d[members])])    # ties -> first member in index order, like np.argmin
            keys[len(reps)] = self.feature_to_key[j]
            reps.append(j)
        self.representatives = reps
        self.feature_to_key = keys
        self._set_bank(self.ref_features[torch.as_tensor(reps, device=self.device)])

    # ---- search
    def _scores(self, q, bank, bank_sqnorm):
        """[Q, N] = 2 q.bank^T - |bank|^2 (arg-max == nearest neighbour); the product runs on the 1x1-convolution MFMA kernel."""
        Q, D = q.shape
        x = q.contiguous().view(Q, D, 1, 1)
        with torch.no_grad():
            prod = ops.linear(x, bank, None)              # [Q, N, 1, 1]
        return 2.0 * prod.view(Q, -1) - bank_sqnorm[None, :]

    @torch.no_grad()
    def kneighbors(self, q_features, n_refs: Optional[int] = None):
        """-> (distances [Q, k] ascending, indices [Q, k]) — sklearn NearestNeighbors.kneighbors semantics
        (dataloader_ref_cluster.py:160-161)."""
        k = int(n_refs or self.n_refs)
        q = torch.as_tensor(np.asarray(q_features) if not torch.is_tensor(q_features) else q_features, dtype=torch.float32).to(self.device)
        if q.dim() == 1:
            q = q[None]
        sc = self._scores(q, self.ref_features, self.sqnorm).contiguous()
        Q, N = sc.shape
        val = torch.empty((Q, k), device=self.device, dtype=torch.float32)
        idx = torch.empty((Q, k), device=self.device, dtype=torch.int32)
        _lib.check(_L().clc_pm_topk(sc.data_ptr(), Q, N, k, val.data_ptr(), idx.data_ptr(), _stream()), "clc_pm_topk")
        qn = (q.double() ** 2).sum(1, keepdim=True).float()
        dist = torch.sqrt(torch.clamp(qn - val, min=0.0))
        return dist, idx.long()

    def query(self, q_features, n_refs: Optional[int] = None):
        """-> list (per query) of reference keys, nearest first (dataloader_ref_cluster.py:160-162)."""
        _, idx = self.kneighbors(q_features, n_refs)
        return [[self.feature_to_key[int(j)] for j in row] for row in idx.cpu()]

    def query_images(self, images, n_refs: Optional[int] = None, rotated: bool = False):
        """Reference frames for a whole BATCH of query images in one pass: features (the extractor: e.g. clc_amd.features.ResNet50Features
        [+ PCAProjection]) and exact k-nearest-neighbour search, where the reference runs one ResNet50 forward and one ball-tree query per
        sample inside __getitem__ (dataloader_ref_cluster.py:149-180).
        images: [N, 3, H, W] normalised tensor (or a sequence the extractor accepts one by one).
        rotated: dataloader_CLC.py:186-200's variant — a second query with the image rotated by 90 degrees, neighbours of both merged as
        np.unique(concatenate)[:n_refs] (sorted by INDEX, as the reference does)."""
        if self.extractor is None:
            raise ValueError("no feature extractor was given: pass extractor=clc_amd.features.ResNet50Features() (load torchvision's "
                             "resnet50 state_dict into it; the pretrained file is not reachable from this build)")
        k = int(n_refs or self.n_refs)

        def feats(x):
            if torch.is_tensor(x) and x.dim() == 4:
                return torch.as_tensor(self.extractor(x.to(self.device)))
            return torch.stack([torch.as_tensor(self.extractor(im)) for im in x])

        _, idx = self.kneighbors(feats(images), k)
        if not rotated:
            return [[self.feature_to_key[int(j)] for j in row] for row in idx.cpu()]
        rot = torch.rot90(images, 1, dims=(2, 3)) if torch.is_tensor(images) else [torch.rot90(torch.as_tensor(im), 1, dims=(-2, -1)) for im in images]
        _, idx2 = self.kneighbors(feats(rot), k)
        out = []
        for a, b in zip(idx.cpu().numpy(), idx2.cpu().numpy()):
            merged = np.unique(np.concatenate([a, b]))[:k]
            out.append([self.feature_to_key[int(j)] for j in merged])
        return out
