"""Reference retrieval core: exact k-nearest-neighbour search and cluster representatives on the GPU against
the reference's own host computation (sklearn NearestNeighbors(ball_tree) / the member-closest-to-centre rule,
/root/reference/dataloader_ref_cluster.py:64,106-146,160-162)."""
import numpy as np
import pytest
import torch


def _bank(n, d, seed):
    rng = np.random.default_rng(seed)
    return rng.normal(size=(n, d)).astype(np.float32)


def test_host_reference_rule_self_check():
    """the oracle used below (sklearn ball_tree) agrees with a brute-force fp64 search — pins the checker."""
    from sklearn.neighbors import NearestNeighbors

    bank, q = _bank(500, 64, 0), _bank(7, 64, 1)
    _, idx = NearestNeighbors(n_neighbors=3, algorithm="ball_tree").fit(bank).kneighbors(q)
    d = ((q[:, None, :].astype(np.float64) - bank[None].astype(np.float64)) ** 2).sum(-1)
    assert np.array_equal(idx, np.argsort(d, axis=1, kind="stable")[:, :3])


@pytest.mark.gpu
@pytest.mark.parametrize("n,d,k", [(5000, 2048, 3), (777, 256, 1), (64, 128, 8)])
def test_kneighbors_matches_sklearn(dev, n, d, k):
    from sklearn.neighbors import NearestNeighbors

    from clc_amd.retrieval import ReferenceIndex

    bank, q = _bank(n, d, 2), _bank(33, d, 3)
    keys = {i: f"ref_{i:05d}.png" for i in range(n)}
    dist_ref, idx_ref = NearestNeighbors(n_neighbors=k, algorithm="ball_tree").fit(bank).kneighbors(q)
    ix = ReferenceIndex(bank, keys, n_refs=k, device=dev)
    dist, idx = ix.kneighbors(q)
    assert np.array_equal(idx.cpu().numpy(), idx_ref), "neighbour sets / order differ from sklearn ball_tree"
    assert np.allclose(dist.cpu().numpy(), dist_ref, rtol=2e-4, atol=2e-3)
    assert ix.query(q[:2]) == [[keys[j] for j in row] for row in idx_ref[:2]]
    # a query that IS a bank row finds itself first, at distance ~0
    d0, i0 = ix.kneighbors(bank[17])
    assert int(i0[0, 0]) == 17 and float(d0[0, 0]) < 0.05


@pytest.mark.gpu
def test_cluster_representatives_match_reference_rule(dev):
    """given the clustering (labels, centres), the representatives are the members closest to their centres
    (dataloader_ref_cluster.py:123-133) and the thinned bank answers queries like sklearn on the same representatives."""
    from sklearn.cluster import MiniBatchKMeans
    from sklearn.neighbors import NearestNeighbors

    from clc_amd.retrieval import ReferenceIndex

    bank = _bank(3000, 256, 5)
    keys = {i: f"k{i}" for i in range(len(bank))}
    km = MiniBatchKMeans(n_clusters=40, random_state=42, batch_size=1000)
    labels = km.fit_predict(bank)
    want, want_keys = [], {}
    for i in range(40):
        members = np.where(labels == i)[0]
        if len(members) == 0:
            continue
        j = members[np.argmin(np.linalg.norm(bank[members] - km.cluster_centers_[i], axis=1))]
        want_keys[len(want)] = keys[j]
        want.append(j)
    ix = ReferenceIndex(bank, keys, n_refs=2, device=dev)
    ix.cluster_features(40, labels=labels, centers=km.cluster_centers_)
    assert ix.representatives == [int(j) for j in want] and ix.feature_to_key == want_keys
    q = _bank(9, 256, 6)
    _, idx_ref = NearestNeighbors(n_neighbors=2, algorithm="ball_tree").fit(bank[want]).kneighbors(q)
    assert np.array_equal(ix.kneighbors(q)[1].cpu().numpy(), idx_ref)
    # the default path runs the reference's own estimator call (same random_state): same representatives
    ix2 = ReferenceIndex(bank, keys, n_clusters=40, n_refs=2, device=dev)
    assert ix2.representatives == ix.representatives
