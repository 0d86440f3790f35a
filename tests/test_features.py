"""SURVEY 8(f)-2, the feature side of reference retrieval: ResNet-50 pool / spatial-pyramid features on the HIP kernels
(clc_amd.features) against the plain-PyTorch restatement (oracle/resnet.py) on seeded weights, PCA projection against sklearn, the
FIFO cache against the reference's semantics, and the batched query path end to end.
Reference: /root/reference/dataloader_ref_cluster.py:41-44, 149-180, 241-261; /root/reference/dataloader_CLC.py:23-40, 110-138, 186-209, 250-294.
"""
import numpy as np
import pytest
import torch


def test_extractor_has_torchvision_names_and_size():
    """23.51 M parameters (ResNet-50 without its fc layer) under torchvision's key names: a torchvision checkpoint loads as is."""
    from clc_amd.features import ResNet50Features
    from oracle import resnet

    o = resnet.seed_weights(resnet.ResNet50(), 0)
    p = ResNet50Features()
    assert sum(q.numel() for q in p.parameters()) == 23508032 == sum(q.numel() for q in o.parameters())
    res = p.load_state_dict(o.state_dict())
    assert not res.missing_keys and not res.unexpected_keys
    for k in ("conv1.weight", "bn1.running_var", "layer1.0.downsample.0.weight", "layer1.0.downsample.1.running_mean", "layer4.2.conv3.weight", "layer3.5.bn2.bias"):
        assert k in p.state_dict(), k


def test_kv_cache_is_the_references_fifo():
    """dataloader_CLC.py:23-40: when full, the OLDEST inserted key goes; get() does not refresh."""
    from clc_amd.features import KVCache

    c = KVCache(max_size=3)
    for k in "abc":
        c.add(k, k.upper())
    assert c.get("a") == "A" and len(c) == 3
    c.add("d", "D")
    assert c.get("a") is None and c.get("b") == "B" and c.get("d") == "D" and len(c) == 3
    c.add("e", "E")
    assert c.get("b") is None and [c.get(k) for k in "cde"] == ["C", "D", "E"]


@pytest.mark.gpu
def test_resnet50_features_vs_plain_torch(dev):
    from clc_amd.features import ResNet50Features
    from oracle import resnet

    o = resnet.seed_weights(resnet.ResNet50(), 3)
    p = ResNet50Features()
    p.load_state_dict(o.state_dict())
    p = p.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    for shape in ((3, 3, 224, 224), (1, 3, 256, 320)):
        x = torch.randn(shape, generator=g)
        with torch.no_grad():
            t_o = o.trunk(x)
            f_o = o(x)
            s_o = resnet.spatial_pyramid_pooling(t_o)
        t_p = p.trunk(x.to(dev))
        assert tuple(t_p.shape) == tuple(t_o.shape)
        e_t = (t_p.cpu() - t_o).abs().max().item() / t_o.abs().max().item()
        f_p, s_p = p(x.to(dev)).cpu(), p.forward_spp(x.to(dev)).cpu()
        assert f_p.shape == (shape[0], 2048) and s_p.shape == (shape[0], 2048 * 21)
        e_f = (f_p - f_o).abs().max().item() / f_o.abs().max().item()
        e_s = (s_p - s_o).abs().max().item() / s_o.abs().max().item()
        print(shape, "layer4 rel err", e_t, "pool", e_f, "spp", e_s)
        assert e_t < 2e-4 and e_f < 2e-4 and e_s < 2e-4, (e_t, e_f, e_s)
    # weights changed in place -> fold() again
    with torch.no_grad():
        p.layer4[2].bn3.weight.mul_(2.0)
        o.layer4[2].bn3.weight.mul_(2.0)
    p.fold()
    with torch.no_grad():
        assert (p(x.to(dev)).cpu() - o(x)).abs().max().item() / o(x).abs().max().item() < 2e-4


@pytest.mark.gpu
def test_pca_projection_and_batched_query(dev):
    """PCA-256 on the device == sklearn's transform; query_images() = features + exact kNN for a whole batch in one call, equal to the
    reference's per-sample path (ResNet features -> ball-tree NearestNeighbors) run on the oracle's features; the rotated-query variant
    merges the two neighbour lists as np.unique(...)[:n_refs] (dataloader_CLC.py:196-200)."""
    from sklearn.decomposition import PCA
    from sklearn.neighbors import NearestNeighbors

    from clc_amd.features import PCAProjection, ResNet50Features
    from clc_amd.retrieval import ReferenceIndex
    from oracle import resnet

    o = resnet.seed_weights(resnet.ResNet50(), 4)
    p = ResNet50Features()
    p.load_state_dict(o.state_dict())
    p = p.to(dev).eval()
    g = torch.Generator().manual_seed(9)
    bank_imgs = torch.randn((24, 3, 96, 96), generator=g)
    with torch.no_grad():
        bank = o(bank_imgs).numpy()
    # PCA
    pca = PCA(n_components=16).fit(bank)
    proj = PCAProjection.from_sklearn(pca, dev)
    got = proj.transform(torch.from_numpy(bank).to(dev)).cpu().numpy()
    want = pca.transform(bank)
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    # batched query: the bank images themselves (slightly perturbed) must retrieve themselves first, then the reference's order
    index = ReferenceIndex(bank, {i: f"ref{i}" for i in range(24)}, n_refs=3, device=dev, extractor=p)
    queries = bank_imgs[:6] + 0.01 * torch.randn((6, 3, 96, 96), generator=g)
    keys = index.query_images(queries)
    with torch.no_grad():
        qf = o(queries).numpy()
    nn_ = NearestNeighbors(n_neighbors=3, algorithm="ball_tree").fit(bank)
    _, idx = nn_.kneighbors(qf)
    assert keys == [[f"ref{j}" for j in row] for row in idx], (keys, idx)
    assert [k[0] for k in keys] == [f"ref{i}" for i in range(6)]
    # rotated variant
    keys_r = index.query_images(queries, rotated=True)
    with torch.no_grad():
        qr = o(torch.rot90(queries, 1, dims=(2, 3))).numpy()
    _, idx2 = nn_.kneighbors(qr)
    want_r = [[f"ref{j}" for j in np.unique(np.concatenate([a, b]))[:3]] for a, b in zip(idx, idx2)]
    assert keys_r == want_r, (keys_r, want_r)
