"""Patch-matching numeric kernels: oracle pinned to the genuine reference functions' outputs
(tests/golden/patch_matching.npz, AST-extracted from /root/reference/models/Patch_Matching.py); HIP ops against both."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "patch_matching.npz")


def test_oracle_matches_reference_golden():
    from oracle import patch_matching as opm

    g = {k: torch.from_numpy(v) for k, v in np.load(GOLD).items()}
    assert torch.equal(opm.create_gaussian_masks(64, 96, 16, 16), g["mask"])
    corr = opm.L2_or_pearson_corr(g["q"], g["r"], 16, 16)
    np.testing.assert_allclose(corr.numpy(), g["corr"].numpy(), rtol=0, atol=1e-5)
    fin = opm.SI_Finder_at_Image_Domain(g["x_dec"], g["y_img"], 16, 16, g["y_dec"], mask=g["mask"])
    assert torch.equal(fin, g["finder"])
    wr = opm.SI_Wraper(g["corr"] * g["mask"], 16, 16, 24, g["y_img"][0:1], k=3, temperature=15)
    np.testing.assert_allclose(wr.numpy(), g["wraper_k3"].numpy(), rtol=0, atol=1e-6)
    st = opm.SI_Wraper(g["corr"] * g["mask"], 16, 16, 24, g["y_img"][0:1], k=3, temperature=15, is_stack=True)
    assert torch.equal(st, g["wraper_k3_stack"])
    x = g["x_dec"][0:1] * 255
    np.testing.assert_allclose(opm.rgb_transform(opm.reduce_mean_and_std_normalize_images(x))[:, 2].numpy(),
                               (0.5 * ((x[:, 0] - 93.70454143384742) / 73.56493292844912 + (x[:, 2] - 94.84678088809876) / 76.74838442810665)).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_matches_reference_golden(dev):
    from clc_amd import patch_matching as pm

    g = {k: torch.from_numpy(v) for k, v in np.load(GOLD).items()}
    mask = pm.create_gaussian_masks(64, 96, 16, 16, device=dev)
    assert (mask.cpu() - g["mask"]).abs().max().item() < 1e-6
    # normalise + colour transform
    q_in = g["x_dec"][0:1].reshape(1, 3, 4, 16, 6, 16).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3, 16, 16).contiguous()
    q = pm.rgb_transform_normalized(q_in.to(dev), 255.0)
    assert (q.cpu() - g["q"]).abs().max().item() < 1e-4
    # Pearson map: the kernels accumulate the CENTRED products (patch - its mean) and take the window variance in double, so the
    # map agrees with the genuine function's output (itself within 5e-6 of an fp64 evaluation) to 2e-5
    corr = pm.L2_or_pearson_corr(g["q"].to(dev), g["r"].to(dev), 16, 16)
    assert corr.shape == g["corr"].shape
    assert (corr.cpu() - g["corr"]).abs().max().item() < 2e-5
    # the matches themselves: same winning positions -> identical copied patches (a tie within 2e-5 may pick the other position)
    fin = pm.SI_Finder_at_Image_Domain(g["x_dec"].to(dev), g["y_img"].to(dev), 16, 16, g["y_dec"].to(dev), mask=mask)
    agree = (fin.cpu() == g["finder"]).float().mean().item()
    assert agree > 0.999, agree
    wr = pm.SI_Wraper((g["corr"] * g["mask"]).to(dev), 16, 16, 24, g["y_img"][0:1].to(dev), k=3, temperature=15)
    assert (wr.cpu() - g["wraper_k3"]).abs().max().item() < 1e-5
    # is_stack=True: the k candidates unweighted, candidate-major along the channels (Patch_Matching.py:235-236)
    st = pm.SI_Wraper((g["corr"] * g["mask"]).to(dev), 16, 16, 24, g["y_img"][0:1].to(dev), k=3, temperature=15, is_stack=True)
    assert st.shape == (1, 9, 64, 96) and torch.equal(st.cpu(), g["wraper_k3_stack"])


@pytest.mark.gpu
def test_hip_pearson_config3_shape(dev):
    """BASELINE config 3 side benchmark shape: 256x256 image, 16x16 patches -> 256 queries, 241x241 map, top-3."""
    from clc_amd import patch_matching as pm
    from oracle import patch_matching as opm
    from oracle.recipe import synthetic_image

    y = synthetic_image(1, 256, 256, 3, smooth=True)
    x = torch.roll(y, shifts=(9, 4), dims=(2, 3))
    q = opm.rgb_transform(opm.reduce_mean_and_std_normalize_images(x.reshape(1, 3, 16, 16, 16, 16).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3, 16, 16) * 255))
    r = opm.rgb_transform(opm.reduce_mean_and_std_normalize_images(y * 255))
    ref = opm.L2_or_pearson_corr(q, r, 16, 16)
    out = pm.L2_or_pearson_corr(q.to(dev), r.to(dev), 16, 16)
    assert out.shape == (1, 256, 241, 241)
    assert (out.cpu() - ref).abs().max().item() < 3e-5
    # arg-max agreement
    a = out.cpu().reshape(256, -1).argmax(1)
    b = ref.reshape(256, -1).argmax(1)
    assert (a == b).float().mean().item() > 0.99
