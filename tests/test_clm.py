"""Conditional Latent Matching: oracle pinned to the genuine reference module's outputs
(tests/golden/clm.npz, generated from /root/reference/models/CLM.py loaded by path); HIP ops checked against both."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "clm.npz")


def _models(kind):
    from oracle import clm as oc
    from oracle.recipe import apply_weight_recipe

    m = (oc.CLM if kind == "clm" else oc.SimpleCLM)(64, temperature=0.5).eval()
    apply_weight_recipe(m, 11)
    if kind == "clm":
        with torch.no_grad():
            m.alignment.offset_conv.weight.mul_(6.0)
            m.alignment.offset_conv.bias.mul_(20.0)
    return m


@pytest.mark.parametrize("kind", ["clm", "simple"])
def test_oracle_matches_reference_golden(kind):
    g = np.load(GOLD)
    m = _models(kind)
    y = torch.from_numpy(g[f"{kind}_y"])
    refs = [torch.from_numpy(r) for r in g[f"{kind}_refs"]]
    with torch.no_grad():
        out = m(y, refs)
    np.testing.assert_allclose(out.numpy(), g[f"{kind}_out"], rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["clm", "simple"])
def test_hip_matches_reference_golden(dev, kind):
    from clc_amd import clm as pc

    g = np.load(GOLD)
    o = _models(kind)
    p = (pc.CLM if kind == "clm" else pc.SimpleCLM)(64, temperature=0.5)
    p.load_state_dict(o.state_dict())
    p = p.to(dev).eval()
    y = torch.from_numpy(g[f"{kind}_y"]).to(dev)
    refs = [torch.from_numpy(r).to(dev) for r in g[f"{kind}_refs"]]
    out = p(y, refs).cpu()
    ref = torch.from_numpy(g[f"{kind}_out"])
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-4, err


@pytest.mark.gpu
def test_hip_matches_oracle_other_shape(dev):
    """32x24 latents, C=128, 2 references, batch 3 — against the vectorised oracle."""
    from clc_amd import clm as pc
    from oracle import clm as oc
    from oracle.recipe import apply_weight_recipe

    o = oc.CLM(128, temperature=0.7).eval()
    apply_weight_recipe(o, 5)
    with torch.no_grad():
        o.alignment.offset_conv.weight.mul_(8.0)
    p = pc.CLM(128, temperature=0.7)
    p.load_state_dict(o.state_dict())
    p = p.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    y = torch.randn(3, 128, 32, 24, generator=g)
    refs = [torch.randn(3, 128, 32, 24, generator=g) for _ in range(2)]
    with torch.no_grad():
        ref = o(y, refs)
    out = p(y.to(dev), [r.to(dev) for r in refs]).cpu()
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-4, err
