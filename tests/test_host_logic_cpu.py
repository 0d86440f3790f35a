"""Host-side logic that needs no GPU: state_dict contract, checkpoint loading, flat arenas, loss bookkeeping."""
import numpy as np
import pytest
import torch


def test_state_dict_contract_matches_oracle():
    from clc_amd import models as pm
    from oracle import graph as og

    for mk_p, mk_o in ((lambda: pm.CLC(N=64, num_ref_frames=1), lambda: og.CLC(N=64, num_ref_frames=1)),
                       (lambda: pm.CLC(N=64, num_ref_frames=3), lambda: og.CLC(N=64, num_ref_frames=3)),
                       (lambda: pm.TCM(N=64), lambda: og.TCM(N=64))):
        p, o = mk_p(), mk_o()
        sp, so = p.state_dict(), o.state_dict()
        assert list(sp.keys()) == list(so.keys())
        assert all(sp[k].shape == so[k].shape and sp[k].dtype == so[k].dtype for k in so)
        assert [n for n, _ in p.named_parameters() if n.endswith(".quantiles")] == ["entropy_bottleneck.quantiles"]


def test_load_state_dict_reference_semantics():
    from clc_amd import models as pm
    from oracle import graph as og
    from oracle.recipe import apply_weight_recipe

    o = og.CLC(N=64, num_ref_frames=1)
    apply_weight_recipe(o, 1)
    o.update(force=True)
    sd = {("module." + k): v for k, v in o.state_dict().items()}
    sd = {k[len("module."):]: v for k, v in sd.items()}      # train_CLC.py:458-464 strips the DataParallel prefix
    sd["some.unknown.key"] = torch.zeros(3)                    # CLC.load_state_dict filters unknown keys (CLC_run.py:599-618)
    p = pm.CLC(N=64, num_ref_frames=1)
    p.load_state_dict(sd)
    for k, v in o.state_dict().items():
        assert torch.equal(p.state_dict()[k], v), k
    # conv weights stay channels_last ([Cout][kh][kw][Cin]) after loading
    assert p.g_a[0].conv1.weight.is_contiguous(memory_format=torch.channels_last)
    assert p.gaussian_conditional.quantized_cdf.shape == (64, 3133)
    assert p.update() is False and p.update(force=True) is True
    t = pm.TCM(N=64)
    with pytest.raises(RuntimeError):
        t.load_state_dict({"g_a.0.conv1.weight": torch.zeros(128, 3, 3, 3)})   # strict, like tcm.py:488


def test_config_validation():
    from clc_amd import models as pm

    with pytest.raises(ValueError):
        pm.CLC(config=[2, 2, 2, 2, 2, 4], N=64)
    with pytest.raises(ValueError):
        pm.TCM(N=64, drop_path_rate=0.1)
    m = pm.CLC(N=64, num_ref_frames=2)
    assert m.ref_feature_adapter[0].weight.shape == (128, 640, 1, 1)


def test_flat_arena_keeps_shapes_and_strides():
    from clc_amd.train import FlatArena

    a = torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    b = torch.randn(5)
    c = torch.randn(6, 7)
    ar = FlatArena([a, b, c])
    for t, v in zip((a, b, c), ar.views):
        assert v.shape == t.shape and v.stride() == t.stride()
        v.copy_(t)
        assert torch.equal(v, t)
    assert ar.flat.numel() % 64 == 0 and ar.offsets == [0, 320, 384]
    ar.flat.zero_()
    assert float(ar.views[0].abs().sum()) == 0.0


def test_eval_helpers_match_reference_definitions():
    """pad/crop/psnr as eval_CLC.py:133-166."""
    from oracle.loss import compute_psnr, crop, pad

    x = torch.rand(1, 3, 200, 300)
    xp, padding = pad(x, 128)
    assert xp.shape == (1, 3, 256, 384) and padding == (42, 42, 28, 28)
    assert torch.equal(crop(xp, padding), x)
    assert abs(compute_psnr(x, x + 0.1) - 20.0) < 1e-3
