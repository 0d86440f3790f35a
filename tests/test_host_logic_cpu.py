"""Host-side logic that needs no GPU: state_dict contract, checkpoint loading, flat arenas, loss bookkeeping."""
import os

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


def test_compat_install_resolves_reference_import_surface():
    """clc_amd.compat.install(): every module-level import of train_CLC.py:17-26, eval_CLC.py:1-17 and models/CLC_run.py:1-20 that
    belongs to the path (not torchvision / tensorboard / the dataset pipeline) resolves, to classes of this package."""
    import subprocess
    import sys

    code = r"""
import sys
import clc_amd.compat as c
done = c.install()
from compressai.datasets import ImageFolder
from compressai.zoo import models as zoo
from pytorch_msssim import ms_ssim
from models import TCM, CLC
from compressai.entropy_models import EntropyBottleneck, GaussianConditional
from compressai.ans import BufferedRansEncoder, RansDecoder
from compressai.models import CompressionModel
from compressai.layers import (AttentionBlock, ResidualBlock, ResidualBlockUpsample, ResidualBlockWithStride, conv3x3, subpel_conv3x3)
from timm.models.layers import trunc_normal_, DropPath
import clc_amd.models, clc_amd.entropy_models, clc_amd.ans, clc_amd.layers
assert CLC is clc_amd.models.CLC and TCM is clc_amd.models.TCM and zoo["clc"] is CLC
assert EntropyBottleneck is clc_amd.entropy_models.EntropyBottleneck and RansDecoder is clc_amd.ans.RansDecoder
assert ResidualBlockWithStride is clc_amd.layers.ResidualBlockWithStride and issubclass(CLC, CompressionModel)
m = CLC(N=64, num_ref_frames=1)          # constructs on the CPU (kernels only run on the GPU)
assert sum(p.numel() for p in m.parameters()) == 70643852 or round(sum(p.numel() for p in m.parameters()) / 1e6, 2) == 70.64
assert isinstance(DropPath(0.0), __import__("torch").nn.Identity)
print("ok", len(done))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stderr[-2000:]


def test_host_tables_follow_load_state_dict_and_update():
    """The coder's host copy of the CDF tables (EntropyModel.host_tables, one D2H copy per model instead of CLC_run.py:654-656's
    .tolist() per call) must not survive a change of the tables: update() replaces the buffers, load_state_dict copies INTO them."""
    from clc_amd.entropy_models import GaussianConditional
    from clc_amd.models.clc import get_scale_table

    a = GaussianConditional(None)
    a.update_scale_table(get_scale_table())
    cdf_a = a.host_tables()[0].copy()
    assert a.host_tables()[0] is a.host_tables()[0]            # cached between calls
    b = GaussianConditional(None)
    b.update_scale_table(get_scale_table(min=0.2, max=64, levels=64))   # other tables, same shapes? no: narrower -> other width
    sd = b.state_dict()
    c = GaussianConditional(None)
    c.update_scale_table(get_scale_table())
    before = c.host_tables()
    assert (before[0] == cdf_a).all()
    for n in ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"):   # the reference resizes the buffers before loading (CLC_run.py:599-618)
        getattr(c, n).resize_(sd[n].shape)
    c.load_state_dict(sd)
    after = c.host_tables()
    assert after[0].shape == tuple(sd["_quantized_cdf"].shape) and (after[0] == sd["_quantized_cdf"].numpy()).all()
    assert (after[1] == sd["_cdf_length"].numpy()).all() and (after[2] == sd["_offset"].numpy()).all()
    # in-place edit of a buffer (no load_state_dict hook involved): the version counters catch it
    c._offset.add_(1)
    assert (c.host_tables()[2] == sd["_offset"].numpy() + 1).all()
    # update(force) replaces the buffer objects
    c.update_scale_table(get_scale_table(), force=True)
    assert (c.host_tables()[0] == cdf_a).all()


def test_container_header_errors_and_kernel_tag():
    """clc_amd.codec container: a blob shorter than its header raises ValueError (not struct.error); the header records which generation
    of context-model kernels encoded the image (the decoder must reproduce the encoder's float context bit for bit)."""
    from clc_amd import codec

    blob = codec.pack([[b"yyyy"], [b"zz"]], (4, 4), (256, 256), n_refs=1)
    strings, shape, meta = codec.unpack(blob)
    assert strings == [[b"yyyy"], [b"zz"]] and tuple(shape) == (4, 4) and meta["image_hw"] == (256, 256) and meta["n_refs"] == 1
    tag = codec.kernel_config_tag()
    assert 0 < tag < 128 and meta["kernel_config_tag"] == tag and meta["same_kernel_config"]
    for bad in (b"", b"CLC1", blob[:23], b"XXXX" + blob[4:], blob + b"!"):
        with pytest.raises(ValueError):
            codec.unpack(bad)
    # a container of another kernel generation — or of a build from before the tag existed (0) — is REFUSED where decoding starts:
    # the context model would leave the encoder's bit-exact means / scales and the arithmetic decoder would run off silently
    for other in (0, tag + 1):
        old = blob[:7] + bytes([other]) + blob[8:]
        with pytest.raises(codec.KernelConfigMismatch):
            codec.unpack(old)
        with pytest.raises(codec.KernelConfigMismatch):
            codec.unpack_item(old)
        s2, _, m2 = codec.unpack(old, strict=False)      # inspection stays possible
        assert s2 == strings and m2["kernel_config_tag"] == other and not m2["same_kernel_config"]
        with pytest.raises(codec.KernelConfigMismatch):
            codec.check_kernel_config(m2)
    # order-affecting tuning keys are part of the tag: the reduced-precision mode (key 14) and the attention tiling of the forward
    # pass (key 16, bits 1 and 4) select kernels that sum in another order; bit 2 of key 16 is a backward kernel and is not
    from clc_amd import lib

    L = lib.load()
    # keys 22 (halo kernel: same bits), 23 (Winograd kernels) and 24 (split-bf16 filter gradients) reach training launches only — recorded forwards, data
    # and filter gradients — never a launch of the codec path: not part of the tag
    for key, val, changes in ((14, 1, True), (16, 7, True), (16, 1, False), (4, 256, True), (13, 0, False), (22, 0, False), (23, 0, False), (24, 0, False)):
        prev = L.clc_set_tuning(key, val)
        try:
            t2 = codec.kernel_config_tag()
            assert (t2 != tag) == changes and (t2 >= 128) == changes, (key, val, t2)
            if changes:
                with pytest.raises(codec.KernelConfigMismatch):
                    codec.unpack(blob)                    # written under the default tuning, read under another
        finally:
            L.clc_set_tuning(key, prev)
    assert codec.kernel_config_tag() == tag and L.clc_get_tuning(16) == 3 and L.clc_get_tuning(23) == 7 and L.clc_get_tuning(24) == 3
    assert L.clc_set_tuning(25, 1) < 0        # (one past the last key)
    # the state is captured WHEN THE KERNELS RAN (compress() returns it as `kernel_config`), not when pack() is called: an image encoded
    # under a temporary non-default state and packed after the default was restored carries the encoding state — as a version-2 header
    # with the full 32-bit hash beside the 7-bit tag (two non-default states cannot be mistaken for each other)
    default_cfg = codec.kernel_config()
    assert default_cfg[0] == tag
    prev = L.clc_set_tuning(14, 1)
    try:
        cfg_a = codec.kernel_config()
    finally:
        L.clc_set_tuning(14, prev)
    prev = L.clc_set_tuning(4, 256)
    try:
        cfg_b = codec.kernel_config()
    finally:
        L.clc_set_tuning(4, prev)
    assert cfg_a[0] >= 128 and cfg_b[0] >= 128 and cfg_a[1] != cfg_b[1] and cfg_a[1] != default_cfg[1]
    item = {"strings": [[b"yyyy"], [b"zz"]], "shape": (4, 4), "kernel_config": cfg_a}
    blob2 = codec.pack_item(item, (256, 256), n_refs=1)           # packed under the DEFAULT state, encoded under state a
    assert blob2[4] == 2 and len(blob2) == 28 + 6 and blob[4] == 1 and len(blob) == 24 + 6
    with pytest.raises(codec.KernelConfigMismatch):
        codec.unpack(blob2)
    _, _, m3 = codec.unpack(blob2, strict=False)
    assert m3["kernel_config_tag"] == cfg_a[0] and m3["kernel_config_hash"] == cfg_a[1] and not m3["same_kernel_config"]
    prev = L.clc_set_tuning(14, 1)
    try:
        s4, _, m4 = codec.unpack(blob2)                           # decodes under the state that encoded
        assert s4 == item["strings"] and m4["same_kernel_config"]
        # same 7-bit tag, other hash -> refused (forged: state b's hash under state a's tag)
        forged = blob2[:24] + cfg_b[1].to_bytes(4, "little") + blob2[28:]
        with pytest.raises(codec.KernelConfigMismatch):
            codec.unpack(forged)
    finally:
        L.clc_set_tuning(14, prev)
    for bad in (blob2[:27], blob2 + b"!"):
        with pytest.raises(ValueError):
            codec.unpack(bad, strict=False)
