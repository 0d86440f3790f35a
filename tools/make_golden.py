#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (numbers only — never reference source).

Runs in the BUILD CONTAINER (needs /root/reference); the fixtures it writes are what travels.
  graph_*.npz   outputs of the reference's OWN model classes (imported through tools/ref_shim.py, i.e.
                genuine CLC/TCM/WMSA/Block/ConvTransBlock/SWAtten code over the oracle's restated leaves)
                on recipe weights (oracle/recipe.py) and seeded inputs  -> pins the oracle's graph wiring
  blocks.npz    outputs of the reference's WMSA / Block / SwinBlock / ConvTransBlock classes -> pins window
                attention, shift mask, relative-position indexing, LN/MLP exactly
  rans_kat.json known-answer bitstreams from the pure-Python coder (oracle/rans_py.py) incl. bypass escapes,
                and a digest of the Gaussian CDF tables
Usage: python tools/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import leaves, rans_py  # noqa: E402
from oracle.loss import RateDistortionLoss  # noqa: E402
from oracle.recipe import apply_weight_recipe, synthetic_image  # noqa: E402
from tools import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def summarize(out, x):
    crit = RateDistortionLoss(0.0067)(out, x)
    xh, ly, lz = out["x_hat"], out["likelihoods"]["y"], out["likelihoods"]["z"]
    p = out["para"]
    return {
        "bpp": np.float64(crit["bpp_loss"].item()), "mse": np.float64(crit["mse_loss"].item()), "loss": np.float64(crit["loss"].item()),
        "x_hat_patch": xh[0, :, 96:112, 96:112].numpy(), "x_hat_mean": np.float64(xh.double().mean().item()),
        "x_hat_abs_sum": np.float64(xh.double().abs().sum().item()),
        "lik_y_logsum": np.float64(torch.log(ly.double()).sum().item()), "lik_z_logsum": np.float64(torch.log(lz.double()).sum().item()),
        "lik_y_patch": ly[0, ::40, 4:8, 4:8].numpy(), "lik_z": lz[0, ::16].numpy(),
        "y_patch": p["y"][0, ::32, :8, :8].numpy(), "means_patch": p["means"][0, ::32, :8, :8].numpy(),
        "scales_patch": p["scales"][0, ::32, :8, :8].numpy(),
    }


def graph_goldens(ref):
    for name, R in (("clc_r1", 1), ("clc_r3", 3), ("tcm", 0)):
        torch.manual_seed(0)
        m = (ref.CLC(N=64, num_ref_frames=R) if R else ref.TCM(N=64)).eval()
        apply_weight_recipe(m, 0)
        x = synthetic_image(1, 256, 256, 100, smooth=True)
        refs = [synthetic_image(1, 256, 256, 101 + i, smooth=True) for i in range(R)]
        with torch.no_grad():
            out = m(x, refs) if R else m(x)
        s = summarize(out, x)
        s["n_params"] = np.int64(sum(p.numel() for p in m.parameters()))
        s["n_state"] = np.int64(len(m.state_dict()))
        # which parameters receive a gradient (decides the live set of the fused optimizer / DDP buckets)
        m.zero_grad()
        o2 = m(x, refs) if R else m(x)
        RateDistortionLoss(0.0067)(o2, x)["loss"].backward()
        dormant = sorted({n.split(".")[0] for n, p in m.named_parameters() if p.grad is None})
        s["dormant_prefixes"] = np.array(",".join(dormant))
        s["n_live"] = np.int64(sum(p.numel() for p in m.parameters() if p.grad is not None))
        np.savez_compressed(os.path.join(OUT, f"graph_{name}.npz"), **s)
        print(name, {k: (float(v) if np.ndim(v) == 0 and v.dtype.kind == "f" else None) for k, v in s.items() if np.ndim(v) == 0 and v.dtype.kind == "f"},
              "dormant:", dormant, "live:", int(s["n_live"]))


def block_goldens(ref):
    mod = sys.modules["models.CLC_run"]
    out = {}
    g = torch.Generator().manual_seed(7)
    for typ in ("W", "SW"):
        for (C, hd, ws, H, W) in ((64, 8, 8, 16, 24), (64, 32, 4, 8, 8), (128, 16, 8, 16, 16)):
            tag = f"{typ}_{C}_{hd}_{ws}_{H}x{W}"
            x = torch.randn(2, H, W, C, generator=g)
            msa = mod.WMSA(C, C, hd, ws, typ).eval()
            apply_weight_recipe(msa, 1)
            blk = mod.Block(C, C, hd, ws, 0, typ).eval()
            apply_weight_recipe(blk, 2)
            with torch.no_grad():
                out[f"wmsa_{tag}_x"] = x.numpy()
                out[f"wmsa_{tag}_y"] = msa(x).numpy()
                out[f"block_{tag}_y"] = blk(x).numpy()
    xs = torch.randn(1, 128, 16, 16, generator=g)
    ctb = mod.ConvTransBlock(64, 64, 16, 8, 0, "SW").eval()
    apply_weight_recipe(ctb, 3)
    swa = mod.SWAtten(384, 384, 16, 8, 0, inter_dim=128).eval()
    apply_weight_recipe(swa, 4)
    xa = torch.randn(1, 384, 16, 16, generator=g)
    with torch.no_grad():
        out["ctb_x"], out["ctb_y"] = xs.numpy(), ctb(xs).numpy()
        out["swatten_x"], out["swatten_y"] = xa.numpy(), swa(xa).numpy()
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)
    print("blocks:", len(out), "arrays")


def block_large_goldens(ref):
    """The genuine Block (models/CLC_run.py:172-193) at the size where the product switches to its fused large-map launches
    (ln1 + qkv, window attention, the wave-private 1x1 kernel, the fused LayerNorm + MLP; M = 2 x 128 x 128 = 32 768 tokens, C = 64,
    head_dim 8, window 8 — the g_a / g_s blocks of the 128 x 128 maps): forward, input gradient and every parameter gradient under a seeded
    dy -> tests/golden/block_large.npz.  x / dy are re-drawn by the test from the stored seeds (their sha256 is stored); kept are three
    8 x 8 x 64 patches + f64 sums of the output and of the input gradient, and the parameter gradients whole (51 k numbers per block)."""
    mod = sys.modules["models.CLC_run"]
    out = {"seed_x": np.int64(20250), "seed_dy": np.int64(20251), "shape": np.array([2, 128, 128, 64], dtype=np.int64)}
    x0 = torch.randn(2, 128, 128, 64, generator=torch.Generator().manual_seed(20250))
    dy = torch.randn(2, 128, 128, 64, generator=torch.Generator().manual_seed(20251))
    out["x_sha256"] = np.array(hashlib.sha256(x0.numpy().tobytes()).hexdigest())
    out["dy_sha256"] = np.array(hashlib.sha256(dy.numpy().tobytes()).hexdigest())
    patches = ((0, 0, 0), (0, 60, 100), (1, 120, 120))     # (image, row, col): a corner window, an interior one, the shifted wrap-around corner
    out["patch_origins"] = np.array(patches, dtype=np.int64)
    for typ in ("W", "SW"):
        blk = mod.Block(64, 64, 8, 8, 0, typ).train()
        apply_weight_recipe(blk, 2)
        x = x0.clone().requires_grad_(True)
        y = blk(x)
        y.backward(dy)
        for k, (b, r, c) in enumerate(patches):
            out[f"{typ}_y_patch{k}"] = y.detach()[b, r:r + 8, c:c + 8].numpy()
            out[f"{typ}_dx_patch{k}"] = x.grad[b, r:r + 8, c:c + 8].numpy()
        for nm, t in (("y", y.detach()), ("dx", x.grad)):
            out[f"{typ}_{nm}_sum"] = np.float64(t.double().sum().item())
            out[f"{typ}_{nm}_abs_sum"] = np.float64(t.double().abs().sum().item())
            out[f"{typ}_{nm}_row_sums"] = t.double().sum(dim=(1, 2)).numpy()          # [2, 64] per image and channel
        for n, q in blk.named_parameters():
            out[f"{typ}_grad_{n}"] = q.grad.numpy()
        print("block_large", typ, "y sum", float(out[f"{typ}_y_sum"]), "dx abs sum", float(out[f"{typ}_dx_abs_sum"]),
              "params", [n for n, _ in blk.named_parameters()])
    # ... and the genuine ConvTransBlock (CLC_run.py:195-220: conv1_1 | ResidualBlock + identity || Block | conv1_2 + residual) at the same size:
    # [2, 128, 128, 128] NCHW, 32 768 pixels -> the wave-private 1x1 kernel (conv1_1 / conv1_2), the 64-channel 3x3 layers, the fused Block,
    # the gradient folds / slots of the product's ConvTransBlock
    xc0 = torch.randn(2, 128, 128, 128, generator=torch.Generator().manual_seed(20252))
    dyc = torch.randn(2, 128, 128, 128, generator=torch.Generator().manual_seed(20253))
    out["ctb_seed_x"], out["ctb_seed_dy"] = np.int64(20252), np.int64(20253)
    out["ctb_x_sha256"] = np.array(hashlib.sha256(xc0.numpy().tobytes()).hexdigest())
    ctb = mod.ConvTransBlock(64, 64, 8, 8, 0, "SW").train()
    apply_weight_recipe(ctb, 3)
    xc = xc0.clone().requires_grad_(True)
    yc = ctb(xc)
    yc.backward(dyc)
    for k, (b, r, c) in enumerate(patches):
        out[f"ctb_y_patch{k}"] = yc.detach()[b, :, r:r + 8, c:c + 8].numpy()
        out[f"ctb_dx_patch{k}"] = xc.grad[b, :, r:r + 8, c:c + 8].numpy()
    for nm, t in (("y", yc.detach()), ("dx", xc.grad)):
        out[f"ctb_{nm}_abs_sum"] = np.float64(t.double().abs().sum().item())
        out[f"ctb_{nm}_chan_sums"] = t.double().sum(dim=(2, 3)).numpy()               # [2, 128] per image and channel
    for n, q in ctb.named_parameters():
        out[f"ctb_grad_{n}"] = q.grad.numpy()
    print("block_large ctb: y abs sum", float(out["ctb_y_abs_sum"]), "params", len(list(ctb.named_parameters())))
    np.savez_compressed(os.path.join(OUT, "block_large.npz"), **out)
    print("block_large:", len(out), "arrays,", os.path.getsize(os.path.join(OUT, "block_large.npz")) // 1024, "KB")


def dormant_goldens(ref):
    """The modules the reference constructs and never calls (CLC_run.py:284-313, 359-369): the genuine in-file CLM class and a
    multi_ref_fusion stack on recipe weights and seeded inputs -> pins the product's / the oracle's versions of them."""
    mod = sys.modules["models.CLC_run"]
    g = torch.Generator().manual_seed(11)
    clm = mod.CLM(192, head_dim=32, window_size=4).eval()
    apply_weight_recipe(clm, 5)
    x, r = torch.randn(2, 192, 8, 8, generator=g), torch.randn(2, 192, 8, 8, generator=g)
    fus = torch.nn.Sequential(mod.conv1x1(192 * 2, 256), torch.nn.GELU(), mod.conv1x1(256, 192)).eval()
    apply_weight_recipe(fus, 6)
    with torch.no_grad():
        a = clm(x, r)
        out = {"x": x.numpy(), "ref": r.numpy(), "clm_y": a.numpy(), "fusion_y": fus(torch.cat([x, a], dim=1)).numpy()}
    np.savez_compressed(os.path.join(OUT, "dormant.npz"), **out)
    print("dormant:", {k: v.shape for k, v in out.items()})


def rans_goldens():
    gc = leaves.GaussianConditional(None)
    gc.update_scale_table(leaves.get_scale_table())
    cdf, ln, off = gc.quantized_cdf.tolist(), gc.cdf_length.tolist(), gc.offset.tolist()
    arr = gc.quantized_cdf.numpy().astype(np.int32)
    kat = {"gaussian_cdf_sha256": hashlib.sha256(arr.tobytes()).hexdigest(), "gaussian_cdf_shape": list(arr.shape),
           "cdf_row0": cdf[0][:6], "cdf_len_first_last": [ln[0], ln[-1]], "offset_first_last": [off[0], off[-1]],
           "cdf_row20_head": cdf[20][:12], "scale_table_sha256": hashlib.sha256(gc.scale_table.numpy().tobytes()).hexdigest(), "cases": []}
    rng = np.random.default_rng(1234)
    cases = [
        ("empty", [], []),
        ("single_zero", [0], [0]),
        ("in_range_small", [0, 1, -1, 2, -2, 0, 0, 1], [5, 5, 9, 12, 12, 0, 63, 30]),
        ("escapes", [100, -100, 5000, -5000, 70000, -70000, 2 ** 20, -(2 ** 20), 0], [0, 0, 3, 3, 10, 10, 63, 63, 1]),
        ("long_unary_escape", [2 ** 27, -(2 ** 27) - 1], [0, 63]),
    ]
    idx = rng.integers(0, 64, 4000).astype(np.int32)
    sig = gc.scale_table.numpy()[idx]
    sym = np.round(rng.normal(0, sig)).astype(np.int32)
    sym[rng.random(4000) < 0.02] = rng.integers(-3000, 3000, int((rng.random(4000) < 0.02).sum()) or 1)[0]
    cases.append(("random_4000", sym.tolist(), idx.tolist()))
    for name, s, i in cases:
        stream = rans_py.RansEncoder().encode_with_indexes(s, i, cdf, ln, off)
        dec = rans_py.RansDecoder().decode_with_indexes(stream, i, cdf, ln, off)
        assert dec == list(s), name
        kat["cases"].append({"name": name, "symbols": list(map(int, s)), "indexes": list(map(int, i)), "stream_hex": stream.hex()})
    # CDF quantiser known answers
    kat["pmf_cases"] = []
    for n in (2, 3, 7, 40):
        p = rng.random(n).astype(np.float32) ** 3
        p[rng.integers(0, n)] = 0.0  # force a zero-width bin -> frequency stealing
        p /= p.sum()
        kat["pmf_cases"].append({"pmf": [float(v) for v in p], "cdf": rans_py.pmf_to_quantized_cdf(p.tolist(), 16)})
    with open(os.path.join(OUT, "rans_kat.json"), "w") as f:
        json.dump(kat, f)
    print("rans kat:", [c["name"] for c in kat["cases"]])


def clm_goldens():
    """Outputs of the genuine /root/reference/models/CLM.py (orphan module, loaded by file path) -> tests/golden/clm.npz."""
    mod = ref_shim.import_reference_file("models/CLM.py", "ref_clm_file")
    out = {}
    g = torch.Generator().manual_seed(42)
    for name, cls in (("clm", mod.CLM), ("simple", mod.SimpleCLM)):
        m = cls(64, temperature=0.5).eval()
        apply_weight_recipe(m, 11)
        if name == "clm":  # offsets of a few pixels so the bilinear / border logic is exercised
            with torch.no_grad():
                m.alignment.offset_conv.weight.mul_(6.0)
                m.alignment.offset_conv.bias.mul_(20.0)
        y = torch.randn(2, 64, 16, 16, generator=g)
        refs = [torch.randn(2, 64, 16, 16, generator=g) for _ in range(3)]
        with torch.no_grad():
            o = m(y, refs)
        out[f"{name}_y"], out[f"{name}_refs"], out[f"{name}_out"] = y.numpy(), torch.stack(refs).numpy(), o.numpy()
        if name == "clm":
            with torch.no_grad():
                yt, rt = m.feature_transform(y), m.feature_transform(refs[0])
                sim = torch.softmax(torch.bmm(yt.view(2, 64, -1).transpose(1, 2), rt.view(2, 64, -1)) / 0.5, dim=-1)
                out["clm_colsum0"] = sim.sum(1).numpy()
    np.savez_compressed(os.path.join(OUT, "clm.npz"), **out)
    print("clm:", {k: v.shape for k, v in out.items()})


def patch_matching_goldens():
    """Outputs of the genuine numeric functions of models/Patch_Matching.py (AST-extracted: the file cannot be imported),
    `.cuda()` patched to the identity -> tests/golden/patch_matching.npz."""
    names = ["L2_or_pearson_corr", "create_gaussian_masks", "SI_Wraper", "SI_Finder_at_Image_Domain", "rgb_transform",
             "reduce_mean_and_std_normalize_images"]
    ns = ref_shim.extract_reference_functions("models/Patch_Matching.py", names)
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ph = pw = 16
        y_img = synthetic_image(2, 64, 96, 8, smooth=True)
        y_dec = (y_img + 0.02 * torch.randn(y_img.shape, generator=torch.Generator().manual_seed(1))).clamp(0, 1)
        x_dec = (torch.roll(y_img, shifts=(5, -7), dims=(2, 3)) + 0.03 * torch.randn(y_img.shape, generator=torch.Generator().manual_seed(2))).clamp(0, 1)
        mask = ns["create_gaussian_masks"](64, 96, ph, pw)
        out = {"x_dec": x_dec.numpy(), "y_img": y_img.numpy(), "y_dec": y_dec.numpy(), "mask": mask.numpy()}
        xp = x_dec[0:1].reshape(1, 3, 4, 16, 6, 16).permute(0, 2, 4, 1, 3, 5).reshape(-1, 3, 16, 16)
        q = ns["rgb_transform"](ns["reduce_mean_and_std_normalize_images"](xp * 255))
        r = ns["rgb_transform"](ns["reduce_mean_and_std_normalize_images"](y_dec[0:1] * 255))
        corr = ns["L2_or_pearson_corr"](q, r, ph, pw)
        out["q"], out["r"], out["corr"] = q.numpy(), r.numpy(), corr.numpy()
        out["finder"] = ns["SI_Finder_at_Image_Domain"](x_dec, y_img, ph, pw, y_dec, mask=mask).numpy()
        out["wraper_k3"] = ns["SI_Wraper"](corr * mask, ph, pw, 24, y_img[0:1], k=3, temperature=15).numpy()
        out["wraper_k3_stack"] = ns["SI_Wraper"](corr * mask, ph, pw, 24, y_img[0:1], k=3, temperature=15, is_stack=True).numpy()
    finally:
        torch.Tensor.cuda = orig
    np.savez_compressed(os.path.join(OUT, "patch_matching.npz"), **out)
    print("patch matching:", {k: v.shape for k, v in out.items()})


def codec_goldens(ref):
    """Streams of the reference's OWN compress() / decompress() (models/CLC_run.py:629-716, 738-814; models/tcm.py) run through
    the shim on the seeded 256x256 inputs -> tests/golden/codec_<name>.npz: y / z stream bytes, shape, the tensors the coder
    saw (y, z, means, scales — so a coder can be checked on IDENTICAL inputs, independent of float summation order in the
    transforms) and a checksum of the decoded x_hat.  Pins symbol order, slice concatenation and the `strings` layout."""
    for name, R in (("clc_r1", 1), ("clc_r3", 3), ("tcm", 0)):
        torch.manual_seed(0)
        m = (ref.CLC(N=64, num_ref_frames=R) if R else ref.TCM(N=64)).eval()
        apply_weight_recipe(m, 0)
        m.update(force=True)
        x = synthetic_image(1, 256, 256, 100, smooth=True)
        refs = [synthetic_image(1, 256, 256, 101 + i, smooth=True) for i in range(R)]
        with torch.no_grad():
            enc = m.compress(x, refs) if R else m.compress(x)
            dec = m.decompress(enc["strings"], enc["shape"], refs) if R else m.decompress(enc["strings"], enc["shape"])
            fwd = m(x, refs) if R else m(x)
            z = m.h_a(fwd["para"]["y"])
        assert isinstance(enc["strings"][0], list) and len(enc["strings"][0]) == 1 and len(enc["strings"][1]) == 1
        assert torch.equal(dec["x_hat"], fwd["x_hat"].clamp(0, 1)), "reference decoder != encoder-side reconstruction"
        xh = dec["x_hat"]
        out = {
            "y_stream": np.frombuffer(enc["strings"][0][0], dtype=np.uint8), "z_stream": np.frombuffer(enc["strings"][1][0], dtype=np.uint8),
            "shape": np.array(list(enc["shape"]), dtype=np.int64),
            "y": fwd["para"]["y"].numpy(), "means": fwd["para"]["means"].numpy(), "scales": fwd["para"]["scales"].numpy(), "z": z.numpy(),
            "x_hat_sha256": np.array(hashlib.sha256(xh.numpy().tobytes()).hexdigest()),
            "x_hat_patch": xh[0, :, 96:112, 96:112].numpy(), "x_hat_mean": np.float64(xh.double().mean().item()),
            "bpp": np.float64(8.0 * (len(enc["strings"][0][0]) + len(enc["strings"][1][0])) / (256 * 256)),
        }
        np.savez_compressed(os.path.join(OUT, f"codec_{name}.npz"), **out)
        print("codec", name, "y bytes", len(enc["strings"][0][0]), "z bytes", len(enc["strings"][1][0]), "bpp", float(out["bpp"]))


def pins_goldens(ref):
    """Reference-held arithmetic of the hot path's entry / exit points, executed as the reference wrote it -> tests/golden/pins.npz:
      likelihood   CLC._likelihood / _standardized_cumulative (models/CLC_run.py:718-736, the model's own copy of the Gaussian-bin
                   likelihood) on a grid that covers sigma below the 0.11 bound, sigma at the scale-table thresholds and |v| from 0 to 40
      rd loss      class RateDistortionLoss (train_CLC.py:36-59), both distortion types (its `ms_ssim` import resolved to the oracle's
                   restatement of pytorch_msssim — that leaf stays unpinned; the bpp term, the MSE term and the weighting are pinned)
      optimizers   configure_optimizers (train_CLC.py:81-117) on the genuine CLC(N=64, R=1): which parameter names go to the main / aux
                   AdamW, in which order, with which hyper-parameters
      eval         compute_psnr / compute_bpp / pad / crop (eval_CLC.py:133-166) on 200x300 and 512x768 inputs."""
    import types

    from oracle.loss import ms_ssim as oracle_ms_ssim

    out = {}
    # ---- (a) the model's own _likelihood
    m = ref.CLC(N=64, num_ref_frames=1).eval()
    g = torch.Generator().manual_seed(2024)
    table = leaves.get_scale_table()
    nxt = lambda t, d: torch.nextafter(t, torch.full_like(t, d))
    sig = torch.cat([torch.tensor([0.0, 1e-4, 0.05, 0.1099, 0.11, 0.1101, 0.5, 1.0, 7.5, 64.0, 256.0, 300.0]), table, nxt(table, 1e9), nxt(table, -1e9)])
    val = torch.cat([torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 5.0, 8.0, 12.0, 20.0, 40.0]), torch.rand(19, generator=g) * 6])
    S, V = torch.meshgrid(sig, val, indexing="ij")
    means = (torch.rand(S.shape, generator=g) - 0.5) * 4
    sign = torch.where(torch.rand(S.shape, generator=g) < 0.5, -1.0, 1.0)
    inputs = means + sign * V
    with torch.no_grad():
        lik = m._likelihood(inputs, S.contiguous(), means)
        lik_nomean = m._likelihood(inputs, S.contiguous())
    out.update(lik_inputs=inputs.numpy(), lik_scales=S.contiguous().numpy(), lik_means=means.numpy(), lik=lik.numpy(), lik_nomean=lik_nomean.numpy())
    # ---- (b) RateDistortionLoss
    ns = ref_shim.extract_reference_functions("train_CLC.py", ["compute_msssim", "RateDistortionLoss", "configure_optimizers"], {"ms_ssim": oracle_ms_ssim})
    tgt = synthetic_image(2, 176, 192, 31, smooth=True)
    xh = (tgt + 0.03 * torch.randn(tgt.shape, generator=g)).clamp(0, 1)
    ly = torch.rand(2, 320, 11, 12, generator=g).clamp_min(1e-9) ** 2
    lz = torch.rand(2, 192, 3, 3, generator=g).clamp_min(1e-9)
    ly[0, 0, 0, :4] = 1e-9
    out.update(rd_x_hat=xh.numpy(), rd_target=tgt.numpy(), rd_lik_y=ly.numpy(), rd_lik_z=lz.numpy())
    for typ in ("mse", "ms_ssim"):
        for lm in (0.0067, 0.05):
            with torch.no_grad():
                r = ns["RateDistortionLoss"](lmbda=lm, type=typ)({"x_hat": xh, "likelihoods": {"y": ly, "z": lz}}, tgt)
            for k, v in r.items():
                out[f"rd_{typ}_{lm}_{k}"] = np.float64(v.double().item())
                out[f"rd_{typ}_{lm}_{k}_f32"] = np.float32(v.item())
    # ---- (c) configure_optimizers
    args = types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3)
    opt, aux = ns["configure_optimizers"](m, args)
    names = {id(p): n for n, p in m.named_parameters()}
    desc = lambda o: {"class": type(o).__name__, "names": [names[id(p)] for p in o.param_groups[0]["params"]],
                      **{k: (list(v) if isinstance(v, tuple) else v) for k, v in o.param_groups[0].items() if k in ("lr", "betas", "eps", "weight_decay", "amsgrad")}}
    out["optimizers_json"] = np.array(json.dumps({"main": desc(opt), "aux": desc(aux), "n_groups": [len(opt.param_groups), len(aux.param_groups)]}))
    # ---- (d) eval helpers
    ev = ref_shim.extract_reference_functions("eval_CLC.py", ["compute_psnr", "compute_bpp", "pad", "crop"])
    for tag, (h, w) in (("200x300", (200, 300)), ("512x768", (512, 768)), ("256x256", (256, 256)), ("1x129", (1, 129))):
        x = torch.rand(1, 3, h, w, generator=torch.Generator().manual_seed(1000 + h + w))   # (tests re-draw it; pad_x_200x300 below is stored)
        xp, padding = ev["pad"](x, 128)
        back = ev["crop"](xp, padding)
        assert torch.equal(back, x)
        out[f"pad_{tag}_padding"] = np.array(padding, dtype=np.int64)
        out[f"pad_{tag}_shape"] = np.array(xp.shape, dtype=np.int64)
        out[f"pad_{tag}_sha256"] = np.array(hashlib.sha256(xp.numpy().tobytes()).hexdigest())
        out[f"pad_{tag}_in_sha256"] = np.array(hashlib.sha256(x.numpy().tobytes()).hexdigest())
    xs = torch.rand(1, 3, 200, 300, generator=torch.Generator().manual_seed(77))
    out["pad_x_200x300"] = xs.numpy()
    xp, padding = ev["pad"](xs, 128)
    out["pad_x_200x300_padded"] = xp.numpy()
    out["pad_x_200x300_padding"] = np.array(padding, dtype=np.int64)
    a, b = xh, tgt
    out["psnr"] = np.float64(ev["compute_psnr"](a, b))
    out["bpp"] = np.float64(ev["compute_bpp"]({"x_hat": xh, "likelihoods": {"y": ly, "z": lz}}))
    np.savez_compressed(os.path.join(OUT, "pins.npz"), **out)
    print("pins:", len(out), "arrays; lik grid", tuple(lik.shape), "rd", {k: float(v) for k, v in out.items() if k.startswith("rd_mse_0.0067") and not k.endswith("f32")},
          "psnr", float(out["psnr"]), "bpp", float(out["bpp"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_shim.import_reference_models()
    if len(sys.argv) > 1 and sys.argv[1] == "codec":
        codec_goldens(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pm":
        patch_matching_goldens()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "dormant":
        dormant_goldens(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "pins":
        pins_goldens(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "block_large":
        block_large_goldens(ref)
        return
    graph_goldens(ref)
    codec_goldens(ref)
    block_goldens(ref)
    block_large_goldens(ref)
    dormant_goldens(ref)
    rans_goldens()
    clm_goldens()
    patch_matching_goldens()
    pins_goldens(ref)


if __name__ == "__main__":
    main()
