"""HIP kernels against REFERENCE-HELD arithmetic (tests/golden/pins.npz from `tools/make_golden.py pins`: the reference's own code executed in
the build container) — the Gaussian-bin likelihood of CLC_run.py:718-736, the RD criterion of train_CLC.py:36-59 and the metrics of
eval_CLC.py:133-166.  All calls go through the C ABI (clc_gauss_lik_fwd, clc_log2_sum_partials, clc_sqdiff_partials, clc_ssim_scale_fwd)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CL = torch.channels_last


@pytest.fixture(scope="module")
def pins():
    return np.load(os.path.join(GOLD, "pins.npz"))


def test_hip_gaussian_likelihood_vs_reference_model_method(dev, pins):
    """ops.gaussian_likelihood (training mode, zero noise => evaluated AT the given inputs) vs CLC._likelihood, floor 1e-9 applied as
    GaussianConditional.forward does.  The bin mass is a DIFFERENCE of two erfc values, upper - lower: its error is bounded by the
    operands' rounding, not by its own size (sigma = 256: both operands are ~0.5 and one float32 ulp of them is 4e-5 of a 1.5e-3 bin mass
    — on either side: the reference's float32 formula has the same conditioning).  Bar: |got - want| <= 4 eps * upper + 2e-6 * want
    (eps = 2^-24: two ulps of the larger operand, then 2e-6 relative) for every grid point above the floor; the floor region exactly."""
    from clc_amd import ops

    to4 = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).reshape(1, 32, 12, 17).contiguous(memory_format=CL)   # 204 = 12 * 17 grid rows per column
    x, s, m = (to4(pins[k]).to(dev) for k in ("lik_inputs", "lik_scales", "lik_means"))
    FLOOR = float(np.float32(1e-9))        # the bound is a float32 buffer (likelihood_lower_bound.bound): 9.99999972e-10
    want = to4(pins["lik"]).clamp_min(FLOOR)
    got, y_hat = ops.gaussian_likelihood(x, s, m, torch.zeros_like(x), True)
    got = got.cpu()
    assert torch.isfinite(got).all() and float(got.min()) >= FLOOR
    v = (to4(pins["lik_inputs"]) - to4(pins["lik_means"])).abs().double()
    sig = to4(pins["lik_scales"]).clamp_min(0.11).double()
    upper = 0.5 * torch.erfc(-(0.5 - v) / (sig * math.sqrt(2.0)))          # the larger of the two operands, in double, for the bound only
    above = want > FLOOR
    tol = 4 * 2.0 ** -24 * upper + 2e-6 * want.double()
    excess = ((got.double() - want.double()).abs() - tol)[above]
    assert int(above.sum()) > 3000 and float(excess.max()) <= 0.0, f"max excess over the operand-rounding bound {float(excess.max()):.3e}"
    # where there is no cancellation (upper within 4x of the bin mass) that bound IS a relative one: <= 1e-5
    plain = above & (upper <= 4 * want.double())
    assert int(plain.sum()) > 500 and float(((got - want).abs() / want)[plain].max()) <= 1e-5
    floor = want == FLOOR
    assert int(floor.sum()) > 100 and float((got[floor] - FLOOR).abs().max()) <= 1e-15


def _rd_inputs(pins, dev):
    out = {"x_hat": torch.from_numpy(pins["rd_x_hat"]).to(dev).contiguous(memory_format=CL),
           "likelihoods": {"y": torch.from_numpy(pins["rd_lik_y"]).to(dev).contiguous(memory_format=CL), "z": torch.from_numpy(pins["rd_lik_z"]).to(dev).contiguous(memory_format=CL)}}
    return out, torch.from_numpy(pins["rd_target"]).to(dev)


@pytest.mark.parametrize("typ", ["mse", "ms_ssim"])
@pytest.mark.parametrize("lmbda", [0.0067, 0.05])
def test_hip_rd_loss_vs_reference_class(dev, pins, typ, lmbda):
    """clc_amd.train.RateDistortionLoss vs the reference class on the stored output dict: bpp / MSE 2e-6 relative (fixed-order two-stage f32
    reductions vs torch's), MS-SSIM 5e-6 absolute, the weighted total accordingly."""
    from clc_amd.train import RateDistortionLoss

    out, tgt = _rd_inputs(pins, dev)
    with torch.no_grad():
        r = RateDistortionLoss(lmbda, type=typ)(out, tgt)
    w = lambda k: float(pins[f"rd_{typ}_{lmbda}_{k}"])
    assert abs(r["bpp_loss"].item() - w("bpp_loss")) <= 2e-6 * w("bpp_loss")
    if typ == "mse":
        assert sorted(r) == ["bpp_loss", "loss", "mse_loss"]
        assert abs(r["mse_loss"].item() - w("mse_loss")) <= 2e-6 * w("mse_loss")
    else:
        assert sorted(r) == ["bpp_loss", "loss", "ms_ssim_loss"]
        assert abs(r["ms_ssim_loss"].item() - w("ms_ssim_loss")) <= 5e-6
    assert abs(r["loss"].item() - w("loss")) <= 3e-6 * abs(w("loss"))


def test_hip_eval_metrics_vs_reference_functions(dev, pins):
    from clc_amd import eval as pe

    out, tgt = _rd_inputs(pins, dev)
    assert abs(pe.compute_psnr(out["x_hat"], tgt) - float(pins["psnr"])) <= 1e-4          # dB
    assert abs(pe.compute_bpp(out) - float(pins["bpp"])) <= 2e-6 * float(pins["bpp"])
    x = torch.from_numpy(pins["pad_x_200x300"]).to(dev)
    xp, padding = pe.pad(x, 128)
    assert tuple(padding) == tuple(pins["pad_x_200x300_padding"]) and torch.equal(xp.cpu(), torch.from_numpy(pins["pad_x_200x300_padded"]))
    assert torch.equal(pe.crop(xp, padding), x)
