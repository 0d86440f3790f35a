"""Oracle rate-distortion loss and eval metrics (TEST INFRASTRUCTURE).

Follows /root/reference/train_CLC.py:36-59 (RateDistortionLoss),
/root/reference/eval_CLC.py:133-166 (compute_psnr / compute_bpp / pad / crop) and the
published pytorch_msssim.ms_ssim algorithm (dependency absent → unpinned).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _gauss_win(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _gfilt(x, win):
    C = x.shape[1]
    w = win.to(x)
    x = F.conv2d(x, w.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, w.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)


def ms_ssim(X, Y, data_range=1.0, weights=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333), K=(0.01, 0.03)):
    win = _gauss_win()
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mcs = []
    for i in range(len(weights)):
        mu1, mu2 = _gfilt(X, win), _gfilt(Y, win)
        s11 = _gfilt(X * X, win) - mu1 * mu1
        s22 = _gfilt(Y * Y, win) - mu2 * mu2
        s12 = _gfilt(X * Y, win) - mu1 * mu2
        cs_map = (2 * s12 + C2) / (s11 + s22 + C2)
        ssim_map = ((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * cs_map
        ssim_pc = ssim_map.flatten(2).mean(-1)
        cs = cs_map.flatten(2).mean(-1)
        if i < len(weights) - 1:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim_pc)], dim=0)  # [levels, B, C]
    w = torch.tensor(weights).to(X).view(-1, 1, 1)
    return torch.prod(vals ** w, dim=0).mean()


class RateDistortionLoss(torch.nn.Module):
    def __init__(self, lmbda=1e-2, type="mse"):
        super().__init__()
        self.lmbda, self.type = lmbda, type

    def forward(self, output, target):
        N, _, H, W = target.size()
        num_pixels = N * H * W
        out = {"bpp_loss": sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in output["likelihoods"].values())}
        if self.type == "mse":
            out["mse_loss"] = F.mse_loss(output["x_hat"], target)
            out["loss"] = self.lmbda * 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        else:
            out["ms_ssim_loss"] = ms_ssim(output["x_hat"], target, data_range=1.0)
            out["loss"] = self.lmbda * (1 - out["ms_ssim_loss"]) + out["bpp_loss"]
        return out


def compute_psnr(a, b):
    return -10 * math.log10(torch.mean((a - b) ** 2).item())


def compute_bpp(out_net):
    size = out_net["x_hat"].size()
    num_pixels = size[0] * size[2] * size[3]
    return sum(torch.log(l).sum() / (-math.log(2) * num_pixels) for l in out_net["likelihoods"].values()).item()


def pad(x, p):
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    padding = (left, W - w - left, top, H - h - top)
    return F.pad(x, padding, mode="constant", value=0), padding


def crop(x, padding):
    return F.pad(x, tuple(-p for p in padding))
