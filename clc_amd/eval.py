"""Codec evaluation loop with the reference's definitions (/root/reference/eval_CLC.py:133-166, 314-338).

pad to a multiple of 128 (centred, zeros) -> compress -> decompress -> crop -> bitrate = 8*sum(len(stream))/pixels,
PSNR = -10 log10(mean((x - x_hat)^2)).  The squared-error sum is the fixed-order two-stage HIP reduction.
"""
from __future__ import annotations

import math
import time
from typing import Iterable, List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops


def pad(x, p: int = 128):
    h, w = x.size(2), x.size(3)
    H, W = (h + p - 1) // p * p, (w + p - 1) // p * p
    left, top = (W - w) // 2, (H - h) // 2
    padding = (left, W - w - left, top, H - h - top)
    return F.pad(x, padding, mode="constant", value=0), padding


def crop(x, padding):
    return F.pad(x, tuple(-q for q in padding))


def compute_psnr(a, b) -> float:
    a = a.float().contiguous(memory_format=ops.CL)
    b = b.float().contiguous(memory_format=ops.CL)
    mse = (ops.sqdiff_sum(a, b) / a.numel()).item()
    return -10 * math.log10(mse)


def compute_bpp(out_net) -> float:
    size = out_net["x_hat"].size()
    num_pixels = size[0] * size[2] * size[3]
    return (sum(ops.sum_log2(l) for l in out_net["likelihoods"].values()) / (-num_pixels)).item()


@torch.no_grad()
def evaluate(net, samples: Iterable[Tuple[torch.Tensor, Sequence[torch.Tensor]]], p: int = 128, device="cuda", engine=None):
    """samples: iterable of (image [3,h,w] in [0,1], [reference images]). Returns per-image rows and averages.
    engine: a clc_amd.codec.CodecEngine over `net` (graph-captured codec, same bitstreams); default: net.compress / net.decompress."""
    net.eval()
    net.update()
    rows, t_total = [], 0.0
    for x, refs in samples:
        x = x.unsqueeze(0).to(device)
        refs = [r.unsqueeze(0).to(device) for r in refs]
        refs = [F.interpolate(r, size=x.shape[-2:], mode="bilinear", align_corners=False) if r.shape[-2:] != x.shape[-2:] else r for r in refs]
        x_p, padding = pad(x, p)
        refs_p = [pad(r, p)[0] for r in refs]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if engine is not None:
            enc = engine.compress(x_p, refs_p)[0]
            dec = {"x_hat": engine.decompress([enc], refs_p)}
        else:
            enc = net.compress(x_p, refs_p)
            dec = net.decompress(enc["strings"], enc["shape"], refs_p)
        torch.cuda.synchronize()
        t_total += time.perf_counter() - t0
        x_hat = crop(dec["x_hat"], padding)
        num_pixels = x.size(0) * x.size(2) * x.size(3)
        bitrate = sum(len(s[0]) for s in enc["strings"]) * 8.0 / num_pixels
        rows.append({"bpp": bitrate, "psnr": compute_psnr(x, x_hat)})
    n = max(1, len(rows))
    return {"rows": rows, "avg_bpp": sum(r["bpp"] for r in rows) / n, "avg_psnr": sum(r["psnr"] for r in rows) / n, "avg_time_s": t_total / n}
