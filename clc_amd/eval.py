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


# ----------------------------------------------------------------------------------------------- checkpoints / RD sweep (§8(f)-3)


def load_checkpoint(net, checkpoint, map_location="cpu"):
    """Load a reference-format checkpoint into `net` (/root/reference/eval_CLC.py:284-291, train_CLC.py:459-465): a path or a dict,
    optionally wrapped as {"state_dict": ...}, keys optionally prefixed "module." (nn.DataParallel); extra keys are ignored and
    missing ones keep their values (CLC.load_state_dict, CLC_run.py:599-618); the CDF tables are rebuilt if absent (net.update())."""
    ck = torch.load(checkpoint, map_location=map_location, weights_only=False) if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__") else checkpoint
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    sd = {k.replace("module.", ""): v for k, v in sd.items()}
    net.load_state_dict(sd)
    net.update()
    return {k: ck[k] for k in ("epoch", "loss") if isinstance(ck, dict) and k in ck}


def find_checkpoints(models_dir):
    """The reference's layout (eval_CLC.py:183-204): <models_dir>/<tag>_<lambda>/<lambda>checkpoint_best.pth.tar, sorted by lambda."""
    import glob
    import os

    out = []
    for d in sorted(glob.glob(os.path.join(models_dir, "*_*"))):
        lam = os.path.basename(d).rsplit("_", 1)[-1]
        path = os.path.join(d, f"{lam}checkpoint_best.pth.tar")
        if os.path.exists(path):
            out.append({"path": path, "bitrate": float(lam) if lam.replace(".", "", 1).isdigit() else 0.0})
    out.sort(key=lambda c: c["bitrate"])
    return out


def rd_sweep(make_net, checkpoints, samples, results_dir, device="cuda", use_engine=True, plot=True):
    """eval_CLC.py:main (259-440): evaluate every checkpoint on `samples` (list of (image, [refs])), write rd_results.csv with the
    reference's columns and (when matplotlib is importable) rd_curve.png.  make_net() builds the model for one checkpoint."""
    import csv
    import os

    os.makedirs(results_dir, exist_ok=True)
    results = []
    csv_path = os.path.join(results_dir, "rd_results.csv")
    samples = list(samples)
    with open(csv_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Checkpoint", "Bitrate (bpp)", "PSNR (dB)", "Time (s)"])
        for cp in checkpoints:
            net = make_net().to(device).eval()
            load_checkpoint(net, cp["path"], map_location="cpu")
            engine = None
            if use_engine:
                from .codec import CodecEngine

                engine = CodecEngine(net)
            try:
                r = evaluate(net, samples, device=device, engine=engine)
            finally:
                if engine is not None:
                    engine.close()   # (one engine per checkpoint: its threads, graphs and pinned buffers go with it)
            results.append({"checkpoint": cp["path"], "bitrate": r["avg_bpp"], "psnr": r["avg_psnr"], "time": r["avg_time_s"]})
            w.writerow([cp["path"], f"{r['avg_bpp']:.4f}", f"{r['avg_psnr']:.2f}", f"{r['avg_time_s']:.4f}"])
            f.flush()
    if plot:
        try:
            import matplotlib

            matplotlib.use("Agg")
            import matplotlib.pyplot as plt

            rs = sorted(results, key=lambda r: r["bitrate"])
            plt.figure(figsize=(10, 6))
            plt.plot([r["bitrate"] for r in rs], [r["psnr"] for r in rs], "o-", linewidth=2, markersize=8)
            plt.xlabel("Bitrate (bpp)")
            plt.ylabel("PSNR (dB)")
            plt.title("Rate-Distortion Performance")
            plt.grid(True, linestyle="--", alpha=0.7)
            plt.tight_layout()
            plt.savefig(os.path.join(results_dir, "rd_curve.png"), dpi=150)
            plt.close()
        except ImportError:
            pass
    return results, csv_path
