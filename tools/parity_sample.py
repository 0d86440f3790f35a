#!/usr/bin/env python3
"""Kernel-order decisions on a LARGER parity sample: |d bpp| / |d PSNR| of the HIP forward vs the CPU oracle on N seeded
256x256 images (CLC N=64, 1 reference, eval mode, recipe weights) under several CLC_TUNING settings, one process, one oracle pass.
  python tools/parity_sample.py 16 "16:3" "16:7"
Prints per-variant mean / max and the per-image table (JSON on the last line)."""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clc_amd import lib as _lib
from clc_amd import models as pm
from clc_amd.recipe import apply_weight_recipe, synthetic_image
from oracle import graph as og
from oracle.loss import compute_bpp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
variants = sys.argv[2:] or ["16:3", "16:7"]
dev = torch.device("cuda", 0)
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
o = og.CLC(N=64, num_ref_frames=1).eval()
apply_weight_recipe(o, 0)
p = pm.CLC(N=64, num_ref_frames=1)
p.load_state_dict(o.state_dict())
p = p.to(dev).eval()
L = _lib.load()
seeds = [100 + 10 * i for i in range(n)]
psnr = lambda t, ref: -10 * math.log10(torch.mean((t.double().cpu() - ref.double()) ** 2).item())
rows = {v: [] for v in variants}
for sd in seeds:
    x, r = synthetic_image(1, 256, 256, sd, smooth=True), [synthetic_image(1, 256, 256, sd + 1, smooth=True)]
    with torch.no_grad():
        a = o(x, r)
    bo, po = compute_bpp(a), psnr(a["x_hat"], x)
    for v in variants:
        old = []
        for kv in v.split(","):
            k, val = kv.split(":")
            old.append((int(k), L.clc_set_tuning(int(k), int(val))))
        try:
            with torch.no_grad():
                b = p(x.to(dev), [r[0].to(dev)])
            bb = compute_bpp({"x_hat": b["x_hat"].cpu(), "likelihoods": {k: t.cpu() for k, t in b["likelihoods"].items()}})
            rows[v].append({"seed": sd, "dbpp": abs(bo - bb), "dpsnr_db": abs(po - psnr(b["x_hat"], x))})
        finally:
            for k, val in old:
                L.clc_set_tuning(k, val)
    print(f"seed {sd}: " + "  ".join(f"[{v}] dbpp {rows[v][-1]['dbpp']:.2e}" for v in variants), flush=True)
summary = {v: {"mean_dbpp": sum(q["dbpp"] for q in rs) / n, "max_dbpp": max(q["dbpp"] for q in rs), "mean_dpsnr_db": sum(q["dpsnr_db"] for q in rs) / n,
               "max_dpsnr_db": max(q["dpsnr_db"] for q in rs), "images_over_1e-4": sum(q["dbpp"] > 1e-4 for q in rs)} for v, rs in rows.items()}
for v, s in summary.items():
    print(f"[{v}] mean dbpp {s['mean_dbpp']:.3e} max {s['max_dbpp']:.3e} images over 1e-4: {s['images_over_1e-4']}/{n}; mean dPSNR {s['mean_dpsnr_db']:.2e} dB")
print(json.dumps({"n": n, "summary": summary, "per_image": rows}))
