"""Per-image codec latency on the GPU box: CLC(N=64, n_refs=R).compress() / decompress() at HxW, bitstream size, PSNR.
usage: python tools/bench_codec.py [size] [n_refs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import torch
from clc_amd import models
from oracle.recipe import apply_weight_recipe, synthetic_image

size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
m = models.CLC(N=64, num_ref_frames=R)
apply_weight_recipe(m, 0)
m = m.to(dev).eval()
m.update(force=True)
x = synthetic_image(1, size, size, 100, smooth=True).to(dev)
refs = [synthetic_image(1, size, size, 101 + i, smooth=True).to(dev) for i in range(R)]
for _ in range(2):
    out = m.compress(x, refs); rec = m.decompress(out["strings"], out["shape"], refs)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 5
for _ in range(n):
    out = m.compress(x, refs)
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(n):
    rec = m.decompress(out["strings"], out["shape"], refs)
torch.cuda.synchronize(); t2 = time.perf_counter()
nbytes = sum(len(s) for ss in out["strings"] for s in ss)
psnr = -10 * math.log10(torch.mean((rec["x_hat"] - x) ** 2).item())
print(f"{size}x{size} n_refs={R}: compress {1e3*(t1-t0)/n:.1f} ms, decompress {1e3*(t2-t1)/n:.1f} ms per image; "
      f"{nbytes} bytes = {8*nbytes/size/size:.3f} bpp; PSNR {psnr:.2f} dB (random-init weights)")
