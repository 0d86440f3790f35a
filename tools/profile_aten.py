"""Which aten ops (torch glue around the HIP kernels) run in one eager training step: count and device time per op."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import synthetic_batch
from clc_amd import models
from clc_amd.train import TrainEngine
from oracle.recipe import apply_weight_recipe

dev = torch.device("cuda", 0)
model = models.CLC(N=64, num_ref_frames=1)
apply_weight_recipe(model, 0)
model = model.to(dev).train()
x = synthetic_batch(8, 256, 100, dev)
refs = [synthetic_batch(8, 256, 1000, dev)]
eng = TrainEngine(model, lmbda=0.0067, lr=1e-4, aux_lr=1e-3, clip_max_norm=1.0, use_graph=False)
for _ in range(2):
    eng.step(x, refs)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    eng.step(x, refs)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_device_time_total, e.device_time_total) for e in prof.key_averages()]
rows.sort(key=lambda r: -r[3])
print(f"{'op':50s} {'calls':>6s} {'self dev us':>12s} {'total dev us':>12s}")
for k, c, sd, td in rows:
    if k.startswith("aten::") or k.startswith("_") or "Backward" in k:
        print(f"{k[:50]:50s} {c:6d} {sd:12.0f} {td:12.0f}")

import collections
for name in ("aten::clone", "aten::add", "aten::add_", "aten::mul", "aten::copy_", "aten::cat", "aten::fill_", "aten::zero_", "aten::zeros_like", "aten::sum"):
    by = collections.Counter(); dev_us = collections.Counter()
    for e in prof.events():
        if e.name == name:
            st = [f for f in e.stack if "clc_amd" in f or "autograd" in f][:2]
            key = (" <- ".join(x.split("/")[-1] for x in st) or "(autograd engine)") + "  " + str(e.input_shapes)[:90]
            by[key] += 1; dev_us[key] += e.device_time_total
    print("==", name)
    for k, c in by.most_common(14):
        print(f"   {c:4d}  {dev_us[k]:8.0f} us  {k}")
