#!/usr/bin/env python3
"""Host-side (Python) profile of the reference-surface calls: where the interpreter's time goes in
  loop     the literal training-loop body of /root/reference/train_CLC.py:137-183 on clc_amd.models.CLC (bench.py's reference_loop leg)
  codec    model.compress() / decompress() of one 512x768 image (eval_CLC.py:314-338)
cProfile, sorted by own time.  python tools/profile_host.py loop|codec [N lines]"""
import cProfile
import os
import pstats
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_batch
from clc_amd import models
from clc_amd.recipe import apply_weight_recipe, synthetic_image
from clc_amd.train import RateDistortionLoss, configure_optimizers

what = sys.argv[1] if len(sys.argv) > 1 else "loop"
lines = int(sys.argv[2]) if len(sys.argv) > 2 else 45
dev = torch.device("cuda", 0)
model = models.CLC(N=64, num_ref_frames=1)
apply_weight_recipe(model, 0)
model = model.to(dev)
if what == "loop":
    model.train()
    x = synthetic_batch(8, 256, 100, dev)
    refs = [synthetic_batch(8, 256, 1000, dev)]
    crit = RateDistortionLoss(0.0067)
    opt, aux = configure_optimizers(model, types.SimpleNamespace(learning_rate=1e-4, aux_learning_rate=1e-3))

    def body():
        opt.zero_grad()
        aux.zero_grad()
        out = crit(model(x, refs), x)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        for p in model.parameters():
            if p.grad is not None:
                p.grad.nan_to_num_()
        opt.step()
        a = model.aux_loss()
        a.backward()
        aux.step()
    n = 3
else:
    model.eval()
    model.update(force=True)
    x = synthetic_image(1, 512, 768, 700, smooth=True).to(dev)
    refs = [synthetic_image(1, 512, 768, 701, smooth=True).to(dev)]

    def body():
        enc = model.compress(x, refs)
        model.decompress(enc["strings"], enc["shape"], refs)
    n = 5
for _ in range(3):
    body()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    body()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(lines)
st.sort_stats("cumulative").print_stats(35)
