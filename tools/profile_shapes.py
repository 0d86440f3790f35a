"""Per-shape time table of the conv launches of one eager training step (HIP events around every launch, ops.PROFILE).
Usage (GPU box): python tools/profile_shapes.py [batch]   -> table sorted by summed time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_batch
from clc_amd import models, ops
from clc_amd.train import TrainEngine
from clc_amd.recipe import apply_weight_recipe


def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    model = models.CLC(N=64, num_ref_frames=1)
    apply_weight_recipe(model, 0)
    model = model.to(dev).train()
    x = synthetic_batch(bs, 256, 100, dev)
    refs = [synthetic_batch(bs, 256, 1000, dev)]
    eng = TrainEngine(model, lmbda=0.0067, lr=1e-4, aux_lr=1e-3, clip_max_norm=1.0, use_graph=False)
    for _ in range(2):
        eng.step(x, refs)
    torch.cuda.synchronize()
    ops.PROFILE = []
    torch.cuda._sleep(int(0.15 * 2.4e9))   # the step is enqueued behind a spin kernel: the host's launch latency stays out of the event brackets
    eng._eager_step(x, refs)
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for r in rec:
        a = agg.setdefault((r.label or r.fam, r.variant, r.owner), [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += r.ms()
        a[2] += r.flops
        a[3] += r.nbytes or 0
    tot = sum(a[1] for a in agg.values())
    print(f"total launch time {tot:.2f} ms over {sum(a[0] for a in agg.values())} launches (eager, event-bracketed: includes ~launch gaps)")
    for (shape, variant, owner), (n, ms, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("ROWS", "120"))]:
        print(f"{ms:7.3f} ms  n={n:3d}  {ms / n * 1e3:7.1f} us  {fl / ms / 1e9:6.1f} TF  {nb / ms / 1e6:6.0f} GB/s  v{variant:<8d} {owner:11s} {shape}")


if __name__ == "__main__":
    main()
