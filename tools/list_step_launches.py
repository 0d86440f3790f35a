"""Every C-ABI launch of ONE eager training step (BASELINE configs[1]: batch 8, 256 x 256, 1 reference) with its label (shape), kernel name and
HIP-event time, grouped by kernel: which layers sit on which kernel.   python tools/list_step_launches.py [name filter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from clc_amd import lib, ops
from clc_amd import models as pm
from clc_amd.train import TrainEngine

flt = sys.argv[1] if len(sys.argv) > 1 else ""
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = pm.CLC(N=64, num_ref_frames=1).to(dev).train()
eng = TrainEngine(model, lmbda=0.0067, use_graph=False)
x = torch.rand(8, 3, 256, 256, device=dev)
refs = [torch.rand(8, 3, 256, 256, device=dev)]
for _ in range(2):
    eng.step(x, refs)           # (the first call discovers the parameter set and builds the arenas)
torch.cuda.synchronize()
ops.PROFILE = []
torch.cuda._sleep(int(0.15 * 2.4e9))
eng._eager_step(x, refs)
torch.cuda.synchronize()
rec, ops.PROFILE = ops.PROFILE, None
L = lib.load()
rows = {}
for r in rec:
    rows.setdefault(bench._kernel_name(L, r), []).append(r)
for name, rs in sorted(rows.items(), key=lambda kv: -sum(r.ms() for r in kv[1])):
    if flt not in name:
        continue
    print(f"{name}: {len(rs)} launches, {sum(r.ms() for r in rs):.3f} ms")
    for r in rs:
        print(f"    {r.ms() * 1e3:8.1f} us  {r.flops / 1e9:7.2f} GF  {r.owner or '':12s} {r.label}")
