import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
def log(*a):
    print(f"[{time.time()-T0:7.2f}s]", *a, flush=True)
T0 = time.time()
from clc_amd import models as pm
from clc_amd.train import TrainEngine
from oracle.recipe import apply_weight_recipe
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
use_graph = (sys.argv[2] == "graph") if len(sys.argv) > 2 else False
m = pm.CLC(N=64, num_ref_frames=1)
apply_weight_recipe(m, 0)
m = m.to(dev).train()
log("model ready")
g = torch.Generator().manual_seed(0)
x = (torch.randint(0, 256, (B, 3, 256, 256), generator=g).float() / 255).to(dev)
r = [(torch.randint(0, 256, (B, 3, 256, 256), generator=g).float() / 255).to(dev)]
eng = TrainEngine(m, lmbda=0.0067, use_graph=use_graph)
for i in range(6):
    t = time.time()
    out = eng.step(x, r)
    torch.cuda.synchronize()
    log("step", i, "loss", out["loss"].item(), "dt", round(time.time() - t, 4))
t = time.time()
for i in range(10):
    out = eng.step(x, r)
torch.cuda.synchronize()
log("10 steps", round((time.time() - t) / 10 * 1e3, 2), "ms/step", B * 10 / (time.time() - t), "img/s")
