"""cProfile of the AUTOGRAD ENGINE'S device thread during loss.backward() of the reference-style eager loop (the main-thread profile of
tools/profile_host.py sees it only as run_backward): the profiler is switched on from inside the first Function.backward that runs on that
thread and dumped after the pass.  python tools/profile_backward_thread.py [lines]"""
import cProfile
import os
import pstats
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import synthetic_batch
from clc_amd import models, ops
from clc_amd.recipe import apply_weight_recipe
from clc_amd.train import RateDistortionLoss

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 45
dev = torch.device("cuda", 0)
model = models.CLC(N=64, num_ref_frames=1)
apply_weight_recipe(model, 0)
model = model.to(dev).train()
x = synthetic_batch(8, 256, 100, dev)
refs = [synthetic_batch(8, 256, 1000, dev)]
crit = RateDistortionLoss(0.0067)
state = {"prof": None, "tid": None}


class _Tap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):          # the first node of the backward pass: runs on the engine's device thread
        if state["on"] and state["prof"] is None:
            state["prof"] = cProfile.Profile()
            state["tid"] = threading.get_ident()
            state["prof"].enable()
        return g


def step(on):
    state["on"] = on
    for p in model.parameters():
        p.grad = None
    out = crit(model(x, refs), x)
    _Tap.apply(out["loss"]).backward()
    torch.cuda.synchronize()


for _ in range(3):
    step(False)
step(True)
# the profiler must be disabled on the thread that enabled it: one more tiny backward whose node does that
class _Stop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        if state["prof"] is not None and threading.get_ident() == state["tid"]:
            state["prof"].disable()
        return g
t = torch.ones((), device=dev, requires_grad=True)
_Stop.apply(t * 2).backward()
st = pstats.Stats(state["prof"])
st.sort_stats("tottime").print_stats(lines)
st.sort_stats("cumulative").print_stats(30)
