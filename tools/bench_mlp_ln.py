"""Kernel-level look at the LayerNorm form of the fused Swin-block MLP (csrc/fused_mlp.hip): forward + backward of `x + mlp(ln2(x))` at
8 x 128 x 128, LN inside the launches vs separate LayerNorm launches, and the backward kernel's timing ablations (CLC_TUNE_ABLATE; wrong results).
Times are hipGraph replays (launches back to back, as in the step): an eager kernel trace of the same launches overstates every phase that the
graph's neighbours overlap (csrc/fused_mlp.hip).   python tools/bench_mlp_ln.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clc_amd import layers, lib, ops

CL = torch.channels_last


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    L = lib.load()
    fc1, fc2, ln = layers.Linear(64, 256).to(dev), layers.Linear(256, 64).to(dev), layers.LayerNorm(64).to(dev)
    prms = list(fc1.parameters()) + list(fc2.parameters()) + list(ln.parameters())
    N, H, W = 8, 128, 128
    nbuf = 6
    xs = [torch.randn(N, 64, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
    gs = [torch.randn(N, 64, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]

    def step(i, fused):
        x = xs[i].detach().requires_grad_(True)
        if fused:
            y = ops.mlp_ln(x, ln.weight, ln.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        else:
            f2 = ops.GradFold()
            y = ops.mlp(ln(x, fold_in=f2), fc1.weight, fc1.bias, fc2.weight, fc2.bias, res=x, fold_out=f2)
        y.backward(gs[i])
        for prm in prms:
            prm.grad = None

    legs = [("separate", False, 0, None), ("ln-fused", True, 0, None), ("abl1", True, 1, None), ("abl4", True, 4, None), ("abl5", True, 5, None)]
    for label, fused, abl, stag in legs:
        prev = L.clc_set_tuning(12, abl)
        for i in range(nbuf):
            step(i, fused)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for r in range(reps):
                step(r % nbuf, fused)
        L.clc_set_tuning(12, prev)
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            g.replay()
            t1.record()
            torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1) / reps * 1e3)
        print(f"{label} ln_fused={fused}: {sorted(ts)[2]:.1f} us per fwd+bwd (graph replay, median of 5; min {min(ts):.1f})", flush=True)


if __name__ == "__main__":
    main()
