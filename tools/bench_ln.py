"""LayerNorm forward / backward micro-benchmark (graph-replayed, operands rotated): python tools/bench_ln.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clc_amd import ops
dev = torch.device("cuda:0"); CL = torch.channels_last
for (N, C, H, W) in ((8, 64, 128, 128), (8, 64, 64, 64), (8, 64, 32, 32), (16, 128, 16, 16)):
    nb = max(2, int(600e6 // (N * C * H * W * 4 * 3)) + 1)
    xs = [torch.randn(N, C, H, W, device=dev).contiguous(memory_format=CL).requires_grad_(True) for _ in range(nb)]
    gs = [torch.randn(N, C, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nb)]
    gam, bet = torch.randn(C, device=dev).requires_grad_(True), torch.randn(C, device=dev).requires_grad_(True)
    ys = [ops.layernorm(x, gam, bet) for x in xs]
    def fwd(i):
        with torch.no_grad():
            ops.layernorm(xs[i], gam, bet)
    def bwd(i):
        torch.autograd.grad(ys[i], [xs[i], gam, bet], gs[i], retain_graph=True)
    line = f"{N}x{H}x{W} C{C}:"
    for label, fn, nbytes in (("fwd", fwd, 2), ("bwd", bwd, 3)):
        for i in range(nb):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(int(0.01 * 2.4e9))
            e0.record()
            for i in range(20):
                fn(i % nb)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        us = sorted(ts)[2]
        line += f"  {label} {us:6.1f} us ({nbytes * N * C * H * W * 4 / us / 1e6:5.2f} TB/s)"
    print(line, flush=True)
