"""A/B of the Swin-block MLP, fused (csrc/fused_mlp.hip) vs the two-launch chain, in ONE process: forward alone, and forward + backward
(data gradients + both layers' filter gradients, deferred reductions flushed), operands rotated through more buffers than the Infinity
Cache holds.     python tools/bench_mlp.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clc_amd import layers, ops

CL = torch.channels_last


def timed(fn, reps, nbuf):
    for i in range(nbuf):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i % nbuf)
    g.replay()
    torch.cuda.synchronize()
    return g


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    fc1, fc2 = layers.Linear(64, 256).to(dev), layers.Linear(256, 64).to(dev)
    shapes = ((8, 128, 128), (8, 64, 64), (4, 256, 256), (8, 32, 32))
    if os.environ.get("SHAPES"):
        shapes = [shapes[int(i)] for i in os.environ["SHAPES"].split(",")]
    legs = os.environ.get("LEGS", "fwd,fwd+bwd").split(",")
    for (N, H, W) in shapes:
        M = N * H * W
        nbuf = max(2, int(600e6 // (M * 64 * 4 * 3)) + 1)
        xs = [torch.randn(N, 64, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        rs = [torch.randn(N, 64, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        gs = [torch.randn(N, 64, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        flops = 2.0 * M * 2 * 64 * 256

        def fwd(mode):
            def f(i):
                with torch.no_grad():
                    if mode == "fused":
                        ops.mlp(xs[i], fc1.weight, fc1.bias, fc2.weight, fc2.bias, res=rs[i])
                    else:
                        fc2(fc1(xs[i], act=ops.ACT_GELU), res=rs[i])
            return f

        def fwdbwd(mode):
            def f(i):
                x = xs[i].detach().requires_grad_(True)
                if mode == "fused":
                    y = ops.mlp(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, res=rs[i])
                else:
                    g = ops.ActGate()
                    y = fc2(fc1(x, act=ops.ACT_GELU, gate_out=g), res=rs[i], gate_in=g)
                y.backward(gs[i])
                for prm in list(fc1.parameters()) + list(fc2.parameters()):
                    prm.grad = None
            return f

        old = ops.FUSED_MLP_MIN_PIX
        ops.FUSED_MLP_MIN_PIX = 1024
        try:
            for label, mk, mult in (("fwd", fwd, 1.0), ("fwd+bwd", fwdbwd, 3.0)):
                if label not in legs:
                    continue
                graphs = []
                for m in ("chain", "fused", "fused-recompute"):
                    if label == "fwd" and m == "fused-recompute":
                        continue
                    ops.MLP_SAVE_H = 0 if m == "fused-recompute" else 1
                    graphs.append((m, timed(mk("fused" if m.startswith("fused") else m), reps, nbuf), []))
                for _ in range(7):
                    for m, g, ts in graphs:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(); g.replay(); e1.record()
                        torch.cuda.synchronize()
                        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
                line = f"mlp 64->256->64 {N}x{H}x{W} {label:8s} {flops * mult / 1e9:6.1f} GF:"
                for m, g, ts in graphs:
                    us = sorted(ts)[len(ts) // 2]
                    line += f"  [{m}] {us:7.1f} us {flops * mult / us / 1e6:5.1f} TF"
                print(line, flush=True)
        finally:
            ops.FUSED_MLP_MIN_PIX = old


if __name__ == "__main__":
    main()
