"""Large-map 1x1 convolution micro-benchmark (conv_igemm_dma2_kernel<..., 1>) under CLC_TUNING-style settings, operands rotated
through enough buffers (> 256 MB Infinity Cache) that every launch reads HBM.
   python tools/bench_1x1.py [reps] [key:value,... ...]      e.g.  python tools/bench_1x1.py 24 8:0 8:1"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clc_amd import lib as _lib
from clc_amd import ops

SHAPES = [(8, 128, 128, 128, 128), (8, 64, 128, 128, 256), (8, 256, 128, 128, 64), (8, 64, 128, 128, 192), (8, 64, 128, 128, 64),
          (8, 128, 64, 64, 128), (8, 64, 64, 64, 256)]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    configs = [a for a in sys.argv[2:]] or ["8:0", "8:1"]
    dev = torch.device("cuda", 0)
    L = _lib.load()
    CL = torch.channels_last
    only = os.environ.get("BENCH_1X1_SHAPES")
    shapes = [SHAPES[int(i)] for i in only.split(",")] if only else SHAPES
    for N, Cin, H, W, Cout in shapes:
        per = N * H * W * (Cin + Cout) * 4
        nbuf = max(2, int(600e6 // per) + 1)
        xs = [torch.randn(N, Cin, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        ys = [torch.empty(N, Cout, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        w = torch.randn(Cout, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        for with_res in (False, True):
          if with_res and Cout > 128:
              continue
          line = f"{Cin:3d}->{Cout:3d} {N}x{H}x{W} {'+res' if with_res else '    '} ({per / 1e6:.0f} MB/launch, {nbuf} buffers):"
          rs = [torch.randn(N, Cout, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)] if with_res else [None] * nbuf
          for cfg in configs:
            for kv in cfg.split(","):
                k, v = kv.split(":")
                L.clc_set_tuning(int(k), int(v))
            for i in range(nbuf):
                ops.conv_raw(xs[i], w, b, ks=1, act=ops.ACT_LRELU, out=ys[i], res=rs[i])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(int(0.02 * 2.4e9))
            e0.record()
            for i in range(reps):
                ops.conv_raw(xs[i % nbuf], w, b, ks=1, act=ops.ACT_LRELU, out=ys[i % nbuf], res=rs[i % nbuf])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            line += f"  [{cfg}] {us:6.1f} us"
          print(line, flush=True)
    L.clc_set_tuning(8, 1)


if __name__ == "__main__":
    main()
