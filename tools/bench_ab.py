"""A/B of the large-map convolution kernels under clc_set_tuning settings (default: key 13, the persistent 1x1 kernel, off / on), in ONE process, operands rotated through more
buffers than the 256 MB Infinity Cache holds (as inside the training step); every variant's output is compared bit for bit with
the first setting.     python tools/bench_ab.py [reps] [key:value[,key:value] ...]     e.g. 20 10:0 10:1 10:4"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clc_amd import lib as _lib
from clc_amd import ops

CL = torch.channels_last
SHAPES = [  # N, Cin, H, W, Cout, ks, epilogue tensors (res, pre)
    (8, 128, 128, 128, 128, 3, 0, 0), (8, 64, 128, 128, 64, 3, 0, 0), (8, 128, 64, 64, 512, 3, 0, 0), (8, 128, 64, 64, 128, 3, 0, 0),
    (8, 128, 128, 128, 128, 1, 1, 0), (8, 64, 128, 128, 256, 1, 0, 1), (8, 256, 128, 128, 64, 1, 1, 0), (8, 64, 128, 128, 192, 1, 0, 0),
    (8, 64, 128, 128, 64, 1, 1, 0), (8, 128, 64, 64, 128, 1, 1, 0),
    (8, 128, 32, 32, 512, 3, 0, 0), (8, 320, 16, 16, 512, 3, 0, 0), (8, 128, 32, 32, 128, 3, 1, 0), (8, 64, 32, 32, 64, 3, 0, 0), (8, 320, 32, 32, 320, 3, 0, 0),
    (16, 640, 16, 16, 224, 3, 0, 0), (16, 512, 16, 16, 224, 3, 0, 0), (16, 384, 16, 16, 224, 3, 0, 0), (8, 704, 16, 16, 224, 3, 0, 0), (8, 448, 16, 16, 224, 3, 0, 0),   # 15..19: slice-parameter nets
    (16, 128, 16, 16, 512, 1, 0, 1), (16, 512, 16, 16, 128, 1, 1, 0), (16, 128, 16, 16, 384, 1, 0, 0), (16, 224, 16, 16, 128, 3, 0, 0), (16, 128, 16, 16, 128, 1, 1, 0), (16, 128, 16, 16, 64, 3, 0, 0),   # 20..25: slice-loop Swin / cc layers
    (8, 128, 128, 128, 12, 3, 0, 0),   # 26: the 12-channel tail of g_s
    (8, 64, 64, 64, 64, 3, 0, 0), (8, 64, 64, 64, 64, 3, 1, 0),   # 27, 28: 64-channel 3x3 layers on the 64x64 maps
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    settings = sys.argv[2:] or ["13:0", "13:1"]   # "key:value[,key:value]" per variant (clc_set_tuning)
    dev = torch.device("cuda", 0)
    L = _lib.load()
    only = os.environ.get("SHAPES")
    shapes = [SHAPES[int(i)] for i in only.split(",")] if only else SHAPES
    for N, Cin, H, W, Cout, ks, with_res, with_pre in shapes:
        per = N * H * W * (Cin + Cout * (1 + with_res + with_pre)) * 4
        nbuf = max(2, int(700e6 // per) + 1)
        xs = [torch.randn(N, Cin, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        ys = [torch.empty(N, Cout, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        rs = [torch.randn(N, Cout, H, W, device=dev).contiguous(memory_format=CL) if with_res else None for _ in range(nbuf)]
        ps = [torch.empty(N, Cout, H, W, device=dev).contiguous(memory_format=CL) if with_pre else None for _ in range(nbuf)]
        w = (torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05).contiguous(memory_format=CL)
        b = torch.randn(Cout, device=dev)
        dys = [torch.randn(N, Cout, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        dxs = [torch.empty(N, Cin, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nbuf)]
        wt = ops.filter_transpose(w, Cout, ks * ks, Cin).view(Cin, -1)
        flops = 2.0 * N * H * W * ks * ks * Cin * Cout

        def fwd(i):
            ops.conv_raw(xs[i], w, b, ks=ks, act=(ops.ACT_GELU if with_pre else ops.ACT_LRELU), out=ys[i], res=rs[i], y_pre=ps[i], pre_deriv=bool(with_pre),
                         batch_variant_ok=bool(int(os.environ.get("BATCH_VARIANT", "1"))))

        def dgrad(i):
            ops.conv_raw(dys[i], wt, None, ks=ks, pad=ks // 2, transposed=True, out_hw=(H, W), out=dxs[i], res=(xs[i] if with_res else None))

        for label, fn, outs in (("fwd  ", fwd, ys), ("dgrad", dgrad, dxs)):
            if os.environ.get("ONLY") and os.environ["ONLY"] != label.strip():
                continue
            nbytes = per if label.startswith("fwd") else N * H * W * (Cin * (1 + with_res) + Cout) * 4
            line = f"{Cin:3d}->{Cout:3d} k{ks} {N}x{H}x{W} r{with_res}p{with_pre} {label} {flops / 1e9:6.1f} GF {nbytes / 1e6:4.0f} MB:"
            ref, graphs = None, []
            for sset in settings:
                for kv in sset.split(","):
                    k, v = kv.split(":")
                    L.clc_set_tuning(int(k), int(v))
                for i in range(nbuf):
                    fn(i)
                torch.cuda.synchronize()
                if ref is None:
                    ref = outs[0].clone()
                same = bool(torch.equal(ref, outs[0]))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for i in range(reps):
                        fn(i % nbuf)
                g.replay()
                torch.cuda.synchronize()
                graphs.append((sset, g, same, []))
            for _ in range(7):   # interleaved rounds: every setting sees the same clock / thermal history
                for sset, g, same, ts in graphs:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g.replay(); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3 / reps)
            for sset, g, same, ts in graphs:
                us = sorted(ts)[len(ts) // 2]
                line += f"  [{sset}] {us:6.1f} us {flops / us / 1e6:5.1f} TF {nbytes / us / 1e3:4.0f} GB/s{'' if same else ' !!DIFF'}"
            print(line, flush=True)
    L.clc_set_tuning(13, 1)
    L.clc_set_tuning(12, 0)


if __name__ == "__main__":
    main()
