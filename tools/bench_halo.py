"""A/B of conv_halo3x3_kernel (csrc/conv_halo.hip) against the LDS-tiled kernels on the layers it takes, one process, hipGraph-replayed
(GPU time only, median of `reps` replays of 10 back-to-back launches on rotating operands): python tools/bench_halo.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clc_amd import ops

CL = torch.channels_last
dev = torch.device("cuda:0")
ops._L().clc_set_tuning(22, 3)   # both instantiations (the 64-channel one is off by default)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
SHAPES = [  # name, N, H, W, Cin, Cout, shuffle, transposed
    ("128->128 @8x128^2 fwd", 8, 128, 128, 128, 128, False, False),
    ("128->128 @8x64^2 fwd", 8, 64, 64, 128, 128, False, False),
    ("128->128 @8x64^2 dgrad", 8, 64, 64, 128, 128, False, True),
    ("128->512 @8x64^2 fwd+shuffle", 8, 64, 64, 128, 512, True, False),
    ("128->512 @8x32^2 fwd+shuffle", 8, 32, 32, 128, 512, True, False),
    ("64->64 @8x128^2 fwd", 8, 128, 128, 64, 64, False, False),
    ("64->64 @8x128^2 dgrad", 8, 128, 128, 64, 64, False, True),
    ("64->64 @8x64^2 fwd", 8, 64, 64, 64, 64, False, False),
]
g = torch.Generator().manual_seed(0)
NB = 4   # rotating operand sets (8 x 128 x 128 x 128 x 4 B = 67 MB each: past the L2s, inside the Infinity Cache — as in the step)
for name, N, H, W, Cin, Cout, shuf, tr in SHAPES:
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=CL)
    b = torch.randn(Cout, generator=g).to(dev)
    if tr:   # "x" = dY [N, Cout, H, W] -> dX [N, Cin, H, W]
        xs = [torch.randn(N, Cout, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(NB)]
        outs = [ops.new_act(N, Cin, H, W, xs[0]) for _ in range(NB)]
        wt = ops.filter_transpose(w, Cout, 9, Cin).view(Cin, -1)
        pk = ops.halo_pack(wt, Cin, Cout)
        call = lambda i, wpk: ops.conv_raw(xs[i % NB], wt, None, ks=3, stride=1, pad=1, transposed=True, out_hw=(H, W), out=outs[i % NB], wpk=wpk)
    else:
        xs = [torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(NB)]
        outs = [ops.new_act(N, Cout // 4, 2 * H, 2 * W, xs[0]) if shuf else ops.new_act(N, Cout, H, W, xs[0]) for _ in range(NB)]
        pk = ops.halo_pack(w, Cout, Cin)
        call = lambda i, wpk: ops.conv_raw(xs[i % NB], w, b, ks=3, stride=1, act=1, shuffle=shuf, out=outs[i % NB], wpk=wpk)
    flops = 2.0 * N * H * W * 9 * Cin * Cout
    res = {}
    for label, wpk in (("tiled", None), ("halo", pk)):
        def fn():
            for i in range(10):
                call(i, wpk)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        res[label] = sorted(ts)[len(ts) // 2]
        res[label + "_out"] = outs[0].clone()
    same = torch.equal(res["tiled_out"], res["halo_out"])
    print(f"{name:32s} tiled {res['tiled'] * 1e3:7.1f} us {flops / res['tiled'] / 1e9:6.1f} TF | halo {res['halo'] * 1e3:7.1f} us {flops / res['halo'] / 1e9:6.1f} TF | "
          f"x{res['tiled'] / res['halo']:.3f} | same bits: {same}", flush=True)
