"""Winograd F(2x2, 3x3) kernel (csrc/conv_wino.hip) against the halo / tiled kernels on the layers it takes: accuracy against an fp64 convolution
and time, one process, hipGraph-replayed (median of `reps` replays of 10 back-to-back launches on rotating operands).  python tools/bench_wino.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from clc_amd import ops

CL = torch.channels_last
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
L = ops._L()
MODE = int(os.environ.get("WINO_MODE", "7"))
L.clc_set_tuning(23, MODE)
SHAPES_ALL = [  # name, N, H, W, Cin, Cout, shuffle, transposed
    ("128->128 @8x128^2 fwd", 8, 128, 128, 128, 128, False, False),
    ("128->128 @8x128^2 dgrad", 8, 128, 128, 128, 128, False, True),
    ("128->128 @8x64^2 fwd", 8, 64, 64, 128, 128, False, False),
    ("128->512 @8x64^2 fwd+shuffle", 8, 64, 64, 128, 512, True, False),
    ("128->512 @8x32^2 fwd+shuffle", 8, 32, 32, 128, 512, True, False),
    ("128->512 @8x64^2 dgrad (512 ch in)", 8, 64, 64, 128, 512, False, True),
    ("64->64 @8x128^2 fwd", 8, 128, 128, 64, 64, False, False),
    ("64->64 @8x128^2 dgrad", 8, 128, 128, 64, 64, False, True),
    ("64->64 @8x64^2 fwd", 8, 64, 64, 64, 64, False, False),
    ("64->256 @8x64^2 fwd+shuffle", 8, 64, 64, 64, 256, True, False),
    ("64->64 @8x64^2 dgrad", 8, 64, 64, 64, 64, False, True),
    ("320->320 @8x32^2 fwd", 8, 32, 32, 320, 320, False, False),
    ("320->320 @8x32^2 dgrad", 8, 32, 32, 320, 320, False, True),
    ("128->128 @8x32^2 fwd", 8, 32, 32, 128, 128, False, False),
    ("128->128 @8x32^2 dgrad", 8, 32, 32, 128, 128, False, True),
    ("128->512 @8x32^2 dgrad (512 ch in)", 8, 32, 32, 128, 512, False, True),
    ("64->64 @8x32^2 fwd", 8, 32, 32, 64, 64, False, False),
]
SHAPES = SHAPES_ALL[:int(os.environ.get('WINO_SHAPES', len(SHAPES_ALL)))]
g = torch.Generator().manual_seed(0)
NB = 4
for name, N, H, W, Cin, Cout, shuf, tr in SHAPES:
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=CL)
    b = torch.randn(Cout, generator=g).to(dev)
    if tr:   # "x" = dY [N, Cout, H, W] -> dX [N, Cin, H, W]
        xs = [torch.randn(N, Cout, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(NB)]
        outs = [ops.new_act(N, Cin, H, W, xs[0]) for _ in range(NB)]
        wt = ops.filter_transpose(w, Cout, 9, Cin).view(Cin, -1)
        u = ops.wino_pack(wt, Cin, Cout, flip=True)
        pk = ops.halo_pack(wt, Cin, Cout) if Cout == 128 else None
        call = lambda i, **k: ops.conv_raw(xs[i % NB], wt, None, ks=3, stride=1, pad=1, transposed=True, out_hw=(H, W), out=outs[i % NB], **k)
        ref = F.conv_transpose2d(xs[0].double().cpu(), w.double().cpu(), padding=1)
    else:
        xs = [torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(NB)]
        outs = [ops.new_act(N, Cout // 4, 2 * H, 2 * W, xs[0]) if shuf else ops.new_act(N, Cout, H, W, xs[0]) for _ in range(NB)]
        u = ops.wino_pack(w, Cout, Cin)
        pk = ops.halo_pack(w, Cout, Cin) if Cin == 128 else None
        call = lambda i, **k: ops.conv_raw(xs[i % NB], w, b, ks=3, stride=1, act=0, shuffle=shuf, out=outs[i % NB], **k)
        ref = F.conv2d(xs[0].double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
        if shuf:
            ref = F.pixel_shuffle(ref, 2)
    flops = 2.0 * N * H * W * 9 * Cin * Cout
    res = {}
    for label, kw in (("direct", dict(wpk=pk)), ("wino", dict(wwino=u))):
        def fn():
            for i in range(10):
                call(i, **kw)
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
        o = outs[0].double().cpu()
        res[label + "_err"] = ((o - ref).abs().max() / ref.abs().max()).item()
    print(f"{name:36s} direct {res['direct'] * 1e3:7.1f} us {flops / res['direct'] / 1e9:6.1f} TF (err {res['direct_err']:.1e}) | wino {res['wino'] * 1e3:7.1f} us "
          f"{flops / res['wino'] / 1e9:6.1f} TF-equivalent (err {res['wino_err']:.1e}) | x{res['direct'] / res['wino']:.3f}", flush=True)

if hasattr(L, "clc_wino_debug") or os.environ.get("CLC_LIB_PATH"):   # a -DWINO_AB=32 build: phase time stamps of one item of the LAST wide launch
    import ctypes
    try:
        fn = L.clc_wino_debug
        buf = (ctypes.c_ulonglong * 32)()
        torch.cuda.synchronize()
        fn(buf)
        t = list(buf)
        names = {0: "item start", 1: "halo landed + barrier", 2: "first transform", 11: "after last chunk", 12: "DMA wait", 13: "round 0 barrier", 14: "round 0 staged",
                 15: "round 0 stored", 16: "round 1 barrier", 17: "round 1 staged", 18: "round 1 stored"}
        for k in range(1, 19):
            print(f"  stamp {k:2d} {names.get(k, 'chunk %d barrier' % (k - 3)):24s} +{t[k] - t[k - 1]:7d} ticks  (total {t[k] - t[0]})")
        if t[19] and t[20]:
            print(f"  round 0: staged -> y in registers {t[19] - t[14]} ticks, -> first pixel pair stored {t[20] - t[19]}, -> second {t[15] - t[20]}")
    except AttributeError:
        pass
