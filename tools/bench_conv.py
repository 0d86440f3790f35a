"""Micro-benchmark of single conv launches through the C ABI (forward / data-grad / weight-grad).
usage: python tools/bench_conv.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clc_amd import ops

CL = torch.channels_last
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else None
SHAPES = [  # name, N, Cin, H, W, Cout, ks, stride
    ("c128_128_3x3_128", 8, 128, 128, 128, 128, 3, 1),
    ("c64_64_3x3_128", 8, 64, 128, 128, 64, 3, 1),
    ("c128_512_3x3_64", 8, 128, 64, 64, 512, 3, 1),
    ("c128_128_1x1_128", 8, 128, 128, 128, 128, 1, 1),
    ("c64_256_1x1_128", 8, 64, 128, 128, 256, 1, 1),
    ("c448_224_3x3_16", 8, 448, 16, 16, 224, 3, 1),
    ("c64_64_3x3_16", 8, 64, 16, 16, 64, 3, 1),
    ("c128_64_1x1_16", 8, 128, 16, 16, 64, 1, 1),
    ("c128_128_3x3_32", 8, 128, 32, 32, 128, 3, 1),
    ("n16_c64_64_3x3_16", 16, 64, 16, 16, 64, 3, 1),
    ("n16_c128_64_1x1_16", 16, 128, 16, 16, 64, 1, 1),
    ("n16_c448_224_3x3_16", 16, 448, 16, 16, 224, 3, 1),
    ("n16_c224_128_3x3_16", 16, 224, 16, 16, 128, 3, 1),
    ("c224_128_3x3_16", 8, 224, 16, 16, 128, 3, 1),
    ("c128_512_1x1_16", 8, 128, 16, 16, 512, 1, 1),
    ("c128_384_1x1_16", 8, 128, 16, 16, 384, 1, 1),
    ("c384_128_1x1_16", 8, 384, 16, 16, 128, 1, 1),
    ("c128_128_1x1_16", 8, 128, 16, 16, 128, 1, 1),
    ("c128_64_3x3_16", 8, 128, 16, 16, 64, 3, 1),
    ("c192_512_3x3_4", 8, 192, 4, 4, 512, 3, 1),
    ("c128_1280_3x3_8", 8, 128, 8, 8, 1280, 3, 1),
    ("n16_c704_224_3x3_16", 16, 704, 16, 16, 224, 3, 1),
    ("c768_224_3x3_16", 8, 768, 16, 16, 224, 3, 1),
    ("c512_224_3x3_16", 8, 512, 16, 16, 224, 3, 1),
]
g = torch.Generator().manual_seed(0)
for name, N, Cin, H, W, Cout, ks, s in SHAPES:
    if only and only != name:
        continue
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) * 0.05).to(dev).contiguous(memory_format=CL)
    b = torch.randn(Cout, generator=g).to(dev)
    OH, OW = (H + 2 * (ks // 2) - ks) // s + 1, (W + 2 * (ks // 2) - ks) // s + 1
    dy = torch.randn(N, Cout, OH, OW, generator=g).to(dev).contiguous(memory_format=CL)
    wt = ops.filter_transpose(w, Cout, ks * ks, Cin)
    flops = 2.0 * N * OH * OW * ks * ks * Cin * Cout
    def fwd(): ops.conv_raw(x, w, b, ks=ks, stride=s, act=1)
    def dgrad(): ops.conv_raw(dy, wt.view(Cin, -1), None, ks=ks, stride=s, pad=ks // 2, transposed=True, out_hw=(H, W))
    def wgrad(): ops.wgrad_raw(x, dy, ks=ks, stride=s, pad=ks // 2, Cout=Cout, Cin=Cin, want_bias=True)
    res = []
    from clc_amd import lib as _lib
    variants = [(fwd, "fwd", 2), (dgrad, "dgrad", 2), (wgrad, "wgrad", None)]
    if os.environ.get("AB", "0") == "1":   # A/B of the DMA K-loop variants in ONE process (tuning key 0)
        variants = [(fwd, "fwd.v1", 1), (fwd, "fwd.v2", 2), (dgrad, "dgrad.v1", 1), (dgrad, "dgrad.v2", 2), (wgrad, "wgrad", None)]
    for fn, label, tune in variants:
        if tune is not None:
            _lib.load().clc_set_tuning(0, tune)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        # capture the launches in a hipGraph so the number is GPU time, not Python/ctypes launch overhead
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for _ in range(reps): fn()
        g_.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g_.replay()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res.append(f"{label} {us:8.1f} us {flops / us / 1e6:6.1f} TF")
    print(f"{name:20s} {flops/1e9:7.2f} GF | " + " | ".join(res), flush=True)
