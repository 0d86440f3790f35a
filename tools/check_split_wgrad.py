"""Filter gradients with f32 products formed from three-way bf16 splits (tuning key 24; csrc/conv_wgrad.hip split3 / MFMA_SPLIT6) against the native
v_mfma_f32_32x32x2_f32 kernels: error of both against an fp64 reference, and time (hipGraph-replayed grouped launch).   python tools/check_split_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from clc_amd import ops

dev = torch.device("cuda:0")
MODE = int(os.environ.get("SPLIT_MODE", "3"))
L = ops._L()
CL = torch.channels_last
g = torch.Generator().manual_seed(0)
SHAPES = [("128->128 k3 @8x128^2", 8, 128, 128, 128, 128, 3), ("64->64 k3 @8x128^2", 8, 128, 128, 64, 64, 3), ("128->512 k3 @8x64^2", 8, 64, 64, 128, 512, 3),
          ("128->128 k1 @8x128^2", 8, 128, 128, 128, 128, 1), ("320->320 k3 @8x32^2", 8, 32, 32, 320, 320, 3), ("64->64 k1 @8x128^2", 8, 128, 128, 64, 64, 1), ("64->256 k1 @8x128^2", 8, 128, 128, 64, 256, 1), ("192->64 k1 @8x128^2", 8, 128, 128, 192, 64, 1), ("640->224 k3 @8x16^2", 8, 16, 16, 640, 224, 3)]
for name, N, H, W, Cin, Cout, ks in SHAPES:
    x = torch.randn(N, Cin, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    dy = torch.randn(N, Cout, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    xd, dyd = x.double().cpu(), dy.double().cpu()
    ref = torch.nn.grad.conv2d_weight(xd, (Cout, Cin, ks, ks), dyd, padding=ks // 2).permute(0, 2, 3, 1).reshape(Cout, -1)    # kernel layout [Co][kh][kw][Ci]
    res = {}
    for mode in (0, MODE):
        old = L.clc_set_tuning(24, mode)
        try:
            dw = torch.zeros(Cout * ks * ks * Cin, device=dev)
            prob = dict(x=x, dy=dy, ks=ks, stride=1, pad=ks // 2, Cout=Cout, Cin=Cin, want_bias=False, dw_out=dw, accumulate=0)
            def run():
                return ops.wgrad_batched([prob])
            keep = run(); torch.cuda.synchronize()
            err = ((dw.double().cpu().view(Cout, -1) - ref).abs().max() / ref.abs().max()).item()
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                keep = run()
            torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                keep = run()
            ts = []
            for _ in range(11):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res[mode] = (sorted(ts)[5], err)
        finally:
            L.clc_set_tuning(24, old)
    fl = 2.0 * N * H * W * ks * ks * Cin * Cout
    print(f"{name:24s} f32 MFMA {res[0][0] * 1e3:7.1f} us {fl / res[0][0] / 1e9:6.1f} TF err {res[0][1]:.2e} | bf16 x6 {res[MODE][0] * 1e3:7.1f} us {fl / res[MODE][0] / 1e9:6.1f} TF-eq err {res[MODE][1]:.2e} | x{res[0][0] / res[MODE][0]:.2f}", flush=True)
