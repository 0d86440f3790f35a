import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clc_amd import ops
dev = torch.device("cuda:0"); CL = torch.channels_last
g = torch.Generator().manual_seed(21)
x = (torch.randn(16, 640, 16, 16, generator=g) * 0.5).to(dev).contiguous(memory_format=CL)
w1 = (torch.randn(224, 640, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=CL).requires_grad_(True)
w2 = (torch.randn(224, 640, 3, 3, generator=g) * 0.02).to(dev).contiguous(memory_format=CL).requires_grad_(True)
b1, b2 = torch.randn(224, generator=g).to(dev).requires_grad_(True), torch.randn(224, generator=g).to(dev).requires_grad_(True)
with torch.no_grad():
    y_eval = ops.conv2d(x, w1, b1, act=ops.ACT_LRELU, w2=w2, b2=b2)
    for n in (1, 2, 4, 8):
        y_one = ops.conv2d(x[:n].contiguous(memory_format=CL), w1, b1, act=ops.ACT_LRELU)
        d = (y_one - y_eval[:n]).abs().max().item()
        print("single bs", n, "vs paired bs16:", "same" if d == 0 else f"DIFF {d:.3e}", "| is CL-contig:", x[:n].is_contiguous(memory_format=CL))
    y_p2 = ops.conv2d(x[:2].contiguous(memory_format=CL), w1, b1, act=ops.ACT_LRELU, w2=w2, b2=b2)
    print("paired bs2 image0 vs paired bs16 image0:", torch.equal(y_p2[:1], y_eval[:1]))
