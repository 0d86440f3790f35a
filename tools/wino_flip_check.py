"""Does the Winograd forward (tuning key 23 bit 0) move a quantization decision on the two gradient-parity inputs of tests/test_model_gpu.py?
Counts hyper-latent / latent symbol differences between a recorded forward on the direct kernels (23:2) and on the Winograd kernel (23:3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from clc_amd import lib
from test_model_gpu import _pair, _inputs
dev = torch.device("cuda:0")
L = lib.load()
for name, B, size, wire in (("bs2-256", 2, 256, False), ("wire-512", 1, 512, True)):
    o, p = _pair("clc", 1, dev)
    p.wire_clm = wire
    x, refs = _inputs(B, 1, size=size)
    xd, rd = x.to(dev), [r.to(dev) for r in refs]
    outs = {}
    for t in (2, 3):
        L.clc_set_tuning(23, t)
        out = p(xd, rd)          # grad mode on: recorded forward
        outs[t] = {"y": out["para"]["y"].detach(), "mu": out["para"]["means"].detach(), "sc": out["para"]["scales"].detach(),
                   "lz": out["likelihoods"]["z"].detach(), "ly": out["likelihoods"]["y"].detach(), "xh": out["x_hat"].detach()}
    a, b = outs[2], outs[3]
    zrel = ((a["lz"] - b["lz"]).abs() / a["lz"].clamp_min(1e-9))
    sy = (torch.round(a["y"] - a["mu"]) != torch.round(b["y"] - b["mu"])).sum().item()
    print(name, "y max rel diff", ((a["y"] - b["y"]).abs().max() / a["y"].abs().max()).item(), "| z likelihood elems differing >1%:", (zrel > 1e-2).sum().item(),
          "| y symbol flips:", sy, "| scales max abs diff", (a["sc"] - b["sc"]).abs().max().item(), "| bpp diff",
          ((torch.log(a["lz"]).sum() + torch.log(a["ly"]).sum()) - (torch.log(b["lz"]).sum() + torch.log(b["ly"]).sum())).item() / (-0.6931 * B * size * size))
