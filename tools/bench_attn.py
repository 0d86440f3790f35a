"""Window-attention micro-benchmark (forward / backward through the autograd op, launches queued behind a spin kernel): python tools/bench_attn.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clc_amd import ops

CL = torch.channels_last
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for N, C, H, W, heads, ws in ((8, 64, 128, 128, 8, 8), (8, 64, 64, 64, 4, 8), (8, 64, 32, 32, 2, 8), (16, 128, 16, 16, 8, 8), (16, 64, 8, 8, 2, 4)):
    g = torch.Generator().manual_seed(0)
    nb = 3
    qkvs = [torch.randn(N, 3 * C, H, W, generator=g).to(dev).contiguous(memory_format=CL).requires_grad_(True) for _ in range(nb)]
    rb = (torch.randn(heads, 2 * ws - 1, 2 * ws - 1, generator=g) * 0.1).to(dev).requires_grad_(True)
    douts = [torch.randn(N, C, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(nb)]
    for shift in (False, True):
        def fwd(i):
            with torch.no_grad():
                ops.window_attention(qkvs[i], rb, heads, ws, shift)
        outs = [ops.window_attention(qkvs[i], rb, heads, ws, shift) for i in range(nb)]
        def bwd(i):
            torch.autograd.grad(outs[i], [qkvs[i], rb], douts[i], retain_graph=True)
        line = f"C{C} h{heads} ws{ws} {N}x{H}x{W} shift={int(shift)}:"
        from clc_amd import lib as _lib
        for key16 in [int(v) for v in os.environ.get("ATTN_4B", "0,1").split(",")]:
          _lib.load().clc_set_tuning(16, key16)
          line += f"  [4B={key16}]"
          for label, fn in (("fwd", fwd), ("bwd", bwd)):
              for i in range(nb):
                  fn(i)
              torch.cuda.synchronize()
              ts = []
              for _ in range(5):   # eager launches queued behind a spin kernel: the GPU runs them back to back
                  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                  torch.cuda._sleep(int(0.01 * 2.4e9))
                  e0.record()
                  for i in range(reps):
                      fn(i % nb)
                  e1.record(); torch.cuda.synchronize()
                  ts.append(e0.elapsed_time(e1) * 1e3 / reps)
              line += f"  {label} {sorted(ts)[2]:7.1f} us"
        print(line, flush=True)
