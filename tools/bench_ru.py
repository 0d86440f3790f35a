"""A/B of the fused ResidualUnit forward (one launch) against the three split-K layer launches, graph-replayed.
   python tools/bench_ru.py            (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clc_amd import layers, ops

from clc_amd import lib as _lib
dev = torch.device("cuda:0")
ABL = [int(a) for a in os.environ.get("RU_ABLATE", "0").split(",")]
torch.manual_seed(0)
for abl in ABL:
  _lib.load().clc_set_tuning(12, abl)
  print("ablate", abl)
  for sets, n in ((4, 32), (1, 8)):
      units = [layers.ResidualUnit(128).to(dev) for _ in range(sets)]
      pair = None if sets == 1 else (units[1] if sets == 2 else tuple(units[1:]))
      x = torch.randn(n, 128, 16, 16, device=dev).contiguous(memory_format=torch.channels_last)
      res = {}
      for fused in (0, 1):
          ops.FUSED_RU = fused
          with torch.no_grad():
              for _ in range(3):
                  units[0](x, pair=pair)
              torch.cuda.synchronize()
              g = torch.cuda.CUDAGraph()
              s = torch.cuda.Stream()
              with torch.cuda.stream(s):
                  with torch.cuda.graph(g, stream=s):
                      for _ in range(20):
                          y = units[0](x, pair=pair)
              g.replay()
              torch.cuda.synchronize()
              e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              best = 1e9
              for _ in range(5):
                  e0.record()
                  for _ in range(10):
                      g.replay()
                  e1.record()
                  torch.cuda.synchronize()
                  best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
          res[fused] = best
      fl = 2.0 * n * 256 * (128 * 64 * 2 + 9 * 64 * 64)
      print(f"sets {sets} batch {n}: three launches {res[0]:.1f} us, fused {res[1]:.1f} us ({fl / res[1] / 1e6:.1f} TF)", flush=True)
