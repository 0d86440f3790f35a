"""Graph-replayed A/B of clc_winattn_bwd / fwd (raw C-ABI calls) under key 16 settings, operands rotated through buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clc_amd import ops, lib
dev = torch.device("cuda:0"); CL = torch.channels_last; L = lib.load()
# settings: name -> (tuning key, value); default A/B: the two-workgroups-per-window backward on small grids (key 18)
KEYS = {"16:0": (16, 0), "16:7": (16, 7), "18:0": (18, 0), "18:1": (18, 1)}
SETTINGS = sys.argv[1:] or ["18:0", "18:1"]
for (N, C, H, W, heads, ws) in ((8, 64, 128, 128, 8, 8), (8, 64, 64, 64, 4, 8), (8, 64, 32, 32, 2, 8), (16, 128, 16, 16, 8, 8)):
    nb = 4
    g = torch.Generator().manual_seed(0)
    qkvs = [torch.randn(N, 3 * C, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(nb)]
    rb = (torch.randn(heads, 2 * ws - 1, 2 * ws - 1, generator=g) * 0.1).to(dev)
    douts = [torch.randn(N, C, H, W, generator=g).to(dev).contiguous(memory_format=CL) for _ in range(nb)]
    outs = [torch.empty(N, C, H, W, device=dev).contiguous(memory_format=CL) for _ in range(nb)]
    lses = [torch.empty(N * H * W * heads, device=dev) for _ in range(nb)]
    dqkvs = [torch.empty_like(q) for q in qkvs]
    drb = torch.zeros_like(rb)
    nbytes = L.clc_winattn_bwd_workspace_bytes(N, H, W, heads, ws)
    wsb = torch.empty((nbytes + 3) // 4, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    def fwd(i):
        lib.check(L.clc_winattn_fwd(qkvs[i].data_ptr(), 3 * C, rb.data_ptr(), outs[i].data_ptr(), C, lses[i].data_ptr(), N, H, W, C, heads, ws, 1, st()), "f")
    def bwd(i):
        lib.check(L.clc_winattn_bwd(douts[i].data_ptr(), C, qkvs[i].data_ptr(), 3 * C, rb.data_ptr(), outs[i].data_ptr(), C, lses[i].data_ptr(), dqkvs[i].data_ptr(), 3 * C,
                                    drb.data_ptr(), 1, N, H, W, C, heads, ws, 1, wsb.data_ptr(), nbytes, st()), "b")
    for i in range(nb):
        fwd(i)
    line = f"C{C} h{heads} {N}x{H}x{W}:"
    for label, fn in (("fwd", fwd), ("bwd", bwd)):
        graphs = []
        for key in SETTINGS:
            L.clc_set_tuning(*KEYS[key])
            fn(0); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(20):
                    fn(i % nb)
            gr.replay(); torch.cuda.synchronize()
            graphs.append((key, gr, []))
        for _ in range(7):
            for key, gr, ts in graphs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 20)
        for key, gr, ts in graphs:
            line += f"  {label}[{key}] {sorted(ts)[3]:7.1f} us"
    print(line, flush=True)
L.clc_set_tuning(16, 3)
L.clc_set_tuning(18, 1)
