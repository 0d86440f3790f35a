import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from clc_amd import ops, lib
dev = torch.device("cuda:0")
CL = torch.channels_last
L = lib.load()
for (N, C, H, W, heads, ws) in ((16, 128, 16, 16, 8, 8), (2, 64, 32, 32, 8, 8)):
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(N, 3 * C, H, W, generator=g) * 1.0).to(dev).contiguous(memory_format=CL).requires_grad_(True)
    rb = (torch.randn(heads, 2 * ws - 1, 2 * ws - 1, generator=g) * 0.1).to(dev).requires_grad_(True)
    dout = torch.randn(N, C, H, W, generator=g).to(dev).contiguous(memory_format=CL)
    res = {}
    for k in (0, 1):
        L.clc_set_tuning(16, k)
        out = ops.window_attention(qkv, rb, heads, ws, True)
        gq, gb = torch.autograd.grad(out, [qkv, rb], dout)
        res[k] = (out.detach().double(), gq.double(), gb.double())
    # fp64 reference in torch
    hd = C // heads
    x = qkv.detach().double().permute(0, 2, 3, 1)  # N H W 3C
    x = torch.roll(x, shifts=(-(ws // 2), -(ws // 2)), dims=(1, 2))
    def win(t):  # N H W c -> windows
        n, h, w, c = t.shape
        return t.view(n, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(n, h // ws, w // ws, ws * ws, c)
    xw = win(x)
    q, k_, v = xw[..., :C], xw[..., C:2 * C], xw[..., 2 * C:]
    def heads_split(t):
        return t.reshape(*t.shape[:-1], heads, hd).transpose(-2, -3)   # ... heads T hd
    q, k_, v = heads_split(q), heads_split(k_), heads_split(v)
    s = q @ k_.transpose(-1, -2) / hd ** 0.5
    idx = torch.arange(ws)
    rel = (idx[:, None] - idx[None, :]) + ws - 1
    iy = torch.arange(ws * ws) // ws; ix = torch.arange(ws * ws) % ws
    bias = rb.detach().double()[:, (iy[:, None] - iy[None, :] + ws - 1), (ix[:, None] - ix[None, :] + ws - 1)]
    s = s + bias.to(s.device)
    nwy, nwx = H // ws, W // ws
    m = torch.zeros(nwy, nwx, ws, ws, ws, ws, dtype=torch.bool, device=dev)
    sh = ws - ws // 2
    m[-1, :, :sh, :, sh:, :] = True; m[-1, :, sh:, :, :sh, :] = True
    m[:, -1, :, :sh, :, sh:] = True; m[:, -1, :, sh:, :, :sh] = True
    m = m.reshape(nwy, nwx, ws * ws, ws * ws)
    s = s.masked_fill(m[None, :, :, None], float("-inf"))
    o = torch.softmax(s, -1) @ v
    o = o.transpose(-2, -3).reshape(N, nwy, nwx, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, C)
    o = torch.roll(o, shifts=(ws // 2, ws // 2), dims=(1, 2)).permute(0, 3, 1, 2)
    for k in (0, 1):
        e = (res[k][0] - o).abs().max().item() / o.abs().max().item()
        print(f"C{C} hd{hd} 4B={k}: fwd rel err vs fp64 {e:.3e};  |4B1-4B0| fwd {(res[1][0]-res[0][0]).abs().max().item():.3e} dqkv {(res[1][1]-res[0][1]).abs().max().item() / res[0][1].abs().max().item():.3e} dbias {(res[1][2]-res[0][2]).abs().max().item() / res[0][2].abs().max().item():.3e}")
