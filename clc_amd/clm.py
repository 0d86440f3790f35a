"""Conditional Latent Matching modules of /root/reference/models/CLM.py on the HIP engine (forward only).

Same class names, constructor arguments and parameter names as the reference file (``CLM``, ``SimpleCLM``,
``DeformableAlignment``): ``feature_transform.{0,2}``, ``alignment.{offset_conv,modulation_conv}``, ``attention_conv``,
``fusion_conv.{0,2}``.  The reference module is an orphan (nothing imports it) whose deformable alignment is a Python
quadruple loop; here the similarity softmax is reduced to its column sums on the fly (the only thing the reference's
``weighted_x`` loop uses), the 9-tap modulated bilinear sampling is one kernel, and all convs run on the implicit GEMM.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib as _lib
from . import ops
from .layers import Conv2d
from .lib import ACT_NONE, ACT_RELU, ACT_SIGMOID
from .ops import CL, _L, _stream, dense, new_act, nhwc


def _conv_infer(conv: Conv2d, x, act=ACT_NONE):
    with torch.no_grad():
        return ops.conv_raw(x, ops.to_kernel_weight(conv.weight), conv.bias, ks=conv.kernel_size[0], stride=1, act=act)


class DeformableAlignment(nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.offset_conv = Conv2d(input_dim * 2, 2 * 3 * 3, 3)
        self.modulation_conv = Conv2d(input_dim * 2, 3 * 3, 3)

    @torch.no_grad()
    def forward(self, x, colsum):
        """x: reference latent [B,C,H,W]; colsum: [B, H*W] column sums of the similarity softmax."""
        x = dense(x)
        B, Cc, H, W = x.shape
        cat = new_act(B, 2 * Cc, H, W, x)
        L = _L()
        _lib.check(L.clc_copy2d(x.data_ptr(), Cc, cat.data_ptr(), 2 * Cc, B * H * W, Cc, _stream()), "clc_copy2d")
        _lib.check(L.clc_clm_scale_rows(x.data_ptr(), Cc, colsum.data_ptr(), cat.data_ptr() + 4 * Cc, 2 * Cc, B * H * W, Cc, _stream()), "clc_clm_scale_rows")
        offset = _conv_infer(self.offset_conv, cat)                      # [B,18,H,W] pixel-major
        modulation = _conv_infer(self.modulation_conv, cat, ACT_SIGMOID)  # sigmoid fused in the epilogue
        out = new_act(B, Cc, H, W, x)
        _lib.check(L.clc_clm_deform(x.data_ptr(), Cc, offset.data_ptr(), 18, modulation.data_ptr(), 9, out.data_ptr(), Cc, B, H, W, Cc, _stream()), "clc_clm_deform")
        return out


def _fuse(feats, atts, y, gate):
    y = dense(y)
    B, Cc, H, W = y.shape
    feats = [dense(f) for f in feats]
    atts = [dense(a) for a in atts]
    out = new_act(B, Cc, H, W, y)
    fp = _lib.ptr_array([f.data_ptr() for f in feats])
    ap = _lib.ptr_array([a.data_ptr() for a in atts])
    _lib.check(_L().clc_clm_fuse(fp, ap, len(feats), Cc, 1, y.data_ptr(), Cc, out.data_ptr(), Cc, B * H * W, Cc, int(gate), _stream()), "clc_clm_fuse")
    return out


class CLM(nn.Module):
    """Conditional Latent Matching (CLM.py:62-128)."""

    def __init__(self, input_dim, temperature=0.5):
        super().__init__()
        self.temperature = temperature
        self.feature_transform = nn.Sequential(Conv2d(input_dim, input_dim, 1), nn.ReLU(inplace=True), Conv2d(input_dim, input_dim, 1))
        self.alignment = DeformableAlignment(input_dim)
        self.attention_conv = Conv2d(input_dim, 1, 1)
        self.fusion_conv = nn.Sequential(Conv2d(input_dim, input_dim, 3), nn.ReLU(inplace=True), Conv2d(input_dim, input_dim, 3))

    def _ft(self, x):
        return _conv_infer(self.feature_transform[2], _conv_infer(self.feature_transform[0], x, ACT_RELU))

    @torch.no_grad()
    def forward(self, y, y_refs):
        if not y.is_cuda:
            raise _lib.ClcError("clc_amd.clm runs on the GPU only")
        y = y.float().contiguous(memory_format=CL)
        B, Cc, H, W = y.shape
        if len(y_refs) > 8:
            raise ValueError("at most 8 reference latents")
        y_t = self._ft(y)
        aligned, atts = [], []
        L = _L()
        for y_ref in y_refs:
            y_ref = y_ref.float().contiguous(memory_format=CL)
            y_ref_t = self._ft(y_ref)
            nbytes = L.clc_clm_sim_colsum_workspace_bytes(B, H * W)
            ws = torch.empty((nbytes + 3) // 4, device=y.device, dtype=torch.float32)
            colsum = torch.empty(B * H * W, device=y.device, dtype=torch.float32)
            _lib.check(L.clc_clm_sim_colsum(y_t.data_ptr(), Cc, y_ref_t.data_ptr(), Cc, B, H * W, Cc, float(self.temperature), colsum.data_ptr(),
                                            ws.data_ptr(), nbytes, _stream()), "clc_clm_sim_colsum")
            a = self.alignment(y_ref, colsum)
            aligned.append(a)
            atts.append(_conv_infer(self.attention_conv, a))
        s = _fuse(aligned, atts, y, gate=False)
        return _conv_infer(self.fusion_conv[2], _conv_infer(self.fusion_conv[0], s, ACT_RELU))


class SimpleCLM(nn.Module):
    """Simplified variant (CLM.py:130-187)."""

    def __init__(self, input_dim, temperature=0.5):
        super().__init__()
        self.temperature = temperature
        self.feature_transform = Conv2d(input_dim, input_dim, 1)
        self.attention_conv = Conv2d(input_dim, 1, 1)
        self.fusion_conv = nn.Sequential(Conv2d(input_dim, input_dim, 3), nn.ReLU(inplace=True))

    @torch.no_grad()
    def forward(self, y, y_refs):
        if not y.is_cuda:
            raise _lib.ClcError("clc_amd.clm runs on the GPU only")
        y = y.float().contiguous(memory_format=CL)
        feats = [_conv_infer(self.feature_transform, r.float().contiguous(memory_format=CL)) for r in y_refs]
        atts = [_conv_infer(self.attention_conv, f) for f in feats]
        s = _fuse(feats, atts, y, gate=True)
        return _conv_infer(self.fusion_conv[0], s, ACT_RELU)
