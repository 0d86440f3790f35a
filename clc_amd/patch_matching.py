"""Patch-matching side-information ops — the numeric core of /root/reference/models/Patch_Matching.py on HIP kernels.

Same function names / argument meaning as the reference file for the self-contained numeric functions
(``L2_or_pearson_corr`` :854-910, ``create_gaussian_masks`` :779-807, ``SI_Wraper`` :218-240,
``SI_Finder_at_Image_Domain`` :87-122, ``rgb_transform`` :926-934, ``reduce_mean_and_std_normalize_images`` :913-924).
The reference file itself is an un-importable orphan (turtle / cv2 / compressai_local imports, pdb traces); its
nn.Module zoo is out of scope.  Tensors are planar NCHW fp32 on the GPU; forward only.
"""
from __future__ import annotations

import torch

from . import lib as _lib
from .ops import _L, _stream


def _chk(t, what):
    if not t.is_cuda:
        raise _lib.ClcError(f"{what}: GPU tensors only (no CPU fallback)")
    return t.float().contiguous()


def rgb_transform_normalized(x, in_scale=1.0):
    """rgb_transform(reduce_mean_and_std_normalize_images(x * in_scale)) fused: [N,3,H,W] -> [N,3,H,W]."""
    x = _chk(x, "rgb_transform_normalized")
    N, C, H, W = x.shape
    assert C == 3
    out = torch.empty_like(x)
    _lib.check(_L().clc_pm_prep(x.data_ptr(), out.data_ptr(), N, H, W, float(in_scale), _stream()), "clc_pm_prep")
    return out


def create_gaussian_masks(img_h, img_w, patch_h, patch_w, device="cuda"):
    P = (img_h * img_w) // (patch_h * patch_w)
    out = torch.empty((1, P, img_h - patch_h + 1, img_w - patch_w + 1), device=device, dtype=torch.float32)
    _lib.check(_L().clc_pm_gauss_mask(out.data_ptr(), img_h, img_w, patch_h, patch_w, _stream()), "clc_pm_gauss_mask")
    return out


def L2_or_pearson_corr(x, y, patch_h, patch_w, mask=None):
    """x: patches [P,C,ph,pw]; y: image [1,C,H,W] -> Pearson correlation map [1,P,H-ph+1,W-pw+1] (times ``mask``)."""
    x, y = _chk(x, "L2_or_pearson_corr"), _chk(y, "L2_or_pearson_corr")
    P, C, ph, pw = x.shape
    assert (ph, pw) == (patch_h, patch_w) and y.shape[0] == 1 and y.shape[1] == C
    H, W = y.shape[2], y.shape[3]
    out = torch.empty((1, P, H - ph + 1, W - pw + 1), device=x.device, dtype=torch.float32)
    L = _L()
    nbytes = L.clc_pm_pearson_workspace_bytes(P, C, H, W, ph, pw)
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
    m = _chk(mask, "mask") if mask is not None else None
    _lib.check(L.clc_pm_pearson(x.data_ptr(), P, y.data_ptr(), C, H, W, ph, pw, m.data_ptr() if m is not None else None, out.data_ptr(),
                                ws.data_ptr(), nbytes, _stream()), "clc_pm_pearson")
    return out


def _topk(cross_corr, k):
    cc = _chk(cross_corr, "topk")
    _, P, ch, cw = cc.shape
    val = torch.empty((P, k), device=cc.device, dtype=torch.float32)
    idx = torch.empty((P, k), device=cc.device, dtype=torch.int32)
    _lib.check(_L().clc_pm_topk(cc.data_ptr(), P, ch * cw, k, val.data_ptr(), idx.data_ptr(), _stream()), "clc_pm_topk")
    return val, idx


def SI_Wraper(cross_corr, patch_h, patch_w, patchs_num, y, k=1, temperature=15, is_stack=False):
    """softmax(top-k value * temperature)-weighted gather of the k best patches of ``y`` per query patch, re-tiled
    ([1, C, H, W]); is_stack=True: the k candidates unweighted, stacked along the channels ([1, k*C, H, W], candidate-major —
    Patch_Matching.py:235-236)."""
    y = _chk(y, "SI_Wraper")
    _, C, H, W = y.shape
    assert patchs_num == (H // patch_h) * (W // patch_w) == cross_corr.shape[1]
    val, idx = _topk(cross_corr, k)
    if is_stack:
        out = torch.empty((1, k * C, H, W), device=y.device, dtype=torch.float32)
        for j in range(k):   # candidate j of every query patch, copied as it is (k = 1 gather without weights)
            idx_j = idx[:, j].contiguous()
            _lib.check(_L().clc_pm_gather(y.data_ptr(), C, H, W, patch_h, patch_w, None, idx_j.data_ptr(), 1, -1.0,
                                          out[:, j * C:(j + 1) * C].data_ptr(), _stream()), "clc_pm_gather")
        return out
    out = torch.empty((1, C, H, W), device=y.device, dtype=torch.float32)
    _lib.check(_L().clc_pm_gather(y.data_ptr(), C, H, W, patch_h, patch_w, val.data_ptr(), idx.data_ptr(), k, float(temperature), out.data_ptr(), _stream()), "clc_pm_gather")
    return out


def SI_Finder_at_Image_Domain(x_dec, y_imgs, patch_h, patch_w, y_dec, mask=None):
    """Per image: best-matching (Pearson x Gaussian prior) patch of y_dec for every patch of x_dec, copied from y_imgs."""
    x_dec, y_imgs, y_dec = _chk(x_dec, "SI_Finder"), _chk(y_imgs, "SI_Finder"), _chk(y_dec, "SI_Finder")
    N, C, H, W = x_dec.shape
    outs = []
    for n in range(N):
        xq = rgb_transform_normalized(x_dec[n:n + 1], 255.0)
        patches = xq.reshape(1, C, H // patch_h, patch_h, W // patch_w, patch_w).permute(0, 2, 4, 1, 3, 5).reshape(-1, C, patch_h, patch_w).contiguous()
        r = rgb_transform_normalized(y_dec[n:n + 1], 255.0)
        corr = L2_or_pearson_corr(patches, r, patch_h, patch_w, mask=mask)
        _, idx = _topk(corr, 1)
        out = torch.empty((1, C, y_imgs.shape[2], y_imgs.shape[3]), device=x_dec.device, dtype=torch.float32)
        yi = y_imgs[n:n + 1].contiguous()
        _lib.check(_L().clc_pm_gather(yi.data_ptr(), C, yi.shape[2], yi.shape[3], patch_h, patch_w, None, idx.data_ptr(), 1, -1.0, out.data_ptr(), _stream()), "clc_pm_gather")
        outs.append(out)
    return torch.cat(outs, dim=0)
