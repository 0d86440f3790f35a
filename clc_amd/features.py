"""Retrieval features on the MI355X: torchvision's ResNet-50 up to the pooling layer, as the reference's dataset classes run it
(SURVEY.md 8(f)-2).

Reference: `models.resnet50(pretrained=True)` with `fc = nn.Identity()`, eval mode, applied to ONE image per `__getitem__` call on
the GPU (/root/reference/dataloader_ref_cluster.py:41-44, 149-180, 241-261); the unused variant adds spatial-pyramid max pooling
(levels 1, 2, 4), PCA to 256 dimensions, a FIFO key-value cache and a second, 90-degree-rotated query
(/root/reference/dataloader_CLC.py:23-40, 110-138, 186-209, 250-294).

Here: the same network (module / parameter / buffer names of torchvision, so `load_state_dict` takes its checkpoints; the pretrained
file itself is not reachable from this build) evaluated for a whole BATCH of images on the kernels of libclc_hip.so:
  * eval-mode BatchNorm folded into the filters and biases once per weight load (a host step, like the CDF tables),
  * the 7x7 / stride-2 stem as patch rows (clc_im2col_small) + a 1x1 convolution on the MFMA kernel, like the RGB heads of g_a,
  * every bottleneck = three launches: 1x1 + ReLU | 3x3 (stride 1 or 2) + ReLU | 1x1 + identity + ReLU in one epilogue (the
    down-sampling 1x1 / stride-s convolution of the first block of a stage is a fourth),
  * max-pool / adaptive pooling kernels (clc_maxpool2d, clc_adaptive_pool2d), PCA projection on the 1x1 MFMA kernel.
Inference only (the reference never trains the extractor).  No CPU fallback.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import lib as _lib
from . import ops
from .ops import ACT_NONE, ACT_RELU, CL, _L, _stream

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # transforms.Normalize of dataloader_ref_cluster.py:35-39


class _ConvBN(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d parameters under torchvision's names (`<conv>.weight`, `<bn>.{weight,bias,running_mean,
    running_var,num_batches_tracked}` are registered by the parent); this helper only folds and launches."""


def _fold(conv_w, bn_w, bn_b, mean, var, eps=1e-5):
    """eval-mode BN(conv(x)) = conv'(x) + b':  w' = w * g / sqrt(var + eps), b' = beta - mean * g / sqrt(var + eps)  (double, then fp32)"""
    scale = bn_w.double() / torch.sqrt(var.double() + eps)
    w = (conv_w.double() * scale.view(-1, 1, 1, 1)).float()
    b = (bn_b.double() - mean.double() * scale).float()
    return w, b


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.stride = stride
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4)) if downsample else None


class ResNet50Features(nn.Module):
    """`resnet50` with `fc = Identity` -> [N, 2048] pooled features (`forward`), or the spatial-pyramid variant (`forward_spp`)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inpl = 64
        for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), start=1):
            layers = [Bottleneck(inpl, planes, stride, downsample=(stride != 1 or inpl != planes * 4))]
            inpl = planes * 4
            layers += [Bottleneck(inpl, planes) for _ in range(1, blocks)]
            setattr(self, f"layer{li}", nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Identity()
        self._folded = None

    # ---- weights: BN folded once per (re)load; kernel layout [Cout][kh][kw][Cin]
    def _load_from_state_dict(self, *a, **k):
        self._folded = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._folded = None
        return super()._apply(fn, *a, **k)

    def fold(self):
        """(re)compute the folded filters; call after changing weights in place."""
        dev = self.conv1.weight.device
        f = {}

        def put(name, conv, bn):
            w, b = _fold(conv.weight.detach(), bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            if w.shape[2] == 1:
                f[name] = (w.reshape(w.shape[0], w.shape[1]).contiguous().to(dev), b.to(dev))
            else:
                f[name] = (w.contiguous(memory_format=CL).to(dev), b.to(dev))

        w, b = _fold(self.conv1.weight.detach(), self.bn1.weight.detach(), self.bn1.bias.detach(), self.bn1.running_mean, self.bn1.running_var, self.bn1.eps)
        # stem as a 1x1 convolution over 7*7*3 = 147 patch values (padded to 148 columns): row order (kh, kw, c) = clc_im2col_small's
        f["stem"] = (torch.nn.functional.pad(w.permute(0, 2, 3, 1).reshape(64, 147), (0, 1)).contiguous().to(dev), b.to(dev))
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                pre = f"layer{li}.{bi}."
                put(pre + "1", blk.conv1, blk.bn1)
                put(pre + "2", blk.conv2, blk.bn2)
                put(pre + "3", blk.conv3, blk.bn3)
                if blk.downsample is not None:
                    put(pre + "d", blk.downsample[0], blk.downsample[1])
        self._folded = f
        return self

    # ---- forward
    @torch.no_grad()
    def trunk(self, x):
        """[N, 3, H, W] normalised images -> layer4 output [N, 2048, H/32, W/32] (pixel-major)"""
        if not x.is_cuda:
            raise _lib.ClcError("clc_amd.features runs on the GPU only (no CPU fallback by design)")
        if self._folded is None:
            self.fold()
        f = self._folded
        x = x.float().contiguous(memory_format=CL)
        col = ops.im2col_small(x, 7, 2, 148)
        t = ops.conv_raw(col, f["stem"][0], f["stem"][1], ks=1, act=ACT_RELU)
        N, Cc, H, W = t.shape
        OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        p = ops.new_act(N, Cc, OH, OW, t)
        _lib.check(_L().clc_maxpool2d(t.data_ptr(), Cc, p.data_ptr(), Cc, N, H, W, Cc, 3, 2, 1, OH, OW, _stream()), "clc_maxpool2d")
        t = p
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                pre = f"layer{li}.{bi}."
                idn = t
                if blk.downsample is not None:
                    idn = ops.conv_raw(t, f[pre + "d"][0], f[pre + "d"][1], ks=1, stride=blk.stride, pad=0)
                u = ops.conv_raw(t, f[pre + "1"][0], f[pre + "1"][1], ks=1, act=ACT_RELU)
                u = ops.conv_raw(u, f[pre + "2"][0], f[pre + "2"][1], ks=3, stride=blk.stride, act=ACT_RELU)
                t = ops.conv_raw(u, f[pre + "3"][0], f[pre + "3"][1], ks=1, act=ACT_RELU, res=idn, res_first=True)   # relu(bn3(conv3) + identity)
        return t

    @staticmethod
    def _pool(t, L, is_max):
        N, Cc, H, W = t.shape
        out = torch.empty((N, Cc * L * L), device=t.device, dtype=torch.float32)
        _lib.check(_L().clc_adaptive_pool2d(t.data_ptr(), Cc, out.data_ptr(), N, H, W, Cc, L, int(is_max), _stream()), "clc_adaptive_pool2d")
        return out

    @torch.no_grad()
    def forward(self, x):
        """== self.feature_extractor(img_tensor) of the reference (dataloader_ref_cluster.py:258), batched: [N, 2048]"""
        return self._pool(self.trunk(x), 1, False)

    @torch.no_grad()
    def forward_spp(self, x, levels=(1, 2, 4)):
        """spatial_pyramid_pooling(layer4(x)) (dataloader_CLC.py:250-256, 282-283): [N, 2048 * (1 + 4 + 16)]"""
        t = self.trunk(x)
        return torch.cat([self._pool(t, l, True) for l in levels], dim=1)


def normalize_images(images_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [N, H, W, 3] (or [N, 3, H, W]) -> ToTensor + Normalize(ImageNet) of the reference's transform (the resize to 224x224 is the
    data loader's business: PIL there)."""
    x = images_u8
    if x.dim() == 4 and x.shape[-1] == 3:
        x = x.permute(0, 3, 1, 2)
    x = x.float() / 255.0
    mean = torch.tensor(IMAGENET_MEAN, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


class PCAProjection:
    """sklearn PCA(n_components).transform on the device: (x - mean) @ components^T (dataloader_CLC.py:124-127, 288-289); fit stays the
    reference's own sklearn estimator on the host (a once-per-dataset step)."""

    def __init__(self, mean, components, device="cuda"):
        self.mean = torch.as_tensor(np.asarray(mean), dtype=torch.float32, device=device)
        self.components = torch.as_tensor(np.asarray(components), dtype=torch.float32, device=device).contiguous()   # [n_components, D]
        if self.components.shape[1] % 4:
            raise ValueError("feature dimension must be a multiple of 4")

    @classmethod
    def from_sklearn(cls, pca, device="cuda"):
        return cls(pca.mean_, pca.components_, device)

    @torch.no_grad()
    def transform(self, feats):
        x = (feats.float() - self.mean[None]).contiguous()
        Q, D = x.shape
        return ops.conv_raw(x.view(Q, D, 1, 1), self.components, None, ks=1, act=ACT_NONE).view(Q, -1)


class KVCache:
    """The reference's FIFO feature cache (dataloader_CLC.py:23-40): same eviction order, O(1) operations."""

    def __init__(self, max_size=1000):
        self.max_size = int(max_size)
        self.cache = OrderedDict()

    def add(self, key, value):
        if key in self.cache:          # (the reference re-appends the key and keeps the old slot: the value is overwritten)
            self.cache[key] = value
            return
        if len(self.cache) >= self.max_size:
            self.cache.popitem(last=False)
        self.cache[key] = value

    def get(self, key):
        return self.cache.get(key)

    def __len__(self):
        return len(self.cache)
