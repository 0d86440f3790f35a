"""Deterministic by-name weight recipe + synthetic inputs (shared by bench.py, tests and the checker; no arithmetic of the path).

Golden fixtures cannot carry a 280 MB state_dict, so every floating-point entry of a
model's state_dict is regenerated from a CPU generator seeded by crc32(name): the
reference classes (through tools/ref_shim.py), the oracle and the product model all
receive bit-identical weights from the same call.  The scales are chosen so that the
latent y, the predicted scales and the likelihoods are non-degenerate (not all clamped).
"""
from __future__ import annotations

import re
import zlib

import torch

CONV_GAIN = 0.6
_SKIP_SUFFIX = ("pedestal", ".bound", "target", "_offset", "_quantized_cdf", "_cdf_length", "scale_table", "scale_bound")


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


@torch.no_grad()
def apply_weight_recipe(model: torch.nn.Module, seed: int = 0) -> None:
    sd = model.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if name.endswith(_SKIP_SUFFIX) or not t.is_floating_point() or t.numel() == 0:
            continue
        g = _gen(name, seed)
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "beta":  # GDN beta (stored re-parametrised: sqrt(v + pedestal))
            v = torch.sqrt(1.0 + 0.2 * torch.rand(shape, generator=g) + 2.0 ** -36)
        elif leaf == "gamma":
            C = shape[0]
            v = torch.sqrt(0.1 * torch.eye(C) + (0.2 / C) * torch.rand(shape, generator=g) + 2.0 ** -36)
        elif leaf == "relative_position_params":
            v = 0.5 * torch.randn(shape, generator=g)
        elif leaf == "quantiles":
            med = 0.3 * torch.randn(shape[0], 1, 1, generator=g)
            w = 3.0 + 4.0 * torch.rand(shape[0], 1, 2, generator=g)
            v = torch.cat([med - w[..., :1], med, med + w[..., 1:]], dim=-1)
        elif leaf.startswith("_matrix"):
            v = t.detach().cpu().clone() + 0.2 * torch.randn(shape, generator=g)
        elif leaf.startswith("_bias"):
            v = torch.rand(shape, generator=g) - 0.5
        elif leaf.startswith("_factor"):
            v = 0.2 * torch.randn(shape, generator=g)
        elif leaf == "weight" and t.dim() >= 2:
            fan_in = t[0].numel()
            gain = CONV_GAIN
            if re.search(r"cc_(mean|scale)_transforms\.\d+\.4\.weight$", name):
                gain *= 12.0  # keep predicted means / scales O(1) so likelihoods are informative
            elif name == "g_a.9.weight":
                gain *= 5.0  # latent y with a few quantisation bins of spread
            v = torch.randn(shape, generator=g) * gain * (1.0 / fan_in) ** 0.5
        elif leaf == "weight":  # LayerNorm gain
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == "bias":
            v = 0.05 * torch.randn(shape, generator=g)
            if re.search(r"cc_scale_transforms\.\d+\.4\.bias$", name):
                v = v + 0.6
        else:
            raise KeyError(f"weight recipe has no rule for {name} {shape}")
        t.copy_(v.to(t.dtype))


def synthetic_image(batch: int, h: int, w: int, seed: int = 100, smooth: bool = False) -> torch.Tensor:
    """uint8 noise / 255 (mirrors normalize_to_tensor, /root/reference/dataloader_ref_cluster.py:182-194);
    ``smooth`` low-pass filters it so codec tests see compressible content."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.randint(0, 256, (batch, 3, h, w), generator=g, dtype=torch.uint8).float() / 255.0
    if smooth:
        k = 15
        x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (k // 2,) * 4, mode="reflect"), k, stride=1)
        x = (x - x.amin(dim=(2, 3), keepdim=True)) / (x.amax(dim=(2, 3), keepdim=True) - x.amin(dim=(2, 3), keepdim=True))
        x = torch.round(x * 255.0) / 255.0
    return x
