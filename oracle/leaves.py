"""Oracle restatement of the CompressAI (1.2.x) leaves the CLC hot path executes.

TEST INFRASTRUCTURE — not product code (see oracle/__init__.py).

The library itself is a third-party dependency of the reference that is neither
vendored under /root/reference nor installed in this image (un-pinned:
/root/reference/README.md:41,70 only say ``compressai``; the API used at
/root/reference/models/CLC_run.py:1-11,319,528,599-618 fits 1.2.x).  The
definitions below restate its *published* algorithm
and are anchored on the reference's call sites cited per class.  PARITY UNPINNED
for the leaf arithmetic (no reference test or golden vector exists).
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import rans_py

# --------------------------------------------------------------------------- ops


class _LowerBoundFn(torch.autograd.Function):
    """max(x, b) with the CompressAI gradient rule grad*[(x>=b)|(grad<0)]."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        pass_through = (x >= bound) | (g < 0)
        return pass_through.type(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum: float = 0, reparam_offset: float = 2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
        self.lower_bound = LowerBound(bound)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        out = self.lower_bound(x)
        return out ** 2 - self.pedestal


class GDN(nn.Module):
    """y = x * rsqrt(beta + sum_j gamma_ij x_j^2)   (inverse: * sqrt)."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        C = x.size(1)
        beta = self.beta_reparam(self.beta)
        gamma = self.gamma_reparam(self.gamma).reshape(C, C, 1, 1)
        norm = F.conv2d(x ** 2, gamma, beta)
        norm = torch.sqrt(norm) if self.inverse else torch.rsqrt(norm)
        return x * norm


def conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=3, stride=stride, padding=1)


def conv1x1(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=1, stride=stride)


def subpel_conv3x3(i, o, r=1):
    return nn.Sequential(nn.Conv2d(i, o * r ** 2, kernel_size=3, padding=1), nn.PixelShuffle(r))


class ResidualBlockWithStride(nn.Module):
    # used at /root/reference/models/CLC_run.py:274-276,335,338,353,376
    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def forward(self, x):
        identity = x
        out = self.gdn(self.conv2(self.leaky_relu(self.conv1(x))))
        if self.skip is not None:
            identity = self.skip(x)
        return out + identity


class ResidualBlockUpsample(nn.Module):
    # used at /root/reference/models/CLC_run.py:345,348,354,385,394
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def forward(self, x):
        out = self.igdn(self.conv(self.leaky_relu(self.subpel_conv(x))))
        return out + self.upsample(x)


class ResidualBlock(nn.Module):
    # used at /root/reference/models/CLC_run.py:210
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def forward(self, x):
        identity = x
        out = self.leaky_relu(self.conv2(self.leaky_relu(self.conv1(x))))
        if self.skip is not None:
            identity = self.skip(x)
        return out + identity


class AttentionBlock(nn.Module):
    # base class of SWAtten, /root/reference/models/CLC_run.py:222-244
    def __init__(self, N):
        super().__init__()

        class ResidualUnit(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv = nn.Sequential(
                    conv1x1(N, N // 2), nn.ReLU(inplace=True),
                    conv3x3(N // 2, N // 2), nn.ReLU(inplace=True),
                    conv1x1(N // 2, N))
                self.relu = nn.ReLU(inplace=True)

            def forward(self, x):
                return self.relu(self.conv(x) + x)

        self.conv_a = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit())
        self.conv_b = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit(), conv1x1(N, N))

    def forward(self, x):
        return x + self.conv_a(x) * torch.sigmoid(self.conv_b(x))


# --------------------------------------------------------------- CDF quantiser


def pmf_to_quantized_cdf(pmf: Tensor, precision: int = 16) -> Tensor:
    """SURVEY.md A.4 (CompressAI C++ ``pmf_to_quantized_cdf``), float32 rounding."""
    return torch.IntTensor(rans_py.pmf_to_quantized_cdf(pmf.tolist(), precision))


# -------------------------------------------------------------- entropy models


class _Coder:
    """Entropy-coder backend used by compress()/decompress(); default = pure Python."""

    @staticmethod
    def encode_with_indexes(symbols, indexes, cdf, cdf_len, offset) -> bytes:
        enc = rans_py.BufferedRansEncoder()
        enc.encode_with_indexes(symbols, indexes, cdf, cdf_len, offset)
        return enc.flush()

    @staticmethod
    def decode_with_indexes(stream, indexes, cdf, cdf_len, offset):
        dec = rans_py.RansDecoder()
        dec.set_stream(stream)
        return dec.decode_stream(indexes, cdf, cdf_len, offset)


class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound: float = 1e-9, entropy_coder=None, entropy_coder_precision: int = 16):
        super().__init__()
        self.entropy_coder = entropy_coder or _Coder()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())

    offset = property(lambda self: self._offset)
    quantized_cdf = property(lambda self: self._quantized_cdf)
    cdf_length = property(lambda self: self._cdf_length)

    def quantize(self, inputs, mode, means=None):
        if mode == "noise":
            noise = torch.empty_like(inputs).uniform_(-0.5, 0.5)
            return inputs + noise
        outputs = inputs.clone()
        if means is not None:
            outputs -= means
        outputs = torch.round(outputs)
        if mode == "dequantize":
            if means is not None:
                outputs += means
            return outputs
        assert mode == "symbols", mode
        return outputs.int()

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            outputs = inputs.type_as(means)
            outputs += means
        else:
            outputs = inputs.type(dtype)
        return outputs

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    def compress(self, inputs, indexes, means=None):
        symbols = self.quantize(inputs, "symbols", means)
        cdf = self._quantized_cdf.tolist()
        cdf_len = self._cdf_length.reshape(-1).int().tolist()
        off = self._offset.reshape(-1).int().tolist()
        return [
            self.entropy_coder.encode_with_indexes(
                symbols[i].reshape(-1).int().tolist(), indexes[i].reshape(-1).int().tolist(), cdf, cdf_len, off)
            for i in range(symbols.size(0))
        ]

    def decompress(self, strings, indexes, dtype=torch.float, means=None):
        cdf = self._quantized_cdf.tolist()
        cdf_len = self._cdf_length.reshape(-1).int().tolist()
        off = self._offset.reshape(-1).int().tolist()
        outputs = torch.empty(indexes.size(), dtype=torch.float32)
        for i, s in enumerate(strings):
            vals = self.entropy_coder.decode_with_indexes(s, indexes[i].reshape(-1).int().tolist(), cdf, cdf_len, off)
            outputs[i] = torch.tensor(vals, dtype=torch.float32).reshape(outputs[i].size())
        return self.dequantize(outputs, means, dtype)


class EntropyBottleneck(EntropyModel):
    """Factorised density; call sites /root/reference/models/CLC_run.py:483,526-530,643-644,749."""

    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        f = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(torch.full((channels, f[i + 1], f[i]), float(init))))
            self.register_parameter(f"_bias{i:d}", nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i:d}", nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = nn.Parameter(torch.Tensor([-self.init_scale, 0, self.init_scale]).repeat(channels, 1, 1))
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2].detach()

    def _logits_cumulative(self, inputs, stop_gradient):
        logits = inputs
        for i in range(len(self.filters) + 1):
            matrix = getattr(self, f"_matrix{i:d}")
            bias = getattr(self, f"_bias{i:d}")
            if stop_gradient:
                matrix, bias = matrix.detach(), bias.detach()
            logits = torch.matmul(F.softplus(matrix), logits) + bias
            if i < len(self.filters):
                factor = getattr(self, f"_factor{i:d}")
                if stop_gradient:
                    factor = factor.detach()
                logits = logits + torch.tanh(factor) * torch.tanh(logits)
        return logits

    def _likelihood(self, inputs):
        lower = self._logits_cumulative(inputs - 0.5, stop_gradient=False)
        upper = self._logits_cumulative(inputs + 0.5, stop_gradient=False)
        sign = -torch.sign(lower + upper).detach()
        return torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))

    def forward(self, x, training=None):
        if training is None:
            training = self.training
        perm = list(range(x.dim()))
        perm[0], perm[1] = perm[1], perm[0]
        x = x.permute(*perm).contiguous()
        shape = x.size()
        values = x.reshape(x.size(0), 1, -1)
        outputs = self.quantize(values, "noise" if training else "dequantize", self._get_medians())
        likelihood = self._likelihood(outputs)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        outputs = outputs.reshape(shape).permute(*perm).contiguous()
        likelihood = likelihood.reshape(shape).permute(*perm).contiguous()
        return outputs, likelihood

    def loss(self):
        logits = self._logits_cumulative(self.quantiles, stop_gradient=True)
        return torch.abs(logits - self.target).sum()

    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        with torch.no_grad():
            medians = self.quantiles[:, 0, 1]
            minima = torch.clamp(torch.ceil(medians - self.quantiles[:, 0, 0]).int(), min=0)
            maxima = torch.clamp(torch.ceil(self.quantiles[:, 0, 2] - medians).int(), min=0)
            self._offset = -minima
            pmf_start = medians - minima
            pmf_length = maxima + minima + 1
            max_length = int(pmf_length.max().item())
            samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
            lower = self._logits_cumulative(samples - 0.5, stop_gradient=True)
            upper = self._logits_cumulative(samples + 0.5, stop_gradient=True)
            sign = -torch.sign(lower + upper)
            pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
            tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
            self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
            self._cdf_length = pmf_length + 2
        return True

    def _build_indexes(self, size):
        N, C = size[0], size[1]
        view = [1, C] + [1] * (len(size) - 2)
        return torch.arange(C).view(*view).int().repeat(N, 1, *size[2:])

    def _medians_like(self, x_size):
        med = self._get_medians().reshape(1, -1, *([1] * (len(x_size) - 2)))
        return med.expand(x_size[0], *([-1] * (len(x_size) - 1)))

    def compress(self, x):
        return super().compress(x, self._build_indexes(x.size()), self._medians_like(x.size()))

    def decompress(self, strings, size):
        out_size = (len(strings), self._quantized_cdf.size(0), *size)
        med = self._medians_like(out_size)
        return super().decompress(strings, self._build_indexes(out_size), med.dtype, med)


class GaussianConditional(EntropyModel):
    """SURVEY.md A.3; call sites /root/reference/models/CLC_run.py:484,489,569,654-656,689-690,791-795."""

    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        self.tail_mass = float(tail_mass)
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer("scale_table", self._prepare(scale_table) if scale_table else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]))

    @staticmethod
    def _prepare(scale_table):
        return torch.Tensor(tuple(float(s) for s in scale_table))

    @staticmethod
    def _standardized_cumulative(x):
        return 0.5 * torch.erfc(-(2 ** -0.5) * x)

    def update_scale_table(self, scale_table, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        self.scale_table = self._prepare(scale_table)
        self.update()
        return True

    def update(self):
        import scipy.stats

        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(self.scale_table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = self.scale_table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        self._offset = -pmf_center
        self._cdf_length = pmf_length + 2

    def _likelihood(self, inputs, scales, means=None):
        values = inputs - means if means is not None else inputs
        scales = self.lower_bound_scale(scales)
        values = torch.abs(values)
        upper = self._standardized_cumulative((0.5 - values) / scales)
        lower = self._standardized_cumulative((-0.5 - values) / scales)
        return upper - lower

    def forward(self, inputs, scales, means=None, training=None):
        if training is None:
            training = self.training
        outputs = self.quantize(inputs, "noise" if training else "dequantize", means)
        likelihood = self._likelihood(outputs, scales, means)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        return outputs, likelihood

    def build_indexes(self, scales):
        scales = self.lower_bound_scale(scales)
        indexes = scales.new_full(scales.size(), len(self.scale_table) - 1).int()
        for s in self.scale_table[:-1]:
            indexes -= (scales <= s).int()
        return indexes


# ------------------------------------------------------------ CompressionModel

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def _resize_registered_buffers(module, module_name, names, state_dict):
    for n in names:
        key = f"{module_name}.{n}"
        if key in state_dict:
            buf = dict(module.named_buffers()).get(n)
            if buf is not None and buf.numel() == 0:
                buf.resize_(state_dict[key].size())


class CompressionModel(nn.Module):
    """CompressAI 1.2.x base class (legacy ctor kwarg used at /root/reference/models/CLC_run.py:319)."""

    def __init__(self, entropy_bottleneck_channels=None, init_weights=None):
        super().__init__()
        if entropy_bottleneck_channels is not None:
            self.entropy_bottleneck = EntropyBottleneck(entropy_bottleneck_channels)

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = False
        for _, m in self.named_modules():
            if isinstance(m, EntropyBottleneck):
                updated |= m.update(force=force)
            if isinstance(m, GaussianConditional):
                updated |= m.update_scale_table(scale_table, force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        for name, m in self.named_modules():
            if not any(k.startswith(name) for k in state_dict.keys()):
                continue
            if isinstance(m, EntropyBottleneck):
                _resize_registered_buffers(m, name, ["_quantized_cdf", "_offset", "_cdf_length"], state_dict)
            if isinstance(m, GaussianConditional):
                _resize_registered_buffers(m, name, ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
        return nn.Module.load_state_dict(self, state_dict, strict=strict)
