"""Entropy models with the CompressAI surface (EntropyBottleneck, GaussianConditional) on HIP kernels.

Surface kept: ``forward``, ``quantize``, ``dequantize``,
``build_indexes``, ``update``, ``update_scale_table``, ``compress``, ``decompress``, ``loss``,
``_get_medians`` and the ``quantized_cdf / cdf_length / offset`` buffers, as used at
/root/reference/models/CLC_run.py:483-491,526-530,569,643-644,654-656,689-690,749,791-795.

Likelihoods (fwd + bwd incl. both LowerBound gradient rules) run in clc_gauss_lik_* / clc_eb_lik_*;
``quantize("symbols")`` + ``build_indexes`` run in clc_quantize_build_indexes (bit-exact integer
path); CDF tables come from clc_pmf_to_quantized_cdf and the streams from the C++ rANS coder
(clc_amd.ans).  Table construction (``update()``) is a once-per-model host step.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch
import torch.nn as nn

from . import ans, ops
from . import lib as _lib
from .layers import LowerBound


def pmf_to_quantized_cdf(pmf, precision: int = 16):
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(p.size + 1, dtype=np.int32)
    _lib.check(_lib.load().clc_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data), "clc_pmf_to_quantized_cdf")
    return torch.from_numpy(out)


class EntropyModel(nn.Module):
    def __init__(self, likelihood_bound: float = 1e-9, entropy_coder=None, entropy_coder_precision: int = 16):
        super().__init__()
        if abs(likelihood_bound - 1e-9) > 1e-15:
            raise ValueError("the HIP likelihood kernels are built for likelihood_bound = 1e-9 (the reference's value)")
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = True
        self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._host_tables = None  # (cdf, cdf_length, offset) as contiguous int32 numpy arrays, cached for the coder

    offset = property(lambda self: self._offset)
    quantized_cdf = property(lambda self: self._quantized_cdf)
    cdf_length = property(lambda self: self._cdf_length)

    def host_tables(self):
        """int32 numpy views of the CDF tables (one D2H copy per model, not per call as in CLC_run.py:654-656)."""
        # stale when the buffer object was replaced (update()) OR rewritten in place (load_state_dict copies into it: _version moves)
        key = (self._quantized_cdf, self._quantized_cdf._version, self._cdf_length._version, self._offset._version)
        if self._host_tables is None or self._host_tables[3][0] is not key[0] or self._host_tables[3][1:] != key[1:]:
            if self._quantized_cdf.numel() == 0:
                raise ValueError("CDF tables are empty: call update() first")
            cdf = np.ascontiguousarray(self._quantized_cdf.detach().cpu().numpy().astype(np.int32))
            ln = np.ascontiguousarray(self._cdf_length.detach().cpu().numpy().astype(np.int32).reshape(-1))
            off = np.ascontiguousarray(self._offset.detach().cpu().numpy().astype(np.int32).reshape(-1))
            self._host_tables = (cdf, ln, off, key)
        return self._host_tables[:3]

    def _load_from_state_dict(self, *args, **kwargs):
        self._host_tables = None   # the tables are about to be overwritten in place
        return super()._load_from_state_dict(*args, **kwargs)

    def quantize(self, inputs, mode, means=None):
        if mode == "noise":
            return inputs + torch.empty_like(inputs).uniform_(-0.5, 0.5)
        outputs = inputs - means if means is not None else inputs.clone()
        outputs = torch.round(outputs)
        if mode == "dequantize":
            return outputs + means if means is not None else outputs
        if mode != "symbols":
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        return outputs.int()

    @staticmethod
    def dequantize(inputs, means=None, dtype=torch.float):
        if means is not None:
            return inputs.type_as(means) + means
        return inputs.type(dtype)

    def _pmf_to_cdf(self, pmf, tail_mass, pmf_length, max_length):
        pmf, tail_mass = pmf.detach().cpu(), tail_mass.detach().cpu()
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: int(pmf_length[i])], tail_mass[i]), dim=0)
            c = pmf_to_quantized_cdf(prob.numpy(), self.entropy_coder_precision)
            cdf[i, : c.size(0)] = c
        return cdf


class EntropyBottleneck(EntropyModel):
    def __init__(self, channels, *args, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3), **kwargs):
        super().__init__(*args, **kwargs)
        if tuple(filters) != (3, 3, 3, 3):
            raise ValueError("the HIP factorised-density kernel is built for filters=(3,3,3,3)")
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

    def _params(self):
        return ([getattr(self, f"_matrix{k}") for k in range(5)], [getattr(self, f"_bias{k}") for k in range(5)],
                [getattr(self, f"_factor{k}") for k in range(4)])

    def _get_medians(self):
        return self.quantiles[:, :, 1:2].detach()

    def likelihood_and_ste(self, x, training=None):
        """(likelihood, ste_round(x - median) + median) in one kernel — what CLC.forward needs (CLC_run.py:526-530)."""
        training = self.training if training is None else training
        noise = torch.empty_like(x, memory_format=ops.CL).uniform_(-0.5, 0.5) if training else None
        m, b, f = self._params()
        return ops.eb_likelihood(x, noise, self.quantiles.detach(), training, m, b, f)

    def forward(self, x, training=None):
        training = self.training if training is None else training
        noise = torch.empty_like(x, memory_format=ops.CL).uniform_(-0.5, 0.5) if training else None
        m, b, f = self._params()
        lik, z_hat = ops.eb_likelihood(x, noise, self.quantiles.detach(), training, m, b, f)
        outputs = x + noise if training else z_hat.detach()
        return outputs, lik

    def loss(self):
        m, b, f = self._params()
        return ops.eb_aux_loss(self.quantiles, self.target, m, b, f)

    # ---- table construction: host-side, once per model.  Always evaluated on the HOST CPU in fp32 so
    # that encoder and decoder build bit-identical integer tables whatever device the model lives on. ----
    def _logits_cumulative_host(self, inputs):
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(torch.nn.functional.softplus(getattr(self, f"_matrix{i:d}").detach().cpu()), logits)
            logits = logits + getattr(self, f"_bias{i:d}").detach().cpu()
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, f"_factor{i:d}").detach().cpu()) * torch.tanh(logits)
        return logits

    @torch.no_grad()
    def update(self, force=False):
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        q = self.quantiles.detach().cpu()
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
        lower = self._logits_cumulative_host(samples - 0.5)
        upper = self._logits_cumulative_host(samples + 0.5)
        sign = -torch.sign(lower + upper)
        pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        self._offset = (-minima).to(dev)
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        self._host_tables = None
        return True

    # ---- codec ----
    def _symbols_nchw(self, x):
        med = self._get_medians().reshape(1, -1, 1, 1)
        return torch.round(x - med).to(torch.int32)

    def compress(self, x):
        cdf, ln, off = self.host_tables()
        N, Cc, H, W = x.shape
        sym = self._symbols_nchw(x).contiguous().cpu().numpy()  # [N,C,H,W] element order == the reference's reshape(-1)
        idx = np.ascontiguousarray(np.broadcast_to(np.arange(Cc, dtype=np.int32).reshape(Cc, 1, 1), (Cc, H, W)))
        return [ans.encode(sym[i].reshape(-1), idx.reshape(-1), cdf, ln, off) for i in range(N)]

    def decompress(self, strings, size):
        cdf, ln, off = self.host_tables()
        Cc = self._quantized_cdf.size(0)
        H, W = int(size[0]), int(size[1])
        idx = np.ascontiguousarray(np.broadcast_to(np.arange(Cc, dtype=np.int32).reshape(Cc, 1, 1), (Cc, H, W))).reshape(-1)
        out = np.empty((len(strings), Cc, H, W), dtype=np.int32)
        for i, s in enumerate(strings):
            out[i] = ans.decode(s, idx, cdf, ln, off).reshape(Cc, H, W)
        dev = self.quantiles.device
        vals = torch.from_numpy(out).to(dev).float().contiguous(memory_format=ops.CL)
        return vals + self._get_medians().reshape(1, -1, 1, 1)


class GaussianConditional(EntropyModel):
    def __init__(self, scale_table, *args, scale_bound=0.11, tail_mass=1e-9, **kwargs):
        super().__init__(*args, **kwargs)
        if abs(float(scale_bound) - 0.11) > 1e-12:
            raise ValueError("the HIP likelihood kernels are built for scale_bound = 0.11 (the reference's value)")
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
        self.scale_table = self._prepare(scale_table).to(self.scale_table.device)
        self.update()
        return True

    @torch.no_grad()
    def update(self):
        import scipy.stats

        dev = self.scale_table.device
        table = self.scale_table.cpu()
        multiplier = -scipy.stats.norm.ppf(self.tail_mass / 2)
        pmf_center = torch.ceil(table * multiplier).int()
        pmf_length = 2 * pmf_center + 1
        max_length = int(torch.max(pmf_length).item())
        samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
        samples_scale = table.unsqueeze(1).float()
        upper = self._standardized_cumulative((0.5 - samples) / samples_scale)
        lower = self._standardized_cumulative((-0.5 - samples) / samples_scale)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        self._quantized_cdf = self._pmf_to_cdf(pmf, tail_mass, pmf_length, max_length).to(dev)
        self._offset = (-pmf_center).to(dev)
        self._cdf_length = (pmf_length + 2).to(dev)
        self._host_tables = None

    def likelihood_and_ste(self, y, scales, means, training=None, noise=None, lik_out=None):
        """(likelihood, ste_round(y - means) + means) in one kernel (CLC_run.py:569-571).  noise: pre-drawn U(-1/2, 1/2) of y's
        shape (the model draws all slices' noise in one launch); drawn here when None."""
        training = self.training if training is None else training
        if training and noise is None:
            noise = torch.empty_like(y, memory_format=ops.CL).uniform_(-0.5, 0.5)
        if not training:
            noise = None
        return ops.gaussian_likelihood(y, scales, means, noise, training, lik_out)

    def forward(self, inputs, scales, means=None, training=None):
        training = self.training if training is None else training
        if means is None:
            means = torch.zeros_like(inputs, memory_format=ops.CL)
        noise = torch.empty_like(inputs, memory_format=ops.CL).uniform_(-0.5, 0.5) if training else None
        lik, y_hat = ops.gaussian_likelihood(inputs, scales, means, noise, training)
        outputs = inputs + noise if training else y_hat.detach()
        return outputs, lik

    def build_indexes(self, scales):
        _, idx, _ = ops.quantize_build_indexes(scales, torch.zeros_like(scales, memory_format=ops.CL), scales, self.scale_table)
        return idx

    def quantize_and_index(self, y, means, scales):
        """symbols, indexes, y_hat = quantize(y,'symbols',means), build_indexes(scales), symbols+means — one kernel."""
        return ops.quantize_build_indexes(y, means, scales, self.scale_table)
