"""Oracle restatement of the CLC / TCM model graph in plain PyTorch (CPU fp32, NCHW).

TEST INFRASTRUCTURE — not product code (see oracle/__init__.py).

Follows (behaviour, parameter names and registration order, not text):
  WMSA / Block            /root/reference/models/CLC_run.py:108-193  (== models/tcm.py:139-236)
  ConvTransBlock          /root/reference/models/CLC_run.py:195-220
  SWAtten / SwinBlock     /root/reference/models/CLC_run.py:222-266
  ReferenceEncoder        /root/reference/models/CLC_run.py:269-281
  in-file CLM (dormant)   /root/reference/models/CLC_run.py:284-313
  CLC                     /root/reference/models/CLC_run.py:316-814
  TCM                     /root/reference/models/tcm.py:310-626
Wiring is pinned against the reference's own classes in tools/make_golden.py
(fixtures: tests/golden/graph_*.npz).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .leaves import (AttentionBlock, CompressionModel, EntropyBottleneck, GaussianConditional, ResidualBlock,
                     ResidualBlockUpsample, ResidualBlockWithStride, conv1x1, conv3x3, get_scale_table,
                     subpel_conv3x3, _resize_registered_buffers)
from . import rans_py


def ste_round(x):
    return torch.round(x) - x.detach() + x


def _conv(i, o, kernel_size=5, stride=2):
    return nn.Conv2d(i, o, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


class WMSA(nn.Module):
    """Window / shifted-window multi-head self attention on [B,H,W,C] tokens."""

    def __init__(self, input_dim, output_dim, head_dim, window_size, type):
        super().__init__()
        self.input_dim, self.output_dim, self.head_dim = input_dim, output_dim, head_dim
        self.n_heads = input_dim // head_dim
        self.window_size, self.type = window_size, type
        self.scale = head_dim ** -0.5
        self.embedding_layer = nn.Linear(input_dim, 3 * input_dim, bias=True)
        self.relative_position_params = nn.Parameter(
            torch.nn.init.trunc_normal_(torch.zeros(self.n_heads, 2 * window_size - 1, 2 * window_size - 1), std=0.02))
        self.linear = nn.Linear(input_dim, output_dim)

    def rel_bias(self):
        ws = self.window_size
        idx = torch.arange(ws)
        cord = torch.stack(torch.meshgrid(idx, idx, indexing="ij"), -1).reshape(-1, 2)  # (i, j) row-major
        rel = cord[:, None, :] - cord[None, :, :] + ws - 1
        return self.relative_position_params[:, rel[..., 0], rel[..., 1]]  # [heads, ws², ws²]

    def shift_mask(self, hw, ww):
        ws, s = self.window_size, self.window_size - self.window_size // 2
        m = torch.zeros(hw, ww, ws, ws, ws, ws, dtype=torch.bool)
        if self.type != "W":
            m[-1, :, :s, :, s:, :] = True
            m[-1, :, s:, :, :s, :] = True
            m[:, -1, :, :s, :, s:] = True
            m[:, -1, :, s:, :, :s] = True
        return m.reshape(hw * ww, ws * ws, ws * ws)

    def forward(self, x):
        B, H, W, C = x.shape
        ws, nh, hd = self.window_size, self.n_heads, self.head_dim
        sh = ws // 2
        if self.type != "W":
            x = torch.roll(x, shifts=(-sh, -sh), dims=(1, 2))
        hw, ww = H // ws, W // ws
        xw = x.reshape(B, hw, ws, ww, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, hw * ww, ws * ws, C)
        qkv = self.embedding_layer(xw)  # [B, nw, np, 3C]; channel = three*C + head*hd + c
        qkv = qkv.reshape(B, hw * ww, ws * ws, 3 * nh, hd).permute(3, 0, 1, 2, 4)  # [3nh, B, nw, np, hd]
        q, k, v = qkv[:nh], qkv[nh:2 * nh], qkv[2 * nh:]
        sim = torch.matmul(q, k.transpose(-1, -2)) * self.scale + self.rel_bias()[:, None, None]
        if self.type != "W":
            sim = sim.masked_fill(self.shift_mask(hw, ww)[None, None], float("-inf"))
        out = torch.matmul(torch.softmax(sim, dim=-1), v)  # [nh, B, nw, np, hd]
        out = out.permute(1, 2, 3, 0, 4).reshape(B, hw * ww, ws * ws, C)
        out = self.linear(out)
        out = out.reshape(B, hw, ww, ws, ws, self.output_dim).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, self.output_dim)
        if self.type != "W":
            out = torch.roll(out, shifts=(sh, sh), dims=(1, 2))
        return out


class Block(nn.Module):
    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path, type="W", input_resolution=None):
        super().__init__()
        assert type in ("W", "SW")
        self.type = type
        self.ln1 = nn.LayerNorm(input_dim)
        self.msa = WMSA(input_dim, input_dim, head_dim, window_size, type)
        self.drop_path = nn.Identity()  # drop_path_rate is 0 on every path
        self.ln2 = nn.LayerNorm(input_dim)
        self.mlp = nn.Sequential(nn.Linear(input_dim, 4 * input_dim), nn.GELU(), nn.Linear(4 * input_dim, output_dim))

    def forward(self, x):
        x = x + self.msa(self.ln1(x))
        return x + self.mlp(self.ln2(x))


class ConvTransBlock(nn.Module):
    def __init__(self, conv_dim, trans_dim, head_dim, window_size, drop_path, type="W"):
        super().__init__()
        self.conv_dim, self.trans_dim = conv_dim, trans_dim
        self.trans_block = Block(trans_dim, trans_dim, head_dim, window_size, drop_path, type)
        self.conv1_1 = nn.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv1_2 = nn.Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1, 1, 0, bias=True)
        self.conv_block = ResidualBlock(conv_dim, conv_dim)

    def forward(self, x):
        c, t = torch.split(self.conv1_1(x), (self.conv_dim, self.trans_dim), dim=1)
        c = self.conv_block(c) + c
        t = self.trans_block(t.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return x + self.conv1_2(torch.cat((c, t), dim=1))


class SwinBlock(nn.Module):
    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path):
        super().__init__()
        self.block_1 = Block(input_dim, output_dim, head_dim, window_size, drop_path, type="W")
        self.block_2 = Block(input_dim, output_dim, head_dim, window_size, drop_path, type="SW")
        self.window_size = window_size

    def forward(self, x):
        if x.size(-1) <= self.window_size or x.size(-2) <= self.window_size:
            # the reference's small-map path pads to ws+1 and then fails in the window partition
            # (/root/reference/models/CLC_run.py:255-266); it is never reached for inputs >= 256.
            raise ValueError("SwinBlock: feature map must be larger than the window (input must be >= 256x256)")
        t = self.block_2(self.block_1(x.permute(0, 2, 3, 1)))
        return t.permute(0, 3, 1, 2)


class SWAtten(AttentionBlock):
    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path, inter_dim=192):
        if inter_dim is not None:
            super().__init__(N=inter_dim)
            self.non_local_block = SwinBlock(inter_dim, inter_dim, head_dim, window_size, drop_path)
            self.in_conv = conv1x1(input_dim, inter_dim)
            self.out_conv = conv1x1(inter_dim, output_dim)
        else:
            super().__init__(N=input_dim)
            self.non_local_block = SwinBlock(input_dim, input_dim, head_dim, window_size, drop_path)

    def forward(self, x):
        x = self.in_conv(x)
        z = self.non_local_block(x)
        out = self.conv_a(x) * torch.sigmoid(self.conv_b(z)) + x
        return self.out_conv(out)


class ReferenceEncoder(nn.Module):
    def __init__(self, N=128, M=320):
        super().__init__()
        self.encoder = nn.Sequential(
            ResidualBlockWithStride(3, N, stride=2), ResidualBlockWithStride(N, N, stride=2),
            ResidualBlockWithStride(N, M, stride=2), conv3x3(M, M, stride=2))

    def forward(self, x):
        return self.encoder(x)


class CLM(nn.Module):
    """In-file CLM: constructed by CLC (parameters live in state_dict) but never called."""

    def __init__(self, channels, head_dim=8, window_size=8):
        super().__init__()
        self.alignment = SWAtten(channels * 2, channels, head_dim, window_size, 0, inter_dim=channels)
        self.fusion = nn.Sequential(conv1x1(channels * 2, channels), nn.GELU(), conv1x1(channels, channels))

    def forward(self, x, ref_feat):
        aligned = self.alignment(torch.cat([x, ref_feat], dim=1))
        return self.fusion(torch.cat([x, aligned], dim=1))


def _cc(cin, cout):
    return nn.Sequential(_conv(cin, 224, 3, 1), nn.GELU(), _conv(224, 128, 3, 1), nn.GELU(), _conv(128, cout, 3, 1))


def _stage(N, head_dim, ws, tail):
    return [ConvTransBlock(N, N, head_dim, ws, 0.0, "W" if i % 2 == 0 else "SW") for i in range(2)] + [tail]


class _SliceCodecMixin:
    """Shared channel-slice loop of forward / compress / decompress."""

    def _params_for_slice(self, i, latent_means, latent_scales, y_hat_slices, ref_features, y_shape):
        support = y_hat_slices if self.max_support_slices < 0 else y_hat_slices[: self.max_support_slices]
        mean_support = self.atten_mean[i](torch.cat([latent_means] + support, dim=1))
        scale_support = self.atten_scale[i](torch.cat([latent_scales] + support, dim=1))
        if ref_features is not None:
            mu = self.ref_cc_mean_transforms[i](torch.cat([mean_support, ref_features], dim=1))
            scale = self.ref_cc_scale_transforms[i](torch.cat([scale_support, ref_features], dim=1))
        else:
            mu = self.cc_mean_transforms[i](mean_support)
            scale = self.cc_scale_transforms[i](scale_support)
        mu = mu[:, :, : y_shape[0], : y_shape[1]]
        scale = scale[:, :, : y_shape[0], : y_shape[1]]
        return mean_support, mu, scale

    def _lrp(self, i, mean_support, y_hat_slice, ref_features):
        sup = torch.cat([mean_support, y_hat_slice], dim=1)
        if ref_features is not None:
            lrp = self.ref_lrp_transforms[i](torch.cat([sup, ref_features], dim=1))
        else:
            lrp = self.lrp_transforms[i](sup)
        return 0.5 * torch.tanh(lrp)

    def _hyper(self, z_hat):
        return self.h_scale_s(z_hat), self.h_mean_s(z_hat)

    def _ref(self, ref_frames):
        return None

    def _fuse_z(self, z):
        return z

    def forward(self, x, ref_frames=None):
        ref_features = self._ref(ref_frames)
        y = self.g_a(x)
        y_shape = y.shape[2:]
        z = self._fuse_z(self.h_a(y))
        _, z_likelihoods = self.entropy_bottleneck(z)
        z_offset = self.entropy_bottleneck._get_medians()
        z_hat = ste_round(z - z_offset) + z_offset
        latent_scales, latent_means = self._hyper(z_hat)
        y_hat_slices, y_lik, mus, scales = [], [], [], []
        for i, y_slice in enumerate(y.chunk(self.num_slices, 1)):
            mean_support, mu, scale = self._params_for_slice(i, latent_means, latent_scales, y_hat_slices, ref_features, y_shape)
            mus.append(mu)
            scales.append(scale)
            _, lik = self.gaussian_conditional(y_slice, scale, mu)
            y_lik.append(lik)
            y_hat_slice = ste_round(y_slice - mu) + mu
            y_hat_slice = y_hat_slice + self._lrp(i, mean_support, y_hat_slice, ref_features)
            y_hat_slices.append(y_hat_slice)
        x_hat = self.g_s(torch.cat(y_hat_slices, dim=1))
        return {"x_hat": x_hat,
                "likelihoods": {"y": torch.cat(y_lik, dim=1), "z": z_likelihoods},
                "para": {"means": torch.cat(mus, dim=1), "scales": torch.cat(scales, dim=1), "y": y}}

    def compress(self, x, ref_frames=None):
        ref_features = self._ref(ref_frames)
        y = self.g_a(x)
        y_shape = y.shape[2:]
        z = self._fuse_z(self.h_a(y))
        z_strings = self.entropy_bottleneck.compress(z)
        z_hat = self.entropy_bottleneck.decompress(z_strings, z.size()[-2:])
        latent_scales, latent_means = self._hyper(z_hat)
        gc = self.gaussian_conditional
        cdf, cdf_len, off = gc.quantized_cdf.tolist(), gc.cdf_length.reshape(-1).int().tolist(), gc.offset.reshape(-1).int().tolist()
        symbols, indexes, y_hat_slices = [], [], []
        for i, y_slice in enumerate(y.chunk(self.num_slices, 1)):
            mean_support, mu, scale = self._params_for_slice(i, latent_means, latent_scales, y_hat_slices, ref_features, y_shape)
            index = gc.build_indexes(scale)
            y_q = gc.quantize(y_slice, "symbols", mu)
            y_hat_slice = y_q + mu
            symbols.extend(y_q.reshape(-1).tolist())
            indexes.extend(index.reshape(-1).tolist())
            y_hat_slice = y_hat_slice + self._lrp(i, mean_support, y_hat_slice, ref_features)
            y_hat_slices.append(y_hat_slice)
        enc = rans_py.BufferedRansEncoder()
        enc.encode_with_indexes(symbols, indexes, cdf, cdf_len, off)
        return {"strings": [[enc.flush()], z_strings], "shape": z.size()[-2:]}

    def decompress(self, strings, shape, ref_frames=None):
        ref_features = self._ref(ref_frames)
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        latent_scales, latent_means = self._hyper(z_hat)
        y_shape = [z_hat.shape[2] * 4, z_hat.shape[3] * 4]
        gc = self.gaussian_conditional
        cdf, cdf_len, off = gc.quantized_cdf.tolist(), gc.cdf_length.reshape(-1).int().tolist(), gc.offset.reshape(-1).int().tolist()
        dec = rans_py.RansDecoder()
        dec.set_stream(strings[0][0])
        y_hat_slices = []
        for i in range(self.num_slices):
            mean_support, mu, scale = self._params_for_slice(i, latent_means, latent_scales, y_hat_slices, ref_features, y_shape)
            index = gc.build_indexes(scale)
            rv = dec.decode_stream(index.reshape(-1).tolist(), cdf, cdf_len, off)
            rv = torch.Tensor(rv).reshape(1, -1, y_shape[0], y_shape[1])
            y_hat_slice = gc.dequantize(rv, mu)
            y_hat_slice = y_hat_slice + self._lrp(i, mean_support, y_hat_slice, ref_features)
            y_hat_slices.append(y_hat_slice)
        x_hat = self.g_s(torch.cat(y_hat_slices, dim=1)).clamp_(0, 1)
        return {"x_hat": x_hat}


def _build_backbone(self, N, M, head_dim):
    """Registers g_a, g_s (call order matters for parameter registration order)."""
    ws = self.window_size
    self.g_a = nn.Sequential(
        ResidualBlockWithStride(3, 2 * N, 2),
        *_stage(N, head_dim[0], ws, ResidualBlockWithStride(2 * N, 2 * N, stride=2)),
        *_stage(N, head_dim[1], ws, ResidualBlockWithStride(2 * N, 2 * N, stride=2)),
        *_stage(N, head_dim[2], ws, conv3x3(2 * N, M, stride=2)))
    self.g_s = nn.Sequential(
        ResidualBlockUpsample(M, 2 * N, 2),
        *_stage(N, head_dim[3], ws, ResidualBlockUpsample(2 * N, 2 * N, 2)),
        *_stage(N, head_dim[4], ws, ResidualBlockUpsample(2 * N, 2 * N, 2)),
        *_stage(N, head_dim[5], ws, subpel_conv3x3(2 * N, 3, 2)))


def _build_hyper(self, N):
    self.h_a = nn.Sequential(ResidualBlockWithStride(320, 2 * N, 2), *_stage(N, 32, 4, conv3x3(2 * N, 192, stride=2)))
    self.h_mean_s = nn.Sequential(ResidualBlockUpsample(192, 2 * N, 2), *_stage(N, 32, 4, subpel_conv3x3(2 * N, 320, 2)))
    self.h_scale_s = nn.Sequential(ResidualBlockUpsample(192, 2 * N, 2), *_stage(N, 32, 4, subpel_conv3x3(2 * N, 320, 2)))


def _slice_w(self, i, extra=0):
    return 320 + (320 // self.num_slices) * min(i + extra, 5 + extra)


class CLC(_SliceCodecMixin, CompressionModel):
    def __init__(self, config=[2, 2, 2, 2, 2, 2], head_dim=[8, 16, 32, 32, 16, 8], drop_path_rate=0, N=128, M=320,
                 num_slices=5, max_support_slices=5, num_ref_frames=3, use_ref=True, **kwargs):
        super().__init__(entropy_bottleneck_channels=N)
        assert list(config) == [2] * 6 and drop_path_rate == 0
        self.config, self.head_dim, self.window_size = config, head_dim, 8
        self.num_slices, self.max_support_slices = num_slices, max_support_slices
        self.num_ref_frames, self.use_ref, self.M = num_ref_frames, use_ref, M
        S = 320 // num_slices
        _build_backbone(self, N, M, head_dim)
        self.ref_encoder = ReferenceEncoder(N, M)
        self.feature_alignment = nn.ModuleList([CLM(192, head_dim=32, window_size=4) for _ in range(num_ref_frames)])
        self.multi_ref_fusion = nn.Sequential(conv1x1(192 * (num_ref_frames + 1), 256), nn.GELU(), conv1x1(256, 192))
        _build_hyper(self, N)
        sw = lambda i: nn.Sequential(SWAtten(_slice_w(self, i), _slice_w(self, i), 16, self.window_size, 0, inter_dim=128))
        self.atten_mean = nn.ModuleList(sw(i) for i in range(num_slices))
        self.atten_scale = nn.ModuleList(sw(i) for i in range(num_slices))
        self.ref_cc_mean_transforms = nn.ModuleList(_cc(_slice_w(self, i) + 64, S) for i in range(num_slices))
        self.ref_cc_scale_transforms = nn.ModuleList(_cc(_slice_w(self, i) + 64, S) for i in range(num_slices))
        self.cc_mean_transforms = nn.ModuleList(_cc(_slice_w(self, i), S) for i in range(num_slices))
        self.cc_scale_transforms = nn.ModuleList(_cc(_slice_w(self, i), S) for i in range(num_slices))
        self.lrp_transforms = nn.ModuleList(_cc(_slice_w(self, i, 1), S) for i in range(num_slices))
        self.ref_lrp_transforms = nn.ModuleList(_cc(_slice_w(self, i, 1) + 64, S) for i in range(num_slices))
        self.ref_feature_adapter = nn.Sequential(conv1x1(M * num_ref_frames, 128), nn.GELU(), conv1x1(128, 64))
        self.entropy_bottleneck = EntropyBottleneck(192)
        self.gaussian_conditional = GaussianConditional(None)

    def _ref(self, ref_frames):
        self._ref_latents = None
        if ref_frames is None or not self.use_ref:
            return None
        feats = [self.ref_encoder(r) for r in ref_frames]
        self._ref_latents = feats
        return self.ref_feature_adapter(torch.cat(feats, dim=1))

    # SURVEY §8(f)-4 — NOT reference behaviour (the reference constructs feature_alignment / multi_ref_fusion, CLC_run.py:359-369,
    # and never calls them).  wire_clm = True applies them the one way their constructor shapes admit: CLM(192) pairs the hyper-latent
    # z with a 192-channel feature of each reference at z's resolution — h_a of that reference's latent — and multi_ref_fusion
    # (192 * (R + 1) -> 256 -> 192) fuses z with the R aligned features.  Encoder side only: the decoder receives the fused z_hat.
    wire_clm = False

    def _fuse_z(self, z):
        if not self.wire_clm or getattr(self, "_ref_latents", None) is None:
            return z
        aligned = [self.feature_alignment[r](z, self.h_a(f)) for r, f in enumerate(self._ref_latents)]
        return self.multi_ref_fusion(torch.cat([z] + aligned, dim=1))

    extract_ref_features = _ref

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=False):
        own = self.state_dict()
        filtered = {k: v for k, v in state_dict.items() if k in own}
        _resize_registered_buffers(self.gaussian_conditional, "gaussian_conditional",
                                   ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
        return super().load_state_dict(filtered, strict=False)


class TCM(_SliceCodecMixin, CompressionModel):
    def __init__(self, config=[2, 2, 2, 2, 2, 2], head_dim=[8, 16, 32, 32, 16, 8], drop_path_rate=0, N=128, M=320,
                 num_slices=5, max_support_slices=5, **kwargs):
        super().__init__(entropy_bottleneck_channels=N)
        assert list(config) == [2] * 6 and drop_path_rate == 0
        self.config, self.head_dim, self.window_size = config, head_dim, 8
        self.num_slices, self.max_support_slices, self.M = num_slices, max_support_slices, M
        S = 320 // num_slices
        _build_backbone(self, N, M, head_dim)
        _build_hyper(self, N)
        sw = lambda i: nn.Sequential(SWAtten(_slice_w(self, i), _slice_w(self, i), 16, self.window_size, 0, inter_dim=128))
        self.atten_mean = nn.ModuleList(sw(i) for i in range(num_slices))
        self.atten_scale = nn.ModuleList(sw(i) for i in range(num_slices))
        self.cc_mean_transforms = nn.ModuleList(_cc(_slice_w(self, i), S) for i in range(num_slices))
        self.cc_scale_transforms = nn.ModuleList(_cc(_slice_w(self, i), S) for i in range(num_slices))
        self.lrp_transforms = nn.ModuleList(_cc(_slice_w(self, i, 1), S) for i in range(num_slices))
        self.entropy_bottleneck = EntropyBottleneck(192)
        self.gaussian_conditional = GaussianConditional(None)

    def forward(self, x, ref_frames=None):
        return super().forward(x, None)

    def compress(self, x, ref_frames=None):
        return super().compress(x, None)

    def decompress(self, strings, shape, ref_frames=None):
        return super().decompress(strings, shape, None)

    def update(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= self.entropy_bottleneck.update(force=force)
        return updated

    def load_state_dict(self, state_dict, strict=True):
        _resize_registered_buffers(self.gaussian_conditional, "gaussian_conditional",
                                   ["_quantized_cdf", "_offset", "_cdf_length", "scale_table"], state_dict)
        return super().load_state_dict(state_dict, strict=strict)
