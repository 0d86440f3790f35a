"""MI355X-native layer library with the CompressAI / reference module surface.

Same class names, constructor arguments, parameter / buffer names and shapes as
``compressai.layers`` and the in-tree blocks of
/root/reference/models/CLC_run.py:108-313, so ``state_dict`` round-trips with reference
checkpoints — but every forward runs on the HIP kernels of libclc_hip.so through
``clc_amd.ops`` with the blocks' elementwise tails fused into conv epilogues:

  ResidualBlockWithStride   conv3x3/s2+LeakyReLU | conv3x3 | GDN(+skip) — GDN = 1x1 conv with square-on-load
                            and x*rsqrt(.) epilogue, the skip branch added in the same epilogue
  ResidualBlockUpsample     subpel conv (PixelShuffle fused in the store)+LeakyReLU | conv3x3 | IGDN(+upsample)
  ResidualBlock             second conv carries LeakyReLU + k*identity in its epilogue
  ResidualUnit              relu(conv(x) + x) in the last 1x1's epilogue
  Block                     LN | qkv GEMM | window attention | proj(+residual) | LN | fc1+GELU | fc2(+residual)

Activations stay logical NCHW in channels_last memory (= NHWC), so no permute ever runs.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, CL


class _LowerBoundFn(torch.autograd.Function):
    """max(x, bound); gradient passes where x >= bound or the gradient pushes x up (CompressAI LowerBound)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).type(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class NonNegativeParametrizer(nn.Module):
    def __init__(self, minimum: float = 0, reparam_offset: float = 2 ** -18):
        super().__init__()
        pedestal = float(reparam_offset) ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self.lower_bound = LowerBound((float(minimum) + pedestal) ** 0.5)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return self.lower_bound(x) ** 2 - self.pedestal


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (weight kept channels_last = [Cout][kh][kw][Cin]); forward on the implicit-GEMM kernel."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, padding=None, bias=True):
        padding = kernel_size // 2 if padding is None else padding
        if kernel_size not in (1, 3) or padding != kernel_size // 2 or stride not in (1, 2):
            raise ValueError("clc_amd.layers.Conv2d supports 1x1 / 3x3 'same' convolutions with stride 1 or 2")
        super().__init__(in_ch, out_ch, kernel_size, stride=stride, padding=padding, bias=bias)
        self.weight.data = self.weight.data.contiguous(memory_format=CL)
        self.weight._clc_is_filter = True

    def forward(self, x, act=ACT_NONE, res=None, res_scale=1.0, res_first=False, shuffle=False, pair=None, fold_in=None, fold_out=None, out=None,
                grad_slot=None, park_dx=None, gate_in=None, gate_out=None):
        """pair: a second Conv2d of the same shape applied to the second half of the batch in the same launch — or a tuple of three:
        this layer and they take one quarter of the batch each.  fold_in / fold_out: ops.GradFold of a residual block (see there)."""
        wx = None
        if isinstance(pair, (tuple, list)):
            pair, q3, q4 = pair
            wx = ((q3.weight, q3.bias), (q4.weight, q4.bias))
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride[0], act=act, res=res, res_scale=res_scale,
                          res_first=res_first, shuffle=shuffle, w2=pair.weight if pair is not None else None,
                          b2=pair.bias if pair is not None else None, fold_in=fold_in, fold_out=fold_out, out=out, grad_slot=grad_slot,
                          park_dx=park_dx, gate_in=gate_in, gate_out=gate_out, wx=wx)


class Linear(nn.Linear):
    """nn.Linear over the channel dim of a pixel-major [N,C,H,W] tensor (tokens are pixels)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.weight._clc_is_filter = True

    def forward(self, x, act=ACT_NONE, res=None, pair=None, fold_in=None, fold_out=None, out=None, gate_in=None, gate_out=None):
        return ops.linear(x, self.weight, self.bias, act=act, res=res, w2=pair.weight if pair is not None else None,
                          b2=pair.bias if pair is not None else None, fold_in=fold_in, fold_out=fold_out, out=out, gate_in=gate_in, gate_out=gate_out)


def _halves(x):
    return ops.split_batch(x)   # views; the backward is one concatenation


class LayerNorm(nn.LayerNorm):
    def forward(self, x, pair=None, fold_in=None, grad_slot=None):
        if pair is None:
            return ops.layernorm(x, self.weight, self.bias, fold_in, grad_slot)
        return ops.layernorm(x, self.weight, self.bias, fold_in, grad_slot, pair.weight, pair.bias)   # one launch, per-half parameters


class GELU(nn.Module):
    """Placeholder keeping the reference's Sequential indices (``mlp.1`` / ``*.1``, ``*.3``); fused into the producer."""

    def forward(self, x):  # pragma: no cover - never called, the producer conv applies it
        raise RuntimeError("GELU is fused into the preceding layer's epilogue")


class PixelShuffle2(nn.Module):
    """Placeholder for nn.PixelShuffle(2) (fused into the producing conv's store)."""

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("PixelShuffle is fused into the preceding conv's store")


def conv3x3(i, o, stride=1):
    return Conv2d(i, o, 3, stride=stride)


def conv1x1(i, o, stride=1):
    return Conv2d(i, o, 1, stride=stride)


class SubpelConv3x3(nn.Sequential):
    """Sequential(Conv2d(i, o*r^2, 3), PixelShuffle(r)) — parameter names ``0.weight`` / ``0.bias``."""

    def __init__(self, i, o, r=2):
        assert r == 2
        super().__init__(Conv2d(i, o * r * r, 3), PixelShuffle2())

    def forward(self, x, act=ACT_NONE, res=None, pair=None, fold_in=None, park_dx=None, gate_out=None):
        return self[0](x, act=act, res=res, shuffle=True, pair=pair[0] if pair is not None else None, fold_in=fold_in, park_dx=park_dx, gate_out=gate_out)


def subpel_conv3x3(i, o, r=1):
    return SubpelConv3x3(i, o, r)


class GDN(nn.Module):
    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def _consts(self):
        """(gamma bound, beta bound, pedestal) as the float32 values of the CompressAI buffers (read once: no sync per step)."""
        c = getattr(self, "_clc_consts", None)
        if c is None:
            c = (float(self.gamma_reparam.lower_bound.bound.item()), float(self.beta_reparam.lower_bound.bound.item()),
                 float(self.gamma_reparam.pedestal.item()))
            assert float(self.beta_reparam.pedestal.item()) == c[2]
            self._clc_consts = c
        return c

    def forward(self, x, res=None, pair=None):
        """pair: a second GDN of the same shape for the second half of the batch (same launches, per-half parameters)."""
        gb, bb, ped = self._consts()
        if pair is not None:
            assert pair._consts() == (gb, bb, ped) and pair.inverse == self.inverse
            return ops.gdn_param(x, self.gamma, self.beta, gb, bb, ped, inverse=self.inverse, res=res, gamma2=pair.gamma, beta2=pair.beta)
        return ops.gdn_param(x, self.gamma, self.beta, gb, bb, ped, inverse=self.inverse, res=res)


class ResidualBlockWithStride(nn.Module):
    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.gdn = GDN(out_ch)
        self.skip = conv1x1(in_ch, out_ch, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def forward(self, x):
        cin = self.conv1.in_channels
        if cin <= 3 and self.skip is not None and self.conv1.stride[0] == 2 and not x.requires_grad:
            # RGB head: K = 27 is far too small for the implicit-GEMM loader's 16-B channel groups.  One patch-row pass
            # (27 -> 32 zero-padded columns), then conv1 AND the 1x1/s2 skip conv (its input pixel is the 3x3 window's centre
            # tap, columns 4*cin .. 5*cin) are 1x1 convolutions over those rows on the MFMA kernel.
            col = ops.im2col_small(x, 3, 2, 32)
            co = self.conv1.out_channels
            w1, ws = ops.stem_filters(self.conv1.weight, self.skip.weight)
            g = ops.ActGate()   # conv2's data gradient arrives already multiplied by LeakyReLU'
            t = self.conv2(ops.linear(col, w1, self.conv1.bias, act=ACT_LRELU, gate_out=g), gate_in=g)
            return self.gdn(t, res=ops.linear(col, ws, self.skip.bias))
        if self.skip is not None and x.requires_grad:
            # x feeds conv1 and the skip conv: the skip's input gradient (computed first in backward) is parked and added in
            # conv1's data-gradient epilogue instead of by an autograd add kernel
            f, g = ops.GradFold(), ops.ActGate()
            t = self.conv2(self.conv1(x, act=ACT_LRELU, fold_in=f, gate_out=g), gate_in=g)
            return self.gdn(t, res=self.skip(x, park_dx=f))
        g = ops.ActGate()
        t = self.conv2(self.conv1(x, act=ACT_LRELU, gate_out=g), gate_in=g)
        identity = self.skip(x) if self.skip is not None else x
        return self.gdn(t, res=identity)


class ResidualBlockUpsample(nn.Module):
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv = conv3x3(out_ch, out_ch)
        self.igdn = GDN(out_ch, inverse=True)
        self.upsample = subpel_conv3x3(in_ch, out_ch, upsample)

    def forward(self, x, pair=None):
        f = ops.GradFold() if x.requires_grad else None   # upsample's input gradient is added in subpel_conv's data-gradient epilogue
        g = ops.ActGate()                                 # conv's data gradient arrives already multiplied by LeakyReLU'
        if pair is not None:   # two blocks side by side on a batch-stacked input (the mean / scale hyper-synthesis nets)
            t = self.conv(self.subpel_conv(x, act=ACT_LRELU, pair=pair.subpel_conv, fold_in=f, gate_out=g), pair=pair.conv, gate_in=g)
            return self.igdn(t, res=self.upsample(x, pair=pair.upsample, park_dx=f), pair=pair.igdn)
        t = self.conv(self.subpel_conv(x, act=ACT_LRELU, fold_in=f, gate_out=g), gate_in=g)
        return self.igdn(t, res=self.upsample(x, park_dx=f))


class ResidualBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = conv3x3(in_ch, out_ch)
        self.leaky_relu = nn.LeakyReLU(inplace=True)
        self.conv2 = conv3x3(out_ch, out_ch)
        self.skip = conv1x1(in_ch, out_ch) if in_ch != out_ch else None

    def forward(self, x, extra_identity=0.0, out=None, grad_slot=None, pair=None):
        """lrelu(conv2(lrelu(conv1 x))) + identity (+ extra_identity * x, used by ConvTransBlock's `+ conv_x`)."""
        if self.skip is None:
            f = ops.GradFold() if x.requires_grad else None   # d(identity) is added in conv1's data-gradient epilogue
            g = ops.ActGate()
            t = self.conv1(x, act=ACT_LRELU, fold_in=f, grad_slot=grad_slot, pair=pair.conv1 if pair is not None else None, gate_out=g)
            return self.conv2(t, act=ACT_LRELU, res=x, res_scale=1.0 + extra_identity, fold_out=f, out=out,
                              pair=pair.conv2 if pair is not None else None, gate_in=g)
        assert pair is None
        assert out is None
        g = ops.ActGate()
        t = self.conv1(x, act=ACT_LRELU, gate_out=g)
        out = self.conv2(t, act=ACT_LRELU, res=self.skip(x), gate_in=g)
        return out + extra_identity * x if extra_identity else out


class ResidualUnit(nn.Module):
    def __init__(self, N):
        super().__init__()
        self.conv = nn.Sequential(conv1x1(N, N // 2), nn.ReLU(inplace=True), conv3x3(N // 2, N // 2), nn.ReLU(inplace=True),
                                  conv1x1(N // 2, N))
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, pair=None):
        """pair: a second ResidualUnit (second half of the batch) or a tuple of three (one quarter of the batch each)."""
        if isinstance(pair, (tuple, list)):
            q = [tuple(m.conv[k] for m in pair) for k in range(5)]
        else:
            q = pair.conv if pair is not None else (None,) * 5
        units = (self,) + (tuple(pair) if isinstance(pair, (tuple, list)) else ((pair,) if pair is not None else ()))
        if ops.residual_unit_fusable(x, units):
            # the 16x16 latent maps of the slice loop: the three layers in ONE launch, their data gradients in one more (csrc/fused_ru.hip)
            return ops.residual_unit(x, [(m.conv[0].weight, m.conv[0].bias, m.conv[2].weight, m.conv[2].bias, m.conv[4].weight, m.conv[4].bias)
                                         for m in units])
        f = ops.GradFold(gated=True) if x.requires_grad else None
        g0, g1 = ops.ActGate(), ops.ActGate()   # each ReLU' rides in the NEXT layer's data-gradient epilogue
        t = self.conv[0](x, act=ACT_RELU, pair=q[0], fold_in=f, gate_out=g0)
        t = self.conv[2](t, act=ACT_RELU, pair=q[2], gate_in=g0, gate_out=g1)
        return self.conv[4](t, act=ACT_RELU, res=x, res_first=True, pair=q[4], fold_out=f, gate_in=g1)


class AttentionBlock(nn.Module):
    def __init__(self, N):
        super().__init__()
        self.conv_a = nn.Sequential(ResidualUnit(N), ResidualUnit(N), ResidualUnit(N))
        self.conv_b = nn.Sequential(ResidualUnit(N), ResidualUnit(N), ResidualUnit(N), conv1x1(N, N))

    def forward(self, x):
        return ops.gate(self.conv_a(x), self.conv_b(x), x)


# ----------------------------------------------------------------------------- Swin-style blocks


class WMSA(nn.Module):
    """Window multi-head self-attention (/root/reference/models/CLC_run.py:108-169) on pixel-major tensors."""

    def __init__(self, input_dim, output_dim, head_dim, window_size, type):
        super().__init__()
        self.input_dim, self.output_dim, self.head_dim = input_dim, output_dim, head_dim
        self.n_heads = input_dim // head_dim
        self.window_size, self.type = window_size, type
        self.scale = head_dim ** -0.5
        self.embedding_layer = Linear(input_dim, 3 * input_dim, bias=True)
        self.relative_position_params = nn.Parameter(
            torch.nn.init.trunc_normal_(torch.zeros(self.n_heads, 2 * window_size - 1, 2 * window_size - 1), std=0.02))
        self.linear = Linear(input_dim, output_dim)

    def attend(self, qkv, res=None, fold_out=None):
        """everything after the embedding (the caller computed qkv, e.g. fused with the LayerNorm in front)"""
        a = ops.window_attention(qkv, self.relative_position_params, self.n_heads, self.window_size, self.type != "W")
        return self.linear(a, res=res, fold_out=fold_out)

    def forward(self, x, res=None, pair=None, fold_out=None):
        if pair is None:
            return self.attend(self.embedding_layer(x), res=res, fold_out=fold_out)
        qkv = self.embedding_layer(x, pair=pair.embedding_layer)
        a = ops.window_attention(qkv, self.relative_position_params, self.n_heads, self.window_size, self.type != "W",
                                 relbias2=pair.relative_position_params)   # one launch, per-half relative-position tables
        return self.linear(a, res=res, pair=pair.linear, fold_out=fold_out)


class Block(nn.Module):
    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path, type="W", input_resolution=None):
        super().__init__()
        assert type in ("W", "SW")
        if drop_path:
            raise ValueError("drop_path > 0 is not supported (the reference always passes 0)")
        self.type = type
        self.ln1 = LayerNorm(input_dim)
        self.msa = WMSA(input_dim, input_dim, head_dim, window_size, type)
        self.drop_path = nn.Identity()
        self.ln2 = LayerNorm(input_dim)
        self.mlp = nn.Sequential(Linear(input_dim, 4 * input_dim), GELU(), Linear(4 * input_dim, output_dim))

    def forward(self, x, pair=None, out=None, grad_slot=None):
        if pair is not None and (out is not None or grad_slot is not None):   # paired Block inside a paired ConvTransBlock
            f1 = ops.GradFold() if x.requires_grad else None
            x = self.msa(self.ln1(x, pair=pair.ln1, fold_in=f1, grad_slot=grad_slot), res=x, pair=pair.msa, fold_out=f1)
            f2 = ops.GradFold() if x.requires_grad else None
            g = ops.ActGate()
            h = self.mlp[0](self.ln2(x, pair=pair.ln2, fold_in=f2), act=ACT_GELU, pair=pair.mlp[0], gate_out=g)
            return self.mlp[2](h, res=x, pair=pair.mlp[2], fold_out=f2, out=out, gate_in=g)
        if pair is None:
            # x + f(LN(x)) twice: the residual gradients are added inside the LayerNorm backward passes
            f1 = ops.GradFold() if x.requires_grad else None
            emb = self.msa.embedding_layer
            if ops.lnlin_fusable(x, emb.weight):
                # large maps: ln1 + the qkv embedding in ONE launch; backward: the embedding's data gradient + ln1's backward pass + the
                # residual gradient in one more (csrc/fused_mlp.hip) — same bits as the separate launches
                qkv = ops.ln_linear(x, self.ln1.weight, self.ln1.bias, emb.weight, emb.bias, fold_in=f1, grad_slot=grad_slot)
                x = self.msa.attend(qkv, res=x, fold_out=f1)
            else:
                x = self.msa(self.ln1(x, fold_in=f1, grad_slot=grad_slot), res=x, fold_out=f1)
            if ops.mlp_ln_fusable(x, self.mlp[0].weight, self.mlp[2].weight):
                # ... and ln2 with them: LN in registers forward, its whole backward pass in the data-gradient launch's epilogue
                return ops.mlp_ln(x, self.ln2.weight, self.ln2.bias, self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias, out=out)
            f2 = ops.GradFold() if x.requires_grad else None
            if ops.mlp_fusable(x, self.mlp[0].weight, self.mlp[2].weight):
                # large maps: fc1 + GELU + fc2 + residual in ONE launch, the 256-channel hidden tensor never leaves the registers; the
                # backward pass recomputes it from the LayerNorm output (csrc/fused_mlp.hip) — same bits as the two launches below
                return ops.mlp(self.ln2(x, fold_in=f2), self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias,
                               res=x, fold_out=f2, out=out)
            g = ops.ActGate()
            h = self.mlp[0](self.ln2(x, fold_in=f2), act=ACT_GELU, gate_out=g)
            return self.mlp[2](h, res=x, fold_out=f2, out=out, gate_in=g)
        assert out is None
        f1 = ops.GradFold() if x.requires_grad else None
        x = self.msa(self.ln1(x, pair=pair.ln1, fold_in=f1), res=x, pair=pair.msa, fold_out=f1)
        f2 = ops.GradFold() if x.requires_grad else None
        g = ops.ActGate()
        h = self.mlp[0](self.ln2(x, pair=pair.ln2, fold_in=f2), act=ACT_GELU, pair=pair.mlp[0], gate_out=g)
        return self.mlp[2](h, res=x, pair=pair.mlp[2], fold_out=f2, gate_in=g)


class ConvTransBlock(nn.Module):
    def __init__(self, conv_dim, trans_dim, head_dim, window_size, drop_path, type="W"):
        super().__init__()
        self.conv_dim, self.trans_dim = conv_dim, trans_dim
        self.trans_block = Block(trans_dim, trans_dim, head_dim, window_size, drop_path, type)
        self.conv1_1 = Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1)
        self.conv1_2 = Conv2d(conv_dim + trans_dim, conv_dim + trans_dim, 1)
        self.conv_block = ResidualBlock(conv_dim, conv_dim)

    def forward(self, x, pair=None):
        """pair: a second ConvTransBlock of the same shape for the second half of a batch-stacked input."""
        q = pair
        f = ops.GradFold() if x.requires_grad else None   # d(x) of the outer residual rides in conv1_1's data-gradient epilogue
        u = self.conv1_1(x, fold_in=f, pair=q.conv1_1 if q is not None else None)
        slots = ops.GradSlots() if u.requires_grad else None
        cd, td = self.conv_dim, self.trans_dim
        c, t = ops.split_channels(u, (cd, td), slots)   # strided views, read in place by the kernels
        # the two branches write their results straight into the channel halves of conv1_2's input (no concatenation copy)
        buf = ops.new_act(u.shape[0], self.conv_dim + self.trans_dim, u.shape[2], u.shape[3], u)
        c = self.conv_block(c, extra_identity=1.0, out=buf[:, :cd], grad_slot=(slots, cd + td, 0) if slots is not None else None,
                            pair=q.conv_block if q is not None else None)
        t = self.trans_block(t, out=buf[:, cd:], grad_slot=(slots, cd + td, cd) if slots is not None else None,
                             pair=q.trans_block if q is not None else None)
        return self.conv1_2(ops.cat_halves(c, t, buf), res=x, fold_out=f, pair=q.conv1_2 if q is not None else None)


class SwinBlock(nn.Module):
    def __init__(self, input_dim, output_dim, head_dim, window_size, drop_path):
        super().__init__()
        self.block_1 = Block(input_dim, output_dim, head_dim, window_size, drop_path, type="W")
        self.block_2 = Block(input_dim, output_dim, head_dim, window_size, drop_path, type="SW")
        self.window_size = window_size

    def forward(self, x, pair=None, out=None):
        """out (paired form only): destination of the second block's result (a batch half of a wider buffer)."""
        if x.size(-1) <= self.window_size or x.size(-2) <= self.window_size:
            raise ValueError("SwinBlock: feature map must be larger than the window (input must be >= 256x256)")
        if pair is not None:
            return self.block_2(self.block_1(x, pair=pair.block_1), pair=pair.block_2, out=out)
        assert out is None
        return self.block_2(self.block_1(x))


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
            self.in_conv = self.out_conv = None

    def forward(self, x, pair=None, in_fold=None, in_slot=None):
        """pair: a second SWAtten of the same shape; x then holds both inputs stacked along the batch ([2B, C, H, W]) and
        every convolution / linear of the two modules runs as ONE launch over both halves.
        in_fold / in_slot: GradFold / gradient slot of the first convolution (ops.SliceSupport)."""
        if pair is not None:
            quad = ops.QUAD_UNITS and x.shape[0] % 2 == 0 and (x.shape[0] // 2) * x.shape[2] * x.shape[3] % 128 == 0
            u_buf = None
            if quad and self.in_conv is not None:
                # the stacked input of the quad chain, [x ; Swin(x)], is WRITTEN in place by its two producers (this convolution and the
                # Swin block's last linear) instead of concatenated afterwards
                u_buf = ops.new_act(2 * x.shape[0], self.in_conv.out_channels, x.shape[2], x.shape[3], x)
            if self.in_conv is not None:
                x = self.in_conv(x, pair=pair.in_conv, fold_in=in_fold, grad_slot=in_slot, out=u_buf[: x.shape[0]] if u_buf is not None else None)
            else:
                assert in_fold is None and in_slot is None
            def branch_a():
                a = x
                for m, q in zip(self.conv_a, pair.conv_a):
                    a = m(a, pair=q)
                return a

            if quad:
                # conv_a's three ResidualUnits (on x) and conv_b's (on the Swin output) are same-shaped layers on different data: stacked
                # along the batch they run as ONE chain of 9 launches with four filter sets (this net's a, the pair's a, this net's b,
                # the pair's b on the quarters) instead of two chains of 9 — these 16x16-map layers are latency-bound
                # x has three consumers (the quad chain, the Swin block, the gate's identity term): one alias each, their gradients summed
                # in one launch (ops.fanout) instead of two pairwise adds
                x_u, x_sw, x = ops.fanout(x, 3)
                if u_buf is not None:
                    u = ops.cat_batch(x_u, self.non_local_block(x_sw, pair=pair.non_local_block, out=u_buf[x.shape[0]:]), u_buf)
                else:
                    u = torch.cat((x_u, self.non_local_block(x_sw, pair=pair.non_local_block)), dim=0)
                for k in range(3):
                    u = self.conv_a[k](u, pair=(pair.conv_a[k], self.conv_b[k], pair.conv_b[k]))
                a, b = ops.split_batch(u)
                # ... and the gradients of the two halves are written into the halves of one buffer by their producers (the gate's d(a),
                # conv_b[3]'s data gradient), which split_batch's backward hands on as it is
                bs = ops.BatchSlots() if u.requires_grad else None
                b = self.conv_b[3](b, pair=pair.conv_b[3], grad_slot=(bs, 0, 1) if bs is not None else None)
                out = ops.gate(a, b, x, a_slot=(bs, 0) if bs is not None else None)
                return self.out_conv(out, pair=pair.out_conv) if self.out_conv is not None else out
            fork_a = ops.BRANCH_STREAMS and ops.PROFILE is None and "swatten_a" in ops.BRANCH_SLOTS
            if fork_a:   # conv_a(x) is independent of the Swin -> conv_b branch: run it on a forked stream
                with ops.fork("swatten_a", [x]) as f:
                    a = branch_a()
            b = self.non_local_block(x, pair=pair.non_local_block)
            for m, q in zip(self.conv_b, pair.conv_b):
                b = m(b, pair=q)
            if fork_a:
                f.join(a)
            else:
                a = branch_a()
            out = ops.gate(a, b, x)
            return self.out_conv(out, pair=pair.out_conv) if self.out_conv is not None else out
        if self.in_conv is not None:
            x = self.in_conv(x)
        if ops.BRANCH_STREAMS and ops.PROFILE is None and "swatten_a" in ops.BRANCH_SLOTS and "scale" not in ops.BRANCH_SLOTS:
            # (not inside the forked scale branch: nested forks crash hipGraph capture_end on ROCm 7.2)
            with ops.fork("swatten_a", [x]) as f:       # conv_a(x) is independent of the Swin -> conv_b branch
                a = self.conv_a(x)
            b = self.conv_b(self.non_local_block(x))
            f.join(a)
        else:
            a = self.conv_a(x)
            b = self.conv_b(self.non_local_block(x))
        out = ops.gate(a, b, x)
        return self.out_conv(out) if self.out_conv is not None else out
