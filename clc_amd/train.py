"""Training-step engine for the CLC hot path on MI355X.

Reference semantics reproduced (file:line into /root/reference):
  RateDistortionLoss          train_CLC.py:36-59    bpp = sum log(lik) / (-ln2 * N*H*W); loss = lmbda*255^2*mse + bpp
  configure_optimizers        train_CLC.py:81-117   AdamW(all but *.quantiles, lr) + AdamW(*.quantiles, aux lr)
  train_one_epoch inner loop  train_CLC.py:137-183  zero_grad, forward, loss.backward, clip_grad_norm_(1.0), nan_to_num_,
                                                    optimizer.step, aux_loss.backward, aux_optimizer.step
  data parallelism            run_ddp.sh:1-7 / train_CLC.py:472-473   one process per GPU, gradient mean over ranks

MI355X-first design of the step (not nn.DataParallel, not stock DDP):
  * the parameters that actually receive gradients ("live": 49.4 M of 70.6 M at N=64, R=1 — the cc_*/lrp_* twins and
    the dormant CLM modules never do) are re-homed into ONE flat fp32 arena, their .grad into a second
    one, Adam moments into two more.  Views keep every tensor's shape and channels_last strides.
  * clip_grad_norm_ + nan_to_num_ + AdamW for all live parameters = 3 launches over the arenas (clc_adamw_step).
  * gradient exchange = a few large all-reduces over the flat gradient arena (RCCL over xGMI; ring all-reduce is
    per-link bound, so buckets are big: 64 MiB), issued on a side stream and overlapped with the aux step.
  * the whole step (fwd + bwd + exchange-free part + optimizer) is captured once into a hipGraph and replayed.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import lib as _lib
from . import ops


# ------------------------------------------------------------------------------------------ loss
FUSED_RD_LOSS = os.environ.get("CLC_FUSED_RD_LOSS", "1") != "0"


def ms_ssim(X, Y, data_range=1.0):
    """pytorch_msssim.ms_ssim semantics (train_CLC.py:33-34) on the HIP MS-SSIM kernels (clc_ssim_scale_fwd/bwd)."""
    return ops.ms_ssim(X, Y.float().contiguous(memory_format=ops.CL), data_range=data_range)


class RateDistortionLoss(nn.Module):
    """Same constructor / output dict as the reference's criterion (train_CLC.py:36-59)."""

    def __init__(self, lmbda=1e-2, type="mse"):
        super().__init__()
        self.lmbda, self.type = lmbda, type

    def forward(self, output, target):
        N, _, H, W = target.size()
        num_pixels = N * H * W
        out = {}
        liks = output["likelihoods"]
        if self.type == "mse" and FUSED_RD_LOSS and isinstance(liks, dict) and list(liks.keys()) == ["y", "z"] and output["x_hat"].is_cuda:
            # the whole scalar tail (three sums, the divisions, lmbda * 255^2 * mse + bpp) in one launch; same bits as the expressions below
            tgt = target.float().contiguous(memory_format=ops.CL)
            out["bpp_loss"], out["mse_loss"], out["loss"] = ops.rd_loss_mse(liks["y"], liks["z"], output["x_hat"], tgt, self.lmbda, num_pixels)
            return out
        # sum(log(l)) / (-ln2 * n) == sum(log2(l)) / (-n); the sums are fixed-order two-stage reductions
        out["bpp_loss"] = sum(ops.sum_log2(l) for l in output["likelihoods"].values()) / (-num_pixels)
        if self.type == "mse":
            tgt = target.float().contiguous(memory_format=ops.CL)
            out["mse_loss"] = ops.sqdiff_sum(output["x_hat"], tgt) / tgt.numel()
            out["loss"] = self.lmbda * 255 ** 2 * out["mse_loss"] + out["bpp_loss"]
        else:
            out["ms_ssim_loss"] = ms_ssim(output["x_hat"], target, data_range=1.0)
            out["loss"] = self.lmbda * (1 - out["ms_ssim_loss"]) + out["bpp_loss"]
        return out


def configure_optimizers(net, args):
    """Reference-compatible (torch.optim.AdamW pair) — used when the reference trainer drives the model unchanged."""
    params = {n for n, p in net.named_parameters() if not n.endswith(".quantiles") and p.requires_grad}
    aux = {n for n, p in net.named_parameters() if n.endswith(".quantiles") and p.requires_grad}
    d = dict(net.named_parameters())
    assert not (params & aux) and len(params | aux) == len(d)
    opt = torch.optim.AdamW((d[n] for n in sorted(params)), lr=args.learning_rate)
    aux_opt = torch.optim.AdamW((d[n] for n in sorted(aux)), lr=args.aux_learning_rate)
    return opt, aux_opt


# ------------------------------------------------------------------------------- flat arenas


class FlatArena:
    """Re-homes a list of tensors into one flat buffer, keeping each tensor's shape and strides (device-agnostic)."""

    def __init__(self, tensors: List[torch.Tensor], align: int = 64):
        self.offsets, total = [], 0
        for t in tensors:
            self.offsets.append(total)
            total += (t.numel() + align - 1) // align * align
        self.numel = total
        dev = tensors[0].device if tensors else "cpu"
        self.flat = torch.zeros(max(total, 1), dtype=torch.float32, device=dev)
        self.views = [self.view_like(i, t) for i, t in enumerate(tensors)]

    def view_like(self, i, t):
        seg = self.flat[self.offsets[i]: self.offsets[i] + t.numel()]
        return seg.as_strided(t.shape, t.stride()) if t.numel() else seg.view(t.shape)


def _dense_strides_ok(t: torch.Tensor) -> bool:
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class GradSync:
    """Gradient mean over ranks as a few large all-reduces over a flat fp32 buffer (RCCL on GPU, gloo in CPU tests).

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by one link, so small buckets
    only add latency — default bucket = 64 MiB.  ``start()`` launches asynchronously, ``finish()`` waits and scales — unless the
    consumer folds the 1 / world factor into a pass it makes anyway (``fold_scale``: FusedAdamW's norm / update kernels take it
    as ``grad_scale``, which saves a read-modify-write pass over the whole arena per step).
    """

    def __init__(self, flat: torch.Tensor, bucket_bytes: int = 64 << 20, group=None, phases=None):
        """phases: list of (begin, end) element ranges of `flat` that become ready at different times (reverse-graph order: the
        gradients of the layers nearest the loss first); start(k) launches phase k's buckets.  Default: one phase = everything."""
        import torch.distributed as dist

        self.dist = dist
        self.flat, self.group = flat, group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        # CLC_FORCE_COLLECTIVES=1: a 1-rank process group still issues every all-reduce (a one-GPU box can then execute the RCCL path —
        # communicator, bucket views, stream ordering around the graph replays — that an 8-GPU run will take)
        self.active = self.world > 1 or (inited and os.environ.get("CLC_FORCE_COLLECTIVES", "0") == "1")
        self.backend = dist.get_backend(group) if inited else "none"
        n = max(1, bucket_bytes // 4)
        self.phases = []
        for a, b in (phases or [(0, flat.numel())]):
            self.phases.append([flat[i: min(i + n, b)] for i in range(a, b, n)])
        self.buckets = [bk for ph in self.phases for bk in ph]
        self.pending = []
        self.fold_scale = False   # True: finish() leaves the SUM in place, the optimizer applies 1 / world (TrainEngine._discover)
        self.launched = 0         # all-reduces issued so far (tests / the bench line's `config.collectives_per_step`)

    def start(self, phase=None):
        if not self.active:
            return
        bks = self.buckets if phase is None else self.phases[phase]
        self.pending += [self.dist.all_reduce(b, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True) for b in bks]
        self.launched += len(bks)

    def finish(self):
        if not self.active:
            return
        for w in self.pending:
            w.wait()
        self.pending = []
        if not self.fold_scale and self.world > 1:
            self.flat.mul_(1.0 / self.world)


def broadcast_parameters(module: nn.Module, src: int = 0, group=None):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(group) == 1 and os.environ.get("CLC_FORCE_COLLECTIVES", "0") != "1":
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel():
            dist.broadcast(t.data, src=src, group=group)


# ------------------------------------------------------------------------------ fused optimizer


class FusedAdamW:
    """clip_grad_norm_ + nan_to_num_ + AdamW (PyTorch defaults) over flat arenas: 3 kernel launches for all parameters."""

    def __init__(self, params: List[nn.Parameter], lr: float, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm: float = 0.0):
        import ctypes as C
        import numpy as np

        for p in params:
            if not _dense_strides_ok(p.data):
                raise ValueError("FusedAdamW needs dense (contiguous or channels_last) parameters")
        self.params = params
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.grad_scale = 1.0     # 1 / world when the gradient arena holds the rank SUM (GradSync.fold_scale)
        self.p_arena = FlatArena([p.data for p in params])
        self.g_arena = FlatArena([p.data for p in params])
        with torch.no_grad():
            for p, pv, gv in zip(params, self.p_arena.views, self.g_arena.views):
                pv.copy_(p.data)
                p.data = pv           # parameter now lives in the arena (same shape / strides)
                p.grad = gv           # persistent gradient view: autograd accumulates in place
                p._clc_direct = True  # ops' backward kernels may accumulate straight into p.grad (no add kernels)
        n = self.p_arena.numel
        dev = self.p_arena.flat.device
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        # optimizer state the kernels read from DEVICE memory, so a captured hipGraph keeps following it:
        #   step_dev = {t, 1 - beta1^t, sqrt(1 - beta2^t)} (advanced by clc_adam_tick), lr_dev = the current learning rate
        self.step_dev = torch.zeros(4, dtype=torch.float32, device=dev)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        L = _lib.load()
        chunk = L.clc_optim_chunk_elems()
        entry = _lib.ParamEntry(self.p_arena.flat.data_ptr(), self.g_arena.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), n)
        self.table = torch.frombuffer(bytearray(bytes(entry)), dtype=torch.uint8).to(dev)
        self.n_chunks = (n + chunk - 1) // chunk
        chunks = np.zeros((self.n_chunks, 2), dtype=np.int32)
        chunks[:, 1] = np.arange(self.n_chunks, dtype=np.int64) * chunk
        if n >= 2 ** 31:
            raise ValueError("arena too large for int32 chunk offsets")
        self.chunks = torch.from_numpy(chunks).to(dev)
        self.partials = torch.empty(self.n_chunks, dtype=torch.float32, device=dev)

    @property
    def grad_flat(self):
        return self.g_arena.flat

    def zero_grad(self):
        self.g_arena.flat.zero_()

    def set_lr(self, lr: float):
        """Learning-rate schedule hook (MultiStepLR in the reference, train_CLC.py:453,497): the kernels read the rate from
        device memory, so this also takes effect on an already captured hipGraph."""
        self.lr = float(lr)
        self.lr_dev.fill_(self.lr)

    def state_snapshot(self):
        return [t.clone() for t in (self.p_arena.flat, self.m, self.v, self.step_dev)]

    def state_restore(self, snap):
        for dst, src in zip((self.p_arena.flat, self.m, self.v, self.step_dev), snap):
            dst.copy_(src)

    def step(self):
        L, st = ops._L(), ops._stream()
        _lib.check(L.clc_adam_tick(self.step_dev.data_ptr(), float(self.betas[0]), float(self.betas[1]), st), "clc_adam_tick")
        sq = None
        if self.max_norm > 0:
            _lib.check(L.clc_grad_sqnorm_partials(self.table.data_ptr(), self.chunks.data_ptr(), self.n_chunks, self.partials.data_ptr(), float(self.grad_scale), st),
                       "clc_grad_sqnorm_partials")
            _lib.check(L.clc_sum_partials(self.partials.data_ptr(), self.n_chunks, 1.0, self.sqnorm.data_ptr(), 0, st), "clc_sum_partials")
            sq = self.sqnorm.data_ptr()
        _lib.check(L.clc_adamw_step(self.table.data_ptr(), self.chunks.data_ptr(), self.n_chunks, sq, float(self.max_norm), self.lr_dev.data_ptr(),
                                    float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), self.step_dev.data_ptr(), float(self.grad_scale), st), "clc_adamw_step")


# ------------------------------------------------------------------------------------ the engine


class FilterTransposer:
    """Keeps [Cin][T][Cout] copies of every conv / linear filter for the data-gradient kernels, refreshed by ONE launch
    per step (clc_filter_transpose_batched) instead of one small launch per layer inside backward."""

    def __init__(self, params: List[nn.Parameter]):
        ws = [p for p in params if p.dim() in (2, 4) and getattr(p, "_clc_is_filter", False)]
        self.n = len(ws)
        if not ws:
            return
        dev = ws[0].device
        total = sum(p.numel() for p in ws)
        self.buf = torch.empty(total, dtype=torch.float32, device=dev)
        entries, off, tiles = [], 0, 0
        for p in ws:
            Cout, Cin = p.shape[0], p.shape[1]
            T = p.shape[2] * p.shape[3] if p.dim() == 4 else 1
            wt = self.buf[off: off + p.numel()].view(Cin, T * Cout)
            off += p.numel()
            p._clc_wt = wt
            entries.append(_lib.TransposeEntry(p.data_ptr(), wt.data_ptr(), Cout, T, Cin, tiles))
            tiles += T * ((Cout + 31) // 32) * ((Cin + 31) // 32)
        self.total_tiles = tiles
        raw = b"".join(bytes(e) for e in entries)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def refresh(self):
        if self.n:
            _lib.check(ops._L().clc_filter_transpose_batched(self.table.data_ptr(), self.n, self.total_tiles, ops._stream()), "clc_filter_transpose_batched")


class HaloPacker:
    """Fragment-order images of the filters the halo-resident 3x3 kernel takes (csrc/conv_halo.hip: [k * C, C, 3, 3], C = 128 or 64, forward;
    the transposed image of the [C, C, 3, 3] ones for their data gradients), refreshed by ONE launch per step (clc_filter_pack_halo_batched)
    right behind the batched transpose, whose output it reads.  Which filters: the ones the discovery pass marked (ops.halo_packed sets
    `_clc_halo_use` on a forward use, `_clc_halo_use_t` on a data-gradient use).  ops.halo_packed() hands the images out while
    ops.WT_CACHE_VALID is set."""

    def __init__(self, params: List[nn.Parameter]):
        jobs = []   # (parameter, source tensor, rows, K, attribute)
        for p in params:
            if p.dim() != 4 or not getattr(p, "_clc_is_filter", False) or ops.to_kernel_weight(p) is not p:
                continue
            if getattr(p, "_clc_halo_use", False):
                jobs.append((p, p, p.shape[0], p.shape[1], "_clc_hpk"))
            wt = getattr(p, "_clc_wt", None)
            if getattr(p, "_clc_halo_use_t", False) and wt is not None:    # [Cin][9][Cout]: rows = Cin, K = Cout
                jobs.append((p, wt, p.shape[1], p.shape[0], "_clc_hpk_t"))
        self.n = len(jobs)
        if not jobs:
            return
        dev = jobs[0][0].device
        total = sum(rows * 9 * K for _, _, rows, K, _ in jobs)
        self.buf = torch.empty(total, dtype=torch.float32, device=dev)
        entries, off, blocks = [], 0, 0
        for p, src, rows, K, attr in jobs:
            n = rows * 9 * K
            out = self.buf[off: off + n]
            off += n
            setattr(p, attr, out)
            entries.append(_lib.HaloPackEntry(src.data_ptr(), out.data_ptr(), int(rows), int(K), blocks))
            blocks += (n // 4 + 255) // 256
        self.total_blocks = blocks
        self.table = torch.frombuffer(bytearray(b"".join(bytes(e) for e in entries)), dtype=torch.uint8).to(dev)

    def refresh(self):
        if self.n:
            _lib.check(ops._L().clc_filter_pack_halo_batched(self.table.data_ptr(), self.n, self.total_blocks, ops._stream()), "clc_filter_pack_halo_batched")


class WinoPacker:
    """Winograd-transformed images U = G g G^T of the filters conv_wino_kernel takes (csrc/conv_wino.hip), refreshed by ONE launch per step
    (clc_filter_wino_batched) behind the batched transpose.  Which filters: the ones the discovery pass marked (ops.wino_packed:
    `_clc_wino_use` forward, `_clc_wino_use_t` data gradient).  ops.wino_packed() hands the images out while ops.WT_CACHE_VALID is set."""

    def __init__(self, params: List[nn.Parameter]):
        jobs = []   # (parameter, source, rows, K, flip, attribute)
        for p in params:
            if p.dim() != 4 or not getattr(p, "_clc_is_filter", False) or ops.to_kernel_weight(p) is not p:
                continue
            if getattr(p, "_clc_wino_use", False):
                jobs.append((p, p, p.shape[0], p.shape[1], 0, "_clc_wu"))
            wt = getattr(p, "_clc_wt", None)
            if getattr(p, "_clc_wino_use_t", False) and wt is not None:    # [Cin][9][Cout]: rows = Cin, K = Cout, taps flipped
                jobs.append((p, wt, p.shape[1], p.shape[0], 1, "_clc_wu_t"))
        self.n = len(jobs)
        if not jobs:
            return
        dev = jobs[0][0].device
        total = sum(-(-rows // 128) * 128 * 16 * K for _, _, rows, K, _, _ in jobs)   # (rows padded to the kernels' 128-row filter tiles)
        self.buf = torch.empty(total, dtype=torch.float32, device=dev)
        entries, off, blocks = [], 0, 0
        for p, src, rows, K, flip, attr in jobs:
            n = -(-rows // 128) * 128 * 16 * K
            out = self.buf[off: off + n]
            off += n
            setattr(p, attr, out)
            entries.append(_lib.WinoEntry(src.data_ptr(), out.data_ptr(), int(rows), int(K), int(flip), blocks))
            blocks += (rows * K // 4 + 255) // 256
        self.total_blocks = blocks
        self.table = torch.frombuffer(bytearray(b"".join(bytes(e) for e in entries)), dtype=torch.uint8).to(dev)

    def refresh(self):
        if self.n:
            _lib.check(ops._L().clc_filter_wino_batched(self.table.data_ptr(), self.n, self.total_blocks, ops._stream()), "clc_filter_wino_batched")


class GDNReparamCache:
    """The effective (re-parametrised) gamma / beta of every GDN module — and gamma transposed for the data-gradient conv — refreshed by ONE
    launch per step (clc_gdn_reparam_fwd_batched) instead of one launch per module inside the forward pass.  Like the transposed filter
    images, the cached tensors are this step's only between refresh() and the optimizer update (ops.WT_CACHE_VALID)."""

    def __init__(self, model: nn.Module, live: List[nn.Parameter]):
        from .layers import GDN

        ids = {id(p) for p in live}
        mods = [m for m in model.modules() if isinstance(m, GDN) and id(m.gamma) in ids and id(m.beta) in ids and m.gamma.is_contiguous()]
        self.n = len(mods)
        if not mods:
            return
        dev = mods[0].gamma.device
        entries, blocks, self.keep = [], 0, []
        for m in mods:
            Cc = m.gamma.shape[0]
            gb, bb, ped = m._consts()
            g_eff = torch.empty((Cc, Cc), device=dev, dtype=torch.float32)
            g_eff_t = torch.empty((Cc, Cc), device=dev, dtype=torch.float32)
            b_eff = torch.empty((Cc,), device=dev, dtype=torch.float32)
            m.gamma._clc_gdn_eff = (g_eff, g_eff_t, b_eff)
            self.keep.append((g_eff, g_eff_t, b_eff))
            entries.append(_lib.GDNEntry(m.gamma.data_ptr(), m.beta.data_ptr(), g_eff.data_ptr(), g_eff_t.data_ptr(), b_eff.data_ptr(), Cc, blocks, gb, bb, ped))
            blocks += (Cc * Cc + Cc + 255) // 256
        self.total_blocks = blocks
        raw = b"".join(bytes(e) for e in entries)
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)

    def refresh(self):
        if self.n:
            _lib.check(ops._L().clc_gdn_reparam_fwd_batched(self.table.data_ptr(), self.n, self.total_blocks, ops._stream()), "clc_gdn_reparam_fwd_batched")


class TrainEngine:
    """One data-parallel training step of the reference loop (train_CLC.py:137-183), hipGraph-captured.

    engine = TrainEngine(model, lmbda=0.0067, lr=1e-4, aux_lr=1e-3, clip_max_norm=1.0)
    out = engine.step(x, refs)      # dict of device scalars: loss, bpp_loss, mse_loss, aux_loss
    """

    LATE_PREFIXES = ("g_a.", "ref_encoder.", "ref_feature_adapter.")   # modules whose backward runs last (they feed everything else)

    def __init__(self, model: nn.Module, lmbda: float, loss_type: str = "mse", lr: float = 1e-4, aux_lr: float = 1e-3,
                 clip_max_norm: float = 1.0, use_graph: bool = True, with_optimizer: bool = True, train_mode: bool = True, side_stream: bool = True,
                 criterion=None, optimizer_factory=None, precision: Optional[str] = None):
        """criterion / optimizer_factory: replaceable parts (defaults: the reference's RD loss and the fused HIP AdamW) — the
        multi-process CPU test drives the step structure with plain-torch stand-ins.  A custom criterion sees the reference forward's
        FULL output dict (`x_hat`, `likelihoods`, `para{means, scales, y}`, CLC_run.py:593-597); only the built-in RateDistortionLoss,
        which reads `x_hat` and the likelihoods alone, lets the model skip assembling `para`.
        precision: None (leave the process-wide setting alone), "f32" or "bf16" — clc_amd.set_precision() is applied around every step
        of this engine (the captured hipGraph keeps the kernels it was captured with)."""
        if precision not in (None, "f32", "bf16"):
            raise ValueError("precision must be None, 'f32' or 'bf16'")
        self.precision = precision
        self.model, self.criterion = model, (criterion or RateDistortionLoss(lmbda, loss_type))
        self.lean_outputs = criterion is None
        self._make_opt = optimizer_factory or (lambda params, lr, max_norm: FusedAdamW(params, lr=lr, max_norm=max_norm))
        self.lr, self.aux_lr, self.clip = lr, aux_lr, clip_max_norm
        self.use_graph, self.with_optimizer = use_graph, with_optimizer
        self.train_mode = train_mode   # False: deterministic rounding instead of noise (tests)
        self.side_stream = side_stream # filter gradients on a second stream, concurrent with the data-gradient chain
        self.opt: Optional[FusedAdamW] = None
        self.aux_opt: Optional[FusedAdamW] = None
        self.sync: Optional[GradSync] = None
        self.graph = None
        self._static = None
        self._out = None
        self.comm_events = None   # a list: the replayed multi-rank step brackets sync.finish() with HIP events (exposed_comm_ms())

    # -- discovery: which parameters does this (model, inputs) combination actually train?
    def _discover(self, x, refs):
        self._live_refs = refs is not None
        self.model.train(self.train_mode)
        for p in self.model.parameters():
            p.grad = None
        out = self.criterion(self.model(x, refs), x)
        out["loss"].backward()
        named = [(n, p) for n, p in self.model.named_parameters() if p.grad is not None and not n.endswith(".quantiles")]
        # arena order = [encoders | everything downstream]: the backward pass reaches the analysis transform and the reference
        # branch LAST, so their gradients are the second (small) exchange phase and everything else can go on the wire while
        # their backward still runs (overlap in reverse-graph order)
        late = [p for n, p in named if n.startswith(self.LATE_PREFIXES)]
        early = [p for n, p in named if not n.startswith(self.LATE_PREFIXES)]
        live = late + early
        aux = [p for n, p in self.model.named_parameters() if n.endswith(".quantiles")]
        if self.side_stream:
            ops.enable_wgrad_stream(True)
            ops.enable_branch_streams(True)
        ops.enable_deferred_reductions(True)
        self.opt = self._make_opt(live, self.lr, self.clip)
        self.aux_opt = self._make_opt(aux, self.aux_lr, 0.0)
        self.transposer = FilterTransposer(live)
        self.halo_packer = HaloPacker(live)
        self.wino_packer = WinoPacker(live)
        self.gdn_cache = GDNReparamCache(self.model, live)
        cut = self.opt.p_arena.offsets[len(late)] if (late and early) else 0
        n_el = self.opt.grad_flat.numel()
        self.early_params = early
        self.sync = GradSync(self.opt.grad_flat, phases=([(cut, n_el), (0, cut)] if cut else None))
        # (wire_clm: the reference latents also feed h_a, so (y, ref_features) no longer separates the encoders from the rest)
        self.two_phase = bool(cut) and hasattr(self.model, "_boundary_ok") and not getattr(self.model, "wire_clm", False)
        self.aux_sync = GradSync(self.aux_opt.grad_flat)
        # the rank mean's 1 / world rides in the optimizer's own passes over the gradients where the optimizer can take it
        for sync, opt in ((self.sync, self.opt), (self.aux_sync, self.aux_opt)):
            if self.with_optimizer and sync.active and hasattr(opt, "grad_scale"):
                opt.grad_scale, sync.fold_scale = 1.0 / sync.world, True

    def _one_like(self, t):
        """the gradient seed of a scalar loss, allocated once per device (outside any capture: the eager warm-up steps come first)"""
        one = getattr(self, "_one", None)
        if one is None or one.device != t.device or one.shape != t.shape:
            one = self._one = torch.ones_like(t)
        return one

    def _model_out(self, x, refs):
        # (the built-in criterion reads x_hat and the likelihoods only: the concatenated means / scales of the output dict are not assembled)
        self.model._lean_outputs = self.lean_outputs
        try:
            return self.model(x, refs)
        finally:
            self.model._lean_outputs = False

    def _fwd_bwd(self, x, refs):
        self.opt.zero_grad()
        self.aux_opt.zero_grad()
        self.transposer.refresh()
        self.halo_packer.refresh()
        self.wino_packer.refresh()
        self.gdn_cache.refresh()
        ops.WT_CACHE_VALID = True      # (the transposed filter images are this step's: see ops.WT_CACHE_VALID)
        try:
            out = self.criterion(self._model_out(x, refs), x)
            out["loss"].backward(self._one_like(out["loss"]))   # (a cached 1: no ones_like fill launch per step)
        finally:
            ops.WT_CACHE_VALID = False
        ops.join_side_streams()   # filter gradients computed on the side stream are complete from here on
        return out

    # -- the same pass cut in two at the outputs of the analysis transform / reference branch (multi-GPU overlap)
    def _fwd_bwd_early(self, x, refs):
        """forward + loss + backward of everything DOWNSTREAM of (y, ref_features); their gradients are parked on the boundary."""
        self.opt.zero_grad()
        self.aux_opt.zero_grad()
        self.transposer.refresh()
        self.halo_packer.refresh()
        self.wino_packer.refresh()
        self.gdn_cache.refresh()
        self.model._keep_boundary = True
        ops.WT_CACHE_VALID = True
        try:
            out = self.criterion(self._model_out(x, refs), x)
        finally:
            self.model._keep_boundary = False
            ops.WT_CACHE_VALID = False
        self._bt = [t for t in self.model._boundary if t is not None and t.requires_grad]
        self.model._boundary = None
        for t in self._bt:
            t.grad = None
        # (retain_graph: the engine would otherwise also release the saved tensors of the boundary's producers, which stage 2 needs)
        ops.WT_CACHE_VALID = True
        try:
            torch.autograd.backward([out["loss"]], [self._one_like(out["loss"])], inputs=list(self.early_params) + self._bt, retain_graph=True)
        finally:
            ops.WT_CACHE_VALID = False
        ops.join_side_streams()
        return out

    def _bwd_late(self):
        """backward of the analysis transform and the reference branch from the parked boundary gradients."""
        ops.WT_CACHE_VALID = True
        try:
            torch.autograd.backward(self._bt, [t.grad for t in self._bt])
        finally:
            ops.WT_CACHE_VALID = False
        ops.join_side_streams()
        self._bt = None

    def _opt_steps(self, out):
        if self.with_optimizer:
            self.opt.step()
        aux_loss = self.model.aux_loss()
        aux_loss.backward(self._one_like(aux_loss))
        out["aux_loss"] = aux_loss.detach()
        return out

    def _finish(self, out):
        if self.with_optimizer:
            self.aux_sync.start()
            self.aux_sync.finish()
            self.aux_opt.step()
        return {k: v.detach() for k, v in out.items()}

    def _eager_step(self, x, refs, split=False):
        if split and self.two_phase:
            out = self._fwd_bwd_early(x, refs)
            self.sync.start(0)        # everything downstream of the encoders goes on the wire ...
            self._bwd_late()          # ... while the encoders' backward runs
            self.sync.start(1)
        else:
            out = self._fwd_bwd(x, refs)
            self.sync.start()
        self.sync.finish()
        out = self._opt_steps(out)
        return self._finish(out)

    def set_lr(self, lr: float = None, aux_lr: float = None):
        """Learning-rate schedule hook (the reference steps a MultiStepLR once per epoch, train_CLC.py:453,497).  The rates live
        in device memory, so the change also reaches an already captured hipGraph."""
        if lr is not None:
            self.lr = float(lr)
            if self.opt is not None:
                self.opt.set_lr(lr)
        if aux_lr is not None:
            self.aux_lr = float(aux_lr)
            if self.aux_opt is not None:
                self.aux_opt.set_lr(aux_lr)

    def exposed_comm_ms(self, x, refs=None, steps: int = 5):
        """Average time per step the compute stream spends WAITING for the gradient exchange (HIP events around sync.finish(), after
        graph A2): what the all-reduce costs beyond the encoders' backward it overlaps.  None for the single-graph step."""
        if self.graph is None or not isinstance(self.graph, tuple):
            return None
        self.comm_events = []
        try:
            for _ in range(steps):
                self.step(x, refs)
            torch.cuda.synchronize()
            ts = [a.elapsed_time(b) for a, b in self.comm_events]
        finally:
            self.comm_events = None
        return sum(ts) / max(1, len(ts))

    @staticmethod
    def _signature(x, refs):
        return (tuple(x.shape), None if refs is None else tuple(tuple(r.shape) for r in refs))

    def step(self, x, refs=None):
        if self.precision is None:
            return self._step(x, refs)
        from . import set_precision

        old = set_precision(self.precision)
        try:
            return self._step(x, refs)
        finally:
            set_precision(old)

    def _step(self, x, refs=None):
        ops.WEIGHTS_EPOCH += 1   # this engine's kernels update the parameters through raw pointers (no version-counter bump): cached images keyed on it go stale
        refs = list(refs) if refs is not None else None
        if self.opt is None:
            self._discover(x, refs)
        if (refs is not None) != self._live_refs:
            # the set of parameters that receive gradients (and with it the flat arenas / DDP buckets) depends on whether
            # references are given (CLC_run.py:550-580 picks the ref_* or the plain slice nets)
            raise ValueError("TrainEngine was set up %s reference frames; build a second engine for the other mode"
                             % ("with" if self._live_refs else "without"))
        # CLC_FORCE_SPLIT_GRAPHS=1 exercises the multi-GPU structure (graph A | exchange | graph B) on one GPU
        single = not self.sync.active and os.environ.get("CLC_FORCE_SPLIT_GRAPHS", "0") != "1"
        if self.graph is not None:
            single = not isinstance(self.graph, tuple)   # (the structure is fixed once captured)
        if not self.use_graph:
            return self._eager_step(x, refs, split=not single)
        sig = self._signature(x, refs)
        if self.graph is not None and sig != self._static_sig:
            # a batch of another shape (the reference's DataLoader has no drop_last, train_CLC.py:428-434: the last batch of an
            # epoch is short): run it eagerly — same arithmetic, launch overhead for this one step only
            return self._eager_step(x, refs, split=not single)
        if self.graph is None:
            # warm up on a side stream (allocator + lazy kernel attributes), then capture.  The warm-up steps must not count as
            # training steps: parameters, Adam moments and step counters are restored afterwards, so the FIRST replay is the
            # first optimizer update (the reference applies exactly one per batch, train_CLC.py:137-183)
            # (kept channels_last: the model's / criterion's `.contiguous(channels_last)` is then a no-op instead of a copy per step and image)
            cl = lambda t: t.float().clone(memory_format=ops.CL) if t.dim() == 4 else t.clone()
            self._static = (cl(x), [cl(r) for r in refs] if refs is not None else None)
            self._static_sig = sig
            snap = (self.opt.state_snapshot(), self.aux_opt.state_snapshot())
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._eager_step(*self._static, split=not single)
                self.opt.state_restore(snap[0])
                self.aux_opt.state_restore(snap[1])
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            del snap
            if single:
                self.graph = torch.cuda.CUDAGraph()
                with ops.capture_guard(), torch.cuda.graph(self.graph, capture_error_mode=ops.graph_capture_mode()):
                    self._out = self._eager_step(*self._static)
            elif self.two_phase:
                # collectives stay outside the graphs: A1 = forward + backward down to the encoders' outputs | exchange of those
                # gradients starts | A2 = the encoders' backward (overlaps the exchange) | rest of the exchange | B = optimizer + aux
                self.graph = (torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph())
                with ops.capture_guard(), torch.cuda.graph(self.graph[0], capture_error_mode=ops.graph_capture_mode()):
                    self._mid = self._fwd_bwd_early(*self._static)
                with ops.capture_guard(), torch.cuda.graph(self.graph[1], pool=self.graph[0].pool(), capture_error_mode=ops.graph_capture_mode()):
                    self._bwd_late()
                with ops.capture_guard(), torch.cuda.graph(self.graph[2], pool=self.graph[0].pool(), capture_error_mode=ops.graph_capture_mode()):
                    self._out2 = self._opt_steps(dict(self._mid))
            else:
                # collectives stay outside the graphs: graph A = fwd+bwd, exchange, graph B = optimizer + aux
                self.graph = (torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph())
                with ops.capture_guard(), torch.cuda.graph(self.graph[0], capture_error_mode=ops.graph_capture_mode()):
                    self._mid = self._fwd_bwd(*self._static)
                with ops.capture_guard(), torch.cuda.graph(self.graph[1], pool=self.graph[0].pool(), capture_error_mode=ops.graph_capture_mode()):
                    self._out2 = self._opt_steps(dict(self._mid))
        sx, srefs = self._static
        sx.copy_(x, non_blocking=True)
        if refs is not None:
            for d, r in zip(srefs, refs):
                d.copy_(r, non_blocking=True)
        if single:
            self.graph.replay()
            return self._out
        self.graph[0].replay()
        if len(self.graph) == 3:
            self.sync.start(0)
            self.graph[1].replay()
            self.sync.start(1)
        else:
            self.sync.start()
        if self.comm_events is not None:
            # exposed part of the exchange on the GPU timeline: the compute stream reaches e0 when graph A2 is done and e1 when the
            # last bucket has landed (finish() makes this stream wait for the collective stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.sync.finish()
            e1.record()
            self.comm_events.append((e0, e1))
        else:
            self.sync.finish()
        self.graph[-1].replay()
        return self._finish(dict(self._out2))
