"""Autograd-aware operator layer over the C ABI (libclc_hip.so).

Tensors keep the reference's logical NCHW shapes but live in channels_last memory
(= the NHWC layout the kernels use), so every ``permute``/``Rearrange`` of the reference
(/root/reference/models/CLC_run.py:215,217,260,263) is a no-op here and ``split``/``chunk``
are strided views the kernels read directly through their leading-dimension argument.

Every op runs on the current HIP stream, allocates only through the PyTorch caching
allocator, never synchronises -> the whole training step is hipGraph-capturable.
There is no CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
from torch.autograd import Function

from . import lib as _lib
from .lib import (ACT_GELU, ACT_HALFTANH, ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SAVED_DERIV, IN_NONE, IN_SQUARE, NORM_GDN,
                  NORM_IGDN, NORM_MUL2, NORM_NONE)

CL = torch.channels_last

# Filter gradients are only needed by the optimizer, so they are DEFERRED: queued per layer and launched WGRAD_GROUP at a
# time through clc_conv2d_wgrad_batched (one grid per tile shape + one slab reduce for the whole group) instead of 2-3
# launch-bound kernels per layer.  Where they run:
#   * in line on the launching stream (default).  Under hipGraph replay on ROCm 7.2 a graph with parallel branches pays
#     ~4.4 us of cross-queue marker latency in front of EVERY kernel that follows an event record, and the branches barely
#     overlap (tools/graph_fork_probe.py; bench: 1 graph queue 216 img/s vs 4 queues 210) — a linear graph is faster;
#   * on a side stream (CLC_WGRAD_STREAM=1), concurrent with the data-gradient chain: the better choice for eager execution.
# Tensors a deferred launch still reads are kept alive in the device state's `keepalive` until join_side_streams().
WGRAD_DEFER = False


class _DeviceState:
    """Everything mutable that the deferred machinery keeps between ops, ONE instance PER DEVICE.

    The reference's in-tree multi-GPU mode is nn.DataParallel (/root/reference/train_CLC.py:74-79, 472-473): one process, one forward
    thread per GPU, and the autograd engine's own worker thread per GPU for the backward.  Queues that were module globals would be
    shared by those threads; keyed by device they are touched by one thread at a time (a replica's forward thread, then its device's
    backward thread), which is all the ordering they need.  (Per-THREAD state would be wrong: loss.backward() runs the CUDA nodes on
    the engine's device thread while join_side_streams() is called from the caller's.)"""

    __slots__ = ("pending", "pending_flop", "pending_streams", "pending_post", "last_defer_vid", "pending_reduce", "keepalive",
                 "branch_pool", "wgrad_stream", "group_ws")

    def __init__(self):
        self.pending = {}            # kernel-family id -> [(descriptor, keep-alive tuple)]
        self.pending_flop = 0.0
        self.pending_streams = {}    # streams that queued a problem since the last flush (their work must be ordered before the launch)
        self.pending_post = {}       # kernel-family id -> callables run right after that family's launch, on its stream
        self.last_defer_vid = 0      # family of the problem queued last (a post hook registers itself under it)
        self.pending_reduce = []
        self.keepalive = []          # tensors a deferred / cross-stream launch still reads, until join_side_streams()
        self.branch_pool = {}
        self.wgrad_stream = None     # side stream of the filter gradients (CLC_WGRAD_STREAM=1), else None
        self.group_ws = {}           # None / "capture" -> [stream-K partial-tile workspace (642 MiB), stream that used it last]


_STATES = {}


_HAS_GPU = []


def _S() -> _DeviceState:
    if not _HAS_GPU:
        _HAS_GPU.append(torch.cuda.is_available())
    # (no GPU: the multi-process CPU tests drive TrainEngine's step structure with plain-torch stand-ins — one idle state, key -1)
    dev = torch.cuda.current_device() if _HAS_GPU[0] else -1
    st = _STATES.get(dev)
    if st is None:
        st = _STATES.setdefault(dev, _DeviceState())   # (setdefault: two replica threads may arrive here together)
    return st


def enable_wgrad_stream(enable=True):
    """Turn on deferred, grouped filter gradients (and the side stream when CLC_WGRAD_STREAM=1) — the TrainEngine's mode."""
    global WGRAD_DEFER
    WGRAD_DEFER = bool(enable)
    _S().wgrad_stream = torch.cuda.Stream() if (enable and os.environ.get("CLC_WGRAD_STREAM", "0") == "1") else None


WGRAD_GROUP = int(os.environ.get("CLC_WGRAD_GROUP", "64"))             # problems per grouped launch (library cap: 64)
# ... or as soon as this much work is queued.  Default: effectively never (flush by count).
WGRAD_FLUSH_GFLOP = float(os.environ.get("CLC_WGRAD_FLUSH_GFLOP", "1000"))


def _group_ws():
    """(ptr, bytes) of this device's stream-K workspace.  Launches on one stream are ordered; when the launching stream changes
    (warm-up stream -> training stream, main stream <-> filter-gradient side stream) the new stream first waits for the old one.
    The buffer is allocated OUTSIDE hipGraph capture (TrainEngine's eager warm-up steps come first); a capture that meets no
    buffer gets a private one that lives in the graph's pool, under its own key, and is never handed to eager launches."""
    tab = _S().group_ws
    cur = torch.cuda.current_stream()
    capturing = torch.cuda.is_current_stream_capturing()
    ent = tab.get(None)
    if ent is None:
        nbytes = _L().clc_conv2d_wgrad_group_workspace_bytes()
        if capturing:
            ent = tab.get("capture")
            if ent is None:
                ent = tab["capture"] = [torch.empty((nbytes + 3) // 4, device="cuda", dtype=torch.float32), cur]
        else:
            ent = tab[None] = [torch.empty((nbytes + 3) // 4, device="cuda", dtype=torch.float32), cur]
    ws, last = ent
    if last.cuda_stream != cur.cuda_stream:
        # (a capturing stream cannot wait on work recorded outside its capture: the eager predecessor has long been synchronised
        # by then — TrainEngine.step synchronises between warm-up and capture)
        if not capturing or _is_capturing(last):
            cur.wait_stream(last)
        ent[1] = cur
    return ws.data_ptr(), ws.numel() * 4


def _is_capturing(stream):
    with torch.cuda.stream(stream):
        return torch.cuda.is_current_stream_capturing()


def release_workspaces():
    """Teardown hook: drop the stream-K workspaces, branch-stream pools and queues of every device of this process (re-created on
    demand).  Call it only between steps: queued filter gradients are dropped with their queues."""
    _STATES.clear()


def _launch_wgrad_group(arr, n):
    wp, wb = _group_ws()
    _lib.check(_lib.load().clc_conv2d_wgrad_batched_sk(arr, n, wp, wb, _stream()), "clc_conv2d_wgrad_batched_sk")


def _flush_family(vid, target):
    """launch the queued problems of ONE kernel family (clc_conv2d_wgrad_variant id) as a stream-K group, then its post hooks"""
    S = _S()
    items = S.pending.pop(vid, [])
    posts = S.pending_post.pop(vid, [])
    with torch.cuda.stream(target):
        if items:
            arr = (_lib.WgradDesc * len(items))(*[d for d, _ in items])
            if PROFILE is None:
                _launch_wgrad_group(arr, len(items))
            else:   # bracketed by events and credited with its problems' algorithmic FLOPs (bench.py's roofline leg)
                # one bracket per KERNEL: the C side runs a family's LDS-DMA-staged problems (no fused activation derivative on dy, no
                # squared input) and the register-staged rest as two launches — the same split here, one call each
                parts = ([it for it in items if not it[0].dys and it[0].in_op == IN_NONE], [it for it in items if it[0].dys or it[0].in_op != IN_NONE])
                dfl = lambda d: 2.0 * d.N * d.OH * d.OW * d.ks * d.ks * d.Cin * d.Cout
                for staged_by_dma, its in zip((True, False), parts):
                    if not its:
                        continue
                    ds = [d for d, _ in its]
                    sub = (_lib.WgradDesc * len(ds))(*ds)
                    fl = sum(dfl(d) for d in ds)
                    by_owner = {}
                    for d, k in its:
                        by_owner[k[-1]] = by_owner.get(k[-1], 0.0) + dfl(d)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _launch_wgrad_group(sub, len(ds))
                    e1.record()
                    PROFILE.append(ProfRec("conv_wgrad_group", vid, fl, e0, e1, f"{len(ds)} problems dma={int(staged_by_dma)}", owner_flops=by_owner,
                                           relaunch=(lambda sub=sub, n=len(ds), keep=[k for _, k in its]: _launch_wgrad_group(sub, n))))
        for fn in posts:
            fn()
    S.keepalive.append([k for _, k in items])


def _flush_target():
    S = _S()
    cur = torch.cuda.current_stream()
    target = S.wgrad_stream if S.wgrad_stream is not None else cur
    for sid, st in S.pending_streams.items():   # the operands were produced on these streams
        if sid != target.cuda_stream:
            target.wait_stream(st)
    S.pending_streams.clear()
    return target


def flush_wgrads():
    """Launch everything queued.  Problems wait in one queue PER KERNEL FAMILY (a family's queue is launched as soon as it holds
    WGRAD_GROUP problems): a flush then costs one stream-K grid (+ fix-up) per family that has work, with full groups,
    instead of one under-filled grid of every family for each 64 problems in arrival order."""
    S = _S()
    if not S.pending and not S.pending_post:
        return
    target = _flush_target()
    for vid in sorted(set(S.pending) | set(S.pending_post)):
        _flush_family(vid, target)
    S.pending_flop = 0.0


# Deferred parameter-gradient reductions (LayerNorm gamma/beta, relative-position bias): the backward kernels leave their
# per-block partial rows in their workspaces and ONE clc_partial_reduce_batched call per step sums them into the gradient
# arena (join_side_streams) — instead of a 5-16 us reduce launch behind each of the ~110 backward kernels.
DEFER_REDUCTIONS = False


def enable_deferred_reductions(enable=True):
    global DEFER_REDUCTIONS
    DEFER_REDUCTIONS = bool(enable)


def defer_reduce(partial, nblocks, n, out0, out1, split, offset=0):
    """offset: first float of this entry's partial rows inside `partial`."""
    e = _lib.ReduceEntry()
    e.partial, e.nblocks, e.n = partial.data_ptr() + 4 * int(offset), int(nblocks), int(n)
    e.out0, e.out1, e.split, e.accumulate = out0.data_ptr(), (out1.data_ptr() if out1 is not None else None), int(split), 1
    _S().pending_reduce.append((e, (partial, out0, out1)))


def flush_reductions():
    S = _S()
    if not S.pending_reduce:
        return
    arr = (_lib.ReduceEntry * len(S.pending_reduce))(*[e for e, _ in S.pending_reduce])
    _lib.check(_L().clc_partial_reduce_batched(arr, len(S.pending_reduce), _stream()), "clc_partial_reduce_batched")
    S.keepalive.append([k for _, k in S.pending_reduce])
    S.pending_reduce.clear()


def join_side_streams():
    flush_wgrads()
    S = _S()
    if S.wgrad_stream is not None:
        torch.cuda.current_stream().wait_stream(S.wgrad_stream)
    flush_reductions()
    S.keepalive.clear()


# Branch streams: independent sub-graphs of the latency-bound 16x16 slice loop (mean vs scale parameter nets, the
# conv_a vs Swin->conv_b branches of SWAtten) run on forked HIP streams and join again, so the ~10 us kernels of different
# branches overlap on the 256 CUs; under hipGraph capture the fork/join becomes parallel graph branches. Autograd runs
# each op's backward on the stream of its forward, so the backward overlaps the same way.
BRANCH_STREAMS = False
# Paired layers: the mean- and scale-parameter nets of a slice run as one launch per layer over a stacked batch (CLC_PAIR=0:
# two launches, optionally on forked streams).
PAIR_SLICES = int(os.environ.get("CLC_PAIR", "1"))   # default on: half the launches of the slice loop, no reliance on hipGraph branch concurrency
SUPPORT_BUFFER = int(os.environ.get("CLC_SUPPORT_BUFFER", "1"))   # slice loop: one support buffer + one gradient buffer instead of per-slice concatenations (SliceSupport)
MATERIALIZE_DZ = int(os.environ.get("CLC_MATERIALIZE_DZ", "32768"))   # rows from which a 3x3 layer's activation backward is its own pass (0: always fused into the gradient kernels' loaders)
QUAD_UNITS = int(os.environ.get("CLC_QUAD_UNITS", "1"))   # paired SWAttens: the ResidualUnits of conv_a and conv_b of both nets in one chain (4 filter sets)
PAIR_HYPER = int(os.environ.get("CLC_PAIR_HYPER", "1"))   # also pair the mean / scale hyper-synthesis nets (h_mean_s, h_scale_s)
BRANCH_SLOTS = set(os.environ.get("CLC_BRANCH", "scale").split(","))   # which forks are taken (debug knob)


def enable_branch_streams(enable=True):
    global BRANCH_STREAMS
    BRANCH_STREAMS = bool(enable)


class fork:
    """with ops.fork(slot, inputs) as f: out = branch(...);  f.join(out) afterwards on the parent stream."""

    def __init__(self, slot, inputs):
        self.parent = torch.cuda.current_stream()
        key = (self.parent.cuda_stream, slot)
        pool = _S().branch_pool
        if key not in pool:
            pool[key] = torch.cuda.Stream()
        self.stream = pool[key]
        self.inputs = [t for t in inputs if t is not None]

    def __enter__(self):
        self.stream.wait_stream(self.parent)
        # tensors crossing streams are kept alive until the end of the step (ops.join_side_streams) instead of
        # record_stream(): every later use of a branch stream starts with wait_stream(parent), which orders any reuse of
        # their memory after the consumers — and record_stream's deferred events crash hipGraph capture_end (ROCm 7.2)
        _S().keepalive.append(self.inputs)
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        self.parent.wait_stream(self.stream)
        _S().keepalive.append(outputs)


class GradFold:
    """Folds the gradient add of a two-consumer tensor into a kernel epilogue.

    x feeds (a) a residual add inside a later layer's epilogue and (b) an earlier layer (conv / LayerNorm) of the branch
    that ends in that add — ResidualUnit, ResidualBlock, Block and ConvTransBlock all have this shape (CLC_run.py:172-220,
    CompressAI layers).  Backward reaches (a) first: instead of returning d(residual) to autograd (which would launch an
    elementwise add when (b)'s gradient arrives), (a) parks it here and (b)'s gradient kernel adds it in its epilogue.  If the
    order is ever the other way round, (a) finds `consumed` set and returns its gradient to autograd as usual."""

    __slots__ = ("pending", "scale", "gate", "consumed", "gated")

    def __init__(self, gated=False):
        """gated: the folding layer is a convolution (its data-gradient epilogue can apply an activation-derivative gate)."""
        self.pending, self.scale, self.gate, self.consumed, self.gated = None, 1.0, None, False, bool(gated)

    def can_park(self):
        return not self.consumed and self.pending is None

    def park(self, g, scale=1.0, gate=None):
        """-> True when parked (the caller then returns None for that input).  gate = (saved, act, saved_is_pre): the parked
        gradient still has to be multiplied by act'(saved) — only conv data-gradient epilogues can do that."""
        if not self.can_park():
            return False
        self.pending, self.scale, self.gate = g, float(scale), gate
        return True

    def take(self):
        self.consumed = True
        g, s, gate = self.pending, self.scale, self.gate
        self.pending = self.gate = None
        return g, s, gate


ACT_GATES = int(os.environ.get("CLC_ACT_GATE", "1"))   # 0: every layer applies its own activation derivative (A/B knob)


class ActGate:
    """Hands a layer's activation derivative to the ONE convolution that consumes its output.

    y = act(conv_a(x)); z = conv_b(y).  In backward conv_b's data-gradient kernel produces dL/dy; multiplying it by act'(.) in that
    kernel's epilogue gives conv_a the gradient of its PRE-activation directly, so conv_a's own data- and filter-gradient kernels
    need no operand prologue (they run on the LDS-DMA path, and skip one tensor read each).  conv_a registers what the derivative
    needs (its output for LeakyReLU / ReLU, the stored derivative for GELU); conv_b applies it and sets `done`; conv_a then treats
    its incoming gradient as already gated.  Only valid when conv_b is the sole consumer of y."""

    __slots__ = ("saved", "act", "pre", "done")

    def __init__(self):
        self.saved, self.act, self.pre, self.done = None, ACT_NONE, False, False


class GradSlots:
    """One gradient buffer for the consumers of a channel split (ConvTransBlock: conv1_1's output feeds the conv branch and
    the transformer branch, CLC_run.py:212-214): each consumer's backward kernel writes its gradient straight into its
    channel range of the buffer, and the split's backward returns the buffer as it is — no strided gather copies."""

    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def view(self, like, total_c, off, n):
        if self.buf is None:
            self.buf = new_act(like.shape[0], total_c, like.shape[2], like.shape[3], like)
        return self.buf[:, off:off + n]


class BatchSlots:
    """GradSlots for the BATCH halves of a stacked tensor: the two consumers of split_batch(u) write their gradients straight into the
    halves of one buffer (a convolution through `grad_slot=(slots, 0, half)`, the gate through `slot=`), and _SplitBatchFn.backward
    hands that buffer on as it is — no concatenation."""

    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def view(self, like, total_c, half, n):
        N = like.shape[0]
        if self.buf is None:
            self.buf = new_act(2 * N, like.shape[1], like.shape[2], like.shape[3], like)
        return self.buf[half * N:(half + 1) * N]


# When set to a list, EVERY launch through the C ABI is bracketed by HIP events on the launch stream and recorded as a ProfRec
# (bench.py's roofline leg, tools/profile_shapes.py): convolutions / filter gradients with their tile-variant id, algorithmic FLOPs
# and bytes (recorded at their call sites), everything else (LayerNorm, attention, elementwise, optimizer ...) through the
# _ProfiledLib proxy under the name of its C entry point.
PROFILE = None
PROFILE_OWNER = "other"   # which sub-network the launches belong to: the model sets it forward, every Function restores it backward


class ProfRec:
    __slots__ = ("fam", "variant", "flops", "e0", "e1", "label", "nbytes", "owner", "owner_flops", "relaunch")

    def __init__(self, fam, variant, flops, e0, e1, label="", nbytes=0.0, owner=None, owner_flops=None, relaunch=None):
        self.fam, self.variant, self.flops, self.e0, self.e1, self.label, self.nbytes = fam, variant, flops, e0, e1, label, nbytes
        self.owner = PROFILE_OWNER if owner is None else owner
        self.owner_flops = owner_flops    # grouped launches: {owner: FLOPs} of the problems inside
        self.relaunch = relaunch          # re-issues the same launch (same descriptors, operands kept alive): graph-replay timing

    def ms(self):
        return self.e0.elapsed_time(self.e1)


def set_owner(name):
    global PROFILE_OWNER
    PROFILE_OWNER = name


def _own(ctx):
    """forward: remember the owning sub-network for this node's backward (profiling only)"""
    if PROFILE is not None:
        ctx.owner = PROFILE_OWNER


def _reown(ctx):
    if PROFILE is not None:
        set_owner(getattr(ctx, "owner", "other"))


_PROF_HINT = [0.0, ""]   # algorithmic FLOPs / label of the NEXT proxied launch (set by the op that knows them)


def _prof_hint(flops=0.0, label=""):
    if PROFILE is not None:
        _PROF_HINT[0], _PROF_HINT[1] = float(flops), label


class _ProfiledLib:
    """libclc_hip.so with every launching entry point bracketed by HIP events (PROFILE mode only)."""

    _NO_LAUNCH = ("clc_last_error", "clc_version", "clc_set_tuning", "clc_conv2d_wgrad_variant", "clc_optim_chunk_elems", "clc_gauss_lik_partials",
                  "clc_ssim_init", "clc_pmf_to_quantized_cdf")

    def __init__(self, L):
        self._lib_ = L
        self._cache = {}

    def __getattr__(self, name):
        f = self._cache.get(name)
        if f is not None:
            return f
        raw = getattr(self._lib_, name)
        if name.endswith(("_bytes", "_blocks")) or name.startswith("clc_rans") or name in self._NO_LAUNCH:
            f = raw
        else:
            def f(*a, _raw=raw, _name=name):
                if PROFILE is None:
                    return _raw(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _raw(*a)
                e1.record()
                PROFILE.append(ProfRec(_name, 0, _PROF_HINT[0], e0, e1, _PROF_HINT[1]))
                _PROF_HINT[0], _PROF_HINT[1] = 0.0, ""
                return rc
        self._cache[name] = f
        return f


_PLIB = [None]


def _L():
    if PROFILE is None:
        return _lib.load()
    if _PLIB[0] is None:
        _PLIB[0] = _ProfiledLib(_lib.load())
    return _PLIB[0]


def graph_capture_mode() -> str:
    """capture_error_mode for torch.cuda.graph().  With a process group alive, ProcessGroupNCCL's watchdog THREAD polls its work events
    (hipEventQuery) at any time; under the default global capture mode that call is illegal while another thread captures
    ("operation not permitted when stream is capturing" -> the watchdog aborts the process: met on the MI355X the first time RCCL ran this
    code, tests/test_rccl_one_rank_gpu.py).  thread_local restricts the legality checks to the capturing thread."""
    import torch.distributed as dist

    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


class capture_guard:
    """Around a hipGraph capture: the cyclic garbage collector stays off.  A collection that happens to run inside a capture may finalise
    objects of an EARLIER engine (captured graphs, pinned host buffers: hipFreeHost / hipGraphDestroy) — calls that are illegal while a stream
    captures and abort the process (met in the full GPU test run: "Fatal Python error: Aborted ... Garbage-collecting" inside
    CodecEngine._capture).  torch.cuda.graph collects BEFORE it starts capturing, not during."""

    def __enter__(self):
        import gc

        self._was = gc.isenabled()
        gc.collect()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc

        if self._was:
            gc.enable()
        return False


def _stream():
    """raw handle of the current HIP stream of the current device.  (torch.cuda.current_stream() builds a Stream object through four layers of
    Python — 8 us a call on the host, 1 400 calls per eager backward pass: a fifth of the reference-style loop's host time.  The two C
    accessors cost 0.3 us.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.ClcError(f"{what}: tensor is on {t.device}; the clc_amd product path runs on the GPU only "
                            "(no CPU fallback by design)")
    if t.dtype != torch.float32:
        raise _lib.ClcError(f"{what}: dtype {t.dtype} (fp32 only)")


def nhwc(t: torch.Tensor):
    """(tensor, ptr, N, H, W, C, ld) of a logical-NCHW tensor stored pixel-major; copies only if it must."""
    assert t.dim() == 4, t.shape
    N, Cc, H, W = t.shape
    ok = t.stride(1) == 1 or Cc == 1
    ld = t.stride(3) if W > 1 else (t.stride(2) if H > 1 else (t.stride(0) if N > 1 else Cc))
    if W > 1 and H > 1:
        ok = ok and t.stride(2) == W * ld
    if N > 1:
        ok = ok and t.stride(0) == H * W * ld
    ok = ok and ld >= Cc
    if not ok:
        t = t.contiguous(memory_format=CL)
        if t.stride(1) != 1 and Cc > 1:  # ambiguous-stride corner cases
            t = torch.empty((N, Cc, H, W), device=t.device, dtype=t.dtype, memory_format=CL).copy_(t)
        ld = Cc
    return t, t.data_ptr(), N, H, W, Cc, ld


def dense(t: torch.Tensor) -> torch.Tensor:
    """pixel-major AND channel-dense (ld == C); copies a channel-slice view."""
    t, _p, N, H, W, Cc, ld = nhwc(t)
    if ld != Cc:
        t = torch.empty((N, Cc, H, W), device=t.device, dtype=t.dtype, memory_format=CL).copy_(t)
    return t


def new_act(N, Cc, H, W, like: torch.Tensor):
    return torch.empty((N, Cc, H, W), device=like.device, dtype=torch.float32, memory_format=CL)


def to_kernel_weight(w: torch.Tensor) -> torch.Tensor:
    """[Co,Ci,kh,kw] -> physically [Co][kh][kw][Ci] (no copy when already channels_last)."""
    if w.dim() == 2:
        return w if w.is_contiguous() else w.contiguous()
    if w.is_contiguous(memory_format=CL) or (w.shape[2] == 1 and w.shape[3] == 1 and w.is_contiguous()):
        return w
    return w.contiguous(memory_format=CL)


# ------------------------------------------------------------------------------------------ conv


# ---- halo-resident 3x3 kernel (csrc/conv_halo.hip): the filter is ALSO needed in fragment order.  Where the packed image comes from:
#   * inside a TrainEngine step (WT_CACHE_VALID): one batched launch per step packs every eligible filter and transposed filter
#     (clc_amd.train.HaloPacker -> w._clc_hpk / w._clc_hpk_t), like the transposed images themselves;
#   * anywhere else: packed PER USE (one 5-us launch in front of a >= 75-us convolution).  No cache across calls: nothing tells this module
#     that `w.data.copy_(...)` or a kernel writing through a raw pointer changed the weights (neither moves a version counter), and a stale
#     image would be silently wrong results.  Inside a captured graph (CodecEngine) the pack launch is part of the graph, so a replay
#     always packs the weights of the moment.
HALO = os.environ.get("CLC_HALO", "1") != "0"
WEIGHTS_EPOCH = 0   # bumped by TrainEngine per step (its kernels update the arena through raw pointers); informational


def halo_ok(N, H, W, Cin, rows, ks, stride):
    """input-channel count (128 / 64) if a [rows][3][3][Cin] filter on an N x H x W map may take the halo kernel, else 0 (the C side re-checks
    everything and falls through to the tiled kernels)"""
    if not (HALO and ks == 3 and stride == 1 and Cin in (128, 64) and rows % Cin == 0 and H % 8 == 0 and W % 16 == 0):
        return 0
    if not (_L().clc_get_tuning(22) & (1 if Cin == 128 else 2)):   # tuning key 22: bit 0 = 128-channel layers, bit 1 = 64-channel layers (off by default)
        return 0
    return Cin if N * (H // 8) * (W // 16) * (rows // Cin) >= 128 else 0


def halo_pack(wk, rows, K=128):
    """[rows][9][K] filter rows (K-contiguous; K = 128 or 64) -> fragment order (clc_filter_pack_halo)."""
    out = torch.empty(rows * 9 * K, device=wk.device, dtype=torch.float32)
    _lib.check(_L().clc_filter_pack_halo(wk.data_ptr(), out.data_ptr(), int(rows), int(K), _stream()), "clc_filter_pack_halo")
    return out


def halo_packed(w, transposed_image=None):
    """packed image of parameter `w`'s forward filter, or (transposed_image given: the [Cin][9][Cout] image of this step) of its transposed one.
    The parameter is marked as a user (`_clc_halo_use` / `_clc_halo_use_t`): clc_amd.train.HaloPacker packs exactly the marked ones."""
    tr = transposed_image is not None
    setattr(w, "_clc_halo_use_t" if tr else "_clc_halo_use", True)
    if WT_CACHE_VALID:
        pk = getattr(w, "_clc_hpk_t" if tr else "_clc_hpk", None)
        if pk is not None:
            return pk
    if tr:
        return halo_pack(transposed_image, transposed_image.shape[0], w.shape[0])
    return halo_pack(to_kernel_weight(w), w.shape[0], w.shape[1])


# ---- Winograd F(2x2, 3x3) kernel (csrc/conv_wino.hip): needs the TRANSFORMED filter U = G g G^T in fragment order.  Same provenance rules as the
# halo kernel's image: per-step batched launch inside a TrainEngine step (clc_amd.train.WinoPacker -> w._clc_wu / w._clc_wu_t), per use elsewhere.
# (switched by tuning key 23 / the CLC_WINO environment variable: bit 0 = forward launches of a recorded (training) pass, bit 1 = data gradients.
#  Never taken without autograd recording: eval forwards, the parity measurement and the codec keep the direct kernels and their bits.)


def wino_ok(N, H, W, Cin, rows, ks, stride, transposed=False):
    """may a [rows][3][3][Cin] filter on an N x H x W map take the Winograd kernels?  (the C side re-checks everything and falls through)"""
    mode = _L().clc_get_tuning(23)
    if not (mode & (2 if transposed else 1)) or not (ks == 3 and stride == 1 and H % 8 == 0 and W % 16 == 0 and Cin <= 1024):
        return False
    # the C side's rule (clc_conv_wino_launch): the 128-wide kernel from 192 items up, else the 64-wide one (bit 2) from 128 items up
    ptiles = N * (H // 8) * (W // 16)
    if Cin % 128 == 0 and rows % 128 == 0 and not (mode & 8) and ptiles * (rows // 128) >= 192:
        return True
    return bool(mode & 4) and Cin % 64 == 0 and rows % 64 == 0 and ptiles * (rows // 64) >= 128


def wino_pack(wk, rows, K, flip=False):
    """[rows][9][K] filter rows -> U = G g G^T in fragment order (clc_filter_wino; 16 / 9 of the size, rows padded to 128).  flip: taps reversed
    (data gradients)."""
    out = torch.empty(-(-rows // 128) * 128 * 16 * K, device=wk.device, dtype=torch.float32)
    _lib.check(_L().clc_filter_wino(wk.data_ptr(), out.data_ptr(), int(rows), int(K), int(bool(flip)), _stream()), "clc_filter_wino")
    return out


def wino_packed(w, transposed_image=None):
    tr = transposed_image is not None
    setattr(w, "_clc_wino_use_t" if tr else "_clc_wino_use", True)
    if WT_CACHE_VALID:
        u = getattr(w, "_clc_wu_t" if tr else "_clc_wu", None)
        if u is not None:
            return u
    if tr:
        return wino_pack(transposed_image, transposed_image.shape[0], w.shape[0], flip=True)
    return wino_pack(to_kernel_weight(w), w.shape[0], w.shape[1])


def conv_raw(x, w, bias=None, *, ks, stride=1, pad=None, act=ACT_NONE, in_op=IN_NONE, norm=NORM_NONE, mul=None,
             res=None, res_scale=1.0, res_first=False, y_pre=None, shuffle=False, transposed=False, out=None, out_hw=None,
             xs=None, xs_act=ACT_NONE, xs_pre=False, w2=None, bias2=None, pre_deriv=False, res_gate=None, out_gate=None, wx=None, batch_variant_ok=False,
             wpk=None, wwino=None):
    """One clc_conv2d launch. ``w`` must already be in kernel layout [Cout][ks][ks][Cin].
    wpk: the same filter in the halo kernel's fragment order (halo_packed / halo_pack), or None.
    batch_variant_ok: the summation order may depend on the batch size (training forward passes only, never the codec path).
    wx: ((w3, bias3), (w4, bias4)) — with (w2, bias2), four filter sets on the four quarters of the batch."""
    _require_gpu(x, "conv2d")
    x, xp, N, H, W, Cin, ldx = nhwc(x)
    Cout = w.shape[0]
    pad = ks // 2 if pad is None else pad
    if transposed:
        OH, OW = out_hw
    else:
        OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    if out is None:
        out = new_act(N, Cout // 4, 2 * OH, 2 * OW, x) if shuffle else new_act(N, Cout, OH, OW, x)
    o, op, _, _, _, _, ldy = nhwc(out)
    assert o is out, "conv2d: output buffer must be pixel-major"
    d = _lib.ConvDesc()
    d.x, d.N, d.H, d.W, d.Cin, d.ldx = xp, N, H, W, Cin, ldx
    d.w = w.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.y, d.OH, d.OW, d.Cout, d.ldy = op, OH, OW, Cout, ldy
    d.ks, d.stride, d.pad = ks, stride, pad
    d.transposed, d.in_op, d.act, d.norm, d.shuffle = int(transposed), in_op, act, norm, int(shuffle)
    keep = [x, w, bias, out]
    if w2 is not None:   # second half of the batch on a second filter set (paired layers)
        d.w2 = w2.data_ptr()
        d.bias2 = bias2.data_ptr() if bias2 is not None else None
        if wx is not None:   # ... or four sets on its quarters
            (w3, b3), (w4, b4) = wx
            d.w3, d.w4 = w3.data_ptr(), w4.data_ptr()
            d.bias3 = b3.data_ptr() if b3 is not None else None
            d.bias4 = b4.data_ptr() if b4 is not None else None
            keep += [w3, b3, w4, b4]
    if mul is not None:
        m, mp, *_r, ldm = nhwc(mul)
        d.mul, d.ldm = mp, ldm
        keep.append(m)
    if res is not None:
        r, rp, *_r, ldr = nhwc(res)
        d.res, d.ldr, d.res_scale, d.res_first = rp, ldr, float(res_scale), int(res_first)
        keep.append(r)
        if res_gate is not None:   # (saved tensor, activation, saved-is-pre-activation): res term *= act'(saved)
            gt, gp, *_r, ldg = nhwc(res_gate[0])
            d.res_gate, d.ldg, d.res_gate_act, d.res_gate_pre = gp, ldg, int(res_gate[1]), int(res_gate[2])
            keep.append(gt)
    if out_gate is not None:   # (saved tensor, activation, saved-is-pre-activation): whole result *= act'(saved)
        ot, otp, *_r, ldog = nhwc(out_gate[0])
        d.out_gate, d.ldog, d.out_gate_act, d.out_gate_pre = otp, ldog, int(out_gate[1]), int(out_gate[2])
        keep.append(ot)
    if y_pre is not None:
        q, qp, *_r, ldp = nhwc(y_pre)
        assert q is y_pre
        d.y_pre, d.ldp = qp, ldp
        d.pre_deriv = int(pre_deriv)
    if xs is not None:   # fused activation backward: x <- x * act'(xs)
        xs_t, xsp, *_r, ldxs = nhwc(xs)
        d.xs, d.ldxs, d.xs_act, d.xs_pre = xsp, ldxs, xs_act, int(xs_pre)
        keep.append(xs_t)
    d.batch_variant_ok = int(bool(batch_variant_ok))
    if wpk is not None:   # the same filter in conv_halo3x3_kernel's fragment order: clc_conv2d takes that kernel when the launch qualifies
        d.w_packed = wpk.data_ptr()
        keep.append(wpk)
    if wwino is not None:   # ... or its Winograd transform (takes precedence)
        d.w_wino = wwino.data_ptr()
        keep.append(wwino)
    if (transposed and H * W <= 1024) or (batch_variant_ok and ks == 3 and OH * OW <= 256):   # scratch for the K split of under-filled grids (0 bytes: no split)
        nws = _lib.load().clc_conv2d_workspace_bytes(C.byref(d))
        if nws:
            ws = torch.empty((nws + 3) // 4, device=x.device, dtype=torch.float32)
            d.workspace, d.workspace_bytes = ws.data_ptr(), nws
            keep.append(ws)
    if PROFILE is None:
        _lib.check(_lib.load().clc_conv2d(C.byref(d), _stream()), "clc_conv2d")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        variant = _lib.check(_lib.load().clc_conv2d(C.byref(d), _stream()), "clc_conv2d")
        e1.record()
        pix = N * H * W if transposed else N * OH * OW
        opix = N * OH * OW
        # algorithmic HBM bytes of the launch: operand read once, result written once, every epilogue tensor once
        nbytes = 4.0 * (N * H * W * Cin + Cout * ks * ks * Cin
                        + opix * Cout * (1 + (res is not None) + (mul is not None) + (y_pre is not None) + (out_gate is not None)))
        PROFILE.append(ProfRec("conv_igemm" if variant != 1 else "conv_direct_small", variant, 2.0 * pix * ks * ks * Cin * Cout, e0, e1,
                               f"{'dgrad' if transposed else 'fwd'} {Cin}->{Cout} k{ks} s{stride} {N}x{H}x{W}" + (" +actbwd" if xs is not None else "") + (" +sq" if in_op == IN_SQUARE else "")
                               + "".join(t for t, on in ((" b", bias is not None), (f" a{act}", act != ACT_NONE), (" res", res is not None), (" mul", mul is not None),
                                                         (" pre", y_pre is not None), (" rg", res_gate is not None), (" og", out_gate is not None), (" w2", w2 is not None and wx is None), (" w4", wx is not None)) if on),
                               nbytes, relaunch=(lambda d=d, keep=keep: _lib.load().clc_conv2d(C.byref(d), _stream()))))
    return out


def im2col_small(x, ks, stride, ldc=32):
    """Patch rows of a few-channel image: [N,C,H,W] -> [N,ldc,OH,OW] (pixel-major rows of ks*ks*C values, zero-padded to ldc).
    Not differentiable: the RGB input of the analysis / reference encoders needs no gradient."""
    _require_gpu(x, "im2col_small")
    assert not x.requires_grad
    x, xp, N, H, W, Cc, ldx = nhwc(x)
    pad = ks // 2
    OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    col = new_act(N, ldc, OH, OW, x)
    _lib.check(_L().clc_im2col_small(xp, ldx, N, H, W, Cc, ks, stride, pad, col.data_ptr(), ldc, OH, OW, _stream()), "clc_im2col_small")
    return col


class _StemPackFn(Function):
    """(3x3/s2 filter [co,cin,3,3], 1x1 skip filter [co,cin,1,1]) -> their 32-column patch-row matrices (clc_stem_pack): one launch
    forward, one launch adding both gradients straight into the optimizer's gradient buffers backward."""

    @staticmethod
    def forward(ctx, w3, w1x1):
        co, cin = w3.shape[0], w3.shape[1]
        k3, k1 = to_kernel_weight(w3), to_kernel_weight(w1x1)
        w1 = torch.empty((co, 32), device=w3.device, dtype=torch.float32)
        ws = torch.empty((co, 32), device=w3.device, dtype=torch.float32)
        _lib.check(_L().clc_stem_pack(k3.data_ptr(), k1.data_ptr(), w1.data_ptr(), ws.data_ptr(), co, cin, _stream()), "clc_stem_pack")
        ctx.refs = (w3, w1x1)
        return w1, ws

    @staticmethod
    def backward(ctx, dw1, dws):
        w3, w1x1 = ctx.refs
        co, cin = w3.shape[0], w3.shape[1]
        g3, g1 = _direct_grad(w3), _direct_grad(w1x1)
        if g3 is not None and g1 is not None and to_kernel_weight(w3) is w3 and to_kernel_weight(w1x1) is w1x1:
            _lib.check(_L().clc_stem_unpack_add(dw1.contiguous().data_ptr(), dws.contiguous().data_ptr(), g3.data_ptr(), g1.data_ptr(), co, cin,
                                                _stream()), "clc_stem_unpack_add")
            return None, None
        return (dw1[:, : 9 * cin].reshape(co, 3, 3, cin).permute(0, 3, 1, 2), dws[:, 4 * cin: 5 * cin].reshape(co, cin, 1, 1))


def stem_filters(w3, w1x1):
    return _StemPackFn.apply(w3, w1x1)


def filter_transpose(w, Cout, T, Cin):
    wt = torch.empty(Cin * T * Cout, device=w.device, dtype=torch.float32)
    _lib.check(_L().clc_filter_transpose(w.data_ptr(), wt.data_ptr(), Cout, T, Cin, _stream()), "clc_filter_transpose")
    return wt.view(Cin, T * Cout)


def wgrad_batched(problems):
    """problems: list of dicts of wgrad_raw keyword arguments (with x, dy, dw_out[, db_out]); one clc_conv2d_wgrad_batched
    call on the current stream, accumulating into the given buffers."""
    queued = []
    for kw in problems:
        queued.append(wgrad_raw(**kw, _collect=True))
    arr = (_lib.WgradDesc * len(queued))(*[d for d, _ in queued])
    _launch_wgrad_group(arr, len(queued))
    return [k for _, k in queued]   # operands / workspaces: keep until the stream has run the launches


def wgrad_raw(x, dy, *, ks, stride, pad, Cout, Cin, want_bias, in_op=IN_NONE, dw_out=None, db_out=None, dys=None, dys_act=ACT_NONE,
              dys_pre=False, defer=False, _collect=False, accumulate=None):
    """Returns (dw [Cout, ks*ks*Cin] flat kernel layout, dbias or None).  With dw_out / db_out (persistent gradient
    buffers in kernel layout) the result is ACCUMULATED into them and (None, None) is returned.  defer=True (direct mode
    on the side stream only) queues the problem for the next grouped launch (flush_wgrads)."""
    x, xp, N, H, W, _, ldx = nhwc(x)
    dy, dp, _, OH, OW, _, lddy = nhwc(dy)
    direct = dw_out is not None
    dw = dw_out if direct else torch.empty(Cout * ks * ks * Cin, device=x.device, dtype=torch.float32)
    db = db_out if direct else (torch.empty(Cout, device=x.device, dtype=torch.float32) if want_bias else None)
    d = _lib.WgradDesc()
    d.x, d.N, d.H, d.W, d.Cin, d.ldx = xp, N, H, W, Cin, ldx
    d.dy, d.OH, d.OW, d.Cout, d.lddy = dp, OH, OW, Cout, lddy
    d.dw, d.dbias = dw.data_ptr(), (db.data_ptr() if db is not None else None)
    d.ks, d.stride, d.pad, d.in_op, d.accumulate = ks, stride, pad, in_op, int(direct if accumulate is None else accumulate)
    dys_t = None
    if dys is not None:   # fused activation backward: dy <- dy * act'(dys)
        dys_t, dysp, *_r, lddys = nhwc(dys)
        d.dys, d.lddys, d.dys_act, d.dys_pre = dysp, lddys, dys_act, int(dys_pre)
    # grouped (stream-K) launches keep their partial tiles in the device's group workspace: only the small-Cin split path still
    # needs per-problem slabs there
    grouped = _collect or defer
    nbytes = (_L().clc_conv2d_wgrad_sk_workspace_bytes if grouped else _L().clc_conv2d_wgrad_workspace_bytes)(C.byref(d))
    ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32) if nbytes else None
    d.workspace, d.workspace_bytes = (ws.data_ptr() if ws is not None else None), nbytes
    if _collect:
        return d, (x, dy, dys_t, dw, db, ws)
    if defer:
        S = _S()
        cur = torch.cuda.current_stream()   # the operands are produced on this stream: ordered before the launch at flush time
        S.pending_streams[cur.cuda_stream] = cur
        vid = _L().clc_conv2d_wgrad_variant(C.byref(d))
        S.last_defer_vid = vid
        q = S.pending.setdefault(vid, [])
        q.append((d, (x, dy, dys_t, dw, db, ws, PROFILE_OWNER)))   # (last entry: owning sub-network, profiling only)
        S.pending_flop += 2.0 * N * OH * OW * ks * ks * Cin * Cout
        if S.pending_flop >= WGRAD_FLUSH_GFLOP * 1e9:
            flush_wgrads()
        elif len(q) >= WGRAD_GROUP:
            _flush_family(vid, _flush_target())
        return None, None
    if PROFILE is None:
        _lib.check(_lib.load().clc_conv2d_wgrad(C.byref(d), _stream()), "clc_conv2d_wgrad")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        variant = _lib.check(_lib.load().clc_conv2d_wgrad(C.byref(d), _stream()), "clc_conv2d_wgrad")
        e1.record()
        PROFILE.append(ProfRec("conv_wgrad" if variant != 1 else "wgrad_small", variant, 2.0 * N * OH * OW * ks * ks * Cin * Cout, e0, e1,
                               f"wgrad {Cin}->{Cout} k{ks} s{stride} {N}x{H}x{W}"))
    return (None, None) if direct else (dw, db)


def _direct_grad(t):
    """Persistent gradient buffer of a parameter laid out exactly like it (set up by clc_amd.train.FusedAdamW), or None."""
    if t is None or not getattr(t, "_clc_direct", False):
        return None
    g = t.grad
    if g is None or g.stride() != t.stride() or g.shape != t.shape:
        return None
    return g


def _dw_to_param_layout(dw_flat, w_param):
    """kernel layout [Co][kh][kw][Ci] -> a tensor shaped like the parameter (logical OIHW / [out,in])."""
    if w_param.dim() == 2:
        return dw_flat.view(w_param.shape)
    Co, Ci, kh, kw = w_param.shape
    if kh == 1 and kw == 1:
        # [Co][1][1][Ci] IS [Co][Ci][1][1]: hand the gradient out with the PARAMETER's strides (a 1x1 filter is "contiguous" and
        # "channels_last" at once, with different stride tuples) — stock DistributedDataParallel checks a gradient's strides against
        # its bucket view's (the gradient layout contract) and falls back to a copy + warning when they differ
        return dw_flat.as_strided(w_param.shape, w_param.stride()) if w_param.stride(1) == 1 else dw_flat.view(Co, Ci, 1, 1)
    return dw_flat.view(Co, kh, kw, Ci).permute(0, 3, 1, 2)  # logical OIHW, channels_last strides


def act_bwd(dy, saved, use_pre, act):
    dy, dyp, N, H, W, Cc, lddy = nhwc(dy)
    saved, sp, *_r, lds = nhwc(saved)
    dz = new_act(N, Cc, H, W, dy)
    _lib.check(_L().clc_act_bwd(dyp, lddy, sp, lds, int(use_pre), act, dz.data_ptr(), Cc, N * H * W, Cc, _stream()), "clc_act_bwd")
    return dz


def unshuffle_act_bwd(dy, saved, use_pre, act):
    """dy [N, C/4, 2H, 2W] (gradient of a PixelShuffle(2)-stored conv output) -> dz [N, C, H, W] = unshuffle(dy * act'(saved))."""
    dy, dyp, N, H2, W2, Q, lddy = nhwc(dy)
    sp, lds = None, 0
    if saved is not None:
        saved, sp, *_r, lds = nhwc(saved)
    dz = new_act(N, Q * 4, H2 // 2, W2 // 2, dy)
    _lib.check(_L().clc_unshuffle_act_bwd(dyp, lddy, sp, lds, int(use_pre), act, dz.data_ptr(), N, H2 // 2, W2 // 2, Q * 4, _stream()), "clc_unshuffle_act_bwd")
    return dz


class _ConvFn(Function):
    """y = act(conv(x, w) + b) + res_scale * res, optionally PixelShuffle(2)-stored.  With (w2, b2) the second half of
    the batch is convolved with the second filter set in the same launch (paired layers)."""

    @staticmethod
    def forward(ctx, x, w, b, res, ks, stride, act, res_scale, shuffle, res_first, w2=None, b2=None, fold_in=None, fold_out=None, out_buf=None,
                grad_slot=None, park_dx=None, gate_in=None, gate_out=None, w3=None, b3=None, w4=None, b4=None):
        _own(ctx)
        if x.dim() != 4 or x.shape[1] != w.shape[1]:
            # (the kernels take Cin from the activation: a mismatch would walk off the end of the filter buffer)
            raise _lib.ClcError(f"conv2d: input {tuple(x.shape)} does not match the filter {tuple(w.shape)} (expected {w.shape[1]} input channels)")
        wk = to_kernel_weight(w)
        ctx.park_dx = park_dx   # GradFold that takes this layer's input gradient (a sibling layer on the same input adds it in its epilogue)
        wk2 = to_kernel_weight(w2) if w2 is not None else None
        wkx = ((to_kernel_weight(w3), b3), (to_kernel_weight(w4), b4)) if w3 is not None else None
        need_grad = _recording(ctx)
        # the activation derivative needs the pre-activation whenever the output does not determine it
        save_pre = need_grad and (act == ACT_GELU or (act in (ACT_LRELU, ACT_RELU, ACT_HALFTANH) and res is not None and not res_first))
        N, _, H, W = x.shape
        Cout = w.shape[0]
        pad = ks // 2
        OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y_pre = None
        if save_pre:
            y_pre = new_act(N, Cout // 4, 2 * OH, 2 * OW, x) if shuffle else new_act(N, Cout, OH, OW, x)
        # GELU: the epilogue stores gelu'(v) (it has v in a register) instead of v, so the backward is one multiply inside
        # the gradient kernels' loaders — no erf/exp there and no elementwise dz pass
        deriv = save_pre and act == ACT_GELU
        # (Winograd: TRAINING forward passes only — its bits differ from the direct kernels', and inference / codec results must not move)
        wwino = wino_packed(w) if (need_grad and w2 is None and wk is w and wino_ok(N, H, W, w.shape[1], Cout, ks, stride)) else None
        wpk = halo_packed(w) if (wwino is None and w2 is None and halo_ok(N, H, W, w.shape[1], Cout, ks, stride) and wk is w) else None
        y = conv_raw(x, wk, b, ks=ks, stride=stride, act=act, res=res, res_scale=res_scale, res_first=res_first, y_pre=y_pre, shuffle=shuffle,
                     w2=wk2, bias2=b2, pre_deriv=deriv, out=out_buf, wx=wkx, batch_variant_ok=need_grad, wpk=wpk, wwino=wwino)
        if out_buf is not None:   # written in place into the caller's (strided) buffer: hand autograd a fresh alias of it
            y = out_buf.detach()
        ctx.cfg = (ks, stride, ACT_SAVED_DERIV if deriv else act, res_scale, shuffle, b is not None, res is not None, res_first)
        saved_act = y_pre if save_pre else (y if act in (ACT_LRELU, ACT_RELU, ACT_HALFTANH) else None)
        ctx.use_pre = save_pre
        ctx.gates = (gate_in, gate_out)
        if gate_out is not None and ACT_GATES and need_grad and act in (ACT_LRELU, ACT_RELU, ACT_GELU) and res is None:
            # what the consumer's data-gradient epilogue needs to apply this layer's activation derivative.  detach(): a plain alias
            # without grad_fn — the gate must not close a reference cycle y -> grad_fn -> ctx -> gate -> y (the activations would
            # then live until a garbage-collection pass instead of until the backward pass has used them)
            gate_out.saved, gate_out.act, gate_out.pre = saved_act.detach(), (ACT_SAVED_DERIV if deriv else act), save_pre
        ctx.bias_ref = b
        ctx.pair = (w2, b2)
        ctx.quad = (w3, b3, w4, b4)
        ctx.grad_slot = grad_slot   # (GradSlots, total channels, offset): where the data gradient is to be written
        ctx.folds = (fold_in, fold_out)   # GradFold: fold_in is added in this layer's data-gradient epilogue, fold_out receives d(res)
        ctx.save_for_backward(x, w, saved_act)
        return y

    @staticmethod
    def _wgrad(x, dz, w, bias_ref, has_b, need_w, need_b, ks, stride, pad, fw):
        """filter / bias gradient of one filter set: straight into the gradient arena when possible (returns None, None)."""
        Cout, Cin = w.shape[0], w.shape[1]
        gw, gb = _direct_grad(w), (_direct_grad(bias_ref) if has_b else None)
        if gw is not None and (gb is not None or not has_b) and to_kernel_weight(w) is w:
            # write straight into the persistent gradient arena (accumulate) — no temporary, no autograd add kernel
            if WGRAD_DEFER and (PROFILE is None or WGRAD_GROUP > 1):
                side = _S().wgrad_stream
                if WGRAD_GROUP > 1 or side is None:
                    wgrad_raw(x, dz, ks=ks, stride=stride, pad=pad, Cout=Cout, Cin=Cin, want_bias=has_b, dw_out=gw, db_out=gb, defer=True, **fw)
                else:
                    cur = torch.cuda.current_stream()
                    side.wait_stream(cur)
                    _S().keepalive.append((x, dz, fw.get("dys")))
                    with torch.cuda.stream(side):
                        wgrad_raw(x, dz, ks=ks, stride=stride, pad=pad, Cout=Cout, Cin=Cin, want_bias=has_b, dw_out=gw, db_out=gb, **fw)
            else:
                wgrad_raw(x, dz, ks=ks, stride=stride, pad=pad, Cout=Cout, Cin=Cin, want_bias=has_b, dw_out=gw, db_out=gb, **fw)
            return None, None
        dwf, db = wgrad_raw(x, dz, ks=ks, stride=stride, pad=pad, Cout=Cout, Cin=Cin, want_bias=need_b, **fw)
        return (_dw_to_param_layout(dwf, w) if need_w else None), db

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        ks, stride, act, res_scale, shuffle, has_b, has_res, res_first = ctx.cfg
        x, w, saved_act = ctx.saved_tensors
        w2, b2 = ctx.pair
        Cout, Cin = w.shape[0], w.shape[1]
        need_x, need_w, need_b, need_res = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2], has_res and ctx.needs_input_grad[3]
        # the activation derivative is applied inside the data-/weight-gradient kernels' loaders (no dz tensor, no extra
        # pass) unless the unshuffle copy or a pre-activation residual needs dz materialised
        # (only for the one-instruction derivatives; GELU arrives here as ACT_SAVED_DERIV: its derivative was stored forward)
        fold_in, fold_out = ctx.folds
        gate_in, gate_out = ctx.gates
        if gate_out is not None and gate_out.done:
            act = ACT_NONE          # the consumer's data-gradient kernel already multiplied dy by this layer's activation derivative
        # residual added BEFORE the activation (ResidualUnit): d(res) = dy * act'(out).  When the branch's first layer will fold it
        # into its data-gradient epilogue, park the RAW dy with the gate and skip the activation-backward pass altogether
        gate_park = (need_res and res_first and not shuffle and act in (ACT_LRELU, ACT_RELU) and fold_out is not None
                     and fold_out.gated and fold_out.can_park())
        fuse = act in (ACT_LRELU, ACT_RELU, ACT_SAVED_DERIV) and not shuffle and (not (need_res and res_first) or gate_park)
        # Large-map 3x3 layers whose derivative no consumer applied (ActGate): ONE elementwise pass dz = dy * act'(.) and then the
        # LDS-DMA data- and filter-gradient kernels beat the fused loaders of the register-staged ones (64 -> 64 @ 8x128x128:
        # 160 + 119 us fused vs 22 + 106 + 102 us)
        if fuse and MATERIALIZE_DZ and ks == 3 and act != ACT_NONE and not gate_park and dy.shape[0] * dy.shape[2] * dy.shape[3] >= MATERIALIZE_DZ and ctx.needs_input_grad[0]:
            fuse = False
        one_pass_unshuffle = shuffle and not (need_res and res_first)
        if one_pass_unshuffle:   # PixelShuffle(2) backward and the activation backward in ONE pass over dy
            dz = unshuffle_act_bwd(dy, saved_act if act != ACT_NONE else None, ctx.use_pre, act)
        else:
            dz = dy if (act == ACT_NONE or fuse) else act_bwd(dy, saved_act, ctx.use_pre, act)
        fa = dict(xs=saved_act, xs_act=act, xs_pre=ctx.use_pre) if fuse else {}
        dres = None
        if need_res and gate_park:
            fold_out.park(dy, res_scale, gate=(saved_act, act, ctx.use_pre))
        elif need_res:
            dsrc = dz if res_first else dy  # residual added before / after the activation
            if fold_out is not None and not shuffle and fold_out.park(dsrc, res_scale):
                dres = None          # the branch's first layer adds res_scale * dsrc in its gradient epilogue
            else:
                dres = dsrc if res_scale == 1.0 else dsrc * res_scale
        if shuffle and not one_pass_unshuffle:  # undo PixelShuffle(2): [N, C/4, 2H, 2W] -> [N, C, H, W]
            dz = torch.nn.functional.pixel_unshuffle(dz, 2).contiguous(memory_format=CL)
        dx = dw = db = dw2 = db2 = None
        w3, b3, w4, b4 = ctx.quad
        dwq = [None] * 4   # dw3, db3, dw4, db4
        pad = ks // 2
        if need_w or need_b:
            if w3 is not None:   # four filter sets: one problem per set, on its quarter of the batch
                q = x.shape[0] // 4
                res4 = []
                for k, (wk_, bk_) in enumerate(((w, ctx.bias_ref), (w2, b2), (w3, b3), (w4, b4))):
                    sl = slice(k * q, (k + 1) * q)
                    fwk = dict(dys=saved_act[sl], dys_act=act, dys_pre=ctx.use_pre) if fuse else {}
                    res4.append(_ConvFn._wgrad(x[sl], dz[sl], wk_, bk_, has_b, need_w, need_b, ks, stride, pad, fwk))
                (dw, db), (dw2, db2) = res4[0], res4[1]
                dwq = [res4[2][0], res4[2][1], res4[3][0], res4[3][1]]
            elif w2 is None:
                fw = dict(dys=saved_act, dys_act=act, dys_pre=ctx.use_pre) if fuse else {}
                dw, db = _ConvFn._wgrad(x, dz, w, ctx.bias_ref, has_b, need_w, need_b, ks, stride, pad, fw)
            else:   # one problem per filter set, on its half of the batch
                h = x.shape[0] // 2
                fw1 = dict(dys=saved_act[:h], dys_act=act, dys_pre=ctx.use_pre) if fuse else {}
                fw2 = dict(dys=saved_act[h:], dys_act=act, dys_pre=ctx.use_pre) if fuse else {}
                dw, db = _ConvFn._wgrad(x[:h], dz[:h], w, ctx.bias_ref, has_b, need_w, need_b, ks, stride, pad, fw1)
                dw2, db2 = _ConvFn._wgrad(x[h:], dz[h:], w2, b2, has_b, need_w, need_b, ks, stride, pad, fw2)
        if need_x:
            def wt_of(wp):
                wt = getattr(wp, "_clc_wt", None) if WT_CACHE_VALID else None   # refreshed once per step by the batched transpose (clc_amd.train)
                if wt is None:
                    wt = filter_transpose(to_kernel_weight(wp), Cout, ks * ks, Cin)
                return wt.view(Cin, -1)
            extra, extra_scale, gate = fold_in.take() if fold_in is not None else (None, 1.0, None)
            gs = ctx.grad_slot
            dx_out = gs[0].view(x, gs[1], gs[2], x.shape[1]) if gs is not None else None
            og = None
            if gate_in is not None and gate_in.saved is not None and extra is None and ctx.park_dx is None:
                # this layer is the only consumer of the producer's activated output: hand it d(pre-activation)
                og = (gate_in.saved, gate_in.act, gate_in.pre)
                gate_in.done = True
            wt1 = wt_of(w)
            # (data gradient of a 128 -> 128 layer: rows = Cin, K = 9 x Cout = 9 x 128)
            wwino = (wino_packed(w, wt1) if (w2 is None and not fa and wino_ok(dz.shape[0], dz.shape[2], dz.shape[3], Cout, Cin, ks, stride, transposed=True)) else None)
            wpk = (halo_packed(w, wt1) if (wwino is None and w2 is None and not fa and halo_ok(dz.shape[0], dz.shape[2], dz.shape[3], Cout, Cin, ks, stride)) else None)
            dx = conv_raw(dz, wt1, None, ks=ks, stride=stride, pad=pad, transposed=True, out_hw=(x.shape[2], x.shape[3]),
                          w2=(wt_of(w2) if w2 is not None else None), wx=(((wt_of(w3), None), (wt_of(w4), None)) if w3 is not None else None),
                          res=extra, res_scale=extra_scale, res_gate=gate, out=dx_out, out_gate=og, wpk=wpk, wwino=wwino, **fa)
        elif fold_in is not None:
            fold_in.consumed = True
        if dx is not None and ctx.park_dx is not None and ctx.park_dx.park(dx):
            dx = None
        return (dx, dw, db, dres, None, None, None, None, None, None, dw2, db2, None, None, None, None, None, None, None) + tuple(dwq)


def conv2d(x, w, b=None, *, stride=1, act=ACT_NONE, res=None, res_scale=1.0, shuffle=False, res_first=False, w2=None, b2=None,
           fold_in=None, fold_out=None, out=None, grad_slot=None, park_dx=None, gate_in=None, gate_out=None, wx=None):
    """out: optional destination (a pixel-major view, e.g. a channel slice of a wider buffer) written in place.
    (w2, b2) [+ wx = ((w3, b3), (w4, b4))]: 2 [4] filter sets on the halves [quarters] of the batch, one launch."""
    ks = w.shape[2] if w.dim() == 4 else 1
    (w3, b3), (w4, b4) = wx if wx is not None else ((None, None), (None, None))
    _note_grad_mode()
    return _ConvFn.apply(x, w, b, res, ks, stride, act, float(res_scale), bool(shuffle), bool(res_first), w2, b2, fold_in, fold_out, out,
                         grad_slot, park_dx, gate_in, gate_out, w3, b3, w4, b4)


# Inside Function.forward grad mode is always off and ctx.needs_input_grad only reflects the inputs' requires_grad flags — under
# torch.no_grad() with parameters that require grad (model.eval() inference, compress() / decompress()) it still says True.  The public
# wrappers therefore note the caller's grad mode right before .apply(); the forwards save activations for backward — and, for the
# convolutions, allow a batch-dependent summation order (clc_conv_desc.batch_variant_ok) — only when a graph is really being recorded.
import threading as _threading

_GRAD_MODE = _threading.local()


def _note_grad_mode():
    _GRAD_MODE.on = torch.is_grad_enabled()


def _recording(ctx) -> bool:
    return bool(getattr(_GRAD_MODE, "on", True)) and any(ctx.needs_input_grad)


# The [Cin][T][Cout] filter images clc_amd.train.FilterTransposer attaches to the parameters (`_clc_wt`) are refreshed at the START of an
# engine step; after that step's optimizer update they are one update behind.  They are therefore trusted only while the engine is
# running (or capturing) the forward/backward of a step — a plain autograd backward on the same model afterwards transposes on the fly.
WT_CACHE_VALID = False

FUSED_RU = int(os.environ.get("CLC_FUSED_RU", "1"))   # ResidualUnits on 16x16 maps with 128 channels: one launch forward, one for the data gradient


def residual_unit_fusable(x, sets) -> bool:
    return bool(FUSED_RU) and x.dim() == 4 and tuple(x.shape[1:]) == (128, 16, 16) and len(sets) in (1, 2, 4) and x.shape[0] % len(sets) == 0


def _ru_desc(x, xp, ldx, N, H, W, Cc, outs, filters, biases):
    d = _lib.RUDesc()
    d.x, d.ldx = xp, ldx
    d.t1, d.t2, d.y = (o.data_ptr() for o in outs)
    d.N, d.H, d.W, d.C, d.sets = N, H, W, Cc, len(filters)
    for k, (f1, f2, f3) in enumerate(filters):
        d.w1[k], d.w2[k], d.w3[k] = f1.data_ptr(), f2.data_ptr(), f3.data_ptr()
        if biases is not None:
            d.b1[k], d.b2[k], d.b3[k] = (bt.data_ptr() for bt in biases[k])
    return d


def residual_unit_fwd_raw(x, sets):
    """One clc_residual_unit_fwd launch.  sets: 1, 2 or 4 tuples (w1, b1, w2, b2, w3, b3) of nn.Conv2d parameters (equal parts of the
    batch each).  Returns the two intermediate activations and the output."""
    _require_gpu(x, "residual_unit")
    x, xp, N, H, W, Cc, ldx = nhwc(x)
    outs = (new_act(N, Cc // 2, H, W, x), new_act(N, Cc // 2, H, W, x), new_act(N, Cc, H, W, x))
    filters = [tuple(to_kernel_weight(wt) for wt in (st[0], st[2], st[4])) for st in sets]
    d = _ru_desc(x, xp, ldx, N, H, W, Cc, outs, filters, [(st[1], st[3], st[5]) for st in sets])
    _prof_hint(2.0 * N * H * W * (Cc * (Cc // 2) * 2 + 9 * (Cc // 2) * (Cc // 2)), f"residual unit fwd {N}x{H}x{W} C{Cc} sets{len(sets)}")
    _lib.check(_L().clc_residual_unit_fwd(C.byref(d), _stream()), "clc_residual_unit_fwd")
    return outs


def _wt_of(wp):
    """[Cin][taps][Cout] image of a filter parameter: refreshed once per step by the batched transpose (clc_amd.train), else made here."""
    wt = getattr(wp, "_clc_wt", None) if WT_CACHE_VALID else None
    if wt is None:
        ks = wp.shape[2] if wp.dim() == 4 else 1
        wt = filter_transpose(to_kernel_weight(wp), wp.shape[0], ks * ks, wp.shape[1])
    return wt.view(wp.shape[1], -1)


def residual_unit_dgrad_raw(dy, y, t2, t1, sets):
    """One clc_residual_unit_dgrad launch: (g2, g1, dx) = the pre-activation gradients of layers 2 and 1 and the unit's input gradient."""
    dy, dp, N, H, W, Cc, ldd = nhwc(dy)
    outs = (new_act(N, Cc // 2, H, W, dy), new_act(N, Cc // 2, H, W, dy), new_act(N, Cc, H, W, dy))
    filters = [(_wt_of(st[4]), _wt_of(st[2]), _wt_of(st[0])) for st in sets]   # the chain runs backwards: layer 3's filter first
    d = _ru_desc(dy, dp, ldd, N, H, W, Cc, outs, filters, None)
    for t in (y, t2, t1):
        assert t.is_contiguous(memory_format=CL)
    d.saved_y, d.saved_t2, d.saved_t1 = y.data_ptr(), t2.data_ptr(), t1.data_ptr()
    _prof_hint(2.0 * N * H * W * (Cc * (Cc // 2) * 2 + 9 * (Cc // 2) * (Cc // 2)), f"residual unit dgrad {N}x{H}x{W} C{Cc} sets{len(sets)}")
    _lib.check(_L().clc_residual_unit_dgrad(C.byref(d), _stream()), "clc_residual_unit_dgrad")
    return outs


class _ResidualUnitFn(Function):
    """y = relu(x + conv1x1(relu(conv3x3(relu(conv1x1 x))))) on the 16x16 latent maps: ONE launch forward, ONE launch for the whole data
    gradient (csrc/fused_ru.hip) and the three layers' ordinary (grouped, deferred) filter-gradient problems.  params: per filter set
    (w1, b1, w2, b2, w3, b3); the sets take equal parts of the batch."""

    @staticmethod
    def forward(ctx, x, nsets, *params):
        _own(ctx)
        sets = [params[6 * k: 6 * k + 6] for k in range(nsets)]
        t1, t2, y = residual_unit_fwd_raw(x, sets)
        ctx.nsets = nsets
        ctx.save_for_backward(x, t1, t2, y, *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        x, t1, t2, y, *params = ctx.saved_tensors
        nsets, need = ctx.nsets, ctx.needs_input_grad
        sets = [params[6 * k: 6 * k + 6] for k in range(nsets)]
        g2, g1, dx = residual_unit_dgrad_raw(dy, y, t2, t1, sets)
        per = x.shape[0] // nsets
        grads = []
        for k, (w1, b1, w2, b2, w3, b3) in enumerate(sets):
            sl, nb = slice(k * per, (k + 1) * per), 2 + 6 * k
            out = [None] * 6
            # layer 3: its dy operand is dy . [y > 0], applied in the filter-gradient kernel's loader; layers 2 / 1 get g2 / g1 as they are
            for j, (xin, dz, w, b, ks, fw) in enumerate(((x[sl], g1[sl], w1, b1, 1, {}), (t1[sl], g2[sl], w2, b2, 3, {}),
                                                         (t2[sl], dy[sl], w3, b3, 1, dict(dys=y[sl], dys_act=ACT_RELU, dys_pre=False)))):
                if need[nb + 2 * j] or need[nb + 2 * j + 1]:
                    out[2 * j], out[2 * j + 1] = _ConvFn._wgrad(xin, dz, w, b, True, need[nb + 2 * j], need[nb + 2 * j + 1], ks, 1, ks // 2, fw)
            grads += out
        return (dx if need[0] else None, None, *grads)


def residual_unit(x, units):
    """units: 1, 2 or 4 tuples (w1, b1, w2, b2, w3, b3) — see _ResidualUnitFn."""
    return _ResidualUnitFn.apply(x, len(units), *[t for u in units for t in u])


# ----------------------------------------------------------------------------- fused Swin-block MLP (csrc/fused_mlp.hip)

FUSED_MLP = int(os.environ.get("CLC_FUSED_MLP", "1"))              # 0: fc1 + GELU and fc2 as two clc_conv2d launches (A/B knob; same bits)
FUSED_MLP_MIN_PIX = int(os.environ.get("CLC_FUSED_MLP_MIN", "32768"))   # pixels from which the persistent fused kernel pays (one workgroup per CU)
FUSED_MLP_LN = int(os.environ.get("CLC_FUSED_MLP_LN", "1"))        # 1: the LayerNorm in front (Block.ln2) inside the fused kernels too (same bits; not with MLP_SAVE_H)
MLP_SAVE_H = int(os.environ.get("CLC_MLP_SAVE_H", "0"))            # training: 0 = nothing stored, the backward kernel recomputes fc1 from the LayerNorm output;
                                                                   # 1 = the forward pass stores fc1's pre-activation and the backward kernel reads it (same bits;
                                                                   #     111 vs 133 us per launch at 8x128x128, and the same step time: the 134 MB it writes cost as much)


def mlp_fusable(x, w1, w2, pair=None) -> bool:
    """Linear(64 -> 256) GELU Linear(256 -> 64) on a map with enough pixels to give every CU a workgroup (the ConvTransBlocks' Swin blocks
    on 128x128 / 64x64 maps at batch 8; the kernels produce the bits of the two-launch chain, so the choice may look at the batch)."""
    if not FUSED_MLP or pair is not None or x.dim() != 4 or x.shape[1] != 64 or tuple(w1.shape) != (256, 64) or tuple(w2.shape) != (64, 256):
        return False
    M = x.shape[0] * x.shape[2] * x.shape[3]
    return M >= FUSED_MLP_MIN_PIX and M % 32 == 0 and M < (1 << 24)


def mlp_ln_fusable(x, w1, w2, pair=None) -> bool:
    """mlp_fusable and the LayerNorm in front can ride along: the kernels then read the block's raw (dense) input."""
    if not (FUSED_MLP_LN and not MLP_SAVE_H and mlp_fusable(x, w1, w2, pair)):
        return False
    return nhwc(x)[6] == x.shape[1]


def mlp_fwd_raw(x, w1, b1, w2, b2, res=None, out=None, h_out=None, ln=None):
    """One clc_mlp_fwd launch: out = res + fc2(gelu(fc1(x))); h_out ([N,256,H,W], optional) receives fc1's pre-activation.
    ln = (gamma, beta): out = x + fc2(gelu(fc1(LN(x)))) from the raw x (res and h_out must be None)."""
    _require_gpu(x, "mlp")
    x, xp, N, H, W, Cin, ldx = nhwc(x)
    if out is None:
        out = new_act(N, w2.shape[0], H, W, x)
    o, op, *_r, ldy = nhwc(out)
    assert o is out, "mlp: output buffer must be pixel-major"
    d = _lib.MlpDesc()
    d.x, d.ldx, d.w1, d.w2 = xp, ldx, w1.data_ptr(), w2.data_ptr()
    d.b1 = b1.data_ptr() if b1 is not None else None
    d.b2 = b2.data_ptr() if b2 is not None else None
    keep = [x, out]
    if res is not None:
        r, rp, *_r, ldr = nhwc(res)
        d.res, d.ldr = rp, ldr
        keep.append(r)
    d.y, d.ldy, d.M, d.Cin, d.Chid, d.Cout = op, ldy, N * H * W, Cin, w1.shape[0], w2.shape[0]
    if h_out is not None:
        assert h_out.is_contiguous(memory_format=CL)
        d.h = h_out.data_ptr()
    if ln is not None:
        assert res is None and h_out is None
        d.ln_gamma, d.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
        if len(ln) > 2 and ln[2] is not None:   # training: LN(x) kept for the backward launch
            assert ln[2].is_contiguous(memory_format=CL)
            d.ln_out = ln[2].data_ptr()
    _prof_hint(2.0 * N * H * W * 2 * Cin * w1.shape[0], f"mlp fwd {Cin}->{w1.shape[0]}->{w2.shape[0]} {N}x{H}x{W}")
    _lib.check(_L().clc_mlp_fwd(C.byref(d), _stream()), "clc_mlp_fwd")
    return out


def mlp_bwd_raw(x, dy, w1, b1, w2t, h_saved=None, ln=None):
    """One clc_mlp_bwd launch -> (dx, dh, g): the block's input gradient, and the two [N,256,H,W] tensors its filter gradients contract
    (dh = d(fc1 pre-activation), g = gelu(fc1(x)) — fc1(x) read from h_saved, or recomputed from x when that is None).
    ln = (gamma, beta, LN(x) as the forward launch stored it): x is the raw input of `x + mlp(LN(x))`; -> (dx of the whole expression, dh, g,
    partial rows [nb][2][C] of dgamma / dbeta, nb)."""
    x, xp, N, H, W, Cin, ldx = nhwc(x)
    dy, dp, *_r, lddy = nhwc(dy)
    Ch = w1.shape[0]
    dx, dh, g = new_act(N, Cin, H, W, x), new_act(N, Ch, H, W, x), new_act(N, Ch, H, W, x)
    d = _lib.MlpDesc()
    d.x, d.ldx, d.w1, d.w2 = xp, ldx, w1.data_ptr(), w1.data_ptr()   # (w2 itself is not read backward: its transposed image is)
    d.b1 = b1.data_ptr() if b1 is not None else None
    d.M, d.Cin, d.Chid, d.Cout = N * H * W, Cin, Ch, dy.shape[1]
    d.dy, d.lddy, d.w2t, d.dx, d.lddx, d.dh, d.g = dp, lddy, w2t.data_ptr(), dx.data_ptr(), Cin, dh.data_ptr(), g.data_ptr()
    if h_saved is not None:
        d.h = h_saved.data_ptr()
    if ln is not None:
        assert h_saved is None and ldx == Cin and ln[2].is_contiguous(memory_format=CL)
        nb = _L().clc_mlp_blocks(N * H * W)
        ws = torch.empty(nb * 2 * Cin, device=x.device, dtype=torch.float32)
        d.ln_gamma, d.ln_beta, d.ln_out, d.ln_ws = ln[0].data_ptr(), ln[1].data_ptr(), ln[2].data_ptr(), ws.data_ptr()
    # algorithmic work = the two data gradients (the recomputed fc1 is this path's overhead, not counted)
    _prof_hint(2.0 * N * H * W * 2 * Cin * Ch, f"mlp dgrad {Cin}->{Ch}->{dy.shape[1]} {N}x{H}x{W}")
    _lib.check(_L().clc_mlp_bwd(C.byref(d), _stream()), "clc_mlp_bwd")
    if ln is not None:
        return dx, dh, g, ws, nb
    return dx, dh, g


class _MlpFn(Function):
    """y = res + fc2(gelu(fc1(x))): ONE launch forward (hidden tensor kept in registers), ONE launch for the data gradient with the hidden
    tensor recomputed from x, and the two layers' ordinary (grouped, deferred) filter-gradient problems.  The residual's gradient is parked
    on `fold_out` (the LayerNorm in front adds it in its backward kernel), as _ConvFn does for fc2."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, res, fold_out, out_buf, ln_g=None, ln_b=None):
        _own(ctx)
        ln = (ln_g, ln_b) if ln_g is not None else None
        hs = new_act(x.shape[0], w1.shape[0], x.shape[2], x.shape[3], x) if (MLP_SAVE_H and _recording(ctx) and ln is None) else None
        if ln is not None and _recording(ctx):
            hs = new_act(x.shape[0], x.shape[1], x.shape[2], x.shape[3], x)   # LN(x), stored by the forward launch
            ln = (ln_g, ln_b, hs)
        y = mlp_fwd_raw(x, w1, b1, w2, b2, res, out_buf, None if ln is not None else hs, ln)
        if out_buf is not None:
            y = out_buf.detach()
        ctx.fold_out, ctx.has_res = fold_out, res is not None
        ctx.has_b = (b1 is not None, b2 is not None)
        ctx.ln = ln[:2] if ln is not None else None
        ctx.save_for_backward(x, w1, b1, w2, b2, hs)
        return y

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        x, w1, b1, w2, b2, hs = ctx.saved_tensors
        need = ctx.needs_input_grad
        dres = dlg = dlb = None
        if ctx.has_res and need[5]:
            if not (ctx.fold_out is not None and ctx.fold_out.park(dy, 1.0)):
                dres = dy
        xw = x   # the x operand of fc1's filter gradient
        if ctx.ln is None:
            dx, dh, g = mlp_bwd_raw(x, dy, w1, b1, _wt_of(w2), hs)
        else:
            # the LayerNorm's backward pass (residual gradient folded in) happens inside the launch; its parameter gradients as _LayerNormFn's
            xw = hs
            dx, dh, g, ws, nb = mlp_bwd_raw(x, dy, w1, b1, _wt_of(w2), None, ctx.ln + (hs,))
            Cc = x.shape[1]
            gd, bd = _direct_grad(ctx.ln[0]), _direct_grad(ctx.ln[1])
            direct = gd is not None and bd is not None
            if direct and DEFER_REDUCTIONS:
                defer_reduce(ws, nb, 2 * Cc, gd, bd, Cc)
            else:
                tot = ws.view(nb, 2, Cc).sum(0)
                if direct:
                    gd.add_(tot[0])
                    bd.add_(tot[1])
                else:
                    dlg, dlb = tot[0], tot[1]
        dw1 = db1 = dw2 = db2 = None
        if need[1] or (ctx.has_b[0] and need[2]):
            dw1, db1 = _ConvFn._wgrad(xw, dh, w1, b1, ctx.has_b[0], need[1], ctx.has_b[0] and need[2], 1, 1, 0, {})
        if need[3] or (ctx.has_b[1] and need[4]):
            dw2, db2 = _ConvFn._wgrad(g, dy, w2, b2, ctx.has_b[1], need[3], ctx.has_b[1] and need[4], 1, 1, 0, {})
        return (dx if need[0] else None), dw1, db1, dw2, db2, dres, None, None, dlg, dlb


def mlp(x, w1, b1, w2, b2, *, res=None, fold_out=None, out=None):
    """res + fc2(gelu(fc1(x))) of a Swin block (x = the LayerNorm's output), fused (see mlp_fusable)."""
    _note_grad_mode()
    return _MlpFn.apply(x, w1, b1, w2, b2, res, fold_out, out)


def mlp_ln(x, ln_g, ln_b, w1, b1, w2, b2, *, out=None):
    """x + fc2(gelu(fc1(LN(x)))) of a Swin block from its raw input, LayerNorm included (see mlp_ln_fusable)."""
    _note_grad_mode()
    return _MlpFn.apply(x, w1, b1, w2, b2, None, None, out, ln_g, ln_b)


# ------------------------------------------------------------------ LayerNorm + Linear: ln1 and the attention's embedding (csrc/fused_mlp.hip)

FUSED_GDN_BWD = int(os.environ.get("CLC_FUSED_GDN_BWD", "1"))   # 0: clc_gdn_bwd_elem + a transposed 1x1 clc_conv2d (A/B knob; same bits)
FUSED_LNLIN = int(os.environ.get("CLC_FUSED_LNLIN", "1"))   # 0: clc_layernorm_fwd + a 1x1 clc_conv2d launch (A/B knob; same bits)


def lnlin_fusable(x, w, pair=None) -> bool:
    """nn.LayerNorm(64) + nn.Linear(64 -> 192) on a map with enough pixels for the persistent kernels (as mlp_fusable)."""
    if not FUSED_LNLIN or pair is not None or x.dim() != 4 or x.shape[1] != 64 or tuple(w.shape) != (192, 64):
        return False
    M = x.shape[0] * x.shape[2] * x.shape[3]
    return M >= FUSED_MLP_MIN_PIX and M % 32 == 0 and M < (1 << 24)


class _LnLinearFn(Function):
    """w . LN(x) + b in ONE launch; backward: the Linear's data gradient, the LayerNorm's backward pass and the block's residual gradient
    (fold_in) in one more, written into `grad_slot` like _LayerNormFn; the filter gradient stays a grouped, deferred problem on the LN(x)
    the forward launch stored."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, w, b, fold_in, grad_slot):
        _own(ctx)
        _require_gpu(x, "ln_linear")
        x, xp, N, H, W, Cc, ldx = nhwc(x)
        y = new_act(N, w.shape[0], H, W, x)
        ln_out = new_act(N, Cc, H, W, x) if _recording(ctx) else None
        d = _lib.LnLinDesc()
        d.x, d.ldx, d.ln_gamma, d.ln_beta, d.w = xp, ldx, ln_g.data_ptr(), ln_b.data_ptr(), w.data_ptr()
        d.b = b.data_ptr() if b is not None else None
        d.y, d.ln_out = y.data_ptr(), (ln_out.data_ptr() if ln_out is not None else None)
        d.M, d.Cin, d.Cout = N * H * W, Cc, w.shape[0]
        _prof_hint(2.0 * N * H * W * Cc * w.shape[0], f"ln+linear fwd {Cc}->{w.shape[0]} {N}x{H}x{W}")
        _lib.check(_L().clc_lnlin_fwd(C.byref(d), _stream()), "clc_lnlin_fwd")
        ctx.fold_in, ctx.grad_slot, ctx.has_b = fold_in, grad_slot, b is not None
        ctx.save_for_backward(x, ln_g, ln_b, w, b, ln_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        x, ln_g, ln_b, w, b, ln_out = ctx.saved_tensors
        need = ctx.needs_input_grad
        x, xp, N, H, W, Cc, ldx = nhwc(x)
        dy, dyp, *_r, lddy = nhwc(dy)
        if lddy != dy.shape[1]:
            dy = dy.contiguous(memory_format=CL)
            dyp = dy.data_ptr()
        gs = ctx.grad_slot
        dx = gs[0].view(x, gs[1], gs[2], Cc) if gs is not None else new_act(N, Cc, H, W, x)
        extra, extra_scale, gate = ctx.fold_in.take() if ctx.fold_in is not None else (None, 1.0, None)
        assert gate is None, "gated residual gradients are folded by conv data-gradient kernels only"
        ep, lde = None, 0
        if extra is not None:
            if extra_scale != 1.0:
                extra = extra * extra_scale
            extra, ep, *_q, lde = nhwc(extra)
        M = N * H * W
        nb = _L().clc_mlp_blocks(M)
        ws = torch.empty(nb * 2 * Cc, device=x.device, dtype=torch.float32)
        d = _lib.LnLinDesc()
        d.x, d.ldx, d.ln_gamma, d.ln_beta = xp, ldx, ln_g.data_ptr(), ln_b.data_ptr()
        d.M, d.Cin, d.Cout = M, Cc, w.shape[0]
        d.dy, d.wt, d.dx, d.lddx, d.dadd, d.ldadd, d.ln_ws = dyp, _wt_of(w).data_ptr(), dx.data_ptr(), nhwc(dx)[6], ep, lde, ws.data_ptr()
        _prof_hint(2.0 * M * Cc * w.shape[0], f"ln+linear dgrad {w.shape[0]}->{Cc} {N}x{H}x{W}")
        _lib.check(_L().clc_lnlin_bwd(C.byref(d), _stream()), "clc_lnlin_bwd")
        dlg = dlb = None
        gd, bd = _direct_grad(ln_g), _direct_grad(ln_b)
        direct = gd is not None and bd is not None
        if direct and DEFER_REDUCTIONS:
            defer_reduce(ws, nb, 2 * Cc, gd, bd, Cc)
        else:
            tot = ws.view(nb, 2, Cc).sum(0)
            if direct:
                gd.add_(tot[0])
                bd.add_(tot[1])
            else:
                dlg, dlb = tot[0], tot[1]
        dw = db = None
        if need[3] or (ctx.has_b and need[4]):
            dw, db = _ConvFn._wgrad(ln_out, dy, w, b, ctx.has_b, need[3], ctx.has_b and need[4], 1, 1, 0, {})
        return (dx if need[0] else None), dlg, dlb, dw, db, None, None


def ln_linear(x, ln_g, ln_b, w, b, *, fold_in=None, grad_slot=None):
    """linear(LN(x)) of a Swin block's attention branch from the block's raw input (see lnlin_fusable)."""
    _note_grad_mode()
    return _LnLinearFn.apply(x, ln_g, ln_b, w, b, fold_in, grad_slot)


def linear(x, w, b=None, *, act=ACT_NONE, res=None, w2=None, b2=None, fold_in=None, fold_out=None, out=None, gate_in=None, gate_out=None):
    """nn.Linear on channels (tokens are pixels): x [N,Cin,H,W] pixel-major, w [Cout,Cin]."""
    _note_grad_mode()
    return _ConvFn.apply(x, w, b, res, 1, 1, act, 1.0, False, False, w2, b2, fold_in, fold_out, out, None, None, gate_in, gate_out)


class _CatHalvesFn(Function):
    """cat((a, b), dim=1) where a and b were WRITTEN as the two channel ranges of `buf` by their producers (conv2d(out=...)):
    no copy forward; backward hands each producer its channel range of the gradient as a strided view."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.ca = a.shape[1]
        assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf.data_ptr() + 4 * ctx.ca and a.shape[1] + b.shape[1] == buf.shape[1]
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        return g[:, : ctx.ca], g[:, ctx.ca:], None


class _FanOutFn(Function):
    """x -> n aliases of x, one per consumer; the backward sums the consumers' gradients in ONE launch (clc_sum_n, fixed order) instead of
    autograd's chain of pairwise adds (n - 1 launches).  For tensors with many consumers in the slice loop: the reference features
    (every slice's cc and lrp nets), the attention blocks' input."""

    @staticmethod
    def forward(ctx, x, n):
        _own(ctx)
        ctx.meta = tuple(x.shape)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        _reown(ctx)
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        N, Cc, H, W = ctx.meta
        views = [nhwc(g) for g in gs]
        if Cc % 4 or len(gs) > 12 or any(v[1] % 16 or v[6] % 4 for v in views):
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        out = new_act(N, Cc, H, W, gs[0])
        ptrs = _lib.ptr_array([v[1] for v in views])
        lds = (C.c_int * len(views))(*[v[6] for v in views])
        _lib.check(_L().clc_sum_n(ptrs, lds, len(views), out.data_ptr(), Cc, N * H * W, Cc, _stream()), "clc_sum_n")
        return out, None


FANOUT = int(os.environ.get("CLC_FANOUT", "1"))   # 0: leave multi-consumer gradients to autograd's pairwise accumulation (A/B knob)


def fanout(x, n):
    """n aliases of x for n consumers (see _FanOutFn); a tensor that needs no gradient is handed out as it is"""
    if n <= 1 or not FANOUT or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(n, 1)
    return _FanOutFn.apply(x, n)


class _CatBatchFn(Function):
    """cat((a, b), dim=0) where a and b were WRITTEN as the two batch halves of `buf` by their producers (out=): no copy forward;
    backward hands each producer its half of the gradient as a view."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.h = a.shape[0]
        assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf.data_ptr() + a.numel() * a.element_size() and a.shape[0] + b.shape[0] == buf.shape[0]
        return buf.detach()

    @staticmethod
    def backward(ctx, g):
        return g[: ctx.h], g[ctx.h:], None


def cat_batch(a, b, buf):
    return _CatBatchFn.apply(a, b, buf)


def cat_halves(a, b, buf):
    return _CatHalvesFn.apply(a, b, buf)


class _CutFn(Function):
    """Identity marking a place where a backward pass may be cut in two (clc_amd.train.TrainEngine's two-phase backward).
    torch.autograd.backward(..., inputs=[t]) EXECUTES t.grad_fn once more than it needs to (the engine marks the producer of a
    captured non-leaf tensor as needed) and a Python Function cannot see that its results are unwanted (ctx.needs_input_grad is
    static) — a convolution there would write its filter gradient twice.  With this no-op as the producer, nothing is."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


def cut(x):
    return _CutFn.apply(x) if x is not None and x.requires_grad else x


class _FlushPointFn(Function):
    """Identity whose backward launches the filter gradients queued so far (the synthesis transform's, when placed at its input):
    on the side stream (CLC_WGRAD_STREAM=1) these MFMA-bound grids then run beside the latency-bound backward of the slice loop."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        flush_wgrads()
        return g


def flush_point(x):
    return _FlushPointFn.apply(x) if (WGRAD_DEFER and x.is_cuda and _S().wgrad_stream is not None and x.requires_grad) else x


# ----------------------------------------------------------------------------------- split / chunk


class _SplitBatchFn(Function):
    """x [2B, ...] -> (x[:B], x[B:]) as views; the backward is ONE concatenation (autograd's own slice backward would
    zero-fill two full-size tensors and add them)."""

    @staticmethod
    def forward(ctx, x):
        h = x.shape[0] // 2
        ctx.meta = (x.shape, x.dtype, x.device)
        return x[:h], x[h:]

    @staticmethod
    def backward_pad(ga):
        """[ga; 0]"""
        return torch.cat((ga, torch.zeros_like(ga)), dim=0)

    @staticmethod
    def backward(ctx, ga, gb):
        shape, dtype, device = ctx.meta
        h = shape[0] // 2
        if ga is not None and gb is not None:
            # the two halves of ONE buffer, written in place by their producer (_GaussLikFn.backward): that buffer IS the concatenation
            base = ga._base
            if (base is not None and base is gb._base and tuple(base.shape) == tuple(shape) and base.is_contiguous(memory_format=CL)
                    and ga.data_ptr() == base.data_ptr() and gb.data_ptr() == base.data_ptr() + h * base.stride(0) * base.element_size()
                    and ga.stride() == base.stride() and gb.stride() == base.stride()):
                return base
        if ga is None:
            ga = torch.zeros((h,) + tuple(shape[1:]), dtype=dtype, device=device).contiguous(memory_format=CL)
        if gb is None:
            gb = torch.zeros((shape[0] - h,) + tuple(shape[1:]), dtype=dtype, device=device).contiguous(memory_format=CL)
        return torch.cat((ga, gb), dim=0)


def split_batch(x):
    return _SplitBatchFn.apply(x)


class _WholeAndFirstHalfFn(Function):
    """x [2B, ...] -> (x, x[:B]) for a tensor that is consumed whole by one layer and by its first half by another (the stacked
    output of the paired attention blocks: both halves feed the paired cc transforms, the mean half also the lrp transform,
    CLC_run.py:560-581).  Backward: the half's gradient is added INTO the whole's gradient (this node is its only holder): one
    half-size add, where a split + autograd's accumulation would zero-fill, concatenate and add full-size tensors."""

    @staticmethod
    def forward(ctx, x):
        h = x.shape[0] // 2
        return x.view_as(x), x[:h]

    @staticmethod
    def backward(ctx, g_whole, g_half):
        if g_whole is None:
            return _SplitBatchFn.backward_pad(g_half)
        if g_half is not None:
            h = g_whole.shape[0] // 2
            b = g_whole._base
            # in place only into memory this node provably holds alone: a fresh tensor, or the channel range torch.cat's backward
            # narrows out of its gradient for this input (the other ranges go to the other inputs).  Any other view — a GradSlots
            # range, a batch slice, a reshaped alias — may have further holders: add out of place there.
            cat_range = (b is not None and b.dim() == 4 and g_whole.dim() == 4 and g_whole.stride() == b.stride()
                         and g_whole.shape[0] == b.shape[0] and g_whole.shape[2:] == b.shape[2:])
            if not (b is None or cat_range):
                g_whole = g_whole.clone(memory_format=CL)
            g_whole[:h] += g_half
        return g_whole


def whole_and_first_half(x):
    return _WholeAndFirstHalfFn.apply(x)



class _SplitFn(Function):
    """torch.split along channels returning strided views, with a backward that writes the incoming slice gradients
    straight into ONE buffer (one strided-copy kernel per slice) — autograd's own SliceBackward would zero-fill a full-size
    tensor per slice and then add them all up (2n+1 full passes instead of n partial ones)."""

    @staticmethod
    def forward(ctx, x, slots, *sizes):
        _own(ctx)
        ctx.sizes = sizes
        ctx.slots = slots
        ctx.shape = x.shape
        outs, o = [], 0
        for sz in sizes:
            outs.append(x[:, o:o + sz])
            o += sz
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        _reown(ctx)
        N, Cc, H, W = ctx.shape
        ref = next(g for g in grads if g is not None)
        buf = ctx.slots.buf if ctx.slots is not None else None
        if buf is not None and tuple(buf.shape) == tuple(ctx.shape):   # every consumer wrote its range of the shared buffer
            o, ok = 0, True
            for sz, g in zip(ctx.sizes, grads):
                ok = ok and g is not None and g.data_ptr() == buf.data_ptr() + 4 * o and g.stride() == buf.stride() and g.shape[1] == sz
                o += sz
            if ok:
                return (buf, None) + (None,) * len(ctx.sizes)
        dx = torch.empty((N, Cc, H, W), device=ref.device, dtype=torch.float32, memory_format=CL)
        rows, o = N * H * W, 0
        for sz, g in zip(ctx.sizes, grads):
            dst = dx.data_ptr() + 4 * o
            if g is None:
                dx[:, o:o + sz].zero_()
            else:
                g, gp, *_r, ldg = nhwc(g)
                _lib.check(_L().clc_copy2d(gp, ldg, dst, Cc, rows, sz, _stream()), "clc_copy2d")
            o += sz
        return (dx, None) + (None,) * len(ctx.sizes)


class SliceSupport:
    """The input of the slice loop's parameter nets without the per-slice concatenations (CLC_run.py:541-566).

    Slice i's mean- / scale-parameter nets read cat([latent_means | latent_scales] + y_hat_slices[:min(i, 5)]): the same leading
    channels every time.  ONE buffer P [2B, C0 + n * S] holds them — rows [0, B) the mean net's input, rows [B, 2B) the scale
    net's; the stacked latents are copied in once, support slice j once (into both halves) — and slice i's nets read the channel
    prefix P[:, :C0 + S * min(i, n)] through its leading dimension: 3 concatenation launches per slice become 1 copy per
    support slice.  Backward mirrors it: every slice's first convolution adds its data gradient onto the same prefix of ONE
    gradient buffer G in its epilogue (GradFold + GradSlots), so the ~100 gradient-accumulation adds autograd would launch for
    the many consumers of the latents and of each support slice collapse to one add per support slice (its two halves).
    Support slice j's gradient is complete once slice j + 1 — its earliest consumer, the last to run backward — has added
    its share; that node hands it to autograd, and slice 0's node hands over the latents' gradient.
    Requires n_support >= num_slices - 1 (the reference's max_support_slices = num_slices): only then does slice i + 1 read slice
    i's result, so autograd's dependencies force the last-to-first backward order the shared buffer relies on — the caller
    (models/clc.py) falls back to per-slice concatenations otherwise."""

    def __init__(self, both, n_support, S):
        self.C0, self.S, self.n = both.shape[1], int(S), int(n_support)
        self.both = both
        self.P = new_act(both.shape[0], self.C0 + self.n * self.S, both.shape[2], both.shape[3], both)
        # the buffer is filled THROUGH an alias with its own version counter: the prefixes handed out by take() are saved for backward by
        # their consumers, and the later slices' copies (into channels those prefixes do not cover) must not invalidate them
        self.Pw = torch.empty(0, device=both.device, dtype=both.dtype).set_(self.P.untyped_storage(), self.P.storage_offset(), self.P.size(), self.P.stride())
        self.Pw[:, :self.C0].copy_(both.detach())
        self.grad = both.requires_grad and torch.is_grad_enabled()
        self.slots = GradSlots()
        if self.grad:
            self.slots.buf = new_act(*self.P.shape, both)

    def width(self, i):
        return self.C0 + self.S * min(i, self.n)

    def add(self, j, y_hat):
        """support slice j (j < n) -> both halves of the buffer, one launch"""
        N2, Ct, H, W = self.P.shape
        self.Pw.view(2, N2 // 2, Ct, H, W)[:, :, self.C0 + j * self.S:self.C0 + (j + 1) * self.S].copy_(y_hat.detach())

    def take(self, i, supports, last):
        """-> (input of slice i's nets, fold_in, grad_slot) for their first convolution.  last: slice i is the final slice (its
        backward runs first: it initialises the gradient buffer instead of adding to it)."""
        assert len(supports) == min(i, self.n)
        x = _SupportTakeFn.apply(self, i, self.both, *supports)
        if not self.grad:
            return x, None, None
        fold = None
        if not last:
            fold = GradFold()
            fold.park(self.slots.buf[:, :self.width(i)])
        return x, fold, (self.slots, self.P.shape[1], 0)


class _SupportTakeFn(Function):
    @staticmethod
    def forward(ctx, sup, i, both, *supports):
        ctx.sup, ctx.i, ctx.k = sup, i, len(supports)
        return sup.P[:, :sup.width(i)].detach()

    @staticmethod
    def backward(ctx, dx):
        sup, i = ctx.sup, ctx.i
        G = sup.slots.buf   # (dx is its prefix: the first convolution of the nets wrote / accumulated in place)
        B = G.shape[0] // 2
        grads = [None] * ctx.k
        if 1 <= i <= sup.n:   # slice i is the earliest consumer of support slice i - 1: everything later has already added its share
            c0 = sup.C0 + (i - 1) * sup.S
            grads[i - 1] = G[:B, c0:c0 + sup.S] + G[B:, c0:c0 + sup.S]
        g_both = G[:, :sup.C0] if i == 0 else None
        return (None, None, g_both) + tuple(grads)


class _GatherChannelsFn(Function):
    """The channel concatenation of tensors that were WRITTEN into consecutive channel ranges of one buffer (out=): returns the
    buffer, and the backward hands each producer its channel range of the incoming gradient as a view — no copy either way."""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.sizes = [p.shape[1] for p in parts]
        return buf.detach()

    @staticmethod
    def backward(ctx, dy):
        outs, o = [], 0
        for sz in ctx.sizes:
            outs.append(dy[:, o:o + sz])
            o += sz
        return (None,) + tuple(outs)


def gather_channels(buf, parts):
    return _GatherChannelsFn.apply(buf, *parts)


def split_channels(x, sizes, slots=None):
    """Channel split as zero-copy views (kernels read them through their leading dimension).  slots: a GradSlots the
    consumers write their gradients into (see there)."""
    x, *_ = nhwc(x)
    return _SplitFn.apply(x, slots, *[int(s) for s in sizes])


# ------------------------------------------------------------------------------------------- GDN


class _GDNFn(Function):
    """y = x * rsqrt(beta + gamma . x^2)  (inverse: * sqrt), fused: 1x1 conv with square-on-load + norm epilogue.
    Optional residual ``res`` is added in the same epilogue (ResidualBlockWithStride / Upsample tail)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, inverse):
        _own(ctx)
        N, Cc, H, W = x.shape
        need_grad = _recording(ctx)
        v = new_act(N, Cc, H, W, x) if need_grad else None
        g = gamma if gamma.is_contiguous() else gamma.contiguous()
        y = conv_raw(x, g, beta, ks=1, in_op=IN_SQUARE, norm=NORM_IGDN if inverse else NORM_GDN, mul=x, y_pre=v, res=res)
        ctx.inverse = inverse
        ctx.has_res = res is not None
        ctx.save_for_backward(x, g, v)
        return y

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        x, g, v = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dy, xx = dense(dy), dense(x)
        n = N * Cc * H * W
        dxd, dv = new_act(N, Cc, H, W, x), new_act(N, Cc, H, W, x)
        _lib.check(_L().clc_gdn_bwd_elem(dy.data_ptr(), xx.data_ptr(), v.data_ptr(), dxd.data_ptr(), dv.data_ptr(), n, int(ctx.inverse), _stream()), "clc_gdn_bwd_elem")
        dgamma = dbeta = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dgf, dbeta = wgrad_raw(xx, dv, ks=1, stride=1, pad=0, Cout=Cc, Cin=Cc, want_bias=True, in_op=IN_SQUARE)
            dgamma = dgf.view(Cc, Cc)
        if ctx.needs_input_grad[0]:
            gt = filter_transpose(g, Cc, 1, Cc)
            t = conv_raw(dv, gt, None, ks=1, transposed=True, out_hw=(H, W))
            dx = new_act(N, Cc, H, W, x)
            _lib.check(_L().clc_gdn_bwd_combine(dxd.data_ptr(), xx.data_ptr(), t.data_ptr(), dx.data_ptr(), n, _stream()), "clc_gdn_bwd_combine")
        return dx, dgamma, dbeta, (dy if ctx.has_res and ctx.needs_input_grad[3] else None), None


def gdn(x, gamma_eff, beta_eff, inverse=False, res=None):
    _note_grad_mode()
    return _GDNFn.apply(x, gamma_eff, beta_eff, res, bool(inverse))


class _GDNParamFn(Function):
    """GDN on the RAW parameters: the NonNegativeParametrizer of gamma and beta is one launch forward (which also emits
    gamma_eff transposed for the data-gradient conv) and one launch backward, writing straight into the gradient arena —
    instead of ~20 parameter-sized torch launches per GDN module and step.
    (gamma2, beta2): paired modules — the second half of the batch is normalised with the second module's parameters."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, inverse, gamma_bound, beta_bound, pedestal, gamma2=None, beta2=None):
        _own(ctx)
        N, Cc, H, W = x.shape
        need_grad = _recording(ctx)
        sets = [(gamma, beta)] + ([(gamma2, beta2)] if gamma2 is not None else [])
        eff = []
        for g, b in sets:
            cached = getattr(g, "_clc_gdn_eff", None) if WT_CACHE_VALID else None   # (clc_amd.train.GDNReparamCache: all modules in one launch per step)
            if cached is not None and g.is_contiguous():
                eff.append((g,) + tuple(cached))
                continue
            g_eff = torch.empty((Cc, Cc), device=x.device, dtype=torch.float32)
            g_eff_t = torch.empty((Cc, Cc), device=x.device, dtype=torch.float32)
            b_eff = torch.empty((Cc,), device=x.device, dtype=torch.float32)
            gm = g if g.is_contiguous() else g.contiguous()
            _lib.check(_L().clc_gdn_reparam_fwd(gm.data_ptr(), b.data_ptr(), Cc, gamma_bound, beta_bound, pedestal, g_eff.data_ptr(),
                                                g_eff_t.data_ptr(), b_eff.data_ptr(), _stream()), "clc_gdn_reparam_fwd")
            eff.append((gm, g_eff, g_eff_t, b_eff))
        v = new_act(N, Cc, H, W, x) if need_grad else None
        w2 = eff[1][1] if len(eff) == 2 else None
        y = conv_raw(x, eff[0][1], eff[0][3], ks=1, in_op=IN_SQUARE, norm=NORM_IGDN if inverse else NORM_GDN, mul=x, y_pre=v, res=res,
                     w2=w2, bias2=(eff[1][3] if len(eff) == 2 else None))
        ctx.cfg = (inverse, res is not None, gamma_bound, beta_bound, len(sets))
        ctx.params = sets
        ctx.save_for_backward(x, v, *[t for e in eff for t in (e[0], e[2])])
        return y

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        x, v, *rest = ctx.saved_tensors
        inverse, has_res, gamma_bound, beta_bound, nset = ctx.cfg
        gms, gts = rest[0::2], rest[1::2]
        N, Cc, H, W = x.shape
        dy, xx = dense(dy), dense(x)
        n = N * Cc * H * W
        # large 128-channel maps: the elementwise part, the gamma^T product and the combination in ONE launch (csrc/fused_mlp.hip: gdn_bwd_kernel; same bits)
        fused = FUSED_GDN_BWD and nset == 1 and Cc == 128 and N * H * W >= max(FUSED_MLP_MIN_PIX, 32768) and (N * H * W) % 32 == 0   # (clc_gdn_bwd_fused builds M >= 32 768 only, whatever CLC_FUSED_MLP_MIN says)
        fused = fused and n * 4 < (1 << 31)
        dx = None
        if fused:
            dv, dx = new_act(N, Cc, H, W, x), new_act(N, Cc, H, W, x)
            _prof_hint(2.0 * N * H * W * Cc * Cc, f"gdn dgrad {Cc} {N}x{H}x{W}")
            _lib.check(_L().clc_gdn_bwd_fused(dy.data_ptr(), xx.data_ptr(), v.data_ptr(), gts[0].data_ptr(), dv.data_ptr(), dx.data_ptr(), N * H * W, Cc,
                                               int(inverse), _stream()), "clc_gdn_bwd_fused")
        else:
            dxd, dv = new_act(N, Cc, H, W, x), new_act(N, Cc, H, W, x)
            _lib.check(_L().clc_gdn_bwd_elem(dy.data_ptr(), xx.data_ptr(), v.data_ptr(), dxd.data_ptr(), dv.data_ptr(), n, int(inverse), _stream()), "clc_gdn_bwd_elem")
        out_grads = [None, None] * 2
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            hb = N // nset
            for k, ((gamma, beta), gm) in enumerate(zip(ctx.params, gms)):
                xk, dvk = (xx, dv) if nset == 1 else (xx[k * hb:(k + 1) * hb], dv[k * hb:(k + 1) * hb])   # this module's half of the batch
                gg, gb = _direct_grad(gamma), _direct_grad(beta)
                direct = gg is not None and gb is not None and gm is gamma
                if direct and WGRAD_DEFER:
                    # d(gamma_eff) joins the next grouped filter-gradient launch; the re-parametrisation's backward runs right after it
                    dgf = torch.empty(Cc * Cc, device=x.device, dtype=torch.float32)
                    dbe = torch.empty(Cc, device=x.device, dtype=torch.float32)
                    wgrad_raw(xk, dvk, ks=1, stride=1, pad=0, Cout=Cc, Cin=Cc, want_bias=True, in_op=IN_SQUARE, dw_out=dgf, db_out=dbe,
                              accumulate=False, defer=True)

                    def post(gm=gm, beta=beta, dgf=dgf, dbe=dbe, gg=gg, gb=gb):
                        _lib.check(_L().clc_gdn_reparam_bwd(gm.data_ptr(), beta.data_ptr(), Cc, gamma_bound, beta_bound, dgf.data_ptr(),
                                                            dbe.data_ptr(), gg.data_ptr(), gb.data_ptr(), 1, _stream()), "clc_gdn_reparam_bwd")
                    _S().pending_post.setdefault(_S().last_defer_vid, []).append(post)
                    _S().keepalive.append((gm, beta, dgf, dbe))
                else:
                    dgf, dbe = wgrad_raw(xk, dvk, ks=1, stride=1, pad=0, Cout=Cc, Cin=Cc, want_bias=True, in_op=IN_SQUARE)
                    if not direct:
                        gg, gb = torch.empty_like(gm), torch.empty_like(beta)
                    _lib.check(_L().clc_gdn_reparam_bwd(gm.data_ptr(), beta.data_ptr(), Cc, gamma_bound, beta_bound, dgf.data_ptr(), dbe.data_ptr(),
                                                        gg.data_ptr(), gb.data_ptr(), int(direct), _stream()), "clc_gdn_reparam_bwd")
                    if not direct:
                        out_grads[2 * k], out_grads[2 * k + 1] = gg, gb
        if ctx.needs_input_grad[0] and not fused:
            # dx = dx_direct + 2 x (gamma^T dv): the 2x factor and the add ride in the 1x1 data-gradient conv's epilogue
            dx = conv_raw(dv, gts[0], None, ks=1, transposed=True, out_hw=(H, W), norm=NORM_MUL2, mul=xx, res=dxd,
                          w2=(gts[1] if nset == 2 else None))
        return (dx, out_grads[0], out_grads[1], (dy if has_res and ctx.needs_input_grad[3] else None), None, None, None, None,
                out_grads[2], out_grads[3])


def gdn_param(x, gamma, beta, gamma_bound, beta_bound, pedestal, inverse=False, res=None, gamma2=None, beta2=None):
    _note_grad_mode()
    return _GDNParamFn.apply(x, gamma, beta, res, bool(inverse), float(gamma_bound), float(beta_bound), float(pedestal), gamma2, beta2)


# ------------------------------------------------------------------------------------- LayerNorm


class _LayerNormFn(Function):
    """nn.LayerNorm over channels.  (gamma2, beta2): paired modules — the second half of the batch uses them."""

    @staticmethod
    def forward(ctx, x, gamma, beta, fold_in=None, grad_slot=None, gamma2=None, beta2=None):
        _own(ctx)
        ctx.grad_slot = grad_slot
        x, xp, N, H, W, Cc, ldx = nhwc(x)
        rows = N * H * W
        y = new_act(N, Cc, H, W, x)
        ctx.fold_in = fold_in
        need = _recording(ctx)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32) if need else None
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if need else None
        mp, rp = (mean.data_ptr(), rstd.data_ptr()) if need else (None, None)
        if gamma2 is None:
            _lib.check(_L().clc_layernorm_fwd(xp, ldx, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), Cc, mp, rp, rows, Cc, _stream()), "clc_layernorm_fwd")
        else:
            assert N % 2 == 0
            _lib.check(_L().clc_layernorm_fwd_pair(xp, ldx, gamma.data_ptr(), beta.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(), rows // 2,
                                                   y.data_ptr(), Cc, mp, rp, rows, Cc, _stream()), "clc_layernorm_fwd_pair")
        ctx.refs = (beta, gamma2, beta2)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        _reown(ctx)
        x, gamma, mean, rstd = ctx.saved_tensors
        beta, gamma2, beta2 = ctx.refs
        paired = gamma2 is not None
        x, xp, N, H, W, Cc, ldx = nhwc(x)
        dy, dyp, *_r, lddy = nhwc(dy)
        rows = N * H * W
        gs = ctx.grad_slot
        dx = gs[0].view(x, gs[1], gs[2], Cc) if gs is not None else new_act(N, Cc, H, W, x)
        lddx = nhwc(dx)[6]
        sets = [(gamma, beta)] + ([(gamma2, beta2)] if paired else [])
        grads = [(_direct_grad(g), _direct_grad(b)) for g, b in sets]
        direct = all(g is not None and b is not None for g, b in grads)
        if not direct:
            grads = [(torch.empty_like(g), torch.empty_like(g)) for g, _ in sets]
        nbytes = _L().clc_layernorm_bwd_workspace_bytes(rows, Cc)
        ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
        extra, extra_scale, gate = ctx.fold_in.take() if ctx.fold_in is not None else (None, 1.0, None)
        assert gate is None, "gated residual gradients are folded by conv data-gradient kernels only"
        ep, lde = None, 0
        if extra is not None:
            if extra_scale != 1.0:
                extra = extra * extra_scale
            extra, ep, *_q, lde = nhwc(extra)
        defer = direct and DEFER_REDUCTIONS
        gp = [(None, None) if defer else (g.data_ptr(), b.data_ptr()) for g, b in grads]
        if not paired:
            _lib.check(_L().clc_layernorm_bwd(dyp, lddy, xp, ldx, gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), lddx, ep, lde,
                                              gp[0][0], gp[0][1], int(direct), rows, Cc, ws.data_ptr(), nbytes, _stream()), "clc_layernorm_bwd")
        else:
            _lib.check(_L().clc_layernorm_bwd_pair(dyp, lddy, xp, ldx, gamma.data_ptr(), gamma2.data_ptr(), rows // 2, mean.data_ptr(), rstd.data_ptr(),
                                                   dx.data_ptr(), lddx, ep, lde, gp[0][0], gp[0][1], gp[1][0], gp[1][1], int(direct), rows, Cc,
                                                   ws.data_ptr(), nbytes, _stream()), "clc_layernorm_bwd_pair")
        if defer:
            nb = _L().clc_layernorm_bwd_blocks(rows, int(paired))
            per = nb // len(sets)
            for k, (g, b) in enumerate(grads):   # partial rows [nb][2][C]: module k owns rows [k*per, (k+1)*per)
                defer_reduce(ws, per, 2 * Cc, g, b, Cc, offset=k * per * 2 * Cc)
        if direct:
            return (dx, None, None, None, None, None, None)
        return (dx, grads[0][0], grads[0][1], None, None) + ((grads[1][0], grads[1][1]) if paired else (None, None))


def layernorm(x, gamma, beta, fold_in=None, grad_slot=None, gamma2=None, beta2=None):
    _note_grad_mode()
    return _LayerNormFn.apply(x, gamma, beta, fold_in, grad_slot, gamma2, beta2)


# ------------------------------------------------------------------------------ window attention


class _WinAttnFn(Function):
    """qkv [N,3C,H,W] pixel-major -> attention output [N,C,H,W]; relbias [heads,2ws-1,2ws-1].
    relbias2: paired modules — the second half of the batch uses the second module's table."""

    @staticmethod
    def forward(ctx, qkv, relbias, heads, ws, shift, relbias2=None):
        _own(ctx)
        qkv, qp, N, H, W, C3, ldq = nhwc(qkv)
        Cc = C3 // 3
        out = new_act(N, Cc, H, W, qkv)
        need = _recording(ctx)
        lse = torch.empty(N * H * W * heads, device=qkv.device, dtype=torch.float32) if need else None
        tabs = [relbias] + ([relbias2] if relbias2 is not None else [])
        rbs = [t if t.is_contiguous() else t.contiguous() for t in tabs]
        ctx.rb_params = [t if r is t else None for t, r in zip(tabs, rbs)]
        lp = lse.data_ptr() if need else None
        _prof_hint(4.0 * N * H * W * ws * ws * Cc, f"winattn fwd C{Cc} h{heads} ws{ws} {N}x{H}x{W}")   # QK^T + PV: 2 x 2 T hd per token and head
        if relbias2 is None:
            _lib.check(_L().clc_winattn_fwd(qp, ldq, rbs[0].data_ptr(), out.data_ptr(), Cc, lp, N, H, W, Cc, heads, ws, int(shift), _stream()), "clc_winattn_fwd")
        else:
            _lib.check(_L().clc_winattn_fwd_pair(qp, ldq, rbs[0].data_ptr(), rbs[1].data_ptr(), out.data_ptr(), Cc, lp, N, H, W, Cc, heads, ws, int(shift),
                                                 _stream()), "clc_winattn_fwd_pair")
        ctx.cfg = (heads, ws, shift, len(tabs))
        ctx.save_for_backward(qkv, out, lse, *rbs)
        return out

    @staticmethod
    def backward(ctx, dout):
        _reown(ctx)
        heads, ws, shift, ntab = ctx.cfg
        qkv, out, lse, *rbs = ctx.saved_tensors
        paired = ntab == 2
        qkv, qp, N, H, W, C3, ldq = nhwc(qkv)
        Cc = C3 // 3
        dout, dop, *_r, lddo = nhwc(dout)
        dqkv = new_act(N, C3, H, W, qkv)
        grbs = [_direct_grad(t) for t in ctx.rb_params]
        direct = all(g is not None for g in grbs)
        drbs = grbs if direct else [torch.empty_like(r) for r in rbs]
        nbytes = _L().clc_winattn_bwd_workspace_bytes(N, H, W, heads, ws)
        wsb = torch.empty((nbytes + 3) // 4, device=qkv.device, dtype=torch.float32)
        defer = direct and DEFER_REDUCTIONS
        dp = [None if defer else d.data_ptr() for d in drbs]
        _prof_hint(8.0 * N * H * W * ws * ws * Cc, f"winattn bwd C{Cc} h{heads} ws{ws} {N}x{H}x{W}")   # (counted as 2 x forward, like the convolutions)
        if not paired:
            _lib.check(_L().clc_winattn_bwd(dop, lddo, qp, ldq, rbs[0].data_ptr(), out.data_ptr(), Cc, lse.data_ptr(), dqkv.data_ptr(), C3,
                                            dp[0], int(direct), N, H, W, Cc, heads, ws, int(shift), wsb.data_ptr(), nbytes, _stream()), "clc_winattn_bwd")
        else:
            _lib.check(_L().clc_winattn_bwd_pair(dop, lddo, qp, ldq, rbs[0].data_ptr(), rbs[1].data_ptr(), out.data_ptr(), Cc, lse.data_ptr(),
                                                 dqkv.data_ptr(), C3, dp[0], dp[1], int(direct), N, H, W, Cc, heads, ws, int(shift), wsb.data_ptr(),
                                                 nbytes, _stream()), "clc_winattn_bwd_pair")
        if defer:
            n_ = heads * (2 * ws - 1) * (2 * ws - 1)
            nb = _L().clc_winattn_bwd_blocks(N, H, W, heads, ws, int(paired))
            per = nb // ntab
            for k, d in enumerate(drbs):
                defer_reduce(wsb, per, n_, d, None, n_, offset=k * per * n_)
        if direct:
            return dqkv, None, None, None, None, None
        return dqkv, drbs[0], None, None, None, (drbs[1] if paired else None)


def window_attention(qkv, relbias, heads, ws, shift, relbias2=None):
    _note_grad_mode()
    return _WinAttnFn.apply(qkv, relbias, int(heads), int(ws), bool(shift), relbias2)


# ------------------------------------------------------------------------------------------ gate


class _GateFn(Function):
    """a * sigmoid(b) + idn."""

    @staticmethod
    def forward(ctx, a, b, idn, a_slot=None):
        _own(ctx)
        a, b, idn = dense(a), dense(b), dense(idn)
        out = new_act(*a.shape, a)
        _lib.check(_L().clc_gate_fwd(a.data_ptr(), b.data_ptr(), idn.data_ptr(), out.data_ptr(), a.numel(), _stream()), "clc_gate_fwd")
        ctx.a_slot = a_slot   # (BatchSlots, half): where d(a) is to be written (a is one batch half of a stacked tensor)
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        _reown(ctx)
        a, b = ctx.saved_tensors
        g = dense(g)
        da = ctx.a_slot[0].view(a, 0, ctx.a_slot[1], 0) if ctx.a_slot is not None else new_act(*a.shape, a)
        db = new_act(*a.shape, a)
        _lib.check(_L().clc_gate_bwd(g.data_ptr(), a.data_ptr(), b.data_ptr(), da.data_ptr(), db.data_ptr(), a.numel(), _stream()), "clc_gate_bwd")
        return da, db, g, None


def gate(a, b, idn, a_slot=None):
    return _GateFn.apply(a, b, idn, a_slot)


# --------------------------------------------------------------------------------------- entropy


class _GaussLikFn(Function):
    """GaussianConditional.forward likelihood (training: additive noise; eval: dequantised)."""

    @staticmethod
    def forward(ctx, y, scale, mu, noise, training, lik_out=None):
        y, yp, N, H, W, Cc, ldy = nhwc(y)
        scale, sp, *_a, ldsc = nhwc(scale)
        mu, mp, *_b, ldmu = nhwc(mu)
        rows = N * H * W
        ldl = Cc
        if lik_out is not None:   # a channel range of a wider buffer (the model gathers the slices' likelihoods without a concatenation)
            lo, _lp, *_q, ldl = nhwc(lik_out)
            assert lo is lik_out and tuple(lik_out.shape) == (N, Cc, H, W)
            lik = lik_out
        else:
            lik = new_act(N, Cc, H, W, y)
        y_hat = new_act(N, Cc, H, W, y)
        npn, ldn = (None, 0)
        if training:
            noise, npn, *_c, ldn = nhwc(noise)
        _lib.check(_L().clc_gauss_lik_fwd(yp, ldy, mp, ldmu, sp, ldsc, npn, ldn, lik.data_ptr(), ldl, y_hat.data_ptr(), Cc, rows, Cc,
                                          0 if training else 1, None, 0, _stream()), "clc_gauss_lik_fwd")
        ctx.training = training
        ctx.save_for_backward(y, scale, mu, noise if training else None)
        return (lik.detach() if lik_out is not None else lik), y_hat

    @staticmethod
    def backward(ctx, dlik, dyhat):
        y, scale, mu, noise = ctx.saved_tensors
        y, yp, N, H, W, Cc, ldy = nhwc(y)
        scale, sp, *_a, ldsc = nhwc(scale)
        mu, mp, *_b, ldmu = nhwc(mu)
        dlik, dlp, *_c, lddl = nhwc(dlik)
        rows = N * H * W
        npn, ldn = (None, 0)
        if ctx.training:
            noise, npn, *_d, ldn = nhwc(noise)
        dy = new_act(N, Cc, H, W, y) if ctx.training else None
        # d(mu) and d(scale) as the two batch halves of ONE buffer: the paired slice nets produce [mu; scale] as one stacked tensor, and
        # _SplitBatchFn.backward hands such a pair back as it is instead of concatenating it
        both = new_act(2 * N, Cc, H, W, y)
        dmu = both[:N] if ctx.training else None
        dsc = both[N:]
        add_p, ldadd = None, 0
        if ctx.training and dyhat is not None:   # the straight-through gradient of y_hat rides in the same pass (was an elementwise add)
            dyhat, add_p, *_e, ldadd = nhwc(dyhat)
        _lib.check(_L().clc_gauss_lik_bwd(dlp, lddl, yp, ldy, mp, ldmu, sp, ldsc, npn, ldn,
                                          dy.data_ptr() if dy is not None else None, Cc, dmu.data_ptr() if dmu is not None else None, Cc,
                                          dsc.data_ptr(), Cc, rows, Cc, 0 if ctx.training else 1, add_p, ldadd, _stream()), "clc_gauss_lik_bwd")
        # y_hat = round(y - mu) + mu with the straight-through estimator: d y_hat/dy = 1, d y_hat/dmu = 0
        if dy is None:
            dy = dyhat
        return dy, dsc, dmu, None, None, None


def gaussian_likelihood(y, scale, mu, noise, training, lik_out=None):
    """-> (likelihood, y_hat) with y_hat = ste_round(y - mu) + mu  (CLC_run.py:569-571).  lik_out: optional destination (a pixel-major
    view, e.g. a channel range of the buffer that collects all slices' likelihoods)."""
    return _GaussLikFn.apply(y, scale, mu, noise, bool(training), lik_out)


def _eb_ptrs(mats, biases, factors):
    return (_lib.ptr_array([t.data_ptr() for t in mats]), _lib.ptr_array([t.data_ptr() for t in biases]),
            _lib.ptr_array([t.data_ptr() for t in factors]))


class _EBLikFn(Function):
    """EntropyBottleneck likelihood of z [N,C,H,W]; params = 5 matrices, 5 biases, 4 factors."""

    @staticmethod
    def forward(ctx, z, noise, quantiles, training, *params):
        mats, biases, factors = params[0:5], params[5:10], params[10:14]
        z, zp, N, H, W, Cc, ldz = nhwc(z)
        rows = N * H * W
        lik = new_act(N, Cc, H, W, z)
        z_hat = new_act(N, Cc, H, W, z)
        npn, ldn = (None, 0)
        if training:
            noise, npn, *_c, ldn = nhwc(noise)
        pm, pb, pf = _eb_ptrs(mats, biases, factors)
        _lib.check(_L().clc_eb_lik_fwd(zp, ldz, npn, ldn, quantiles.data_ptr(), pm, pb, pf, lik.data_ptr(), Cc, z_hat.data_ptr(), Cc, rows, Cc,
                                       0 if training else 1, _stream()), "clc_eb_lik_fwd")
        ctx.training = training
        ctx.save_for_backward(z, noise if training else None, quantiles, *params)
        return lik, z_hat

    @staticmethod
    def backward(ctx, dlik, dzhat):
        z, noise, quantiles, *params = ctx.saved_tensors
        mats, biases, factors = params[0:5], params[5:10], params[10:14]
        z, zp, N, H, W, Cc, ldz = nhwc(z)
        dlik, dlp, *_c, lddl = nhwc(dlik)
        rows = N * H * W
        npn, ldn = (None, 0)
        if ctx.training:
            noise, npn, *_d, ldn = nhwc(noise)
        grads = [torch.empty_like(p) for p in params]
        pm, pb, pf = _eb_ptrs(mats, biases, factors)
        gm, gb, gf = _eb_ptrs(grads[0:5], grads[5:10], grads[10:14])
        dz = new_act(N, Cc, H, W, z) if ctx.training else None
        _lib.check(_L().clc_eb_lik_bwd(dlp, lddl, zp, ldz, npn, ldn, quantiles.data_ptr(), pm, pb, pf, gm, gb, gf,
                                       dz.data_ptr() if dz is not None else None, Cc, rows, Cc, 0 if ctx.training else 1, _stream()), "clc_eb_lik_bwd")
        if dzhat is not None:  # z_hat = ste_round(z - med) + med
            dz = dzhat if dz is None else dz + dzhat
        direct = [_direct_grad(p) for p in params]
        if all(d is not None for d in direct):
            # the optimizer's gradient arena: ONE multi-tensor add instead of autograd's 14 single-workgroup accumulation kernels
            torch._foreach_add_(direct, grads)
            grads = [None] * len(params)
        return (dz, None, None, None, *grads)


def eb_likelihood(z, noise, quantiles, training, mats, biases, factors):
    """-> (likelihood, z_hat) with z_hat = ste_round(z - median) + median  (CLC_run.py:526-530)."""
    return _EBLikFn.apply(z, noise, quantiles, bool(training), *mats, *biases, *factors)


class _EBAuxFn(Function):
    @staticmethod
    def forward(ctx, quantiles, target, *params):
        mats, biases, factors = params[0:5], params[5:10], params[10:14]
        Cc = quantiles.shape[0]
        part = torch.empty(Cc, device=quantiles.device, dtype=torch.float32)
        dq = torch.empty_like(quantiles)
        pm, pb, pf = _eb_ptrs(mats, biases, factors)
        _lib.check(_L().clc_eb_aux(quantiles.data_ptr(), pm, pb, pf, target.data_ptr(), part.data_ptr(), dq.data_ptr(), Cc, _stream()), "clc_eb_aux")
        out = torch.zeros(1, device=quantiles.device, dtype=torch.float32)
        _lib.check(_L().clc_sum_partials(part.data_ptr(), Cc, 1.0, out.data_ptr(), 0, _stream()), "clc_sum_partials")
        ctx.save_for_backward(dq)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return (dq * g, None) + (None,) * 14


def eb_aux_loss(quantiles, target, mats, biases, factors):
    return _EBAuxFn.apply(quantiles, target, *[m.detach() for m in mats], *[b.detach() for b in biases], *[f.detach() for f in factors])


def quantize_build_indexes(y, mu, scale, scale_table):
    """-> (symbols int32 [N,C,H,W], indexes int32, y_hat) — INT path of compress()."""
    y, yp, N, H, W, Cc, ldy = nhwc(y)
    mu, mp, *_a, ldmu = nhwc(mu)
    scale, sp, *_b, ldsc = nhwc(scale)
    rows = N * H * W
    sym = torch.empty((N, H, W, Cc), device=y.device, dtype=torch.int32)
    idx = torch.empty((N, H, W, Cc), device=y.device, dtype=torch.int32)
    y_hat = new_act(N, Cc, H, W, y)
    _lib.check(_L().clc_quantize_build_indexes(yp, ldy, mp, ldmu, sp, ldsc, scale_table.data_ptr(), scale_table.numel(), sym.data_ptr(),
                                               idx.data_ptr(), y_hat.data_ptr(), Cc, rows, Cc, _stream()), "clc_quantize_build_indexes")
    # logical NCHW views over the NHWC int buffers
    return sym.permute(0, 3, 1, 2), idx.permute(0, 3, 1, 2), y_hat


# --------------------------------------------------------------------------------- RD loss pieces


class _SumLog2Fn(Function):
    """sum(log2(lik)) over a likelihood tensor, two-stage fixed-order reduction."""

    @staticmethod
    def forward(ctx, lik):
        lik, lp, N, H, W, Cc, ld = nhwc(lik)
        rows = N * H * W
        nb = max(1, min(1024, (rows * Cc + 1023) // 1024))
        part = torch.empty(nb, device=lik.device, dtype=torch.float32)
        out = torch.empty(1, device=lik.device, dtype=torch.float32)
        _lib.check(_L().clc_log2_sum_partials(lp, ld, rows, Cc, part.data_ptr(), nb, _stream()), "clc_log2_sum_partials")
        _lib.check(_L().clc_sum_partials(part.data_ptr(), nb, 1.0, out.data_ptr(), 0, _stream()), "clc_sum_partials")
        ctx.save_for_backward(lik)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (lik,) = ctx.saved_tensors
        lik, lp, N, H, W, Cc, ld = nhwc(lik)
        g = g.contiguous()
        d = new_act(N, Cc, H, W, lik)
        _lib.check(_L().clc_scaled_recip(lp, ld, N * H * W, Cc, g.data_ptr(), 1.0 / math.log(2.0), d.data_ptr(), Cc, _stream()), "clc_scaled_recip")
        return d


def sum_log2(lik):
    return _SumLog2Fn.apply(lik)


class _SqDiffSumFn(Function):
    """sum((a - b)^2); gradient flows to ``a`` only (b is the target image)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = dense(a), dense(b)
        n = a.numel()
        nb = max(1, min(1024, (n + 4095) // 4096))
        part = torch.empty(nb, device=a.device, dtype=torch.float32)
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        _lib.check(_L().clc_sqdiff_partials(a.data_ptr(), b.data_ptr(), n, part.data_ptr(), nb, _stream()), "clc_sqdiff_partials")
        _lib.check(_L().clc_sum_partials(part.data_ptr(), nb, 1.0, out.data_ptr(), 0, _stream()), "clc_sum_partials")
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        d = new_act(*a.shape, a)
        _lib.check(_L().clc_scaled_diff(a.data_ptr(), b.data_ptr(), a.numel(), g.data_ptr(), 2.0, d.data_ptr(), _stream()), "clc_scaled_diff")
        return d, None


def sqdiff_sum(a, b):
    return _SqDiffSumFn.apply(a, b)


class _RDLossMseFn(Function):
    """(bpp_loss, mse_loss, loss) of the reference's RateDistortionLoss in its MSE form (/root/reference/train_CLC.py:43-59):
        bpp = (sum log2 lik_y + sum log2 lik_z) / (-num_pixels);  mse = mean (x_hat - x)^2;  loss = lmbda * 255^2 * mse + bpp
    as 4 launches forward (three partial-sum passes + clc_rd_combine) and 4 backward, instead of 6 reduction launches + 5 scalar tensor
    expressions forward and their autograd nodes backward.  Every rounding step of the reference's expressions is kept (see
    rd_combine_kernel): the same bits as the chain of ops it replaces."""

    @staticmethod
    def forward(ctx, lik_y, lik_z, x_hat, target, lmbda, num_pixels):
        L, st = _L(), _stream()
        parts = []
        for lik in (lik_y, lik_z):
            lk, lp, N, H, W, Cc, ld = nhwc(lik)
            rows = N * H * W
            nb = max(1, min(1024, (rows * Cc + 1023) // 1024))
            part = torch.empty(nb, device=lk.device, dtype=torch.float32)
            _lib.check(L.clc_log2_sum_partials(lp, ld, rows, Cc, part.data_ptr(), nb, st), "clc_log2_sum_partials")
            parts.append((part, nb, lk))
        a, b = dense(x_hat), dense(target)
        n = a.numel()
        nb = max(1, min(1024, (n + 4095) // 4096))
        psq = torch.empty(nb, device=a.device, dtype=torch.float32)
        _lib.check(L.clc_sqdiff_partials(a.data_ptr(), b.data_ptr(), n, psq.data_ptr(), nb, st), "clc_sqdiff_partials")
        out = [torch.empty(1, device=a.device, dtype=torch.float32) for _ in range(3)]
        ctx.consts = (float(-num_pixels), float(n), float(lmbda * 255 ** 2))
        _lib.check(L.clc_rd_combine(parts[0][0].data_ptr(), parts[0][1], parts[1][0].data_ptr(), parts[1][1], psq.data_ptr(), nb, *ctx.consts,
                                    out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st), "clc_rd_combine")
        ctx.save_for_backward(parts[0][2], parts[1][2], a, b)
        return tuple(o.reshape(()) for o in out)

    @staticmethod
    def backward(ctx, g_bpp, g_mse, g_loss):
        ly, lz, a, b = ctx.saved_tensors
        L, st = _L(), _stream()
        gs = torch.empty(2, device=a.device, dtype=torch.float32)
        ptr = lambda g: (g.contiguous().data_ptr() if g is not None else None)
        _lib.check(L.clc_rd_grad_scalars(ptr(g_bpp), ptr(g_mse), ptr(g_loss), *ctx.consts, gs.data_ptr(), gs.data_ptr() + 4, st), "clc_rd_grad_scalars")
        grads = []
        for lik in (ly, lz):
            lk, lp, N, H, W, Cc, ld = nhwc(lik)
            d = new_act(N, Cc, H, W, lk)
            _lib.check(L.clc_scaled_recip(lp, ld, N * H * W, Cc, gs.data_ptr(), 1.0 / math.log(2.0), d.data_ptr(), Cc, st), "clc_scaled_recip")
            grads.append(d)
        dx = new_act(*a.shape, a)
        _lib.check(L.clc_scaled_diff(a.data_ptr(), b.data_ptr(), a.numel(), gs.data_ptr() + 4, 2.0, dx.data_ptr(), st), "clc_scaled_diff")
        return grads[0], grads[1], dx, None, None, None


def rd_loss_mse(lik_y, lik_z, x_hat, target, lmbda, num_pixels):
    return _RDLossMseFn.apply(lik_y, lik_z, x_hat, target, lmbda, num_pixels)


# ------------------------------------------------------------------------------------- MS-SSIM


class _MsSsimMeansFn(Function):
    """Per-scale (mean cs, mean ssim) of x vs y over `levels` dyadic scales -> [levels, B*C, 2]; gradient flows to x."""

    @staticmethod
    def forward(ctx, x, y, levels, data_range):
        x, y = dense(x), dense(y)
        B, Cc, H, W = x.shape
        L = _L()
        _lib.check(L.clc_ssim_init(), "clc_ssim_init")
        xs, ys = [x], [y]
        means = torch.empty((levels, B * Cc, 2), device=x.device, dtype=torch.float32)
        for s in range(levels):
            xc, yc = xs[-1], ys[-1]
            h, w = xc.shape[2], xc.shape[3]
            nbytes = L.clc_ssim_workspace_bytes(B, h, w, Cc)
            ws = torch.empty((nbytes + 3) // 4, device=x.device, dtype=torch.float32)
            _lib.check(L.clc_ssim_scale_fwd(xc.data_ptr(), Cc, yc.data_ptr(), Cc, B, h, w, Cc, float(data_range), means[s].data_ptr(),
                                            ws.data_ptr(), nbytes, _stream()), "clc_ssim_scale_fwd")
            if s + 1 < levels:
                if h % 2 or w % 2:
                    raise _lib.ClcError("ms_ssim: image sides must stay even across the scales (e.g. multiples of 16)")
                xn, yn = new_act(B, Cc, h // 2, w // 2, x), new_act(B, Cc, h // 2, w // 2, x)
                _lib.check(L.clc_avgpool2(xc.data_ptr(), Cc, xn.data_ptr(), B, h, w, Cc, _stream()), "clc_avgpool2")
                _lib.check(L.clc_avgpool2(yc.data_ptr(), Cc, yn.data_ptr(), B, h, w, Cc, _stream()), "clc_avgpool2")
                xs.append(xn)
                ys.append(yn)
        ctx.data_range = float(data_range)
        ctx.save_for_backward(*xs, *ys)
        return means

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        levels = len(saved) // 2
        xs, ys = saved[:levels], saved[levels:]
        g = g.contiguous()
        L = _L()
        dnext = None
        for s in reversed(range(levels)):
            xc, yc = xs[s], ys[s]
            B, Cc, h, w = xc.shape
            nbytes = L.clc_ssim_workspace_bytes(B, h, w, Cc)
            ws = torch.empty((nbytes + 3) // 4, device=xc.device, dtype=torch.float32)
            dx = new_act(B, Cc, h, w, xc)
            _lib.check(L.clc_ssim_scale_bwd(xc.data_ptr(), Cc, yc.data_ptr(), Cc, B, h, w, Cc, ctx.data_range, g[s].data_ptr(),
                                            dnext.data_ptr() if dnext is not None else None, dx.data_ptr(), Cc, ws.data_ptr(), nbytes, _stream()), "clc_ssim_scale_bwd")
            dnext = dx
        return dnext, None, None, None


def ms_ssim(x, y, data_range=1.0, weights=(0.0448, 0.2856, 0.3001, 0.2363, 0.1333)):
    """pytorch_msssim.ms_ssim(X, Y, data_range, size_average=True) — the windowed statistics run in HIP, the final
    [5, B, C] product of powers is a handful of scalar-sized torch ops."""
    B, Cc = x.shape[0], x.shape[1]
    means = _MsSsimMeansFn.apply(x, y, len(weights), float(data_range)).view(len(weights), B, Cc, 2)
    # exponents are host scalars: no host->device tensor is built here, so the loss can be captured into a hipGraph
    # (TrainEngine(loss_type='ms_ssim') = the reference's --type ms-ssim mode)
    L = len(weights)
    out = None
    for s in range(L):
        v = torch.relu(means[s, :, :, 0 if s + 1 < L else 1]) ** float(weights[s])
        out = v if out is None else out * v
    return out.mean()
