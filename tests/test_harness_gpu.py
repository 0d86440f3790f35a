"""GPU parity of the harness arithmetic that runs inside every timed step:
the fused clip_grad_norm_ + nan_to_num_ + AdamW step, the aux (quantiles) step, the hipGraph-captured engine against the
reference training loop (/root/reference/train_CLC.py:137-183) written with torch.optim.AdamW, the learning-rate hook on a
captured graph, and clc_amd.eval's pad / crop / PSNR / bitrate (/root/reference/eval_CLC.py:133-166, 324-338).
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CL = torch.channels_last


def _make_params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 32, 3, 3), (17,), (128, 64), (5, 3, 1, 1), (4097,), (224, 128, 3, 3), (1, 1), (320, 320)]
    cpu = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    cpu = [t.contiguous(memory_format=CL) if t.dim() == 4 else t for t in cpu]
    return cpu


def _grads(cpu_params, seed, scale):
    g = torch.Generator().manual_seed(1000 + seed)
    return [torch.randn(p.shape, generator=g) * scale for p in cpu_params]


@pytest.mark.parametrize("max_norm", [1.0, 0.0])
def test_fused_adamw_matches_torch(dev, max_norm):
    """6 steps of clip_grad_norm_(max_norm) + nan_to_num_ + torch.optim.AdamW (train_CLC.py:164-179) on CPU vs clc_adamw_step:
    clip active (step 1), clip inactive (2), a NaN gradient (3: the norm is NaN, every gradient becomes 0), an Inf gradient
    (4: coefficient 0, inf*0 = NaN -> 0), ordinary steps after the non-finite ones (5-6: moments must have stayed finite)."""
    from clc_amd.train import FusedAdamW

    cpu = _make_params(dev)
    ref = [torch.nn.Parameter(p.clone()) for p in cpu]
    prm = [torch.nn.Parameter(p.clone().to(dev)) for p in cpu]
    prm = [torch.nn.Parameter(p.data.contiguous(memory_format=CL)) if p.dim() == 4 else p for p in prm]
    opt_ref = torch.optim.AdamW(ref, lr=1e-3, foreach=False)
    opt = FusedAdamW(prm, lr=1e-3, max_norm=max_norm)
    for step, scale in enumerate([1.0, 1e-4, 1.0, 1.0, 0.5, 1e-3], start=1):
        gs = _grads(cpu, step, scale)
        if step == 3:
            gs[2][5, 7] = float("nan")
        if step == 4:
            gs[0][1, 2, 0, 1] = float("inf")
            gs[5][3, 3, 1, 1] = float("-inf")
        old = [p.detach().clone() for p in ref]
        for p, g in zip(ref, gs):
            p.grad = g.clone()
        if max_norm > 0:
            torch.nn.utils.clip_grad_norm_(ref, max_norm)
        for p in ref:
            p.grad.nan_to_num_()
        opt_ref.step()
        opt.zero_grad()
        for p, g in zip(prm, gs):
            p.grad.copy_(g.to(dev))   # the persistent gradient views of the arena
        opt.step()
        torch.cuda.synchronize()
        for i, (a, b, o) in enumerate(zip(prm, ref, old)):
            upd_ref = (b.detach() - o).double()
            scale_u = max(upd_ref.abs().max().item(), 1e-12)
            err = (a.detach().cpu().double() - b.detach().double()).abs()
            # both sides round the new parameter to fp32 (up to 1 ulp of |p| each); beyond that the update itself must agree to 1e-5 of its
            # size — a wrong bias correction, clip coefficient or step count is an error of 10 % .. 10x of the update
            allowed = 2.0 ** -22 * b.detach().abs().double() + 1e-5 * scale_u
            assert bool((err <= allowed).all()), f"step {step} tensor {i}: max excess {(err - allowed).max().item():.3e} (update scale {scale_u:.3e})"
            st = opt_ref.state[b]
            off = opt.p_arena.offsets[i]
            m = opt.m[off: off + b.numel()].cpu()
            v = opt.v[off: off + b.numel()].cpu()
            phys = (lambda t: t.permute(0, 2, 3, 1).reshape(-1)) if b.dim() == 4 else (lambda t: t.reshape(-1))
            if max_norm > 0:   # clipping turns every non-finite gradient into 0 before it reaches the moments
                assert torch.isfinite(m).all() and torch.isfinite(v).all()
            # tolerances: the clip coefficient comes from a float sum of ~3e5 squares in another order than torch.norm's (1e-6
            # relative on g, twice that on g^2); m = m0 + 0.1 (g - m0) cancels, so it is compared on the tensor's scale
            m_ref, v_ref = phys(st["exp_avg"]), phys(st["exp_avg_sq"])
            fin = torch.isfinite(m_ref)
            torch.testing.assert_close(m, m_ref, rtol=5e-6, atol=2e-6 * float(m_ref[fin].abs().max()) + 1e-30, msg=lambda s_: f"step {step} tensor {i} exp_avg: {s_}")
            torch.testing.assert_close(v, v_ref, rtol=1e-5, atol=1e-30, msg=lambda s_: f"step {step} tensor {i} exp_avg_sq: {s_}")
            # the gradient the optimizer saw (clipped, nan_to_num'ed) is left in place, like p.grad in the reference loop
            torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=5e-6, atol=1e-30, msg=lambda s_: f"step {step} tensor {i} grad: {s_}")


def test_adamw_inf_without_clip_saturates(dev):
    """nan_to_num_ maps +-inf to +-FLT_MAX when no clipping multiplies it by 0 first (train_CLC.py:176-178 with clip_max_norm=0)."""
    from clc_amd.train import FusedAdamW

    p_ref = torch.nn.Parameter(torch.linspace(-1, 1, 300))
    p = torch.nn.Parameter(p_ref.detach().clone().to(dev))
    o_ref, o = torch.optim.AdamW([p_ref], lr=1e-3, foreach=False), FusedAdamW([p], lr=1e-3, max_norm=0.0)
    g = torch.linspace(-2, 2, 300)
    g[10], g[20], g[30] = float("inf"), float("-inf"), float("nan")
    for _ in range(2):
        p_ref.grad = g.clone()
        p_ref.grad.nan_to_num_()
        o_ref.step()
        o.zero_grad()
        p.grad.copy_(g.to(dev))
        o.step()
    torch.cuda.synchronize()
    assert torch.allclose(p.detach().cpu(), p_ref.detach(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("world", [2, 8, 3])
def test_adamw_grad_scale_is_the_rank_mean(dev, world):
    """N > 1: the gradient arena holds the rank SUM after the all-reduce and the 1 / world of DDP's mean (run_ddp.sh:1-7) is folded
    into the norm and update kernels (`grad_scale`) instead of a pass of its own.  For a power-of-two world that is the SAME bits as
    scaling first (the product is exact); for world 3 it must agree to rounding."""
    from clc_amd.train import FusedAdamW

    cpu = _make_params(dev)
    mk = lambda: [torch.nn.Parameter(p.clone().to(dev).contiguous(memory_format=CL) if p.dim() == 4 else p.clone().to(dev)) for p in cpu]
    pa, pb = mk(), mk()
    a, b = FusedAdamW(pa, lr=1e-3, max_norm=1.0), FusedAdamW(pb, lr=1e-3, max_norm=1.0)
    b.grad_scale = 1.0 / world
    for step, scale in enumerate([1.0, 1e-4, 0.3], start=1):
        gs = _grads(cpu, step, scale)
        for p, q, g in zip(pa, pb, gs):
            p.grad.copy_(g.to(dev))                 # the mean ...
            q.grad.copy_((g * world).to(dev))       # ... and the sum an all-reduce leaves behind
        a.step()
        b.step()
    torch.cuda.synchronize()
    if world in (2, 8):
        assert torch.equal(a.p_arena.flat, b.p_arena.flat) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
        assert torch.equal(a.g_arena.flat, b.g_arena.flat)   # the gradient left in place is the scaled, clipped one
    else:
        torch.testing.assert_close(a.p_arena.flat, b.p_arena.flat, rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(a.g_arena.flat, b.g_arena.flat, rtol=1e-6, atol=1e-12)


def test_aux_loss_and_aux_step(dev):
    """EntropyBottleneck.loss() (aux loss, train_CLC.py:181) and its quantiles gradient vs the oracle leaf, then three aux-optimizer
    steps (AdamW on *.quantiles, lr 1e-3, train_CLC.py:100-104,182-183) vs torch.optim.AdamW."""
    from clc_amd.entropy_models import EntropyBottleneck
    from clc_amd.recipe import apply_weight_recipe
    from clc_amd.train import FusedAdamW
    from oracle import leaves

    o = leaves.EntropyBottleneck(192)
    apply_weight_recipe(o, 5)
    p = EntropyBottleneck(192)
    p.load_state_dict(o.state_dict())
    p = p.to(dev)
    opt_ref = torch.optim.AdamW([o.quantiles], lr=1e-3, foreach=False)
    opt = FusedAdamW([p.quantiles], lr=1e-3, max_norm=0.0)
    for step in range(3):
        opt_ref.zero_grad()
        lo = o.loss()
        lo.backward()
        opt.zero_grad()
        lp = p.loss()
        lp.backward()
        assert abs(lo.item() - lp.item()) <= 2e-5 * abs(lo.item()), (step, lo.item(), lp.item())
        gq, gr = p.quantiles.grad.cpu(), o.quantiles.grad
        assert (gq - gr).abs().max().item() <= 1e-4 * gr.abs().max().item(), step
        for prm in (p._matrix0, p._bias2, p._factor1):   # the aux loss detaches the density parameters (CompressAI loss())
            assert prm.grad is None
        old = o.quantiles.detach().clone()
        opt_ref.step()
        opt.step()
        upd_ref, upd = o.quantiles.detach() - old, p.quantiles.detach().cpu() - old
        assert (upd - upd_ref).abs().max().item() <= 2e-4 * upd_ref.abs().max().item(), step


def _reference_loop(model, x, refs, steps, lmbda, lr, aux_lr, clip, loss_type="mse"):
    """train_CLC.py:137-183 verbatim in behaviour: zero_grad, forward, loss.backward, clip_grad_norm_, nan_to_num_, AdamW, aux."""
    from clc_amd.train import RateDistortionLoss

    params = [p for n, p in model.named_parameters() if not n.endswith(".quantiles")]
    aux = [p for n, p in model.named_parameters() if n.endswith(".quantiles")]
    opt, aux_opt = torch.optim.AdamW(params, lr=lr), torch.optim.AdamW(aux, lr=aux_lr)
    crit = RateDistortionLoss(lmbda, loss_type)
    seq = []
    for _ in range(steps):
        opt.zero_grad()
        aux_opt.zero_grad()
        out = crit(model(x, refs), x)
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        for p in model.parameters():
            if p.grad is not None:
                p.grad.nan_to_num_()
        opt.step()
        aux_loss = model.aux_loss()
        aux_loss.backward()
        aux_opt.step()
        seq.append((out["loss"].item(), out["bpp_loss"].item(), aux_loss.item()))
    return seq


def _model(dev, R=1, seed=0):
    from clc_amd import models as pm
    from clc_amd.recipe import apply_weight_recipe

    m = pm.CLC(N=64, num_ref_frames=R)
    apply_weight_recipe(m, seed)
    return m.to(dev)


def _inputs(dev, B, R, size=256):
    from clc_amd.recipe import synthetic_image

    return synthetic_image(B, size, size, 100, smooth=True).to(dev), [synthetic_image(B, size, size, 101 + i, smooth=True).to(dev) for i in range(R)]


@pytest.mark.parametrize("wgrad_arith", ["f32_mfma", "default"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_engine_equals_reference_training_loop(dev, use_graph, wgrad_arith, request):
    """TrainEngine (flat arenas, fused optimizer, hipGraph) against the reference loop driven by torch.optim.AdamW on the same
    product model: 4 steps, the loss / bpp / aux-loss sequences and the final parameters.  Also pins that the captured path
    applies exactly ONE update per call (the graph warm-up must not train).
    wgrad_arith: the engine's GROUPED filter-gradient launches form their f32 products from bf16 splits by default (tuning key 24), the reference
    loop's single launches use the f32 MFMAs — both at the same error against fp64, but two independent roundings instead of one summation
    order and another.  "f32_mfma" pins key 24 to 0 (identical arithmetic on both sides: the strict sliver bar); "default" keeps it and
    allows the larger sliver of sign-of-noise elements (measured 0.6-2.2 %; a wrong update rule moves ALL elements)."""
    from clc_amd import lib
    from clc_amd.train import TrainEngine

    if wgrad_arith == "f32_mfma":
        L = lib.load()
        request.addfinalizer(lambda old=L.clc_set_tuning(24, 0): L.clc_set_tuning(24, old))

    x, refs = _inputs(dev, 2, 1)
    m_ref = _model(dev).eval()          # eval-mode rounding: deterministic (the noise proxy would need shared RNG streams)
    ref_seq = _reference_loop(m_ref, x, refs, 4, 0.0067, 1e-4, 1e-3, 1.0)
    m = _model(dev)
    eng = TrainEngine(m, lmbda=0.0067, lr=1e-4, aux_lr=1e-3, clip_max_norm=1.0, use_graph=use_graph, train_mode=False)
    seq = []
    for _ in range(4):
        out = eng.step(x, refs)
        seq.append((out["loss"].item(), out["bpp_loss"].item(), out["aux_loss"].item()))
    for i, (a, b) in enumerate(zip(seq, ref_seq)):
        for u, v, name in zip(a, b, ("loss", "bpp", "aux")):
            # the two runs flush their filter gradients in different groups (another stream-K split of each K range: fp32 summation
            # order only); the loss falls sixfold over these four clip-limited steps, so a last-bit difference in a gradient doubles
            # per step — the bar starts at 3e-4 and doubles with it (a wrong update rule is off by percents at step 1)
            assert abs(u - v) <= 3e-4 * (2 ** i) * max(1.0, abs(v)), f"step {i} {name}: engine {u} vs reference loop {v}"
    assert ref_seq[3][0] < ref_seq[0][0]
    assert float(eng.opt.step_dev[0].item()) == 4.0 and float(eng.aux_opt.step_dev[0].item()) == 4.0
    # Final parameters.  Adam's first update is lr * sign(g): an element whose true gradient is zero (e.g. the key bias of an
    # attention layer: softmax is shift-invariant) moves by +-lr on the sign of rounding noise, in the reference as much as here,
    # so an element-wise bar is ill-posed.  Instead: every element moved by at most ~4 lr in 4 steps, and all but a sliver of
    # the ~49 M elements agree to a few % of ONE step (a wrong step count / bias correction / lr moves ALL of them).
    pr = dict(m_ref.named_parameters())
    n_el = n_bad = 0
    for n, p in m.named_parameters():
        d = (p.detach() - pr[n].detach()).abs()
        assert float(d.max()) <= 2.2 * 4 * 1e-4 * (10.0 if n.endswith(".quantiles") else 1.0), n
        n_el += d.numel()
        n_bad += int((d > 0.05 * 1e-4 * (10.0 if n.endswith(".quantiles") else 1.0)).sum())
    assert n_bad / n_el < (5e-3 if wgrad_arith == "f32_mfma" else 4e-2), (n_bad, n_el)


def test_set_lr_reaches_captured_graph(dev):
    """MultiStepLR (train_CLC.py:453,497) changes lr between epochs: the captured step must follow."""
    from clc_amd.train import TrainEngine

    x, refs = _inputs(dev, 2, 1)
    m = _model(dev)
    eng = TrainEngine(m, lmbda=0.0067, lr=1e-4, aux_lr=1e-3, use_graph=True, train_mode=False)
    eng.step(x, refs)
    w = m.g_a[0].conv1.weight
    q = m.entropy_bottleneck.quantiles
    before, qb = w.detach().clone(), q.detach().clone()
    eng.set_lr(0.0, 0.0)
    eng.step(x, refs)
    assert torch.equal(w.detach(), before) and torch.equal(q.detach(), qb), "lr = 0 must freeze the parameters (update and weight decay scale with lr)"
    eng.set_lr(1e-4, 1e-3)
    eng.step(x, refs)
    d1 = (w.detach() - before).abs().max().item()
    assert 1e-6 < d1 < 5e-4, d1
    before2 = w.detach().clone()
    eng.set_lr(1e-5)
    eng.step(x, refs)
    d2 = (w.detach() - before2).abs().max().item()
    assert d2 < 0.25 * d1, (d1, d2)


def test_ms_ssim_engine_graph_and_shape_change(dev):
    """--type ms-ssim (train_CLC.py:56-57) under hipGraph capture == eager, bit for bit; a batch of another shape (the last,
    short batch of an epoch: the reference DataLoader has no drop_last) falls back to an eager step instead of failing."""
    from clc_amd.train import TrainEngine

    x, refs = _inputs(dev, 2, 1)
    seqs = {}
    for use_graph in (False, True):
        eng = TrainEngine(_model(dev), lmbda=0.05, loss_type="ms_ssim", use_graph=use_graph, train_mode=False)
        seqs[use_graph] = [eng.step(x, refs)["loss"].item() for _ in range(3)]
        assert "ms_ssim_loss" in eng.step(x, refs)
    assert seqs[False] == seqs[True], seqs
    out = eng.step(x[:1], [refs[0][:1]])   # short batch on the captured engine
    assert math.isfinite(out["loss"].item())
    with pytest.raises(ValueError):
        eng.step(x, None)


def test_two_phase_backward_equals_single_backward(dev):
    """The backward pass cut at the encoders' outputs (multi-GPU overlap: clc_amd.train.TrainEngine._fwd_bwd_early / _bwd_late)
    produces the single-pass gradients: every parameter, to fp32 summation order (the filter gradients are flushed in two
    stream-K groups instead of one), and the encoders' parameters sit in front of the arena (second exchange phase)."""
    from clc_amd.train import TrainEngine

    x, refs = _inputs(dev, 2, 1)
    m = _model(dev)
    eng = TrainEngine(m, lmbda=0.0067, use_graph=False, train_mode=False)
    eng._discover(x, refs)
    assert eng.two_phase and len(eng.sync.phases) == 2
    names = {id(p): n for n, p in m.named_parameters()}
    order = [names[id(p)] for p in eng.opt.params]
    n_late = sum(n.startswith(TrainEngine.LATE_PREFIXES) for n in order)
    assert 0 < n_late < len(order) and all(n.startswith(TrainEngine.LATE_PREFIXES) for n in order[:n_late])
    assert not any(n.startswith(TrainEngine.LATE_PREFIXES) for n in order[n_late:])
    cut = eng.opt.p_arena.offsets[n_late]
    assert eng.sync.phases[0][0].data_ptr() == eng.opt.grad_flat[cut:].data_ptr() and eng.sync.phases[1][0].data_ptr() == eng.opt.grad_flat.data_ptr()
    out1 = eng._fwd_bwd(x, refs)
    g1 = eng.opt.grad_flat.clone()
    out2 = eng._fwd_bwd_early(x, refs)
    assert float(eng.opt.grad_flat[:cut].abs().max()) == 0.0, "the encoders' gradients must not exist before the second stage"
    eng._bwd_late()
    g2 = eng.opt.grad_flat.clone()
    assert out1["loss"].item() == out2["loss"].item()
    worst = 0.0
    for i, p in enumerate(eng.opt.params):
        a = g1[eng.opt.p_arena.offsets[i]: eng.opt.p_arena.offsets[i] + p.numel()]
        b = g2[eng.opt.p_arena.offsets[i]: eng.opt.p_arena.offsets[i] + p.numel()]
        scale = a.abs().max().item()
        if scale > 1e-12:
            worst = max(worst, (a - b).abs().max().item() / scale)
    assert worst < 2e-5, worst


def test_force_split_graphs_matches_single_graph(dev, monkeypatch):
    """The multi-GPU step structure (graph A1 = forward + backward down to the encoders' outputs | first exchange phase | graph A2 =
    the encoders' backward | second phase | graph B = optimizer + aux) forced on one GPU: same first loss, bit for bit (identical
    forward), the following ones within training noise (the filter gradients are summed in another order and Adam's first step is
    lr * sign(g), so rounding-boundary flips of the quantiser separate the trajectories), and run-to-run reproducible."""
    from clc_amd.train import TrainEngine

    x, refs = _inputs(dev, 2, 1)
    seqs = []
    for split in ("0", "1"):
        monkeypatch.setenv("CLC_FORCE_SPLIT_GRAPHS", split)
        eng = TrainEngine(_model(dev), lmbda=0.0067, use_graph=True, train_mode=False)
        seqs.append([eng.step(x, refs)["loss"].item() for _ in range(4)])
        if split == "1":
            assert isinstance(eng.graph, tuple) and len(eng.graph) == 3, "two-phase backward structure not taken"
            eng2 = TrainEngine(_model(dev), lmbda=0.0067, use_graph=True, train_mode=False)
            assert [eng2.step(x, refs)["loss"].item() for _ in range(4)] == seqs[1], "split path is not reproducible"
    assert seqs[0][0] == seqs[1][0]
    for a, b in zip(*seqs):
        assert abs(a - b) <= 1e-2 * abs(a), seqs


def test_split_graphs_later_step_matches_single_graph_from_the_same_state(dev, monkeypatch):
    """the loss-sequence bars above cannot see a 1 %-level error in step >= 2 of the three-graph path (trajectories
    separate through rounding flips).  Here the comparison is made FROM THE SAME STATE: two replayed steps of the three-graph engine,
    then its parameters and both optimizers' moments / step counters are copied into a single-graph engine, and both replay step 3 on the
    same batch.  Same loss bit for bit (identical forward on identical parameters); first and second moments after the step — linear /
    quadratic in the step's gradients — within fp32 summation order per parameter tensor (the filter gradients are flushed in other
    stream-K groups); parameters within 3 % of one step's size where Adam's quotient is well conditioned."""
    from clc_amd.train import TrainEngine

    x, refs = _inputs(dev, 2, 1)
    monkeypatch.setenv("CLC_FORCE_SPLIT_GRAPHS", "1")
    split = TrainEngine(_model(dev), lmbda=0.0067, use_graph=True, train_mode=False)
    for _ in range(2):
        split.step(x, refs)
    assert isinstance(split.graph, tuple) and len(split.graph) == 3
    monkeypatch.setenv("CLC_FORCE_SPLIT_GRAPHS", "0")
    single = TrainEngine(_model(dev), lmbda=0.0067, use_graph=True, train_mode=False)
    single.step(x, refs)                       # (builds its arenas and captures its graph)
    assert not isinstance(single.graph, tuple)
    torch.cuda.synchronize()
    single.opt.state_restore(split.opt.state_snapshot())
    single.aux_opt.state_restore(split.aux_opt.state_snapshot())
    before = split.opt.p_arena.flat.clone()
    la, lb = split.step(x, refs)["loss"].item(), single.step(x, refs)["loss"].item()
    assert la == lb, (la, lb)
    lr = split.opt.lr
    worst_m = worst_v = 0.0
    for i, p in enumerate(split.opt.params):
        sl = slice(split.opt.p_arena.offsets[i], split.opt.p_arena.offsets[i] + p.numel())
        for name, a, b in (("m", split.opt.m[sl], single.opt.m[sl]), ("v", split.opt.v[sl], single.opt.v[sl])):
            scale = a.abs().max().item()
            if scale > 1e-20:
                err = (a - b).abs().max().item() / scale
                if name == "m":
                    worst_m = max(worst_m, err)
                else:
                    worst_v = max(worst_v, err)
    assert worst_m < 5e-5 and worst_v < 1e-4, (worst_m, worst_v)
    pa, pb = split.opt.p_arena.flat, single.opt.p_arena.flat
    moved = (pa - before).abs()
    assert float(moved.max()) > 0.5 * lr                                   # the step did move the parameters
    well = split.opt.v.sqrt() > 1e-6                                        # Adam's quotient m / (sqrt(v) + eps) is well conditioned here
    assert float(((pa - pb).abs()[well]).max()) <= 0.03 * lr, float(((pa - pb).abs()[well]).max()) / lr


# ------------------------------------------------------------------------------------------------- eval helpers (row 16)


def test_eval_pad_crop_psnr_bitrate(dev):
    """clc_amd.eval on a 200x300 image (not a multiple of 128): pad / crop geometry and values == the oracle's (eval_CLC.py:133-166),
    PSNR from the HIP squared-error reduction == the oracle's formula, evaluate() rows == bitrate / PSNR recomputed by hand
    from compress()/decompress() (eval_CLC.py:324-338)."""
    from clc_amd import eval as pe
    from clc_amd.recipe import synthetic_image
    from oracle import loss as ol

    x = synthetic_image(1, 200, 300, 7, smooth=True)
    xp_o, pad_o = ol.pad(x, 128)
    xp, pad_p = pe.pad(x.to(dev), 128)
    assert pad_p == pad_o == (42, 42, 28, 28) and tuple(xp.shape) == (1, 3, 256, 384)
    assert torch.equal(xp.cpu(), xp_o)
    assert torch.equal(pe.crop(xp, pad_p).cpu(), x) and torch.equal(ol.crop(xp_o, pad_o), x)
    noisy = (x + 0.01 * torch.randn(x.shape, generator=torch.Generator().manual_seed(3))).clamp(0, 1)
    assert abs(pe.compute_psnr(x.to(dev), noisy.to(dev)) - ol.compute_psnr(x, noisy)) <= 1e-4
    m = _model(dev).eval()
    r = synthetic_image(1, 200, 300, 8, smooth=True)
    res = pe.evaluate(m, [(x[0], [r[0]])], p=128, device=dev)
    # by hand, eval_CLC.py:324-338
    enc = m.compress(xp, [pe.pad(r.to(dev), 128)[0]])
    dec = m.decompress(enc["strings"], enc["shape"], [pe.pad(r.to(dev), 128)[0]])
    x_hat = ol.crop(dec["x_hat"].cpu(), pad_o)
    bitrate = sum(len(s[0]) for s in enc["strings"]) * 8.0 / (200 * 300)
    assert res["rows"][0]["bpp"] == bitrate
    assert abs(res["rows"][0]["psnr"] - ol.compute_psnr(x, x_hat)) <= 1e-3
    assert res["avg_bpp"] == bitrate and res["avg_time_s"] > 0
    # likelihood-based bpp helper (eval_CLC.py:158-166 compute_bpp)
    with torch.no_grad():
        out = m(xp, [pe.pad(r.to(dev), 128)[0]])
    cpu_out = {"x_hat": out["x_hat"].cpu(), "likelihoods": {k: v.cpu() for k, v in out["likelihoods"].items()}}
    assert abs(pe.compute_bpp(out) - ol.compute_bpp(cpu_out)) <= 1e-5


def test_checkpoint_loading_and_rd_sweep(dev, tmp_path):
    """§8(f)-3: reference-format checkpoints ({"state_dict": {"module.<key>": ...}}, the directory layout eval_CLC.py:183-204 globs,
    CDF buffers included so the resize path of load_state_dict runs) load into the HIP model; the sweep writes the reference's CSV
    (eval_CLC.py:395-411) with the numbers evaluate() gives by hand."""
    import csv

    from clc_amd import eval as pe
    from clc_amd import models as pm
    from clc_amd.recipe import apply_weight_recipe, synthetic_image
    from oracle import graph as og

    cps = []
    for lam, seed in (("0.0067", 0), ("0.025", 1)):
        o = og.CLC(N=64, num_ref_frames=1)
        apply_weight_recipe(o, seed)
        o.update(force=True)
        d = tmp_path / f"0322_{lam}"
        d.mkdir()
        sd = {"module." + k: v for k, v in o.state_dict().items()}
        sd["module.some_future_buffer"] = torch.zeros(3)          # extra keys are ignored (CLC_run.py:604-607)
        torch.save({"epoch": 7, "state_dict": sd, "loss": 1.0}, d / f"{lam}checkpoint_best.pth.tar")
        cps.append(o)
    found = pe.find_checkpoints(str(tmp_path))
    assert [c["bitrate"] for c in found] == [0.0067, 0.025]
    net = pm.CLC(N=64, num_ref_frames=1).to(dev)
    meta = pe.load_checkpoint(net, found[1]["path"])
    assert meta["epoch"] == 7
    for k, v in cps[1].state_dict().items():
        assert torch.equal(net.state_dict()[k].cpu(), v), k
    samples = [(synthetic_image(1, 200, 300, 40 + i, smooth=True)[0], [synthetic_image(1, 180, 260, 50 + i, smooth=True)[0]]) for i in range(2)]
    results, csv_path = pe.rd_sweep(lambda: pm.CLC(N=64, num_ref_frames=1), found, samples, str(tmp_path / "res"), device=dev)
    rows = list(csv.reader(open(csv_path)))
    assert rows[0] == ["Checkpoint", "Bitrate (bpp)", "PSNR (dB)", "Time (s)"] and len(rows) == 3
    by_hand = pe.evaluate(net.eval(), samples, device=dev)      # eager model.compress / decompress, checkpoint 2
    assert abs(results[1]["bitrate"] - by_hand["avg_bpp"]) < 1e-12 and abs(results[1]["psnr"] - by_hand["avg_psnr"]) < 1e-9
    assert rows[2][1] == f"{by_hand['avg_bpp']:.4f}" and rows[2][2] == f"{by_hand['avg_psnr']:.2f}"
    assert results[0]["bitrate"] != results[1]["bitrate"]


def test_reduced_precision_mode_bf16(dev):
    """SURVEY 8(f)-4, second half: the opt-in reduced-precision mode (bf16-in / f32-accumulate MFMA in the convolutions of the
    analysis / synthesis transforms and the reference encoder — the counterpart of train_CLC.py:143-174's autocast branch).
    Stated here, at BASELINE configs[1]'s model on the seeded 256x256 image: |d bpp| and |d PSNR| of the bf16-mode forward against
    the CPU oracle (f32), and the gradient error of one bf16-mode training step against the f32-mode step.
    Bars for THIS mode (it is not the parity mode): |d bpp| <= 2e-2, |d PSNR| <= 0.3 dB, per-parameter gradient error <= 10 % of
    the gradient's largest element for all but a handful of parameters; the mode must actually change the bits (it ran)."""
    import clc_amd
    from clc_amd import models as pm
    from clc_amd.recipe import apply_weight_recipe, synthetic_image
    from clc_amd.train import RateDistortionLoss, TrainEngine
    from oracle import graph as og
    from oracle.loss import compute_bpp

    o = og.CLC(N=64, num_ref_frames=1).eval()
    apply_weight_recipe(o, 0)
    p = pm.CLC(N=64, num_ref_frames=1)
    p.load_state_dict(o.state_dict())
    p = p.to(dev).eval()
    x, r = synthetic_image(1, 256, 256, 100, smooth=True), [synthetic_image(1, 256, 256, 101, smooth=True)]
    xd, rd = x.to(dev), [r[0].to(dev)]
    psnr = lambda t: -10 * math.log10(torch.mean((t.double().cpu() - x.double()) ** 2).item())
    with torch.no_grad():
        a = o(x, r)
        f32 = p(xd, rd)
        assert clc_amd.set_precision("bf16") == "f32"
        try:
            b16 = p(xd, rd)
        finally:
            clc_amd.set_precision("f32")
        again = p(xd, rd)
    assert clc_amd.get_precision() == "f32"
    assert torch.equal(again["x_hat"], f32["x_hat"]), "leaving the mode must restore the f32 kernels"
    assert not torch.equal(b16["x_hat"], f32["x_hat"]), "the bf16 mode did not change anything: did it run?"
    cb = lambda out: compute_bpp({"x_hat": out["x_hat"].cpu(), "likelihoods": {k: v.cpu() for k, v in out["likelihoods"].items()}})
    dbpp, dpsnr = abs(cb(b16) - compute_bpp(a)), abs(psnr(b16["x_hat"]) - psnr(a["x_hat"]))
    print(f"bf16 mode vs oracle: |d bpp| = {dbpp:.2e}, |d PSNR| = {dpsnr:.3f} dB  (f32 mode: {abs(cb(f32) - compute_bpp(a)):.1e}, {abs(psnr(f32['x_hat']) - psnr(a['x_hat'])):.1e})")
    assert dbpp <= 2e-2 and dpsnr <= 0.3, (dbpp, dpsnr)

    # gradients of one training step, bf16 mode vs f32 mode (same weights, same batch, deterministic rounding)
    xb = synthetic_image(2, 256, 256, 7, smooth=True).to(dev)
    rb = [synthetic_image(2, 256, 256, 8, smooth=True).to(dev)]
    grads = {}
    for mode in ("f32", "bf16"):
        m = pm.CLC(N=64, num_ref_frames=1)
        apply_weight_recipe(m, 0)
        m = m.to(dev)
        eng = TrainEngine(m, lmbda=0.0067, use_graph=False, train_mode=False, precision=mode)
        clc_amd.set_precision(mode)
        try:
            eng._discover(xb, rb)
            out = eng._fwd_bwd(xb, rb)
        finally:
            clc_amd.set_precision("f32")
        torch.cuda.synchronize()
        grads[mode] = ({n: q.grad.clone() for n, q in m.named_parameters() if q.grad is not None}, out["loss"].item())
    assert abs(grads["bf16"][1] - grads["f32"][1]) <= 2e-2 * abs(grads["f32"][1])
    errs = []
    for n, g in grads["f32"][0].items():
        d = g.abs().max().item()
        if d > 1e-12:
            errs.append(((grads["bf16"][0][n] - g).abs().max().item() / d, n))
    errs.sort(reverse=True)
    worst, med = errs[0][0], errs[len(errs) // 2][0]
    print(f"bf16 vs f32 gradients: worst {worst:.3f} ({errs[0][1]}), median {med:.2e}, > 10 %: {sum(e > 0.1 for e, _ in errs)} of {len(errs)}")
    assert med < 2e-2 and sum(e > 0.1 for e, _ in errs) <= len(errs) // 50, errs[:10]


def test_plain_autograd_after_engine_steps_uses_current_filters(dev):
    """The [Cin][T][Cout] filter images the engine attaches to the parameters for its data-gradient kernels are one optimizer update
    behind once a step has finished: a plain autograd backward on the same model afterwards must not use them (ops.WT_CACHE_VALID)."""
    from clc_amd import ops
    from clc_amd.train import RateDistortionLoss, TrainEngine

    x, refs = _inputs(dev, 2, 1)
    m = _model(dev)
    eng = TrainEngine(m, lmbda=0.0067, use_graph=False, train_mode=False)
    for _ in range(2):
        eng.step(x, refs)
    assert ops.WT_CACHE_VALID is False
    # gradient of the loss w.r.t. the INPUT image through plain autograd, with and without the engine's cached images present
    def input_grad():
        xi = x.clone().requires_grad_(True)
        out = RateDistortionLoss(0.0067)(m(xi, refs), xi.detach())
        (g,) = torch.autograd.grad(out["loss"], [xi])
        return g
    g_with = input_grad()
    saved = {}
    for p in m.parameters():
        if hasattr(p, "_clc_wt"):
            saved[p] = p._clc_wt
            del p._clc_wt
    assert saved, "the engine attached no transposed filter images"
    g_without = input_grad()
    for p, wt in saved.items():
        p._clc_wt = wt
    assert torch.equal(g_with, g_without)
