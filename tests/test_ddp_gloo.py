"""N>1 path on CPU: two processes, gloo backend — gradient averaging over the flat arena and parameter broadcast
(the same GradSync / broadcast_parameters code runs over RCCL on the GPUs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clc_amd.train import FlatArena, GradSync, broadcast_parameters

        torch.manual_seed(100 + rank)  # different init per rank on purpose
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 4, 1))
        net[0].weight.data = net[0].weight.data.contiguous(memory_format=torch.channels_last)
        broadcast_parameters(net)
        ref = [p.detach().clone() for p in net.parameters()]
        params = list(net.parameters())
        arena = FlatArena([p.data for p in params])
        for p, v in zip(params, arena.views):
            p.grad = v
        sync = GradSync(arena.flat, bucket_bytes=256)   # tiny buckets -> several all-reduces
        assert sync.world == world and len(sync.buckets) > 1
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(rank))
        arena.flat.zero_()
        net(x).square().mean().backward()
        local = arena.flat.clone()
        sync.start()
        sync.finish()
        q.put((rank, [t.tolist() for t in ref], local.tolist(), arena.flat.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_mean_and_broadcast():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ref0, loc0, avg0), (_, ref1, loc1, avg1) = res
    assert ref0 == ref1, "parameters were not broadcast from rank 0"
    want = (torch.tensor(loc0) + torch.tensor(loc1)) / 2
    assert torch.allclose(torch.tensor(avg0), want, atol=1e-7) and avg0 == avg1
    assert loc0 != loc1


# ------------------------------------------------------------------------------------------------------------------------
# The TrainEngine step structure of the multi-GPU path, end to end, on two CPU processes: forward + loss + backward cut at the
# encoders' outputs (phase-0 gradients go on the wire while the encoders' backward runs), second exchange phase, optimizer, aux step.
# The HIP pieces (RD loss kernels, fused AdamW) are replaced by plain-torch stand-ins through the engine's own hooks.


class _StubOpt:
    """FusedAdamW's interface over flat arenas, plain SGD arithmetic (the exchange / phase logic is what is under test)."""

    def __init__(self, params, lr, max_norm=0.0):
        from clc_amd.train import FlatArena

        self.params, self.lr = params, lr
        self.p_arena, self.g_arena = FlatArena([p.data for p in params]), FlatArena([p.data for p in params])
        with torch.no_grad():
            for p, pv, gv in zip(params, self.p_arena.views, self.g_arena.views):
                pv.copy_(p.data)
                p.data, p.grad = pv, gv

    grad_flat = property(lambda self: self.g_arena.flat)

    def zero_grad(self):
        self.g_arena.flat.zero_()

    def step(self):
        self.p_arena.flat.sub_(self.lr * self.g_arena.flat)

    def set_lr(self, lr):
        self.lr = lr


class _StubCodec(torch.nn.Module):
    """g_a / ref branch -> latent -> the rest, with the boundary hook of clc_amd.models.clc._SliceCodec.forward."""

    _boundary_ok = True
    _boundary = None

    def __init__(self):
        super().__init__()
        self.g_a = torch.nn.Conv2d(3, 6, 3, padding=1)
        self.ref_encoder = torch.nn.Conv2d(3, 2, 1)
        self.h_a = torch.nn.Conv2d(6, 6, 1)
        self.g_s = torch.nn.Conv2d(8, 3, 3, padding=1)
        self.quantiles = torch.nn.Parameter(torch.tensor([-1.0, 0.0, 1.0]))

    def forward(self, x, refs):
        y = self.g_a(x)
        rf = self.ref_encoder(refs[0])
        if getattr(self, "_keep_boundary", False):
            self._boundary = (y, rf)
        return {"x_hat": self.g_s(torch.cat([self.h_a(y) + y, rf], 1))}

    def aux_loss(self):
        return (self.quantiles ** 2).sum()


def _engine_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from clc_amd.train import TrainEngine, broadcast_parameters

        torch.manual_seed(7 + rank)
        net = _StubCodec()
        broadcast_parameters(net)
        ref_net = _StubCodec()
        ref_net.load_state_dict(net.state_dict())
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(rank))
        r = [torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(10 + rank))]
        crit = lambda out, tgt: {"loss": ((out["x_hat"] - tgt) ** 2).mean()}
        eng = TrainEngine(net, lmbda=0.0, lr=0.1, aux_lr=0.1, clip_max_norm=0.0, use_graph=False, side_stream=False, criterion=crit,
                          optimizer_factory=lambda params, lr, max_norm: _StubOpt(params, lr, max_norm))
        eng.step(x, r)
        assert eng.two_phase and len(eng.sync.phases) == 2 and eng.sync.world == world
        # what plain autograd computes locally on the pre-step parameters
        crit(ref_net(x, r), x)["loss"].backward()
        local = {n: p.grad.clone() for n, p in ref_net.named_parameters() if p.grad is not None}
        names = [n for n, _ in net.named_parameters() if not n.endswith("quantiles")]
        # arena order: encoders first (second exchange phase), everything downstream after (first phase)
        order = [n for n in names if n.startswith(TrainEngine.LATE_PREFIXES)] + [n for n in names if not n.startswith(TrainEngine.LATE_PREFIXES)]
        assert [id(p) for p in eng.opt.params] == [id(dict(net.named_parameters())[n]) for n in order]
        q.put((rank, {n: local[n].tolist() for n in names}, {n: dict(net.named_parameters())[n].grad.tolist() for n in names},
               {n: p.detach().tolist() for n, p in net.named_parameters()}))
    finally:
        dist.destroy_process_group()


def test_two_rank_engine_step_two_phase_exchange():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, loc0, avg0, par0), (_, loc1, avg1, par1) = res
    for n in loc0:
        want = (torch.tensor(loc0[n]) + torch.tensor(loc1[n])) / 2
        assert torch.allclose(torch.tensor(avg0[n]), want, atol=1e-6), f"{n}: exchanged gradient is not the mean over ranks"
        assert avg0[n] == avg1[n]
    assert par0 == par1, "ranks diverged after one step"


def _forced_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CLC_FORCE_COLLECTIVES="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from clc_amd.train import GradSync

        flat = torch.arange(1000, dtype=torch.float32)
        sync = GradSync(flat, bucket_bytes=1024, phases=[(400, 1000), (0, 400)])
        assert sync.world == 1 and sync.active and sync.backend == "gloo"
        sync.start(0)
        sync.start(1)
        sync.finish()
        q.put((sync.launched, len(sync.buckets), bool(torch.equal(flat, torch.arange(1000, dtype=torch.float32)))))
    finally:
        dist.destroy_process_group()


def test_one_rank_group_with_forced_collectives_issues_every_bucket():
    """CLC_FORCE_COLLECTIVES=1 (the knob that makes a one-GPU box execute the RCCL path, tests/test_rccl_one_rank_gpu.py): a 1-rank group
    still launches one all-reduce per bucket of every phase and leaves the sum (= the values) in place, unscaled."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), q))
    p.start()
    launched, n_buckets, same = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert launched == n_buckets == 5 and same


def test_without_the_knob_a_one_rank_group_stays_silent():
    from clc_amd.train import GradSync

    sync = GradSync(torch.zeros(8))
    assert not sync.active and sync.world == 1
    sync.start()
    sync.finish()
    assert sync.launched == 0
