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
