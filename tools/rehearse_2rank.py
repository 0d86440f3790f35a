#!/usr/bin/env python3
"""Asserting rehearsal of the multi-GPU training step on a ONE-GPU box (the driver's 8-GPU runs use RCCL: /root/reference/run_ddp.sh:1-7).

Two ranks share cuda:0 and exchange gradients over gloo (RCCL refuses two ranks on one device), with real kernels and the real step
structure: graph A1 (forward + backward down to the encoders' outputs) | exchange phase 0 | graph A2 (encoders' backward) | exchange
phase 1 | graph B (optimizer + aux).  Both ranks get IDENTICAL shards, so the mean of the two equal gradients IS the gradient
(x + x = 2x and 2x / 2 = x are exact in binary floating point): the 2-rank run must reproduce, BIT FOR BIT, the loss sequence and the
parameters of a 1-rank run with the same three-graph structure (CLC_FORCE_SPLIT_GRAPHS=1).

  python tools/rehearse_2rank.py --single OUT.json                        # 1 rank, forced split graphs -> reference values
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/rehearse_2rank.py --check OUT.json
(tools/gpu_rehearse_2rank.sh runs both and keeps the report under gpurun_out/.)"""
import argparse
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps):
    import torch

    from bench import synthetic_batch
    from clc_amd import models
    from clc_amd.recipe import apply_weight_recipe
    from clc_amd.train import TrainEngine, broadcast_parameters

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    model = models.CLC(N=64, num_ref_frames=1)
    apply_weight_recipe(model, 0)
    model = model.to(dev)
    broadcast_parameters(model)
    x = synthetic_batch(2, 256, 100, dev)               # the SAME shard on every rank
    refs = [synthetic_batch(2, 256, 1000, dev)]
    eng = TrainEngine(model, lmbda=0.0067, lr=1e-4, aux_lr=1e-3, clip_max_norm=1.0, use_graph=True, train_mode=False)   # deterministic rounding
    losses = []
    for _ in range(steps):
        out = eng.step(x, refs)
        losses.append(float(out["loss"].item()))
    torch.cuda.synchronize()
    flat = eng.opt.p_arena.flat.detach().cpu().numpy()
    aux = eng.aux_opt.p_arena.flat.detach().cpu().numpy()
    return {"losses": losses, "param_sha256": hashlib.sha256(flat.tobytes()).hexdigest(), "aux_sha256": hashlib.sha256(aux.tobytes()).hexdigest(),
            "param_l2": float((flat.astype("float64") ** 2).sum() ** 0.5), "n_params": int(flat.size), "two_phase": bool(eng.two_phase),
            "graphs": (len(eng.graph) if isinstance(eng.graph, tuple) else 1), "world": int(eng.sync.world),
            "collectives_issued": int(eng.sync.launched + eng.aux_sync.launched)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--single", help="write the 1-rank reference values to this JSON")
    ap.add_argument("--check", help="2-rank run: compare with this JSON")
    ap.add_argument("--rccl1", help="ONE rank, backend nccl (= RCCL), CLC_FORCE_COLLECTIVES=1: every all-reduce of the three-graph step is really "
                    "issued on a 1-rank communicator; compare with this JSON (the 1-rank forced-split run without a process group)")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    if a.rccl1:
        import socket

        import torch
        import torch.distributed as dist

        os.environ["CLC_FORCE_COLLECTIVES"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        try:
            res = run(a.steps)
            with open(a.rccl1) as fh:
                want = json.load(fh)
            ok = res["losses"] == want["losses"] and res["param_sha256"] == want["param_sha256"] and res["aux_sha256"] == want["aux_sha256"]
            report = {"ranks": 1, "backend": dist.get_backend(), "steps": a.steps, "graphs": res["graphs"], "two_phase": res["two_phase"],
                      "collectives_issued": res["collectives_issued"], "losses_rccl": res["losses"], "losses_no_group": want["losses"],
                      "bit_identical_to_no_group": bool(ok), "param_sha256": res["param_sha256"],
                      "structure": "graph A1 | RCCL all-reduce phase 0 | graph A2 | phase 1 | graph B | aux all-reduce, 1-rank nccl communicator on one MI355X"}
            print("REHEARSAL " + json.dumps(report), flush=True)
            assert res["graphs"] == 3 and res["two_phase"] and res["collectives_issued"] > 0, res
            assert ok, ("the RCCL run differs from the run without a process group", res["losses"], want["losses"])
        finally:
            dist.barrier()
            dist.destroy_process_group()
        return
    if a.single:
        os.environ["CLC_FORCE_SPLIT_GRAPHS"] = "1"
        res = run(a.steps)
        assert res["world"] == 1 and res["graphs"] == 3, res
        with open(a.single, "w") as fh:
            json.dump(res, fh)
        print("single-rank reference:", json.dumps(res))
        return
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        res = run(a.steps)
        assert res["world"] == world == 2 and res["graphs"] == 3 and res["two_phase"], res
        with open(a.check) as fh:
            want = json.load(fh)
        ok = res["losses"] == want["losses"] and res["param_sha256"] == want["param_sha256"] and res["aux_sha256"] == want["aux_sha256"]
        gathered = [None] * world
        dist.all_gather_object(gathered, (res["losses"], res["param_sha256"]))
        same_on_all = all(g == gathered[0] for g in gathered)
        if rank == 0:
            report = {"ranks": world, "steps": a.steps, "losses_2rank": res["losses"], "losses_1rank": want["losses"],
                      "bit_identical_to_1rank": bool(ok), "ranks_agree": bool(same_on_all), "param_sha256": res["param_sha256"],
                      "structure": "graph A1 | gloo all-reduce phase 0 | graph A2 | phase 1 | graph B, two ranks on one MI355X, identical shards"}
            print("REHEARSAL " + json.dumps(report), flush=True)
        assert same_on_all, "ranks diverged"
        assert ok, ("2-rank run differs from the 1-rank run of the same structure", res["losses"], want["losses"])
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
