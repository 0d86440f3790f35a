"""Multi-GPU readiness without multi-GPU hardware (SURVEY 8(e), configs[3]): two ranks on ONE MI355X over gloo, real kernels, the real
three-graph step (A1 | exchange | A2 | exchange | B), identical shards -> bit-identical to the 1-rank run of the same structure.
The ranks are CHILD processes (torch.distributed.run), launched before they touch the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_device_reproduce_the_single_rank_run(dev, tmp_path):
    ref = str(tmp_path / "ref.json")
    env = dict(os.environ, PYTHONPATH=ROOT)
    a = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rehearse_2rank.py"), "--single", ref, "--steps", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert a.returncode == 0, a.stderr[-2000:]
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(ROOT, "tools", "rehearse_2rank.py"), "--check", ref, "--steps", "2"], capture_output=True, text=True, timeout=360, env=env, cwd=ROOT)
    assert b.returncode == 0, (b.stdout[-1500:], b.stderr[-2500:])
    line = next(l for l in b.stdout.splitlines() if l.startswith("REHEARSAL "))
    rep = json.loads(line[len("REHEARSAL "):])
    assert rep["ranks"] == 2 and rep["bit_identical_to_1rank"] and rep["ranks_agree"] and len(rep["losses_2rank"]) == 2


def test_bench_gpus2_self_launches_two_ranks(dev):
    """The driver's multi-GPU command line, bare: `python bench.py --gpus 2` must start its own two ranks (fresh child processes,
    /root/reference/run_ddp.sh:7), run the three-graph step with the gradient exchange, and put rank 0's JSON line on stdout.
    One-GPU box: both ranks on cuda:0, exchange over gloo (CLC_SINGLE_DEVICE / CLC_DIST_BACKEND); the roofline leg — an eager step
    with the all-reduce inside — runs on both ranks as it will under RCCL."""
    env = dict(os.environ, PYTHONPATH=ROOT, CLC_SINGLE_DEVICE="1", CLC_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, timeout=360, env=env, cwd=ROOT)   # (under the GPU harness's 420-s silence limit: a hung child must fail THIS test, not get the whole run killed)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    rep = json.loads(lines[0])
    assert rep["n_gpus"] == 2 and rep["config"]["ranks_seen"] == 2 and rep["config"]["global_batch"] == 8
    assert rep["scaling"] == "weak" and rep["value"] > 0 and rep["config"]["parallelism"] == "dp2"
    assert "roofline" in rep and rep["roofline"]["frac"] > 0
