"""`python bench.py --gpus N` (N > 1) must launch its own ranks (the driver runs it bare; /root/reference/run_ddp.sh:7 is the recipe it
mirrors: `python -m torch.distributed.run --nproc_per_node=8 ...`).  CPU-side checks: argument assembly, the JSON line of rank 0
arriving on the launcher's stdout, and the children's exit code becoming the launcher's."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_command_assembly():
    import bench

    cmd = bench.launcher_command(4, 29999, ["--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]   # the script's own arguments, unchanged, after the script
    # default: the rendezvous picks its own port (no probe-then-bind window), still on the loopback address
    cmd = bench.launcher_command(2, None, ["--gpus", "2"])
    assert "--rdzv-endpoint=127.0.0.1:0" in cmd and "--rdzv-backend=c10d" in cmd and "--master-port" not in cmd
    assert cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"


def _run(args, **env):
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)


def test_bare_gpus2_launches_two_ranks_and_relays_rank0_json():
    r = _run(["--gpus", "2", "--launch-selftest"])
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE JSON line: rank 0's
    rep = json.loads(lines[0])
    assert rep == {"launch_selftest": True, "n_gpus": 2, "ranks_seen": 2}
    assert "launching the ranks" in r.stderr and "torch.distributed.run" in r.stderr


def test_child_failure_becomes_the_launchers_exit_code():
    """no GPU here: every rank stops with `bench.py needs an MI355X` -> torchrun fails -> the launcher must fail too (and print no JSON)"""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_recorded_launches_map_to_the_profilers_kernel_names():
    """bench.py's roofline leg joins its live per-launch HIP-event times with `rocprofv3 --kernel-trace` names: every variant id the conv entry
    point can return must have a name (an unnamed family stopped the 2-rank bench once), and the names are the instantiations' own."""
    import types
    sys.path.insert(0, ROOT)
    import bench
    from clc_amd import lib

    L = lib.load()
    rec = lambda variant, label="fwd 128->128": types.SimpleNamespace(fam="conv_igemm", variant=variant, label=label)
    assert bench._kernel_name(L, rec((13 << 20) | (1 << 4) | 1)) == "conv_wino_kernel<true>"
    assert bench._kernel_name(L, rec((13 << 20) | (4 << 4) | 2, "dgrad 512->128")) == "conv_wino_kernel<false>"
    assert bench._kernel_name(L, rec((13 << 20) | (1 << 12) | (1 << 4) | 2, "dgrad 64->64")) == "conv_wino64_kernel<false>"
    assert bench._kernel_name(L, rec((12 << 20) | (2 << 4) | 2, "dgrad 128->128")) == "conv_halo3x3_kernel<128, true, false>"
    assert bench._kernel_name(L, rec((12 << 20) | (1 << 4) | 1)) == "conv_halo3x3_kernel<64, false, true>"
    for fam in (1, 2, 3, 4, 5, 8, 10, 11, 12, 13):
        assert bench._kernel_name(L, rec((fam << 20) | (2 << 16) | (16 << 3) | 4)).split("<")[0].endswith("kernel")
