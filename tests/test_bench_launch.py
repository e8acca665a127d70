"""`python bench.py --gpus N` as the round-end driver may run it: no launcher around it, the script starts its own
ranks (one per GPU, torch.distributed.run on the loopback address) and relays rank 0's single JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_launch_command_shape():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "3"]
    assert 0 < bench.free_port() < 65536


@pytest.mark.skipif(__import__("torch").cuda.is_available(), reason="failure path of the launcher: needs a host without a GPU")
def test_self_launch_reports_failing_ranks():
    """Without a GPU every rank dies at set_device: the parent must exit non-zero and print no JSON line."""
    env = dict(os.environ, LV_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--seq", "4096", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_two_ranks_gloo():
    """The exact plain command, two ranks sharing the box's one GPU (gloo: RCCL refuses two ranks per device)."""
    env = dict(os.environ, LV_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--seq", "65536", "--steps", "1", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and len(line["per_rank"]) == 2
    assert line["config"]["parallelism"] == "cp2" and line["value"] > 0 and line["scaling"] == "strong"
    assert {r["rank"] for r in line["per_rank"]} == {0, 1}
