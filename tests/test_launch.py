"""CPU: `bench.py --gpus N` launches itself (no torchrun around it): the parent spawns N ranks through
`python -m torch.distributed.run`, the ranks form a process group (gloo here, RCCL on the GPU box), and rank 0's JSON line
comes back on stdout.  Ref: P/pretrain_AnatoMask_DDP.py:192-240 expects an external torchrun; the driver runs `python bench.py`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_timeline(ex):
    """the per-collective record of the N > 1 JSON line (trainer.exchange_timeline): one [issue_ms, done_ms] pair per collective, measured
    from the start of backward, issued in backward-completion order, every one done after it was issued, and the exposed remainder."""
    tl = ex["timeline"]
    pieces = tl["pieces"]
    assert len(pieces) == ex["collectives_per_step"] and sum(p["bytes"] for p in pieces) == ex["gradient_bytes_per_step"]
    issues = [p["issue_ms"] for p in pieces]
    assert issues == sorted(issues) and issues[0] >= 0.0
    assert all(p["done_ms"] is not None and p["done_ms"] >= p["issue_ms"] for p in pieces)
    assert issues[0] <= tl["backward_end_ms"]                  # the exchange starts before backward has ended
    assert tl["exposed_ms"] is not None and tl["exposed_ms"] >= 0.0 and ex["exposed_ms_last_step"] == tl["exposed_ms"]
    assert abs(tl["exposed_ms"] - max(0.0, max(p["done_ms"] for p in pieces) - tl["backward_end_ms"])) < 2e-3


@pytest.mark.timeout(180)
def test_bench_self_launch_two_ranks_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-launch", "--steps", "3", "--warmup", "1"], capture_output=True,
                       text=True, timeout=170, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # ONE line, from rank 0
    out = json.loads(lines[0])
    ex = out.pop("exchange")
    ranks = out.pop("ranks")
    assert out == {"dry_run": True, "n_gpus": 2, "world": 2, "rank_sum": 1.0, "backend": "gloo", "mean_gradient_ok": True, "ranks_seen": 2}
    assert sorted(r["rank"] for r in ranks) == [0, 1] and len({r["device"] for r in ranks}) == 2 and all(r["window_s"] > 0 for r in ranks)
    # the rank body's N > 1 branches ran for real (gloo): timed window, exchange-off window, the `exchange` record of the JSON line
    assert ex["ranks"] == 2 and ex["backend"] == "gloo" and ex["gradient_bytes_per_step"] > 50e6          # STUNet-S: 13.3 M parameters
    assert ex["collectives_per_step"] >= 4 and ex["largest_collective_bytes"] <= 8 << 20
    assert ex["first_collective_after_tag"].startswith("dec") and ex["ms_per_step_without_exchange"] >= 0
    _check_timeline(ex)


@pytest.mark.timeout(400)
def test_bench_self_launch_eight_ranks_gloo():
    """The shape of the driver's 8-GPU run, on CPU: 8 ranks, every one seen, the mean of 8 different gradients, one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-launch", "--steps", "2", "--warmup", "1"], capture_output=True,
                       text=True, timeout=380, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["world"] == 8 and out["ranks_seen"] == 8 and out["rank_sum"] == 28.0 and out["mean_gradient_ok"]
    assert sorted(r_["rank"] for r_ in out["ranks"]) == list(range(8))
    assert out["exchange"]["ranks"] == 8 and out["exchange"]["collectives_per_step"] >= 4
    _check_timeline(out["exchange"])


def test_launch_command_and_noop_under_a_launcher(monkeypatch):
    from anatomask_amd import launch
    cmd = launch.launch_command(4, "bench.py", ["--gpus", "4", "--steps", "3"], port=12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "3"]
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert not launch.launched()
    monkeypatch.setenv("WORLD_SIZE", "4"); monkeypatch.setenv("RANK", "1")
    assert launch.launched()                               # under torchrun bench.py is a rank, never a launcher
