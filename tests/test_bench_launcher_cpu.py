"""`python bench.py --gpus N` (N > 1, no RANK in the environment) starts the N ranks itself: one child process group
through torch.distributed.run on 127.0.0.1, rank 0's JSON line forwarded, the child's exit code returned, and the
parent never initialises the GPU."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_torchrun_one_process_per_gpu():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_gpus_n_without_rank_launches_children_and_forwards_one_line(monkeypatch, capsys):
    calls = []
    line = json.dumps({"metric": "frame-pairs/sec (match+triangulate+PnP-RANSAC)", "value": 1.0, "n_gpus": 2})

    def fake_run(cmd, stdout=None, env=None):
        calls.append((cmd, env))
        return types.SimpleNamespace(returncode=0, stdout=("NCCL version banner\n" + line + "\n").encode())

    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "Rank", lambda *a, **k: pytest.fail("the parent must not become a rank / touch the GPU"))
    rc = bench.main(["--gpus", "2", "--steps", "2"])
    out = capsys.readouterr()
    assert rc == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert "--nproc-per-node=2" in cmd and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out.out.strip() == line                       # exactly the one JSON line on stdout
    assert "NCCL version banner" in out.err


def test_child_failure_is_the_parents_exit_code(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(subprocess, "run", lambda cmd, stdout=None, env=None: types.SimpleNamespace(returncode=3, stdout=b""))
    assert bench.main(["--gpus", "8"]) == 3


def test_as_a_rank_it_does_not_launch(monkeypatch):
    """Under an external launcher (RANK set) bench.py is one of the ranks: no child processes."""
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: pytest.fail("a rank must not launch ranks"))
    seen = {}

    class FakeRank:
        def __init__(self, args): seen["gpus"] = args.gpus; self.rank = 1; self.world = 2
        def close(self): seen["closed"] = True

    monkeypatch.setattr(bench, "Rank", FakeRank)
    monkeypatch.setattr(bench, "run_trajectory", lambda args, rk: seen.setdefault("ran", (args.frames, args.features)))
    assert bench.main(["--gpus", "2"]) == 0
    assert seen == {"gpus": 2, "ran": (3450, 2000), "closed": True}
