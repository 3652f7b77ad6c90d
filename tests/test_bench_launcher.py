"""`python bench.py --gpus N` without a torch.distributed.run launch starts its N ranks by itself (one process per GPU, RCCL), as CHILD processes and
before anything initialises the GPU.  CPU test of the launcher path: the command it forms and the hand-over (no GPU, nothing is really started)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_command_shape():
    import bench

    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], port=29511)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    port = int(bench.launcher_command(2, [])[bench.launcher_command(2, []).index("--master-port") + 1])
    assert 1024 < port < 65536   # a free port is picked when none is given


def test_gpus_n_self_launches_children_before_touching_the_gpu(monkeypatch):
    import subprocess

    import torch

    import bench

    started = {}

    def fake_run(cmd, env=None, **kw):
        started["cmd"], started["env"] = cmd, env

        class R:
            returncode = 0
        return R()

    def no_gpu(*a, **k):
        raise AssertionError("the launching process must not initialise the GPU")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(torch.cuda, "is_available", no_gpu)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert "--nproc-per-node=4" in started["cmd"] and started["cmd"][-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert started["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert started["env"].get("NCCL_MAX_NCHANNELS") == "32"     # RCCL's CU footprint is bounded in the children's environment (ddp.rccl_channel_env)


def test_mismatched_world_size_is_an_error(monkeypatch):
    import bench

    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match" in str(e.value.code)
