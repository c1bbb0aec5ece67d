"""bench.py --gpus N: the contract with the driver (`python bench.py --gpus N` alone must run N ranks; under a launcher the
two numbers must agree; fewer devices than N is an error, not a silent N = 1 line)."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(gpus):
    return argparse.Namespace(gpus=gpus)


def test_single_gpu_needs_no_launcher(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.launch_ranks_if_needed(_args(1)) is None


def test_launcher_world_size_must_match(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.launch_ranks_if_needed(_args(4)) is None          # under torch.distributed.run: nothing to do
    with pytest.raises(SystemExit, match="--gpus 2 but the launcher started WORLD_SIZE=4"):
        bench.launch_ranks_if_needed(_args(2))


def test_fewer_devices_than_ranks_is_loud(monkeypatch):
    import torch

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("BENCH_DEVICE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="needs 8 devices, this box has 1"):
        bench.launch_ranks_if_needed(_args(8))


def test_self_launch_command(monkeypatch):
    """N > 1 without a launcher: re-exec under torch.distributed.run with one process per GPU on 127.0.0.1."""
    import subprocess

    import torch

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks_if_needed(_args(4))
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
