"""`python bench.py --gpus N` must start N ranks itself when nobody did (BASELINE config 5 is only measurable that way), as a child
process started before any GPU call -- never by re-executing a process that has initialised the GPU."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def bench(monkeypatch):
    monkeypatch.syspath_prepend(ROOT)
    return importlib.import_module("bench")


def test_gpus_2_builds_a_two_rank_torchrun_launch(bench, monkeypatch):
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    rc = bench.maybe_self_launch(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert rc == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert "--standalone" in cmd and "--master-port" not in cmd               # torchrun binds the rendezvous port itself: no bind-close-reuse race
    assert cmd[cmd.index("--local-addr") + 1] == "127.0.0.1"                  # the container hostname may not resolve
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]          # the ranks get the caller's arguments unchanged
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(env["OMP_NUM_THREADS"]) >= 1
    assert bench.maybe_self_launch(["--gpus=4"]) == 0 and "--nproc-per-node=4" in calls[1][0]


def test_no_self_launch_for_one_gpu_or_inside_a_launch(bench, monkeypatch):
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: pytest.fail("must not spawn"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.maybe_self_launch([]) is None
    assert bench.maybe_self_launch(["--gpus", "1", "--steps", "2"]) is None
    monkeypatch.setenv("WORLD_SIZE", "8")                                             # already one of torchrun's ranks
    assert bench.maybe_self_launch(["--gpus", "8"]) is None


def test_self_launch_happens_before_torch_is_imported():
    """The parent must not have touched the GPU: the launch decision runs above `import torch` in bench.py."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("_rc = maybe_self_launch()") < src.index("\nimport torch\n")


def test_child_exit_code_is_propagated(tmp_path):
    """End to end on CPU: a 2-rank launch whose ranks fail (no GPU here) makes `bench.py --gpus 2` fail with a non-zero code,
    and the parent process itself never needed a GPU to get that far."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert "bench.py needs a GPU" in r.stderr
