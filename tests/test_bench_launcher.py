"""CPU tests of bench.py's launcher: `python bench.py --gpus N` (no WORLD_SIZE) must start N ranks itself, as a child
process composed before anything touches the GPU, and must pass the caller's arguments through unchanged."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_compose_launch_cmd_is_the_drivers_shape():
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.compose_launch_cmd(8, argv, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv, "the ranks must see exactly the caller's arguments"


def test_launch_only_when_no_world_size():
    a = bench.parse(["--gpus", "4"])
    assert bench.needs_launch(a, {})
    assert not bench.needs_launch(a, {"WORLD_SIZE": "4", "RANK": "0"})   # already under torch.distributed.run
    assert not bench.needs_launch(bench.parse([]), {})                    # N = 1 runs in-process
    assert not bench.needs_launch(bench.parse(["--gpus", "1"]), {})


def test_parent_spawns_child_and_returns_its_code(monkeypatch):
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    ran = []
    monkeypatch.setattr(bench, "main_rank", lambda a: ran.append(a))
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 8)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "1", "--master-port", "29444"])
    assert e.value.code == 7, "the parent exits with the child's status"
    assert not ran, "the launching parent must not run a rank (it would touch the GPU)"
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-5:] == ["--gpus", "2", "--steps", "1", "--master-port", "29444"][-5:]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_too_few_gpus_is_an_error(monkeypatch):
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: 1)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert e.value.code == 2


def test_config_defaults():
    a = bench.parse(["--config", "4"])
    assert (a.nq, a.nref, a.d, a.k) == (1_500_000, 12_000_000, 128, 30)
    a = bench.parse(["--config", "5", "--nq", "1000"])
    assert (a.nq, a.nref, a.d, a.k, a.genes) == (1000, 4_000_000, 2, 90, 30_000)
    a = bench.parse([])
    assert (a.config, a.nq, a.nref, a.d, a.k) == (2, 1_500_000, 1_500_000, 50, 30)


def test_gpu_count_never_loads_hip(monkeypatch, tmp_path):
    """The launching parent counts GPUs from the environment / the KFD topology files: it must not import torch or call HIP
    (a parent that has initialised the GPU and then spawns ranks is what the pool forbids)."""
    import subprocess

    code = ("import sys, bench; bench.visible_gpu_count(); "
            "assert 'torch' not in sys.modules, 'visible_gpu_count imported torch'; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3,5")
    assert bench.visible_gpu_count() == 3
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "1")
    assert bench.visible_gpu_count() == 1
