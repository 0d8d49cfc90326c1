"""bench.py's own launcher (CPU): `python bench.py --gpus N` without WORLD_SIZE starts torch.distributed.run as a CHILD
process with the driver's own argument form, relays rank 0's JSON line to stdout and everything else to stderr, and
returns the child's exit code.  (The run itself needs GPUs: tests/test_gpu_pipeline.py::test_bench_starts_its_own_ranks.)"""
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _FakeProc:
    def __init__(self, lines, rc):
        self.stdout = iter(lines)
        self._rc = rc

    def wait(self):
        return self._rc


@pytest.mark.parametrize("rc", [0, 7])
def test_spawn_ranks_command_and_relay(monkeypatch, rc):
    import subprocess
    import bench
    seen = {}

    def fake_popen(cmd, env=None, stdout=None, text=None):
        seen.update(cmd=cmd, env=env)
        return _FakeProc(["W0000 some launcher chatter\n", '{"metric": "x", "n_gpus": 4}\n'], rc)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    out, err = io.StringIO(), io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    monkeypatch.setattr(sys, "stderr", err)
    got = bench.spawn_ranks(4)
    assert got == rc
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]          # the same arguments, after the script
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert out.getvalue() == '{"metric": "x", "n_gpus": 4}\n' and "chatter" in err.getvalue()


def test_main_spawns_only_without_a_launcher(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: calls.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [8]
