"""bench.py's launcher contract: `python bench.py --gpus N` with no WORLD_SIZE in the environment must turn itself into
the torch.distributed.run command the driver would have used (one rank per GPU, rendezvous on 127.0.0.1) instead of
exiting -- the first 8-GPU run must not be wasted on a usage message."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_torchrun_command_is_formed():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.torchrun_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], port=29533)
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29533"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    # a port the environment names is honoured (the driver passes --master-port to its own launcher)
    os.environ["MASTER_PORT"] = "29611"
    try:
        assert bench.torchrun_command(2, [])[bench.torchrun_command(2, []).index("--master-port") + 1] == "29611"
    finally:
        del os.environ["MASTER_PORT"]


def test_gpus_n_without_launcher_reexecs(tmp_path):
    """no GPU needed: the re-exec happens before anything is imported; a stub `torch.distributed.run` records its argv"""
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "run.py").write_text("import sys, json\nprint('STUB ' + json.dumps(sys.argv[1:]))\n")
    env = dict(os.environ, PYTHONPATH=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"], env=env,
                         capture_output=True, text=True, timeout=120)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("STUB ")]
    assert line, out.stdout + out.stderr
    argv = json.loads(line[0][5:])
    assert argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[-4:] == ["--gpus", "4", "--steps", "3"]


@pytest.mark.gpu
def test_bench_via_torchrun_at_one_gpu():
    """the same re-exec path on a GPU box at --gpus 1: the JSON line comes out of the re-executed launcher"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--via-torchrun", "--config", "fast",
                          "--steps", "3", "--warmup", "1", "--headline-only", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "re-executing as" in out.stderr, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert lines, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads(lines[-1])
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0
