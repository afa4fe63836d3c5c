"""bench.py's self-launch path without a GPU: `--gpus N --dry-run` prints the torch.distributed.run child command that would start
the ranks (one rank per GPU, 127.0.0.1 rendezvous) and the configurations an N > 1 run measures, and exits before any GPU call."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dry(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--dry-run"] + list(extra), capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_2_dry_run_prints_the_child_command():
    d = _dry()
    cmd = d["cmd"]
    assert d["dry_run"] is True and d["ranks"] == 2
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[cmd.index("--master-port") + 2] == os.path.join(ROOT, "bench.py")
    tail = cmd[cmd.index("--master-port") + 3:]
    assert tail == ["--gpus", "2", "--steps", "4", "--warmup", "1"]          # --dry-run is not forwarded to the ranks
    assert d["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # a multi-GPU run measures the weak-scaling headline only ...
    assert d["configs_measured"] == ["2"]
    # ... unless --others asks for BASELINE.json's DDP configurations (4 = joint trainer, 5 = dual_gan fp8) behind it
    d = _dry("--others")
    assert d["configs_measured"] == ["2", "4a", "4b", "5"] and d["cmd"][-1] == "--others"


def test_dry_run_with_a_single_config():
    d = _dry("--config", "5", "--no-others")
    assert d["configs_measured"] == ["5"]
    assert d["cmd"][-2:] == ["5", "--no-others"]
