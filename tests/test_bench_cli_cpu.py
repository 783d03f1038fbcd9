"""bench.py pieces that need no GPU: the self-launcher of `--gpus N` (the ranks are started as a CHILD
`python -m torch.distributed.run` before anything touches the GPU) and the CPU-baseline worker process."""
import json
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=300, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_gpus_n_starts_its_own_ranks_as_a_child_launcher():
    d = run("--gpus", "4", "--steps", "3", "--warmup", "1", "--strong", "--dry-run-launch")
    cmd = d["would_run"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1].isdigit()
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail[:2] == ["--gpus", "4"] and "--strong" in tail and tail[tail.index("--steps") + 1] == "3"


def test_cpu_baseline_worker_is_a_plain_cpu_process():
    d = run("--cpu-worker", "0.3", "--cpu-chain", "2", env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": ""})
    assert d["kind"] == "port" and d["cores"] == 1 and d["unit"] == "spin-updates/s" and d["value"] > 1e5
