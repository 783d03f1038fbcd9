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


def test_fused_plans_stay_within_the_memory_budget_for_default_arguments():
    """one planning call for a default-length run (10^4 sweeps: 200 windows of 50) would take ~2 GB at N = 10^3 and
    ~16 GB at N = 10^4, times 30 phase launches in NMC.run_restarts; Engine.sweep_philox_windows plans piece by piece."""
    from conftest import load_product
    P = load_product()
    eng = P.engine
    assert eng.fused_window(10000) == 50
    for n, n_long in ((1000, 40), (10000, 400), (11264, 0)):
        per_window = eng.fused_plan_bytes(n, n_long, 50)
        assert 50 * n * 24 < per_window < 50 * (n + n_long) * 90 + 6 * 2 ** 20       # 24 ... 76 bytes per position + padding
        piece = eng.sweeps_per_plan_piece(per_window, 10000, 50, eng.Engine.FUSED_PLAN_BUDGET)
        assert piece % 50 == 0 and 50 <= piece <= 10000
        assert (piece // 50) * per_window <= eng.Engine.FUSED_PLAN_BUDGET
        # strided records: pieces start on recorded sweeps
        assert eng.sweeps_per_plan_piece(per_window, 10000, 50, eng.Engine.FUSED_PLAN_BUDGET, record_stride=4) % 100 == 0
    # a budget below one window still makes progress, one window at a time
    assert eng.sweeps_per_plan_piece(eng.fused_plan_bytes(10000, 0, 50), 10000, 50, 1) == 50
