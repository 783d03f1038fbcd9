"""Fused-window schedule vs the sweep-by-sweep schedule: same bits (spins, tracked energies), and launch time."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, T, W = 10_000, int(os.environ.get("R", 8)), 10, int(os.environ.get("W", 4))
J, h = make_instance(N, seed=20250225)
inst = P.Instance(J, h)
betas = np.geomspace(0.05, 4.0, R)
out = {}
for mode in ("plain", "fused"):
    with P.Engine(inst, None, R) as eng:
        eng.set_spins(init_spins(R, N))
        eng.pt_init(betas)
        if mode == "fused":
            t0 = time.perf_counter()
            k = eng.plan_philox_fused(0, W, T, 99)
            print("windows planned:", k, "in", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
            t0 = time.perf_counter()
            eng.plan_philox_fused(0, W, T, 99)
            print("  second call (buffers allocated):", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
        else:
            t0 = time.perf_counter()
            eng.plan_philox(0, W * T, 99)
            eng.energy()
            print("plain plan in", round((time.perf_counter() - t0) * 1e3, 2), "ms", flush=True)
        eng.timing_reset(True)
        for w in range(W):
            eng.sweep_philox(T, 99, sweep0=w * T, beta=None)
        tm = eng.timing_total()
        st = eng.last_schedule_stats()
        out[mode] = (eng.get_spins(), eng.energy())
        print(mode, "us/launch", tm["ms_sweep"] / tm["launches_sweep"] * 1e3, "levels/sweep", st["levels"] / max(1, st["orders"]), flush=True)
print("spins equal:", np.array_equal(out["plain"][0], out["fused"][0]), "energies equal:", np.array_equal(out["plain"][1], out["fused"][1]))
