"""Fused-window vs sweep-by-sweep schedule across instance sizes (same bits, us per launch)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
T, W = 10, int(os.environ.get("W", 20))
for N, R in ((300, 1024), (1000, 256), (1000, 1024), (1600, 512), (2048, 512), (4096, 256), (7000, 256), (10000, 256)):
    J, h = make_instance(N, seed=5)
    inst = P.Instance(J, h)
    res = {}
    for mode in ("plain", "fused"):
        with P.Engine(inst, None, R) as eng:
            eng.set_spins(init_spins(R, N)); eng.pt_init(np.geomspace(0.05, 4.0, R))
            k = eng.plan_philox_fused(0, W, T, 7) if mode == "fused" else eng.plan_philox(0, W * T, 7)
            eng.timing_reset(True)
            for w in range(W):
                eng.sweep_philox(T, 7, sweep0=w * T, beta=None)
            tm = eng.timing_total(); st = eng.last_schedule_stats()
            res[mode] = (eng.get_spins(), eng.energy(), tm["ms_sweep"] / tm["launches_sweep"] * 1e3, st["levels"] / st["orders"], k)
    a, b = res["plain"], res["fused"]
    print(f"N={N} R={R}: plain {a[2]:.1f} us ({a[3]:.1f} lv/sweep)  fused {b[2]:.1f} us ({b[3]:.1f} lv/sweep, planned {b[4]})  "
          f"same bits: {np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])}", flush=True)
