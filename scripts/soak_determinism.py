"""Long determinism screen of the bench workload (fused windows + swaps): two runs of ROUNDS rounds, same bits."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, T, ROUNDS, SEED = 10_000, int(os.environ.get("R", 256)), 10, int(os.environ.get("ROUNDS", 600)), 0xA5A50000
PAIRS = round(0.3 * R)
J, h = make_instance(N)
res = []
for rep in range(2):
    with P.Engine(J, h, R) as eng:
        eng.set_spins(init_spins(R, N)); eng.pt_init(np.geomspace(0.05, 4.0, R))
        pl = P.engine.RoundPlanner(eng, 0, ROUNDS, T, SEED, budget_bytes=16 << 30)
        eng.pt_plan(0, ROUNDS, SEED, PAIRS)
        for r in range(ROUNDS):
            pl.sweep(r)
            eng.pt_swap_philox(r, SEED, PAIRS, want_log=False)
        spins = eng.get_spins()
        res.append((spins, eng.energy(), eng.energy_of(spins), eng.pt_slots()))
a, b = res
print("rounds", ROUNDS, "same spins", np.array_equal(a[0], b[0]), "same slots", np.array_equal(a[3], b[3]),
      "tracked == recomputed", np.array_equal(a[1], a[2]), "min E", a[1].min())
