"""Experiment: is the level loop bound by L2-miss latency of the streamed schedule?  Re-run the SAME sweeps
(schedule stays in L2/MALL) vs fresh sweeps each round."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()

def run(N, R, S, rounds, same, plan_total):
    J, h = make_instance(N)
    eng = P.Engine(J, h, R)
    eng.set_spins(init_spins(R, N))
    tab = np.repeat(np.geomspace(0.05, 4.0, R)[:, None], S, axis=1)
    eng.plan_philox(0, plan_total, 42)
    eng.sweep_philox(S, 42, sweep0=0, beta=tab); eng.energy()
    ms = 0.0
    for r in range(rounds):
        s0 = 0 if same else S * (r + 1)
        eng.sweep_philox(S, 42, sweep0=s0, beta=tab)
        ms += eng.last_timing()["ms_sweep"]
    print(f"N={N} R={R} S={S} same={same} plan={plan_total}: {ms/rounds*1e3/S:.1f} us/sweep  {R*N*S*rounds/(ms*1e-3):.3e} upd/s", flush=True)
    eng.close()

run(10000, 256, 4, 20, True, 4)
run(10000, 256, 4, 20, False, 4 * 22)
run(10000, 256, 1, 20, True, 1)
run(10000, 32, 4, 20, True, 4)
run(10000, 8, 4, 20, True, 4)
run(10000, 1, 4, 20, True, 4)
