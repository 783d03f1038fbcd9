"""Timing of k_rounds_fused (many rounds per launch) against k_sweep_fused + k_pt_swap (a launch per round) on the bench shape: us per
round by HIP events around the whole loop.  PRECISION, ROUNDS (per launch), NLMC_LIB / NLMC_DBG_FLAGS as for kloop.py."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, T, ROUNDS = (int(os.environ.get(k, d)) for k, d in (("N", 10000), ("R", 256), ("T", 10), ("ROUNDS", 256)))
PREC, PAIRS = os.environ.get("PRECISION", "f64"), 77
J, h = make_instance(N)
for mode in ("persistent", "deferred", "per-round"):
    with P.Engine(J, h, R) as eng:
        eng.set_spins(init_spins(R, N)); eng.pt_init(np.geomspace(0.05, 4.0, R))
        assert eng.plan_philox_fused(0, 2 * ROUNDS, T, 7) == 2 * ROUNDS
        eng.pt_plan(0, 2 * ROUNDS, 7, PAIRS)
        for rep in range(2):                      # first pass = warm-up
            eng.timing_reset(True, every=1 if mode == "persistent" else 8)
            eng.energy(); t0 = time.perf_counter()
            if mode == "persistent":
                assert eng.pt_rounds_fused(ROUNDS, T, 7, rep * ROUNDS * T, rep * ROUNDS, PAIRS, precision=PREC), eng.rounds_fused_refusal
            elif mode == "deferred":
                assert eng.pt_rounds_deferred(ROUNDS, T, 7, rep * ROUNDS * T, rep * ROUNDS, PAIRS, precision=PREC), eng.rounds_fused_refusal
            else:
                for r in range(rep * ROUNDS, (rep + 1) * ROUNDS):
                    eng.sweep_philox(T, 7, sweep0=r * T, beta=None, precision=PREC)
                    eng.pt_swap_philox(r, 7, PAIRS, want_log=False)
            eng.energy(); dt = time.perf_counter() - t0
            tm = eng.timing_total()
        print(f"{os.environ.get('TAG', '')} {PREC} {mode}: {dt / ROUNDS * 1e6:.1f} us per round wall, kernel {tm['ms_sweep'] / max(1, tm['launches_timed']) * 1e3:.1f} us per round "
              f"(events), E_min {eng.energy().min():.0f}", flush=True)
