"""Quick throughput probe of the sweep kernel (not the bench contract; see bench.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product  # noqa: E402
from helpers import make_instance, init_spins  # noqa: E402

P = load_product()


def probe(N, R, S, rounds, precision="f32", plan=True, with_energy=False):
    J, h = make_instance(N)
    eng = P.Engine(J, h, R)
    eng.set_spins(init_spins(R, N))
    betas = np.geomspace(0.05, 4.0, R)
    eng.pt_init(betas) if False else None
    tab = np.repeat(betas[:, None], S, axis=1)
    if plan:
        eng.plan_philox(0, S * (rounds + 2), 42)
    # warmup
    eng.sweep_philox(S, 42, sweep0=0, beta=tab, precision=precision, want_energy=with_energy)
    eng.energy()
    t0 = time.perf_counter()
    ms_k = 0.0
    ms_l = 0.0
    for r in range(rounds):
        eng.sweep_philox(S, 42, sweep0=S * (r + 1), beta=tab, precision=precision, want_energy=with_energy)
        tm = eng.last_timing()
        ms_k += tm["ms_sweep"]
        ms_l += tm["ms_levelize"]
    eng.energy()
    dt = time.perf_counter() - t0
    st = eng.last_schedule_stats()
    upd = R * N * S * rounds
    print(f"N={N} R={R} S={S} rounds={rounds} {precision} plan={plan} NT={os.environ.get('NLMC_SWEEP_NT','auto')}: "
          f"wall {upd / dt:.3e} upd/s | kernel {upd / (ms_k * 1e-3):.3e} upd/s ({ms_k / rounds * 1e3:.1f} us/launch, "
          f"levelize {ms_l / rounds * 1e3:.1f} us) | levels/sweep {st['levels'] / max(1, st['orders']):.1f} "
          f"| alg GB/s {upd / (ms_k * 1e-3) * 63 / 1e9:.0f}", flush=True)
    eng.close()


if __name__ == "__main__":
    probe(10000, 256, 10, 20)
    probe(10000, 256, 10, 20, plan=False)
    probe(10000, 256, 10, 20, precision="f64")
    probe(10000, 256, 100, 3)
    probe(1000, 256, 100, 5)
    probe(1000, 64, 100, 5)
    probe(10000, 1024, 10, 5)
    probe(10000, 32, 10, 20)
