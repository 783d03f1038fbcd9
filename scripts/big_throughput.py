"""Throughput of the global-memory kernels (csrc/nlmc_big.h: chains too long for LDS): shared-order philox sweeps of R chains."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
for N, R, S in ((30_000, 256, 20), (100_000, 256, 10), (1_000_000, 64, 4), (30_000, 1, 20), (100_000, 1, 10), (1_000_000, 1, 4)):
    J, h = make_instance(N)
    inst = P.Instance(J, h)
    with P.Engine(inst, None, R) as eng:
        eng.set_spins(init_spins(R, N) if N <= 100_000 else np.tile(init_spins(1, N), (R, 1)))
        for prec in ("f32", "f64"):
            eng.sweep_philox(S, 7, sweep0=0, beta=1.0, precision=prec)      # warm
            eng.timing_reset(True)
            t0 = time.perf_counter()
            eng.sweep_philox(S, 7, sweep0=S, beta=1.0, precision=prec)
            eng.energy_tracked()
            dt = time.perf_counter() - t0
            tm = eng.timing_total()
            eng.timing_reset(False)
            st = eng.last_schedule_stats()
            print(f"N={N} R={R} S={S} {prec}: {dt * 1e3:.1f} ms wall ({tm['ms_levelize']:.1f} ms levelize, {tm['ms_sweep']:.1f} ms sweeps; "
                  f"{st['levels'] / max(1, st['orders']):.1f} levels per sweep) -> {R * N * S / dt:.3e} updates/s", flush=True)
