"""Throughput seen by a user of the drop-in class (not the bench.py headline): NPT(J, h, rng="philox").run(...)."""
import os, sys, time, io, contextlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
for (N, R, sweeps, rounds, pairs) in ((10_000, 256, 1000, 100, 77), (1000, 32, 10_000, 100, 10)):
    J, h = make_instance(N)
    for trace in ("float64", "int8", None):
        for rep in range(3):                                  # third call: library, caches and the host allocator are warm
            obj = P.NPT(J, h, rng="philox", seed=1)
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = time.perf_counter()
                M, E = obj.run(np.geomspace(0.05, 4.0, R), R, [False] * R, num_sweeps_MCMC=sweeps, num_sweeps_read=sweeps,
                               num_swap_attempts=rounds, num_swapping_pairs=pairs, return_trace=trace)
                dt = time.perf_counter() - t0
        print(f"NPT.run philox N={N} R={R} sweeps={sweeps} rounds={rounds} return_trace={trace}: {dt:.3f} s wall incl. instance "
              f"upload, planning, read-out -> {R * N * sweeps / dt:.3e} updates/s ; min energy {E.min():.1f} ; swaps accepted "
              f"{obj.swap_accepted.mean():.2f}", flush=True)

# APT_ICM(J, h, rng="philox").run(icm_feedback=True) at the C5 shape: 32 temperatures x 8 sub-replicas, 100 rounds of 10 sweeps
N, R, K, sweeps, rounds, pairs = 10_000, 32, 8, 1000, 100, 10
J, h = make_instance(N)
for trace in ("float64", None):
    for rep in range(3):
        obj = P.APT_ICM(J, h, rng="philox", seed=1)
        obj.num_subreplicas = K
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            M, E = obj.run(np.geomspace(0.05, 4.0, R), R, num_sweeps_MCMC=sweeps, num_sweeps_read=sweeps, num_swap_attempts=rounds,
                           num_swapping_pairs=pairs, icm_feedback=True, return_trace=trace)
            dt = time.perf_counter() - t0
    print(f"APT_ICM.run philox N={N} R={R} x {K} sub-replicas sweeps={sweeps} rounds={rounds} return_trace={trace}: {dt:.3f} s wall "
          f"-> {R * K * N * sweeps / dt:.3e} updates/s ; min energy {E.min():.1f}", flush=True)

# NMC(J, h, rng="philox").run(): single chain, the reference's headline call (NMC/examples/general_example.py)
for (N, s0, s, cycles) in ((1000, 1000, 1000, 4), (10_000, 1000, 1000, 2)):
    J, h = make_instance(N)
    obj = P.NMC(J, h, rng="philox", seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        M, E, emin = obj.run(s0, s, cycles, 1, 1, 20, 3, 3, 0.01, 0.9, 0.9999999, 0.999999, 100, np.finfo(float).eps)
        dt = time.perf_counter() - t0
    tot = s0 + 3 * cycles * s
    print(f"NMC.run philox N={N}: {tot} sweeps of one chain + {cycles} backbone inferences in {dt:.3f} s "
          f"({dt / tot * 1e6:.1f} us per sweep all in) ; min energy {emin:.1f}", flush=True)
