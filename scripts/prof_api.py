"""cProfile of `NPT(J, h, rng="philox").run` at the C4 shape, once per return_trace mode, under the conditions a caller
sees: results of the previous call still alive (fresh host memory for the next M)."""
import os, sys, io, contextlib, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N, R, sweeps, rounds, pairs = 10_000, 256, 1000, 100, 77
J, h = make_instance(N)
def go(trace):
    obj = P.NPT(J, h, rng="philox", seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        return obj.run(np.geomspace(0.05, 4.0, R), R, [False] * R, num_sweeps_MCMC=sweeps, num_sweeps_read=sweeps,
                       num_swap_attempts=rounds, num_swapping_pairs=pairs, return_trace=trace)
keep = []
for trace in ("float64", "int8", None):
    keep.append(go(trace))
    pr = cProfile.Profile(); pr.enable(); keep.append(go(trace)); pr.disable()
    print(f"==== return_trace={trace}")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(10)
    print("\n".join(l[:160] for l in s.getvalue().splitlines() if l.strip()))
