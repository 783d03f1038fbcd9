import os, sys, time, io, contextlib, cProfile, pstats
import numpy as np
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
from conftest import load_product
from helpers import make_instance
P = load_product()
N, R, sweeps, rounds, pairs = 10_000, 256, 1000, 100, 77
J, h = make_instance(N)
TRACE = {"none": None}.get(os.environ.get("TRACE", "float64"), os.environ.get("TRACE", "float64"))
def go():
    obj = P.NPT(J, h, rng="philox", seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        return obj.run(np.geomspace(0.05, 4.0, R), R, [False] * R, num_sweeps_MCMC=sweeps, num_sweeps_read=sweeps, num_swap_attempts=rounds, num_swapping_pairs=pairs, return_trace=TRACE)
go()
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
