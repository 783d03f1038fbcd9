"""C3 as titled in BASELINE.json ("NPT (APT + NMC) ... x 8 restarts"): N = 10^3, 32-rung ladder, the 8 coldest slots run
NMC_task every round (backbone inference + 3 phases), the others plain sweeps; 10^4 sweeps, 100 swap rounds; philox mode,
device-resident.  RESTARTS (default "1,8,32") ladders batched; spin-updates/s counts replicas x spins x num_sweeps_MCMC
like the metric does (the NMC phases' surplus sweeps are not counted).  CONTEXTS=k: the ladders in k contexts on the one GPU
(device_ids=[0] * k: whole ladders per context run out of step).  PROFILE=1: cProfile of the host side of the last run."""
import os, sys, time, contextlib, io, cProfile, pstats
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N, R = int(os.environ.get("N", 1000)), 32
J, h = make_instance(N)
betas = np.geomspace(0.1, 3.0, R)
doNMC = [False] * (R - 8) + [True] * 8
for nr in [int(v) for v in os.environ.get("RESTARTS", "1,8,32").split(",")]:
    for rep in range(2):
        obj = P.NPT(J, h, rng="philox", seed=5)
        pr = cProfile.Profile()
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            pr.enable()
            M, E = obj.run(betas, R, doNMC, num_sweeps_MCMC=10000, num_sweeps_read=10000, num_swap_attempts=100,
                           num_swapping_pairs=10, num_cycles=1, global_beta=3.0, lambda_start=3.0, num_restarts=nr,
                           device_ids=([0] * min(nr, int(os.environ["CONTEXTS"])) if os.environ.get("CONTEXTS") else None),
                           return_trace=os.environ.get("TRACE", "float64") if os.environ.get("TRACE", "float64") != "none" else None)
            pr.disable()
        dt = time.perf_counter() - t
    print(f"num_restarts={nr}" + (f" in {min(nr, int(os.environ['CONTEXTS']))} contexts" if os.environ.get("CONTEXTS") else "") + f": {dt:.3f} s, {nr * R * N * 10000 / dt:.3e} spin-updates/s, best energy over restarts {obj.restart_energies.min():.1f}, "
          f"swap acceptance {obj.swap_accepted.mean():.2f}", flush=True)
if os.environ.get("PROFILE"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(16); print(s.getvalue()[:4000])
