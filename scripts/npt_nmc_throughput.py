"""C3 as titled in BASELINE.json ("NPT (APT + NMC)"): N = 10^3, 32-rung ladder, the 8 coldest replicas run NMC cycles
(backbone inference + 3 phases per swap round), the others plain sweeps; 10^4 sweeps, 100 swap rounds; philox mode."""
import os, sys, time, contextlib, io, cProfile, pstats
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N, R = 1000, 32
J, h = make_instance(N)
betas = np.geomspace(0.1, 3.0, R)
doNMC = [False] * (R - 8) + [True] * 8
for lbp in ("device", "host"):
    obj = P.NPT(J, h, rng="philox", seed=5, lbp=lbp)
    pr = cProfile.Profile()
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable()
        M, E = obj.run(betas, R, doNMC, num_sweeps_MCMC=10000, num_sweeps_read=10000, num_swap_attempts=100,
                       num_swapping_pairs=10, num_cycles=1, global_beta=3.0, lambda_start=3.0)
        pr.disable()
    dt = time.perf_counter() - t
    print(f"lbp={lbp}: {dt:.2f} s, {R * N * 10000 / dt:.3e} spin-updates/s, min E {E.min():.1f}", flush=True)
    if lbp == "device":
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(14); print(s.getvalue()[:3500])
