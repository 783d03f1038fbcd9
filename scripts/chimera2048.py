"""Real-instance sanity: Chimera-2048 droplet instance 001, listed optimum -3336.773333."""
import os, sys, time, contextlib, io
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product, GOLDEN
P = load_product()
d = os.path.join(GOLDEN, "instances")
W, h = P.instances.txt_to_A_droplet(os.path.join(d, "chimera2048__001.txt"))
tok = open(os.path.join(d, "chimera2048__groundstate_001.txt")).read().split()
e_gs = float(tok[2]); s = (2 * np.array(tok[3:3 + 2048], dtype=int) - 1).astype(np.int8)
J = -W; hh = -np.asarray(h).reshape(-1); nf = abs(J).max()
with P.Engine(P.Instance(J / nf, hh / nf), None, 1) as eng:
    print("listed", e_gs, "evaluated", eng.energy_of(s[None])[0] * nf)
for R, sweeps, rounds, bmax in ((32, 20000, 2000, 30.0), (32, 100000, 10000, 30.0), (48, 400000, 40000, 40.0)):
    betas = np.geomspace(0.5, bmax, R)
    obj = P.APT_ICM(J / nf, hh / nf, rng="philox", seed=3)
    t = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        M, E = obj.run(betas, R, num_sweeps_MCMC=sweeps, num_sweeps_read=sweeps, num_swap_attempts=rounds, num_swapping_pairs=R // 3, icm_feedback=True)
    print("ICM", R, sweeps, rounds, bmax, "best", E.min() * nf, "gap", E.min() * nf - e_gs, "acc", obj.swap_accepted.mean(), "t", time.perf_counter() - t, flush=True)
