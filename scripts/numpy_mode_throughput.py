import os, sys, time, io, contextlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
for N in (1000, 2048):
    J, h = make_instance(N)
    Jd = J.toarray()
    for rep in range(2):
        np.random.seed(1)
        obj = P.NMC(Jd, h)                     # default rng="numpy": bit-exact with the reference's stream
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            M = obj.MCMC(2000, np.sign(np.random.rand(N) - 0.5), 2.0, Jd, h)
        dt = time.perf_counter() - t0
    print(f"numpy-mode MCMC N={N}: 2000 sweeps in {dt:.3f} s = {dt/2000*1e6:.1f} us per sweep = {dt/2000/N*1e9:.1f} ns per spin update", flush=True)
