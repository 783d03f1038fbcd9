"""cProfile of `NMC(J, h, rng="philox").run` (one chain, N = 10^3: the reference's headline call in the device-RNG mode)."""
import os, sys, io, contextlib, cProfile, pstats
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N = int(os.environ.get("N", 1000))
J, h = make_instance(N)
def go():
    obj = P.NMC(J, h, rng="philox", seed=1)
    with contextlib.redirect_stdout(io.StringIO()):
        return obj.run(1000, 1000, 4, 1, 1, 20, 3, 3, 0.01, 0.9, 0.9999999, 0.999999, 100, np.finfo(float).eps)
go()
pr = cProfile.Profile(); pr.enable(); go(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print("\n".join(l[:150] for l in s.getvalue().splitlines() if l.strip()))
