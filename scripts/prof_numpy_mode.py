import os, sys, io, contextlib, cProfile, pstats
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N=1000
J, h = make_instance(N); Jd = J.toarray()
obj = P.NMC(Jd, h)
np.random.seed(1)
obj.MCMC(200, np.sign(np.random.rand(N) - 0.5), 2.0, Jd, h)
pr = cProfile.Profile(); pr.enable()
M = obj.MCMC(2000, np.sign(np.random.rand(N) - 0.5), 2.0, Jd, h)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
print("\n".join(l[:150] for l in s.getvalue().splitlines() if l.strip()))
