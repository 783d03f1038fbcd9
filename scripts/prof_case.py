"""Small fixed case for rocprofv3: N=1e4, R=256, 10 sweeps per launch, 6 launches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, S, rounds = 10000, 256, 10, 6
J, h = make_instance(N)
eng = P.Engine(J, h, R)
eng.set_spins(init_spins(R, N))
tab = np.repeat(np.geomspace(0.05, 4.0, R)[:, None], S, axis=1)
eng.plan_philox(0, S * rounds, 42)
for r in range(rounds):
    eng.sweep_philox(S, 42, sweep0=S * r, beta=tab)
print(eng.energy()[:3])
eng.close()
