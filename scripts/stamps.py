"""Diagnostic build only (NLMC_LIB=.../libnlmc_hip_stamps.so NLMC_STAMP_FILE=...): where do a wave's cycles go?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, S = 10000, 256, 10
J, h = make_instance(N)
eng = P.Engine(J, h, R)
eng.set_spins(init_spins(R, N))
tab = np.repeat(np.geomspace(0.05, 4.0, R)[:, None], S, axis=1)
eng.plan_philox(0, S * 3, 42)
for r in range(3):
    eng.sweep_philox(S, 42, sweep0=S * r, beta=tab)
eng.energy()
eng.close()
d = np.fromfile(os.environ["NLMC_STAMP_FILE"], dtype=np.int64).reshape(R, 16, 8)
print("inside update_spin, s_memtime cycles per ACTIVE level (median over chains), lane 0 of each wave:")
for w in (0, 1, 4, 8, 12, 15):
    al = np.maximum(d[:, w, 3], 1)
    f = lambda j: np.median(d[:, w, j] / al)
    print(f" wave {w:2d}: lds-gather {f(4):7.0f}  field-fma {f(5):7.0f}  decide(exp2,ur,s[k]) {f(2):7.0f}  energy+write {f(0):7.0f}   update total {f(1):7.0f}")
