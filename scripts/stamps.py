"""Diagnostic build only (NLMC_LIB=.../libnlmc_hip_stamps.so NLMC_STAMP_FILE=...): where do a wave's cycles go?
PRECISION=f64: the fp64-field kernel (8 waves)."""
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
PREC = os.environ.get("PRECISION", "f32")
eng.plan_philox(0, S * 3, 42, precision=PREC)
for r in range(3):
    eng.sweep_philox(S, 42, sweep0=S * r, beta=tab, precision=PREC)
eng.energy()
eng.close()
raw = np.fromfile(os.environ["NLMC_STAMP_FILE"], dtype=np.int64)
d = raw[:R * 16 * 8].reshape(R, 16, 8)
lv = raw[R * 16 * 8:]
print("chain 0, wave 0: barrier-to-barrier cycles per level (mean over the launch's sweeps) vs level width:")
print("  " + "  ".join(f"{int(lv[48 + i])}:{lv[i] / S:.0f}" for i in range(48) if lv[48 + i] > 0))
print("inside update_spin, s_memtime cycles per call (median over chains), lane 0 of each wave; sweep totals per wave:")
for w in ((0, 1, 3, 4, 8, 12, 15) if PREC == "f32" else (0, 1, 3, 4, 7)):
    calls = np.maximum(d[:, w, 4], 1)
    f = lambda j: np.median(d[:, w, j] / calls)
    print(f" wave {w:2d}: lds-gather {f(0):6.0f}  field {f(1):6.0f}  decide {f(2):6.0f}  energy+write {f(3):6.0f}  | calls/sweep "
          f"{np.median(d[:, w, 4]) / S:5.1f}  fill {np.median(d[:, w, 5]) / S:6.0f}  epilogue {np.median(d[:, w, 7]) / S:5.0f}  "
          f"kernel cycles/sweep {np.median(d[:, w, 6]) / S:8.0f}")
