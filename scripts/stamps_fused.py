"""Diagnostic build only (NLMC_LIB=.../libnlmc_hip_stamps.so NLMC_STAMP_FILE=...): cycle split of k_sweep_fused."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, T, W = (int(os.environ.get(k, d)) for k, d in (('N', 10000), ('R', 256), ('T', 10), ('W', 6)))
J, h = make_instance(N)
eng = P.Engine(J, h, R)
eng.set_spins(init_spins(R, N)); eng.pt_init(np.geomspace(0.05, 4.0, R))
assert eng.plan_philox_fused(0, W, T, 7) == W
for w in range(W):
    eng.sweep_philox(T, 7, sweep0=w * T, beta=None)
nl = eng.last_schedule_stats()["levels"]
eng.energy()
eng.close()
raw = np.fromfile(os.environ["NLMC_STAMP_FILE"], dtype=np.int64)
d = raw[:R * 16 * 8].reshape(R, 16, 8)
lv = raw[R * 16 * 8:]
print(f"{nl} levels in the window; chain 0, wave 0: barrier-to-barrier cycles of the first 48 levels (width:cycles):")
print("  " + "  ".join(f"{int(lv[48 + i])}:{lv[i]:.0f}" for i in range(48) if lv[48 + i] > 0))
print("per wave (median over chains), cycles per level: work (fetch issue + threshold production + update) vs barrier wait")
for w in range(16):
    m = lambda j: np.median(d[:, w, j])
    print(f" wave {w:2d}: chunks {m(4):5.0f}  | per level: work {m(5) / nl:6.0f}  barrier-wait {m(6) / nl:6.0f}  | kernel total {m(7):9.0f} cycles = {m(7) / nl:6.0f} per level")
