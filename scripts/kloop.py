"""Kernel-loop timing of the fused-window sweep kernel on the bench workload (N = 10^4, 256 chains, 10 sweeps per
launch): us per launch by HIP events.  Knobs: NLMC_LIB (variant build), NLMC_FUSED_WORKERS, W (windows), N, R, T,
GAUSS=1 (Gaussian couplings: 8-byte schedule entries), INT3=1 (couplings in +-{1,2,3}: 4-byte entries), PRECISION=f64 (the fp64 mode on
the same windows)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, T, W = (int(os.environ.get(k, d)) for k, d in (("N", 10000), ("R", 256), ("T", 10), ("W", 40)))
PREC = os.environ.get("PRECISION", "f32")
J, h = make_instance(N, gaussian=bool(int(os.environ.get("GAUSS", "0"))))
if os.environ.get("INT3"):                       # couplings in +-{1,2,3}: the 4-byte entry format
    J = J.copy(); J.data = J.data * (1 + (np.arange(J.nnz) % 3)); J = ((J + J.T) / 2).tocsr(); J.data = np.sign(J.data) * np.ceil(np.abs(J.data))
inst = P.Instance(J, h)
with P.Engine(inst, None, R) as eng:
    eng.set_spins(init_spins(R, N)); eng.pt_init(np.geomspace(0.05, 4.0, R))
    k = eng.plan_philox_fused(0, W, T, 7)
    for w in range(3):
        eng.sweep_philox(T, 7, sweep0=w * T, beta=None, precision=PREC)
    eng.timing_reset(True)
    for w in range(3, W):
        eng.sweep_philox(T, 7, sweep0=w * T, beta=None, precision=PREC)
    tm = eng.timing_total(); st = eng.last_schedule_stats()
    us = tm["ms_sweep"] / tm["launches_sweep"] * 1e3
    print(f"{os.environ.get('TAG', '')} {PREC} N={N} R={R} T={T}: {us:.1f} us/launch  {R * N * T / us * 1e6:.3e} upd/s  "
          f"{st['levels'] / st['orders']:.2f} lv/sweep  planned {k}  E_min {eng.energy().min():.0f}", flush=True)
