"""k_sweep_fused with and without per-sweep outputs (running minimum + argmin state) on the same windows: us per launch
by HIP events.  N, R, T, W (windows), FLAGS=1 (phase flags on: a random backbone mask, NC phase)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, T, W = int(os.environ.get("N", 1000)), int(os.environ.get("R", 64)), int(os.environ.get("T", 50)), int(os.environ.get("W", 40))
J, h = make_instance(N)
with P.Engine(J, h, R) as eng:
    eng.set_spins(init_spins(R, N))
    assert eng.plan_philox_fused(0, W, T, 3) == W
    if os.environ.get("FLAGS"):
        fl = np.where(np.random.default_rng(1).random((R, N)) < 0.5, 2, 0).astype(np.uint8)
        eng.set_flags(fl, 20.0)
    for name, kw in (("plain", {}), ("want_min + state", dict(want_min=True, want_state=True)), ("+ energy trace", dict(want_min=True, want_state=True, want_energy=True)),
                     ("+ recorded spins", dict(want_min=True, want_state=True, record_stride=1))):
        for rep in range(2):
            eng.timing_reset(True, 1)
            for w in range(W):
                eng.sweep_philox(T, 3, sweep0=w * T, beta=2.0, **kw)
            t = eng.timing_total()
        print(f"N={N} R={R} T={T}: {name:<20} {t['ms_sweep'] * 1e3 / t['launches_timed']:.1f} us per launch, "
              f"{R * N * T * t['launches_timed'] / (t['ms_sweep'] * 1e-3):.3e} updates/s in the kernel", flush=True)
