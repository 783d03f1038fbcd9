"""Randomised differential test of the fused-window kernels against the sweep-by-sweep kernels (which the oracle tests
pin): random graphs (degree mixes incl. hubs and isolated spins), the three schedule entry formats (+-J / small integers /
Gaussian), integer or real fields, self-couplings, phase flags, with and without per-sweep outputs, both arithmetic modes (the
fp64 mode fuses on +-J / integer instances only; on some of those cases the test knob NLMC_F64_TIE_MASK widens its exact path).  Same bits or it
prints the failing case.  CASES (default 60), SEED."""
import os, sys
import numpy as np
import scipy.sparse as sp
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import init_spins
P = load_product()
CASES, SEED0 = int(os.environ.get("CASES", 60)), int(os.environ.get("SEED", 1))
bad = 0
for case in range(CASES):
    rng = np.random.default_rng(SEED0 * 1000 + case)
    n = int(rng.integers(256, 3000))
    mean_deg = float(rng.choice([3, 6, 10, 14]))
    m = int(n * mean_deg / 2)
    i = rng.integers(0, n, m); j = rng.integers(0, n, m)
    if rng.random() < 0.5:                                     # hubs
        for hub in rng.choice(n, 3, replace=False):
            d = int(rng.integers(17, 90))
            i = np.concatenate([i, np.full(d, hub)]); j = np.concatenate([j, rng.integers(0, n, d)])
    keep = i != j
    A = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
    A = sp.triu(((A + A.T) > 0).astype(np.float64), 1).tocsr()
    kind = str(rng.choice(["pmj", "int", "gauss"]))
    A.data = {"pmj": lambda: rng.choice([-1.0, 1.0], A.nnz), "int": lambda: rng.choice([-3.0, -2.0, -1.0, 1.0, 2.0, 3.0], A.nnz),
              "gauss": lambda: rng.normal(0, 1, A.nnz)}[kind]()
    A = (A + A.T).tolil()
    diag = rng.random() < 0.3
    if diag:
        for k in rng.choice(n, 20, replace=False):
            A[k, k] = rng.choice([-1.0, 1.0]) if kind != "gauss" else rng.normal()
    A = A.tocsr(); A.sort_indices()
    h = rng.integers(-1, 2, n).astype(float) if (kind != "gauss" and rng.random() < 0.5) else (np.zeros(n) if kind != "gauss" else rng.normal(0, 0.3, n))
    R, T, W = int(rng.integers(1, 5)), int(rng.integers(3, 9)), int(rng.integers(1, 4))
    use_flags, outs = rng.random() < 0.4, rng.random() < 0.5
    flags = rng.choice([0, 0, 0, 1, 2, 3], size=(R, n)).astype(np.uint8) if use_flags else None
    beta = np.repeat(np.geomspace(0.2, 2.5, R)[:, None], T * W, axis=1)
    per_sweep = rng.random() < 0.4                             # a temperature per sweep (fp64 mode: runs sweep by sweep then)
    if per_sweep:
        beta = beta * np.linspace(0.5, 1.0, T * W)[None, :]
    inst = P.Instance(A, h)
    prec = "f64" if (kind != "gauss" and rng.random() < 0.5) else "f32"
    tie = str(rng.choice(["", "", "0xFFFF0000", "0"])) if prec == "f64" else ""
    os.environ.pop("NLMC_F64_TIE_MASK", None)
    if tie:
        os.environ["NLMC_F64_TIE_MASK"] = tie                  # read at Engine creation
    res = []
    for fused in (True, False):
        with P.Engine(inst, None, R) as eng:
            eng.set_spins(init_spins(R, n))
            if flags is not None:
                eng.set_flags(flags, 7.0)
            planned = eng.plan_philox_fused(0, W, T, 99) if fused else 0
            kw = dict(record_stride=2, want_energy=True, want_min=True, want_state=True) if outs else {}
            if outs:
                o = eng.sweep_philox(T * W, 99, sweep0=0, beta=beta, precision=prec, **kw)
            else:
                o = None
                for w in range(W):
                    eng.sweep_philox(T, 99, sweep0=w * T, beta=beta[:, w * T:(w + 1) * T], precision=prec)
            res.append((eng.get_spins(), eng.energy_tracked(), o, planned, eng.last_schedule_stats()["orders"]))
    a, b = res
    ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    if outs:
        ok = ok and all(np.array_equal(a[2][k], b[2][k]) for k in ("spins", "energy", "min_energy", "argmin", "argmin_state"))
    used = a[3] == W and a[4] == T
    print(f"case {case}: n={n} deg~{mean_deg} {kind} diag={diag} h={'int' if h.any() and kind != 'gauss' else ('real' if h.any() else '0')} "
          f"{prec}{'/tie=' + tie if tie else ''} R={R} T={T} W={W} temps={'per-sweep' if per_sweep else 'per-chain'} flags={use_flags} outs={outs} fused_used={used} -> {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
