"""Chunks per level of the fused windows bench.py runs on (N = 10^4, 3N edges, 10 sweeps per window): how wide the levels
are, how many are a single chunk, runs of consecutive single-chunk levels."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N, T, W = int(os.environ.get("N", 10000)), int(os.environ.get("T", 10)), int(os.environ.get("W", 64))
J, h = make_instance(N)
with P.Engine(J, h, 1) as eng:
    assert eng.plan_philox_fused(0, W, T, 0xA5A50000) == W
    rows = [eng.plan_levels(w) for w in range(W)]
nl = np.array([len(r) for r in rows])
allc = np.concatenate(rows)
print(f"N={N} T={T}: {W} windows, levels per window {nl.min()}..{nl.max()} (mean {nl.mean():.1f}), chunks per window {np.mean([r.sum() for r in rows]):.0f}, "
      f"positions used per update {np.mean([r.sum() for r in rows]) * 64 / (N * T):.3f}")
h_ = np.bincount(allc, minlength=17)
print("chunks per level : " + " ".join(f"{k:>5d}" for k in range(1, len(h_))))
print("share of levels  : " + " ".join(f"{100 * h_[k] / len(allc):5.1f}" for k in range(1, len(h_))))
runs = []
for r in rows:
    k = 0
    for v in list(r) + [99]:
        if v == 1:
            k += 1
        else:
            if k:
                runs.append(k)
            k = 0
runs = np.array(runs) if runs else np.zeros(1, int)
print(f"single-chunk levels: {100 * h_[1] / len(allc):.1f} % of all levels, in runs of mean length {runs.mean():.2f} (max {runs.max()}); "
      f"levels of <= 2 chunks: {100 * (h_[1] + h_[2]) / len(allc):.1f} %; <= 4 chunks: {100 * h_[1:5].sum() / len(allc):.1f} %")
print("first window, chunks per level:", " ".join(str(v) for v in rows[0]))
