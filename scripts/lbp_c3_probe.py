"""Backbone inference at the C3 shape (N = 10^3, 64 seeds = 8 NMC replicas x 8 restarts): time per batch, lambdas and BP
iterations per seed (the slowest seed sets the launch time).  DUMP=dir: per-seed lambdas / iterations / marginals as npz."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
EPS = np.finfo(float).eps
N = int(os.environ.get("N", 1000)); B = int(os.environ.get("B", 64))
J, h = make_instance(N)
inst = P.Instance(J, h)
g = P.lbp.EdgeGraph(inst)
eps = g.epsilon(inst.h)
lam0 = float(os.environ.get("LAM0", 3.0)); beta = float(os.environ.get("BETA", 3.0))
lams = P.lbp.lambda_list(lam0, 0.01, 0.9)
with P.Engine(inst, None, B) as eng:
    eng.set_spins((2 * np.random.default_rng(1).integers(0, 2, size=(B, N)) - 1).astype(np.int8))
    for sw in (0, 100, 1000):
        if sw:
            eng.sweep_philox(sw, 7, sweep0=sw, beta=beta)
        ms = eng.get_spins().astype(np.float64)
        for rep in range(2):
            t = time.perf_counter()
            o = eng.lbp_convexified(ms, eps, lams, beta, EPS, 100, float(np.tanh(19.06)) - EPS)
            dt = time.perf_counter() - t
        if os.environ.get("DUMP"):
            os.makedirs(os.environ["DUMP"], exist_ok=True)
            np.savez(os.path.join(os.environ["DUMP"], f"lbp_probe_sw{sw}.npz"), n_lambdas=o["n_lambdas"], iters=o["iters"], status=o["status"],
                     mag=o["mag"], ms=ms)
        its = np.array([int((o["iters"][p][:o["n_lambdas"][p]] + 1).sum()) for p in range(B)])
        print(f"N={N} seeds={B} after {sw} sweeps: {dt*1e3:.2f} ms; lambdas per seed min/med/max {o['n_lambdas'].min()}/{int(np.median(o['n_lambdas']))}/{o['n_lambdas'].max()} of {len(lams)}; "
              f"BP iterations per seed min/med/max {its.min()}/{int(np.median(its))}/{its.max()}; us per iteration of the slowest {dt*1e6/its.max():.2f}; status {np.unique(o['status'])}", flush=True)
