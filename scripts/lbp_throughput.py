"""Backbone inference: host edge-list LBP (lbp.py) vs the device kernel, N = 10^3 and 10^4."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
import oracle
P = load_product()
EPS = np.finfo(float).eps
for N in (1000, 10000):
    J, h = make_instance(N, seed=3)
    inst = P.Instance(J, h)
    g = P.lbp.EdgeGraph(inst)
    eps = g.epsilon(inst.h)
    csr = oracle.Csr(J)
    ms = []
    for p in range(4):
        s = np.where(np.random.default_rng(p).random(N) < 0.5, -1, 1).astype(np.int8)
        cb = np.tile(np.array(oracle.cb_pair(3.0)), (100, 1))
        ms.append(oracle.sweeps_philox(csr, h, s, cb, 1, p, want_M=False)[1].astype(float))
    ms = np.stack(ms)
    t = time.perf_counter()
    hc, hm = P.lbp.lbp_convexified(inst, 0.5, 0.01, 0.9, ms[0].copy(), eps, EPS, 100, 0.999999, 0.99999, 2.5, graph=g, want_marginals=True)
    t_host = time.perf_counter() - t
    with P.Engine(inst, None, 1) as eng:
        lams = P.lbp.lambda_list(0.5, 0.01, 0.9)
        eng.lbp_convexified(ms[:1], eps, lams, 2.5, EPS, 100, 1.0)          # graph build + warm-up
        for B in (1, 4, 64, 256):
            mb = np.concatenate([ms] * (B // 4)) if B >= 4 else ms[:1]
            for rep in range(2):                     # (second call: buffers of this batch size exist)
                t = time.perf_counter()
                o = eng.lbp_convexified(mb, eps, lams, 2.5, EPS, 100, float(np.tanh(19.06)) - EPS)
                dt = time.perf_counter() - t
            iters = int((o["iters"][0][:o["n_lambdas"][0]] + 1).sum())
            print(f"N={N} problems={B}: device {dt*1e3:.2f} ms ({dt/B*1e3:.3f} ms/problem, {iters} BP iterations in problem 0, "
                  f"{o['n_lambdas'][0]} lambdas); host 1 problem {t_host*1e3:.1f} ms ({len(hm)} lambdas)", flush=True)
