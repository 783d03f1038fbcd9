import os, sys, time
import numpy as np
REPO = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
EPS = np.finfo(float).eps
N = 10000
J, h = make_instance(N, seed=3)
inst = P.Instance(J, h)
g = P.lbp.EdgeGraph(inst)
eps = g.epsilon(inst.h)
lams = P.lbp.lambda_list(0.5, 0.01, 0.9)
ms = np.where(np.random.default_rng(0).random((4, N)) < 0.5, -1.0, 1.0)
with P.Engine(inst, None, 1) as eng:
    eng.set_spins(ms[:1].astype(np.int8)); eng.sweep_philox(200, 3, beta=3.0)
    s = eng.get_spins().astype(float)
    ms = np.repeat(s, 4, axis=0)
    eng.lbp_convexified(ms[:1], eps, lams, 2.5, EPS, 100, 1.0)
    for B in (1, 4):
        for rep in range(2):
            t = time.perf_counter()
            o = eng.lbp_convexified(ms[:B], eps, lams, 2.5, EPS, 100, float(np.tanh(19.06)) - EPS)
            dt = time.perf_counter() - t
        it = int((o["iters"][0][:o["n_lambdas"][0]] + 1).sum())
        print(f"GROUP={os.environ.get('NLMC_LBP_GROUP')} P={B}: {dt*1e3:.2f} ms, {it} iterations, {dt*1e6/it:.1f} us/iteration, status {o['status']}", flush=True)
