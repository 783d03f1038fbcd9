import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()
N, R, S, rounds = 10000, int(os.environ.get("RR", "256")), 10, 10
J, h = make_instance(N)
if os.environ.get("CAPDEG"):
    import scipy.sparse as sp
    cap = int(os.environ["CAPDEG"])
    A = J.tolil()
    for k in range(N):                      # drop the highest-index neighbours of over-full rows (symmetrically)
        while len(A.rows[k]) > cap:
            j = A.rows[k][-1]
            A[k, j] = 0; A[j, k] = 0
            A.rows[k] = [c for c in A.rows[k] if c != j]; A.data[k] = A.data[k][:len(A.rows[k])]
            A.rows[j] = [c for c in A.rows[j] if c != k]; A.data[j] = A.data[j][:len(A.rows[j])]
    J = sp.csr_matrix(A); J.eliminate_zeros()
    print("capped: nnz", J.nnz, "max deg", np.diff(J.indptr).max())
eng = P.Engine(J, h, R)
eng.set_spins(init_spins(R, N))
tab = np.repeat(np.geomspace(0.05, 4.0, R)[:, None], S, axis=1)
eng.plan_philox(0, S * (rounds + 1), 42)
eng.sweep_philox(S, 42, sweep0=0, beta=tab); eng.energy()
ms = 0.0
for r in range(rounds):
    eng.sweep_philox(S, 42, sweep0=S * (r + 1), beta=tab)
    ms += eng.last_timing()["ms_sweep"]
print(f"DBG={os.environ.get('NLMC_DBG','0')} NT={os.environ.get('NLMC_SWEEP_NT','auto')} R={R}: {ms/rounds*1e3/S:.1f} us/sweep", flush=True)
eng.close()
