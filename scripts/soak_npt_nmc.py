"""Determinism screen of the C3-as-titled path (NPT rounds whose 8 coldest slots run NMC_task, RESTARTS ladders batched): two runs,
then the same ladders in CONTEXTS contexts on the one GPU -- restart energies, final slots and swap logs must be the same bits."""
import os, sys, io, contextlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from conftest import load_product
from helpers import make_instance
P = load_product()
N, R = 1000, 32
NR, K = int(os.environ.get("RESTARTS", 16)), int(os.environ.get("CONTEXTS", 4))
J, h = make_instance(N)
betas = np.geomspace(0.1, 3.0, R)
doNMC = [False] * (R - 8) + [True] * 8


def run(device_ids=None):
    obj = P.NPT(J, h, rng="philox", seed=5)
    with contextlib.redirect_stdout(io.StringIO()):
        M, E = obj.run(betas, R, doNMC, num_sweeps_MCMC=4000, num_sweeps_read=4000, num_swap_attempts=40, num_swapping_pairs=10,
                       num_cycles=1, global_beta=3.0, lambda_start=3.0, num_restarts=NR, device_ids=device_ids, return_trace="int8")
    return M, E, obj.restart_energies, obj.final_slots, obj.swap_log_all


a, b, c = run(), run(), run([0] * K)
same = lambda x, y: all(np.array_equal(p, q) for p, q in zip((x[0], x[1], x[2], x[3], *x[4]), (y[0], y[1], y[2], y[3], *y[4])))
print(f"restarts {NR}: run == rerun {same(a, b)}; one context == {K} contexts {same(a, c)}; best energy {a[2].min():.1f}")
