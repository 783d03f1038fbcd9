"""TEST / BASELINE INFRASTRUCTURE ONLY -- the reference's CPU path with its own cost structure, restated in NumPy.

`oracle/nlo.c` states the same algorithm the way a compiled sequential program would (one row of J per update).
The reference itself pays, per SPIN update, a tuple of the whole state, a full sparse mat-vec and NumPy/SciPy call
overhead (NMC/nmc.py:70-88).  BASELINE.json asks for "the reference NumPy CPU path timed on the same box's host cores";
the reference cannot travel to the GPU box, so bench.py times this restatement beside the C port.  It draws from the
global legacy NumPy stream in the reference's order (one permutation per sweep, one rand() per update), so under
np.random.seed it returns the reference's matrix M bit for bit (tests/test_oracle_golden.py pins it on a golden).
"""
import numpy as np
from scipy.sparse import csr_matrix


def mcmc_numpy_path(num_sweeps, m_start, beta, J, h):
    """Fixed-beta MCMC of NMC/nmc.py:28-91 with the reference's per-update work.  Returns M [N, num_sweeps]."""
    N = J.shape[0]
    state = np.array(m_start, dtype=np.float64).reshape(-1, 1)
    out = np.zeros((N, num_sweeps))
    A = csr_matrix(J)                                   # conversion on every call, like the reference (:53)
    field = np.array(h, dtype=np.float64).reshape(-1, 1)
    for sweep in range(num_sweeps):
        order = np.random.permutation(N)
        for spin in order:
            _key = tuple(state.ravel())                 # built unconditionally by the reference (:73), O(N) per update
            local = A.dot(state) + field                # full mat-vec for one component (:86)
            state[spin] = np.sign(np.tanh(beta * local[spin]) - 2 * np.random.rand() + 1)
        out[:, sweep] = state.ravel()
    return out
