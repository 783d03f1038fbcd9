"""Host-side (Python) logic that stays on the CPU in the reference too: temperature schedules, phase flags,
stream drawing in the reference's program order.  No spin is updated here."""
import numpy as np

from . import _abi


def beta_schedule(num_sweeps, beta, anneal=False, sweeps_per_beta=1, initial_beta=0):
    """Per-sweep inverse temperature of MCMC() (NMC/nmc.py:56-69).  The index is incremented BEFORE use, so
    linspace's first value is never applied; num_betas = num_sweeps // sweeps_per_beta."""
    run = np.full(int(num_sweeps), float(beta), dtype=np.float64)
    if anneal:
        nb = int(num_sweeps) // int(sweeps_per_beta)
        vals = np.linspace(initial_beta, beta, nb)
        idx = 0
        for jj in range(int(num_sweeps)):
            if jj % sweeps_per_beta == 0 and idx < nb - 1:
                idx += 1
            run[jj] = vals[idx]
    return run


def draw_legacy_stream(n_sweeps, n):
    """One chain's draws in the reference's order: per sweep np.random.permutation(N) then N x np.random.rand()
    (NMC/nmc.py:71,87).  Consumes the global legacy NumPy stream exactly like the reference does."""
    perm = np.empty((n_sweeps, n), dtype=np.int32)
    u = np.empty((n_sweeps, n), dtype=np.float64)
    for t in range(n_sweeps):
        perm[t] = np.random.permutation(n)
        u[t] = np.random.rand(n)
    return perm, u


def phase_flags(n, m_init, clusters, phase):
    """Flags realising the three NMC phases (NMC/nmc.py:377-381, 398-401, 419-421):
       'C'  : cluster rows of (J,h) / temp_x, every other spin frozen by h = 10000 * m_init
       'NC' : cluster spins frozen by h = 10000 * m_init
       'ALL': plain (J,h)."""
    fl = np.zeros(n, dtype=np.uint8)
    if phase == "ALL":
        return fl
    m_init = np.asarray(m_init)
    cl = np.asarray(clusters, dtype=np.int64)
    frozen_code = np.where(m_init > 0, _abi.SPIN_FROZEN_UP, _abi.SPIN_FROZEN_DOWN).astype(np.uint8)
    if phase == "C":
        fl[:] = frozen_code
        fl[cl] = _abi.SPIN_SCALED
    elif phase == "NC":
        fl[cl] = frozen_code[cl]
    else:
        raise ValueError(phase)
    return fl
