"""oracle/refport.py -- TEST INFRASTRUCTURE ONLY.

Sequential CPU restatement of the reference's drivers around the C sweep (oracle/nlo.c).  One chain at a time,
dense N x N bookkeeping, global legacy `np.random` / stdlib `random` streams consumed in the reference's program
order -- so that, after `np.random.seed(s); random.seed(s)`, every output equals the reference's bit for bit
(pinned by tests/test_oracle_golden.py against tests/golden/*.npz).

Each method cites the reference lines it restates (paths relative to the reference checkout).  Plotting,
printing, the LRU hash table and the process pool are not restated: the pool is
replaced by in-order execution, which is the deterministic reading of the reference.
"""
import random as _pyrandom

import numpy as np

from . import Csr, sweeps_stream, clusters as _c_clusters

EPS = np.finfo(float).eps
FREEZE = 10000  # NMC/nmc.py:381,401


def _beta_schedule(num_sweeps, beta, anneal, sweeps_per_beta, initial_beta):
    """NMC/nmc.py:56-69 -- note the pre-incremented index: beta_vals[0] is never used."""
    run = np.full(num_sweeps, float(beta))
    if anneal:
        nb = num_sweeps // sweeps_per_beta
        vals = np.linspace(initial_beta, beta, nb)
        idx = 0
        for jj in range(num_sweeps):
            if jj % sweeps_per_beta == 0 and idx < nb - 1:
                idx += 1
            run[jj] = vals[idx]
    return run


def mcmc(num_sweeps, m_start, beta, J, h, anneal=False, sweeps_per_beta=1, initial_beta=0):
    """NMC/nmc.py:28-91 (== NPT/npt.py:47-110).  Returns M [N, num_sweeps] float64."""
    csr = J if isinstance(J, Csr) else Csr(J)
    N = csr.n
    M = np.zeros((N, num_sweeps))  # raises ValueError for negative counts like the reference (:52)
    if num_sweeps == 0:
        return M
    run = _beta_schedule(num_sweeps, beta, anneal, sweeps_per_beta, initial_beta)
    perm = np.empty((num_sweeps, N), dtype=np.int32)
    u = np.empty((num_sweeps, N))
    for t in range(num_sweeps):  # stream order: permutation(N) then N x rand()  (:71,:87)
        perm[t] = np.random.permutation(N)
        u[t] = np.random.rand(N)
    Mi, _ = sweeps_stream(csr, np.asarray(h, dtype=np.float64).reshape(-1), m_start, run, perm, u)
    M[:, :] = Mi.T
    return M


def trace_energies(M, J, h):
    """NMC/nmc.py:386-387 (dense expression, same operand order)."""
    return [-(M[:, i].T @ J @ M[:, i] / 2 + M[:, i].T @ h) for i in range(M.shape[1])]


def atanh_saturated(x):
    """NMC/nmc.py:230-255."""
    hi = np.tanh(19.06)
    return np.arctanh(np.clip(x, -hi + EPS, hi - EPS))


def loopy_bp(J, h, beta, h_msgs, u_msgs, tolerance, max_iterations):
    """NMC/nmc.py:168-228 (dense messages).  Returns (magnetizations, iteration, h_msgs, u_msgs)."""
    N = J.shape[0]
    h = np.asarray(h).reshape(-1)
    tJ = np.tanh(beta * J)
    it = 0
    for it in range(max_iterations):
        h_old, u_old = h_msgs.copy(), u_msgs.copy()
        for i in range(N):
            tot = h[i] + np.sum(u_msgs[:, i])
            h_msgs[i, :] = tot - u_msgs[:, i]
            h_msgs[i, i] = 0
        u_msgs = (1 / beta) * atanh_saturated(tJ * np.tanh(beta * h_msgs))
        du = np.max(np.abs(u_msgs - u_old)) / np.max(np.abs(u_msgs) + np.abs(u_old))
        dh = np.max(np.abs(h_msgs - h_old)) / np.max(np.abs(h_msgs) + np.abs(h_old))
        if du < tolerance and dh < tolerance:
            break
    mag = np.tanh(beta * (h + np.sum(u_msgs, axis=0)))
    return mag, it, h_msgs, u_msgs


def find_clusters(J, mag, thr0, thr_cut, step):
    """NMC/nmc.py:257-318."""
    seeds = np.where(np.abs(mag) >= thr0)[0]
    cl = []
    for s in seeds:
        if any(s in c for c in cl):
            continue
        nb = np.where(J[s, :] != 0)[0]
        nb = np.setdiff1d(nb, np.hstack(cl) if cl else [])
        cl.append(np.append(s, np.intersect1d(nb, seeds)))
    thr = thr0 - step
    while thr > thr_cut:
        for i, c in enumerate(cl):
            nb = np.unique(np.where(J[c, :] != 0)[1])
            nb = np.setdiff1d(nb, np.hstack(cl) if cl else [])
            cl[i] = np.append(cl[i], nb[np.abs(mag[nb]) >= thr])
        thr -= step
    return cl


def lbp_convexified(J, h, lambda_start, lambda_end, factor, m_star, epsilon, tolerance, max_iterations, thr0, thr_cut,
                    beta, want_marginals=False):
    """NMC/nmc.py:93-166.  Returns clusters (and the per-lambda marginals when asked)."""
    h = np.asarray(h).reshape(-1)
    m_star = np.asarray(m_star).reshape(-1)
    lam = lambda_start
    N = J.shape[0]
    h_msgs = np.zeros((N, N))
    u_msgs = J * m_star.reshape(1, -1)
    marg_all = {}
    prev = None
    while lam >= lambda_end:
        h_lam = h + lam * m_star * epsilon
        mag, it, h_msgs, u_msgs = loopy_bp(J, h_lam.copy(), beta, h_msgs.copy(), u_msgs.copy(), tolerance,
                                           max_iterations)
        if it == max_iterations - 1 and lam == lambda_start:
            raise ValueError('LBP diverged at initial lambda, please try a larger lambda_start or increase '
                             'max_iterations or beta')
        elif it == max_iterations - 1:
            lambda_end = lam
            mag = prev
        else:
            prev = mag
        marg_all[lam] = mag
        lam = lam * factor
        if round(lam, 6) == 0:
            break
    cl = find_clusters(J, mag, thr0, thr_cut, 0.01)
    return (cl, marg_all) if want_marginals else cl


def nmc_subroutine(J, h, variant, m_star, num_cycles, S, full_update_frequency, M_skip, global_beta, temp_x, lam0, lam1,
                   lamf, thr0, thr_cut, max_iterations, tolerance, all_clusters=None):
    """NMC/nmc.py:320-440 (variant 'nmc') and NPT/npt.py:357-477 (variant 'npt').

    'nmc': clusters and the phase matrices are rebuilt every cycle from the running m_star.
    'npt': LBP once per call; h_c / h_nc persist across cycles (freeze biases accumulate)."""
    N = len(h)
    epsilon = np.abs(h) + np.sum(np.abs(J), axis=1)
    all_spins = np.arange(N)
    m_init = m_star
    provided = all_clusters is not None
    cols = S * num_cycles * 3 // M_skip
    M_all = np.zeros((N, cols))
    E_all = np.zeros(cols)
    at = 0

    def detect(ms):
        cl = lbp_convexified(J, h, lam0, lam1, lamf, ms.copy(), epsilon, tolerance, max_iterations, thr0, thr_cut,
                             global_beta)
        return np.concatenate(cl).astype(int) if cl else np.array([], dtype=int)

    def record(M):
        nonlocal at, m_init
        en = trace_energies(M, J, h)
        M_all[:, at:at + S // M_skip] = M[:, ::M_skip]
        E_all[at:at + S // M_skip] = en[::M_skip]
        at += S // M_skip
        m_init = M[:, int(np.argmin(en))]
        return en

    if variant == 'npt':
        if not provided:
            all_clusters = detect(m_star)
        non = np.setdiff1d(all_spins, all_clusters)
        J_c, h_c = J.copy(), h.copy()
        J_c[all_clusters, :] = J_c[all_clusters, :] / temp_x
        h_c[all_clusters] /= temp_x
        J_nc, h_nc = J.copy(), h.copy()
        csr_c, csr_nc, csr_full = Csr(J_c), Csr(J_nc), Csr(J)

    for cycle in range(num_cycles):
        if variant == 'nmc':
            if not provided:
                all_clusters = detect(m_star)
            non = np.setdiff1d(all_spins, all_clusters)
            J_c, h_c = J.copy(), h.copy()
            J_c[all_clusters, :] = J_c[all_clusters, :] / temp_x
            h_c[all_clusters] /= temp_x
            csr_c, csr_full = Csr(J_c), Csr(J)
            csr_nc = csr_full
        h_c[non] = m_init[non] * FREEZE
        record(mcmc(S, m_init.copy(), global_beta, csr_c, h_c))
        if variant == 'nmc':
            h_nc = h.copy()
        h_nc[all_clusters] = m_init[all_clusters] * FREEZE
        record(mcmc(S, m_init.copy(), global_beta, csr_nc, h_nc))
        if cycle % full_update_frequency == 0:
            record(mcmc(S, m_init.copy(), global_beta, csr_full, h))
            if variant == 'nmc':
                m_star = m_init.copy()
    M_all, E_all = M_all[:, :at], E_all[:at]
    return M_all, E_all, np.min(E_all), all_clusters


class RefNMC:
    """NMC/nmc.py:13-520."""

    def __init__(self, J, h):
        self.J = J
        self.h = np.asarray(h).reshape(-1)

    def run(self, num_sweeps_initial=10000, num_sweeps_per_NMC_phase=10000, num_NMC_cycles=10, full_update_frequency=1,
            M_skip=1, temp_x=20, global_beta=2.5, lambda_start=0.5, lambda_end=0.01, lambda_reduction_factor=0.9,
            threshold_initial=0.999999, threshold_cutoff=0.99999, max_iterations=100, tolerance=EPS):
        nf = np.max(np.abs(self.J))
        self.J = self.J / nf
        self.h = self.h / nf
        N = len(self.h)
        m_init = np.sign(2 * np.random.rand(N) - 1)
        M = mcmc(num_sweeps_initial, m_init.copy(), global_beta, self.J, self.h, anneal=True, sweeps_per_beta=1,
                 initial_beta=0)
        en = trace_energies(M, self.J, self.h)
        m_star = M[:, int(np.argmin(en))].copy()
        Mo, Eo, Emin, _ = nmc_subroutine(self.J, self.h, 'nmc', m_star, num_NMC_cycles, num_sweeps_per_NMC_phase,
                                         full_update_frequency, M_skip, global_beta, temp_x, lambda_start, lambda_end,
                                         lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations,
                                         tolerance)
        return Mo, Eo, Emin


def select_pairs(all_pairs, k):
    """NPT/npt.py:514-533 (stdlib `random.randint`: a second, separately seeded stream)."""
    avail = list(all_pairs)
    out = []
    for _ in range(k):
        if not avail:
            raise ValueError("Cannot find non-overlapping pairs.")
        p = avail[_pyrandom.randint(0, len(avail) - 1)]
        out.append(p)
        avail = [q for q in avail if q[0] not in p and q[1] not in p]
    return out


def dense_energy(m, J, h):
    """NPT/npt.py:657."""
    return -m.T @ J @ m / 2 - m.T @ h


class RefNPT:
    """NPT/npt.py:15-700 with the pool run in order."""

    def __init__(self, J, h):
        self.J = J
        self.h = np.asarray(h).reshape(-1)

    def run(self, beta_list, num_replicas, doNMC, num_sweeps_MCMC=1000, num_sweeps_read=1000, num_swap_attempts=100,
            num_swapping_pairs=1, num_cycles=10, full_update_frequency=1, M_skip=1, temp_x=20, global_beta=2.5,
            lambda_start=0.5, lambda_end=0.01, lambda_reduction_factor=0.9, threshold_initial=0.999999,
            threshold_cutoff=0.99999, max_iterations=100, tolerance=EPS):
        R = num_replicas
        S = num_sweeps_MCMC // num_swap_attempts
        S_read = num_sweeps_read // num_swap_attempts
        S_nmc = int(np.ceil(num_sweeps_MCMC / num_swap_attempts / 3 / num_cycles))
        nf = np.max(np.abs(self.J))
        self.J = self.J / nf
        self.h = self.h / nf
        if len(doNMC) != R:
            raise ValueError("The length of doNMC does not match the number of replicas.")
        N = self.J.shape[0]
        all_pairs = [(i, i + 1) for i in range(1, R)]
        M = np.zeros((R * N, S))
        m_start = np.sign(2 * np.random.rand(R * N, 1) - 1)
        csr = Csr(self.J)
        log_pairs, log_acc = [], []
        for _ in range(num_swap_attempts):
            for r in range(R):
                blk = m_start[r * N:(r + 1) * N]
                if not doNMC[r]:
                    Mr = mcmc(S, blk.copy(), beta_list[r], csr, self.h)
                else:
                    Mr, _, _, _ = nmc_subroutine(self.J, self.h, 'npt', blk.copy().flatten(), num_cycles, S_nmc,
                                                 full_update_frequency, M_skip, global_beta, temp_x, lambda_start,
                                                 lambda_end, lambda_reduction_factor, threshold_initial,
                                                 threshold_cutoff, max_iterations, tolerance)
                M[r * N:(r + 1) * N, :] = Mr[:, -S:]
            m_start = M[:, -1].copy().reshape(-1, 1)
            last = M[:, -1]
            for (a, b) in select_pairs(all_pairs, num_swapping_pairs):
                ma = last[(a - 1) * N:a * N].copy()
                mb = last[(b - 1) * N:b * N].copy()
                dE = dense_energy(mb, self.J, self.h) - dense_energy(ma, self.J, self.h)
                dB = beta_list[b - 1] - beta_list[a - 1]
                log_pairs.append((a, b))
                ok = np.random.rand() < min(1, np.exp(dB * dE))
                log_acc.append(int(ok))
                if ok:
                    m_start[(a - 1) * N:a * N] = mb.reshape(-1, 1)
                    m_start[(b - 1) * N:b * N] = ma.reshape(-1, 1)
        Energy = np.zeros(R)
        for r in range(R):  # min over the FIRST S_read columns (NPT/npt.py:685-692, :41)
            blk = M[r * N:(r + 1) * N, :]
            Energy[r] = np.min([-(blk[:, i].T @ self.J @ blk[:, i] / 2 + blk[:, i].T @ self.h)
                                for i in range(S_read)])
        self.swap_pairs = np.array(log_pairs, dtype=np.int32).reshape(-1, 2)
        self.swap_accepted = np.array(log_acc, dtype=np.int8)
        return M, Energy


class RefAPT_ICM:
    """NPT/apt_ICM.py:14-305."""
    num_subreplicas = 10  # :177
    katzgraber = True     # :178

    def __init__(self, J, h):
        self.J = J
        h = np.array(h) if isinstance(h, list) else h
        self.h = h[:, np.newaxis] if h.ndim == 1 else h

    def run(self, beta_list, num_replicas, num_sweeps_MCMC=1000, num_sweeps_read=1000, num_swap_attempts=100,
            num_swapping_pairs=1):
        R, K = num_replicas, self.num_subreplicas
        S = num_sweeps_MCMC // num_swap_attempts
        S_read = num_sweeps_read // num_swap_attempts
        N = self.J.shape[0]
        J, hcol = self.J, self.h
        hflat = hcol.reshape(-1)
        csr = Csr(J)
        all_pairs = [(i, i + 1) for i in range(1, R)]
        M = np.zeros((N * R, S * K))
        m_start = np.sign(2 * np.random.rand(N * R, K) - 1)
        log_pairs, log_acc = [], []
        for _ in range(int(num_swap_attempts)):
            for r in range(R):
                for j in range(K):
                    Mt = mcmc(S, m_start[r * N:(r + 1) * N, j], beta_list[r], csr, hflat)
                    M[r * N:(r + 1) * N, j * S:(j + 1) * S] = Mt
                    m_start[r * N:(r + 1) * N, j] = Mt[:, -1]
            for r in range(R):  # Houdayer move on the FIRST column of each sub-replica block (:215-246)
                shuf = np.random.permutation(K)
                for p in range(K // 2):
                    ja, jb = shuf[2 * p], shuf[2 * p + 1]
                    s1 = M[r * N:(r + 1) * N, ja * S].copy()
                    s2 = M[r * N:(r + 1) * N, jb * S].copy()
                    cl = _c_clusters(csr, s1.astype(np.int8), s2.astype(np.int8))
                    if cl:
                        pick = cl[np.random.randint(len(cl))]
                        if self.katzgraber and len(pick) > N // 2:
                            s1 = -s1
                        else:
                            s1[pick], s2[pick] = s2[pick].copy(), s1[pick].copy()
                        M[r * N:(r + 1) * N, ja * S] = s1
                        M[r * N:(r + 1) * N, jb * S] = s2
            sel = select_pairs(all_pairs, num_swapping_pairs)
            for j in range(K):
                last = M[:, (j + 1) * S - 1]
                for (a, b) in sel:
                    ma = last[(a - 1) * N:a * N].copy()
                    mb = last[(b - 1) * N:b * N].copy()
                    Ea = -ma.T @ J @ ma / 2 - ma.T @ hcol
                    Eb = -mb.T @ J @ mb / 2 - mb.T @ hcol
                    dE, dB = Eb - Ea, beta_list[b - 1] - beta_list[a - 1]
                    log_pairs.append((a, b))
                    ok = bool(np.random.rand() < min(1, np.exp(dB * dE)))
                    log_acc.append(int(ok))
                    if ok:
                        m_start[(a - 1) * N:a * N, j] = mb
                        m_start[(b - 1) * N:b * N, j] = ma
        Energy = np.zeros(R)
        for r in range(R):
            blk = M[r * N:(r + 1) * N, :]
            Energy[r] = np.min([(-1 * (blk[:, i].T @ J @ blk[:, i] / 2 + blk[:, i].T @ hcol)).item()
                                for i in range(S_read)])
        self.swap_pairs = np.array(log_pairs, dtype=np.int32).reshape(-1, 2)
        self.swap_accepted = np.array(log_acc, dtype=np.int8)
        return M, Energy


class RefAPTPreprocessor:
    """NPT/apt_preprocessor.py:12-204 (file outputs and plots not restated)."""

    def __init__(self, J, h):
        self.J = J
        self.h = np.asarray(h, dtype=np.float64)
        self.N = J.shape[0]

    def run(self, num_sweeps_MCMC=1000, num_sweeps_read=1000, num_rng=100, beta_start=0.5, alpha=1.25,
            sigma_E_val=1000, beta_max=30):
        import scipy.sparse as sp
        Jd = self.J.toarray() if sp.issparse(self.J) else np.asarray(self.J)
        nf = np.max(np.abs(Jd))
        Jd = Jd / nf
        h = self.h.reshape(-1) / nf
        csr = Csr(Jd)
        beta, sigma = [beta_start], []
        sigma_E = sigma_E_val
        sigma_min = 0.5 * np.min(np.abs(Jd[Jd != 0]))
        saved = np.zeros((num_rng, self.N))
        it = 1
        while sigma_E > sigma_min:
            if it != 1:
                beta.append(beta[-1] + alpha / sigma_E)
            Energy = np.zeros((num_rng, num_sweeps_read))
            for j in range(num_rng):
                m0 = np.sign(2. * np.random.rand(self.N, 1) - 1) if it == 1 else saved[j, :].copy().reshape(-1, 1)
                M = mcmc(num_sweeps_MCMC, m0.copy(), beta[-1], csr, h)
                mm = M[:, -num_sweeps_read:]
                for kk in range(num_sweeps_read):
                    m = mm[:, kk].copy().reshape(1, -1)
                    Energy[j, kk] = (-(m @ (Jd / 2) @ m.T + m @ h.reshape(-1, 1))).item()
                saved[j, :] = mm[:, -1]
            sigma_E = np.mean(np.std(Energy, axis=1))
            if beta[-1] > beta_max:
                break
            sigma.append(sigma_E)
            it += 1
        return beta, sigma
