"""Backbone inference on the host: convexified loopy belief propagation + cluster growth.

Reference: NMC/nmc.py:93-318 (== NPT/npt.py:129-355).  SURVEY.md section 2 row 6 keeps this step on the host (it is
the "next" row f-1, not part of the sweep path); it is restated here on the EDGE LIST (2*nnz messages) instead of
the reference's dense N x N message matrices, so that an N = 10^4 instance needs ~1 MB instead of 1.6 GB.

Message layout: for every stored entry e = (i -> j) of the CSR matrix (i.e. J[i, j] != 0)
    u[e]  == u_msgs[i, j]   and   hm[e] == h_msgs[i, j]        (reference index order)
`rev[e]` is the position of the transposed entry (j -> i).  Entries with J == 0 carry u == 0 forever and never
influence anything (tanh(0) == 0), which is why dropping them is exact.
"""
import numpy as np

from .npsum import SparsePairwiseSum

EPS = np.finfo(float).eps
_SAT = np.tanh(19.06)


def atanh_saturated(x):
    """NMC/nmc.py:230-255."""
    return np.arctanh(np.clip(x, -_SAT + EPS, _SAT - EPS))


class EdgeGraph:
    """Edge-list view of a symmetric-pattern CSR matrix."""

    def __init__(self, inst):
        self.n = inst.n
        self.indptr = inst.indptr.astype(np.int64)
        self.src = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.indptr))   # i of entry (i -> j)
        self.dst = inst.indices.astype(np.int64)                                        # j
        self.indptr32 = np.ascontiguousarray(inst.indptr, dtype=np.int32)
        self.dst32 = np.ascontiguousarray(inst.indices, dtype=np.int32)
        self.val = inst.data
        # rev[e] = index of (j -> i): sort entries by (dst, src); the sorted sequence enumerates (i' = dst, j' = src)
        order = np.lexsort((self.src, self.dst))
        if order.shape[0] != self.src.shape[0] or not (np.array_equal(self.src[order], self.dst)
                                                       and np.array_equal(self.dst[order], self.src)):
            raise ValueError("LBP needs a structurally symmetric J")
        self.rev = order
        self.not_diag = self.src != self.dst
        self.deg_offdiag = np.bincount(self.src[self.not_diag], minlength=self.n)
        # np.sum over a dense length-N row/column == pairwise summation; restated exactly on the non-zeros
        self._pw = SparsePairwiseSum(self.n, self.n, self.src, self.dst)

    def row_abs_sum(self):
        """np.sum(np.abs(J), axis=1)  (NMC/nmc.py:353)."""
        return self._pw(np.abs(self.val))

    def epsilon(self, h):
        """epsilon = |h| + sum_j |J_ij|  (NMC/nmc.py:353)."""
        return np.abs(h) + self.row_abs_sum()


def _colsum(g, u):
    """sum_k u_msgs[k, i] for every i, in NumPy's pairwise association (np.sum(u_msgs[:, i]), NMC/nmc.py:201).
    The entries (k -> i) are the reverses of the entries (i -> k) of row i, in ascending k."""
    return g._pw(u[g.rev])


def _colsum_seq(g, u):
    """np.sum(u_msgs, axis=0) (NMC/nmc.py:216): an axis-0 reduction of a C-ordered matrix adds row after row,
    i.e. sequentially in ascending k."""
    return np.bincount(g.src, weights=u[g.rev], minlength=g.n)


def loopy_bp(g, h, beta, state, tolerance, max_iterations):
    """NMC/nmc.py:168-228 on the edge list.

    state = (hm, u, tot): messages on the stored entries plus `tot[i]`, the value every NON-edge entry of row i of
    the reference's dense h_msgs holds (h_msgs[i, j] = total_i for J_ij == 0, j != i) -- needed only because the
    reference's convergence test takes maxima over the full N x N matrices.
    Returns (magnetizations, last_iteration, state)."""
    hm, u, tot = state
    tJ = np.tanh(beta * g.val)
    has_non_nbr = g.deg_offdiag < (g.n - 1)          # node i has some j != i with J_ij == 0
    any_non_nbr = bool(np.any(has_non_nbr))
    it = 0
    for it in range(max_iterations):
        hm_old, u_old, tot_old = hm, u, tot
        tot = h + _colsum(g, u)
        hm = tot[g.src] - u[g.rev]                   # h_msgs[i, j] = total_i - u_msgs[j, i]
        hm = np.where(g.not_diag, hm, 0.0)           # h_msgs[i, i] = 0
        u = (1.0 / beta) * atanh_saturated(tJ * np.tanh(beta * hm))
        with np.errstate(invalid="ignore", divide="ignore"):
            du = (np.max(np.abs(u - u_old)) / np.max(np.abs(u) + np.abs(u_old))) if u.size else np.nan
            num = float(np.max(np.abs(hm - hm_old))) if hm.size else 0.0
            den = float(np.max(np.abs(hm) + np.abs(hm_old))) if hm.size else 0.0
            if any_non_nbr:
                num = max(num, float(np.max(np.abs(tot - tot_old)[has_non_nbr])))
                den = max(den, float(np.max((np.abs(tot) + np.abs(tot_old))[has_non_nbr])))
            dh = np.float64(num) / np.float64(den)
        if du < tolerance and dh < tolerance:
            break
    mag = np.tanh(beta * (h + _colsum_seq(g, u)))
    return mag, it, (hm, u, tot)


def find_clusters(g, mag, threshold_initial, threshold_cutoff, threshold_step, flat=False):
    """NMC/nmc.py:257-318 with CSR neighbour lists instead of dense row scans (same visiting order, same output):
    the compiled host routine behind include/nlmc.h: nlmc_find_clusters."""
    from . import _abi
    L = _abi.lib()
    mag = _abi.as_c(mag, np.float64).reshape(-1)
    ip, ix, v = g.indptr32, g.dst32, _abi.as_c(g.val, np.float64)
    cap = int(max(g.n, ip[-1]) + g.n)
    members, sizes, cnt = np.zeros(cap, np.int32), np.zeros(g.n, np.int32), np.zeros(1, np.int32)
    _abi.check(L.nlmc_find_clusters(g.n, _abi.ptr(ip), _abi.ptr(ix), _abi.ptr(v), _abi.ptr(mag), float(threshold_initial),
                                    float(threshold_cutoff), float(threshold_step), _abi.ptr(members), cap,
                                    _abi.ptr(sizes), _abi.ptr(cnt)))
    k = int(cnt[0])
    flat_members = members[:int(sizes[:k].sum())].astype(np.int64)
    if flat:
        return flat_members                    # == np.concatenate(clusters): all that NMC_subroutine consumes
    cuts = np.cumsum(sizes[:k])
    return [flat_members[a:b] for a, b in zip(np.concatenate([[0], cuts[:-1]]), cuts)]


def find_clusters_py(g, mag, threshold_initial, threshold_cutoff, threshold_step):
    """Plain NumPy statement of the same routine (cross-check of nlmc_find_clusters in tests/test_host_lbp.py)."""
    n = g.n
    amag = np.abs(mag)
    seeds = np.where(amag >= threshold_initial)[0]
    seed_mask = np.zeros(n, dtype=bool)
    seed_mask[seeds] = True
    in_cluster = np.zeros(n, dtype=bool)
    clusters = []

    def nbrs(nodes):
        out = [g.dst[g.indptr[k]:g.indptr[k + 1]][g.val[g.indptr[k]:g.indptr[k + 1]] != 0] for k in nodes]
        return np.unique(np.concatenate(out)) if out else np.zeros(0, dtype=np.int64)

    for s in seeds:
        if in_cluster[s]:
            continue
        nb = nbrs([s])
        nb = nb[~in_cluster[nb]]
        common = nb[seed_mask[nb]]
        c = np.append(s, common)
        clusters.append(c)
        in_cluster[c] = True
    thr = threshold_initial - threshold_step
    while thr > threshold_cutoff:
        for i, c in enumerate(clusters):
            nb = nbrs(c)
            nb = nb[~in_cluster[nb]]
            add = nb[amag[nb] >= thr]
            clusters[i] = np.append(clusters[i], add)
            in_cluster[add] = True
        thr -= threshold_step
    return clusters


def lambda_list(lambda_start, lambda_end, lambda_reduction_factor):
    """The lambdas the loop of NMC/nmc.py:133-160 visits when nothing cuts it short."""
    out, lam = [], lambda_start
    while lam >= lambda_end:
        out.append(lam)
        lam = lam * lambda_reduction_factor
        if round(lam, 6) == 0:
            break
    return out


def lbp_convexified_device(eng, graph, lambda_start, lambda_end, lambda_reduction_factor, m_stars, epsilon, tolerance,
                           max_iterations, threshold_initial, threshold_cutoff, global_beta, want_marginals=False,
                           flat=False):
    """NMC/nmc.py:93-166 for a BATCH of seeds m_stars [P, N] with the message passing on the GPU
    (include/nlmc.h: nlmc_lbp_convexified); the cluster growth stays on the host.  Returns a list of P cluster lists
    (flat=True: P concatenated index arrays) -- and the list of P {lambda: marginals} dicts when want_marginals."""
    lams = lambda_list(lambda_start, lambda_end, lambda_reduction_factor)
    ms = np.atleast_2d(np.asarray(m_stars, dtype=np.float64))
    if not lams:                    # the reference's loop body never runs: find_clusters(None) raises TypeError there
        raise TypeError("bad operand type for abs(): 'NoneType'")
    o = eng.lbp_convexified(ms, epsilon, lams, global_beta, tolerance, max_iterations, _SAT - EPS, want_all=want_marginals)
    if np.any(o["status"] != 0):
        raise ValueError('LBP diverged at initial lambda, please try a larger lambda_start or increase '
                         'max_iterations or beta')
    clusters = [find_clusters(graph, o["mag"][p], threshold_initial, threshold_cutoff, 0.01, flat=flat)
                for p in range(ms.shape[0])]
    if not want_marginals:
        return clusters
    margs = [{lams[l]: o["mag_all"][p, l].copy() for l in range(int(o["n_lambdas"][p]))} for p in range(ms.shape[0])]
    return clusters, margs


def lbp_convexified(inst, lambda_start, lambda_end, lambda_reduction_factor, m_star, epsilon, tolerance,
                    max_iterations, threshold_initial, threshold_cutoff, global_beta, graph=None, want_marginals=False):
    """NMC/nmc.py:93-166.  `inst` is an engine.Instance (normalised J, h)."""
    g = graph if graph is not None else EdgeGraph(inst)
    h = inst.h
    m_star = np.asarray(m_star, dtype=np.float64).reshape(-1)
    lam = lambda_start
    # h_msgs = 0 everywhere; u_msgs = J * m_star.reshape(1, -1)   (NMC/nmc.py:128-129)
    state = (np.zeros(g.val.shape[0]), g.val * m_star[g.dst], np.zeros(g.n))
    marg_all = {}
    prev = None
    mag = None
    while lam >= lambda_end:
        h_lam = h + lam * m_star * epsilon
        mag, it, state = loopy_bp(g, h_lam, global_beta, state, tolerance, max_iterations)
        if it == max_iterations - 1 and lam == lambda_start:
            raise ValueError('LBP diverged at initial lambda, please try a larger lambda_start or increase '
                             'max_iterations or beta')
        elif it == max_iterations - 1:
            lambda_end = lam
            mag = prev
        else:
            prev = mag
        marg_all[lam] = mag
        lam = lam * lambda_reduction_factor
        if round(lam, 6) == 0:
            break
    clusters = find_clusters(g, mag, threshold_initial, threshold_cutoff, 0.01)
    return (clusters, marg_all) if want_marginals else clusters
