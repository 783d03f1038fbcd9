"""Machinery shared by the drop-in classes (NMC, NPT, APT_ICM): engine cache, RNG modes, the single-chain MCMC()
entry point, the host-side backbone inference wrappers and the three-phase NMC cycle.

RNG modes
  rng="numpy"  (default)  the host draws `np.random.permutation(N)` + N x `np.random.rand()` per sweep from the global
               legacy NumPy stream in the reference's program order and the GPU consumes them: after
               `np.random.seed(s)` every returned array equals the reference's (spins bit for bit).
  rng="philox"            throughput mode: visiting order and uniforms are generated on the device (counter-based
               Philox4x32-10); same Markov kernel, different stream.

Backbone inference (LBP_convexified inside NMC_subroutine / NMC_task)
  lbp="host"    edge-list restatement on the host with NumPy's own summation order: marginals, stopping lambda and
                clusters equal the reference's bit for bit (default with rng="numpy").
  lbp="device"  the message passing runs in the HIP kernel k_lbp, batched over replicas/restarts (default with
                rng="philox"); equal to rounding -- see tests/test_gpu_lbp.py for what that means for the lambda at
                which the reference's loop stops.

Accepted-and-ignored: `use_hash_table` / `hash_table` (a CPU memoisation trick),
`num_cores` (the process pool is replaced by one batched launch), plotting (opt-in through plot=True).
"""

import os

import numpy as np

from . import hostlogic
from .engine import Engine, Instance, fused_window, trace_layout
from .lbp import (EdgeGraph, lbp_convexified, lbp_convexified_device, loopy_bp, atanh_saturated as _atanh_saturated,
                  find_clusters as _find_clusters)

EPS = np.finfo(float).eps


def _default_rng_mode():
    return os.environ.get("NLMC_RNG", "numpy")


class _EngineCache:
    """One HIP context per (instance, batch size); rebuilt when the caller hands in a different matrix."""

    def __init__(self, device=0):
        self.device = device
        self._key = None
        self._inst = None
        self._J_obj = self._h_obj = None
        self._engines = {}

    @staticmethod
    def _fingerprint(J, h):
        import scipy.sparse as sp
        if sp.issparse(J):
            A = J.tocsr()
            return ("s", A.shape, A.nnz, hash(A.data.tobytes()), hash(A.indices.tobytes()), hash(np.asarray(h).tobytes()))
        A = np.asarray(J)
        return ("d", A.shape, hash(A.tobytes()), hash(np.asarray(h).tobytes()))

    def instance(self, J, h):
        if self._inst is not None and J is self._J_obj and h is self._h_obj:
            return self._inst                      # same objects as last time: skip the O(size) fingerprint
        key = self._fingerprint(J, h)
        self._J_obj, self._h_obj = J, h
        if key != self._key:
            self.close()
            self._inst = Instance(J, h)
            self._key = key
        return self._inst

    def engine(self, J, h, n_chains):
        inst = self.instance(J, h)
        if n_chains not in self._engines:
            self._engines[n_chains] = Engine(inst, None, n_chains, device=self.device)
        return self._engines[n_chains]

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}
        self._key = None
        self._inst = None


class SweepMixin:
    """Shared by NMC / NPT / APT_ICM: one MCMC() call on one chain."""

    def _init_backend(self, rng, seed, device):
        self.rng = rng if rng is not None else _default_rng_mode()
        if self.rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.seed = int(seed) if seed is not None else 0xA5A50000
        self._cache = _EngineCache(device)
        self._phase_cache = _EngineCache(device)
        self._sweep_counter = 0          # philox mode: global sweep index, never reused within one object

    def _check_hash_table(self, hash_table, use_hash_table):
        # NMC/nmc.py:74-76: the reference insists on an LRUCache only when the table is actually used
        if use_hash_table and hash_table is None:
            raise ValueError("hash_table must be an instance of cachetools.LRUCache")

    def _mcmc_on(self, eng, num_sweeps, m_start, beta_run, record=True, flags=None, temp_x=1.0, record_stride=1):
        """Run `num_sweeps` sweeps of ONE chain on `eng` (n_chains == 1).  Returns the engine's output dict (configurations
        after sweeps 0, record_stride, 2 record_stride, ... when `record`)."""
        rs = int(record_stride) if record else 0
        n = eng.n
        s0 = np.asarray(m_start, dtype=np.float64).reshape(-1)
        eng.set_spins(s0.astype(np.int8)[None, :])
        eng.set_flags(None if flags is None else flags[None, :], temp_x)
        if self.rng == "numpy":
            b = np.asarray(beta_run, dtype=np.float64)
            piece = 256 if rs <= 1 else max(rs, 256 // rs * rs)
            if num_sweeps < 3 * piece:
                perm, u = hostlogic.draw_legacy_stream(num_sweeps, n)
                return eng.sweep_stream(perm[None], u[None], b[None, :], record_stride=rs,
                                        want_energy=True, want_min=True, want_state=True)
            # Long runs: the draws of piece i+1 (the reference's own np.random calls, in its order, on ONE worker thread) are made
            # while the GPU sweeps piece i (the C-ABI call releases the GIL): 12 of 48 us per sweep at N = 10^3 leave the
            # critical path.  The pieces are stitched exactly as one call would return them (trace, energies, first argmin).
            from concurrent.futures import ThreadPoolExecutor
            from .engine import _stitch_outputs
            cuts = list(range(0, num_sweeps, piece)) + [num_sweeps]
            outs = []
            with ThreadPoolExecutor(max_workers=1) as ex:
                fut = ex.submit(hostlogic.draw_legacy_stream, cuts[1] - cuts[0], n)
                for i in range(len(cuts) - 1):
                    perm, u = fut.result()
                    if i + 2 < len(cuts):
                        fut = ex.submit(hostlogic.draw_legacy_stream, cuts[i + 2] - cuts[i + 1], n)
                    outs.append(eng.sweep_stream(perm[None], u[None], b[None, cuts[i]:cuts[i + 1]], record_stride=rs,
                                                 want_energy=True, want_min=True, want_state=True))
            return _stitch_outputs(outs, piece, rs, True, True, True)
        beta2 = np.asarray(beta_run)[None, :]
        # Arithmetic of the single-chain device-RNG calls (MCMC(), the phases of NMC.run / NMC_subroutine): a property of the
        # INSTANCE alone -- not of M_skip, not of the number of sweeps.  n >= 256: the "f32" dynamics of the
        # batched path (24-bit fixed-point couplings + logistic thresholds; DESIGN.md section 2 states the tolerance: the
        # chain samples the Boltzmann law of (Jq, hq) 2^-qs with |Jq 2^-qs - J| <= 2^-(qs+1), nothing lost for +-J / integer
        # instances), on fused windows planned piece by piece within a memory budget where the instance qualifies, sweep by
        # sweep otherwise (same bits).  Smaller instances: fp64 fields.  o["energy_recorded"] holds the fp64 energies of the
        # recorded configurations, computed on the device copy of the trace -- what the reference's list comprehension
        # (NMC/nmc.py:386-387) would give for them; NMC_subroutine's argmin hand-off (NMC/nmc.py:394-395) uses THEM whenever every
        # sweep was recorded (M_skip == 1) and the kernel's tracked minimum (exact integers of the quantised model in the "f32"
        # arithmetic) only for strided recording.
        if n >= 256:
            o = eng.sweep_philox_windows(num_sweeps, self.seed, sweep0=self._sweep_counter, beta=beta2, record_stride=rs,
                                         want_energy=True, want_min=True, want_state=True, want_recorded_energy=True)
        else:
            o = eng.sweep_philox(num_sweeps, self.seed, sweep0=self._sweep_counter, beta=beta2, precision="f64", record_stride=rs,
                                 want_energy=True, want_min=True, want_state=True)
        self._sweep_counter += num_sweeps
        return o


class Common(SweepMixin):
    """Methods NMC (NMC/nmc.py) and NPT (NPT/npt.py) share line for line in the reference."""
    _variant = "nmc"

    def __init__(self, J, h, rng=None, seed=None, device=0, lbp=None):
        self.J = J
        self.h = np.asarray(h).reshape(-1)
        self._init_backend(rng, seed, device)
        self.lbp = lbp if lbp is not None else os.environ.get("NLMC_LBP", "device" if self.rng == "philox" else "host")
        if self.lbp not in ("host", "device"):
            raise ValueError("lbp must be 'host' or 'device'")

    def _detect_clusters(self, inst, graph, epsilon, m_stars, lambda_start, lambda_end, lambda_reduction_factor, tolerance,
                         max_iterations, threshold_initial, threshold_cutoff, global_beta):
        """LBP_convexified for a batch of seeds m_stars [P, N] -> list of P index arrays (clusters concatenated)."""
        ms = np.atleast_2d(np.asarray(m_stars, dtype=np.float64))
        if self.lbp == "device":
            eng = self._cache.engine(self.J, self.h, 1)
            return [c.astype(int) for c in
                    lbp_convexified_device(eng, graph, lambda_start, lambda_end, lambda_reduction_factor, ms, epsilon,
                                           tolerance, max_iterations, threshold_initial, threshold_cutoff, global_beta,
                                           flat=True)]
        else:
            cls = [lbp_convexified(inst, lambda_start, lambda_end, lambda_reduction_factor, ms[p].copy(), epsilon,
                                   tolerance, max_iterations, threshold_initial, threshold_cutoff, global_beta, graph=graph)
                   for p in range(ms.shape[0])]
        return [np.concatenate(cl).astype(int) if cl else np.array([], dtype=int) for cl in cls]

    # ------------------------------------------------------------------------------------------------
    def MCMC(self, num_sweeps, m_start, beta, J, h, anneal=False, sweeps_per_beta=1, initial_beta=0,
             hash_table=None, use_hash_table=False):
        """Heat-bath sweeps with a fresh random permutation per sweep (NMC/nmc.py:28-91).
        Returns M [N, num_sweeps] float64, column jj = state after sweep jj.  rng="numpy": the reference's stream and fp64
        arithmetic, bit for bit.  rng="philox": device RNG; N >= 256: couplings in 24-bit fixed point (exact for +-J /
        integer instances, |dJ| <= 2^-(qs+1) max|J| 2^-23 otherwise -- DESIGN.md section 2), smaller N: fp64 fields."""
        N = J.shape[0]
        M = np.zeros((N, num_sweeps))           # negative counts raise ValueError exactly like the reference (:52)
        if num_sweeps == 0:
            return M
        self._check_hash_table(hash_table, use_hash_table)
        run = hostlogic.beta_schedule(num_sweeps, beta, anneal, sweeps_per_beta, initial_beta)
        eng = self._phase_cache.engine(J, np.asarray(h).reshape(-1), 1)
        o = self._mcmc_on(eng, num_sweeps, m_start, run)
        return trace_layout(o["spins"], out=M)        # M[:, jj] = state after sweep jj

    # ------------------------------------------------------------------------------------------------
    def atanh_saturated(self, x):
        return _atanh_saturated(x)

    def LoopyBeliefPropagation(self, J, h, beta, h_msgs, u_msgs, tolerance, max_iterations):
        """NMC/nmc.py:168-228.  Dense message matrices in and out (reference signature); computed on the edge list."""
        inst = Instance(J, np.zeros(J.shape[0]))
        g = EdgeGraph(inst)
        h_msgs = np.asarray(h_msgs)
        u_msgs = np.asarray(u_msgs)
        # non-edge rows of h_msgs hold one value per row (total_i): recover it for the convergence test
        tot = np.zeros(g.n)
        full = np.ones((g.n, g.n), dtype=bool)
        full[g.src, g.dst] = False
        np.fill_diagonal(full, False)
        for i in range(g.n):
            nz = np.nonzero(full[i])[0]
            if nz.size:
                tot[i] = h_msgs[i, nz[0]]
        state = (h_msgs[g.src, g.dst].astype(float), u_msgs[g.src, g.dst].astype(float), tot)
        mag, it, (hm, u, tot) = loopy_bp(g, np.asarray(h, dtype=float).reshape(-1), beta, state, tolerance, max_iterations)
        Hd = np.repeat(tot[:, None], g.n, axis=1)
        Hd[g.src, g.dst] = hm
        np.fill_diagonal(Hd, 0.0)
        Ud = np.zeros((g.n, g.n))
        Ud[g.src, g.dst] = u
        Jd = inst.csr.toarray()
        tH = np.tanh(beta * Hd)
        corr = (np.tanh(beta * Jd) + tH * tH.T) / (1 + np.tanh(beta * Jd) * tH * tH.T + 1e-10)
        corr = corr - np.diag(np.diag(corr))
        return mag, corr, (1 / beta) * _atanh_saturated(mag), (1 / beta) * _atanh_saturated(corr), it, Hd, Ud

    def find_clusters(self, magnetizations, threshold_initial, threshold_cutoff, threshold_step):
        """NMC/nmc.py:257-318."""
        inst = self._cache.instance(self.J, self.h)
        return _find_clusters(EdgeGraph(inst), np.asarray(magnetizations), threshold_initial, threshold_cutoff,
                              threshold_step)

    def LBP_convexified(self, lambda_start, lambda_end, lambda_reduction_factor, m_star, epsilon, tolerance,
                        max_iterations, threshold_initial, threshold_cutoff, global_beta):
        """NMC/nmc.py:93-166.  Returns (clusters, marginals_all_lambdas, mean_marginals_all_lambdas,
        h_tilde_all_lambdas, J_tilde_all_lambdas); the last two are left empty (never consumed by the reference's
        own callers, NMC/nmc.py:369)."""
        from collections import defaultdict
        inst = self._cache.instance(self.J, self.h)
        clusters, marg = lbp_convexified(inst, lambda_start, lambda_end, lambda_reduction_factor, m_star, epsilon,
                                         tolerance, max_iterations, threshold_initial, threshold_cutoff, global_beta,
                                         graph=self._graph(inst), want_marginals=True)
        print(f"\ncluster size = {sum(len(c) for c in clusters)}\n")
        m_all, mean_all = defaultdict(list), defaultdict(list)
        for k, v in marg.items():
            m_all[k] = v
            mean_all[k] = np.mean(v)
        return clusters, m_all, mean_all, defaultdict(list), defaultdict(list)

    def _graph(self, inst):
        if getattr(self, "_graph_of", None) is not inst:
            self._graph_obj = EdgeGraph(inst)
            self._graph_of = inst
        return self._graph_obj

    # ------------------------------------------------------------------------------------------------
    def NMC_subroutine(self, m_star, num_cycles, num_sweeps_per_NMC_phase, full_update_frequency, M_skip, global_beta,
                       temp_x, lambda_start, lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff,
                       max_iterations, tolerance, all_clusters=None, hash_table=None, use_hash_table=False):
        """Three-phase NMC cycles (NMC/nmc.py:320-440): cluster spins hot with the rest frozen, then the clusters
        frozen, then (every full_update_frequency cycles) a plain sweep phase; every phase restarts from the
        argmin-energy column of the previous one.  The phase matrices J_c/h_c/J_nc/h_nc of the reference are not
        materialised: the engine applies them through per-spin flags (include/nlmc.h: nlmc_set_flags)."""
        inst = self._cache.instance(self.J, self.h)
        eng = self._cache.engine(self.J, self.h, 1)
        N = inst.n
        S = int(num_sweeps_per_NMC_phase)
        graph = self._graph(inst)
        epsilon = graph.epsilon(inst.h)
        m_init = np.asarray(m_star, dtype=np.float64).reshape(-1)
        provided = all_clusters is not None
        cols = S * num_cycles * 3 // M_skip
        M_overall = np.zeros((N, cols))
        energy_overall = np.zeros(cols)
        at = 0
        run = np.full(S, float(global_beta))

        def phase(kind, m_from):
            nonlocal at, m_init
            fl = hostlogic.phase_flags(N, m_from, all_clusters, kind)
            w = S // M_skip
            strided = M_skip > 0 and S % M_skip == 0 and M_overall.flags["C_CONTIGUOUS"]
            o = self._mcmc_on(eng, S, m_from, run, flags=None if kind == "ALL" else fl, temp_x=temp_x,
                              record_stride=M_skip if strided else 1)
            en = o["energy"][0]
            if strided:                                    # M[:, ::M_skip] is what the engine recorded: laid out in place
                trace_layout(o["spins"], dst_col=[at], row_len=M_overall.shape[1], out=M_overall)
            else:
                M = o["spins"][0].T.astype(np.float64)
                M_overall[:, at:at + w] = M[:, ::M_skip]   # shape mismatch raises like the reference when S % M_skip
            er = o.get("energy_recorded")                  # fp64 energies of the recorded columns (device-RNG mode, n >= 256)
            energy_overall[at:at + w] = en[::M_skip] if er is None else (er[0] if strided else er[0][::M_skip])
            at += w
            if er is not None and (not strided or M_skip == 1):
                # every sweep was recorded and its fp64 energy is at hand: the hand-off is the reference's own rule, the FIRST
                # argmin of the fp64 energies (NMC/nmc.py:386-395) -- the same column the energies returned above point at (for
                # couplings that are not exact in fixed point the kernel's tracked minimum can sit on another column: ADVICE r3)
                m_init = o["spins"][0][int(np.argmin(er[0]))].astype(np.float64)
            else:
                # strided recording (M_skip > 1): not every column came back; the first minimum of the energies the kernel tracked
                # over ALL sweeps (exact integers of the fixed-point model in the "f32" arithmetic, fp64 in the others)
                m_init = o["argmin_state"][0].astype(np.float64)
            return o

        def detect(ms):
            cl = self._detect_clusters(inst, graph, epsilon, ms, lambda_start, lambda_end, lambda_reduction_factor,
                                       tolerance, max_iterations, threshold_initial, threshold_cutoff, global_beta)[0]
            print(f"\ncluster size = {len(cl)}\n")
            return cl

        if self._variant == "npt" and not provided:      # NPT/npt.py:397-403: LBP once per call
            all_clusters = detect(m_star)
        if self.rng == "philox" and N >= 256:             # (device RNG: the phases' level schedules do not depend on the spins)
            eng.plan_ahead(self._sweep_counter, sum(2 + (1 if c % full_update_frequency == 0 else 0) for c in range(num_cycles)),
                           S, self.seed)
        try:
            for cycle in range(num_cycles):
                if self._variant == "nmc":                   # NMC/nmc.py:365-373: clusters re-detected every cycle
                    print(f'\nCurrent iteration = {cycle + 1}')
                    if not provided:
                        all_clusters = detect(m_star)
                phase("C", m_init)
                phase("NC", m_init)
                if cycle % full_update_frequency == 0:
                    o = phase("ALL", m_init)
                    if self._variant == "nmc":
                        m_star = m_init.copy()
                        print(f'\ncurrent m_star energy = {o["min_energy"][0]:.8f}')
        finally:
            eng.plan_ahead(None, 0, 0, 0)
        M_overall = M_overall[:, :at]
        energy_overall = energy_overall[:at]
        min_energy = np.min(energy_overall)
        return M_overall, energy_overall, min_energy, all_clusters

