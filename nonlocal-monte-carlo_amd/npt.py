"""Drop-in for the reference's `NPT/npt.py`: class NPT(J, h) -- Non-equilibrium Monte Carlo + parallel tempering.

Same names, argument meaning, return shapes and error behaviour as NPT/npt.py:15-700.  Where the reference hands one
task per replica to a process pool (NPT/npt.py:616-640), this class advances ALL replicas of a round in batched
launches of the HIP engine; `num_cores` is accepted and ignored.

rng="numpy" (default): the legacy NumPy stream and stdlib `random` are consumed in exactly the reference's in-order
program order (replica 1's sweeps, replica 2's, ..., then pair selection, then one rand() per attempted swap), so
after `np.random.seed(s); random.seed(s)` M, Energy and the swap log equal the reference run with its pool executed
in order (the pool itself is not reproducible for num_cores > 1).

rng="philox": throughput mode.  The whole run stays on the device: replica states never leave HBM, swaps are label
exchanges decided by a kernel (include/nlmc.h: nlmc_pt_swap_philox), replicas on doNMC slots run their NMC_task
(backbone inference, phase flags, argmin hand-offs; include/nlmc.h: nlmc_pt_mark_slots ... nlmc_set_phase) without a host
round trip, `num_restarts` independent ladders advance in the same launches, and only the last round's trace is read back
for the return value.
"""
import random as _pyrandom

import os

import numpy as np

from . import hostlogic
from .base import Common
from .engine import Engine, prefault_async, trace_layout


class NPT(Common):
    """NMC + Adaptive Parallel Tempering (NPT/npt.py:15)."""
    _variant = "npt"

    # ------------------------------------------------------------------------------------------------
    def replica_energy(self, M, num_sweeps):
        """min and trace of E = -(m^T J m / 2 + m^T h) over the first num_sweeps columns (NPT/npt.py:31-45)."""
        eng = self._cache.engine(self.J, self.h, 1)
        cols = np.stack([np.asarray(M)[:, ii] for ii in range(num_sweeps)]) if num_sweeps > 0 else np.zeros((0, eng.n))
        EE1 = eng.energy_of(cols.astype(np.int8)) if num_sweeps > 0 else np.zeros(0)
        return np.min(EE1), EE1

    def MCMC_task(self, replica_i, num_sweeps_MCMC, m_start, beta_list, use_hash_table=False, hash_table=None):
        """NPT/npt.py:112-127 (1-based replica index)."""
        return self.MCMC(num_sweeps_MCMC, m_start.copy(), beta_list[replica_i - 1], self.J, self.h,
                         hash_table=hash_table, use_hash_table=use_hash_table)

    def NMC_task(self, m_start, num_cycles, num_sweeps_per_NMC_phase, full_update_frequency, M_skip, global_beta,
                 temp_x, lambda_start, lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff,
                 max_iterations, tolerance, use_hash_table=False, hash_table=None):
        """NPT/npt.py:479-512."""
        M_overall, _, _, _ = self.NMC_subroutine(
            m_start, num_cycles, num_sweeps_per_NMC_phase, full_update_frequency, M_skip, global_beta, temp_x,
            lambda_start, lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations,
            tolerance, hash_table=hash_table, use_hash_table=use_hash_table)
        return M_overall

    def select_non_overlapping_pairs(self, all_pairs):
        """NPT/npt.py:514-533: num_swapping_pairs adjacent pairs without a shared replica, drawn with stdlib
        `random.randint` (a stream separate from NumPy's)."""
        available = all_pairs.copy()
        selected = []
        for _ in range(self.num_swapping_pairs):
            if not available:
                raise ValueError("Cannot find non-overlapping pairs.")
            pair = available[_pyrandom.randint(0, len(available) - 1)]
            selected.append(pair)
            available = [p for p in available if p[0] not in pair and p[1] not in pair]
        return selected

    # ------------------------------------------------------------------------------------------------
    def run(self, beta_list, num_replicas, doNMC, num_sweeps_MCMC=1000, num_sweeps_read=1000, num_swap_attempts=100,
            num_swapping_pairs=1, num_cycles=10, full_update_frequency=1, M_skip=1, temp_x=20, global_beta=2.5,
            lambda_start=0.5, lambda_end=0.01, lambda_reduction_factor=0.9, threshold_initial=0.999999,
            threshold_cutoff=0.99999, max_iterations=100, tolerance=np.finfo(float).eps, use_hash_table=False,
            num_cores=8, plot=False, num_restarts=1, device_ids=None, return_trace="float64"):
        """Run the NPT algorithm (NPT/npt.py:535-700).  Returns (M [R*N, S_swap], Energy [R]).

        Additive keyword arguments (defaults = the reference's behaviour): `num_restarts` independent ladders advanced
        together and `device_ids` (GPUs the chains are sharded over; the reference's knob is num_cores, NPT/npt.py:616)
        -- both for rng="philox" runs (with NMC replicas: M_skip == 1); M and Energy describe restart 0,
        `self.restart_energies` [num_restarts, R] all of them.  `return_trace`: "float64" (reference dtype), "int8", or
        None (M is not built).  Dynamics of the philox mode: 24-bit fixed-point couplings + logistic thresholds (DESIGN.md
        section 2 states the tolerance); the argmin hand-off between NMC phases uses the energies tracked in that model."""
        self.num_replicas = num_replicas
        self.num_sweeps_MCMC = num_sweeps_MCMC
        self.num_sweeps_read = num_sweeps_read
        self.num_swap_attempts = num_swap_attempts
        self.num_sweeps_MCMC_per_swap = self.num_sweeps_MCMC // self.num_swap_attempts
        self.num_sweeps_read_per_swap = self.num_sweeps_read // self.num_swap_attempts
        self.num_sweeps_per_NMC_phase_per_swap = int(np.ceil(self.num_sweeps_MCMC / self.num_swap_attempts / 3 / num_cycles))
        self.num_swapping_pairs = num_swapping_pairs
        self.use_hash_table = use_hash_table
        self.doNMC = doNMC

        norm_factor = np.max(np.abs(self.J))            # NPT/npt.py:588-590: run() mutates self.J / self.h
        self.J = self.J / norm_factor
        self.h = self.h / norm_factor
        if len(self.doNMC) != self.num_replicas:
            raise ValueError("The length of doNMC does not match the number of replicas.")
        beta_list = np.asarray(beta_list, dtype=np.float64)

        if return_trace not in ("float64", "int8", None):
            raise ValueError("return_trace must be 'float64', 'int8' or None")
        any_nmc = any(bool(v) for v in doNMC)
        # (with NMC replicas and M_skip > 1 the phases' traces and argmin hand-offs are strided on the device; a phase length that M_skip
        # does not divide raises the reference's own shape error on the host-managed path)
        device_resident = self.rng == "philox" and (not any_nmc or (int(M_skip) >= 1 and self.num_sweeps_per_NMC_phase_per_swap % int(M_skip) == 0))
        if (int(num_restarts) != 1 or (device_ids is not None and len(list(device_ids)) > 1)) and not device_resident:
            raise ValueError("num_restarts / device_ids need rng='philox' (and M_skip == 1 with NMC replicas): the "
                             "reference's own stream order has one ladder in one process")
        if int(num_restarts) < 1:
            raise ValueError("num_restarts must be >= 1")
        if device_resident:
            nmc = None
            if any_nmc:
                nmc = dict(num_cycles=num_cycles, full_update_frequency=full_update_frequency, temp_x=temp_x,
                           global_beta=global_beta, lambda_start=lambda_start, lambda_end=lambda_end,
                           lambda_reduction_factor=lambda_reduction_factor, threshold_initial=threshold_initial,
                           threshold_cutoff=threshold_cutoff, max_iterations=max_iterations, tolerance=tolerance, M_skip=int(M_skip))
            M, Energy = self._run_device_resident(beta_list, int(num_restarts), device_ids, return_trace, nmc)
        else:
            M, Energy = self._run_host_managed(beta_list, num_cycles, full_update_frequency, M_skip, temp_x, global_beta,
                                               lambda_start, lambda_end, lambda_reduction_factor, threshold_initial,
                                               threshold_cutoff, max_iterations, tolerance)
        print(f"\nLatest energy from each replica = {Energy}")
        if plot and M is not None:
            self.plot_energies([self.replica_energy(M[r * self._n:(r + 1) * self._n, :], self.num_sweeps_read_per_swap)[1]
                                for r in range(num_replicas)], beta_list)
        return M, Energy

    # ------------------------------------------------------------------------------------------------
    def _run_host_managed(self, beta_list, num_cycles, full_update_frequency, M_skip, temp_x, global_beta, lambda_start,
                          lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations,
                          tolerance):
        """The reference's bookkeeping (m_start / M arrays on the host), sweeps batched on the GPU."""
        inst = self._cache.instance(self.J, self.h)
        R, N = self.num_replicas, inst.n
        self._n = N
        S = self.num_sweeps_MCMC_per_swap
        S_nmc = self.num_sweeps_per_NMC_phase_per_swap
        all_pairs = [(i, i + 1) for i in range(1, R)]
        M = np.zeros((R * N, S))
        numpy_mode = self.rng == "numpy"
        host_rng = None if numpy_mode else np.random.default_rng(self.seed)
        if numpy_mode:
            m_start = np.sign(2 * np.random.rand(R * N, 1) - 1)
        else:
            m_start = np.sign(2 * host_rng.random((R * N, 1)) - 1)
        mc = [r for r in range(R) if not self.doNMC[r]]
        nm = [r for r in range(R) if self.doNMC[r]]
        eng_m = Engine(inst, None, len(mc), device=self._cache.device) if mc else None
        eng_n = Engine(inst, None, len(nm), device=self._cache.device) if nm else None
        eng_e = self._cache.engine(self.J, self.h, 1)
        graph = self._graph(inst) if nm else None
        epsilon = graph.epsilon(inst.h) if nm else None
        phases = []
        for cycle in range(num_cycles):
            phases += ["C", "NC"] + (["ALL"] if cycle % full_update_frequency == 0 else [])
        log_pairs, log_acc = [], []
        count = np.zeros(self.num_swap_attempts)
        drawer = None
        try:
            w = S_nmc // M_skip
            if nm and S > 0 and len(phases) * w < S:       # NPT/npt.py:643-644: M[block] = M_nmc[:, -S:] cannot be filled
                raise ValueError(f"could not broadcast input array from shape ({N},{len(phases) * w}) into shape ({N},{S})")
            # The np.random calls of a round are the same whatever the spins do: per replica, in order, one permutation + N
            # uniforms per sweep (NPT/npt.py:622-640 run in order), then one uniform per selected pair (NPT/npt.py:668).  In the
            # default mode they are therefore made one round AHEAD on a worker thread, in the reference's order, while the GPU
            # sweeps the current round (they were 61 % of the wall time at N = 10^3 x 8 replicas).
            def draw_round():
                st = {r: hostlogic.draw_legacy_stream(len(phases) * S_nmc if self.doNMC[r] else S, N) for r in range(R)}
                return st, [np.random.rand() for _ in range(self.num_swapping_pairs)]

            fut = None
            if numpy_mode and self.num_swap_attempts > 0:
                from concurrent.futures import ThreadPoolExecutor
                drawer = ThreadPoolExecutor(max_workers=1)
                fut = drawer.submit(draw_round)
            for ii in range(self.num_swap_attempts):
                print(f"\nRunning swap attempt = {ii + 1}")
                # Only the LAST column of a round's block is ever read again (next start state, swap energies) -- except
                # in the final round, whose block is what run() returns: traces are materialised there only.
                last_round = ii == self.num_swap_attempts - 1
                rs = 1 if last_round else 0
                # --- draws in the reference's program order: replica by replica (NPT/npt.py:622-640 run in order)
                streams, swap_u = {}, []
                if numpy_mode:
                    streams, swap_u = fut.result()
                    if not last_round:
                        fut = drawer.submit(draw_round)
                # --- plain replicas: one launch for all of them
                if mc:
                    eng_m.set_flags(None)
                    eng_m.set_spins(np.stack([m_start[r * N:(r + 1) * N, 0] for r in mc]).astype(np.int8))
                    btab = np.repeat(beta_list[mc][:, None], S, axis=1)
                    if numpy_mode:
                        o = eng_m.sweep_stream(np.stack([streams[r][0] for r in mc]), np.stack([streams[r][1] for r in mc]),
                                               btab, record_stride=rs)
                    else:
                        o = eng_m.sweep_philox(S, self.seed, sweep0=self._sweep_counter, beta=btab, record_stride=rs)
                    fin = None if last_round else eng_m.get_spins()
                    for i, r in enumerate(mc):
                        if last_round:
                            M[r * N:(r + 1) * N, :] = o["spins"][i].T
                        else:
                            M[r * N:(r + 1) * N, -1] = fin[i]
                # --- NMC replicas: backbone per replica (one batch), then the phases in lock step
                if nm:
                    clusters = self._detect_clusters(inst, graph, epsilon,
                                                     np.stack([m_start[r * N:(r + 1) * N, 0] for r in nm]), lambda_start,
                                                     lambda_end, lambda_reduction_factor, tolerance, max_iterations,
                                                     threshold_initial, threshold_cutoff, global_beta)
                    m_init = np.stack([m_start[r * N:(r + 1) * N, 0] for r in nm]).astype(np.int8)
                    rec_n = last_round or M_skip != 1          # strided traces: keep the reference's column arithmetic
                    traces = [np.zeros((N, len(phases) * w)) for _ in nm] if rec_n else None
                    at = 0
                    for p, kind in enumerate(phases):
                        fl = np.stack([hostlogic.phase_flags(N, m_init[i], clusters[i], kind) for i in range(len(nm))])
                        eng_n.set_spins(m_init)
                        eng_n.set_flags(None if kind == "ALL" else fl, temp_x)
                        btab = np.full((len(nm), S_nmc), float(global_beta))
                        if numpy_mode:
                            pp = np.stack([streams[r][0][p * S_nmc:(p + 1) * S_nmc] for r in nm])
                            uu = np.stack([streams[r][1][p * S_nmc:(p + 1) * S_nmc] for r in nm])
                            o = eng_n.sweep_stream(pp, uu, btab, record_stride=int(rec_n), want_min=True, want_state=True)
                        else:
                            o = eng_n.sweep_philox(S_nmc, self.seed, sweep0=self._sweep_counter + S + p * S_nmc, beta=btab,
                                                   order="per_chain", record_stride=int(rec_n), want_min=True,
                                                   want_state=True)
                        if rec_n:
                            for i in range(len(nm)):
                                traces[i][:, at:at + w] = o["spins"][i].T[:, ::M_skip]
                        at += w
                        m_init = o["argmin_state"].copy()
                    if rec_n:
                        for i, r in enumerate(nm):
                            M[r * N:(r + 1) * N, :] = traces[i][:, -S:].copy()     # NPT/npt.py:643-644
                    else:
                        fin = eng_n.get_spins()                # last trace column = state after the last phase
                        for i, r in enumerate(nm):
                            M[r * N:(r + 1) * N, -1] = fin[i]
                if not numpy_mode:
                    self._sweep_counter += S + len(phases) * S_nmc
                # --- swap step on the host, exactly as NPT/npt.py:646-680
                m_start = M[:, -1].copy().reshape(-1, 1)
                selected = self.select_non_overlapping_pairs(all_pairs) if numpy_mode else \
                    self._select_pairs_host_rng(all_pairs, host_rng)
                for (sel, nxt) in selected:
                    m_sel = M[(sel - 1) * N:sel * N, -1].copy()
                    m_next = M[(nxt - 1) * N:nxt * N, -1].copy()
                    E = eng_e.energy_of(np.stack([m_sel, m_next]).astype(np.int8))
                    E_sel, E_next = E[0], E[1]
                    beta_sel, beta_next = beta_list[sel - 1], beta_list[nxt - 1]
                    print(f"\nSelected pair indices: {sel}, {nxt}")
                    print(f"β values: {beta_sel}, {beta_next}")
                    print(f"Energies: {E_sel}, {E_next}")
                    log_pairs.append((sel, nxt))
                    u = swap_u[len(log_acc) - ii * self.num_swapping_pairs] if numpy_mode else host_rng.random()
                    ok = u < min(1, np.exp((beta_next - beta_sel) * (E_next - E_sel)))
                    log_acc.append(int(ok))
                    if ok:
                        count[ii] += 1
                        print(f"Swapping {int(sum(count))}th time")
                        m_start[(sel - 1) * N:sel * N] = m_next.reshape(-1, 1)
                        m_start[(nxt - 1) * N:nxt * N] = m_sel.reshape(-1, 1)
            Energy = np.zeros(R)
            for r in range(R):                       # min over the FIRST R_swap columns (NPT/npt.py:685-692, :41)
                Energy[r] = self.replica_energy(M[r * N:(r + 1) * N, :], self.num_sweeps_read_per_swap)[0]
        finally:
            if drawer is not None:
                drawer.shutdown(wait=True)
            for e in (eng_m, eng_n):
                if e is not None:
                    e.close()
        self.swap_pairs = np.array(log_pairs, dtype=np.int32).reshape(-1, 2)
        self.swap_accepted = np.array(log_acc, dtype=np.int8)
        print(f"Swap acceptance rate = {np.count_nonzero(count) / max(1, count.size) * 100:.2f} per cent\n")
        return M, Energy

    def _select_pairs_host_rng(self, all_pairs, rng):
        available = all_pairs.copy()
        selected = []
        for _ in range(self.num_swapping_pairs):
            if not available:
                raise ValueError("Cannot find non-overlapping pairs.")
            pair = available[int(rng.integers(0, len(available)))]
            selected.append(pair)
            available = [p for p in available if p[0] not in pair and p[1] not in pair]
        return selected

    # ------------------------------------------------------------------------------------------------
    def _run_device_resident(self, beta_list, n_restarts=1, device_ids=None, return_trace="float64", nmc=None):
        """Throughput path: all replicas (x restarts) stay on the device(s); label-exchange swaps decided by a kernel; the
        swap log is kept on the device and read once; only the last round's trace comes back (as int8 until the
        reference-shaped float64 M is asked for).  `nmc`: NMC_task parameters when some slots have doNMC set -- those
        chains run backbone inference + the NMC phases inside every round, on the device (distributed.LocalTempering)."""
        from .distributed import LocalTempering, ShardedTempering, ShardedAsLocal, block_partition, launcher_context, all_reduce_np
        from .lbp import lambda_list, _SAT, EPS as _EPS
        inst = self._cache.instance(self.J, self.h)
        R, N = self.num_replicas, inst.n
        self._n = N
        S = self.num_sweeps_MCMC_per_swap
        rounds = self.num_swap_attempts
        G = R * n_restarts
        devs = [self._cache.device] if not device_ids else [int(d) for d in device_ids]
        # One rank of a torch.distributed.run job (RANK / WORLD_SIZE set, device_ids not given): the chains are sharded over the
        # RANKS, one GPU each.  Whole ladders per rank (num_restarts % ranks == 0): every rank decides its own ladders' swaps, no
        # collective, NMC slots allowed.  A ladder cut across ranks: ONE all-gather of the energies per round, issued by the
        # library over RCCL (distributed.ShardedTempering).  M / Energy describe the first restart of the rank (rank 0: restart
        # 0; with a cut ladder every rank returns restart 0, gathered); self.restart_energies holds all restarts on every rank.
        ctx = launcher_context() if device_ids is None else None
        cut = False
        if ctx is not None:
            torch_, dist_, W_, rank_, lrank_ = ctx
            if G % W_:
                raise ValueError("num_replicas * num_restarts must be a multiple of the number of ranks")
            base_, count_ = block_partition(G, W_, rank_)
            # (NLMC_NPT_FORCE_COLLECTIVE: take the cut-ladder driver although the blocks are whole ladders -- rehearses the collective path
            # with fewer ranks than a real cut needs)
            cut = base_ % R != 0 or count_ % R != 0 or (bool(os.environ.get("NLMC_NPT_FORCE_COLLECTIVE")) and not nmc)
            if nmc and cut:
                raise ValueError("with NMC replicas every ladder must lie on one rank: num_restarts % ranks != 0")
            devs = [lrank_]
        if G % len(devs):
            raise ValueError("num_replicas * num_restarts must be a multiple of len(device_ids)")
        if nmc and n_restarts % len(devs):
            raise ValueError("with NMC replicas every ladder must lie on one device: num_restarts % len(device_ids) != 0")
        phases, S_nmc = [], self.num_sweeps_per_NMC_phase_per_swap
        if nmc:
            for cycle in range(nmc["num_cycles"]):
                phases += ["C", "NC"] + (["ALL"] if cycle % nmc["full_update_frequency"] == 0 else [])
            w_nmc = S_nmc // nmc["M_skip"]                 # recorded columns per phase (NPT/npt.py:434-435)
            if S > 0 and len(phases) * w_nmc < S:          # NPT/npt.py:643-644: M[block] = M_nmc[:, -S:] cannot be filled
                raise ValueError(f"could not broadcast input array from shape ({N},{len(phases) * w_nmc}) into shape ({N},{S})")
            lams = lambda_list(nmc["lambda_start"], nmc["lambda_end"], nmc["lambda_reduction_factor"])
            if not lams:                # the reference's lambda loop never runs: find_clusters(None) raises TypeError there
                raise TypeError("bad operand type for abs(): 'NoneType'")
            thr, t = [float(nmc["threshold_initial"])], nmc["threshold_initial"] - 0.01     # NMC/nmc.py:300-316
            while t > nmc["threshold_cutoff"]:
                thr.append(float(t))
                t -= 0.01
        if ctx is None:
            lt = LocalTempering(inst, beta_list, G, self.seed, self.num_swapping_pairs, devs)
        elif not cut:
            lt = LocalTempering(inst, beta_list, G, self.seed, self.num_swapping_pairs, devs, parts=[(base_, count_)])
        else:
            lt = ShardedAsLocal(ShardedTempering(
                lambda i, n, b, g: Engine(i, None, n, device=lrank_, chain_base=b, n_chains_global=g), inst, beta_list, G, self.seed,
                self.num_swapping_pairs, torch=torch_, dist=dist_, device=torch_.device("cuda", lrank_)))
        lo_, hi_ = (0, G) if (ctx is None or cut) else (base_, base_ + count_)       # global chains the arrays below cover
        M_buf = M_touch = None
        if return_trace == "float64" and R * N * S * 8 >= (32 << 20):
            # the reference-shaped float64 M of the return value: allocated now, its pages faulted in by a background thread
            # while the GPU sweeps (first touch of 205 MB at the C4 shape cost 14 of 47 ms after the run)
            M_buf = np.empty((R * N, S), dtype=np.float64)
            M_touch = prefault_async(M_buf)
        try:
            if nmc:
                lt.configure_nmc(self.doNMC, phases, S_nmc, nmc["global_beta"], nmc["temp_x"], self._graph(inst).epsilon(inst.h),
                                 lams, nmc["tolerance"], nmc["max_iterations"], _SAT - _EPS, thr, M_skip=nmc["M_skip"])
            m0 = np.random.default_rng(self.seed).integers(0, 2, size=(G, N), dtype=np.int8)
            m0 <<= 1                                     # 2 b - 1 in place (the same states as `2 * b - 1`, two passes over G x N bytes less)
            m0 -= 1
            lt.set_spins(m0)
            lt.sweeps_done = self._sweep_counter
            lt.nmc_sweeps_done = getattr(self, "_nmc_sweep_counter", 0)
            lt.plan(rounds * S, rounds)
            if self.num_swapping_pairs > 0:
                lt.log_begin(rounds)
            k = self.num_sweeps_read_per_swap
            last, e_last, E_cols, slots_last = None, None, None, np.arange(G, dtype=np.int32) % R
            for ii in range(rounds):
                is_last = ii == rounds - 1
                if is_last:
                    slots_last = lt.slots()
                if not is_last:
                    lt.round(S)
                elif return_trace is not None:
                    outs = lt.round(S, record_stride=1, energy_columns=min(k, S) if (0 < k and S > 0) else 0)
                    if nmc:
                        last, E_cols = self._assemble_mixed(lt, outs, S, N, min(k, S) if (0 < k and S > 0) else 0, "spins")
                    else:
                        last = outs[0]["spins"] if len(outs) == 1 else np.concatenate([o["spins"] for o in outs])  # [G, S, N] int8
                        if 0 < k and S > 0:
                            # replica_energy (NPT/npt.py:31-45, :685-692) of every replica: fp64 energies of the FIRST R_swap
                            # recorded columns, computed on the device copy of the trace (one launch per context)
                            E_cols = np.concatenate([e.energy_of_recorded(min(k, S)) for e in lt.engs])
                else:
                    outs = lt.round(S, want_energy=True)
                    if nmc:
                        e_last, _ = self._assemble_mixed(lt, outs, S, N, 0, "energy")
                    else:
                        e_last = np.concatenate([o["energy"] for o in outs])   # [G, S] tracked
            if cut:                                      # every rank assembles the whole ladder's read-out
                last = None if last is None else lt.gather(last)
                E_cols = None if E_cols is None else lt.gather(E_cols)
                e_last = None if e_last is None else lt.gather(e_last)
            elif ctx is not None and nmc:                # (_assemble_mixed lays its rows out by global chain id)
                last = None if last is None else last[lo_:hi_]
                E_cols = None if E_cols is None else E_cols[lo_:hi_]
                e_last = None if e_last is None else e_last[lo_:hi_]
            gch = np.arange(lo_, hi_)                    # global ids of the chains the arrays cover
            sl_ = slots_last[lo_:hi_]
            self._sweep_counter += rounds * lt.sweeps_per_round(S)
            self._nmc_sweep_counter = lt.nmc_sweeps_done          # (the NMC phases draw from a counter range of their own)
            lt.check()
            Energy = np.zeros(R)
            E_all = np.zeros((n_restarts, R))
            if last is not None and S > 0:
                if k > 0:
                    E_all[gch // R, sl_] = E_cols.min(axis=1)
                    Energy = E_all[lo_ // R].copy()
                else:
                    Energy[0] = np.min(np.zeros(0))                              # np.min of nothing: ValueError (NPT/npt.py:43)
            elif return_trace is None and S > 0 and rounds > 0:
                # no trace comes back: the read-out uses the energies tracked by the sweep kernel (exact for the
                # fixed-point couplings, within 2^-(qs+1) per coupling of the fp64 ones)
                if k <= 0:
                    Energy[0] = np.min(np.zeros(0))
                E_all[gch // R, sl_] = e_last[:, :k].min(axis=1)
                Energy = E_all[lo_ // R].copy()
            M = None
            if return_trace is not None:
                dt = np.float64 if return_trace == "float64" else np.int8
                if M_touch is not None:
                    M_touch.join()
                if last is not None:                     # restart 0 is the one returned in the reference's shape:
                    M = trace_layout(last[:R], sl_[:R], R, dt, out=M_buf)   # block r = the replica at temperature slot r
                else:
                    M = np.zeros((R * N, S), dtype=dt)
            if self.num_swapping_pairs > 0 and rounds > 0:
                p, a = lt.swap_log()                     # [rounds, ladders, pairs, 2] / [rounds, ladders, pairs]
                if ctx is not None and not cut:          # every rank logged its own ladders: all restarts on every rank
                    p = all_reduce_np(torch_, dist_, np.where(p < 0, 0, p + 1).astype(np.int32)) - 1
                    a = all_reduce_np(torch_, dist_, a.astype(np.int32)).astype(np.uint8)
                self.swap_pairs = p[:, lo_ // R].reshape(-1, 2) + 1
                self.swap_accepted = a[:, lo_ // R].reshape(-1).astype(np.int8)
                self.swap_acceptance_per_restart = a.reshape(rounds, n_restarts, -1).mean(axis=(0, 2))
                self.swap_log_all = (p, a)               # slots (0-based) / decisions of every restart
            else:
                self.swap_pairs, self.swap_accepted = np.zeros((0, 2), np.int32), np.zeros(0, np.int8)
            self.final_slots = lt.slots()
            if ctx is not None and not cut:              # the other ranks' restarts
                E_all = all_reduce_np(torch_, dist_, E_all)
            self.restart_energies = E_all
        finally:
            lt.close()
        return M, Energy

    @staticmethod
    def _assemble_mixed(lt, outs, S, N, k, key):
        """Last-round outputs of a ladder set with NMC slots -> per global chain: the block NPT.run keeps (NPT/npt.py:640-644):
        the S sweeps of a plain replica, the LAST S columns of an NMC replica's phases laid end to end.  key "spins": returns
        ([G, S, N] int8, fp64 energies of the first k columns [G, k] or None); key "energy": ([G, S] tracked energies, None)."""
        G = lt.G
        full = np.zeros((G, S, N), dtype=np.int8) if key == "spins" else np.zeros((G, S))
        E_cols = np.zeros((G, k)) if (key == "spins" and k > 0) else None
        for e, (base, _), rec in zip(lt.engs, lt.parts, outs):
            pc, nc = rec["plain_chains"], rec["nmc_chains"]
            full[base + pc] = rec["plain"][key]
            # (recorded traces come back strided already; per-sweep energies are strided here: NPT/npt.py:434-435)
            ms = lt.nmc["M_skip"] if (key == "energy" and lt.nmc) else 1
            tail = np.concatenate([o[key][:, ::ms] for o in rec["nmc"]], axis=1)[:, -S:] if S > 0 else np.zeros_like(full[base + nc])
            full[base + nc] = tail
            if E_cols is not None:
                E_cols[base + pc] = rec["plain_energy_columns"]
                E_cols[base + nc] = e.energy_of(tail[:, :k]).reshape(len(nc), k)
        return full, E_cols

    def plot_energies(self, EE1_list, beta_list):
        """NPT/npt.py:702-717 (presentation only)."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.figure()
        for i in range(self.num_replicas):
            plt.plot(EE1_list[i], label=f"Replica {i + 1} (β={beta_list[i]:.2f})")
        plt.xlabel('Sweeps')
        plt.ylabel('Energy')
        plt.title('Energy traces for different replicas')
        plt.legend()
        plt.savefig('NPT_energy.png')
        plt.close()
