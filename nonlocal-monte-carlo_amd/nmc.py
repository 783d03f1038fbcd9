"""Drop-in for the reference's `NMC/nmc.py`: class NMC(J, h) with MCMC / LBP_convexified / NMC_subroutine / run.

Same names, argument meaning, return shapes and error behaviour as the reference (NMC/nmc.py:13-520); the sweeps,
energies and argmin hand-offs run in the HIP engine behind the C-ABI (include/nlmc.h).  See base.py for the two
RNG modes (`rng="numpy"` reproduces the reference bit for bit under `np.random.seed`).
"""
import numpy as np

from . import hostlogic
from .base import Common


class NMC(Common):
    """The NMC class implements the Non-equilibrium (non-local) Monte Carlo algorithm (NMC/nmc.py:13)."""
    _variant = "nmc"

    # ------------------------------------------------------------------------------------------------
    def run(self, num_sweeps_initial=int(1e4), num_sweeps_per_NMC_phase=int(1e4), num_NMC_cycles=10,
            full_update_frequency=1, M_skip=1, temp_x=20, global_beta=2.5, lambda_start=0.5, lambda_end=0.01,
            lambda_reduction_factor=0.9, threshold_initial=0.999999, threshold_cutoff=0.99999, max_iterations=100,
            tolerance=np.finfo(float).eps, use_hash_table=False, plot=False):
        """Execute the NMC algorithm (NMC/nmc.py:442-520).  Returns (M_overall, energy_overall, min_energy)."""
        norm_factor = np.max(np.abs(self.J))            # NMC/nmc.py:474-476: run() mutates self.J / self.h
        self.J = self.J / norm_factor
        self.h = self.h / norm_factor
        N = len(self.h)
        if self.rng == "numpy":
            m_init = np.sign(2 * np.random.rand(N) - 1)
        else:
            m_init = np.sign(2 * np.random.default_rng(self.seed).random(N) - 1)
        eng = self._cache.engine(self.J, self.h, 1)
        sched = hostlogic.beta_schedule(int(num_sweeps_initial), global_beta, True, 1, 0)
        if num_sweeps_initial < 0:
            raise ValueError("negative dimensions are not allowed")   # np.zeros((N, -k)) in the reference (:52)
        o = self._mcmc_on(eng, int(num_sweeps_initial), m_init, sched, record=False)
        m_star = o["argmin_state"][0].astype(np.float64)
        print(f'\ninitial m_star energy = {o["min_energy"][0]:.8f}')
        M_overall, energy_overall, min_energy, all_clusters = self.NMC_subroutine(
            m_star, num_NMC_cycles, num_sweeps_per_NMC_phase, full_update_frequency, M_skip, global_beta, temp_x,
            lambda_start, lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations,
            tolerance, hash_table=None, use_hash_table=use_hash_table)
        if plot:
            self.plot_results(M_overall, energy_overall, all_clusters, M_skip, num_NMC_cycles, full_update_frequency,
                              num_sweeps_per_NMC_phase)
        return M_overall, energy_overall, min_energy

    # ------------------------------------------------------------------------------------------------
    def run_restarts(self, num_restarts, num_sweeps_initial=int(1e4), num_sweeps_per_NMC_phase=int(1e4), num_NMC_cycles=10,
                     full_update_frequency=1, temp_x=20, global_beta=2.5, lambda_start=0.5, lambda_end=0.01,
                     lambda_reduction_factor=0.9, threshold_initial=0.999999, threshold_cutoff=0.99999,
                     max_iterations=100, tolerance=np.finfo(float).eps, all_clusters=None, _force_host=False):
        """Throughput extension (not in the reference): `num_restarts` independent NMC runs of run()'s algorithm batched
        in ONE context -- anneal, then per cycle the three phases (NMC/nmc.py:365-433), every phase one launch over all
        restarts with per-restart cluster flags, the argmin-energy state of each phase handed to the next.  Device RNG
        (philox) whatever `rng` says, "f32" arithmetic (24-bit fixed-point couplings).  `all_clusters`: one index array used by
        every restart and cycle (skips the host-side backbone inference), or None to infer clusters per restart and
        cycle like run() does.  Returns (min_energy [R], best_state [R, N] int8, energy_of_phase_minima [R, phases])."""
        norm_factor = np.max(np.abs(self.J))
        self.J = self.J / norm_factor
        self.h = self.h / norm_factor
        inst = self._cache.instance(self.J, self.h)
        R, N = int(num_restarts), inst.n
        S0, S = int(num_sweeps_initial), int(num_sweeps_per_NMC_phase)
        eng = self._cache.engine(self.J, self.h, R)
        m = np.sign(2 * np.random.default_rng(self.seed).random((R, N)) - 1).astype(np.int8)
        args = (lambda_start, lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations, tolerance)
        if not _force_host and full_update_frequency >= 1 and (all_clusters is not None or self.lbp == "device"):     # (lbp="host": the bit-exact host inference)
            return self._run_restarts_device(eng, inst, m, S0, S, num_NMC_cycles, int(full_update_frequency), temp_x, global_beta, all_clusters, *args)
        return self._run_restarts_host(eng, inst, m, S0, S, num_NMC_cycles, full_update_frequency, temp_x, global_beta,
                                       all_clusters, *args)

    def _run_restarts_device(self, eng, inst, m, S0, S, num_NMC_cycles, full_update_frequency, temp_x, global_beta, all_clusters, lambda_start,
                             lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations, tolerance):
        """The hand-offs, the inference seeds, the cluster masks and the phase flags all stay on the device (include/nlmc.h:
        nlmc_adopt_best, nlmc_backbone_clusters, nlmc_set_phase); per launch only the minima and argmin states come back for the
        run's own best-of bookkeeping.  The state a cycle's backbone inference is seeded with (m_star, NMC/nmc.py:368-373,433) is the
        state after the last PLAIN phase: with full_update_frequency == 1 the state the cycle starts from, otherwise an older one --
        kept aside on the device (nlmc_backbone_seed) whenever a plain phase (or the anneal) ends."""
        from .lbp import lambda_list, _SAT, EPS as _EPS
        R, N = m.shape
        sweep0 = self._sweep_counter
        best_e, best_s, trail = np.full(R, np.inf), m.copy(), []
        eng.select("all")
        eng.set_spins(m)
        eng.set_flags(None)

        def launch(n_sweeps, beta):
            nonlocal sweep0, best_e
            o = eng.sweep_philox_windows(n_sweeps, self.seed, sweep0=sweep0, beta=beta, want_min=True, want_state=True)
            sweep0 += n_sweeps
            better = o["min_energy"] < best_e
            best_e = np.where(better, o["min_energy"], best_e)
            best_s[better] = o["argmin_state"][better]
            trail.append(o["min_energy"].copy())
            eng.adopt_best()                     # the next launch starts from the argmin column (NMC/nmc.py:394-395)
            eng.energy()                         # ... with its fp64 energy as the tracked one (what set_spins did per launch)

        if S0 > 0:
            sched = hostlogic.beta_schedule(S0, global_beta, True, 1, 0)
            launch(S0, np.repeat(sched[None, :], R, axis=0))
        if S > 0 and num_NMC_cycles > 0:
            eng.backbone_seed(True)                  # m_star = the state after the anneal (NMC/nmc.py:364-365)
            if all_clusters is None:
                lams = lambda_list(lambda_start, lambda_end, lambda_reduction_factor)
                if not lams:
                    raise TypeError("bad operand type for abs(): 'NoneType'")
                thr, t = [float(threshold_initial)], threshold_initial - 0.01
                while t > threshold_cutoff:
                    thr.append(float(t))
                    t -= 0.01
                epsilon = self._graph(inst).epsilon(inst.h)
            else:
                mask = np.zeros((R, N), dtype=np.uint8)
                mask[:, np.asarray(all_clusters, dtype=int)] = 1
                eng.set_cluster_mask(mask)
            n_launches = sum(2 + (1 if cycle % full_update_frequency == 0 else 0) for cycle in range(num_NMC_cycles))
            eng.plan_ahead(sweep0, n_launches, S, self.seed)              # the phase launches' windows, planned together
            try:
                for cycle in range(num_NMC_cycles):
                    if all_clusters is None:
                        eng.backbone_clusters(epsilon, lams, global_beta, tolerance, max_iterations, _SAT - _EPS, thr)
                    plain = cycle % full_update_frequency == 0            # NMC/nmc.py:419: the third phase of this cycle
                    for kind in ("C", "NC") + (("ALL",) if plain else ()):
                        eng.set_phase(kind, temp_x)
                        launch(S, float(global_beta))
                    if plain:
                        eng.backbone_seed(True)                           # m_star = m_init after the plain phase (NMC/nmc.py:433)
                eng.backbone_check()
            finally:
                eng.plan_ahead(None, 0, 0, 0)
                eng.set_flags(None)
                eng.backbone_seed(False)
        self._sweep_counter = sweep0
        # the running minima were tracked in the fixed-point model; report the fp64 energies of the kept states
        return eng.energy_of(best_s), best_s, np.stack(trail, axis=1) if trail else np.zeros((R, 0))

    def _run_restarts_host(self, eng, inst, m, S0, S, num_NMC_cycles, full_update_frequency, temp_x, global_beta, all_clusters,
                           lambda_start, lambda_end, lambda_reduction_factor, threshold_initial, threshold_cutoff, max_iterations,
                           tolerance):
        """Cycles without a plain phase keep an older m_star for the inference than the state the phases continue from: the
        hand-offs go through the host (state and flags uploaded per launch)."""
        R, N = m.shape
        graph = self._graph(inst) if all_clusters is None else None
        epsilon = graph.epsilon(inst.h) if all_clusters is None else None
        sweep0 = self._sweep_counter
        best_e = np.full(R, np.inf)
        best_s = m.copy()
        trail = []

        def launch(state, n_sweeps, beta_tab, flags):
            nonlocal sweep0, best_e, best_s
            eng.set_spins(state)
            eng.set_flags(flags, temp_x)
            # fused windows planned piece by piece against a memory budget (Engine.sweep_philox_windows; instances / lengths
            # the fused kernels do not take run sweep by sweep, same bits)
            o = eng.sweep_philox_windows(n_sweeps, self.seed, sweep0=sweep0, beta=beta_tab, want_min=True, want_state=True)
            sweep0 += n_sweeps
            better = o["min_energy"] < best_e
            best_e = np.where(better, o["min_energy"], best_e)
            best_s[better] = o["argmin_state"][better]
            trail.append(o["min_energy"].copy())
            return o["argmin_state"].copy()

        if S0 > 0:
            sched = hostlogic.beta_schedule(S0, global_beta, True, 1, 0)
            m = launch(m, S0, np.repeat(sched[None, :], R, axis=0), None)
        m_star = m.copy()
        flat = np.full((R, S), float(global_beta)) if S > 0 else None
        n_launches = sum(2 + (1 if cycle % full_update_frequency == 0 else 0) for cycle in range(num_NMC_cycles))
        eng.plan_ahead(sweep0, n_launches, S, self.seed)
        for cycle in range(num_NMC_cycles):
            if S == 0:
                break
            if all_clusters is None:
                cls = self._detect_clusters(inst, graph, epsilon, m_star.astype(float), lambda_start, lambda_end,
                                            lambda_reduction_factor, tolerance, max_iterations, threshold_initial,
                                            threshold_cutoff, global_beta)
            else:
                cls = [np.asarray(all_clusters, dtype=int)] * R
            m = launch(m, S, flat, np.stack([hostlogic.phase_flags(N, m[r], cls[r], "C") for r in range(R)]))
            m = launch(m, S, flat, np.stack([hostlogic.phase_flags(N, m[r], cls[r], "NC") for r in range(R)]))
            if cycle % full_update_frequency == 0:
                m = launch(m, S, flat, None)
                m_star = m.copy()
        eng.plan_ahead(None, 0, 0, 0)
        eng.set_flags(None)
        self._sweep_counter = sweep0
        # the running minima were tracked incrementally from fp32 fields; report the fp64 energies of the kept states
        return eng.energy_of(best_s), best_s, np.stack(trail, axis=1) if trail else np.zeros((R, 0))

    def plot_results(self, M_overall, energy_overall, all_clusters, M_skip, num_NMC_cycles, full_update_frequency,
                     num_sweeps_per_NMC_phase):
        """Energy trace + spin raster (presentation only; the reference writes NMC_energy.png / NMC_spins.png)."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(2, 1, figsize=(10, 8))
        ax[0].imshow(M_overall, aspect='auto', cmap='viridis')
        ax[0].set_ylabel('spin')
        ax[1].plot(energy_overall)
        ax[1].set_xlabel(f'recorded sweep (every {M_skip})')
        ax[1].set_ylabel('Energy')
        fig.savefig('NMC_energy.png')
        plt.close(fig)
