"""Drop-in for the reference's `NPT/apt_ICM.py`: class APT_ICM(J, h) -- parallel tempering with Houdayer's
iso-cluster move (NPT/apt_ICM.py:14-305).

All num_replicas x 10 sub-replica chains advance in one batched launch per round; the disagreement clusters of a
sub-replica pair are found on the device (include/nlmc.h: nlmc_icm_components / nlmc_icm_move).  `num_cores` and the
hash table are accepted and ignored.

rng="numpy" (default) consumes the legacy NumPy stream and stdlib `random` in the reference's program order, so that
M, Energy and the swap log equal the reference after `np.random.seed(s); random.seed(s)` -- including its quirk
(SURVEY.md section 3.4): the cluster move edits only the FIRST recorded column of each sub-replica block of M and
never the states that seed the next round.

rng="philox": sweeps, pair picks and cluster picks are drawn on the device; `icm_feedback=True` additionally lets the
move act on the CURRENT states so that it feeds the dynamics (what Houdayer's move is meant to do).
"""
import random as _pyrandom

import numpy as np

from . import hostlogic
from .base import SweepMixin
from .engine import Engine, RoundPlanner, trace_layout


class APT_ICM(SweepMixin):
    num_subreplicas = 10      # NPT/apt_ICM.py:177
    useKatzgraber = True      # NPT/apt_ICM.py:178

    def __init__(self, J, h, rng=None, seed=None, device=0):
        self.J = J
        if isinstance(h, list):
            h = np.array(h)
        if len(h.shape) == 1:
            h = h[:, np.newaxis]                      # NPT/apt_ICM.py:27-34: h kept as an [N,1] column
        self.h = h
        self._init_backend(rng, seed, device)

    def _hflat(self):
        return np.asarray(self.h, dtype=np.float64).reshape(-1)

    # ------------------------------------------------------------------------------------------------
    def replica_energy(self, M, num_sweeps):
        """NPT/apt_ICM.py:36-50."""
        eng = self._cache.engine(self.J, self._hflat(), 1)
        cols = np.stack([np.asarray(M)[:, ii] for ii in range(num_sweeps)]) if num_sweeps > 0 else np.zeros((0, eng.n))
        EE1 = eng.energy_of(cols.astype(np.int8)) if num_sweeps > 0 else np.zeros(0)
        return np.min(EE1), EE1

    def MCMC(self, num_sweeps, m_start, beta, hash_table=None, use_hash_table=False):
        """NPT/apt_ICM.py:52-93 (J, h taken from self; fixed beta)."""
        N = self.J.shape[0]
        M = np.zeros((N, num_sweeps))
        if num_sweeps == 0:
            return M
        self._check_hash_table(hash_table, use_hash_table)
        eng = self._cache.engine(self.J, self._hflat(), 1)
        o = self._mcmc_on(eng, num_sweeps, m_start, np.full(num_sweeps, float(beta)))
        M[:, :] = o["spins"][0].T
        return M

    def select_non_overlapping_pairs(self, all_pairs):
        """NPT/apt_ICM.py:95-114."""
        available = all_pairs.copy()
        selected = []
        for _ in range(self.num_swapping_pairs):
            if not available:
                raise ValueError("Cannot find non-overlapping pairs.")
            pair = available[_pyrandom.randint(0, len(available) - 1)]
            selected.append(pair)
            available = [p for p in available if p[0] not in pair and p[1] not in pair]
        return selected

    def find_disagreement_clusters(self, state_1, state_2, J):
        """Connected components of the sub-graph induced by the spins on which the two states disagree, ordered by
        smallest member (NPT/apt_ICM.py:116-143).  Computed by the device kernel; returns a list of index lists."""
        h0 = np.zeros(J.shape[0])
        eng = self._phase_cache.engine(J, h0, 2)
        eng.set_spins(np.stack([np.asarray(state_1), np.asarray(state_2)]).astype(np.int8))
        eng.icm_components(0, 1)
        lab = eng.icm_labels()
        roots = np.unique(lab[lab >= 0])
        return [list(np.nonzero(lab == r)[0]) for r in roots]

    # ------------------------------------------------------------------------------------------------
    def run(self, beta_list, num_replicas, num_sweeps_MCMC=1000, num_sweeps_read=1000, num_swap_attempts=100,
            num_swapping_pairs=1, use_hash_table=0, num_cores=8, plot=False, icm_feedback=False, return_trace="float64",
            device_ids=None):
        """NPT/apt_ICM.py:145-305.  Returns (M [R*N, S_swap*10], Energy [R]).  `return_trace` (device-resident path only):
        "float64" (the reference's M), "int8" (same layout, 1 byte per entry) or None (M is not materialised).
        `device_ids` (rng="philox", icm_feedback=True; the reference's knob is num_cores, NPT/apt_ICM.py:145-146): the
        temperature ladder is cut into len(device_ids) slot blocks, one context each -- all sub-replicas of a temperature on one
        GPU (distributed.SlotShardedAPT); random numbers are then keyed by (sub-replica, temperature slot), the results do not
        depend on the number of blocks."""
        if return_trace not in ("float64", "int8", None):
            raise ValueError("return_trace must be 'float64', 'int8' or None")
        self.num_replicas = num_replicas
        self.num_sweeps_MCMC = num_sweeps_MCMC
        self.num_sweeps_read = num_sweeps_read
        self.num_swap_attempts = num_swap_attempts
        self.num_sweeps_MCMC_per_swap = self.num_sweeps_MCMC // self.num_swap_attempts
        self.num_sweeps_read_per_swap = self.num_sweeps_read // self.num_swap_attempts
        self.num_swapping_pairs = num_swapping_pairs
        self.use_hash_table = use_hash_table

        inst = self._cache.instance(self.J, self._hflat())
        R, K, N = num_replicas, self.num_subreplicas, inst.n
        S = self.num_sweeps_MCMC_per_swap
        beta_list = np.asarray(beta_list, dtype=np.float64)
        if device_ids is not None and not (self.rng == "philox" and icm_feedback):
            raise ValueError("device_ids needs rng='philox' and icm_feedback=True (the device-resident path)")
        if self.rng == "philox" and icm_feedback and device_ids is not None:
            return self._run_slot_sharded(inst, beta_list, plot, return_trace, list(device_ids))
        if self.rng == "philox" and icm_feedback:
            return self._run_device_resident(inst, beta_list, plot, return_trace)
        numpy_mode = self.rng == "numpy"
        host_rng = None if numpy_mode else np.random.default_rng(self.seed)
        all_pairs = [(i, i + 1) for i in range(1, R)]
        M = np.zeros((N * R, S * K))
        m_start = np.sign(2 * (np.random.rand(N * R, K) if numpy_mode else host_rng.random((N * R, K))) - 1)
        eng = Engine(inst, None, R * K, device=self._cache.device)         # chain id = r*K + j
        eng_icm = Engine(inst, None, R * K, device=self._cache.device)     # scratch copy of the FIRST columns
        eng_e = self._cache.engine(self.J, self._hflat(), 1)
        btab = np.repeat(np.repeat(beta_list, K)[:, None], S, axis=1)
        log_pairs, log_acc = [], []
        count = np.zeros(self.num_swap_attempts)
        try:
            for ii in range(int(self.num_swap_attempts)):
                print(f"\nRunning swap attempt = {ii + 1}")
                # --- sweeps of every (replica, sub-replica) chain, program order = chain order (:197-213)
                start = np.stack([m_start[r * N:(r + 1) * N, j] for r in range(R) for j in range(K)]).astype(np.int8)
                eng.set_spins(start)
                if numpy_mode:
                    st = [hostlogic.draw_legacy_stream(S, N) for _ in range(R * K)]
                    o = eng.sweep_stream(np.stack([s[0] for s in st]), np.stack([s[1] for s in st]), btab, record_stride=1)
                else:
                    o = eng.sweep_philox(S, self.seed, sweep0=self._sweep_counter, beta=btab, record_stride=1)
                    self._sweep_counter += S
                tr = o["spins"]                                             # [R*K, S, N]
                for r in range(R):
                    for j in range(K):
                        M[r * N:(r + 1) * N, j * S:(j + 1) * S] = tr[r * K + j].T
                        m_start[r * N:(r + 1) * N, j] = tr[r * K + j][-1]
                # --- Houdayer move on the FIRST column of each sub-replica block (:215-246)
                target = eng if (icm_feedback and not numpy_mode) else eng_icm
                if target is eng_icm:
                    eng_icm.set_spins(tr[:, 0, :] if S > 0 else start)
                for r in range(R):
                    shuffled = np.random.permutation(K) if numpy_mode else host_rng.permutation(K)
                    for p in range(K // 2):
                        ja, jb = int(shuffled[2 * p]), int(shuffled[2 * p + 1])
                        ca, cb = r * K + ja, r * K + jb
                        ncl = target.icm_components(ca, cb)
                        if ncl > 0:
                            pick = int(np.random.randint(ncl)) if numpy_mode else int(host_rng.integers(0, ncl))
                            target.icm_move(ca, cb, pick, self.useKatzgraber)
                moved = target.get_spins()
                if target is eng_icm:
                    if S > 0:
                        for r in range(R):
                            for j in range(K):
                                M[r * N:(r + 1) * N, j * S] = moved[r * K + j]
                else:
                    for r in range(R):
                        for j in range(K):
                            m_start[r * N:(r + 1) * N, j] = moved[r * K + j]
                            if S > 0:
                                M[r * N:(r + 1) * N, (j + 1) * S - 1] = moved[r * K + j]
                # --- PT swap per sub-replica on the LAST column (:248-285)
                selected = self.select_non_overlapping_pairs(all_pairs) if numpy_mode else \
                    self._select_pairs_host_rng(all_pairs, host_rng)
                for j in range(K):
                    for (sel, nxt) in selected:
                        m_sel = M[(sel - 1) * N:sel * N, (j + 1) * S - 1].copy()
                        m_next = M[(nxt - 1) * N:nxt * N, (j + 1) * S - 1].copy()
                        E = eng_e.energy_of(np.stack([m_sel, m_next]).astype(np.int8))
                        E_sel, E_next = E[0], E[1]
                        print(f"\nSelected pair indices: {sel}, {nxt}")
                        log_pairs.append((sel, nxt))
                        u = np.random.rand() if numpy_mode else host_rng.random()
                        ok = u < min(1, np.exp((beta_list[nxt - 1] - beta_list[sel - 1]) * (E_next - E_sel)))
                        log_acc.append(int(ok))
                        if ok:
                            count[ii] += 1
                            print(f"swapping {np.sum(count)}th time")
                            m_start[(sel - 1) * N:sel * N, j] = m_next
                            m_start[(nxt - 1) * N:nxt * N, j] = m_sel
            Energy = np.zeros(R)
            for r in range(R):
                Energy[r] = self.replica_energy(M[r * N:(r + 1) * N, :], self.num_sweeps_read_per_swap)[0]
        finally:
            eng.close()
            eng_icm.close()
        self.swap_pairs = np.array(log_pairs, dtype=np.int32).reshape(-1, 2)
        self.swap_accepted = np.array(log_acc, dtype=np.int8)
        print(f"\nLatest energy from each replica = {Energy}")
        if plot:
            self.plot_energies([self.replica_energy(M[r * N:(r + 1) * N, :], self.num_sweeps_read_per_swap)[1]
                                for r in range(R)], beta_list)
        return M, Energy

    def _run_device_resident(self, inst, beta_list, plot, return_trace="float64"):
        """Throughput path (rng="philox", icm_feedback=True): every (sub-replica, replica) chain lives in ONE context,
        laid out sub-replica-major so that each sub-replica's beta ladder is a block of consecutive chains.  Per round:
        batched sweeps at the ladder temperatures -> iso-cluster moves between randomly paired sub-replicas of every
        replica, acting on the CURRENT states (k_icm_components / k_icm_move, picks drawn on the device) -> label-
        exchange swaps decided on the device for every sub-replica ladder.  Configurations never leave HBM until the
        last round's trace is read out."""
        R, K, N = self.num_replicas, self.num_subreplicas, inst.n
        S, rounds = self.num_sweeps_MCMC_per_swap, int(self.num_swap_attempts)
        G = K * R                                       # chain id = j * R + slot-holder index
        host_rng = np.random.default_rng(self.seed)      # initial states only
        eng = Engine(inst, None, G, device=self._cache.device)
        try:
            eng.set_spins((2 * host_rng.integers(0, 2, size=(G, N), dtype=np.int8) - 1).astype(np.int8))
            eng.pt_init(beta_list)
            planner = RoundPlanner(eng, self._sweep_counter, rounds, S, self.seed)
            if self.num_swapping_pairs > 0 and rounds > 0:
                eng.pt_plan(0, rounds, self.seed, self.num_swapping_pairs)
            last, E_rec, slots_last = None, None, np.arange(G, dtype=np.int32) % R
            icm_sizes = []
            log_every = max(1, rounds // 16)
            swaps = self.num_swapping_pairs > 0 and rounds > 0
            if swaps:
                eng.pt_log_begin(0, rounds, self.num_swapping_pairs)      # swap log stays on the device, read once
            for ii in range(rounds):
                is_last = ii == rounds - 1
                if is_last:
                    slots_last = eng.pt_slots()
                o = planner.sweep(ii, record_stride=1 if is_last else 0)
                if is_last and S > 0:
                    last = o["spins"]                                  # [G, S, N] int8
                    E_rec = eng.energy_of_recorded(S)                  # fp64 energies of the trace, from its device copy
                # Houdayer: for every temperature slot the K sub-replicas that currently hold it are shuffled and paired
                # on the device (nlmc_icm_round_ladders); cluster sizes are read back on a sample of the rounds only (a
                # read-back synchronises the stream; it also reports a component search that did not converge)
                logged = rounds <= 16 or is_last or ii % log_every == 0
                info = eng.icm_round_ladders(ii, self.seed, self.useKatzgraber, want_info=logged)
                if logged:
                    icm_sizes.append(info[:, 1].copy())
                if swaps:
                    eng.pt_swap_philox(ii, self.seed, self.num_swapping_pairs, want_log=False)
            acc_log = [eng.pt_log_read()[1]] if swaps else []
            self._sweep_counter += rounds * S
            # M block of replica r = [N, K S]: sub-replica j in columns j S .. (NPT/apt_ICM.py:188,207); Energy[r] = min
            # over the FIRST num_sweeps_read_per_swap columns of the block (NPT/apt_ICM.py:36-50, :290-297)
            dt = np.float64 if return_trace != "int8" else np.int8
            M = None if return_trace is None else np.zeros((N * R, S * K), dtype=dt)
            Energy = np.zeros(R)
            if last is not None:
                cols = (np.arange(G, dtype=np.int32) // R) * S
                if return_trace is not None:
                    M = trace_layout(last, slots_last, R, dt, dst_col=cols, row_len=S * K)
                E_blk = np.empty((R, K * S))
                E_blk[slots_last[:, None], cols[:, None] + np.arange(S)[None, :]] = E_rec
                k = self.num_sweeps_read_per_swap
                Energy = np.min(E_blk[:, :k], axis=1) if k > 0 else np.min(np.zeros(0))    # np.min of nothing: ValueError
            self.final_energies = eng.energy()
            self.final_slots = eng.pt_slots()
            self.swap_accepted = np.concatenate([a.reshape(-1) for a in acc_log]).astype(np.int8) if acc_log else np.zeros(0, np.int8)
            self.icm_cluster_sizes = np.concatenate(icm_sizes) if icm_sizes else np.zeros(0, np.int32)
        finally:
            eng.close()
        print(f"\nLatest energy from each replica = {Energy}")
        if plot and M is not None:
            self.plot_energies([self.replica_energy(M[r * N:(r + 1) * N, :], self.num_sweeps_read_per_swap)[1]
                                for r in range(R)], beta_list)
        return M, Energy

    def _run_slot_sharded(self, inst, beta_list, plot, return_trace, device_ids):
        """The device-resident run with the ladder cut into slot blocks over `device_ids` (NPT/apt_ICM.py:215-285):
        per round sweeps -> Houdayer moves inside every block -> one exchange of tracked energies and boundary configurations between
        the blocks -> identical swap decision everywhere (distributed.SlotShardedAPT)."""
        from .distributed import SlotShardedAPT
        R, K, N = self.num_replicas, self.num_subreplicas, inst.n
        S, rounds = self.num_sweeps_MCMC_per_swap, int(self.num_swap_attempts)
        host_rng = np.random.default_rng(self.seed)      # initial states only
        own = len(device_ids) > 1

        def mk(i, n, b, g, dev=0):
            return Engine(i, None, n, device=int(dev), chain_base=b, n_chains_global=g, own_stream=own)
        apt = SlotShardedAPT(mk, inst, beta_list, K, self.seed, self.num_swapping_pairs, precision="f32",
                             katzgraber=self.useKatzgraber, device_ids=device_ids)
        try:
            apt.sweeps_done = self._sweep_counter
            apt.set_spins_by_slot((2 * host_rng.integers(0, 2, size=(K, R, N), dtype=np.int8) - 1).astype(np.int8))
            apt.plan(rounds, S)
            acc_log, icm_sizes = [], []
            log_every = max(1, rounds // 16)
            for ii in range(rounds):
                is_last = ii == rounds - 1
                logged = rounds <= 16 or is_last or ii % log_every == 0
                log, info = apt.round(S, want_log=self.num_swapping_pairs > 0, want_info=logged,
                                      **(dict(record_stride=1) if (is_last and S > 0) else {}))
                if log is not None and log[1] is not None:
                    acc_log.append(log[1])
                if logged:
                    icm_sizes.append(np.concatenate(info)[:, 1].copy())
            self._sweep_counter += rounds * S
            dt = np.float64 if return_trace != "int8" else np.int8
            M = None if return_trace is None else np.zeros((N * R, S * K), dtype=dt)
            Energy = np.zeros(R)
            if rounds > 0 and S > 0:
                Rw = apt.Rw
                E_blk = np.empty((R, K * S))
                for w, (e, o, sl) in enumerate(zip(apt.engs, apt.last_outputs, apt.last_slots)):
                    c = np.arange(K * Rw, dtype=np.int32)
                    dst, cols = (w * Rw + sl).astype(np.int32), ((c // Rw) * S).astype(np.int32)
                    if M is not None:
                        trace_layout(o["spins"], dst, R, dt, dst_col=cols, row_len=S * K, out=M)
                    E_blk[dst[:, None], cols[:, None] + np.arange(S)[None, :]] = e.energy_of(o["spins"].reshape(-1, N)).reshape(K * Rw, S)
                k = self.num_sweeps_read_per_swap
                Energy = np.min(E_blk[:, :k], axis=1) if k > 0 else np.min(np.zeros(0))
            self.final_states, self.final_energies = apt.gather_by_slot()
            self.swap_accepted = np.concatenate([a.reshape(-1) for a in acc_log]).astype(np.int8) if acc_log else np.zeros(0, np.int8)
            self.icm_cluster_sizes = np.concatenate(icm_sizes) if icm_sizes else np.zeros(0, np.int32)
            apt.check()
        finally:
            apt.close()
        print(f"\nLatest energy from each replica = {Energy}")
        if plot and M is not None:
            self.plot_energies([self.replica_energy(M[r * N:(r + 1) * N, :], self.num_sweeps_read_per_swap)[1]
                                for r in range(R)], beta_list)
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

    def plot_energies(self, EE1_list, beta_list):
        """NPT/apt_ICM.py:307-322 (presentation only)."""
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        plt.figure()
        for i in range(self.num_replicas):
            plt.plot(EE1_list[i], label=f"Replica {i + 1} (β={beta_list[i]:.2f})")
        plt.xlabel('Sweeps')
        plt.ylabel('Energy')
        plt.legend()
        plt.savefig('APT_ICM_energy.png')
        plt.close()
