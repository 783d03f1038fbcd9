"""Host-side handle on the HIP engine (one context = one instance (J,h) x a batch of chains on one GPU).

Everything that touches spins runs in the HIP kernels behind the C-ABI (include/nlmc.h).  This module only
marshals NumPy arrays; it contains no CPU implementation of the sweep and never imports `oracle`.
"""
import ctypes

import numpy as np
import scipy.sparse as sp

from . import _abi


class Instance:
    """(J, h) as the engine wants it: CSR, both triangles, sorted columns, explicit zeros dropped -- the matrix
    `scipy.sparse.csr_matrix(J)` the reference builds at NMC/nmc.py:53 -- without ever densifying a sparse input."""

    def __init__(self, J, h):
        A = sp.csr_matrix(J).astype(np.float64).copy()
        if A.shape[0] != A.shape[1]:
            raise ValueError("J must be square")
        A.eliminate_zeros()
        A.sort_indices()
        self.n = int(A.shape[0])
        self.indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.data = np.ascontiguousarray(A.data, dtype=np.float64)
        self.h = np.ascontiguousarray(np.asarray(h, dtype=np.float64).reshape(-1))
        if self.h.shape[0] != self.n:
            raise ValueError("h must have one entry per spin")
        self.csr = A
        d = A - A.T
        self.symmetric = (d.nnz == 0) or (np.max(np.abs(d.data)) <= 1e-12 * max(1.0, float(np.max(np.abs(A.data)))
                                                                                  if A.nnz else 1.0))

    @property
    def nnz(self):
        return int(self.data.shape[0])


def _beta_table(beta, R, S):
    """beta: scalar | [R] | [S] (anneal schedule, only if R != S or flagged by shape) | [R,S] -> (array, cs, ss)."""
    b = np.asarray(beta, dtype=np.float64)
    if b.ndim == 0:
        return np.ascontiguousarray(b.reshape(1)), 0, 0
    if b.ndim == 2:
        if b.shape != (R, S):
            raise ValueError("beta table must be [n_chains, n_sweeps]")
        if S > 1 and (b == b[:, :1]).all():         # one temperature per chain: the kernels' plain variants (stride 0 over sweeps)
            return np.ascontiguousarray(b[:, 0]), 1, 0
        return np.ascontiguousarray(b), S, 1
    raise ValueError("pass a scalar or a 2-D table; use per_chain()/per_sweep() helpers for 1-D inputs")


class Engine:
    def __init__(self, J, h, n_chains, device=0, stream=None, chain_base=0, n_chains_global=None, own_stream=False):
        self.inst = J if isinstance(J, Instance) else Instance(J, h)
        if not self.inst.symmetric:
            raise ValueError("J must be symmetric: the heat-bath field and the incremental energy assume J == J^T")
        self.n = self.inst.n
        self.n_chains = int(n_chains)
        self.chain_base = int(chain_base)
        self.n_chains_global = int(n_chains_global if n_chains_global is not None else n_chains)
        self._L = _abi.lib()
        h_ = ctypes.c_void_p()
        rc = self._L.nlmc_create(ctypes.byref(h_), int(device), ctypes.c_void_p(stream) if stream else None, self.n,
                                 self.inst.nnz, _abi.ptr(self.inst.indptr), _abi.ptr(self.inst.indices),
                                 _abi.ptr(self.inst.data), _abi.ptr(self.inst.h), self.n_chains, self.chain_base,
                                 self.n_chains_global)
        _abi.check(rc, None)
        self._ctx = h_
        if own_stream:                # several contexts on one device: a stream each, or they execute one after the other
            self._ck(self._L.nlmc_own_stream(self._ctx))
        self.ladder_len = 0
        self._ahead = None
        self.energy_scale = int(self._L.nlmc_energy_scale(self._ctx))
        self.field_scale = int(self._L.nlmc_field_scale(self._ctx))      # qs of the "f32" path: Jq = rint(J 2^qs)

    # -- lifetime ---------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.nlmc_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def _ck(self, rc):
        _abi.check(rc, self._ctx)

    # -- state ------------------------------------------------------------------------------------------
    def set_spins(self, spins):
        s = _abi.as_c(spins, np.int8).reshape(self.n_chains, self.n)
        self._ck(self._L.nlmc_set_spins(self._ctx, _abi.ptr(s)))

    def get_spins(self):
        s = np.empty((self.n_chains, self.n), dtype=np.int8)
        self._ck(self._L.nlmc_get_spins(self._ctx, _abi.ptr(s)))
        return s

    def set_flags(self, flags, temp_x=1.0):
        if flags is None:
            self._ck(self._L.nlmc_set_flags(self._ctx, None, 1.0))
            return
        f = _abi.as_c(flags, np.uint8).reshape(self.n_chains, self.n)
        self._ck(self._L.nlmc_set_flags(self._ctx, _abi.ptr(f), float(temp_x)))

    def energy(self):
        out = np.empty(self.n_chains, dtype=np.float64)
        self._ck(self._L.nlmc_energy(self._ctx, _abi.ptr(out)))
        return out

    def energy_tracked(self):
        """The incrementally tracked energies (no recomputation)."""
        out = np.empty(self.n_chains, dtype=np.float64)
        self._ck(self._L.nlmc_energy_tracked(self._ctx, _abi.ptr(out)))
        return out

    def set_energy_sink(self, dev_ptr):
        """Later sweep calls also write the tracked energies of their final states to this device buffer (None: off)."""
        self._ck(self._L.nlmc_set_energy_sink(self._ctx, ctypes.c_void_p(int(dev_ptr)) if dev_ptr else None))

    def energy_dev(self, dev_ptr):
        self._ck(self._L.nlmc_energy_dev(self._ctx, ctypes.c_void_p(int(dev_ptr))))

    def energy_of(self, configs):
        c = _abi.as_c(configs, np.int8).reshape(-1, self.n)
        out = np.empty(c.shape[0], dtype=np.float64)
        self._ck(self._L.nlmc_energy_of(self._ctx, _abi.ptr(c), c.shape[0], _abi.ptr(out)))
        return out

    def energy_of_recorded(self, count, first=0):
        """fp64 energies of recorded configurations first .. first+count-1 of every chain of the most recent sweep call
        (`record_stride` > 0), taken from the device copy of its trace -> [rows of that call, count]."""
        out = np.empty((getattr(self, "_last_rows", self.n_chains), int(count)), dtype=np.float64)
        self._ck(self._L.nlmc_energy_of_recorded(self._ctx, int(first), int(count), _abi.ptr(out)))
        return out

    # -- sweeps -----------------------------------------------------------------------------------------
    def rows(self):
        """Chains a sweep call acts on: all of them, or the selected subset (select())."""
        return int(self._L.nlmc_subset_count(self._ctx))

    def _outputs(self, S, record_stride, want_energy, want_min, want_state):
        R, n = self.rows(), self.n
        self._last_rows = R
        o = {}
        o["spins"] = np.empty((R, (S + record_stride - 1) // record_stride, n), np.int8) if record_stride else None
        o["energy"] = np.empty((R, S), np.float64) if want_energy else None
        o["min_energy"] = np.empty(R, np.float64) if want_min else None
        o["argmin"] = np.empty(R, np.int32) if want_min else None
        o["argmin_state"] = np.empty((R, n), np.int8) if want_state else None
        return o

    STREAM_CHUNK_BYTES = 384 << 20      # staged permutation + uniforms + schedule per C-ABI call

    def sweep_stream(self, perm, u, beta, record_stride=0, want_energy=False, want_min=False, want_state=False):
        """Reference-stream sweeps (NMC/nmc.py:28-91).  perm,u: [R,S,N]; beta: scalar or [R,S] table.
        Long runs are cut into several C-ABI calls so that the staged stream stays bounded; the pieces are stitched
        back together (trace, energies, first-argmin bookkeeping) exactly as one call would return them."""
        R, n = self.n_chains, self.n
        perm = np.asarray(perm).reshape(R, -1, n)
        S = perm.shape[1]
        u = np.asarray(u).reshape(R, S, n)
        per_sweep = R * n * 36                         # 12 B of stream + ~24 B of per-chain schedule per spin
        chunk = max(1, self.STREAM_CHUNK_BYTES // max(1, per_sweep))
        if record_stride > 1:
            chunk = max(record_stride, chunk // record_stride * record_stride)
        if S > chunk:
            b = np.asarray(beta, dtype=np.float64)
            outs = []
            for t0 in range(0, S, chunk):
                t1 = min(S, t0 + chunk)
                outs.append(self.sweep_stream(perm[:, t0:t1], u[:, t0:t1], b if b.ndim == 0 else b[:, t0:t1], record_stride,
                                              want_energy, want_min, want_state))
            return _stitch_outputs(outs, chunk, record_stride, want_energy, want_min, want_state)
        perm = _abi.as_c(perm, np.int32)
        u = _abi.as_c(u, np.float64)
        tab, cs, ss = _beta_table(beta, R, S)
        o = self._outputs(S, record_stride, want_energy, want_min, want_state)
        self._ck(self._L.nlmc_sweep_stream(self._ctx, S, _abi.ptr(perm), _abi.ptr(u), _abi.ptr(tab), cs, ss,
                                           max(1, record_stride), _abi.ptr(o["spins"]), _abi.ptr(o["energy"]),
                                           _abi.ptr(o["min_energy"]), _abi.ptr(o["argmin"]),
                                           _abi.ptr(o["argmin_state"])))
        return o

    def sweep_philox(self, n_sweeps, seed, sweep0=0, beta=None, precision="f32", order="shared", record_stride=0,
                     want_energy=False, want_min=False, want_state=False):
        """Throughput sweeps (device Philox).  beta: scalar | [R,S] table | None (= PT ladder)."""
        R = self.rows()
        S = int(n_sweeps)
        if beta is None:
            tabp, cs, ss = None, 0, 0
        else:
            tab, cs, ss = _beta_table(beta, R, S)
            tabp = _abi.ptr(tab)
        o = self._outputs(S, record_stride, want_energy, want_min, want_state)
        prec = {"f32": _abi.F32, "f64": _abi.F64}[precision]
        om = {"shared": _abi.ORDER_SHARED, "per_chain": _abi.ORDER_PER_CHAIN}[order]
        self._ck(self._L.nlmc_sweep_philox(self._ctx, prec, om, S, int(sweep0) & 0xFFFFFFFF, int(seed), tabp, cs, ss,
                                           max(1, record_stride), _abi.ptr(o["spins"]), _abi.ptr(o["energy"]),
                                           _abi.ptr(o["min_energy"]), _abi.ptr(o["argmin"]),
                                           _abi.ptr(o["argmin_state"])))
        return o

    FUSED_PLAN_BUDGET = 1 << 30        # bytes of fused-window plan one sweep_philox_windows piece may hold

    def fused_plan_bytes(self, window):
        return fused_plan_bytes(self.n, int(np.count_nonzero(np.diff(self.inst.indptr) > 8)), window)

    def sweep_philox_windows(self, n_sweeps, seed, sweep0=0, beta=None, window=None, budget_bytes=None, record_stride=0,
                             want_energy=False, want_min=False, want_state=False, want_recorded_energy=False):
        """sweep_philox on fused windows planned piece by piece against a memory budget (one planning call for a
        whole run of 10^4 sweeps takes 23 GB at N = 10^3, 230 GB at N = 10^4): the call is cut into pieces of at most
        budget_bytes / fused_plan_bytes windows, each planned, swept and dropped; pieces whose plan is refused (instance or
        window outside what the fused kernels take) run sweep by sweep -- same bits.  Outputs are stitched together exactly
        as one call would return them (trace, energies, first argmin).  want_recorded_energy: o["energy_recorded"] = fp64
        energies of the recorded configurations [rows, n_recorded], computed on the device copy of each piece's trace."""
        S = int(n_sweeps)
        T = fused_window(S) if window is None else int(window)
        budget = self.FUSED_PLAN_BUDGET if budget_bytes is None else int(budget_bytes)
        rec_e = bool(want_recorded_energy and record_stride)
        a = getattr(self, "_ahead", None)
        if (a is not None and window is None and a.window and S == a.S and int(seed) == a.seed and (int(sweep0) - a.sweep0) % max(1, S) == 0
                and 0 <= (int(sweep0) - a.sweep0) // max(1, S) < a.R
                and sweeps_per_plan_piece(self.fused_plan_bytes(a.window), S, a.window, budget, record_stride) == S):
            ii = (int(sweep0) - a.sweep0) // S          # launch ii of the run announced with plan_ahead: its windows are (or get)
            if not (a._fused_from <= ii < a._fused_to):  # planned together with those of the launches that follow
                a._plan(ii, True)
            if a._fused_from <= ii < a._fused_to:
                self.fused_last_call = True
                o = self.sweep_philox(S, seed, sweep0=sweep0, beta=beta, record_stride=record_stride, want_energy=want_energy,
                                      want_min=want_min, want_state=want_state)
                if rec_e:
                    o["energy_recorded"] = self.energy_of_recorded(o["spins"].shape[1])
                return o
        if not T or S % T or S == 0:
            self.fused_last_call = False
            o = self.sweep_philox(S, seed, sweep0=sweep0, beta=beta, record_stride=record_stride, want_energy=want_energy,
                                  want_min=want_min, want_state=want_state)
            if rec_e and S > 0:
                o["energy_recorded"] = self.energy_of_recorded(o["spins"].shape[1])
            return o
        per_piece = sweeps_per_plan_piece(self.fused_plan_bytes(T), S, T, budget, record_stride)
        b = None if beta is None else np.asarray(beta, dtype=np.float64)
        outs, fused = [], True
        if a is not None:                    # this call plans into the slot the announced run's windows live in: they are gone
            a._fused_from = a._fused_to = 0
            self.plan_slot(a.slot)
        for t0 in range(0, S, per_piece):
            t1 = min(S, t0 + per_piece)
            if fused:
                fused = self.plan_philox_fused(sweep0 + t0, (t1 - t0) // T, T, seed) == (t1 - t0) // T     # refused once: stop asking
            outs.append(self.sweep_philox(t1 - t0, seed, sweep0=sweep0 + t0, beta=b if (b is None or b.ndim < 2) else b[:, t0:t1],
                                          record_stride=record_stride, want_energy=want_energy, want_min=want_min,
                                          want_state=want_state))
            if rec_e:
                outs[-1]["energy_recorded"] = self.energy_of_recorded(outs[-1]["spins"].shape[1])
        self.fused_last_call = fused
        o = outs[0] if len(outs) == 1 else _stitch_outputs(outs, per_piece, record_stride, want_energy, want_min, want_state)
        if rec_e and len(outs) > 1:
            o["energy_recorded"] = np.concatenate([p["energy_recorded"] for p in outs], axis=1)
        return o

    def plan_ahead(self, sweep0, n_launches, sweeps_per_launch, seed, budget_bytes=None):
        """Announce `n_launches` sweep_philox_windows calls of `sweeps_per_launch` sweeps each at consecutive sweep indices from
        `sweep0` (the phases of an NMC run): their fused windows depend on the RNG only and are planned a budget's worth of
        LAUNCHES at a time instead of one launch at a time -- a planning launch is latency-bound (~1.5 ms whether it builds 9
        windows or 200; per phase launch that was 27 % of C2 through NMC.run_restarts).  plan_ahead(None) ends it."""
        if sweep0 is None or n_launches <= 0 or sweeps_per_launch <= 0:
            self._ahead = None
            return
        self._ahead = RoundPlanner(self, sweep0, n_launches, sweeps_per_launch, seed,
                                   budget_bytes=self.FUSED_PLAN_BUDGET if budget_bytes is None else budget_bytes, slot=0,
                                   fused_outputs=True)

    def plan_philox(self, sweep0, n_sweeps, seed, precision="f32"):
        prec = {"f32": _abi.F32, "f64": _abi.F64}[precision]
        self._ck(self._L.nlmc_plan_philox(self._ctx, prec, _abi.ORDER_SHARED, int(sweep0), int(n_sweeps), int(seed)))

    def plan_philox_fused(self, sweep0, n_windows, window, seed):
        """Fused-window schedules for `n_windows` launches of exactly `window` sweeps (include/nlmc.h).  Returns the
        number of windows planned (0: the instance does not qualify; calls then take the sweep-by-sweep path)."""
        k = ctypes.c_int32(0)
        self._ck(self._L.nlmc_plan_philox_fused(self._ctx, int(sweep0) & 0xFFFFFFFF, int(n_windows), int(window), int(seed),
                                                ctypes.byref(k)))
        return int(k.value)

    def fused_modes(self, window):
        """Precisions that may run on fused windows of `window` sweeps here: subset of {"f32", "f64"} (nlmc_fused_modes)."""
        m = int(self._L.nlmc_fused_modes(self._ctx, int(window)))
        return {p for bit, p in ((1, "f32"), (2, "f64")) if m & bit}

    def plan_slot(self, slot):
        """Fused-window plans live in two slots (include/nlmc.h: nlmc_plan_slot); planning calls write to the selected one."""
        self._ck(self._L.nlmc_plan_slot(self._ctx, int(slot)))

    def plan_levels(self, window):
        """Chunks (64 schedule positions) per level of a planned fused window of the selected plan slot (diagnostic)."""
        off = np.zeros(1025, dtype=np.int32)
        nl = ctypes.c_int32(0)
        self._ck(self._L.nlmc_plan_get_levels(self._ctx, int(window), _abi.ptr(off), 1025, ctypes.byref(nl)))
        return np.diff(off[:nl.value + 1])

    def plan_reserve_fused(self, n_windows, window):
        """Allocate the fused-window plan buffers for up to n_windows windows (no schedule is built)."""
        self._ck(self._L.nlmc_plan_reserve_fused(self._ctx, int(n_windows), int(window)))

    # -- replica exchange -------------------------------------------------------------------------------
    def pt_init(self, beta_list):
        b = _abi.as_c(beta_list, np.float64).reshape(-1)
        self._ck(self._L.nlmc_pt_init(self._ctx, b.shape[0], _abi.ptr(b)))
        self.ladder_len = int(b.shape[0])

    def pt_slots(self):
        s = np.empty(self.n_chains_global, dtype=np.int32)
        self._ck(self._L.nlmc_pt_get_slots(self._ctx, _abi.ptr(s)))
        return s

    def pt_set_slots(self, slots):
        s = _abi.as_c(slots, np.int32).reshape(self.n_chains_global)
        self._ck(self._L.nlmc_pt_set_slots(self._ctx, _abi.ptr(s)))

    def pt_apply_swap(self, ladder, slot_a, slot_b):
        self._ck(self._L.nlmc_pt_apply_swap(self._ctx, int(ladder), int(slot_a), int(slot_b)))

    def pt_swap_philox(self, round_idx, seed, n_pairs, energies_all_dev=None, want_log=True):
        nl = self.n_chains_global // self.ladder_len
        pairs = np.empty((nl, n_pairs, 2), np.int32) if want_log else None
        acc = np.empty((nl, n_pairs), np.uint8) if want_log else None
        dev = ctypes.c_void_p(int(energies_all_dev)) if energies_all_dev else None
        self._ck(self._L.nlmc_pt_swap_philox(self._ctx, int(round_idx), int(seed), int(n_pairs), dev, _abi.ptr(pairs),
                                             _abi.ptr(acc)))
        return pairs, acc

    def pt_swap_philox_host(self, round_idx, seed, n_pairs, energies_all, want_log=False):
        """Swap round of a sharded context with the all-gathered energies in host memory."""
        nl = self.n_chains_global // self.ladder_len
        pairs = np.empty((nl, n_pairs, 2), np.int32) if want_log else None
        acc = np.empty((nl, n_pairs), np.uint8) if want_log else None
        e = _abi.as_c(energies_all, np.float64).reshape(self.n_chains_global)
        self._ck(self._L.nlmc_pt_swap_philox_host(self._ctx, int(round_idx), int(seed), int(n_pairs), _abi.ptr(e),
                                                  _abi.ptr(pairs), _abi.ptr(acc)))
        return pairs, acc

    # -- sharded ladders: the per-round all-gather issued by the library (include/nlmc.h: nlmc_comm_init) -------------
    @staticmethod
    def comm_unique_id():
        out = np.zeros(128, dtype=np.uint8)
        _abi.check(_abi.lib().nlmc_comm_unique_id(_abi.ptr(out)), None)
        return out

    def comm_init(self, unique_id, world, rank):
        uid = _abi.as_c(unique_id, np.uint8).reshape(128)
        self._ck(self._L.nlmc_comm_init(self._ctx, _abi.ptr(uid), int(world), int(rank)))

    def pt_swap_philox_collective(self, round_idx, seed, n_pairs, refresh_energies=False, want_log=False):
        nl = self.n_chains_global // self.ladder_len
        pairs = np.empty((nl, n_pairs, 2), np.int32) if want_log else None
        acc = np.empty((nl, n_pairs), np.uint8) if want_log else None
        self._ck(self._L.nlmc_pt_swap_philox_collective(self._ctx, int(round_idx), int(seed), int(n_pairs), int(bool(refresh_energies)),
                                                        _abi.ptr(pairs), _abi.ptr(acc)))
        return pairs, acc

    @staticmethod
    def comm_probe():
        """True when this process can load librccl (nlmc_comm_probe); every rank asks before comm_init and the ranks agree."""
        return _abi.lib().nlmc_comm_probe() == _abi.OK

    def comm_check(self, timeout_ms=-1):
        """RCCL asynchronous error flag, and with timeout_ms >= 0 a bounded wait for the collectives queued on the context's stream;
        RuntimeError after aborting the communicator (include/nlmc.h: nlmc_comm_check)."""
        self._ck(self._L.nlmc_comm_check(self._ctx, int(timeout_ms)))

    # -- APT run cut into temperature-slot blocks over ranks (include/nlmc.h: nlmc_apt_shard) -------------------------------
    def apt_shard(self, beta_global, world, rank):
        b = _abi.as_c(beta_global, np.float64).reshape(-1)
        self._ck(self._L.nlmc_apt_shard(self._ctx, b.shape[0], int(world), int(rank), _abi.ptr(b)))
        self.apt_world, self.apt_rank, self.apt_R = int(world), int(rank), int(b.shape[0])

    def apt_pack(self, want_configs=True):
        """(tracked energies [K, Rw] int64 in units of 2^-energy_scale by (ladder, local slot), configurations on local slot 0
        [K, n], on local slot Rw - 1 [K, n])."""
        K = self.n_chains // self.ladder_len
        e = np.empty((K, self.ladder_len), np.int64)
        lo = np.empty((K, self.n), np.int8) if want_configs else None
        hi = np.empty((K, self.n), np.int8) if want_configs else None
        self._ck(self._L.nlmc_apt_pack(self._ctx, _abi.ptr(e), _abi.ptr(lo), _abi.ptr(hi)))
        return e, lo, hi

    def _apt_log(self, n_pairs, want_log):
        K = self.n_chains // self.ladder_len
        return (np.empty((K, n_pairs, 2), np.int32), np.empty((K, n_pairs), np.uint8)) if want_log else (None, None)

    def apt_swap_host(self, round_idx, seed, n_pairs, efix_all, recv_lo, recv_hi, want_log=False):
        pairs, acc = self._apt_log(n_pairs, want_log)
        e = _abi.as_c(efix_all, np.int64).reshape(self.apt_world, -1, self.ladder_len)
        lo = None if recv_lo is None else _abi.as_c(recv_lo, np.int8).reshape(-1, self.n)
        hi = None if recv_hi is None else _abi.as_c(recv_hi, np.int8).reshape(-1, self.n)
        self._ck(self._L.nlmc_apt_swap_host(self._ctx, int(round_idx), int(seed), int(n_pairs), _abi.ptr(e), _abi.ptr(lo), _abi.ptr(hi),
                                            _abi.ptr(pairs), _abi.ptr(acc)))
        return pairs, acc

    def apt_selftest_exchange(self):
        """(recv_lo, recv_hi) [K, n] of the collective path's neighbour exchange run with this rank as its own neighbour."""
        K = self.n_chains // self.ladder_len
        lo, hi = np.empty((K, self.n), np.int8), np.empty((K, self.n), np.int8)
        self._ck(self._L.nlmc_apt_selftest_exchange(self._ctx, _abi.ptr(lo), _abi.ptr(hi)))
        return lo, hi

    def apt_swap_collective(self, round_idx, seed, n_pairs, want_log=False):
        pairs, acc = self._apt_log(n_pairs, want_log)
        self._ck(self._L.nlmc_apt_swap_collective(self._ctx, int(round_idx), int(seed), int(n_pairs), _abi.ptr(pairs), _abi.ptr(acc)))
        return pairs, acc

    def pt_plan(self, round0, n_rounds, seed, n_pairs):
        self._ck(self._L.nlmc_pt_plan(self._ctx, int(round0), int(n_rounds), int(seed), int(n_pairs)))

    def pt_rounds_fused(self, n_rounds, sweeps_per_round, seed, sweep0, round0, n_pairs, precision="f32"):
        """n_rounds whole rounds (sweeps + swap round) in one cooperative launch (include/nlmc.h: nlmc_pt_rounds_fused).  True when the
        rounds were queued, False when the context / plans do not qualify (nothing was run: drive the rounds one by one)."""
        prec = {"f32": _abi.F32, "f64": _abi.F64}[precision]
        rc = self._L.nlmc_pt_rounds_fused(self._ctx, prec, int(n_rounds), int(sweeps_per_round), int(sweep0) & 0xFFFFFFFF,
                                          int(round0) & 0xFFFFFFFF, int(seed), int(n_pairs))
        if rc == _abi.ERR_UNSUPPORTED:
            self.rounds_fused_refusal = _abi.lib().nlmc_last_error(self._ctx).decode()
            return False
        self._ck(rc)
        return True

    def pt_rounds_deferred(self, n_rounds, sweeps_per_round, seed, sweep0, round0, n_pairs, precision="f32"):
        """n_rounds rounds as n_rounds sweep launches (each decides the previous round's swap in its prologue) + one swap launch
        (include/nlmc.h: nlmc_pt_rounds_deferred).  True when queued, False when the context / plans do not qualify."""
        prec = {"f32": _abi.F32, "f64": _abi.F64}[precision]
        rc = self._L.nlmc_pt_rounds_deferred(self._ctx, prec, int(n_rounds), int(sweeps_per_round), int(sweep0) & 0xFFFFFFFF,
                                             int(round0) & 0xFFFFFFFF, int(seed), int(n_pairs))
        if rc == _abi.ERR_UNSUPPORTED:
            self.rounds_fused_refusal = _abi.lib().nlmc_last_error(self._ctx).decode()
            return False
        self._ck(rc)
        return True

    def pt_log_begin(self, round0, n_rounds, n_pairs):
        """Keep the swap log of the next rounds on the device (rounds called with want_log=False)."""
        self._ck(self._L.nlmc_pt_log_begin(self._ctx, int(round0), int(n_rounds), int(n_pairs)))
        self._pt_log_shape = (int(n_rounds), self.n_chains_global // self.ladder_len, int(n_pairs))

    def pt_log_read(self):
        r, nl, p = self._pt_log_shape
        pairs, acc = np.empty((r, nl, p, 2), np.int32), np.empty((r, nl, p), np.uint8)
        if r * nl * p:
            self._ck(self._L.nlmc_pt_log_read(self._ctx, _abi.ptr(pairs), _abi.ptr(acc)))
        return pairs, acc

    def pt_check(self):
        """ValueError("Cannot find non-overlapping pairs.") if a device-decided swap round ran out of pairs (NPT/npt.py:526)."""
        self._ck(self._L.nlmc_pt_check(self._ctx))

    # -- rounds whose marked temperature slots run NMC cycles (include/nlmc.h) -----------------------------
    def mark_slots(self, marks):
        m = None if marks is None else _abi.as_c(np.asarray(marks).astype(bool), np.uint8).reshape(self.ladder_len)
        self._ck(self._L.nlmc_pt_mark_slots(self._ctx, _abi.ptr(m)))

    def select(self, which):
        """Later sweep_philox / adopt_best / backbone_clusters / set_phase calls act on: "all" chains, or the local chains
        currently on "unmarked" / "marked" temperature slots."""
        w = {"all": _abi.CHAINS_ALL, "unmarked": _abi.CHAINS_UNMARKED, "marked": _abi.CHAINS_MARKED}[which]
        self._ck(self._L.nlmc_select_chains(self._ctx, w))

    def overlap_subsets(self, on=True):
        """Queue the marked subset's work on a second stream beside the unmarked chains' sweeps (nlmc_overlap_subsets)."""
        self._ck(self._L.nlmc_overlap_subsets(self._ctx, int(bool(on))))

    def subset(self):
        out = np.empty(self.rows(), dtype=np.int32)
        self._ck(self._L.nlmc_get_subset(self._ctx, _abi.ptr(out)))
        return out

    def track_minimum(self, on=True, stride=1):
        """Sweep calls keep the running minimum + argmin state on the device; `stride` > 1: over sweeps 0, stride, 2 stride, ... only."""
        self._ck(self._L.nlmc_track_minimum(self._ctx, (max(1, int(stride)) if on else 0)))

    def backbone_seed(self, snapshot=True):
        """snapshot=True: later backbone_clusters calls are seeded with the configurations as they are NOW; False: with the current ones."""
        self._ck(self._L.nlmc_backbone_seed(self._ctx, 1 if snapshot else 0))

    def adopt_best(self):
        self._ck(self._L.nlmc_adopt_best(self._ctx))

    def backbone_clusters(self, epsilon, lambdas, beta, tolerance, max_iterations, sat, thresholds):
        eps = _abi.as_c(epsilon, np.float64).reshape(self.n)
        lam = _abi.as_c(lambdas, np.float64).reshape(-1)
        thr = _abi.as_c(thresholds, np.float64).reshape(-1)
        self._ck(self._L.nlmc_backbone_clusters(self._ctx, _abi.ptr(eps), _abi.ptr(lam), lam.shape[0], float(beta), float(tolerance),
                                                int(max_iterations), float(sat), _abi.ptr(thr), thr.shape[0]))

    def backbone_check(self):
        self._ck(self._L.nlmc_backbone_check(self._ctx))

    def cluster_mask(self):
        out = np.empty((self.n_chains, self.n), dtype=np.uint8)
        self._ck(self._L.nlmc_get_cluster_mask(self._ctx, _abi.ptr(out)))
        return out

    def set_cluster_mask(self, mask):
        m = _abi.as_c(np.asarray(mask).astype(bool), np.uint8).reshape(self.n_chains, self.n)
        self._ck(self._L.nlmc_set_cluster_mask(self._ctx, _abi.ptr(m)))

    def set_phase(self, kind, temp_x=1.0):
        k = {"ALL": _abi.PHASE_ALL, "C": _abi.PHASE_BACKBONE_HOT, "NC": _abi.PHASE_BACKBONE_FROZEN}[kind]
        self._ck(self._L.nlmc_set_phase(self._ctx, k, float(temp_x)))

    # -- iso-cluster move -------------------------------------------------------------------------------
    def icm_components(self, chain_a, chain_b):
        out = ctypes.c_int32(0)
        self._ck(self._L.nlmc_icm_components(self._ctx, int(chain_a), int(chain_b), ctypes.byref(out)))
        return int(out.value)

    def icm_move(self, chain_a, chain_b, pick_index, katzgraber=True):
        info = np.zeros(2, np.int32)
        self._ck(self._L.nlmc_icm_move(self._ctx, int(chain_a), int(chain_b), int(pick_index), int(bool(katzgraber)),
                                       _abi.ptr(info)))
        return int(info[0]), int(info[1])

    def icm_labels(self):
        out = np.empty(self.n, dtype=np.int32)
        self._ck(self._L.nlmc_icm_get_labels(self._ctx, _abi.ptr(out)))
        return out

    def icm_round_philox(self, pairs, round_idx, seed, katzgraber=True, want_info=False):
        p = _abi.as_c(pairs, np.int32).reshape(-1, 2)
        info = np.zeros((p.shape[0], 2), np.int32) if want_info else None
        self._ck(self._L.nlmc_icm_round_philox(self._ctx, _abi.ptr(p), p.shape[0], int(round_idx), int(seed),
                                               int(bool(katzgraber)), _abi.ptr(info)))
        return info

    def icm_round_ladders(self, round_idx, seed, katzgraber=True, want_info=False):
        """Houdayer step of one APT round with the pairing decided on the device (include/nlmc.h)."""
        npairs = ctypes.c_int32(0)
        R = self.ladder_len
        K = self.n_chains_global // R
        info = np.zeros((R * (K // 2), 2), np.int32) if want_info else None
        self._ck(self._L.nlmc_icm_round_ladders(self._ctx, int(round_idx), int(seed), int(bool(katzgraber)),
                                                ctypes.byref(npairs), _abi.ptr(info)))
        return info

    # -- backbone inference (loopy BP on the device) -------------------------------------------------------
    def lbp_convexified(self, m_star, epsilon, lambdas, beta, tolerance, max_iterations, sat, want_all=False):
        """Batched lambda loop of LBP_convexified (include/nlmc.h: nlmc_lbp_convexified).  m_star [P, n] float64.
        Returns dict(mag [P, n], n_lambdas [P], iters [P, L], status [P], mag_all [P, L, n] or None)."""
        ms = _abi.as_c(np.atleast_2d(m_star), np.float64)
        P, L = ms.shape[0], len(lambdas)
        if ms.shape[1] != self.n:
            raise ValueError("m_star must be [n_problems, n]")
        eps = _abi.as_c(epsilon, np.float64).reshape(-1)
        lam = _abi.as_c(lambdas, np.float64).reshape(-1)
        mag = np.zeros((P, self.n))
        mag_all = np.zeros((P, L, self.n)) if want_all else None
        nl, it, st = np.zeros(P, np.int32), np.zeros((P, L), np.int32), np.zeros(P, np.int32)
        self._ck(self._L.nlmc_lbp_convexified(self._ctx, P, _abi.ptr(ms), _abi.ptr(eps), _abi.ptr(lam), L, float(beta),
                                              float(tolerance), int(max_iterations), float(sat), _abi.ptr(mag),
                                              _abi.ptr(mag_all), _abi.ptr(nl), _abi.ptr(it), _abi.ptr(st)))
        return {"mag": mag, "n_lambdas": nl, "iters": it, "status": st, "mag_all": mag_all}

    # -- measurement ------------------------------------------------------------------------------------
    def last_timing(self):
        a, b, c = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
        self._ck(self._L.nlmc_last_timing(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"ms_levelize": a.value, "ms_sweep": b.value, "launches_sweep": c.value}

    def timing_reset(self, enable=True, every=1):
        """Accumulate HIP-event timings of the sweep calls from now on (`every` > 1: events around every `every`-th
        fused-window launch only), or stop accumulating."""
        self._ck(self._L.nlmc_timing_reset(self._ctx, max(1, int(every)) if enable else 0))

    def timing_total(self):
        a, b, c, d = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self._L.nlmc_timing_total(self._ctx, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
        return {"ms_levelize": a.value, "ms_sweep": b.value, "launches_sweep": c.value, "launches_timed": d.value}

    def probe_level_round(self, waves=16, conflict_free=True, rounds=20000, n_workgroups=256):
        """ns per level-synchronous round (barrier + 8 LDS byte gathers + write) on this device (nlmc_probe_level_round)."""
        out = ctypes.c_double(0)
        self._ck(self._L.nlmc_probe_level_round(self._ctx, int(waves), int(bool(conflict_free)), int(rounds), int(n_workgroups),
                                                ctypes.byref(out)))
        return out.value

    def last_schedule_stats(self):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        self._ck(self._L.nlmc_last_schedule_stats(self._ctx, ctypes.byref(a), ctypes.byref(b)))
        return {"orders": a.value, "levels": b.value}


def trace_layout(spins, dst_block=None, n_dst_blocks=None, dtype=np.float64, n_threads=0, dst_col=None, row_len=None,
                 out=None):
    """Recorded traces [blocks, S, N] int8 -> the reference's M layout [n_dst_blocks * N, row_len] (`M[r*N:(r+1)*N, :] =
    MCMC(...)`, NPT/npt.py:641; NPT/apt_ICM.py:207 with a column group per sub-replica): block b goes to the rows of block
    dst_block[b], columns dst_col[b] .. dst_col[b]+S-1; int8 or float64; compiled host routine (nlmc_trace_layout), one
    thread per core.  `out`: write into this C-contiguous [n_dst_blocks * N, row_len] array instead of a new one."""
    spins = np.ascontiguousarray(spins, dtype=np.int8)
    B, S, N = spins.shape
    nd = B if n_dst_blocks is None else int(n_dst_blocks)
    rl = S if row_len is None else int(row_len)
    dt = np.dtype(dtype) if out is None else out.dtype
    if dt not in (np.dtype(np.int8), np.dtype(np.float64)):
        raise ValueError("trace_layout: dtype must be int8 or float64")
    db = None if dst_block is None else np.ascontiguousarray(dst_block, dtype=np.int32)
    dc = None if dst_col is None else np.ascontiguousarray(dst_col, dtype=np.int32)
    for x in (db, dc):
        if x is not None and x.shape != (B,):
            raise ValueError("trace_layout: dst_block / dst_col must have one entry per block")
    if out is not None:
        if out.shape != (nd * N, rl) or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]:
            raise ValueError("trace_layout: out must be a writeable C-contiguous [n_dst_blocks * N, row_len] array")
        M = out
    else:
        full = B * S == nd * rl                  # distinct destinations (checked by the routine) then cover all of M
        M = np.empty((nd * N, rl), dtype=dt) if full else np.zeros((nd * N, rl), dtype=dt)
    _abi.check(_abi.lib().nlmc_trace_layout(_abi.ptr(spins), B, S, N, _abi.ptr(db), _abi.ptr(dc), nd, rl, _abi.ptr(M),
                                            dt.itemsize, int(n_threads)), None)
    return M


def _stitch_outputs(outs, chunk, record_stride, want_energy, want_min, want_state):
    """Outputs of consecutive pieces of one logical sweep call (pieces of `chunk` sweeps) -> what one call would return."""
    R = outs[0]["spins"].shape[0] if record_stride else (outs[0]["energy"].shape[0] if want_energy else len(outs[0]["min_energy"]))
    o = {"spins": np.concatenate([p["spins"] for p in outs], axis=1) if record_stride else None,
         "energy": np.concatenate([p["energy"] for p in outs], axis=1) if want_energy else None,
         "min_energy": None, "argmin": None, "argmin_state": None}
    if want_min:
        mins = np.stack([p["min_energy"] for p in outs])           # [pieces, R]
        first = np.argmin(mins, axis=0)                            # np.argmin: first piece holding the minimum
        o["min_energy"] = mins[first, np.arange(R)]
        o["argmin"] = np.array([outs[first[r]]["argmin"][r] + first[r] * chunk for r in range(R)], dtype=np.int32)
        if want_state:
            o["argmin_state"] = np.stack([outs[first[r]]["argmin_state"][r] for r in range(R)])
    return o


def prefault_async(arr, n_threads=0):
    """Touch every page of a freshly allocated array from a background thread (nlmc_host_prefault; the call releases the GIL).
    Returns the thread: join() it before the array is filled."""
    import threading
    a = arr
    t = threading.Thread(target=lambda: _abi.lib().nlmc_host_prefault(_abi.ptr(a), int(a.nbytes), int(n_threads)), daemon=True)
    t.start()
    return t


def device_count():
    return int(_abi.lib().nlmc_device_count())


def fused_plan_bytes(n, n_long, window):
    """Upper estimate of the device memory one planned window of `window` sweeps takes (head 8 + row planes <= 64 + item ids
    4 bytes per schedule position, 2 bytes per update of level scratch): what nlmc_plan_philox_fused allocates per window
    (csrc/nlmc.hip: reserve_fused_plan, fused_pstride).  n_long: rows longer than 8 entries (two positions each)."""
    pstride = -(-int(window) * (int(n) + int(n_long)) // 64) * 64 + 64 * 1024
    return pstride * 76 + int(window) * int(n) * 2 + 8192


def sweeps_per_plan_piece(bytes_per_window, n_sweeps, window, budget_bytes, record_stride=0):
    """Sweeps one piece of Engine.sweep_philox_windows covers: as many whole windows as fit the budget (at least one),
    starting on a recorded sweep when configurations are recorded with a stride."""
    per_piece = max(1, int(budget_bytes) // int(bytes_per_window)) * int(window)
    if record_stride > 1:
        L = int(np.lcm(int(window), int(record_stride)))
        per_piece = per_piece // L * L or int(n_sweeps)
    return min(per_piece, int(n_sweeps))


def fused_window(n_sweeps, lo=3, hi=64):
    """Largest divisor of n_sweeps in [lo, hi] (the launch length of the fused-window schedule), or 0."""
    for t in range(min(hi, n_sweeps), lo - 1, -1):
        if n_sweeps % t == 0:
            return t
    return 0


class RoundPlanner:
    """Level schedules planned ahead for `n_rounds` rounds of `sweeps_per_round` sweeps each (they depend on the RNG
    only), a bounded number of rounds at a time.  Rounds whose sweeps need no per-sweep output run on the fused-window
    schedule when the instance qualifies (one or more launches of `window` sweeps), everything else on the
    sweep-by-sweep schedule; the results are the same bits either way.  With `pt_pairs` > 0 the pair selections of the
    swap rounds `pt_round0 + ii` (they depend on the RNG only, too) are planned with every chunk."""

    BYTES_PER_UPDATE = 140              # packed schedule: head 8 + row window 128 + scratch

    def __init__(self, eng, sweep0, n_rounds, sweeps_per_round, seed, precision="f32", budget_bytes=1 << 30,
                 chunk_rounds=None, pt_pairs=0, pt_round0=0, slot=0, beta=None, fused_outputs=False):
        """`slot`: fused-plan slot of the engine this planner owns; `beta`: None = the ladder temperatures, or one inverse
        temperature for every chain; `fused_outputs`: rounds with per-sweep outputs also run on fused windows (the
        output variant of the kernel) when the plan covers them."""
        self.eng, self.sweep0, self.R, self.S, self.seed, self.precision = eng, int(sweep0), int(n_rounds), int(sweeps_per_round), int(seed), precision
        self.slot, self.beta, self.fused_outputs = int(slot), beta, bool(fused_outputs)
        self.window = fused_window(self.S)
        if self.window and precision != "f32" and (not hasattr(eng, "fused_modes") or precision not in eng.fused_modes(self.window)):
            self.window = 0              # fp64 mode: fused windows only where the field is an exact integer (nlmc_fused_modes)
        per_round = max(1, self.S * eng.n * self.BYTES_PER_UPDATE)
        self.chunk = max(1, min(self.R, int(budget_bytes // per_round)))
        if chunk_rounds:
            self.chunk = max(1, min(self.chunk, int(chunk_rounds)))
        self.pt_pairs, self.pt_round0 = int(pt_pairs), int(pt_round0)
        self._fused_from = self._fused_to = self._plain_from = self._plain_to = 0      # planned round ranges
        self.chunks_planned = 0

    def _plan(self, ii, want_fused):
        r0, r1 = ii, min(self.R, ii + self.chunk)
        self.chunks_planned += 1
        if self.pt_pairs > 0 and hasattr(self.eng, "pt_plan"):
            self.eng.pt_plan(self.pt_round0 + r0, r1 - r0, self.seed, self.pt_pairs)
        if want_fused and self.window:
            if hasattr(self.eng, "plan_slot"):
                self.eng.plan_slot(self.slot)
            k = self.eng.plan_philox_fused(self.sweep0 + r0 * self.S, (r1 - r0) * (self.S // self.window), self.window, self.seed)
            if k == (r1 - r0) * (self.S // self.window):
                self._fused_from, self._fused_to = r0, r1
                return True
            self.window = 0              # the instance does not qualify: stop asking
        if not want_fused or not self.window:
            self.eng.plan_philox(self.sweep0 + r0 * self.S, (r1 - r0) * self.S, self.seed, precision=self.precision)
            self._plain_from, self._plain_to = r0, r1
        return False

    def sweep(self, ii, **outputs):
        """The sweeps of round ii at the PT ladder temperatures.  `outputs`: record_stride / want_* of sweep_philox."""
        needs_plain = any(outputs.get(k) for k in ("record_stride", "want_energy", "want_min", "want_state")) and not self.fused_outputs
        if self.S == 0:
            return self.eng.sweep_philox(0, self.seed, sweep0=self.sweep0, beta=self.beta, precision=self.precision, **outputs)
        if not needs_plain and self.window:
            if not (self._fused_from <= ii < self._fused_to):
                self._plan(ii, True)
            if self._fused_from <= ii < self._fused_to:      # one call: the library walks the windows of the round
                return self.eng.sweep_philox(self.S, self.seed, sweep0=self.sweep0 + ii * self.S, beta=self.beta,
                                             precision=self.precision, **outputs)
        if not (self._plain_from <= ii < self._plain_to) and not needs_plain:
            self._plan(ii, False)
        # (a round with outputs outside the planned range builds its schedule inside the call)
        return self.eng.sweep_philox(self.S, self.seed, sweep0=self.sweep0 + ii * self.S, beta=self.beta, precision=self.precision,
                                     **outputs)
