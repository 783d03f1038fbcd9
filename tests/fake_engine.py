"""Test double with the Engine interface that ShardedTempering drives, backed by the ORACLE (CPU).  It exists so
that the multi-process orchestration (partition, all-gather, replicated swap decision) can be rehearsed with
`gloo` on machines without a GPU.  Never used by the product."""
import ctypes

import numpy as np

import oracle
from oracle import pt as opt


class OracleEngine:
    def __init__(self, inst, n_chains, chain_base, n_chains_global):
        self.csr = oracle.Csr.from_parts(inst.n, inst.indptr, inst.indices, inst.data)
        self.h = inst.h
        self.n, self.n_chains, self.chain_base, self.G = inst.n, n_chains, chain_base, n_chains_global
        self.esc = oracle.field_scale(self.csr, inst.h)[1]
        self.spins = np.ones((n_chains, inst.n), np.int8)
        self.efix = np.zeros(n_chains, np.int64)

    def pt_init(self, betas):
        self.betas = np.asarray(betas, float)
        self.L = self.ladder_len = len(self.betas)
        self.slots = (np.arange(self.G) % self.L).astype(np.int32)

    def set_spins(self, s):
        self.spins = np.asarray(s, np.int8).reshape(self.n_chains, self.n).copy()
        self.efix = np.array([int(np.rint(oracle.energy(self.csr, self.h, self.spins[c]) * 2.0 ** self.esc))
                              for c in range(self.n_chains)], np.int64)

    def get_spins(self):
        return self.spins.copy()

    def plan_philox(self, *a, **k):
        self.planned_plain = getattr(self, "planned_plain", 0) + 1

    def plan_philox_fused(self, sweep0, n_windows, window, seed):
        """The oracle has no level schedule: it either 'plans' every window (fused_ok) or none, so that the planner's
        two branches are both exercised; a fused launch is then just `window` sequential sweeps."""
        self.fused_calls = getattr(self, "fused_calls", 0) + 1
        return n_windows if getattr(self, "fused_ok", True) else 0

    def sweep_philox(self, n_sweeps, seed, sweep0=0, beta=None, precision="f32"):
        assert beta is None
        for c in range(self.n_chains):
            gc = self.chain_base + c
            cb = np.tile(np.array(oracle.cb_pair(self.betas[self.slots[gc]], 1.0, precision == "f64")), (n_sweeps, 1))
            _, s, tr = oracle.sweeps_philox(self.csr, self.h, self.spins[c], cb, seed, gc, sweep0=sweep0, escale=self.esc,
                                            use_f64=precision == "f64", efix0=int(self.efix[c]), want_M=False)
            self.spins[c] = s
            self.efix[c] = tr[-1] if n_sweeps else self.efix[c]

    def _energies(self):
        return self.efix.astype(np.float64) * 2.0 ** -self.esc

    def energy_dev(self, ptr):
        buf = (ctypes.c_double * self.n_chains).from_address(int(ptr))
        buf[:] = list(self._energies())

    def pt_swap_philox(self, rnd, seed, n_pairs, energies_all_dev=None, want_log=True):
        if energies_all_dev:
            E = np.array((ctypes.c_double * self.G).from_address(int(energies_all_dev))[:])
        elif self.n_chains == self.G:
            E = self._energies()
        else:
            # a context that owns whole ladders decides them from its own energies (include/nlmc.h: nlmc_pt_swap_philox)
            assert self.chain_base % self.L == 0 and self.n_chains % self.L == 0
            b, e = self.chain_base, self.chain_base + self.n_chains
            self.slots[b:e], pairs, acc = opt.swap_round(self._energies(), self.slots[b:e], self.betas, self.L, n_pairs, rnd, seed,
                                                         ladder0=b // self.L)
            return pairs, acc
        self.slots, pairs, acc = opt.swap_round(E, self.slots, self.betas, self.L, n_pairs, rnd, seed)
        return pairs, acc

    def energy_tracked(self):
        return self._energies()

    def pt_swap_philox_host(self, rnd, seed, n_pairs, energies_all, want_log=False):
        self.slots, pairs, acc = opt.swap_round(np.asarray(energies_all, float), self.slots, self.betas, self.L, n_pairs, rnd, seed)
        return pairs, acc

    def pt_slots(self):
        return self.slots.copy()

    def close(self):
        pass
