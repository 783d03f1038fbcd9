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
            _, s, tr = oracle.sweeps_philox(self.csr, self.h, self.spins[c], cb, seed, self._rng_id(c), sweep0=sweep0, escale=self.esc,
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

    # ---- APT run cut into temperature-slot blocks (include/nlmc.h: nlmc_apt_shard; csrc/nlmc_apt.h), restated with the oracle ----
    rng_stride = 0

    def apt_shard(self, beta_global, world, rank):
        self.beta_global = np.asarray(beta_global, float)
        self.apt_world, self.apt_rank, self.apt_R = int(world), int(rank), len(self.beta_global)
        assert self.apt_R == world * self.L and self.chain_base == 0 and self.n_chains == self.G
        self.rng_stride, self.rng_base = self.apt_R, rank * self.L

    def pt_set_slots(self, slots):
        self.slots = np.asarray(slots, np.int32).copy()

    def pt_plan(self, *a, **k):
        pass

    def energy_tracked(self):
        return self._energies()

    def _rng_id(self, c):
        return (c // self.L) * self.rng_stride + self.rng_base + int(self.slots[c]) if self.rng_stride else self.chain_base + c

    def _chain_of_slot(self):
        cos = np.empty(self.n_chains, int)
        cos[(np.arange(self.n_chains) // self.L) * self.L + self.slots] = np.arange(self.n_chains)
        return cos

    def icm_round_ladders(self, rnd, seed, katzgraber=True, want_info=False):
        """k_icm_round with the pairing made on the fly: keys philox(ladder, round, GLOBAL slot, ICM_PAIR); the pick is keyed by the
        (ladder, global slot) ids of the two chains."""
        lo, hi = int(seed) & 0xFFFFFFFF, int(seed) >> 32
        K, cos = self.n_chains // self.L, self._chain_of_slot()
        info = []
        for r in range(self.L):
            rg = r + (self.rng_base if self.rng_stride else 0)
            keys = [int(oracle.philox(j, rnd, rg, 6, lo, hi)[0]) for j in range(K)]
            sh = sorted(range(K), key=lambda j: (keys[j], j))
            for q in range(K // 2):
                ja, jb = sh[2 * q], sh[2 * q + 1]
                a, b = cos[ja * self.L + r], cos[jb * self.L + r]
                cl = oracle.clusters(self.csr, self.spins[a], self.spins[b])
                if not cl:
                    info.append((0, 0))
                    continue
                ida, idb = (self._rng_id(a), self._rng_id(b)) if self.rng_stride else (self.chain_base + a, self.chain_base + b)
                pick = (int(oracle.philox(ida, rnd, idb, opt.TAG_ICM, lo, hi)[0]) * len(cl)) >> 32
                info.append((len(cl), len(cl[pick])))
                if katzgraber and len(cl[pick]) > self.n // 2:
                    self.spins[a] = -self.spins[a]
                else:
                    sa, sb = self.spins[a].copy(), self.spins[b].copy()
                    self.spins[a, cl[pick]], self.spins[b, cl[pick]] = sb[cl[pick]], sa[cl[pick]]
                for c in (a, b):
                    self.efix[c] = int(np.rint(oracle.energy(self.csr, self.h, self.spins[c]) * 2.0 ** self.esc))
        return np.array(info, np.int32).reshape(-1, 2) if want_info else None

    def apt_pack(self, want_configs=True):
        K, cos = self.n_chains // self.L, self._chain_of_slot()
        ef = self.efix[cos].reshape(K, self.L).copy()
        lo = self.spins[cos.reshape(K, self.L)[:, 0]].copy()
        hi = self.spins[cos.reshape(K, self.L)[:, -1]].copy()
        return ef, lo, hi

    def apt_swap_host(self, rnd, seed, n_pairs, efix_all, recv_lo, recv_hi, want_log=False):
        """k_apt_swap + k_apt_adopt: the decision of every selected pair of the GLOBAL ladder from the gathered energies; pairs inside
        this block exchange labels, accepted boundary pairs adopt the neighbour's configuration and energy."""
        lo, hi = int(seed) & 0xFFFFFFFF, int(seed) >> 32
        L, R, K, me = self.L, self.apt_R, self.n_chains // self.L, self.apt_rank
        E = np.asarray(efix_all, np.int64).reshape(self.apt_world, K, L)
        pairs_all, acc_all = [], []
        for j in range(K):
            cos = self._chain_of_slot()
            avail, sel = list(range(R - 1)), []
            for p in range(n_pairs):
                if not avail:
                    raise ValueError("Cannot find non-overlapping pairs.")
                r = int(oracle.philox(p, rnd, j, opt.TAG_PAIR, lo, hi)[0])
                i = avail[(r * len(avail)) >> 32]
                sel.append(i)
                avail = [q for q in avail if abs(q - i) > 1]
            for p, i in enumerate(sel):
                wa, wb = i // L, (i + 1) // L
                la, lb = i - wa * L, i + 1 - wb * L
                Ea, Eb = float(E[wa, j, la]) * 2.0 ** -self.esc, float(E[wb, j, lb]) * 2.0 ** -self.esc
                w = oracle.philox(p, rnd, j, opt.TAG_SWAP, lo, hi)
                u = ((int(w[0]) >> 5) * 67108864.0 + (int(w[1]) >> 6)) / 9007199254740992.0
                z = ((self.beta_global[i + 1] - self.beta_global[i]) * (Eb - Ea)) * opt.LOG2E
                acc = u < oracle.lib().nlo_exp2_f64(z)
                if acc:
                    if wa == me and wb == me:
                        ca, cb = cos[j * L + la], cos[j * L + lb]
                        self.slots[ca], self.slots[cb] = lb, la
                        cos[j * L + la], cos[j * L + lb] = cb, ca
                    elif wa == me:
                        ch = cos[j * L + L - 1]
                        self.spins[ch], self.efix[ch] = recv_hi[j], E[me + 1, j, 0]
                    elif wb == me:
                        ch = cos[j * L]
                        self.spins[ch], self.efix[ch] = recv_lo[j], E[me - 1, j, L - 1]
                pairs_all.append((i, i + 1))
                acc_all.append(int(acc))
        return np.array(pairs_all, np.int32).reshape(K, n_pairs, 2), np.array(acc_all, np.uint8).reshape(K, n_pairs)

    def apt_swap_collective(self, rnd, seed, n_pairs, want_log=False):
        assert self.apt_world == 1
        return self.apt_swap_host(rnd, seed, n_pairs, self.apt_pack()[0][None], None, None, want_log)
