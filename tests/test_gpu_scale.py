"""Parity at BASELINE.json's full sizes through properties that do not need a full oracle run:
 * tracked (incremental, fixed-point) energies == fp64 recomputation from the final configurations (exact for +-J);
 * a sample of chains re-run by the sequential oracle (spins bit for bit) at N = 10^4;
 * results independent of how the 256 chains are split over contexts (the multi-GPU invariant);
 * the two RNG modes sample the same law (mean energy per spin at fixed beta agrees within statistical error)."""
import numpy as np
import pytest

import oracle
from helpers import make_instance, init_spins, draw_stream, DeviceBuffer

pytestmark = pytest.mark.gpu


def test_c4_size_energy_consistency_and_oracle_sample(product):
    N, R, S = 10_000, 256, 12
    J, h = make_instance(N)
    csr = oracle.Csr(J)
    betas = np.geomspace(0.05, 4.0, R)
    m0 = init_spins(R, N)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        eng.pt_init(betas)
        E0 = eng.energy()
        o = eng.sweep_philox(S, 0xA5A50000, beta=None, want_energy=True)
        tracked = o["energy"][:, -1]
        spins = eng.get_spins()
        exact = eng.energy()
        esc = eng.energy_scale
    assert np.array_equal(tracked, exact)                     # integer-valued instance: no rounding anywhere
    assert np.all(exact[128:] < E0[128:])                      # the cold half of the ladder relaxed from its random start
    for c in (0, 77, 255):                                     # hot, middle, cold chain against the sequential spec
        cb = np.tile(np.array(oracle.cb_pair(betas[c])), (S, 1))
        _, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, 0xA5A50000, c, escale=esc,
                                            efix0=int(np.rint(E0[c] * 2.0 ** esc)), want_M=False)
        assert np.array_equal(spins[c], s_fin)
        assert np.array_equal(o["energy"][c], tr * 2.0 ** -esc)
        assert exact[c] == oracle.energy(csr, h, s_fin)


def test_c3_size_ladders_split_over_contexts(product):
    """C3: N = 10^3, 32 betas x 8 restarts = 256 chains; 2 'ranks' of 128 chains reproduce the single context,
    including the device-decided swap rounds fed with all-gathered energies."""
    N, L, NL, S, ROUNDS, PAIRS, SEED = 1000, 32, 8, 5, 6, 10, 99
    J, h = make_instance(N, seed=7)
    betas = np.geomspace(0.1, 3.0, L)
    G = L * NL
    m0 = init_spins(G, N)

    def run(parts):
        engs = [product.Engine(J, h, cnt, chain_base=base, n_chains_global=G) for base, cnt in parts]
        try:
            for e, (base, cnt) in zip(engs, parts):
                e.set_spins(m0[base:base + cnt])
                e.pt_init(betas)
            for rnd in range(ROUNDS):
                for e in engs:
                    e.sweep_philox(S, SEED, sweep0=rnd * S, beta=None)
                E_all = np.concatenate([e.energy() for e in engs])
                for e in engs:
                    if len(engs) == 1:
                        e.pt_swap_philox(rnd, SEED, PAIRS, want_log=False)
                    else:        # stand-in for the all-gather: every context sees the same full energy vector
                        buf = DeviceBuffer(E_all)
                        e.pt_swap_philox(rnd, SEED, PAIRS, energies_all_dev=buf.ptr.value, want_log=False)
                        e.pt_slots()           # blocks until the swap kernel has read the buffer
                        buf.free()
            return np.concatenate([e.get_spins() for e in engs]), engs[0].pt_slots()
        finally:
            for e in engs:
                e.close()

    s1, slots1 = run([(0, G)])
    s2, slots2 = run([(0, G // 2), (G // 2, G // 2)])
    assert np.array_equal(s1, s2)
    assert np.array_equal(slots1, slots2)
    assert not np.array_equal(slots1, np.arange(G) % L)        # swaps did happen


def test_both_rng_modes_sample_the_same_law(product):
    """Mean energy per spin at beta = 0.8 on a 3N-edge +-J graph: stream mode (reference arithmetic, NumPy stream)
    vs philox mode (fp32, device RNG).  40 chains x 60 recorded sweeps each after burn-in."""
    N, R, burn, S = 400, 40, 60, 60
    J, h = make_instance(N, seed=21)
    m0 = init_spins(R, N)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        np.random.seed(5)
        perm, u = draw_stream(R, burn + S, N)
        o = eng.sweep_stream(perm, u, 0.8, want_energy=True)
        e_stream = o["energy"][:, burn:].mean(axis=1) / N
        eng.set_spins(m0)
        o = eng.sweep_philox(burn + S, 1234, beta=0.8, want_energy=True)
        e_philox = o["energy"][:, burn:].mean(axis=1) / N
    se = np.sqrt(e_stream.var(ddof=1) / R + e_philox.var(ddof=1) / R)
    assert abs(e_stream.mean() - e_philox.mean()) < 5 * se, (e_stream.mean(), e_philox.mean(), se)
    assert -1.9 < e_philox.mean() < -1.2                      # degree-6 +-J glass at beta = 0.8


def test_c2_size_nmc_phases_batched_restarts(product):
    """C2: N = 10^3, 64 restarts in one launch, cluster phase with a fixed cluster set (first 5 % of the spins)."""
    N, R, S = 1000, 64, 30
    J, h = make_instance(N, seed=3)
    csr = oracle.Csr(J)
    m0 = init_spins(R, N)
    cl = np.arange(N // 20)
    fl = np.zeros((R, N), np.uint8)
    fl[:, cl] = 1
    rest = np.ones(N, bool)
    rest[cl] = False
    fl[:, rest] = np.where(m0[:, rest] > 0, 2, 3)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        eng.set_flags(fl, 20.0)
        E0 = eng.energy()
        o = eng.sweep_philox(S, 777, beta=3.0, want_min=True, want_state=True)
        fin = eng.get_spins()
        esc = eng.energy_scale
    assert np.array_equal(fin[:, rest], m0[:, rest])           # frozen spins never move
    for c in (0, 63):
        cb = np.tile(np.array(oracle.cb_pair(3.0, 20.0)), (S, 1))
        M, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, 777, c, flags=fl[c], escale=esc,
                                            efix0=int(np.rint(E0[c] * 2.0 ** esc)))
        assert np.array_equal(fin[c], s_fin)
        am = int(np.argmin(tr))
        assert o["argmin"][c] == am and np.array_equal(o["argmin_state"][c], M[am])


def test_c5_size_icm_round(product):
    """C5: N = 10^4, 256 chains = 128 sub-replica pairs, one batched iso-cluster round on the device; two pairs are
    re-derived with the oracle's BFS (component count, picked size, resulting states), all energies re-synchronised."""
    N, R = 10_000, 256
    J, h = make_instance(N)
    csr = oracle.Csr(J)
    m0 = init_spins(R, N)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        eng.sweep_philox(3, 5, beta=1.5)
        before = eng.get_spins()
        pairs = np.arange(R, dtype=np.int32).reshape(-1, 2)
        info = eng.icm_round_philox(pairs, 9, 4242, katzgraber=True, want_info=True)
        after = eng.get_spins()
        E = eng.energy()
    for p in (0, 127):
        a, b = pairs[p]
        cl = oracle.clusters(csr, before[a], before[b])
        assert info[p, 0] == len(cl)
        w = int(oracle.philox(int(a), 9, int(b), 5, 4242 & 0xFFFFFFFF, 4242 >> 32)[0])
        pick = (w * len(cl)) >> 32
        assert info[p, 1] == len(cl[pick])
        ea, eb = before[a].copy(), before[b].copy()
        if len(cl[pick]) > N // 2:
            ea = -ea
        else:
            ea[cl[pick]], eb[cl[pick]] = before[b][cl[pick]], before[a][cl[pick]]
        assert np.array_equal(after[a], ea) and np.array_equal(after[b], eb)
        assert E[a] == oracle.energy(csr, h, ea) and E[b] == oracle.energy(csr, h, eb)
    # Houdayer's move conserves the total energy of a pair when it exchanges a cluster (no global flip)
    with product.Engine(J, h, R) as eng:
        Eb = eng.energy_of(before)
    swapped = info[:, 1] <= N // 2
    assert np.allclose((E[0::2] + E[1::2])[swapped], (Eb[0::2] + Eb[1::2])[swapped], rtol=0, atol=1e-9)


def test_c4_size_determinism_race_screen(product):
    """Same seed twice at the bench size (256 x 10^4 spins, 30 sweeps across 3 launches): identical bits.  Any missing
    barrier / LDS ordering problem in the pipelined level loop would show up here as run-to-run differences."""
    N, R = 10_000, 256
    J, h = make_instance(N)
    betas = np.geomspace(0.05, 4.0, R)
    m0 = init_spins(R, N)
    outs = []
    for rep in range(2):
        with product.Engine(J, h, R) as eng:
            eng.set_spins(m0)
            eng.pt_init(betas)
            if rep == 1:
                eng.plan_philox(0, 30, 99)            # planned vs unplanned schedules must not matter either
            for r in range(3):
                eng.sweep_philox(10, 99, sweep0=10 * r, beta=None)
                eng.pt_swap_philox(r, 99, 77, want_log=False)
            outs.append((eng.get_spins(), eng.energy(), eng.pt_slots()))
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][2], outs[1][2])


def test_c4_size_fused_windows_oracle_sample_and_energy(product):
    """The exact workload bench.py times (N = 10^4, 256 replicas, fused windows of 10 sweeps, swap round between
    windows): tracked energies == fp64 recomputation for every replica; the hot, a middle and the cold replica of the
    first window re-run by the sequential oracle, spins bit for bit; same seed twice -> same bits (race screen)."""
    N, R, T, W, SEED, PAIRS = 10_000, 256, 10, 3, 0xA5A50000, 77
    J, h = make_instance(N)
    csr = oracle.Csr(J)
    betas = np.geomspace(0.05, 4.0, R)
    m0 = init_spins(R, N)

    def run(n_windows):
        with product.Engine(J, h, R) as eng:
            eng.set_spins(m0)
            eng.pt_init(betas)
            E0 = eng.energy()
            assert eng.plan_philox_fused(0, n_windows, T, SEED) == n_windows
            eng.pt_plan(0, n_windows, SEED, PAIRS)
            for w in range(n_windows):
                eng.sweep_philox(T, SEED, sweep0=w * T, beta=None)
                assert eng.last_schedule_stats()["levels"] / T < 18.5        # fused list (sweep by sweep: ~21)
                if w + 1 < n_windows:
                    eng.pt_swap_philox(w, SEED, PAIRS, want_log=False)
            tracked = eng.energy()                       # tracked fixed-point energies
            spins = eng.get_spins()
            exact = eng.energy_of(spins)
            return E0, tracked, spins, exact, eng.pt_slots(), eng.energy_scale

    E0, tracked, spins, exact, slots, esc = run(1)
    assert np.array_equal(tracked, exact)
    for c in (0, 131, 255):
        cb = np.tile(np.array(oracle.cb_pair(betas[c])), (T, 1))
        _, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, SEED, c, escale=esc, efix0=int(np.rint(E0[c] * 2.0 ** esc)),
                                            want_M=False)
        assert np.array_equal(spins[c], s_fin)
        assert tracked[c] == tr[-1] * 2.0 ** -esc
    a = run(W)
    b = run(W)
    assert np.array_equal(a[1], a[3])                                        # energies consistent after swaps too
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    assert not np.array_equal(a[4], np.arange(R))                            # swaps happened


def test_c5_size_icm_rounds_properties(product):
    """C5 (APT + iso-cluster moves) at full size -- N = 10^4, 32 temperatures x 8 sub-replicas = 256 chains, rounds of
    10 sweeps + device-paired Houdayer step + swaps -- through properties that need no full oracle run: a Houdayer
    exchange conserves the SUM of the two energies of a pair (checked on the whole batch: total energy before == after
    for rounds where no Katzgraber flip fired -- with h = 0 a global flip conserves it too); tracked == recomputed
    energies; spins stay +-1; every chain keeps exactly one temperature slot per ladder; same seed -> same bits."""
    N, R, K, T, ROUNDS, SEED = 10_000, 32, 8, 10, 3, 77
    J, h = make_instance(N)
    G = R * K
    m0 = init_spins(G, N)

    def run():
        with product.Engine(J, h, G) as eng:
            eng.set_spins(m0)
            eng.pt_init(np.geomspace(0.05, 4.0, R))
            pl = product.engine.RoundPlanner(eng, 0, ROUNDS, T, SEED)
            eng.pt_plan(0, ROUNDS, SEED, 10)
            sums = []
            for r in range(ROUNDS):
                pl.sweep(r)
                e_before = eng.energy()
                info = eng.icm_round_ladders(r, SEED, True, want_info=True)
                e_tracked = eng.energy_tracked()            # the round leaves the tracked energies in sync
                e_after = eng.energy()
                assert np.array_equal(e_tracked, e_after)
                sums.append((e_before.sum(), e_after.sum(), int((info[:, 1] > 0).sum())))
                eng.pt_swap_philox(r, SEED, 10, want_log=False)
            spins = eng.get_spins()
            return spins, eng.energy(), eng.energy_of(spins), eng.pt_slots(), sums

    s1, tracked, exact, slots, sums = run()
    s2, _, _, slots2, _ = run()
    assert np.array_equal(s1, s2) and np.array_equal(slots, slots2)
    assert set(np.unique(s1)) == {-1, 1}
    assert np.array_equal(tracked, exact)
    for before, after, moved in sums:
        assert before == after and moved > 0           # +-J, h = 0: integer energies, the pair sums are conserved exactly
    for j in range(K):
        assert sorted(slots[j * R:(j + 1) * R]) == list(range(R))
