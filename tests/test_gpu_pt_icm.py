"""Device-decided replica exchange and iso-cluster move against small host restatements of their published laws."""
import numpy as np
import pytest

import oracle
from helpers import make_instance, init_spins

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634
TAG_SWAP, TAG_PAIR, TAG_ICM = 3, 4, 5


from oracle.pt import swap_round as expected_swap_round  # noqa: E402


def test_pt_swap_rounds_match_restatement(product):
    J, h = make_instance(120, seed=2, with_h=True)
    L, nl, n_pairs = 8, 3, 3
    G = L * nl
    betas = np.geomspace(0.2, 3.0, L)
    with product.Engine(J, h, G) as eng:
        eng.set_spins(init_spins(G, 120))
        eng.pt_init(betas)
        slots = eng.pt_slots()
        assert np.array_equal(slots, np.arange(G) % L)
        eng.pt_plan(2, 3, 77, n_pairs)          # rounds 2..4 use planned selections: results must not change
        for rnd in range(6):
            eng.sweep_philox(3, 77, sweep0=3 * rnd, beta=None)
            E = eng.energy()
            exp_slots, exp_pairs, exp_acc = expected_swap_round(E, slots, betas, L, n_pairs, rnd, 77)
            pairs, acc = eng.pt_swap_philox(rnd, 77, n_pairs)
            slots = eng.pt_slots()
            assert np.array_equal(pairs, exp_pairs)
            assert np.array_equal(acc, exp_acc)
            assert np.array_equal(slots, exp_slots)
        # every ladder still holds each slot exactly once
        assert all(sorted(slots[g * L:(g + 1) * L]) == list(range(L)) for g in range(nl))
        assert acc.sum() + 1 > 0


def test_pt_ladder_beta_follows_slot(product):
    """A chain that holds slot r sweeps at beta_list[r]: compare with explicit per-chain betas."""
    J, h = make_instance(150, seed=8)
    L = 6
    betas = np.linspace(0.3, 2.4, L)
    m0 = init_spins(L, 150)
    perm = np.array([3, 0, 5, 1, 4, 2], dtype=np.int32)
    with product.Engine(J, h, L) as a, product.Engine(J, h, L) as b:
        a.set_spins(m0)
        a.pt_init(betas)
        a.pt_set_slots(perm)
        a.sweep_philox(5, 9, beta=None)
        b.set_spins(m0)
        b.sweep_philox(5, 9, beta=np.repeat(betas[perm][:, None], 5, axis=1))
        assert np.array_equal(a.get_spins(), b.get_spins())
        a.pt_apply_swap(0, 0, 1)
        s = a.pt_slots()
        assert s[1] == 1 and s[3] == 0


def test_pt_too_many_pairs_raises_like_reference(product):
    J, h = make_instance(60, seed=1)
    with product.Engine(J, h, 4) as eng:
        eng.set_spins(init_spins(4, 60))
        eng.pt_init(np.array([0.5, 1.0, 1.5, 2.0]))
        with pytest.raises(ValueError, match="non-overlapping"):
            for rnd in range(64):       # (1,2) first leaves nothing for a second pair: happens w.p. 1/3 per round
                eng.pt_swap_philox(rnd, 5, 2)


def test_icm_move_matches_oracle_components(product):
    J, h = make_instance(300, seed=6)
    csr = oracle.Csr(J)
    r = np.random.default_rng(4)
    for trial in range(6):
        s1 = r.choice([-1, 1], 300).astype(np.int8)
        s2 = s1.copy()
        s2[r.random(300) < (0.1 + 0.15 * trial)] *= -1
        cl = oracle.clusters(csr, s1, s2)
        with product.Engine(J, h, 2) as eng:
            eng.set_spins(np.stack([s1, s2]))
            assert eng.icm_components(0, 1) == len(cl)
            lab = eng.icm_labels()
            for c in cl:
                assert np.all(lab[c] == c.min())
            assert np.all(lab[(s1 * s2) == 1] == -1)
            if not cl:
                continue
            pick = trial % len(cl)
            ncomp, size = eng.icm_move(0, 1, pick, katzgraber=True)
            got = eng.get_spins()
            e1, e2 = s1.copy(), s2.copy()
            if len(cl[pick]) > 300 // 2:
                e1 = -e1
            else:
                e1[cl[pick]], e2[cl[pick]] = s2[cl[pick]], s1[cl[pick]]
            assert (ncomp, size) == (len(cl), len(cl[pick]))
            assert np.array_equal(got[0], e1) and np.array_equal(got[1], e2)
            # tracked energies were re-synchronised after the non-local move
            assert np.allclose(eng.energy(), [oracle.energy(csr, h, e1), oracle.energy(csr, h, e2)], rtol=0, atol=1e-9)


def test_icm_round_philox_batch(product):
    J, h = make_instance(200, seed=12)
    csr = oracle.Csr(J)
    m0 = init_spins(8, 200)
    pairs = np.array([[0, 5], [2, 3], [7, 1]], dtype=np.int32)
    seed, rnd = 1234567, 3
    with product.Engine(J, h, 8, chain_base=4, n_chains_global=16) as eng:
        eng.set_spins(m0)
        info = eng.icm_round_philox(pairs, rnd, seed, katzgraber=True, want_info=True)
        got = eng.get_spins()
    exp = m0.copy()
    for p, (a, b) in enumerate(pairs):
        cl = oracle.clusters(csr, m0[a], m0[b])
        assert info[p, 0] == len(cl)
        if not cl:
            continue
        w = int(oracle.philox(4 + a, rnd, 4 + b, TAG_ICM, seed & 0xFFFFFFFF, seed >> 32)[0])
        pick = (w * len(cl)) >> 32
        assert info[p, 1] == len(cl[pick])
        if len(cl[pick]) > 100:
            exp[a] = -exp[a]
        else:
            exp[a, cl[pick]], exp[b, cl[pick]] = m0[b, cl[pick]], m0[a, cl[pick]]
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("with_h", [False, True])
def test_icm_round_ladders_pairing_and_moves(product, with_h):
    """nlmc_icm_round_ladders: per temperature slot the K ladders are shuffled by Philox keys and paired
    (NPT/apt_ICM.py:216-222); each pair gets the iso-cluster move of test_icm_round_philox_batch.  Restated on the host
    with the oracle's Philox and cluster routines, after a few swap rounds so that slots are no longer the identity."""
    TAG_ICM_PAIR = 6
    N, R, K = 150, 4, 6
    J, h = make_instance(N, seed=21, with_h=with_h)
    csr = oracle.Csr(J)
    G = R * K
    seed, rnd = 99887766, 5
    with product.Engine(J, h, G) as eng:
        eng.set_spins(init_spins(G, N))
        eng.pt_init(np.geomspace(0.2, 2.0, R))
        eng.sweep_philox(3, seed, beta=None)
        for r0 in range(3):
            eng.pt_swap_philox(r0, seed, 1, want_log=False)
        slots = eng.pt_slots()
        m0 = eng.get_spins()
        info = eng.icm_round_ladders(rnd, seed, katzgraber=True, want_info=True)
        got = eng.get_spins()
        E_tracked = eng.energy_tracked()         # re-synchronised by the round (no drift against a later recomputation)
        qs = eng.field_scale
        E = eng.energy()
    # exact for integer couplings and fields; otherwise within the quantisation bound of the fixed-point model
    tol = 0.0 if not with_h else 2.0 ** -(qs + 1) * (J.nnz / 2 + N) * 4
    assert np.max(np.abs(E_tracked - E)) <= tol
    assert not np.array_equal(slots, np.arange(G) % R)
    exp = m0.copy()
    p = 0
    for r in range(R):
        keys = [int(oracle.philox(j, rnd, r, TAG_ICM_PAIR, seed & 0xFFFFFFFF, seed >> 32)[0]) for j in range(K)]
        sh = sorted(range(K), key=lambda j: (keys[j], j))
        holder = {j: j * R + int(np.where(slots[j * R:(j + 1) * R] == r)[0][0]) for j in range(K)}
        for q in range(K // 2):
            a, b = holder[sh[2 * q]], holder[sh[2 * q + 1]]
            cl = oracle.clusters(csr, m0[a], m0[b])
            assert info[p, 0] == len(cl)
            if cl:
                w = int(oracle.philox(a, rnd, b, TAG_ICM, seed & 0xFFFFFFFF, seed >> 32)[0])
                pick = (w * len(cl)) >> 32
                assert info[p, 1] == len(cl[pick])
                if len(cl[pick]) > N // 2:
                    exp[a] = -exp[a]
                else:
                    exp[a, cl[pick]], exp[b, cl[pick]] = m0[b, cl[pick]], m0[a, cl[pick]]
            p += 1
    assert np.array_equal(got, exp)
    assert np.allclose(E, [oracle.energy(csr, h, s) for s in got], rtol=0, atol=1e-9)


def test_energy_sink_is_written_by_the_sweep_kernels(product):
    """nlmc_set_energy_sink: the buffer the sharded driver all-gathers from holds the tracked energies after every sweep
    call (plain and fused path), without a separate conversion launch."""
    from helpers import DeviceBuffer
    N, R, T = 2048, 6, 5
    J, h = make_instance(N, seed=4)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(init_spins(R, N))
        eng.pt_init(np.geomspace(0.2, 2.0, R))
        buf = DeviceBuffer(np.zeros(R))
        eng.set_energy_sink(buf.ptr.value)
        eng.sweep_philox(T, 9, sweep0=0, beta=None)                       # plain path
        assert np.array_equal(buf.read(R), eng.energy())
        assert eng.plan_philox_fused(T, 1, T, 9) == 1
        eng.sweep_philox(T, 9, sweep0=T, beta=None)                       # fused window
        assert eng.last_schedule_stats()["orders"] == T
        assert np.array_equal(buf.read(R), eng.energy())
        eng.set_energy_sink(None)
        before = buf.read(R)
        eng.sweep_philox(T, 9, sweep0=2 * T, beta=None)
        assert np.array_equal(buf.read(R), before)                        # switched off
        buf.free()


def test_library_issued_all_gather_one_rank_communicator(product):
    """The per-round all-gather of a sharded ladder is issued by the library itself (RCCL bound at run time, communicator from
    nlmc_comm_init) on the kernels' stream.  A one-rank communicator rehearses the whole path on one GPU: sweeps write their
    energies into the gathered vector, ncclAllGather in place, the swap kernel reads it -- same decisions as the
    single-context round, also when the states changed outside a sweep call (refresh_energies)."""
    J, h = make_instance(200, seed=12, with_h=True)
    L, nl, n_pairs = 8, 2, 3
    G = L * nl
    betas = np.geomspace(0.2, 3.0, L)
    with product.Engine(J, h, G) as a, product.Engine(J, h, G) as b:
        for e in (a, b):
            e.set_spins(init_spins(G, 200))
            e.pt_init(betas)
        try:
            a.comm_init(product.Engine.comm_unique_id(), 1, 0)
        except NotImplementedError as ex:                      # no librccl on this box
            pytest.skip(str(ex))
        for rnd in range(5):
            for e in (a, b):
                e.sweep_philox(4, 91, sweep0=4 * rnd, beta=None)
            pa, aa = a.pt_swap_philox_collective(rnd, 91, n_pairs, want_log=True)
            pb, ab = b.pt_swap_philox(rnd, 91, n_pairs)
            assert np.array_equal(pa, pb) and np.array_equal(aa, ab) and np.array_equal(a.pt_slots(), b.pt_slots())
        s = init_spins(G, 200, base=77)
        for e in (a, b):
            e.set_spins(s)                                     # states replaced behind the sweep kernels' back
        pa, aa = a.pt_swap_philox_collective(9, 91, n_pairs, refresh_energies=True, want_log=True)
        pb, ab = b.pt_swap_philox(9, 91, n_pairs)
        assert np.array_equal(pa, pb) and np.array_equal(aa, ab) and np.array_equal(a.pt_slots(), b.pt_slots())
    # a context that does not own an equal block of the chains is refused
    with product.Engine(J, h, 4, chain_base=0, n_chains_global=16) as c:
        c.pt_init(betas)
        with pytest.raises(ValueError):
            c.comm_init(product.Engine.comm_unique_id(), 2, 1)
