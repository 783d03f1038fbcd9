"""k_rounds_fused (include/nlmc.h: nlmc_pt_rounds_fused): many rounds -- sweeps + replica exchange -- in one cooperative launch, and
nlmc_pt_rounds_deferred: one sweep launch per round that decides the PREVIOUS round's swap in its prologue -- against the same rounds
driven with a sweep launch and a swap launch each: spins, tracked energies, slot maps and the swap log must be the same bits."""
import numpy as np
import pytest

from helpers import make_instance, init_spins
from test_gpu_fused64 import integer_instance

pytestmark = pytest.mark.gpu
SEED = 0xA5A50000


def drive(product, inst, G, L, T, rounds, pairs, precision, persistent, m0, chunk=None, split=None):
    betas = np.geomspace(0.1, 3.0, L)
    with product.Engine(inst, None, G) as eng:
        eng.set_spins(m0)
        eng.pt_init(betas)
        assert eng.plan_philox_fused(0, rounds, T, SEED) == rounds
        eng.pt_plan(0, rounds, SEED, pairs)
        eng.pt_log_begin(0, rounds, pairs)
        if persistent:
            at = 0
            batch = eng.pt_rounds_deferred if persistent == "deferred" else eng.pt_rounds_fused
            for k in (split or [rounds]):
                assert batch(k, T, SEED, at * T, at, pairs, precision=precision), getattr(eng, "rounds_fused_refusal", "")
                at += k
        else:
            for r in range(rounds):
                eng.sweep_philox(T, SEED, sweep0=r * T, beta=None, precision=precision)
                eng.pt_swap_philox(r, SEED, pairs, want_log=False)
        p, a = eng.pt_log_read()
        spins, E, slots = eng.get_spins(), eng.energy(), eng.pt_slots()
        exact = eng.energy_of(spins)
    return spins, E, slots, p, a, exact


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_persistent_rounds_equal_rounds_launched_one_by_one(product, precision):
    N, L, nl, T, rounds, pairs = 4000, 8, 3, 5, 7, 3
    J, h = make_instance(N, seed=6)
    inst = product.Instance(J, h)
    G = L * nl
    m0 = init_spins(G, N)
    ref = drive(product, inst, G, L, T, rounds, pairs, precision, False, m0)
    assert ref[4].sum() > 0 and not np.array_equal(ref[2], np.arange(G) % L)
    for mode in (True, "deferred"):
        for split in (None, [3, 4], [1, 1, 5]):
            got = drive(product, inst, G, L, T, rounds, pairs, precision, mode, m0, split=split)
            for x, y in zip(got, ref):
                assert np.array_equal(x, y), (mode, split)
    assert np.array_equal(ref[1], ref[5])                          # tracked == recomputed (+-J)


@pytest.mark.parametrize("case", ["integer_diag", "gaussian"])
def test_persistent_rounds_other_formats(product, case):
    """Compact entries + diagonal + fields + hub rows (fp64 mode), Gaussian couplings (wide entries, f32 mode)."""
    N, L, nl, T, rounds, pairs = 3000, 6, 2, 6, 4, 2
    if case == "integer_diag":
        J, h = integer_instance(N, 4, wmax=2, diag=True, h_step=0.25)
        precision = "f64"
    else:
        J, h = make_instance(N, seed=9, with_h=True, gaussian=True)
        precision = "f32"
    inst = product.Instance(J, h)
    G = L * nl
    m0 = init_spins(G, N)
    ref = drive(product, inst, G, L, T, rounds, pairs, precision, False, m0)
    for mode in (True, "deferred"):
        got = drive(product, inst, G, L, T, rounds, pairs, precision, mode, m0)
        for x, y in zip(got[:5], ref[:5]):
            assert np.array_equal(x, y), mode


def test_persistent_rounds_refusals_and_the_driver(product):
    """What does not qualify is refused BEFORE anything runs (the driver then takes the launch-per-round path): no plan, a window
    that is not one round, phase flags, an fp64 call on inexact couplings; ShardedTempering.run_rounds == round() x n."""
    N, L, T = 2000, 6, 5
    J, h = make_instance(N, seed=2)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.2, 2.5, L)
    m0 = init_spins(L, N)
    with product.Engine(inst, None, L) as eng:
        eng.set_spins(m0)
        eng.pt_init(betas)
        assert not eng.pt_rounds_fused(2, T, SEED, 0, 0, 2) and "plan" in eng.rounds_fused_refusal
        assert eng.plan_philox_fused(0, 4, T, SEED) == 4
        assert not eng.pt_rounds_fused(2, T, SEED, 0, 0, 2) and "pair selections" in eng.rounds_fused_refusal
        eng.pt_plan(0, 4, SEED, 2)
        assert not eng.pt_rounds_fused(2, 2 * T, SEED, 0, 0, 2)                  # a round of two windows
        fl = np.zeros((L, N), np.uint8); fl[:, :10] = 1
        eng.set_flags(fl, 20.0)
        assert not eng.pt_rounds_fused(2, T, SEED, 0, 0, 2) and "flags" in eng.rounds_fused_refusal
        eng.set_flags(None)
        assert np.array_equal(eng.get_spins(), m0)                                # nothing ran
        assert eng.pt_rounds_fused(2, T, SEED, 0, 0, 2)
        eng.pt_check()
    Jg, hg = make_instance(N, seed=2, gaussian=True)
    with product.Engine(Jg, hg, L) as eng:
        eng.set_spins(m0); eng.pt_init(betas)
        assert eng.plan_philox_fused(0, 2, T, SEED) == 2
        eng.pt_plan(0, 2, SEED, 2)
        assert not eng.pt_rounds_fused(2, T, SEED, 0, 0, 2, precision="f64") and "fp64" in eng.rounds_fused_refusal

    def run(mode):
        st = product.distributed.ShardedTempering(lambda i, n, b, g: product.Engine(i, None, n, chain_base=b, n_chains_global=g), inst, betas,
                                                  L, SEED, 2, precision="f64")
        st.set_spins(m0)
        st.plan(11 * T, 11, chunk_rounds=4, lazy=True)
        if mode == "persistent":
            st.run_rounds(11, T, persistent=True)
            assert st.persistent_rounds == 11
        elif mode == "deferred":
            st.run_rounds(11, T)                                   # the default of run_rounds where it applies
            assert st.deferred_rounds == 11
        else:
            for _ in range(11):
                st.round(T)
        out = st.eng.get_spins(), st.eng.energy(), st.eng.pt_slots()
        st.close()
        return out
    a, b, c = run("persistent"), run("per-round"), run("deferred")
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and all(np.array_equal(x, y) for x, y in zip(c, b))


def test_c4_size_persistent_rounds(product):
    """The bench workload (N = 10^4, 256 replicas on one ladder, 10 sweeps per round, 77 pairs, fp64 mode): 6 rounds in one launch ==
    6 x (sweep launch + swap launch); tracked == recomputed energies; twice the same bits."""
    N, G, T, rounds, pairs = 10_000, 256, 10, 6, 77
    J, h = make_instance(N)
    inst = product.Instance(J, h)
    m0 = init_spins(G, N)
    ref = drive(product, inst, G, G, T, rounds, pairs, "f64", False, m0)
    a = drive(product, inst, G, G, T, rounds, pairs, "f64", True, m0)
    b = drive(product, inst, G, G, T, rounds, pairs, "f64", True, m0, split=[2, 4])
    d = drive(product, inst, G, G, T, rounds, pairs, "f64", "deferred", m0, split=[1, 5])
    for x, y, z, v in zip(a, ref, b, d):
        assert np.array_equal(x, y) and np.array_equal(z, y) and np.array_equal(v, y)
    assert np.array_equal(ref[1], ref[5]) and ref[4].sum() > 100
