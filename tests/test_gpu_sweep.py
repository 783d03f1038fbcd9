"""Parity of the HIP sweep path (through the C-ABI) against the oracle and the reference's golden vectors.

Bars: spin configurations bit-exact; energies of the stream mode within 1e-10*max(1,|E|) of the reference
(dense BLAS vs CSR summation + 2^-scale fixed-point quanta); philox-mode energies bit-exact vs the oracle
(both sides accumulate the same integers).
"""
import numpy as np
import pytest

import oracle
from conftest import golden, golden_names
from helpers import make_instance, init_spins, draw_stream

pytestmark = pytest.mark.gpu
E_RTOL = 1e-10


def csr_of(g):
    return oracle.Csr.from_parts(int(g["N"]), g["indptr"], g["indices"], g["data"])


def assert_energy(a, b, rtol=E_RTOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b))), np.max(np.abs(a - b))


@pytest.mark.parametrize("name", golden_names("mcmc_fixed_") + golden_names("mcmc_icmvariant_"))
def test_stream_mode_reproduces_reference_mcmc(product, name):
    g = golden(name)
    csr = csr_of(g)
    S = int(g["num_sweeps"])
    np.random.seed(int(g["seed"]))
    m0 = np.sign(2 * np.random.rand(csr.n) - 1).astype(np.int8)
    perm, u = draw_stream(1, S, csr.n)
    with product.Engine(csr.toarray(), g["h"], 1) as eng:
        eng.set_spins(m0[None])
        o = eng.sweep_stream(perm, u, float(g["beta"]), record_stride=1, want_energy=True, want_min=True,
                             want_state=True)
        assert np.array_equal(o["spins"][0], g["M"])
        assert_energy(o["energy"][0], g["energies"])
        am = int(np.argmin(g["energies"]))
        assert_energy(o["min_energy"][0], g["energies"][am])
        assert np.array_equal(eng.get_spins()[0], g["M"][-1])
        assert_energy(eng.energy()[0], g["energies"][-1])
        assert_energy(eng.energy_of(g["M"]), g["energies"])


@pytest.mark.parametrize("name", golden_names("mcmc_anneal_"))
def test_stream_mode_anneal_schedule(product, name):
    g = golden(name)
    csr = csr_of(g)
    S = int(g["num_sweeps"])
    sched = product.hostlogic.beta_schedule(S, float(g["beta"]), True, int(g["sweeps_per_beta"]),
                                            float(g["initial_beta"]))
    np.random.seed(int(g["seed"]))
    m0 = np.sign(2 * np.random.rand(csr.n) - 1).astype(np.int8)
    perm, u = draw_stream(1, S, csr.n)
    with product.Engine(csr.toarray(), g["h"], 1) as eng:
        eng.set_spins(m0[None])
        o = eng.sweep_stream(perm, u, sched[None, :], record_stride=1)
        assert np.array_equal(o["spins"][0], g["M"])


def test_stream_mode_batched_chains_match_oracle(product):
    """Many chains, own permutation each, in one launch == the oracle run chain by chain."""
    J, h = make_instance(300, seed=5, with_h=True, gaussian=True)
    R, S, N = 12, 9, 300
    csr = oracle.Csr(J)
    m0 = init_spins(R, N)
    np.random.seed(77)
    perm, u = draw_stream(R, S, N)
    betas = np.linspace(0.3, 2.5, R)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        o = eng.sweep_stream(perm, u, np.repeat(betas[:, None], S, axis=1), record_stride=1, want_energy=True)
    for c in range(R):
        M, _ = oracle.sweeps_stream(csr, h, m0[c].astype(float), np.full(S, betas[c]), perm[c], u[c])
        assert np.array_equal(o["spins"][c], M)
        E = [oracle.energy(csr, h, M[t]) for t in range(S)]
        assert_energy(o["energy"][c], E)


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("order", ["shared", "per_chain"])
def test_philox_mode_bit_exact_vs_oracle(product, precision, order):
    J, h = make_instance(400, seed=9, with_h=True, gaussian=True)
    R, S, N = 10, 8, 400
    csr = oracle.Csr(J)
    m0 = init_spins(R, N)
    betas = np.linspace(0.2, 3.0, R)
    seed, sweep0 = 0xA5A50000 + (7 << 32), 5
    f64 = precision == "f64"
    with product.Engine(J, h, R, chain_base=3, n_chains_global=R + 3) as eng:
        eng.set_spins(m0)
        E0 = eng.energy()
        esc = eng.energy_scale
        o = eng.sweep_philox(S, seed, sweep0=sweep0, beta=np.repeat(betas[:, None], S, axis=1), precision=precision,
                             order=order, record_stride=1, want_energy=True, want_min=True, want_state=True)
        final = eng.get_spins()
        E_exact = eng.energy()
    for c in range(R):
        gc = c + 3
        cb = np.tile(np.array(oracle.cb_pair(betas[c], 1.0, f64)), (S, 1))
        ef0 = int(np.rint(E0[c] * 2.0 ** esc))
        M, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, seed, gc, order_group=(gc + 1 if order == "per_chain" else 0),
                                            sweep0=sweep0, escale=esc, use_f64=f64, efix0=ef0)
        assert np.array_equal(o["spins"][c], M), f"chain {c}"
        assert np.array_equal(final[c], s_fin)
        assert np.array_equal(o["energy"][c], tr.astype(np.float64) * 2.0 ** -esc)
        am = int(np.argmin(tr))
        assert o["argmin"][c] == am
        assert np.array_equal(o["argmin_state"][c], M[am])
        assert_energy(E_exact[c], oracle.energy(csr, h, s_fin), rtol=1e-12)
        # incremental fp32/fp64 energy stays close to the exact one
        assert abs(tr[-1] * 2.0 ** -esc - E_exact[c]) <= (1e-4 if not f64 else 1e-9) * max(1.0, abs(E_exact[c]))


def test_philox_mode_phase_flags(product):
    J, h = make_instance(256, seed=3, with_h=True)
    R, S, N = 6, 6, 256
    csr = oracle.Csr(J)
    m0 = init_spins(R, N)
    r = np.random.default_rng(1)
    flags = np.zeros((R, N), np.uint8)
    for c in range(R):
        cl = r.random(N) < 0.2
        if c % 2 == 0:          # phase C: clusters scaled, the rest frozen at its start value
            flags[c, cl] = 1
            flags[c, ~cl] = np.where(m0[c, ~cl] > 0, 2, 3)
        else:                   # phase NC: clusters frozen
            flags[c, cl] = np.where(m0[c, cl] > 0, 2, 3)
    beta, temp_x, seed = 2.0, 20.0, 99
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        eng.set_flags(flags, temp_x)
        E0 = eng.energy()
        esc = eng.energy_scale
        o = eng.sweep_philox(S, seed, beta=beta, record_stride=1, want_energy=True)
    for c in range(R):
        cb = np.tile(np.array(oracle.cb_pair(beta, temp_x)), (S, 1))
        M, _, tr = oracle.sweeps_philox(csr, h, m0[c], cb, seed, c, flags=flags[c], escale=esc,
                                        efix0=int(np.rint(E0[c] * 2.0 ** esc)))
        assert np.array_equal(o["spins"][c], M)
        assert np.array_equal(o["energy"][c], tr.astype(np.float64) * 2.0 ** -esc)
        frozen = flags[c] >= 2
        assert np.all(M[:, frozen] == m0[c, frozen][None, :])


def test_philox_results_do_not_depend_on_sharding_or_windows(product):
    """Same (seed, global chain id, sweep index) -> same bits, however chains are split over contexts or sweeps
    over calls (this is what makes the 1/2/4/8-GPU runs bit-identical)."""
    J, h = make_instance(200, seed=4)
    R, S, N = 8, 10, 200
    m0 = init_spins(R, N)
    betas = np.repeat(np.linspace(0.5, 2.0, R)[:, None], S, axis=1)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        eng.sweep_philox(S, 1234, sweep0=0, beta=betas)
        ref = eng.get_spins()
        Eref = eng.energy()
    out = np.empty_like(ref)
    for base in (0, 4):
        with product.Engine(J, h, 4, chain_base=base, n_chains_global=R) as eng:
            eng.set_spins(m0[base:base + 4])
            eng.sweep_philox(4, 1234, sweep0=0, beta=betas[base:base + 4, :4])
            eng.plan_philox(4, 6, 1234, precision="f32")
            eng.sweep_philox(6, 1234, sweep0=4, beta=betas[base:base + 4, 4:])
            out[base:base + 4] = eng.get_spins()
            assert np.array_equal(eng.energy(), Eref[base:base + 4])
    assert np.array_equal(out, ref)


def test_stream_mode_nmc_phase_flags_match_reference_golden(product):
    """Phase C / NC parameterisation (NMC/nmc.py:377-381,398-401) through flags, one phase at a time."""
    g = golden("nmc_subroutine_nmc_c2_s4")
    csr = csr_of(g)
    J, h, N = csr.toarray(), g["h"], csr.n
    S, cl = int(g["num_sweeps_per_NMC_phase"]), g["clusters"]
    np.random.seed(int(g["seed"]))
    m = np.sign(2 * np.random.rand(N) - 1).astype(np.int8)
    noncl = np.setdiff1d(np.arange(N), cl)
    got = []
    with product.Engine(J, h, 1) as eng:
        for cycle in range(int(g["num_cycles"])):
            for phase in ("C", "NC", "ALL"):
                fl = np.zeros((1, N), np.uint8)
                if phase == "C":
                    fl[0, cl] = 1
                    fl[0, noncl] = np.where(m[noncl] > 0, 2, 3)
                elif phase == "NC":
                    fl[0, cl] = np.where(m[cl] > 0, 2, 3)
                eng.set_spins(m[None])
                eng.set_flags(fl if phase != "ALL" else None, float(g["temp_x"]))
                perm, u = draw_stream(1, S, N)
                o = eng.sweep_stream(perm, u, float(g["global_beta"]), record_stride=1, want_min=True, want_state=True)
                got.append(o["spins"][0])
                m = o["argmin_state"][0].copy()
    assert np.array_equal(np.concatenate(got), g["M_overall"])


def test_stream_mode_chunked_calls_equal_one_call(product):
    """Engine.sweep_stream cuts long runs into several C-ABI calls; traces, energies and argmin are stitched exactly."""
    J, h = make_instance(150, seed=8, with_h=True, gaussian=True)
    R, S, N = 3, 23, 150
    m0 = init_spins(R, N)
    np.random.seed(11)
    perm, u = draw_stream(R, S, N)
    betas = np.repeat(np.array([0.4, 1.0, 2.2])[:, None], S, axis=1)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        one = eng.sweep_stream(perm, u, betas, record_stride=1, want_energy=True, want_min=True, want_state=True)
        eng.set_spins(m0)
        eng.STREAM_CHUNK_BYTES = R * N * 36 * 5          # 5 sweeps per call
        many = eng.sweep_stream(perm, u, betas, record_stride=1, want_energy=True, want_min=True, want_state=True)
    for k in ("spins", "energy", "min_energy", "argmin", "argmin_state"):
        assert np.array_equal(one[k], many[k]), k
