"""Edge cases of the sweep path: degenerate graphs, diagonal couplings, deep (dense) schedules, rows longer than the
packed window, size limits, zero-length calls -- HIP path vs the sequential oracle, bit for bit."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from helpers import init_spins, draw_stream, make_instance

pytestmark = pytest.mark.gpu


def check_philox(product, J, h, R=3, S=5, precision="f32", beta=1.1, seed=321):
    csr = oracle.Csr(J)
    n = csr.n
    m0 = init_spins(R, n)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        E0 = eng.energy()
        esc = eng.energy_scale
        o = eng.sweep_philox(S, seed, beta=beta, precision=precision, record_stride=1, want_energy=True)
        Eend = eng.energy()
    for c in range(R):
        cb = np.tile(np.array(oracle.cb_pair(beta, 1.0, precision == "f64")), (S, 1))
        M, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, seed, c, escale=esc, use_f64=precision == "f64",
                                            efix0=int(np.rint(E0[c] * 2.0 ** esc)))
        assert np.array_equal(o["spins"][c], M)
        assert np.array_equal(o["energy"][c], tr * 2.0 ** -esc)
        assert abs(Eend[c] - oracle.energy(csr, h, s_fin)) <= 1e-9 * max(1.0, abs(Eend[c]))
    return o


def test_single_spin_and_isolated_vertices(product):
    check_philox(product, sp.csr_matrix((1, 1)), np.array([0.7]), R=2, S=4)
    J = sp.lil_matrix((9, 9))
    J[0, 1] = J[1, 0] = -1.0
    J[4, 7] = J[7, 4] = 0.5                       # spins 2,3,5,6,8 have no neighbour at all
    check_philox(product, J.tocsr(), np.linspace(-1, 1, 9), S=6)
    check_philox(product, J.tocsr(), np.linspace(-1, 1, 9), S=6, precision="f64")


def test_diagonal_couplings(product):
    """J_kk != 0 (the reference never forbids it): the field includes the diagonal term like J.dot(m) does, the
    energy bookkeeping does not double count it."""
    J, h = make_instance(120, seed=2, with_h=True, gaussian=True)
    J = (J + sp.diags(np.linspace(-0.5, 0.5, 120))).tocsr()
    check_philox(product, J, h)
    check_philox(product, J, h, precision="f64")
    csr = oracle.Csr(J)
    m0 = init_spins(1, 120)
    np.random.seed(3)
    perm, u = draw_stream(1, 6, 120)
    with product.Engine(J, h, 1) as eng:
        eng.set_spins(m0)
        o = eng.sweep_stream(perm, u, 0.9, record_stride=1, want_energy=True)
    M, _ = oracle.sweeps_stream(csr, h, m0[0].astype(float), np.full(6, 0.9), perm[0], u[0])
    assert np.array_equal(o["spins"][0], M)
    E = np.array([oracle.energy(csr, h, M[t]) for t in range(6)])
    assert np.all(np.abs(o["energy"][0] - E) <= 1e-10 * np.maximum(1, np.abs(E)))


def test_dense_graph_deep_schedule_and_long_rows(product):
    """Complete graph: every level holds one spin (n levels), every row has n-1 > 16 entries (CSR tail path)."""
    r = np.random.default_rng(0)
    n = 48
    A = np.triu(r.standard_normal((n, n)), 1)
    check_philox(product, A + A.T, r.standard_normal(n) * 0.2, S=4)
    check_philox(product, A + A.T, r.standard_normal(n) * 0.2, S=3, precision="f64")
    n = 1100                                       # more levels than the LDS offset table holds: plain level loop
    A = np.triu(np.where(r.random((n, n)) < 0.5, 1.0, -1.0), 1)
    check_philox(product, A + A.T, np.zeros(n), R=1, S=2, beta=0.01)


def test_degree_between_8_and_16_and_above(product):
    """Star-like hubs: rows of 9..16 entries use the second half of the packed window, longer ones the CSR tail."""
    n = 300
    J = sp.lil_matrix((n, n))
    r = np.random.default_rng(5)
    for hub, deg in ((0, 9), (1, 12), (2, 16), (3, 17), (4, 40)):
        for j in r.choice(np.arange(10, n), size=deg, replace=False):
            w = float(r.choice([-1.0, 1.0]) * r.random())
            J[hub, j] = w
            J[j, hub] = w
    check_philox(product, J.tocsr(), r.standard_normal(n) * 0.1, S=6)


def test_zero_sweeps_and_record_stride(product):
    J, h = make_instance(64, seed=1)
    m0 = init_spins(2, 64)
    with product.Engine(J, h, 2) as eng:
        eng.set_spins(m0)
        o = eng.sweep_philox(0, 1, beta=1.0, record_stride=1, want_energy=True)
        assert o["spins"].shape == (2, 0, 64) and o["energy"].shape == (2, 0)
        assert np.array_equal(eng.get_spins(), m0)
        full = eng.sweep_philox(7, 9, beta=1.0, record_stride=1)["spins"]
        eng.set_spins(m0)
        strided = eng.sweep_philox(7, 9, beta=1.0, record_stride=3)["spins"]
        assert strided.shape == (2, 3, 64) and np.array_equal(strided, full[:, ::3])     # M[:, ::M_skip]


def test_size_limits(product):
    n = product._abi.LDS_N
    r = np.random.default_rng(1)
    i = np.arange(n)
    w = r.choice([-1.0, 1.0], n)
    J = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, (i + 1) % n]), np.concatenate([(i + 1) % n, i]))),
                      shape=(n, n)).tocsr()          # ring of the largest size whose spins live in LDS
    csr = oracle.Csr(J)
    m0 = init_spins(1, n)
    with product.Engine(J, np.zeros(n), 1) as eng:
        eng.set_spins(m0)
        E0 = eng.energy()
        for f64 in (False, True):      # (fp64 uniforms of that many spins do not fit in LDS: the global-memory kernels, csrc/nlmc_big.h)
            eng.set_spins(m0)
            o = eng.sweep_philox(2, 4, beta=0.7, record_stride=1, precision="f64" if f64 else "f32")
            cb = np.tile(np.array(oracle.cb_pair(0.7, 1.0, f64)), (2, 1))
            M, _, _ = oracle.sweeps_philox(csr, np.zeros(n), m0[0], cb, 4, 0, escale=eng.energy_scale, use_f64=f64,
                                           efix0=int(np.rint(E0[0] * 2.0 ** eng.energy_scale)))
            assert np.array_equal(o["spins"][0], M)
    n = product._abi.MAX_N + 1
    with pytest.raises(NotImplementedError):
        product.Engine(sp.csr_matrix((n, n)), np.zeros(n), 1)


def test_new_entry_points_degenerate_inputs(product):
    """Fused plan, device backbone inference and the ladder pairing on degenerate inputs: zero windows, a dense graph
    (schedule deeper than the fused level table -> declined, plain path still right), a graph without edges (the
    reference's convergence ratio is 0/0 there: every lambda runs out of iterations -> ValueError), one sub-replica per
    temperature (no pairs), no backbone seeds (no clusters)."""
    SEED = 5
    J, h = make_instance(600, seed=1)
    with product.Engine(J, h, 2) as eng:
        assert eng.plan_philox_fused(0, 0, 10, SEED) == 0
        eng.set_spins(init_spins(2, 600))
        eng.pt_init(np.array([0.5, 1.0]))                         # K = 1 ladder of 2 slots: nothing to pair
        before = eng.get_spins()
        info = eng.icm_round_ladders(0, SEED, True, want_info=True)
        assert info.shape == (0, 2) and np.array_equal(eng.get_spins(), before)
    # dense 300-spin graph: ~300 levels per sweep, 10 sweeps do not fit the 1024-entry fused level table
    rng = np.random.default_rng(0)
    A = rng.normal(size=(300, 300)); A = (A + A.T) / 2; np.fill_diagonal(A, 0.0)
    with product.Engine(A, np.zeros(300), 2) as eng:
        assert eng.plan_philox_fused(0, 2, 10, SEED) == 0         # declined, not an error
    check_philox(product, sp.csr_matrix(A), np.zeros(300), R=2, S=3)
    # no edges at all
    n = 40
    inst = product.Instance(sp.csr_matrix((n, n)), np.linspace(-1, 1, n))
    graph = product.lbp.EdgeGraph(inst)
    with product.Engine(inst, None, 1) as eng:
        with pytest.raises(ValueError, match="LBP diverged at initial lambda"):
            product.lbp.lbp_convexified_device(eng, graph, 0.5, 0.01, 0.9, np.ones((1, n)), graph.epsilon(inst.h),
                                               np.finfo(float).eps, 20, 0.999999, 0.99999, 2.5)
    assert product.lbp.find_clusters(graph, np.zeros(n), 0.999999, 0.99999, 0.01) == []
    assert product.lbp.find_clusters(graph, np.zeros(n), 0.999999, 0.99999, 0.01, flat=True).size == 0


@pytest.mark.parametrize("n,fused", [(300, False), (1500, True)])
def test_energies_of_the_recorded_trace_on_the_device(product, n, fused):
    """nlmc_energy_of_recorded (replica_energy over the first k columns, NPT/npt.py:31-45,685-692, without sending the
    trace back) == nlmc_energy_of of the same configurations == the oracle's energy; refused when nothing was recorded."""
    J, h = make_instance(n, seed=7, with_h=True, gaussian=True)
    csr = oracle.Csr(J)
    R, S = 5, 6
    m0 = init_spins(R, n)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        if fused:
            assert eng.plan_philox_fused(0, 1, S, 99) == 1
        o = eng.sweep_philox(S, 99, beta=np.repeat(np.linspace(0.3, 2.0, R)[:, None], S, axis=1), record_stride=1)
        for first, k in ((0, S), (0, 2), (3, 3)):
            E = eng.energy_of_recorded(k, first)
            assert E.shape == (R, k)
            assert np.array_equal(E, eng.energy_of(o["spins"][:, first:first + k]).reshape(R, k))
        assert abs(E[2, 1] - oracle.energy(csr, h, o["spins"][2, 4])) <= 1e-9 * max(1.0, abs(E[2, 1]))
        with pytest.raises(RuntimeError):
            eng.energy_of_recorded(S + 1)
        eng.sweep_philox(2, 99, sweep0=S, beta=1.0)              # records nothing
        with pytest.raises(RuntimeError):
            eng.energy_of_recorded(1)


def test_stream_rows_that_are_not_permutations_are_refused(product):
    """nlmc_sweep_stream scatters the reference stream by spin on the device (k_stream_scatter) and checks on the way that every
    (chain, sweep) row of `perm` is a permutation of 0..n-1: a repeated or out-of-range entry is a ValueError, and the state is
    untouched; a valid stream right after runs normally."""
    N, R, S = 200, 2, 3
    J, h = make_instance(N, seed=5)
    m0 = init_spins(R, N)
    rng = np.random.default_rng(0)
    perm = np.stack([np.stack([rng.permutation(N) for _ in range(S)]) for _ in range(R)]).astype(np.int32)
    u = rng.random((R, S, N))
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        for bad in ("repeat", "range"):
            p = perm.copy()
            if bad == "repeat":
                p[1, 2, 7] = p[1, 2, 8]
            else:
                p[0, 1, 3] = N
            with pytest.raises(ValueError, match="not a permutation"):
                eng.sweep_stream(p, u, 1.0)
            assert np.array_equal(eng.get_spins(), m0)
        a = eng.sweep_stream(perm, u, 1.0, record_stride=1)["spins"]
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        b = eng.sweep_stream(perm, u, 1.0, record_stride=1)["spins"]
    assert np.array_equal(a, b)
