"""Chains too long for LDS (csrc/nlmc_big.h; the reference takes any N, NMC/nmc.py:49-53): the global-memory kernels against
the oracle at sizes only they take (N = 30 000 and 70 000: spin indices past 16 bits) and, forced on at a small size (the knob
NLMC_FORCE_BIG, read when a context is created), against the LDS kernels on everything a sweep call can return.

Bars as in test_gpu_sweep.py: spins bit-exact, philox-mode energies bit-exact, stream-mode energies within 1e-10."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from helpers import make_instance, init_spins, draw_stream

pytestmark = pytest.mark.gpu
E_RTOL = 1e-10


def assert_energy(a, b, rtol=E_RTOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.all(np.abs(a - b) <= rtol * np.maximum(1.0, np.abs(b))), np.max(np.abs(a - b))


def big_instance(n, seed, gaussian=False, diag=False):
    """ring + random chords (mean degree ~6), a few hub rows, optionally self-couplings"""
    r = np.random.default_rng(seed)
    i = np.concatenate([np.arange(n), r.integers(0, n, 2 * n), np.full(40, 7), np.full(25, n - 3)])
    j = np.concatenate([(np.arange(n) + 1) % n, r.integers(0, n, 2 * n), r.integers(0, n, 40), r.integers(0, n, 25)])
    keep = i != j
    A = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(n, n)).tocsr()
    A = sp.triu(((A + A.T) > 0).astype(np.float64), 1).tocsr()
    A.data = r.normal(0, 1, A.nnz) if gaussian else r.choice([-1.0, 1.0], A.nnz)
    A = (A + A.T).tolil()
    if diag:
        for k in r.choice(n, 30, replace=False):
            A[k, k] = r.normal() if gaussian else r.choice([-1.0, 1.0])
    A = A.tocsr()
    A.sort_indices()
    h = r.normal(0, 0.3, n) if gaussian else r.integers(-1, 2, n).astype(float)
    return A, h


@pytest.mark.parametrize("n,gaussian,diag", [(30_000, True, True), (70_000, False, False)])
def test_stream_mode_of_long_chains_matches_the_oracle(product, monkeypatch, n, gaussian, diag):
    monkeypatch.setenv("NLMC_BIG_PER_LEVEL", "1" if diag else "0")      # (3 chains would take one launch per level by themselves)
    J, h = big_instance(n, 11, gaussian, diag)
    R, S = 3, 3
    csr = oracle.Csr(J)
    m0 = init_spins(R, n)
    np.random.seed(5)
    perm, u = draw_stream(R, S, n)
    betas = np.array([0.4, 1.1, 2.5])
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        assert_energy(eng.energy(), [oracle.energy(csr, h, m0[c].astype(float)) for c in range(R)])
        o = eng.sweep_stream(perm, u, np.repeat(betas[:, None], S, axis=1), record_stride=1, want_energy=True, want_min=True,
                             want_state=True)
        final = eng.get_spins()
        assert_energy(eng.energy_tracked(), eng.energy())
        bad = perm.copy()
        bad[1, 2, 5] = bad[1, 2, 6]
        with pytest.raises(ValueError):
            eng.sweep_stream(bad, u, np.repeat(betas[:, None], S, axis=1))
    for c in range(R):
        M, _ = oracle.sweeps_stream(csr, h, m0[c].astype(float), np.full(S, betas[c]), perm[c], u[c])
        assert np.array_equal(o["spins"][c], M), c
        assert np.array_equal(final[c], M[-1])
        E = np.array([oracle.energy(csr, h, M[t]) for t in range(S)])
        assert_energy(o["energy"][c], E)
        assert o["argmin"][c] == int(np.argmin(o["energy"][c])) and np.array_equal(o["argmin_state"][c], M[o["argmin"][c]])


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("order", ["shared", "per_chain"])
def test_philox_modes_of_long_chains_match_the_oracle(product, monkeypatch, precision, order):
    monkeypatch.setenv("NLMC_BIG_PER_LEVEL", "1" if precision == "f64" else "0")
    n = 30_000 if order == "shared" else 66_000
    J, h = big_instance(n, 3, gaussian=(precision == "f64"), diag=(order == "shared"))
    R, S = 3, 3
    csr = oracle.Csr(J)
    m0 = init_spins(R, n)
    betas = np.array([0.3, 1.0, 2.2])
    seed, sweep0, f64 = 0x1234 + (5 << 32), 9, precision == "f64"
    flags = np.zeros((R, n), np.uint8)
    cl = np.random.default_rng(1).random(n) < 0.1
    flags[1, cl] = 1
    flags[1, ~cl] = np.where(m0[1, ~cl] > 0, 2, 3)
    with product.Engine(J, h, R, chain_base=2, n_chains_global=R + 2) as eng:
        eng.set_spins(m0)
        eng.set_flags(flags, 20.0)
        E0 = eng.energy()
        esc = eng.energy_scale
        o = eng.sweep_philox(S, seed, sweep0=sweep0, beta=np.repeat(betas[:, None], S, axis=1), precision=precision, order=order,
                             record_stride=1, want_energy=True, want_min=True, want_state=True)
        # the same sweeps from a cached plan, in two calls
        eng.set_spins(m0)
        if order == "shared":
            eng.plan_philox(sweep0, S, seed, precision=precision)
        eng.sweep_philox(1, seed, sweep0=sweep0, beta=betas[:, None], precision=precision, order=order)
        eng.sweep_philox(S - 1, seed, sweep0=sweep0 + 1, beta=np.repeat(betas[:, None], S - 1, axis=1), precision=precision, order=order)
        again = eng.get_spins()
    for c in range(R):
        gc = c + 2
        cb = np.tile(np.array(oracle.cb_pair(betas[c], 20.0, f64)), (S, 1))
        M, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, seed, gc, order_group=(gc + 1 if order == "per_chain" else 0),
                                            sweep0=sweep0, flags=flags[c], escale=esc, use_f64=f64, efix0=int(np.rint(E0[c] * 2.0 ** esc)))
        assert np.array_equal(o["spins"][c], M), c
        assert np.array_equal(again[c], s_fin)
        assert np.array_equal(o["energy"][c], tr.astype(np.float64) * 2.0 ** -esc)
        assert o["argmin"][c] == int(np.argmin(tr)) and np.array_equal(o["argmin_state"][c], M[int(np.argmin(tr))])


def test_global_memory_kernels_equal_the_lds_kernels_at_a_size_both_take(product, monkeypatch):
    """NLMC_FORCE_BIG: stream and both philox modes, per-chain temperatures, phase flags, a self-coupling, recorded
    configurations with a stride, energy trace, running minimum with its state, a replica-exchange ladder in between."""
    J, h = make_instance(700, seed=8, with_h=True, gaussian=True)
    J = J.tolil(); J[5, 5] = 0.5; J[40, 40] = -1.25; J = J.tocsr(); J.sort_indices()
    R, S, N = 6, 7, 700
    m0 = init_spins(R, N)
    np.random.seed(3)
    perm, u = draw_stream(R, S, N)
    beta = np.repeat(np.geomspace(0.2, 3.0, R)[:, None], S, axis=1) * np.linspace(0.6, 1.0, S)[None, :]
    flags = np.random.default_rng(2).choice([0, 0, 0, 1, 2, 3], size=(R, N)).astype(np.uint8)

    def run():
        out = []
        with product.Engine(J, h, R) as eng:
            eng.set_spins(m0)
            out.append(eng.energy())
            out.append(eng.sweep_stream(perm, u, beta, record_stride=2, want_energy=True, want_min=True, want_state=True))
            eng.set_flags(flags, 7.0)
            out.append(eng.sweep_stream(perm, u, beta, record_stride=1, want_energy=True))
            for prec in ("f32", "f64"):
                for order in ("shared", "per_chain"):
                    out.append(eng.sweep_philox(S, 17, sweep0=3, beta=beta, precision=prec, order=order, record_stride=3,
                                                want_energy=True, want_min=True, want_state=True))
            eng.set_flags(np.zeros((R, N), np.uint8), 1.0)
            eng.pt_init(np.geomspace(0.2, 3.0, R))
            for rnd in range(4):
                out.append(eng.sweep_philox(3, 17, sweep0=100 + 3 * rnd, beta=None, precision="f64", want_energy=True))
                out.append(eng.pt_swap_philox(rnd, 17, 2))
            out.append(eng.pt_slots())
            out.append(eng.get_spins())
            out.append(eng.energy_tracked())
            out.append(eng.energy_of(m0))
        return out

    monkeypatch.delenv("NLMC_FORCE_BIG", raising=False)
    a = run()
    monkeypatch.setenv("NLMC_FORCE_BIG", "1")
    monkeypatch.setenv("NLMC_BIG_PER_LEVEL", "0")                 # one workgroup per chain
    b = run()
    monkeypatch.setenv("NLMC_BIG_PER_LEVEL", "1")                 # one launch per level
    b2 = run()

    def same(x, y):
        if isinstance(x, dict):
            return all(same(x[k], y[k]) for k in x)
        if isinstance(x, (tuple, list)):
            return len(x) == len(y) and all(same(p, q) for p, q in zip(x, y))
        return np.array_equal(np.asarray(x), np.asarray(y))
    for i, (x, y, z) in enumerate(zip(a, b, b2)):
        assert same(x, y) and same(x, z), i


def test_drop_in_classes_on_a_long_chain(product):
    """NMC.MCMC and NPT.run at N = 26 000 (past the LDS kernels) in the default numpy mode against the oracle's restatement of
    the reference classes on the same random streams."""
    from oracle import refport
    n = 26_000
    J, h = big_instance(n, 21)
    Jd = J.tocsr()
    np.random.seed(4)
    m0 = np.sign(np.random.rand(n) - 0.5)
    obj = product.NMC(Jd, h.copy())
    np.random.seed(9)
    M = obj.MCMC(3, m0.copy(), 1.3, Jd, h)
    np.random.seed(9)
    Mr = refport.mcmc(3, m0.copy(), 1.3, oracle.Csr(J), h)
    assert np.array_equal(np.asarray(M), Mr)

    import contextlib, io, random
    R, betas = 4, np.geomspace(0.3, 2.0, 4)
    args = (betas, R, [False] * R, 8, 8, 4, 1, 1, 1, 1, 20, 2.5, 3, 0.01, 0.9, 0.9999999, 0.999999, 100, np.finfo(float).eps)
    res = []
    for cls in (product.NPT, refport.RefNPT):
        np.random.seed(31)
        random.seed(31)
        obj = cls(Jd, h.copy())
        with contextlib.redirect_stdout(io.StringIO()):
            Mx, Ex = obj.run(*args)
        res.append((np.asarray(Mx), np.asarray(Ex), np.asarray(obj.swap_pairs), np.asarray(obj.swap_accepted)))
    assert res[0][0].shape == (R * n, 2) and np.array_equal(res[0][0], res[1][0])
    assert_energy(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])

    # APT_ICM.run (Houdayer moves between two sub-replicas per temperature): default numpy mode against the oracle's restatement,
    # then the device-resident mode (rng="philox", icm_feedback=True) for its invariants
    res = []
    for cls in (product.APT_ICM, refport.RefAPT_ICM):
        np.random.seed(32)
        random.seed(32)
        obj = cls(Jd, h.copy())
        with contextlib.redirect_stdout(io.StringIO()):
            Mx, Ex = obj.run(betas, R, 8, 8, 4, 1)
        res.append((np.asarray(Mx), np.asarray(Ex), np.asarray(obj.swap_pairs), np.asarray(obj.swap_accepted)))
    assert np.array_equal(res[0][0], res[1][0])
    assert_energy(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    obj = product.APT_ICM(Jd, h.copy(), rng="philox", seed=5)
    with contextlib.redirect_stdout(io.StringIO()):
        Mx, Ex = obj.run(betas, R, 8, 8, 4, 1, icm_feedback=True, return_trace="int8")
    assert np.asarray(Mx).shape[0] == R * n and set(np.unique(Mx)) <= {-1, 1}
    assert len(obj.icm_cluster_sizes) > 0 and np.all(obj.icm_cluster_sizes >= 0)
    assert np.asarray(Ex).shape == (R,) and np.all(np.isfinite(Ex))


def test_houdayer_move_and_backbone_mask_on_long_chains(product, monkeypatch):
    """The iso-cluster kernels and the cluster-mask kernel with their work arrays in global memory: components against the
    oracle's search at N = 30 000, the device-decided round's conservation laws there, and everything against the LDS kernels at
    a size both take (NLMC_FORCE_BIG)."""
    n = 30_000
    J, h = big_instance(n, 5)
    csr = oracle.Csr(J)
    m0 = init_spins(4, n)
    with product.Engine(J, h, 4) as eng:
        eng.set_spins(m0)
        eng.sweep_philox(3, 5, beta=0.9)                      # correlated pairs: many small clusters, one large
        s = eng.get_spins()
        nc = eng.icm_components(0, 1)
        lab = eng.icm_labels()
        cl = oracle.clusters(csr, s[0].astype(float), s[1].astype(float))
        assert nc == len(cl)
        want = np.full(n, -1, np.int64)
        for members in cl:
            want[np.asarray(members)] = int(np.min(members))
        assert np.array_equal(lab, want)
        pick = 7 % nc
        ncomp, size = eng.icm_move(0, 1, pick, True)
        roots = sorted(int(np.min(m)) for m in cl)
        members = np.flatnonzero(want == roots[pick])
        exp = s.copy()
        if size > n // 2:
            exp[0] = -exp[0]
        else:
            exp[0, members], exp[1, members] = s[1, members], s[0, members]
        assert size == len(members) and np.array_equal(eng.get_spins(), exp)
        # device-decided round on the other pair + the first: energies stay tracked exactly, spins off the clusters untouched
        before = eng.get_spins()
        eng.energy()
        info = eng.icm_round_philox([[2, 3], [0, 1]], 4, 99, True, want_info=True)
        after = eng.get_spins()
        assert np.all(info[:, 0] > 0)
        assert np.array_equal(eng.energy_tracked(), eng.energy())
        for a, b in ((2, 3), (0, 1)):
            agree = before[a] == before[b]
            moved = after[a] != before[a]
            if not np.array_equal(after[a], -before[a]):       # (not the global flip of the Katzgraber variant)
                assert not np.any(moved & agree) and np.array_equal(after[a][moved], before[b][moved])
                assert np.array_equal(after[b][moved], before[a][moved])

    J2, h2 = make_instance(900, seed=12)
    m2 = init_spins(8, 900)
    inst2 = product.Instance(J2, h2)
    eps2 = product.lbp.EdgeGraph(inst2).epsilon(inst2.h)
    lams2 = product.lbp.lambda_list(3.0, 0.05, 0.8)
    thr2 = [0.9999 - 0.01 * i for i in range(3)]

    def run():
        out = []
        with product.Engine(J2, h2, 8) as eng:
            eng.set_spins(m2)
            eng.pt_init(np.geomspace(0.3, 2.0, 4))            # 2 ladders of 4 slots
            eng.sweep_philox(4, 3, beta=None)
            eng.energy()
            out.append(eng.icm_round_philox([[0, 4], [1, 5], [2, 6]], 1, 77, True, want_info=True))
            out.append(eng.icm_round_ladders(2, 77, False, want_info=True))
            out.append(eng.get_spins()); out.append(eng.energy_tracked())
            out.append(eng.icm_components(3, 7)); out.append(eng.icm_labels())
            eng.track_minimum(True)
            eng.sweep_philox(3, 3, sweep0=10, beta=None)
            eng.backbone_clusters(eps2, lams2, 3.0, np.finfo(float).eps, 100, float(np.tanh(19.06)) - np.finfo(float).eps, thr2)
            eng.backbone_check()
            out.append(eng.cluster_mask())
            assert 0 < out[-1].sum() < out[-1].size
        return out

    monkeypatch.delenv("NLMC_FORCE_BIG", raising=False)
    a = run()
    monkeypatch.setenv("NLMC_FORCE_BIG", "1")
    b = run()
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(np.asarray(x), np.asarray(y)), i


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_nmc_run_on_a_long_chain(product, rng):
    """NMC.run at N = 26 000 (the reference's dense message arrays of N x N doubles would be 5.4 GB each there: no oracle run) --
    the trace it returns is consistent: energies are those of its columns, the minimum is the minimum."""
    import contextlib, io
    n = 26_000
    J, _ = big_instance(n, 23)
    h = np.zeros(n)
    csr = oracle.Csr(J)
    obj = product.NMC(J, h.copy()) if rng == "numpy" else product.NMC(J, h.copy(), rng="philox", seed=3)
    np.random.seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        M, E, emin = obj.run(6, 4, 1, 1, 1, 20, 3.0, 3.0, 0.05, 0.8, 0.9999, 0.97, 100, np.finfo(float).eps)
    M, E = np.asarray(M), np.asarray(E).reshape(-1)
    assert M.shape == (n, 3 * 4) and E.shape == (M.shape[1],) and set(np.unique(M)) <= {-1.0, 1.0}
    for t in (0, 3, 4, 8, M.shape[1] - 1):
        assert_energy(E[t], oracle.energy(csr, h, M[:, t]))
    assert_energy(emin, E.min())


@pytest.mark.parametrize("per_level", ["0", "1"])
def test_deepest_schedule_of_a_long_chain(product, monkeypatch, per_level):
    """A ring visited in index order: every spin waits for its predecessor, the schedule has as many levels as spins (30 000: the
    levelizer's passes, its histogram bins past the LDS ones, and a level per launch / per barrier with one spin in it)."""
    monkeypatch.setenv("NLMC_BIG_PER_LEVEL", per_level)
    n = 30_000
    i = np.arange(n)
    w = np.random.default_rng(2).choice([-1.0, 1.0], n)
    J = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, (i + 1) % n]), np.concatenate([(i + 1) % n, i]))), shape=(n, n)).tocsr()
    J.sort_indices()
    h = np.zeros(n)
    csr = oracle.Csr(J)
    m0 = init_spins(1, n)
    perm = np.arange(n, dtype=np.int32)[None, None, :]
    u = np.random.default_rng(3).random((1, 1, n))
    with product.Engine(J, h, 1) as eng:
        eng.set_spins(m0)
        o = eng.sweep_stream(perm, u, np.array([[0.8]]), record_stride=1, want_energy=True)
        assert eng.last_schedule_stats()["levels"] == n
    M, _ = oracle.sweeps_stream(csr, h, m0[0].astype(float), np.array([0.8]), perm[0], u[0])
    assert np.array_equal(o["spins"][0], M)
    assert_energy(o["energy"][0], [oracle.energy(csr, h, M[0])])
