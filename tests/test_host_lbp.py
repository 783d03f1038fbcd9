"""Host logic (no GPU): the product's edge-list LBP + cluster growth against the reference's golden marginals and
clusters; schedules and phase flags."""
import numpy as np
import pytest

from conftest import golden, golden_names, load_product


@pytest.fixture(scope="module")
def P():
    return load_product()


def inst_of(P, g):
    import scipy.sparse as sp
    A = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=(int(g["N"]),) * 2)
    return P.Instance(A, g["h"])


@pytest.mark.parametrize("name", golden_names("lbp_"))
def test_lbp_matches_reference(P, name):
    g = golden(name)
    inst = inst_of(P, g)
    graph = P.lbp.EdgeGraph(inst)
    eps = graph.epsilon(inst.h)
    assert np.array_equal(eps, np.abs(inst.h) + np.sum(np.abs(inst.csr.toarray()), axis=1))
    cl, marg = P.lbp.lbp_convexified(inst, float(g["lambda_start"]), float(g["lambda_end"]),
                                     float(g["lambda_reduction_factor"]), g["m_star"].astype(float), eps,
                                     np.finfo(float).eps, int(g["max_iterations"]), float(g["threshold_initial"]),
                                     float(g["threshold_cutoff"]), float(g["global_beta"]), graph=graph,
                                     want_marginals=True)
    lam = np.array(sorted(marg.keys(), reverse=True))
    assert np.array_equal(lam, g["lambdas"])
    got = np.array([marg[l] for l in lam])
    assert np.array_equal(got, g["marginals"])          # bit-exact: same association of every sum
    assert np.array_equal(np.array([len(c) for c in cl]), g["cluster_sizes"])
    cat = np.concatenate(cl).astype(np.int64) if cl else np.zeros(0, np.int64)
    assert np.array_equal(cat, g["clusters_concat"])


def test_beta_schedule_quirk(P):
    # NMC/nmc.py:62-67: the index is pre-incremented, so linspace's first value is never used
    s = P.hostlogic.beta_schedule(6, 3.0, True, 1, 0.0)
    assert np.allclose(s, np.linspace(0, 3, 6)[[1, 2, 3, 4, 5, 5]])
    s = P.hostlogic.beta_schedule(7, 2.0, True, 3, 0.5)
    assert np.allclose(s, np.linspace(0.5, 2.0, 2)[[1, 1, 1, 1, 1, 1, 1]])
    assert np.all(P.hostlogic.beta_schedule(4, 1.5) == 1.5)


def test_phase_flags(P):
    m = np.array([1, -1, 1, -1, 1], dtype=np.int8)
    cl = np.array([1, 3])
    assert list(P.hostlogic.phase_flags(5, m, cl, "C")) == [2, 1, 2, 1, 2]
    assert list(P.hostlogic.phase_flags(5, m, cl, "NC")) == [0, 3, 0, 3, 0]
    assert list(P.hostlogic.phase_flags(5, m, cl, "ALL")) == [0] * 5


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_find_clusters_compiled_equals_numpy_statement(P, seed):
    """include/nlmc.h: nlmc_find_clusters (compiled host routine) against the NumPy statement of NMC/nmc.py:257-318,
    with the threshold-lowering loop active (step 0.01 from 0.95 down to 0.80) and inactive (reference defaults)."""
    from helpers import make_instance
    rng = np.random.default_rng(seed)
    n = 400
    J, h = make_instance(n, seed=seed)
    g = P.lbp.EdgeGraph(P.Instance(J, h))
    mag = np.tanh(rng.normal(0, 2.0, n))
    mag[rng.random(n) < 0.2] = 1.0                      # saturated backbone spins
    for t0, tc in ((0.95, 0.80), (0.999999, 0.99999), (0.5, 0.1), (2.0, 1.5)):
        a = P.lbp.find_clusters(g, mag, t0, tc, 0.01)
        b = P.lbp.find_clusters_py(g, mag, t0, tc, 0.01)
        assert len(a) == len(b)
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
