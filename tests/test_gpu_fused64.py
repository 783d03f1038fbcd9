"""The fp64 mode (the reference's arithmetic: fp64 field, 53-bit uniform, NMC/nmc.py:86-87) on fused windows
(k_sweep_fused<.., F64>, include/nlmc.h: nlmc_fused_modes) and the packed fp64 sweep-by-sweep kernel.

Where every coupling and field is an exact multiple of 2^-qs the fp64 field is an exact integer times 2^-qs and the spec's
acceptance test is an integer threshold per field value; the fused kernel must reproduce the sweep-by-sweep fp64 kernel and
the sequential oracle (oracle/nlo.c, use_f64) bit for bit: spins, tracked energies, per-sweep outputs, swap decisions."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from helpers import make_instance, init_spins

pytestmark = pytest.mark.gpu
SEED = 0xA5A50000


def integer_instance(N, seed, wmax=3, hub_degs=(9, 12, 16, 17, 40), diag=False, h_step=0.0):
    """Random degree-6 graph with couplings in +-{1..wmax}, hub rows of the given degrees (second half of the row window, CSR tail),
    optional integer diagonal, fields in multiples of h_step: every value an exact multiple of a power of two."""
    rng = np.random.default_rng(seed)
    Jb, _ = make_instance(N, seed=seed)
    A = sp.lil_matrix(sp.csr_matrix(Jb))
    for hub, deg in enumerate(hub_degs):
        for j in rng.choice(np.arange(64, N), size=deg, replace=False):
            A[hub, j] = A[j, hub] = float(rng.choice([-1.0, 1.0]))
    A = sp.csr_matrix(A)
    if wmax > 1:
        U = sp.triu(A, 1).tocoo()
        w = U.data * rng.integers(1, wmax + 1, U.nnz)
        A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([U.row, U.col]), np.concatenate([U.col, U.row]))), shape=(N, N)).tocsr()
    if diag:
        A = (A + sp.diags(rng.integers(-2, 3, N).astype(float))).tocsr()
    h = rng.integers(-3, 4, N) * h_step if h_step else np.zeros(N)
    A.sort_indices()
    return A, h


def run(product, inst, R, T, W, betas, fused, precision="f64", swaps=0, m0=None, outputs=False):
    with product.Engine(inst, None, R) as eng:
        eng.set_spins(m0)
        E0 = eng.energy()
        eng.pt_init(betas)
        planned = eng.plan_philox_fused(0, W, T, SEED) if fused else 0
        if swaps:
            eng.pt_plan(0, W, SEED, swaps)
        lv, outs = [], []
        for w in range(W):
            kw = dict(record_stride=2, want_energy=True, want_min=True, want_state=True) if outputs else {}
            outs.append(eng.sweep_philox(T, SEED, sweep0=w * T, beta=None, precision=precision, **kw))
            st = eng.last_schedule_stats()
            lv.append(st["levels"] / max(1, st["orders"]))
            if swaps:
                eng.pt_swap_philox(w, SEED, swaps, want_log=False)
        return {"spins": eng.get_spins(), "E": eng.energy(), "slots": eng.pt_slots(), "planned": planned, "lv": lv,
                "esc": eng.energy_scale, "E0": E0, "outs": outs, "modes": eng.fused_modes(T)}


def check_oracle(J, h, m0, betas, chains, res, S, exact_energy=True):
    csr = oracle.Csr(J)
    for c in chains:
        cb = np.tile(np.array(oracle.cb_pair(betas[c], 1.0, True)), (S, 1))
        _, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, SEED, c, escale=res["esc"], use_f64=True,
                                            efix0=int(np.rint(res["E0"][c] * 2.0 ** res["esc"])), want_M=False)
        assert np.array_equal(res["spins"][c], s_fin), f"chain {c}"
        if exact_energy:
            assert res["E"][c] == tr[-1] * 2.0 ** -res["esc"]
        else:         # inexact couplings: the start energy handed to the oracle is itself a rounded fp64 sum
            assert abs(res["E"][c] - oracle.energy(csr, h, s_fin)) <= 1e-9 * max(1.0, abs(res["E"][c]))


def test_fused_f64_equals_plain_f64_and_the_oracle_pmj(product):
    N, R, T, W = 8000, 6, 5, 3
    J, h = make_instance(N, seed=31)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.05, 4.0, R)
    m0 = init_spins(R, N)
    f = run(product, inst, R, T, W, betas, True, m0=m0)
    p = run(product, inst, R, T, W, betas, False, m0=m0)
    assert f["modes"] == {"f32", "f64"} and f["planned"] == W
    assert max(f["lv"]) < min(p["lv"])                              # the fused kernel really ran (fewer levels per sweep)
    assert np.array_equal(f["spins"], p["spins"]) and np.array_equal(f["E"], p["E"])
    check_oracle(J, h, m0, betas, (0, 3, R - 1), f, T * W)
    # (since round 4 the two arithmetic modes draw an update's decision from the SAME Philox word -- the "f32" mode all 32 bits through
    # its logistic threshold, the fp64 mode its 27 high bits, and 26 more from a second call when they do not decide: the two chains
    # differ only where a uniform falls within ~2^-24 of the acceptance probability, which 7 x 10^5 updates need not contain)
    q = run(product, inst, R, T, W, betas, True, precision="f32", m0=m0)
    assert np.mean(q["spins"] != f["spins"]) < 0.01


@pytest.mark.parametrize("case", ["integer_hubs", "diag_fields", "pmj_hubs_swaps"])
def test_fused_f64_formats_long_rows_diagonal_fields_and_swaps(product, case):
    """Compact entries (integer couplings), lane pairs and the CSR tail (hub rows of 9..40 entries), a diagonal, quarter-integer
    fields, replica exchange between the windows: fused == sweep-by-sweep == oracle."""
    N, R, T, W = 3000, 5, 6, 2
    if case == "integer_hubs":
        J, h = integer_instance(N, 3, wmax=3)
    elif case == "diag_fields":
        J, h = integer_instance(N, 4, wmax=2, diag=True, h_step=0.25)
    else:
        J, h = integer_instance(N, 5, wmax=1)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.1, 2.5, R)
    m0 = init_spins(R, N)
    swaps = 2 if case == "pmj_hubs_swaps" else 0
    f = run(product, inst, R, T, W, betas, True, m0=m0, swaps=swaps)
    p = run(product, inst, R, T, W, betas, False, m0=m0, swaps=swaps)
    assert "f64" in f["modes"] and f["planned"] == W and max(f["lv"]) < min(p["lv"])
    assert np.array_equal(f["spins"], p["spins"]) and np.array_equal(f["E"], p["E"]) and np.array_equal(f["slots"], p["slots"])
    if not swaps:
        check_oracle(J, h, m0, betas, (0, R - 1), f, T * W)


def test_fused_f64_exact_path_of_undecided_high_words(product, monkeypatch):
    """The 27 high bits of a uniform decide an update unless they equal the high word of its threshold (2^-27 per update);
    the test knob NLMC_F64_TIE_MASK makes the kernel treat far more updates as undecided, so the exact path (one Philox call
    for the low word, the sweep index recovered from the level) runs thousands of times: same bits."""
    N, R, T, W = 2500, 4, 7, 2
    J, h = integer_instance(N, 11, wmax=2, h_step=0.5)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.2, 3.0, R)
    m0 = init_spins(R, N)
    a = run(product, inst, R, T, W, betas, True, m0=m0)
    monkeypatch.setenv("NLMC_F64_TIE_MASK", "0xFFFF0000")         # undecided whenever the top 11 of 27 bits agree
    b = run(product, inst, R, T, W, betas, True, m0=m0)
    monkeypatch.setenv("NLMC_F64_TIE_MASK", "0")                  # every update takes the exact path
    c = run(product, inst, R, T, W, betas, True, m0=m0)
    assert a["planned"] == b["planned"] == c["planned"] == W
    assert np.array_equal(a["spins"], b["spins"]) and np.array_equal(a["E"], b["E"])
    assert np.array_equal(a["spins"], c["spins"]) and np.array_equal(a["E"], c["E"])
    check_oracle(J, h, m0, betas, (1,), a, T * W)


def test_fused_f64_per_sweep_outputs(product):
    """Energy trace, running minimum + argmin state, recorded configurations of the fp64 mode on fused windows == sweep by sweep."""
    N, R, T, W = 2600, 4, 6, 2
    J, h = integer_instance(N, 21, wmax=2, diag=True)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.3, 2.0, R)
    m0 = init_spins(R, N)
    f = run(product, inst, R, T, W, betas, True, m0=m0, outputs=True)
    p = run(product, inst, R, T, W, betas, False, m0=m0, outputs=True)
    assert f["planned"] == W and max(f["lv"]) < min(p["lv"])
    for of, op in zip(f["outs"], p["outs"]):
        for k in ("spins", "energy", "min_energy", "argmin", "argmin_state"):
            assert np.array_equal(of[k], op[k]), k
    assert np.array_equal(f["spins"], p["spins"])


def test_fp64_mode_on_inexact_couplings_runs_sweep_by_sweep(product):
    """Gaussian couplings are not multiples of a power of two: no fp64 fused windows, an fp64 call inside a plan takes the
    sweep-by-sweep kernel (and is what the oracle says); phase flags switch the fused fp64 kernel off as well."""
    N, R, T = 2000, 3, 5
    J, h = make_instance(N, seed=9, with_h=True, gaussian=True)
    betas = np.geomspace(0.3, 2.0, R)
    m0 = init_spins(R, N)
    g = run(product, product.Instance(J, h), R, T, 1, betas, True, m0=m0)
    assert g["modes"] == {"f32"} and g["planned"] == 1 and g["lv"][0] > 14        # planned (for f32), not used
    check_oracle(J, h, m0, betas, (0, 2), g, T, exact_energy=False)
    J, h = make_instance(N, seed=9)
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        eng.pt_init(betas)
        assert eng.plan_philox_fused(0, 1, T, SEED) == 1 and "f64" in eng.fused_modes(T)
        fl = np.zeros((R, N), np.uint8)
        fl[:, :100] = 1
        eng.set_flags(fl, 20.0)
        eng.sweep_philox(T, SEED, sweep0=0, beta=None, precision="f64")
        st = eng.last_schedule_stats()
        assert st["levels"] / st["orders"] > 14                                   # sweep by sweep


def test_c4_size_f64_fused_oracle_sample_and_energy(product):
    """The workload of bench.py's headline leg (N = 10^4, 256 replicas, fp64 mode on fused windows of 10 sweeps, swap rounds
    in between): tracked == recomputed energies for every replica, hot / middle / cold replica of the first window against the
    sequential fp64 oracle bit for bit, and the sweep-by-sweep fp64 kernel (16-entry packed window, 8 waves) on the same
    window gives the same bits for all 256."""
    N, R, T, PAIRS = 10_000, 256, 10, 77
    J, h = make_instance(N)
    betas = np.geomspace(0.05, 4.0, R)
    m0 = init_spins(R, N)
    inst = product.Instance(J, h)
    f = run(product, inst, R, T, 1, betas, True, m0=m0)
    assert f["planned"] == 1 and f["lv"][0] < 18.5
    with product.Engine(inst, None, R) as eng:
        assert np.array_equal(f["E"], eng.energy_of(f["spins"]))
    check_oracle(J, h, m0, betas, (0, 131, 255), f, T)
    p = run(product, inst, R, T, 1, betas, False, m0=m0)
    assert p["lv"][0] > 18.5
    assert np.array_equal(f["spins"], p["spins"]) and np.array_equal(f["E"], p["E"])
    a = run(product, inst, R, T, 3, betas, True, m0=m0, swaps=PAIRS)
    b = run(product, inst, R, T, 3, betas, True, m0=m0, swaps=PAIRS)
    assert np.array_equal(a["spins"], b["spins"]) and np.array_equal(a["slots"], b["slots"])
    assert not np.array_equal(a["slots"], np.arange(R))
    with product.Engine(inst, None, R) as eng:
        assert np.array_equal(a["E"], eng.energy_of(a["spins"]))


# ---- the packed fp64 sweep-by-sweep kernel (k_sweep_philox<double, DIAG, true>) ----------------------------------
def sweep_by_sweep_f64(product, J, h, R, S, beta, seed=321):
    m0 = init_spins(R, J.shape[0])
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        E0, esc = eng.energy(), eng.energy_scale
        o = eng.sweep_philox(S, seed, beta=beta, precision="f64", record_stride=1, want_energy=True)
    return m0, E0, esc, o


@pytest.mark.parametrize("case", ["hubs", "diag", "dense"])
def test_packed_f64_window_integer_weights(product, case, monkeypatch):
    """Integer couplings switch the 16-bit packed schedule window of the fp64 kernel on (every J = Jq 2^-qs exactly): rows of 9,
    12, 16, 17 and 40 entries (second half of the window, CSR tail), a diagonal (the DIAG variant), a dense graph (deep schedule,
    every row a tail) -- against the oracle, and against the unpacked window (NLMC_NO_PACK64): identical bits."""
    r = np.random.default_rng(7)
    if case == "hubs":
        J, h = integer_instance(300, 2, wmax=3, h_step=0.5)
    elif case == "diag":
        J, h = integer_instance(240, 6, wmax=2, hub_degs=(9, 17), diag=True)
    else:
        n = 60
        A = np.triu(r.integers(-3, 4, (n, n)).astype(float), 1)
        J, h = sp.csr_matrix(A + A.T), r.integers(-2, 3, n) * 0.5
    R, S, beta, seed = 3, 5, 0.9, 321
    m0, E0, esc, o = sweep_by_sweep_f64(product, J, h, R, S, beta, seed)
    csr = oracle.Csr(J)
    for c in range(R):
        cb = np.tile(np.array(oracle.cb_pair(beta, 1.0, True)), (S, 1))
        M, _, tr = oracle.sweeps_philox(csr, h, m0[c], cb, seed, c, escale=esc, use_f64=True, efix0=int(np.rint(E0[c] * 2.0 ** esc)))
        assert np.array_equal(o["spins"][c], M)
        assert np.array_equal(o["energy"][c], tr * 2.0 ** -esc)
    monkeypatch.setenv("NLMC_NO_PACK64", "1")
    _, _, _, o2 = sweep_by_sweep_f64(product, J, h, R, S, beta, seed)
    assert np.array_equal(o["spins"], o2["spins"]) and np.array_equal(o["energy"], o2["energy"])
