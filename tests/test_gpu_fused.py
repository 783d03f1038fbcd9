"""Fused-window schedule (include/nlmc.h: nlmc_plan_philox_fused; k_levelize_fused / k_sweep_fused).

It is a re-ordering of independent updates only, so the bar is bit-equality: with the sequential oracle (spins and
tracked fixed-point energies) and with the sweep-by-sweep kernel on the same seeds -- plain ladder, NMC phase flags,
an instance with a diagonal, rows longer than the 16-entry packed window, swaps between windows, sharded contexts."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from helpers import make_instance, init_spins

pytestmark = pytest.mark.gpu
SEED = 0xA5A50000


def run_windows(product, inst, R, T, W, betas, fused, flags=None, temp_x=1.0, swaps=0, m0=None, base=0, G=None):
    with product.Engine(inst, None, R, chain_base=base, n_chains_global=G) as eng:
        eng.set_spins(init_spins(R, inst.n, base=1000 + base) if m0 is None else m0)
        eng.pt_init(betas)
        if flags is not None:
            eng.set_flags(flags, temp_x)
        planned = eng.plan_philox_fused(0, W, T, SEED) if fused else 0
        if swaps:
            eng.pt_plan(0, W, SEED, swaps)
        lv = []
        for w in range(W):
            eng.sweep_philox(T, SEED, sweep0=w * T, beta=None)
            st = eng.last_schedule_stats()
            lv.append(st["levels"] / max(1, st["orders"]))
            if swaps:
                eng.pt_swap_philox(w, SEED, swaps, want_log=False)
        return eng.get_spins(), eng.energy(), eng.pt_slots(), planned, lv, eng.energy_scale


def test_fused_equals_oracle_and_plain(product):
    N, R, T, W = 8000, 6, 5, 3
    J, h = make_instance(N, seed=31)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.1, 3.0, R)
    m0 = init_spins(R, N)
    sf, ef, _, planned, lvf, esc = run_windows(product, inst, R, T, W, betas, True, m0=m0)
    sp_, ep, _, _, lvp, _ = run_windows(product, inst, R, T, W, betas, False, m0=m0)
    assert planned == W
    assert max(lvf) < min(lvp)                         # the fused path really ran (fewer levels per sweep)
    assert np.array_equal(sf, sp_) and np.array_equal(ef, ep)
    csr = oracle.Csr(J)
    for c in (0, R - 1):
        cb = np.tile(np.array(oracle.cb_pair(betas[c])), (T * W, 1))
        e0 = int(np.rint(oracle.energy(csr, h, m0[c]) * 2.0 ** esc))
        _, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, SEED, c, escale=esc, efix0=e0, want_M=False)
        assert np.array_equal(sf[c], s_fin)
        assert ef[c] == tr[-1] * 2.0 ** -esc


def test_fused_with_phase_flags_diagonal_and_long_rows(product):
    """Gaussian couplings (the fp64 energy fallback and non-trivial float sums), a non-zero diagonal (DIAG kernel),
    one hub row with 40 neighbours (CSR tail beyond the packed window), cluster / frozen flags with temp_x."""
    N, R, T, W = 9000, 4, 4, 2
    rng = np.random.default_rng(5)
    Jb, h = make_instance(N, seed=8)
    A = sp.lil_matrix(sp.csr_matrix(Jb).multiply(1.0))
    hub = 17
    for j in rng.choice(np.arange(100, N), 40, replace=False):
        A[hub, j] = A[j, hub] = rng.normal()
    A = sp.csr_matrix(A)
    A.data = A.data * rng.normal(1.0, 0.3, A.nnz)
    A = ((A + A.T) * 0.5).tocsr()
    A.setdiag(rng.normal(0, 0.2, N))
    A = A.tocsr()
    hv = rng.normal(0, 0.1, N)
    inst = product.Instance(A, hv)
    betas = np.geomspace(0.3, 2.0, R)
    flags = np.zeros((R, N), np.uint8)
    flags[:, :N // 10] = 1                              # scaled rows
    flags[1, N // 2:] = 2
    flags[2, N // 2:] = 3                              # frozen +-
    a = run_windows(product, inst, R, T, W, betas, True, flags=flags, temp_x=20.0)
    b = run_windows(product, inst, R, T, W, betas, False, flags=flags, temp_x=20.0)
    assert a[3] == W and max(a[4]) < min(b[4])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_fused_with_swaps_and_sharded_contexts(product):
    """Replica exchange between windows; two half-size contexts with global chain ids give the same bits (weak-scaling
    invariant of bench.py), fused and plain."""
    N, G, T, W = 8192, 8, 6, 3
    J, h = make_instance(N, seed=12)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.05, 4.0, G)
    m0 = init_spins(G, N)
    full_f = run_windows(product, inst, G, T, W, betas, True, swaps=2, m0=m0)
    full_p = run_windows(product, inst, G, T, W, betas, False, swaps=2, m0=m0)
    assert full_f[3] == W
    assert np.array_equal(full_f[0], full_p[0]) and np.array_equal(full_f[1], full_p[1]) and np.array_equal(full_f[2], full_p[2])
    # sharded, no swaps in between (a swap needs the all-gather): first window only
    one_f = run_windows(product, inst, G, T, 1, betas, True, m0=m0)
    lo = run_windows(product, inst, G // 2, T, 1, betas, True, m0=m0[:G // 2], base=0, G=G)
    hi = run_windows(product, inst, G // 2, T, 1, betas, True, m0=m0[G // 2:], base=G // 2, G=G)
    assert np.array_equal(one_f[0], np.concatenate([lo[0], hi[0]]))


def test_fused_plan_declines_what_it_cannot_run(product, monkeypatch):
    """Tiny instance (n < 256), window of 2: nothing planned, calls take the plain path; per-chain orders and f64
    ignore a fused plan; calls with per-sweep outputs run on it (snapshot slots in LDS, or in a global ring for large n)."""
    J, h = make_instance(200, seed=3)
    with product.Engine(J, h, 2) as eng:
        assert eng.plan_philox_fused(0, 4, 10, SEED) == 0
    N, T = 8000, 5
    J, h = make_instance(N, seed=31)
    with product.Engine(J, h, 2) as eng:
        assert eng.plan_philox_fused(0, 2, 2, SEED) == 0
        assert eng.plan_philox_fused(0, 2, T, SEED) == 2
        eng.set_spins(init_spins(2, N))
        monkeypatch.setenv("NLMC_NO_FUSED_OUT", "1")
        o = eng.sweep_philox(T, SEED, sweep0=0, beta=1.0, want_energy=True, want_min=True)     # sweep by sweep
        st = eng.last_schedule_stats()
        assert o["energy"].shape == (2, T) and st["orders"] == T and st["levels"] / T > 15
        e_plain = eng.energy()
        monkeypatch.delenv("NLMC_NO_FUSED_OUT")
        eng.set_spins(init_spins(2, N))
        o2 = eng.sweep_philox(T, SEED, sweep0=0, beta=1.0, want_energy=True, want_min=True)    # fused, with outputs
        assert eng.last_schedule_stats()["levels"] / T < st["levels"] / T
        assert np.array_equal(o2["energy"], o["energy"]) and np.array_equal(o2["min_energy"], o["min_energy"])
        eng.set_spins(init_spins(2, N))
        eng.sweep_philox(T, SEED, sweep0=0, beta=1.0)                                           # fused, no outputs
        assert eng.last_schedule_stats()["levels"] / T < st["levels"] / T
        assert np.array_equal(eng.energy(), e_plain)
        eng.set_spins(init_spins(2, N))
        eng.sweep_philox(T, SEED, sweep0=0, beta=1.0, order="per_chain")                        # not what was planned
        assert eng.last_schedule_stats()["orders"] == 2 * T
    N = 10000                      # the snapshot slots do not fit in LDS: they live in a global ring, the call stays fused
    J, h = make_instance(N, seed=32)
    with product.Engine(J, h, 2) as eng:
        assert eng.plan_philox_fused(0, 1, T, SEED) == 1
        eng.set_spins(init_spins(2, N))
        eng.sweep_philox(T, SEED, sweep0=0, beta=1.0, want_min=True)
        lv_out = eng.last_schedule_stats()
        eng.sweep_philox(T, SEED, sweep0=0, beta=1.0)
        assert eng.last_schedule_stats() == lv_out and lv_out["orders"] == T


@pytest.mark.parametrize("N", [300, 1000, 1600, 3000, 5000])
def test_fused_small_workgroups(product, N):
    """Sweep workgroups of 4 / 6 / 10 waves (3 / 5 / 9 workers + one helper wave): same bits as the plain path and as
    the sequential oracle."""
    R, T, W = 5, 7, 2
    J, h = make_instance(N, seed=N)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.2, 2.5, R)
    m0 = init_spins(R, N)
    f = run_windows(product, inst, R, T, W, betas, True, m0=m0)
    p = run_windows(product, inst, R, T, W, betas, False, m0=m0)
    assert f[3] == W and max(f[4]) < min(p[4])
    assert np.array_equal(f[0], p[0]) and np.array_equal(f[1], p[1])
    csr = oracle.Csr(J)
    c = R - 1
    cb = np.tile(np.array(oracle.cb_pair(betas[c])), (T * W, 1))
    e0 = int(np.rint(oracle.energy(csr, h, m0[c]) * 2.0 ** f[5]))
    _, s_fin, tr = oracle.sweeps_philox(csr, h, m0[c], cb, SEED, c, escale=f[5], efix0=e0, want_M=False)
    assert np.array_equal(f[0][c], s_fin) and f[1][c] == tr[-1] * 2.0 ** -f[5]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["pmj", "gauss_flags", "anneal", "pmj_10000_flags", "gauss_10000"])
def test_fused_windows_with_per_sweep_outputs_equal_the_sweep_by_sweep_kernel(product, case, monkeypatch):
    """Calls that want the energy trace, the running minimum + argmin state (NMC/nmc.py:386-395) or recorded
    configurations, or that anneal (one temperature per sweep, NMC/nmc.py:56-69), run on the fused windows too
    (k_sweep_fused<.., OUT>): every output must equal the sweep-by-sweep kernel's bit for bit, over several windows per
    call, with phase flags, and the tracked state must carry on correctly into the next call."""
    from helpers import make_instance, init_spins
    N, R, T, W = 700, 5, 6, 4
    big = "10000" in case                 # n = 10^4: the snapshot slots do not fit in LDS beside the threshold tables and live in
    if big:                               # a global ring (argmin hand-off at C4 / C5 size on fused windows)
        N, R, T, W = 10000, 3, 5, 2
    gauss = case.startswith("gauss")
    J, h = make_instance(N, seed=41, with_h=gauss, gaussian=gauss)
    m0 = init_spins(R, N)
    S = T * W
    if case == "anneal":
        beta = np.stack([np.linspace(0.1, 2.5 + 0.1 * c, S) for c in range(R)])
    else:
        beta = np.repeat(np.geomspace(0.3, 2.5, R)[:, None], S, axis=1)
    flags = None
    if case in ("gauss_flags", "pmj_10000_flags"):
        rng = np.random.default_rng(3)
        flags = rng.choice([0, 0, 0, 1, 2, 3], size=(R, N)).astype(np.uint8)

    def go(fused_out):
        if fused_out:
            monkeypatch.delenv("NLMC_NO_FUSED_OUT", raising=False)
        else:
            monkeypatch.setenv("NLMC_NO_FUSED_OUT", "1")
        with product.Engine(J, h, R) as eng:
            eng.set_spins(m0)
            if flags is not None:
                eng.set_flags(flags, 7.0)
            assert eng.plan_philox_fused(0, 2 * W, T, SEED) == 2 * W
            o1 = eng.sweep_philox(S, SEED, sweep0=0, beta=beta, record_stride=2, want_energy=True, want_min=True, want_state=True)
            lv = eng.last_schedule_stats()
            o2 = eng.sweep_philox(S, SEED, sweep0=S, beta=beta[:, ::-1].copy(), want_min=True, want_state=True)
            return o1, o2, eng.get_spins(), eng.energy_tracked(), lv
    a1, a2, sa, ea, lva = go(True)
    b1, b2, sb, eb, lvb = go(False)
    assert lva["orders"] == T and lvb["orders"] != T                      # the first really ran on fused windows
    for k in ("spins", "energy", "min_energy", "argmin", "argmin_state"):
        assert np.array_equal(a1[k], b1[k]), k
    for k in ("min_energy", "argmin", "argmin_state"):
        assert np.array_equal(a2[k], b2[k]), k
    assert np.array_equal(sa, sb) and np.array_equal(ea, eb)


@pytest.mark.gpu
@pytest.mark.parametrize("weights", ["pmj", "int3", "gauss"])
def test_fused_lane_pairs_when_most_rows_are_long(product, weights):
    """Rows longer than 8 entries take an even / odd lane PAIR of the schedule (entries 0-7 / 8-15), rows longer than 16
    read the rest from the CSR arrays: a graph where most rows are long (mean degree 12, hubs of 17 ... 70 neighbours,
    a few spins without any), all three schedule entry formats -- 16-bit LDS addresses (+-J with integer fields h),
    4-byte entries (couplings in +-{1,2,3}), 8-byte entries (Gaussian) -- fused windows, with and without
    per-sweep outputs, against the sweep-by-sweep kernel and the sequential oracle, bit for bit."""
    N, R, T, W = 1500, 3, 5, 2
    rng = np.random.default_rng(77)
    i = rng.integers(20, N, size=6 * N); j = rng.integers(20, N, size=6 * N)
    hubs = [(1, 17), (2, 24), (3, 33), (4, 70)]
    for hub, d in hubs:
        nb = rng.choice(np.arange(20, N), d, replace=False)
        i = np.concatenate([i, np.full(d, hub)]); j = np.concatenate([j, nb])
    keep = i != j
    A = sp.coo_matrix((np.ones(keep.sum()), (i[keep], j[keep])), shape=(N, N)).tocsr()
    A = ((A + A.T) > 0).astype(np.float64).tocsr()
    A = sp.triu(A, 1).tocsr()
    w = {"pmj": lambda: rng.choice([-1.0, 1.0], size=A.nnz), "int3": lambda: rng.choice([-3.0, -2.0, -1.0, 1.0, 2.0, 3.0], size=A.nnz),
         "gauss": lambda: rng.normal(0, 1, A.nnz) / 4.0}[weights]()
    A.data = w
    A = (A + A.T).tolil()
    if weights == "pmj":                          # a few self-couplings: the DIAG kernel on the address format
        for k in rng.choice(np.arange(20, N), 60, replace=False):
            A[k, k] = rng.choice([-1.0, 1.0])
    A = A.tocsr(); A.sort_indices()
    deg = np.diff(A.indptr)
    assert (deg > 8).mean() > 0.6 and deg.max() >= 70 and (deg == 0).sum() > 0 and ((deg > 16) & (deg <= 33)).sum() >= 3
    hv = rng.normal(0, 0.2, N) if weights == "gauss" else rng.integers(-1, 2, N).astype(np.float64)
    inst = product.Instance(A, hv)
    betas = np.geomspace(0.2, 1.5, R)
    a = run_windows(product, inst, R, T, W, betas, True)
    b = run_windows(product, inst, R, T, W, betas, False)
    assert a[3] == W
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    csr = oracle.Csr(A)
    m0 = init_spins(R, N)
    for c in range(R):
        cb = np.tile(np.array(oracle.cb_pair(betas[c], 1.0, False)), (T * W, 1))
        M, s_fin, tr = oracle.sweeps_philox(csr, hv, m0[c], cb, SEED, c, escale=a[5], use_f64=False,
                                            efix0=int(np.rint(oracle.energy(csr, hv, m0[c]) * 2.0 ** a[5])))
        assert np.array_equal(a[0][c], s_fin)
    # per-sweep outputs on the same windows (energy trace, running minimum, recorded states)
    with product.Engine(inst, None, R) as eng:
        eng.set_spins(m0); eng.pt_init(betas)
        assert eng.plan_philox_fused(0, W, T, SEED) == W
        o = eng.sweep_philox(T * W, SEED, sweep0=0, beta=None, record_stride=1, want_energy=True, want_min=True, want_state=True)
        assert eng.last_schedule_stats()["orders"] == T          # fused windows were used
        assert np.array_equal(o["spins"][:, -1], a[0])
        E = eng.energy_of(o["spins"].reshape(-1, N)).reshape(R, T * W)
        tol = 2.0 ** -(eng.field_scale + 1) * (A.nnz / 2 + N) + 1e-9 * np.abs(E).max()      # stated tolerance
        assert np.abs(o["energy"] - E).max() <= tol
        k = np.argmin(o["energy"], axis=1)
        assert np.array_equal(o["argmin"], k) and np.array_equal(o["argmin_state"], o["spins"][np.arange(R), k])


def test_plan_ahead_and_own_stream_change_no_bit(product):
    """Engine.plan_ahead (the windows of several announced sweep_philox_windows launches planned together) and a context on a
    stream of its own (nlmc_own_stream) are scheduling decisions: states, minima, argmin states and recorded energies of a run of
    phase launches are the same bits with and without them, and a launch that does not fit the announcement (other length)
    still runs (its own plan replaces the announced one, the next announced launch plans again)."""
    N, R, S, L = 1200, 5, 24, 4
    J, h = make_instance(N, seed=41, with_h=True)
    inst = product.Instance(J, h)
    m0 = init_spins(R, N)

    def run(ahead, own):
        out = []
        with product.Engine(inst, None, R, own_stream=own) as eng:
            eng.set_spins(m0)
            if ahead:
                eng.plan_ahead(7, L, S, SEED)
            for i in range(L):
                o = eng.sweep_philox_windows(S, SEED, sweep0=7 + i * S, beta=1.7, record_stride=2, want_energy=True, want_min=True,
                                             want_state=True, want_recorded_energy=True)
                assert eng.fused_last_call
                out.append(o)
                if i == 1:                   # a call outside the announcement, in between
                    out.append(eng.sweep_philox_windows(10, SEED, sweep0=1000, beta=0.9, want_min=True, want_state=True))
                    eng.set_spins(o["argmin_state"])
            if ahead:
                assert eng._ahead.chunks_planned == 2          # all four launches at once, then again after the foreign call
                eng.plan_ahead(None, 0, 0, 0)
            return out, eng.get_spins(), eng.energy()

    ref, s_ref, e_ref = run(False, False)
    for ahead, own in ((True, False), (False, True), (True, True)):
        got, s, e = run(ahead, own)
        assert np.array_equal(s, s_ref) and np.array_equal(e, e_ref)
        for a, b in zip(got, ref):
            for k in ("spins", "energy", "min_energy", "argmin", "argmin_state", "energy_recorded"):
                if k in b and b[k] is not None:
                    assert np.array_equal(a[k], b[k]), (ahead, own, k)
