"""NPT rounds whose doNMC slots run NMC_task (NPT/npt.py:479-512, 622-647), device-resident and restart-batched
(include/nlmc.h: nlmc_pt_mark_slots ... nlmc_set_phase; distributed.LocalTempering.configure_nmc):

* the round protocol at the C3 size of BASELINE.json (N = 10^3, 32 temperatures, the 8 coldest slots NMC, 8 restarts = 256
  chains) against the oracle's sequential spec, chain by chain for a sample of chains: plain chains at their slot's
  temperature, NMC chains through backbone flags -> three phases with argmin hand-offs, then the swap round;
* the backbone mask kernel against the host cluster growth (nlmc_find_clusters) on the same marginals;
* NPT.run(doNMC=..., num_restarts=...) through the drop-in class: restart 0 == the single-ladder run, device split == one
  context;
* the distribution of the minimum energies against 40 runs of the reference itself with mixed doNMC
  (tests/golden/stats_minenergy_pmj96_mixed.npz, tools/make_golden.py statsmix).
"""
import contextlib
import io

import numpy as np
import pytest

import oracle
from oracle.pt import swap_round as expected_swap_round
from conftest import golden
from helpers import make_instance

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


def _thresholds(t0, cutoff):
    thr, t = [float(t0)], t0 - 0.01
    while t > cutoff:
        thr.append(float(t))
        t -= 0.01
    return thr


def _phase_flags(mask, kind):
    if kind == "ALL":
        return None
    m = mask.astype(bool)
    return np.where(m, 1, 2).astype(np.uint8) if kind == "C" else np.where(m, 2, 0).astype(np.uint8)


def test_rounds_with_nmc_slots_match_the_oracle_at_c3_size(product):
    N, R, n_restarts, n_nmc, S, rounds, seed = 1000, 32, 8, 8, 20, 3, 0xC3C3
    G = R * n_restarts
    J, h = make_instance(N)
    inst = product.Instance(J, h)
    csr = oracle.Csr(J)
    betas = np.geomspace(0.1, 3.0, R)
    doNMC = np.array([False] * (R - n_nmc) + [True] * n_nmc)
    phases, S_nmc, gbeta, temp_x, n_pairs = ["C", "NC", "ALL"], 7, 3.0, 20.0, 10
    graph = product.lbp.EdgeGraph(inst)
    eps = graph.epsilon(inst.h)
    lams = product.lbp.lambda_list(3.0, 0.05, 0.8)
    thr = _thresholds(0.9999, 0.97)
    sat = float(np.tanh(19.06)) - EPS
    from nlmc_amd.distributed import LocalTempering
    lt = LocalTempering(inst, betas, G, seed, n_pairs, [0])
    try:
        eng = lt.engs[0]
        esc = eng.energy_scale
        lt.configure_nmc(doNMC, phases, S_nmc, gbeta, temp_x, eps, lams, EPS, 100, sat, thr)
        m0 = (2 * np.random.default_rng(5).integers(0, 2, size=(G, N), dtype=np.int8) - 1).astype(np.int8)
        lt.set_spins(m0)
        lt.plan(rounds * S, rounds)
        assert lt._planners[0].window == 20 and lt._nmc_planners[0].window == 7
        slots = lt.slots()
        state = m0.copy()
        rng = np.random.default_rng(11)
        for ii in range(rounds):
            outs = lt.round(S, record_stride=1, want_energy=True)[0]
            # both sweep paths ran on fused windows (plan slots 0 and 1)
            assert lt._planners[0]._fused_from <= ii < lt._planners[0]._fused_to
            assert lt._nmc_planners[0]._fused_from <= ii * 3 < lt._nmc_planners[0]._fused_to
            pc, nc = outs["plain_chains"], outs["nmc_chains"]
            assert len(pc) == (R - n_nmc) * n_restarts and len(nc) == n_nmc * n_restarts
            assert np.array_equal(np.sort(np.concatenate([pc, nc])), np.arange(G))
            assert np.all(doNMC[slots[nc]]) and not np.any(doNMC[slots[pc]])
            mask = eng.cluster_mask()
            new_state = lt.gather_spins()
            E_track = eng.energy_tracked()
            # --- plain chains: S sweeps at the slot's temperature
            for j in rng.choice(len(pc), size=5, replace=False):
                c = int(pc[j])
                cb = np.tile(np.array(oracle.cb_pair(betas[slots[c]])), (S, 1))
                e0 = int(np.rint(oracle.energy(csr, h, state[c]) * 2.0 ** esc))
                M, s_fin, tr = oracle.sweeps_philox(csr, h, state[c], cb, seed, c, sweep0=ii * S, escale=esc, efix0=e0)
                assert np.array_equal(M, outs["plain"]["spins"][j]), f"round {ii}: plain chain {c}"
                assert np.array_equal(tr * 2.0 ** -esc, outs["plain"]["energy"][j])
                assert np.array_equal(s_fin, new_state[c]) and E_track[c] == tr[-1] * 2.0 ** -esc
            # --- NMC chains: flags from the device's own backbone mask, three phases, argmin hand-offs
            for j in rng.choice(len(nc), size=4, replace=False):
                c = int(nc[j])
                assert 0 < mask[c].sum() < N, "the test instance should give a non-trivial backbone"
                s = state[c].copy()
                cb = np.tile(np.array(oracle.cb_pair(gbeta, temp_x)), (S_nmc, 1))
                for p, kind in enumerate(phases):
                    e0 = int(np.rint(oracle.energy(csr, h, s) * 2.0 ** esc))
                    M, s_fin, tr = oracle.sweeps_philox(csr, h, s, cb, seed, c, sweep0=(1 << 31) + (ii * 3 + p) * S_nmc,
                                                        flags=_phase_flags(mask[c], kind), escale=esc, efix0=e0)
                    assert np.array_equal(M, outs["nmc"][p]["spins"][j]), f"round {ii}: NMC chain {c}, phase {kind}"
                    assert np.array_equal(tr * 2.0 ** -esc, outs["nmc"][p]["energy"][j])
                    s = M[int(np.argmin(tr))].copy()        # first minimum: np.argmin (NPT/npt.py:436)
                assert np.array_equal(s_fin, new_state[c]) and E_track[c] == tr[-1] * 2.0 ** -esc
            # --- the backbone masks: host cluster growth on the marginals of the same inference kernel
            from nlmc_amd.lbp import lbp_convexified_device
            seeds = state[nc[:6]].astype(np.float64)
            with product.Engine(inst, None, 1) as e1:
                cl = lbp_convexified_device(e1, graph, 3.0, 0.05, 0.8, seeds, eps, EPS, 100, 0.9999, 0.97, gbeta, flat=True)
            for q in range(6):
                want = np.zeros(N, dtype=np.uint8)
                want[cl[q]] = 1
                assert np.array_equal(want, mask[nc[q]]), f"round {ii}: backbone mask of chain {nc[q]}"
            # --- swap round on the tracked energies
            exp_slots, _, _ = expected_swap_round(E_track, slots, betas, R, n_pairs, ii, seed)
            slots = lt.slots()
            assert np.array_equal(slots, exp_slots)
            state = new_state
        lt.check()
    finally:
        lt.close()


def _run(product, J, h, **kw):
    N, R = J.shape[0], 8
    betas = np.geomspace(0.2, 2.5, R)
    obj = product.NPT(J, h, rng="philox", seed=kw.pop("seed", 99))
    args = dict(num_sweeps_MCMC=60, num_sweeps_read=30, num_swap_attempts=3, num_swapping_pairs=3, num_cycles=1, global_beta=2.5,
                lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, threshold_initial=0.9999, threshold_cutoff=0.97)
    args.update(kw)
    with contextlib.redirect_stdout(io.StringIO()):
        M, E = obj.run(betas, R, [False] * 5 + [True] * 3, **args)
    return obj, M, E


def test_npt_run_with_nmc_replicas_restarts_and_device_split(product):
    J, h = make_instance(400, seed=4)
    o1, M1, E1 = _run(product, J, h)
    assert M1.shape == (8 * 400, 20) and M1.dtype == np.float64 and E1.shape == (8,)
    assert set(np.unique(M1)) <= {-1.0, 1.0}
    # Energy[r] = min over the first R_swap columns of block r, fp64 (NPT/npt.py:685-692)
    csr = oracle.Csr(o1.J)
    for r in range(8):
        blk = M1[r * 400:(r + 1) * 400]
        assert E1[r] == min(oracle.energy(csr, o1.h, blk[:, t].astype(np.int8)) for t in range(10))
    o2, M2, E2 = _run(product, J, h)                                   # same seed: same bits
    assert np.array_equal(M1, M2) and np.array_equal(E1, E2)
    o8, M8, E8 = _run(product, J, h, num_restarts=8)                   # restart 0 IS the single-ladder run
    assert np.array_equal(M8, M1) and np.array_equal(E8, E1) and np.array_equal(o8.swap_accepted, o1.swap_accepted)
    assert o8.restart_energies.shape == (8, 8) and np.array_equal(o8.restart_energies[0], E1)
    assert len({tuple(row) for row in o8.restart_energies}) > 1        # the other ladders are other trajectories
    o4, M4, E4 = _run(product, J, h, num_restarts=8, device_ids=[0, 0, 0, 0])   # two ladders per context
    assert np.array_equal(M4, M8) and np.array_equal(o4.restart_energies, o8.restart_energies)
    assert np.array_equal(o4.swap_log_all[1], o8.swap_log_all[1])
    oi, Mi, Ei = _run(product, J, h, num_restarts=8, return_trace="int8")
    assert Mi.dtype == np.int8 and np.array_equal(Mi, M8.astype(np.int8)) and np.array_equal(Ei, E8)
    on, Mn, En = _run(product, J, h, num_restarts=8, return_trace=None)
    assert Mn is None and np.array_equal(En, E8)                       # +-J instance: tracked == fp64 energies
    with pytest.raises(ValueError):
        _run(product, J, h, num_restarts=8, device_ids=[0, 0, 0])       # ladders would be cut
    with pytest.raises(ValueError):
        _run(product, J, h, num_restarts=2, M_skip=2)                   # strided NMC traces: host-managed path only
    with pytest.raises(ValueError, match="could not broadcast"):        # 21 phases x 2 sweeps < 60 (NPT/npt.py:643-644)
        _run(product, J, h, num_sweeps_MCMC=180, num_cycles=10, full_update_frequency=100)


def test_npt_run_flags_a_diverged_inference_like_the_reference(product):
    """max_iterations too small for the first lambda: the reference raises ValueError('LBP diverged at initial lambda ...')
    inside NMC_task (NPT/npt.py:178-180); the device-resident run raises it at its end-of-run check."""
    J, h = make_instance(300, seed=6)
    with pytest.raises(ValueError, match="LBP diverged at initial lambda"):
        _run(product, J, h, max_iterations=2)


def test_min_energies_with_nmc_slots_match_the_reference(product):
    """The reference's NPT.run with mixed doNMC (two coldest of six slots run NMC_task), 40 independent runs, against 256
    restarts of the device-resident path under the same budget: mean read-out energy per temperature slot, mean minimum.
    The NMC slots heat their backbones to beta / temp_x every round, which moves their read-out energies by ~15 units
    against plain replicas (stats_minenergy_pmj96): an NMC round that did something else would not pass."""
    g = golden("stats_minenergy_pmj96_mixed")
    csr = oracle.Csr.from_parts(int(g["N"]), g["indptr"], g["indices"], g["data"])
    R, rounds, pairs, nsw = int(g["num_replicas"]), int(g["num_swap_attempts"]), int(g["num_swapping_pairs"]), int(g["num_sweeps_MCMC"])
    n_restarts = 256
    obj = product.NPT(csr.toarray(), g["h"], rng="philox", seed=1618)
    with contextlib.redirect_stdout(io.StringIO()):
        obj.run(g["beta_list"], R, [bool(v) for v in g["doNMC"]], num_sweeps_MCMC=nsw, num_sweeps_read=nsw, num_swap_attempts=rounds,
                num_swapping_pairs=pairs, num_cycles=int(g["num_cycles"]), full_update_frequency=int(g["full_update_frequency"]),
                temp_x=float(g["temp_x"]), global_beta=float(g["global_beta"]), lambda_start=float(g["lambda_start"]),
                lambda_end=float(g["lambda_end"]), lambda_reduction_factor=float(g["lambda_reduction_factor"]),
                threshold_initial=float(g["threshold_initial"]), threshold_cutoff=float(g["threshold_cutoff"]),
                max_iterations=int(g["max_iterations"]), num_restarts=n_restarts, return_trace="int8")
    E = obj.restart_energies
    ref = g["energies"]
    assert E.shape == (n_restarts, R)
    for r in range(R):
        se = np.hypot(ref[:, r].std(ddof=1) / np.sqrt(len(ref)), E[:, r].std(ddof=1) / np.sqrt(n_restarts))
        z = (E[:, r].mean() - ref[:, r].mean()) / max(se, 1e-3)
        assert abs(z) < 4.5, f"slot {r}: ours {E[:, r].mean():.2f} vs reference {ref[:, r].mean():.2f} ({z:.1f} sigma)"
    m_us, m_ref = E.min(axis=1), g["min_energy"]
    se = np.hypot(m_ref.std(ddof=1) / np.sqrt(len(m_ref)), m_us.std(ddof=1) / np.sqrt(n_restarts))
    z = (m_us.mean() - m_ref.mean()) / se
    assert abs(z) < 4.5, f"mean minimum: ours {m_us.mean():.2f} vs reference {m_ref.mean():.2f} ({z:.1f} sigma)"


def test_npt_with_nmc_replicas_on_chimera128_against_the_known_ground_state(product):
    """Known answer that ships with the reference (NMC/examples/Chimera_droplet_instances/chimera128_spinglass_power: instance 001
    and the first line of groundstates_otn2d.txt).  The device-resident NPT with NMC_task on its two coldest slots (16 slots, 16
    restarts, 2*10^4 sweeps) must end within 1 % of the listed ground-state energy and never below it; the plain ladder of the
    same shape reads out the ground state itself (the NMC slots end every round on a short plain phase after heating their
    backbone, so their read-out energies sit above a plain replica's -- in the reference too, stats_minenergy_pmj96_mixed)."""
    import os
    from conftest import GOLDEN
    d = os.path.join(GOLDEN, "instances")
    W, h = product.instances.txt_to_A_droplet(os.path.join(d, "chimera128__001.txt"))
    tok = open(os.path.join(d, "chimera128__groundstate_001.txt")).read().split()
    e_gs = float(tok[2])
    J = -W                                                       # NMC/examples/chimera_example.py: J = -W, h = -h
    hh = -np.asarray(h, dtype=np.float64).reshape(-1)
    nf = abs(J).max()
    R = 16
    best = {}
    for n_nmc in (2, 0):
        obj = product.NPT(J, hh, rng="philox", seed=8)
        with contextlib.redirect_stdout(io.StringIO()):
            obj.run(np.geomspace(0.3, 20.0, R), R, [False] * (R - n_nmc) + [True] * n_nmc, num_sweeps_MCMC=20000,
                    num_sweeps_read=20000, num_swap_attempts=400, num_swapping_pairs=R // 3, num_cycles=1, global_beta=10.0,
                    temp_x=20, lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, num_restarts=16, return_trace=None)
        best[n_nmc] = obj.restart_energies.min() * nf             # run() normalises by max|J| (NPT/npt.py:588-590)
        assert best[n_nmc] >= e_gs - 1e-3, (n_nmc, best[n_nmc], e_gs)
    assert best[2] <= e_gs * (1 - 0.01), (best, e_gs)
    assert abs(best[0] - e_gs) < 1e-3, (best, e_gs)


def test_rounds_with_nmc_slots_planned_and_unplanned_draw_the_same_numbers(product):
    """without a plan the NMC phases of round i used sweep indices that round i + 1's plain sweeps (and phases) used
    again.  The phases now draw from a counter range of their own (distributed.NMC_SWEEP_SPACE), the same with and without a plan:
    planned rounds, unplanned rounds and a plan that covers only the first rounds give the same states and slots."""
    N, R, n_restarts, n_nmc, S, rounds, seed = 600, 8, 2, 3, 6, 4, 0xBEEF
    G = R * n_restarts
    J, h = make_instance(N, seed=3)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.2, 2.5, R)
    doNMC = np.array([False] * (R - n_nmc) + [True] * n_nmc)
    graph = product.lbp.EdgeGraph(inst)
    args = (doNMC, ["C", "NC", "ALL"], 5, 2.5, 20.0, graph.epsilon(inst.h), product.lbp.lambda_list(3.0, 0.05, 0.8), EPS, 100,
            float(np.tanh(19.06)) - EPS, _thresholds(0.9999, 0.97))
    from nlmc_amd.distributed import LocalTempering
    m0 = (2 * np.random.default_rng(5).integers(0, 2, size=(G, N), dtype=np.int8) - 1).astype(np.int8)

    def drive(planned_rounds):
        lt = LocalTempering(inst, betas, G, seed, 2, [0])
        try:
            lt.configure_nmc(*args)
            lt.set_spins(m0)
            if planned_rounds:
                lt.plan(planned_rounds * S, planned_rounds)
            for _ in range(rounds):
                lt.round(S)
            lt.check()
            return lt.gather_spins(), lt.slots(), lt.sweeps_done, lt.nmc_sweeps_done
        finally:
            lt.close()
    a, b, c = drive(rounds), drive(0), drive(2)
    assert a[2:] == (rounds * S, rounds * 3 * 5) == b[2:] == c[2:]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    assert not np.array_equal(a[0], m0)
