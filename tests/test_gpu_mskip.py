"""`M_skip != 1` and `full_update_frequency != 1` on the device-resident paths.

* NPT rounds with NMC slots and M_skip = 2 (NPT/npt.py:434-437: `M = M[:, ::M_skip]` BEFORE the energies and the argmin hand-off):
  the strided traces, the argmin over the recorded columns only and the states after the round against the oracle's sequential spec;
  `NPT.run(..., M_skip=2)` takes the device path (restarts allowed) and returns the reference's shapes.
* `NMC.run_restarts(full_update_frequency=2 / 3)` on the device path (inference seeds kept aside on the device,
  nlmc_backbone_seed) == the host-managed path bit for bit."""
import contextlib
import io

import numpy as np
import pytest

import oracle
from helpers import make_instance
from test_gpu_npt_nmc import _phase_flags, _thresholds, EPS

pytestmark = pytest.mark.gpu


def test_rounds_with_nmc_slots_and_strided_recording_match_the_oracle(product):
    N, R, n_restarts, n_nmc, S, rounds, seed, M_skip = 700, 8, 2, 3, 6, 3, 0xD00D, 2
    G = R * n_restarts
    J, h = make_instance(N, seed=3)
    inst = product.Instance(J, h)
    csr = oracle.Csr(J)
    betas = np.geomspace(0.2, 2.5, R)
    doNMC = np.array([False] * (R - n_nmc) + [True] * n_nmc)
    phases, S_nmc, gbeta, temp_x = ["C", "NC", "ALL"], 8, 2.5, 20.0
    graph = product.lbp.EdgeGraph(inst)
    from nlmc_amd.distributed import LocalTempering
    lt = LocalTempering(inst, betas, G, seed, 2, [0])
    try:
        eng = lt.engs[0]
        esc = eng.energy_scale
        lt.configure_nmc(doNMC, phases, S_nmc, gbeta, temp_x, graph.epsilon(inst.h), product.lbp.lambda_list(3.0, 0.05, 0.8), EPS, 100,
                         float(np.tanh(19.06)) - EPS, _thresholds(0.9999, 0.97), M_skip=M_skip)
        m0 = (2 * np.random.default_rng(5).integers(0, 2, size=(G, N), dtype=np.int8) - 1).astype(np.int8)
        lt.set_spins(m0)
        lt.plan(rounds * S, rounds)
        state = m0.copy()
        for ii in range(rounds):
            outs = lt.round(S, record_stride=1)[0]
            nc = outs["nmc_chains"]
            mask = eng.cluster_mask()
            new_state = lt.gather_spins()
            for j, c in enumerate(nc):
                c = int(c)
                s = state[c].copy()
                cb = np.tile(np.array(oracle.cb_pair(gbeta, temp_x)), (S_nmc, 1))
                for p, kind in enumerate(phases):
                    e0 = int(np.rint(oracle.energy(csr, h, s) * 2.0 ** esc))
                    M, s_fin, tr = oracle.sweeps_philox(csr, h, s, cb, seed, c, sweep0=(1 << 31) + (ii * 3 + p) * S_nmc,
                                                        flags=_phase_flags(mask[c], kind), escale=esc, efix0=e0)
                    got = outs["nmc"][p]["spins"][j]
                    assert got.shape == (S_nmc // M_skip, N)
                    assert np.array_equal(M[::M_skip], got), f"round {ii}: NMC chain {c}, phase {kind}"       # M[:, ::M_skip]
                    s = M[::M_skip][int(np.argmin(tr[::M_skip]))].copy()    # argmin over the RECORDED columns (NPT/npt.py:434-437)
                assert np.array_equal(s_fin, new_state[c])
            state = new_state
        lt.check()
    finally:
        lt.close()


def test_npt_run_with_m_skip_raises_like_the_reference(product):
    """In NPT.run the NMC replicas' block is `M_nmc[:, -S:]` (NPT/npt.py:643-644) with 3 cycles S_nmc / M_skip ~ S / M_skip recorded
    columns: for M_skip > 1 the reference's assignment cannot be filled and raises the broadcast ValueError -- so does the drop-in,
    now from the device-resident path (also with restarts, which the host-managed path refused)."""
    N, R = 400, 8
    J, h = make_instance(N, seed=4)
    betas = np.geomspace(0.2, 2.5, R)
    args = dict(num_sweeps_MCMC=60, num_sweeps_read=30, num_swap_attempts=3, num_swapping_pairs=3, num_cycles=2, global_beta=2.5,
                lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, threshold_initial=0.9999, threshold_cutoff=0.97)
    for kw in (dict(num_restarts=1), dict(num_restarts=3)):
        obj = product.NPT(J.toarray(), h, rng="philox", seed=99)
        with contextlib.redirect_stdout(io.StringIO()):
            # S = 20 sweeps per round, NMC phases of ceil(60 / 3 / 3 / 2) = 4 sweeps, 6 phases, M_skip = 2: 12 recorded columns < 20
            with pytest.raises(ValueError, match=r"could not broadcast input array from shape \(400,12\) into shape \(400,20\)"):
                obj.run(betas, R, [False] * 5 + [True] * 3, M_skip=2, **dict(args, **kw))
            M, E = obj.run(betas, R, [False] * 5 + [True] * 3, M_skip=1, **dict(args, **kw))
        assert M.shape == (R * N, 20) and E.shape == (R,)


@pytest.mark.parametrize("fuf,clusters", [(2, "given"), (3, "given"), (2, "inferred")])
def test_run_restarts_full_update_frequency_device_equals_host(product, fuf, clusters):
    N, R = 500, 6
    J, h = make_instance(N, seed=8)
    kw = dict(num_sweeps_initial=30, num_sweeps_per_NMC_phase=12, num_NMC_cycles=5, full_update_frequency=fuf, global_beta=2.5,
              lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, threshold_initial=0.9999, threshold_cutoff=0.97,
              all_clusters=np.arange(0, N, 7) if clusters == "given" else None)
    res = []
    for force_host in (False, True):
        obj = product.NMC(J.toarray(), h, rng="philox", seed=5, lbp="device")
        with contextlib.redirect_stdout(io.StringIO()):
            res.append(obj.run_restarts(R, _force_host=force_host, **kw))
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    assert res[0][2].shape == (R, 1 + sum(2 + (1 if c % fuf == 0 else 0) for c in range(5)))
