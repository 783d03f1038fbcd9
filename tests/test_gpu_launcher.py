"""NPT.run as one rank of a torch.distributed.run job: restarts sharded over the ranks with NMC_task slots (whole
ladders per rank: no collective), and the cut-ladder driver with the library-issued all-gather (rehearsed with one rank).  The
workers are child processes (tests/launch_worker.py); results must equal the single-process run bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import make_instance

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(tmp_path, world, backend, mode, extra_env=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "launch_worker.py"), str(tmp_path / "o"), backend, mode],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-3000:]
    return [np.load(str(tmp_path / "o") + f".rank{r}.npz") for r in range(world)]


def reference(product, mode):
    N, R = 300, 6
    J, h = make_instance(N, seed=4)
    betas = np.geomspace(0.3, 2.5, R)
    doNMC = [False] * (R - 2) + [True, True] if mode == "nmc" else [False] * R
    obj = product.NPT(J.toarray(), h, rng="philox", seed=11)
    M, E = obj.run(betas, R, doNMC, num_sweeps_MCMC=60, num_sweeps_read=60, num_swap_attempts=6, num_swapping_pairs=2, num_cycles=1,
                   num_restarts=2, global_beta=2.5, lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, threshold_initial=0.9999,
                   threshold_cutoff=0.97)
    return M, E, obj.restart_energies, obj.swap_accepted, obj.swap_log_all[1]


@pytest.mark.parametrize("mode", ["plain", "nmc"])
def test_two_ranks_share_the_restarts(product, tmp_path, mode):
    """Two ranks (gloo group, both on the one GPU of the box), one restart each: rank 0 returns restart 0, rank 1 restart 1, both
    know every restart's energies -- equal to the single-process run with num_restarts = 2.  mode "nmc": the two coldest slots
    run NMC_task inside every round (device-resident, distributed.ShardedTempering-style whole ladders per rank)."""
    M, E, RE, acc, acc_all = reference(product, mode)
    got = launch(tmp_path, 2, "gloo", mode)
    assert np.array_equal(got[0]["M"], M) and np.array_equal(got[0]["E"], E)
    assert np.array_equal(got[0]["swap_accepted"], acc)
    for r in (0, 1):
        assert np.array_equal(got[r]["restart_energies"], RE)
        assert np.array_equal(got[r]["E"], RE[r])
    assert np.array_equal(got[1]["swap_accepted"], acc_all[:, 1].reshape(-1))
    assert not np.array_equal(got[1]["M"], M)


def test_one_rank_cut_ladder_driver_with_the_library_collective(product, tmp_path):
    """A one-rank RCCL job forced onto the cut-ladder driver (ShardedTempering: energies all-gathered by the library on the kernels'
    stream, read-out gathered over the ranks): the same M, Energy and swap decisions as the plain call."""
    M, E, RE, acc, _ = reference(product, "plain")
    try:
        got = launch(tmp_path, 1, "nccl", "plain", {"NLMC_NPT_FORCE_COLLECTIVE": "1"})
    except AssertionError as ex:
        if "librccl" in str(ex) or "NCCL" in str(ex):
            pytest.skip(str(ex)[-300:])
        raise
    assert np.array_equal(got[0]["M"], M) and np.array_equal(got[0]["E"], E)
    assert np.array_equal(got[0]["restart_energies"], RE) and np.array_equal(got[0]["swap_accepted"], acc)


def test_two_ranks_cut_one_ladder(product, tmp_path):
    """Two ranks (gloo group, both on the one GPU), ONE ladder of 6 temperatures: three chains per rank, the swap round decided on
    every rank from the all-gathered energies (distributed.ShardedTempering with real engines on both sides; the collective itself
    goes through the host here -- RCCL refuses two ranks on one GPU), label exchanges applied to the replicated slot table, the
    last round's trace gathered: the same M, Energy and swap decisions as one process."""
    N, R = 300, 6
    J, h = make_instance(N, seed=4)
    betas = np.geomspace(0.3, 2.5, R)
    obj = product.NPT(J.toarray(), h, rng="philox", seed=11)
    M, E = obj.run(betas, R, [False] * R, num_sweeps_MCMC=60, num_sweeps_read=60, num_swap_attempts=6, num_swapping_pairs=2, num_cycles=1)
    got = launch(tmp_path, 2, "gloo", "cut")
    for r in (0, 1):
        assert np.array_equal(got[r]["M"], M) and np.array_equal(got[r]["E"], E)
        assert np.array_equal(got[r]["swap_accepted"], obj.swap_accepted)


def test_two_ranks_cut_the_bench_ladder(product, tmp_path):
    """The bench workload's shape (N = 10^4, one ladder of 256 temperatures, fp64 mode on fused windows, 77 planned pairs per round)
    cut over two ranks of 128 chains, driven like bench.py drives it: spins, tracked energies and the slot table after six rounds
    equal one process holding all 256 chains."""
    from helpers import init_spins
    N, G, S, rounds = 10_000, 256, 10, 6
    J, h = make_instance(N)
    inst = product.Instance(J, h)
    st = product.distributed.ShardedTempering(lambda i, n, b, g: product.Engine(i, None, n, chain_base=b, n_chains_global=g), inst,
                                              np.geomspace(0.05, 4.0, G), G, 20250225, round(0.3 * G), precision="f64")
    st.set_spins(init_spins(G, N))
    st.plan(rounds * S, rounds, chunk_rounds=4, lazy=True)
    st.run_rounds(rounds, S)
    spins, energy, slots = st.eng.get_spins(), st.eng.energy_tracked(), st.eng.pt_slots()
    st.close()
    got = launch(tmp_path, 2, "gloo", "bench")
    for r in (0, 1):
        b, c = int(got[r]["base"]), int(got[r]["count"])
        assert (b, c) == (128 * r, 128)
        assert np.array_equal(got[r]["spins"], spins[b:b + c]) and np.array_equal(got[r]["energy"], energy[b:b + c])
        assert np.array_equal(got[r]["slots"], slots)


def test_two_ranks_share_the_apt_slot_blocks(product, tmp_path):
    """Row e-2 with two PROCESSES (gloo group, both on the one GPU): 8 temperatures x 6 sub-replicas, four temperatures per rank,
    Houdayer moves local, swaps across the block boundary exchange configurations between the processes -- states by (sub-replica,
    slot), tracked energies and the swap log equal one context (tests/test_gpu_apt_sharded.py covers W contexts of ONE process)."""
    from test_gpu_apt_sharded import drive, start_states
    N, R, K, S, rounds, pairs = 600, 8, 6, 5, 6, 3
    J, h = make_instance(N, seed=5)
    ref = drive(product, product.Instance(J, h), np.geomspace(0.2, 2.5, R), K, start_states(K, R, N), 1, S, rounds, pairs, "f64", want_info=False)
    got = launch(tmp_path, 2, "gloo", "apt")
    assert ref[3].sum() > 0
    for r in (0, 1):
        assert np.array_equal(got[r]["cfg"], ref[0]) and np.array_equal(got[r]["en"], ref[1])
        assert np.array_equal(got[r]["pairs"], ref[2]) and np.array_equal(got[r]["acc"], ref[3])
