"""Row e-2 on the GPU: APT + iso-cluster moves with the temperature ladder cut into slot blocks (include/nlmc.h: nlmc_apt_shard,
csrc/nlmc_apt.h; distributed.SlotShardedAPT).  One context == two == four contexts on one device, bit for bit, by (sub-replica,
temperature slot); the one-context run == the oracle-backed restatement of the protocol (tests/fake_engine.py); the library-issued
collective path on a one-rank RCCL communicator; the C5 size (N = 10^4, 32 temperatures x 8 sub-replicas)."""
import numpy as np
import pytest

from helpers import make_instance

pytestmark = pytest.mark.gpu
SEED = 0xA5A50000


def start_states(K, R, N, seed=11):
    rng = np.random.default_rng(seed)
    return (2 * rng.integers(0, 2, size=(K, R, N)) - 1).astype(np.int8)


def drive(product, inst, betas, K, spins, W, S, rounds, pairs, precision="f32", planned=True, factory=None, want_info=True, comm=False):
    def mk(i, n, b, g, dev=0):
        return product.Engine(i, None, n, device=0, chain_base=b, n_chains_global=g, own_stream=W > 1)
    apt = product.distributed.SlotShardedAPT(factory or mk, inst, betas, K, SEED, pairs, precision=precision,
                                             device_ids=None if W == 1 else [0] * W)
    try:
        if comm:
            apt.engs[0].comm_init(product.Engine.comm_unique_id(), 1, 0)
        apt.set_spins_by_slot(spins)
        if planned:
            apt.plan(rounds, S, chunk_rounds=4)
        logs, infos = [], []
        for _ in range(rounds):
            log, info = apt.round(S, want_log=True, want_info=want_info)
            logs.append(log)
            if want_info:
                infos.append(np.concatenate(info))
        cfg, en = apt.gather_by_slot()
        exact = np.stack([e.energy_of(c) for e, c in zip([apt.engs[0]] * cfg.shape[0], cfg)])
        apt.check()
    finally:
        apt.close()
    return cfg, en, np.stack([l[0] for l in logs]), np.stack([l[1] for l in logs]), infos, exact


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_one_two_and_four_contexts_give_the_same_states(product, precision):
    N, R, K, S, rounds, pairs = 600, 8, 6, 5, 6, 3
    J, h = make_instance(N, seed=5)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.2, 2.5, R)
    spins = start_states(K, R, N)
    ref = drive(product, inst, betas, K, spins, 1, S, rounds, pairs, precision)
    assert ref[3].sum() > 0 and any((i[:, 1] > 0).any() for i in ref[4])
    assert np.array_equal(ref[1], ref[5])                          # tracked == recomputed energies (+-J instance: exact)
    for W in (2, 4):
        got = drive(product, inst, betas, K, spins, W, S, rounds, pairs, precision)
        assert np.array_equal(got[0], ref[0]), W
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[1], got[5]), W
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), W
    p, a = ref[2], ref[3]
    assert any(a[r, j, q] and p[r, j, q, 1] % (R // 4) == 0 for r in range(rounds) for j in range(K) for q in range(pairs))
    # without a plan (selection inside the swap kernel, sweep by sweep schedules): same bits
    un = drive(product, inst, betas, K, spins, 2, S, rounds, pairs, precision, planned=False)
    assert np.array_equal(un[0], ref[0]) and np.array_equal(un[3], ref[3])


def test_one_context_equals_the_oracle_restatement_of_the_protocol(product):
    """The device path (slot-keyed Philox in the sweep kernels, k_icm_round's pairing and pick, k_apt_swap) against the same
    protocol driven over the oracle-backed engine double: sweeps by oracle/nlo.c, clusters by nlo_clusters, decisions in Python."""
    from fake_engine import OracleEngine
    N, R, K, S, rounds, pairs = 300, 4, 4, 3, 4, 1
    J, h = make_instance(N, seed=8)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.3, 2.0, R)
    spins = start_states(K, R, N, seed=3)
    dev = drive(product, inst, betas, K, spins, 1, S, rounds, pairs)
    cpu = drive(product, inst, betas, K, spins, 1, S, rounds, pairs, planned=False,
                factory=lambda i, n, b, g, d=None: _Double(OracleEngine(i, n, b, g)))
    assert np.array_equal(dev[0], cpu[0]) and np.array_equal(dev[1], cpu[1])
    assert np.array_equal(dev[2], cpu[2]) and np.array_equal(dev[3], cpu[3])
    assert all(np.array_equal(x, y) for x, y in zip(dev[4], cpu[4]))
    dev2 = drive(product, inst, betas, K, spins, 2, S, rounds, pairs)
    assert np.array_equal(dev2[0], cpu[0])


class _Double:
    """OracleEngine + the two read-outs `drive` uses."""
    def __init__(self, e):
        self._e = e

    def __getattr__(self, k):
        return getattr(self._e, k)

    def energy_of(self, cfg):
        import oracle
        return np.array([oracle.energy(self._e.csr, self._e.h, s) for s in cfg])


def test_library_issued_collective_on_a_one_rank_communicator(product):
    """nlmc_apt_swap_collective with an RCCL communicator of one rank: k_apt_pack -> ncclAllGather in place (int64) -> k_apt_swap,
    on the kernels' stream -- same bits as the run without a communicator; nlmc_comm_check reports a healthy communicator."""
    N, R, K, S, rounds, pairs = 500, 6, 4, 4, 5, 2
    J, h = make_instance(N, seed=2)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.2, 2.5, R)
    spins = start_states(K, R, N)
    ref = drive(product, inst, betas, K, spins, 1, S, rounds, pairs)
    try:
        got = drive(product, inst, betas, K, spins, 1, S, rounds, pairs, comm=True)
    except NotImplementedError as ex:
        pytest.skip(str(ex))
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[3], ref[3])
    with product.Engine(J, h, K * R) as e:
        e.comm_check(1000)                                          # no communicator: nothing to check
        e.pt_init(betas)
        e.apt_shard(betas, 1, 0)
        e.comm_init(product.Engine.comm_unique_id(), 1, 0)
        e.apt_swap_collective(0, SEED, pairs)
        e.comm_check(5000)
        # the grouped ncclSend / ncclRecv pair of the N > 1 path, with this rank as its own neighbour on both sides
        e.set_spins(spins.reshape(K * R, N))
        _, lo, hi = e.apt_pack()
        rlo, rhi = e.apt_selftest_exchange()
        assert np.array_equal(rlo, hi) and np.array_equal(rhi, lo) and not np.array_equal(lo, hi)
        with pytest.raises(RuntimeError):
            e.pt_swap_philox_collective(0, SEED, pairs)             # the chain-block collective is not this mode's


def test_apt_shard_argument_checks(product):
    J, h = make_instance(300, seed=1)
    betas = np.geomspace(0.2, 2.0, 8)
    with product.Engine(J, h, 16) as e:
        with pytest.raises(RuntimeError):
            e.apt_shard(betas, 2, 0)                                # pt_init first
        e.pt_init(betas[:4])
        with pytest.raises(ValueError):
            e.apt_shard(betas, 2, 1)                                # pt_init got block 0, not block 1
        with pytest.raises(ValueError):
            e.apt_shard(betas[:6], 2, 0)                            # R_global != world x local slots
        e.apt_shard(betas, 2, 0)
        with pytest.raises(RuntimeError):
            e.pt_swap_philox(0, SEED, 1)                            # one slot block of a cut ladder: the plain round is refused
    with product.Engine(J, h, 8, chain_base=8, n_chains_global=16) as e:
        e.pt_init(betas[:4])
        with pytest.raises(ValueError):
            e.apt_shard(betas, 2, 0)                                # not self-contained


def test_c5_size_two_contexts_equal_one(product):
    """BASELINE config 5 (N = 10^4, 32 temperatures x 8 sub-replicas, rounds of 10 sweeps, 10 swap pairs): two slot blocks on one
    device == one context, by (sub-replica, slot), with the fp64 mode on fused windows; tracked == recomputed energies."""
    N, R, K, S, rounds, pairs = 10_000, 32, 8, 10, 4, 10
    J, h = make_instance(N)
    inst = product.Instance(J, h)
    betas = np.geomspace(0.05, 4.0, R)
    spins = start_states(K, R, N, seed=5)
    a = drive(product, inst, betas, K, spins, 1, S, rounds, pairs, "f64", want_info=False)
    b = drive(product, inst, betas, K, spins, 2, S, rounds, pairs, "f64", want_info=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
    assert np.array_equal(a[1], a[5]) and a[3].sum() > 0


def test_apt_icm_run_with_device_ids(product, capsys):
    """APT_ICM.run(device_ids=...) (the reference's parallelism knob is num_cores, NPT/apt_ICM.py:145-146): one block and two blocks
    of the ladder return the same M, Energy, swap decisions and final states; shapes are the reference's."""
    N, R = 400, 6
    J, h = make_instance(N, seed=9)
    betas = np.geomspace(0.3, 2.0, R)
    outs = []
    for ids in ([0], [0, 0], [0, 0, 0]):
        apt = product.APT_ICM(J.toarray(), h, rng="philox", seed=7)
        apt.num_subreplicas = 4
        M, E = apt.run(betas, R, num_sweeps_MCMC=40, num_sweeps_read=40, num_swap_attempts=8, num_swapping_pairs=2,
                       icm_feedback=True, device_ids=ids)
        assert M.shape == (R * N, 5 * 4) and E.shape == (R,) and set(np.unique(M)) <= {-1.0, 1.0}
        outs.append((M, E, apt.swap_accepted, apt.final_states, apt.final_energies))
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            assert np.array_equal(x, y)
    assert outs[0][2].sum() > 0
    M, E = outs[0][0], outs[0][1]
    with product.Engine(J, h, 1) as e:
        for r in (0, R - 1):
            blk = M[r * N:(r + 1) * N]
            assert E[r] == e.energy_of(blk.T.astype(np.int8))[:5].min()   # the FIRST num_sweeps_read_per_swap = 5 columns (NPT/apt_ICM.py:36-50,290-297)
    with pytest.raises(ValueError):
        product.APT_ICM(J.toarray(), h).run(betas, R, device_ids=[0])      # numpy-stream mode has no device-resident path
