"""Row e-2 on CPUs: APT + iso-cluster moves with the temperature ladder cut into slot blocks
(distributed.SlotShardedAPT; include/nlmc.h: nlmc_apt_shard), driven over the oracle-backed engine double.  The property the
multi-GPU runs rely on: the states on every (sub-replica, temperature slot) do not depend on the number of shards -- one process
over 1, 2 and 4 shards, and two `gloo` ranks of one shard each."""
import os
import socket
import sys

import numpy as np

from conftest import load_product
from helpers import make_instance

HERE = os.path.dirname(os.path.abspath(__file__))
N, R, K, S, ROUNDS, PAIRS, SEED = 48, 8, 4, 2, 5, 3, 777


def _setup():
    P = load_product()
    J, h = make_instance(N, seed=5, with_h=True, gaussian=True)
    inst = P.Instance(J, h)
    betas = np.geomspace(0.2, 2.5, R)
    rng = np.random.default_rng(11)
    spins = (2 * rng.integers(0, 2, size=(K, R, N)) - 1).astype(np.int8)
    return P, inst, betas, spins


def _drive(P, inst, betas, spins, n_shards=1, torch=None, dist=None):
    from fake_engine import OracleEngine

    def mk(i, n, b, g, dev=None):
        return OracleEngine(i, n, b, g)
    apt = P.distributed.SlotShardedAPT(mk, inst, betas, K, SEED, PAIRS, torch=torch, dist=dist, device="cpu",
                                       device_ids=None if (dist is not None or n_shards == 1) else [0] * n_shards)
    apt.set_spins_by_slot(spins)
    logs, sizes = [], []
    for _ in range(ROUNDS):
        log, info = apt.round(S, want_log=True, want_info=True)
        logs.append(log)
        sizes.append(np.concatenate(info))
    cfg, en = apt.gather_by_slot()
    apt.check()
    apt.close()
    return cfg, en, np.stack([l[0] for l in logs]), np.stack([l[1] for l in logs]), sizes


def test_one_process_one_two_and_four_shards_give_the_same_states():
    P, inst, betas, spins = _setup()
    ref = _drive(P, inst, betas, spins, 1)
    assert ref[3].sum() > 0                                         # swaps were accepted ...
    assert any((i[:, 1] > 0).any() for i in ref[4])                # ... and clusters were moved
    assert not np.array_equal(ref[0], spins)
    for W in (2, 4):
        got = _drive(P, inst, betas, spins, W)
        assert np.array_equal(got[0], ref[0]), W                   # configurations by (sub-replica, slot)
        assert np.array_equal(got[1], ref[1]), W                   # tracked energies
        assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), W      # the full swap log, on every shard
    # a boundary pair was accepted somewhere (otherwise the test would not have moved a configuration between shards)
    pairs, acc = ref[2], ref[3]
    assert any(acc[r, j, p] and pairs[r, j, p, 1] % (R // 4) == 0 for r in range(ROUNDS) for j in range(K) for p in range(PAIRS))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, inst, betas, spins = _setup()
        cfg, en, pairs, acc, _ = _drive(P, inst, betas, spins, torch=torch, dist=dist)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cfg=cfg, en=en, pairs=pairs, acc=acc)
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_reproduce_one_shard(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    P, inst, betas, spins = _setup()
    ref = _drive(P, inst, betas, spins, 1)
    for r in (0, 1):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(got["cfg"], ref[0]) and np.array_equal(got["en"], ref[1])
        assert np.array_equal(got["pairs"], ref[2]) and np.array_equal(got["acc"], ref[3])
