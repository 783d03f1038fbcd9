"""world_size-2 `gloo` rehearsal of the replica-sharded tempering driver on the CPU (engine = oracle-backed test
double).  Checks the property the multi-GPU runs rely on: the trajectory does not depend on the number of ranks."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import load_product
from helpers import make_instance, init_spins

HERE = os.path.dirname(os.path.abspath(__file__))
N, L, NL, S, ROUNDS, PAIRS, SEED = 60, 4, 2, 3, 4, 1, 4242


def _setup():
    P = load_product()
    J, h = make_instance(N, seed=3, with_h=True, gaussian=True)
    inst = P.Instance(J, h)
    betas = np.geomspace(0.3, 2.5, L)
    return P, inst, betas, init_spins(L * NL, N)


def _drive(P, inst, betas, m0, torch=None, dist=None):
    from fake_engine import OracleEngine
    st = P.distributed.ShardedTempering(lambda i, n, b, g: OracleEngine(i, n, b, g), inst, betas, L * NL, SEED, PAIRS,
                                        torch=torch, dist=dist, device="cpu")
    st.set_spins(m0)
    for _ in range(ROUNDS):
        st.round(S)
    return st.gather_spins(), st.eng.pt_slots(), (st.base, st.count)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, inst, betas, m0 = _setup()
        spins, slots, part = _drive(P, inst, betas, m0, torch=torch, dist=dist)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), spins=spins, slots=slots, part=np.array(part))
    finally:
        dist.destroy_process_group()


def test_block_partition():
    P = load_product()
    bp = P.distributed.block_partition
    assert [bp(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert [bp(256, 8, r) for r in range(8)] == [(32 * r, 32) for r in range(8)]
    cover = sorted(i for r in range(3) for i in range(bp(7, 3, r)[0], sum(bp(7, 3, r))))
    assert cover == list(range(7))


def test_two_ranks_reproduce_single_process(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    P, inst, betas, m0 = _setup()
    ref_spins, ref_slots, _ = _drive(P, inst, betas, m0)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert tuple(r0["part"]) == (0, 4) and tuple(r1["part"]) == (4, 4)
    for r in (r0, r1):
        assert np.array_equal(r["spins"], ref_spins)        # same bits for 1 and 2 ranks
        assert np.array_equal(r["slots"], ref_slots)        # replicated slot table stayed consistent
    assert not np.array_equal(ref_slots, np.arange(L * NL) % L)            # the run did exchange temperatures


@pytest.mark.parametrize("fused_ok", [True, False])
def test_planned_rounds_equal_unplanned(fused_ok):
    """ShardedTempering.plan() + RoundPlanner on the CPU test double: rounds run through the planner -- as fused-window
    launches when the engine accepts the plan, sweep by sweep when it declines -- give the same bits as rounds without
    any plan, and the planner asks for the right windows."""
    from fake_engine import OracleEngine
    P, inst, betas, m0 = _setup()
    G, S6, ROUNDS5 = L * NL, 6, 5

    def drive(plan):
        made = []

        def mk(i, n, b, g):
            e = OracleEngine(i, n, b, g)
            e.fused_ok = fused_ok
            made.append(e)
            return e
        st = P.distributed.ShardedTempering(mk, inst, betas, G, SEED, PAIRS, device="cpu")
        st.set_spins(m0)
        if plan:
            st.plan(ROUNDS5 * S6, ROUNDS5)
        for _ in range(ROUNDS5):
            st.round(S6)
        return st.gather_spins(), st.eng.pt_slots(), made[0], st

    a_spins, a_slots, eng_a, st_a = drive(True)
    b_spins, b_slots, _, _ = drive(False)
    assert np.array_equal(a_spins, b_spins) and np.array_equal(a_slots, b_slots)
    assert eng_a.fused_calls == 1                              # asked once, for all rounds
    assert st_a._planner.window == (6 if fused_ok else 0)      # S = 6 sweeps -> one window of 6 per round
    assert getattr(eng_a, "planned_plain", 0) == (0 if fused_ok else 1)


def test_fused_window_choice():
    P = load_product()
    fw = P.engine.fused_window
    assert fw(10) == 10 and fw(100) == 50 and fw(64) == 64 and fw(97) == 0 and fw(2) == 0 and fw(128) == 64 and fw(9) == 9


def test_local_tempering_contexts_of_whole_and_of_cut_ladders_give_the_same_bits():
    """distributed.LocalTempering (one process, several contexts) on the CPU double: 4 ladders of 4 slots in 1 context, in 2 and
    4 contexts (whole ladders: every context decides its own ladders' swaps, nothing passes through the host) and in 8 contexts
    (every ladder cut in two: tracked energies gathered on the host) -- same spins, same slots after every round."""
    from fake_engine import OracleEngine
    P = load_product()
    J, h = make_instance(N, seed=3, with_h=True, gaussian=True)
    inst = P.Instance(J, h)
    Lq, NLq = 4, 4
    G = Lq * NLq
    betas = np.geomspace(0.3, 2.5, Lq)
    m0 = init_spins(G, N)

    def drive(k):
        lt = P.distributed.LocalTempering(inst, betas, G, SEED, 1, [0] * k, engine_factory=lambda i, n, b, g: OracleEngine(i, n, b, g))
        assert lt.whole_ladders == (k <= NLq)
        lt.set_spins(m0)
        hist = []
        for _ in range(ROUNDS):
            lt.round(S)
            hist.append(lt.slots().copy())
        out = lt.gather_spins(), np.array(hist)
        lt.close()
        return out
    ref_s, ref_h = drive(1)
    assert not np.array_equal(ref_h[-1], np.arange(G) % Lq)
    for k in (2, 4, 8):
        s_, h_ = drive(k)
        assert np.array_equal(s_, ref_s), k
        assert np.array_equal(h_, ref_h), k
