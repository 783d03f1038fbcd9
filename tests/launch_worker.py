"""Worker of tests/test_gpu_launcher.py: one rank of a (small) torch.distributed job that calls the drop-in NPT.run.  Started as a
child process with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* set; torch is imported first (before the engine's HIP runtime)."""
import json
import os
import sys

import torch                    # noqa: F401  (first: see DESIGN.md section 10)
import torch.distributed as dist
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    out, backend, mode = sys.argv[1], sys.argv[2], sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from conftest import load_product
    from helpers import make_instance
    P = load_product()
    torch.cuda.set_device(0)
    if backend == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world)     # (two ranks on ONE GPU: RCCL refuses that, gloo does not care)
    if mode == "bench":
        # the bench workload's shape cut over the ranks: ShardedTempering as bench.py drives it (fp64 mode, fused windows, planned
        # pair selections, one all-gather per round), a few rounds
        from helpers import init_spins
        N, G, S, rounds = 10_000, 256, 10, 6
        J, h = make_instance(N)
        inst = P.Instance(J, h)
        betas = np.geomspace(0.05, 4.0, G)
        dev = torch.device("cuda", 0)
        st = P.distributed.ShardedTempering(lambda i, n, b, g: P.Engine(i, None, n, device=0, chain_base=b, n_chains_global=g), inst, betas,
                                            G, 20250225, round(0.3 * G), torch=torch, dist=dist, device=dev, precision="f64")
        st.set_spins(init_spins(G, N))
        st.plan(rounds * S, rounds, chunk_rounds=4, lazy=True)
        st.run_rounds(rounds, S)
        np.savez(out + f".rank{rank}.npz", spins=st.eng.get_spins(), energy=st.eng.energy_tracked(), slots=st.eng.pt_slots(),
                 base=st.base, count=st.count)
        st.close()
        dist.destroy_process_group()
        print(json.dumps({"rank": rank, "ok": True}))
        return
    if mode == "apt":
        # APT + iso-cluster moves, the temperature ladder cut into slot blocks over the ranks (SlotShardedAPT, gloo transport: the
        # all-gather of the energies and the neighbour exchange of boundary configurations through torch.distributed on the host)
        N, R, K, S, rounds, pairs = 600, 8, 6, 5, 6, 3
        J, h = make_instance(N, seed=5)
        inst = P.Instance(J, h)
        rng = np.random.default_rng(11)
        spins = (2 * rng.integers(0, 2, size=(K, R, N)) - 1).astype(np.int8)
        apt = P.distributed.SlotShardedAPT(lambda i, n, b, g: P.Engine(i, None, n, device=0, chain_base=b, n_chains_global=g), inst,
                                           np.geomspace(0.2, 2.5, R), K, 0xA5A50000, pairs, torch=torch, dist=dist, device="cpu", precision="f64")
        apt.set_spins_by_slot(spins)
        apt.plan(rounds, S, chunk_rounds=4)
        logs = [apt.round(S, want_log=True, want_info=False)[0] for _ in range(rounds)]
        cfg, en = apt.gather_by_slot()
        apt.check()
        np.savez(out + f".rank{rank}.npz", cfg=cfg, en=en, pairs=np.stack([l[0] for l in logs]), acc=np.stack([l[1] for l in logs]))
        apt.close()
        dist.destroy_process_group()
        print(json.dumps({"rank": rank, "ok": True}))
        return
    N, R = 300, 6
    J, h = make_instance(N, seed=4)
    betas = np.geomspace(0.3, 2.5, R)
    doNMC = [False] * (R - 2) + [True, True] if mode == "nmc" else [False] * R
    obj = P.NPT(J.toarray(), h, rng="philox", seed=11)
    M, E = obj.run(betas, R, doNMC, num_sweeps_MCMC=60, num_sweeps_read=60, num_swap_attempts=6, num_swapping_pairs=2, num_cycles=1,
                   num_restarts=(1 if mode == "cut" else 2), global_beta=2.5, lambda_start=3.0, lambda_end=0.05, lambda_reduction_factor=0.8, threshold_initial=0.9999,
                   threshold_cutoff=0.97)
    np.savez(out + f".rank{rank}.npz", M=M, E=E, restart_energies=obj.restart_energies, swap_accepted=obj.swap_accepted)
    if dist.is_initialized():
        dist.destroy_process_group()
    print(json.dumps({"rank": rank, "ok": True}))


if __name__ == "__main__":
    main()
