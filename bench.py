#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sweep + replica-exchange path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one replica-exchange round of NPT: S_SWAP heat-bath sweeps of every replica at its ladder temperature
followed by one swap-attempt round.  Workload (SURVEY.md section 8d, config C4 on one GPU): synthetic +-J spin glass,
N = 10^4 spins, exactly 3N edges (mean degree 6), h = 0; 256 replicas PER GPU on a geometric beta ladder 0.05 -> 4
that spans all GPUs (256*N_gpus slots); 10 sweeps per round; round(0.3 * replicas) swap pairs per round.  Weak
scaling: per-GPU work is fixed; the only collective is one all-gather of 256 float64 energies per rank and round.
Inputs (instance, replica states, level schedules) are resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0).  `value` = spin-updates/s of the whole job; `roofline` prices the dominant kernel
(k_sweep_fused) with HIP events recorded on the stream it runs on; `cpu_baseline` times the oracle (a C port of the
same algorithm, one thread) on a bounded sample of the same workload; `cpu_baseline_numpy_path` times the
reference's own NumPy loop structure (restated in oracle/numpy_path.py, pinned to a golden) the same way.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

N_SPINS = 10_000
REPLICAS_PER_GPU = 256
S_SWAP = 10
BETA_MIN, BETA_MAX = 0.05, 4.0
INSTANCE_SEED = 20250225
PHILOX_SEED = 0xA5A50000
BYTES_PER_UPDATE = 63        # 9*d + 9 at d = 6 (fp32 J, int32 col, int8 spins)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per k_sweep_fused launch of THIS workload from the PMC passes committed in
# profiles/r01_f_sweep_hbm_traffic_pmc.csv: (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 read correction applied).
# It is the touched part of the level schedule fetched once per XCD (plus the warm-up touch of the next window's) and the
# spin write-back; PMC cannot be read live.
HBM_TRAFFIC_BYTES_PER_LAUNCH = 85_637_031


def cpu_baseline(J, h, seconds=12.0):
    """oracle/nlo.c (kind "port"): sequential C restatement of the same philox-mode sweep, one chain, one thread."""
    import oracle
    csr = oracle.Csr(J)
    s = np.where(np.random.default_rng(1000).random(csr.n) < 0.5, -1, 1).astype(np.int8)
    chunk = 20
    cb = np.tile(np.array(oracle.cb_pair(1.0)), (chunk, 1))
    oracle.sweeps_philox(csr, h, s, cb[:1], PHILOX_SEED, 0, want_M=False)     # warm-up / page-in
    done, t0, emin, ef = 0, time.perf_counter(), None, 0
    ef = int(np.rint(oracle.energy(csr, h, s) * 2.0 ** 32))
    while time.perf_counter() - t0 < seconds:
        _, s, tr = oracle.sweeps_philox(csr, h, s, cb, PHILOX_SEED, 0, sweep0=done, escale=32, efix0=ef, want_M=False)
        ef = int(tr[-1])
        emin = min(float(tr.min()) * 2.0 ** -32, emin) if emin is not None else float(tr.min()) * 2.0 ** -32
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": done * csr.n / dt, "unit": "spin-updates/s", "cores": 1, "kind": "port",
            "sample": f"1 chain x {csr.n} spins x {done} sweeps at beta=1 ({dt:.1f} s of oracle/nlo.c:nlo_sweeps_philox)",
            "min_energy_seen": emin}


def numpy_path_baseline(J, h, seconds=10.0):
    """oracle/numpy_path.py (kind "port"): the reference's own per-update work (state tuple + full sparse mat-vec +
    NumPy call overhead per SPIN update, NMC/nmc.py:70-88) restated, one process; the reference itself cannot travel to
    this box.  Bounded: whole sweeps until `seconds` are used (one sweep of 10^4 spins costs ~5 s)."""
    from oracle.numpy_path import mcmc_numpy_path
    n = J.shape[0]
    np.random.seed(7)
    m = np.sign(2 * np.random.rand(n) - 1)
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        m = mcmc_numpy_path(1, m, 1.0, J, h)[:, -1]
        done += 1
    dt = time.perf_counter() - t0
    return {"value": done * n / dt, "unit": "spin-updates/s", "cores": 1, "kind": "port",
            "sample": f"1 chain x {n} spins x {done} sweeps at beta=1 ({dt:.1f} s of oracle/numpy_path.py: the reference's "
                      "NumPy loop structure)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    from __graft_entry__ import load
    from helpers import make_instance, init_spins
    P = load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    J, h = make_instance(N_SPINS, seed=INSTANCE_SEED)
    inst = P.Instance(J, h)
    G = REPLICAS_PER_GPU * world
    betas = np.geomspace(BETA_MIN, BETA_MAX, G)
    n_pairs = round(0.3 * G)
    stream = torch.cuda.current_stream().cuda_stream

    def make_engine(i, n, base, g):
        return P.Engine(i, None, n, device=local_rank, stream=stream, chain_base=base, n_chains_global=g)

    st = P.distributed.ShardedTempering(make_engine, inst, betas, G, PHILOX_SEED, n_pairs, torch=torch, dist=dist,
                                        device=torch.device("cuda", local_rank))
    base, count = st.base, st.count
    st.set_spins(np.concatenate([np.zeros((base, N_SPINS), np.int8), init_spins(count, N_SPINS, base=1000 + base),
                                 np.zeros((G - base - count, N_SPINS), np.int8)]) if world > 1 else init_spins(G, N_SPINS))
    total_rounds = a.warmup + a.steps
    st.plan(total_rounds * S_SWAP, total_rounds)  # level schedules + pair selections: resident before timing starts
    e_start = st.eng.energy()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        st.round(S_SWAP)
    st.eng.timing_reset(True)
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.round(S_SWAP)
    sync()
    dt = time.perf_counter() - t0
    tm = st.eng.timing_total()
    st.eng.timing_reset(False)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    e_end = st.eng.energy()
    sched = st.eng.last_schedule_stats()

    if rank == 0:
        updates = float(G) * N_SPINS * S_SWAP * a.steps
        upd_launch = float(count) * N_SPINS * S_SWAP
        ms_launch = tm["ms_sweep"] / max(1, tm["launches_sweep"])
        achieved = upd_launch * BYTES_PER_UPDATE / (ms_launch * 1e-3) / 1e9 if ms_launch > 0 else 0.0
        out = {
            "metric": "spin-updates/s (replicas x spins x sweeps / s), NPT sweep + swap rounds",
            "value": updates / dt,
            "unit": "spin-updates/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "NPT heat-bath sweeps + replica exchange, sparse +-J spin glass (C4 per GPU)",
                       "spins": N_SPINS, "edges": 3 * N_SPINS, "replicas_per_gpu": REPLICAS_PER_GPU,
                       "replicas_total": G, "sweeps_per_round": S_SWAP, "swap_pairs_per_round": n_pairs,
                       "beta_ladder": [BETA_MIN, BETA_MAX], "rng": "philox4x32-10", "order": "one permutation per sweep"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": HBM_TRAFFIC_BYTES_PER_LAUNCH if (world == 1 and count == REPLICAS_PER_GPU) else None,
                         "traffic_source": "profiles/r01_f_sweep_hbm_traffic_pmc.csv",
                         "algorithmic_bytes_per_launch": upd_launch * BYTES_PER_UPDATE,
                         "kernel": "k_sweep_fused<false>", "us_per_launch": ms_launch * 1e3,
                         "bytes_per_update": BYTES_PER_UPDATE, "updates_per_launch": upd_launch,
                         "note": "algorithmic bytes (SURVEY 8d: 63 B per update, no discount for rows shared by chains); "
                                 "spins live in LDS and one level schedule serves all 256 chains, so HBM carries ~5 % of "
                                 "that (traffic) and frac can exceed 1 -- the kernel is bound by VALU issue and per-level "
                                 "latency, see DESIGN.md section 5"},
            "levels_per_sweep": sched["levels"] / max(1, sched["orders"]),
            "min_energy": {"start": float(e_start.min()), "end": float(e_end.min())},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(J, h)
            out["cpu_baseline_numpy_path"] = numpy_path_baseline(J, h)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    st.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
