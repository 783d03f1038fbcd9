#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sweep + replica-exchange path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--strong]

With N > 1 and no torch.distributed environment this script starts `python -m torch.distributed.run --nproc-per-node N
bench.py ...` itself as a CHILD process (before anything touches the GPU) and exits with the child's code; launched by
torch.distributed.run it is one rank per GPU over RCCL.

A "step" is ROUNDS_PER_STEP = 5120 replica-exchange rounds of NPT (~0.6 s of GPU time: the driver's --steps 20 times ~12 s
per leg, the default 4 steps 2.4 s); a round is S_SWAP = 10 heat-bath sweeps of every replica at its ladder temperature
followed by one swap-attempt round.  TWO legs of equal standing run with the same --steps / --warmup, GPU work first, CPU
baselines last: the headline leg in the reference's arithmetic (fp64 field, 53-bit uniform: NMC/nmc.py:86-87 -- the "f64"
mode, on fused windows because the +-J instance makes the fp64 field an exact integer, bit-identical to the sweep-by-sweep
fp64 kernel and the fp64 oracle) and the fixed-point "f32" throughput mode (`fixed_point_f32`; its own stated tolerance,
DESIGN.md section 2).  Workload (SURVEY.md section 8d, config C4 on
one GPU): synthetic +-J spin glass, N = 10^4 spins, exactly 3N edges (mean degree 6), h = 0; 256 replicas PER GPU on a
geometric beta ladder 0.05 -> 4 that spans all GPUs (256*N_gpus slots; --strong: 256 replicas in total, 256/N per
GPU); round(0.3 * replicas) swap pairs per round.  The only collective is one all-gather of the local float64 energies
per rank and round.

EVERYTHING a round needs is inside the timed region: the per-sweep visiting orders and their level schedules
(k_levelize_fused -- the reference draws its permutation inside the sweep loop, NMC/nmc.py:62-71) and the pair
selections (k_pt_select) are built chunk by chunk (256 rounds) by the round that first needs them, after t0.  Only
the instance, the replica states and the ladder are resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0).  `value` = end-to-end spin-updates/s of the whole job (headline leg); `value_kernel_loop` = the same
updates over the summed durations of the sweep kernel alone; `roofline` prices the dominant kernel (HIP events recorded on the
stream it runs on) against the bound that binds it -- levels per second of a level-synchronous workgroup against the floor of one
level (barrier + LDS gather + write), MEASURED in the same run by nlmc_probe_level_round -- and `roofline_algorithmic` keeps the
contract figure of SURVEY.md section 8d (algorithmic bytes over the HBM peak; above 1 because the design does not move those bytes); `cpu_baseline` times the oracle (a C port of the same algorithm, one thread) on a
bounded sample of the same workload (`cpu_baseline_all_cores`: one such process per host core); `cpu_baseline_numpy_path` times the reference's own NumPy loop structure
(restated in oracle/numpy_path.py, pinned to a golden) the same way.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

N_SPINS = 10_000
REPLICAS_PER_GPU = 256
S_SWAP = 10
ROUNDS_PER_STEP = int(os.environ.get("NLMC_BENCH_ROUNDS_PER_STEP", "5120"))   # (the override is for the profiler passes of scripts/profile_round.sh)
# 5120 rounds x ~0.116 ms: a step is ~0.6 s, the driver's --steps 20 times ~12 s per leg (VERDICT r3 #1c)
EVENT_EVERY = int(os.environ.get("NLMC_BENCH_EVENT_EVERY", "8"))   # HIP events around every 8th sweep-kernel launch of the timed region
PLAN_CHUNK_ROUNDS = 256      # rounds whose schedules are built together (one workgroup per window: fills the chip)
BETA_MIN, BETA_MAX = 0.05, 4.0
INSTANCE_SEED = 20250225
PHILOX_SEED = 0xA5A50000
BYTES_PER_UPDATE = 63        # 9*d + 9 at d = 6 (4-byte J, int32 col, int8 spins)
BYTES_PER_UPDATE_F64 = 91    # 13*d + 13: fp64 J (parity-grade field sum)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md, L2 (per XCD): ~34.5 TB/s aggregate
VALU_LANES_PER_CU_CLK = 128  # 4 SIMD-32 per CU, one wave64 instruction per 2 cycles per SIMD (MI355X_MICROARCH.md)
CLOCK_GHZ = 2.4
N_CUS = 256
SCHEDULE_BYTES_PER_POSITION = 24 # what the sweep kernel streams per schedule position from L2: item head 8 B + 8 x 2 B row
                                 # entries (the sign format of +-J instances; 8 x 4 B = 40 B with 16-bit couplings); a row
                                 # longer than 8 entries takes two positions (an even / odd lane pair)


def load_pmc():
    """Per-launch PMC figures of the sweep kernels on THIS workload, keyed by leg ("f64" / "f32") (collected by
    scripts/profile_round.sh in separate --pmc passes, committed under profiles/; PMC cannot be read live).  Returns {} when the
    file is absent."""
    path = os.path.join(REPO, "profiles", "current_sweep_pmc.json")
    try:
        with open(path) as f:
            return json.load(f)
    except OSError:
        return {}


def cpu_baseline(J, h, seconds=12.0, chain=0, use_f64=True):
    """oracle/nlo.c (kind "port"): sequential C restatement of the same philox-mode sweep (the arithmetic of the headline leg),
    one chain, one thread."""
    import numpy as np
    import oracle
    csr = oracle.Csr(J)
    s = np.where(np.random.default_rng(1000 + chain).random(csr.n) < 0.5, -1, 1).astype(np.int8)
    chunk = 20
    cb = np.tile(np.array(oracle.cb_pair(1.0, 1.0, use_f64)), (chunk, 1))
    oracle.sweeps_philox(csr, h, s, cb[:1], PHILOX_SEED, chain, use_f64=use_f64, want_M=False)     # warm-up / page-in
    esc = oracle.field_scale(csr, h)[1]
    done, t0, emin, ef = 0, time.perf_counter(), None, 0
    ef = int(np.rint(oracle.energy(csr, h, s) * 2.0 ** esc))
    while time.perf_counter() - t0 < seconds:
        _, s, tr = oracle.sweeps_philox(csr, h, s, cb, PHILOX_SEED, chain, sweep0=done, escale=esc, efix0=ef, use_f64=use_f64, want_M=False)
        ef = int(tr[-1])
        emin = min(float(tr.min()) * 2.0 ** -esc, emin) if emin is not None else float(tr.min()) * 2.0 ** -esc
        done += chunk
    dt = time.perf_counter() - t0
    return {"value": done * csr.n / dt, "unit": "spin-updates/s", "cores": 1, "kind": "port",
            "sample": f"1 chain x {csr.n} spins x {done} sweeps at beta=1 ({dt:.1f} s of oracle/nlo.c:nlo_sweeps_philox, "
                      f"{'f64' if use_f64 else 'f32'} mode)",
            "min_energy_seen": emin}


def cpu_baseline_all_cores(seconds=8.0, use_f64=True):
    """The same oracle loop in one CHILD process per host core of this process's CPU set (each its own chain; children
    are started with subprocess and never touch the GPU): what the C port does with the whole host."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                        # a container's CPU share (cgroup v2: "<quota> <period>" or "max <period>")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = min(cores, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        pass
    cores = min(cores, 64)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["OMP_NUM_THREADS"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds), "--cpu-chain", str(i),
                               "--headline", "f64" if use_f64 else "f32"],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True) for i in range(cores)]
    vals = []
    for pr in procs:
        out, _ = pr.communicate(timeout=seconds * 10 + 120)
        if pr.returncode == 0 and out.strip():
            vals.append(json.loads(out.strip().splitlines()[-1])["value"])
    if not vals:
        return None
    return {"value": float(sum(vals)), "unit": "spin-updates/s", "cores": len(vals), "kind": "port",
            "sample": f"{len(vals)} processes x 1 chain x {N_SPINS} spins, {seconds:.0f} s each of oracle/nlo.c:nlo_sweeps_philox"}


def numpy_path_baseline(J, h, seconds=10.0):
    """oracle/numpy_path.py (kind "port"): the reference's own per-update work (state tuple + full sparse mat-vec +
    NumPy call overhead per SPIN update, NMC/nmc.py:70-88) restated, one process; the reference itself cannot travel to
    this box.  Bounded: whole sweeps until `seconds` are used (one sweep of 10^4 spins costs ~5 s)."""
    import numpy as np
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


def self_launch(a, argv):
    """N > 1 without a torch.distributed environment: become the launcher.  Runs BEFORE torch is imported -- nothing
    in this process has touched the GPU, and the ranks are children (no exec after GPU initialisation)."""
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.dry_run_launch:
        print(json.dumps({"would_run": cmd}))
        return 0
    return subprocess.call(cmd, env=env)


C5_BETAS_PER_GPU, C5_SUBREPLICAS, C5_ROUNDS_PER_STEP = 32, 8, int(os.environ.get("NLMC_BENCH_C5_ROUNDS_PER_STEP", "1024"))


def run_c5(a, P, inst, torch, dist, world, rank, local_rank):
    """BASELINE config 5: APT + Houdayer iso-cluster moves on the 10^4-spin instance, 32 temperatures x 8 sub-replicas = 256 chains
    PER GPU; the ladder of 32 x n_gpus temperatures is cut into slot blocks, one per GPU (distributed.SlotShardedAPT: all
    sub-replicas of a temperature on one GPU; per round ONE all-gather of 256 int64 tracked energies per rank and ONE exchange of
    the 8 boundary configurations with each neighbour).  A step = 1024 rounds of (10 sweeps, 128 iso-cluster moves per GPU, one
    swap round of round(0.3 R) pairs per sub-replica ladder).  Prints one JSON line (rank 0)."""
    import numpy as np
    R, K = C5_BETAS_PER_GPU * world, C5_SUBREPLICAS
    betas = np.geomspace(BETA_MIN, BETA_MAX, R)
    n_pairs = round(0.3 * R)
    stream = torch.cuda.current_stream().cuda_stream
    gloo = os.environ.get("NLMC_BENCH_BACKEND", "nccl") != "nccl"          # rehearsal (see main): collectives on host tensors
    dev = "cpu" if gloo else torch.device("cuda", local_rank)

    def mk(i, n, b, g, d=None):
        return P.Engine(i, None, n, device=local_rank, stream=stream, chain_base=b, n_chains_global=g)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def leg(precision):
        apt = P.distributed.SlotShardedAPT(mk, inst, betas, K, PHILOX_SEED, n_pairs, torch=torch, dist=dist, device=dev, precision=precision)
        rng = np.random.default_rng(1000)
        spins = (2 * rng.integers(0, 2, size=(K, R, N_SPINS), dtype=np.int8) - 1).astype(np.int8)
        apt.set_spins_by_slot(spins)
        e0 = apt.gather_by_slot()[1].min()
        wr, tr = a.warmup * C5_ROUNDS_PER_STEP, a.steps * C5_ROUNDS_PER_STEP
        apt.plan(wr + tr, S_SWAP, chunk_rounds=PLAN_CHUNK_ROUNDS)
        for _ in range(wr):
            apt.round(S_SWAP)
        sync()
        t0 = time.perf_counter()
        for _ in range(tr):
            apt.round(S_SWAP)
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        cfg, en = apt.gather_by_slot()
        ok = bool(np.array_equal(en[:, :C5_BETAS_PER_GPU], np.stack([apt.engs[0].energy_of(c[:C5_BETAS_PER_GPU]) for c in cfg])))
        coll = ("rccl ncclAllGather + grouped ncclSend/ncclRecv issued by libnlmc_hip.so on the kernels' stream" if apt.lib_collective
                else "torch.distributed all_gather + batch_isend_irecv" if (dist is not None and world > 1) else None)
        apt.check()
        apt.close()
        return {"value": float(K * R) * N_SPINS * S_SWAP * tr / dt, "unit": "spin-updates/s", "precision": precision, "rounds_timed": tr,
                "seconds_timed": dt, "ms_per_round": dt / tr * 1e3, "collective": coll,
                "min_energy": {"start": float(e0), "end": float(en.min())}, "tracked_energies_equal_fp64_recomputation": ok}

    first, second = a.headline, ("f32" if a.headline == "f64" else "f64")
    r1 = leg(first)
    r2 = None if a.no_second_leg else leg(second)
    if rank == 0:
        out = {"metric": "spin-updates/s (replicas x spins x sweeps / s), APT sweeps + iso-cluster moves + swap rounds (BASELINE config 5)",
               "value": r1["value"], "unit": "spin-updates/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": r1["seconds_timed"] / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64 field + 53-bit uniform (the reference's arithmetic)" if first == "f64" else "i32 fixed-point field + f32 threshold",
               "data": "synthetic",
               "config": {"workload": "APT + Houdayer iso-cluster moves, sparse +-J spin glass (C5 per GPU)", "spins": N_SPINS,
                          "temperatures_per_gpu": C5_BETAS_PER_GPU, "subreplicas": K, "replicas_per_gpu": C5_BETAS_PER_GPU * K,
                          "sweeps_per_round": S_SWAP, "rounds_per_step": C5_ROUNDS_PER_STEP, "swap_pairs_per_round": n_pairs,
                          "sharding": "temperature-slot blocks, all sub-replicas of a temperature on one GPU"},
               "ranks_seen": world}
        out.update({k: v for k, v in r1.items() if k not in ("value", "unit", "precision")})
        if r2 is not None:
            out["second_leg"] = r2
        out["cpu_baseline"] = None
        if gloo:
            out["rehearsal"] = f"process group 'gloo', ranks share {torch.cuda.device_count()} GPU(s): code-path rehearsal, not a measurement"
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4, help="timed steps; one step = %d swap rounds" % ROUNDS_PER_STEP)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--strong", action="store_true", help="256 replicas in total instead of 256 per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-leg", action="store_true", help="skip the fixed-point leg")
    ap.add_argument("--headline", choices=["f64", "f32"], default="f64", help=argparse.SUPPRESS)
    ap.add_argument("--persistent", action="store_true", help="a planned chunk of rounds per launch (k_rounds_fused) instead of a launch per round")
    ap.add_argument("--workload", choices=["c4", "c5"], default="c4",
                    help="c4 (default, the headline): replica-sharded NPT; c5: APT + iso-cluster moves, 32 temperatures x 8 sub-replicas "
                         "per GPU, the temperature ladder cut into slot blocks over the GPUs (BASELINE config 5)")
    ap.add_argument("--dry-run-launch", action="store_true", help="print the launcher command of --gpus N and exit")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)     # child of cpu_baseline_all_cores
    ap.add_argument("--cpu-chain", type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.cpu_worker > 0:                       # CPU-only child process: never touches the GPU
        from helpers import make_instance
        J, h = make_instance(N_SPINS, seed=INSTANCE_SEED)
        print(json.dumps(cpu_baseline(J, h, seconds=a.cpu_worker, chain=a.cpu_chain, use_f64=(a.headline == "f64"))), flush=True)
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a, sys.argv[1:]))

    import numpy as np
    import torch
    from __graft_entry__ import load
    from helpers import make_instance, init_spins
    P = load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # NLMC_BENCH_BACKEND=gloo: REHEARSAL of the multi-rank code path on a box with fewer GPUs than ranks (RCCL refuses two ranks on
    # one GPU): the ranks share the GPUs there are, the collectives go through the host.  Its numbers mean nothing; the line says so.
    backend = os.environ.get("NLMC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    red_dev = "cuda" if backend == "nccl" else "cpu"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run (also with one rank)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the one JSON line,
        # so fd 1 points at stderr until the communicator exists (forced by one small collective)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            t = torch.zeros(1, device=red_dev)
            dist.all_reduce(t)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    J, h = make_instance(N_SPINS, seed=INSTANCE_SEED)
    inst = P.Instance(J, h)
    if a.workload == "c5":
        run_c5(a, P, inst, torch, dist, world, rank, local_rank)
        if dist is not None:
            dist.destroy_process_group()
        return
    positions_per_update = 1.0 + float(np.count_nonzero(np.diff(J.indptr) > 8)) / N_SPINS
    assert np.all(np.abs(J.data) == 1.0)                   # the bench instance is +-J: 2-byte schedule entries
    sched_bytes = SCHEDULE_BYTES_PER_POSITION * positions_per_update
    G = REPLICAS_PER_GPU if a.strong else REPLICAS_PER_GPU * world
    if G % world:
        raise SystemExit(f"{G} replicas do not split evenly over {world} ranks")
    betas = np.geomspace(BETA_MIN, BETA_MAX, G)
    n_pairs = round(0.3 * G)
    stream = torch.cuda.current_stream().cuda_stream

    def make_engine(i, n, base, g):
        return P.Engine(i, None, n, device=local_rank, stream=stream, chain_base=base, n_chains_global=g)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(precision, steps, warmup):
        """warmup + steps steps of ROUNDS_PER_STEP rounds; every schedule / pair selection of the timed rounds is built
        inside the timed region (lazy chunks)."""
        st = P.distributed.ShardedTempering(make_engine, inst, betas, G, PHILOX_SEED, n_pairs, torch=torch, dist=dist,
                                            device=torch.device("cuda", local_rank), precision=precision)
        base, count = st.base, st.count
        st.set_spins(np.concatenate([np.zeros((base, N_SPINS), np.int8), init_spins(count, N_SPINS, base=1000 + base),
                                     np.zeros((G - base - count, N_SPINS), np.int8)]) if world > 1 else init_spins(G, N_SPINS))
        e_start = st.eng.energy()
        fused = precision in st.eng.fused_modes(S_SWAP)
        chunk = PLAN_CHUNK_ROUNDS if fused else 8
        wr = warmup * ROUNDS_PER_STEP
        if wr:
            # (buffers for a full chunk are allocated here: memory allocation is not part of a round's work)
            st.plan(wr * S_SWAP, wr, chunk_rounds=chunk, lazy=True, reserve_rounds=chunk)
            st.run_rounds(wr, S_SWAP, persistent=a.persistent)
        tr = steps * ROUNDS_PER_STEP
        st.plan(tr * S_SWAP, tr, chunk_rounds=chunk, lazy=True)     # a fresh, EMPTY planner: nothing is built yet
        st.eng.timing_reset(True, every=EVENT_EVERY if fused else 1)
        sync()
        t0 = time.perf_counter()
        # one process, whole ladders: one launch per round (the sweep launch decides the previous round's swap in its prologue); ranks
        # that exchange energies: sweep launch + all-gather + swap launch per round (--persistent: a planned chunk of rounds per launch,
        # k_rounds_fused -- same bits, and measured the same speed)
        st.run_rounds(tr, S_SWAP, persistent=a.persistent)
        sync()
        dt = time.perf_counter() - t0
        tm = st.eng.timing_total()
        st.eng.timing_reset(False)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        e_end = st.eng.energy()
        e_exact = st.eng.energy_of(st.eng.get_spins())            # tracked fixed-point energies == fp64 recomputation (untimed)
        sched = st.eng.last_schedule_stats()
        chunks = st._planner.chunks_planned if getattr(st, "_planner", None) is not None else 0
        coll = ("rccl ncclAllGather issued by libnlmc_hip.so on the kernels' stream" if st.lib_collective else
                "torch.distributed all_gather_into_tensor" if st.collective else None)
        # the floor of one level-synchronous step on THIS device, measured now (include/nlmc.h: nlmc_probe_level_round)
        probe = {"conflict_free_ns": st.eng.probe_level_round(16, True), "random_gather_ns": st.eng.probe_level_round(16, False)}
        st.close()
        return {"dt": dt, "tm": tm, "rounds": tr, "count": count, "e_start": e_start, "e_end": e_end, "sched": sched,
                "chunks": chunks, "collective": coll, "fused": fused, "probe": probe, "precision": precision,
                "persistent_rounds": getattr(st, "persistent_rounds", 0), "deferred_rounds": getattr(st, "deferred_rounds", 0),
                "tracked_equals_recomputed": bool(np.array_equal(e_end, e_exact))}

    KERNEL = {("f64", True): "k_sweep_fused<DIAG=false,FLAGS=false,OUT=false,FMT_ADDR,F64=true>", ("f64", False): "k_sweep_philox<double,false,PK=true>",
              ("f32", True): "k_sweep_fused<DIAG=false,FLAGS=false,OUT=false,FMT_ADDR,F64=false>", ("f32", False): "k_sweep_philox<float,false>"}
    DTYPE = {"f64": "f64 field + 53-bit uniform (the reference's arithmetic, NMC/nmc.py:86-87; on this +-J instance the fp64 field is an "
                    "exact integer and the fp64 acceptance test an exact integer threshold per field value: bit-identical to the "
                    "sweep-by-sweep fp64 kernel and the fp64 oracle)",
             "f32": "i32 field (24-bit fixed-point J) + f32 logistic threshold from 32 random bits (stated tolerance: DESIGN.md section 2)"}

    def leg_report(r):
        """Figures of one leg.  `roofline`: the bound that binds -- a workgroup advances one level per barrier round, and a round
        costs at least the probe's round (barrier + 8 LDS byte gathers in flight + 1 write at 16 waves, measured in this run);
        `roofline_algorithmic`: SURVEY 8d's bytes per update over the HBM peak (the contract line; not a bound of this design)."""
        dt, tm, tr, count = r["dt"], r["tm"], r["rounds"], r["count"]
        updates = float(G) * N_SPINS * S_SWAP * tr
        upd_launch = float(count) * N_SPINS * S_SWAP
        ms_launch = tm["ms_sweep"] / max(1, tm["launches_timed"])       # average over the launches that had events
        sec_launch = ms_launch * 1e-3
        launches_per_round = tm["launches_sweep"] / max(1, tr)
        bpu = BYTES_PER_UPDATE_F64 if r["precision"] == "f64" else BYTES_PER_UPDATE
        pmc = load_pmc().get(r["precision"], {}) if (world == 1 and count == REPLICAS_PER_GPU and r["fused"]) else {}
        levels_launch = r["sched"]["levels"] if r["fused"] else r["sched"]["levels"] / max(1, r["sched"]["orders"]) * S_SWAP / max(1.0, launches_per_round)
        ach_lv = levels_launch / sec_launch if ms_launch > 0 else 0.0
        kernel = KERNEL[(r["precision"], r["fused"])]
        if r["deferred_rounds"]:
            # one launch per round: the sweep launch decides the previous round's swap in its prologue (nlmc_pt_rounds_deferred)
            kernel = kernel.replace(">", ",DEFER=true>")
        if r["persistent_rounds"]:
            # rounds run inside k_rounds_fused: "launch" below = one ROUND of it (10 sweeps of every chain + the in-kernel swap round;
            # events around every chunk launch, divided by its rounds)
            kernel = kernel.replace("k_sweep_fused", "k_rounds_fused (per round of)")
        peak_lv = 1e9 / r["probe"]["conflict_free_ns"]
        achieved = upd_launch * bpu / sec_launch / 1e9 if ms_launch > 0 else 0.0
        out = {
            "value": updates / dt, "unit": "spin-updates/s", "dtype": DTYPE[r["precision"]], "rounds_timed": tr,
            "seconds_timed": dt, "ms_per_round": dt / tr * 1e3,
            "value_kernel_loop": (upd_launch / sec_launch / launches_per_round) * world if ms_launch > 0 else None,
            "kernel": kernel, "us_per_launch": ms_launch * 1e3, "rounds_in_persistent_launches": r["persistent_rounds"],
            "rounds_with_swap_in_the_sweep_launch": r["deferred_rounds"],
            "sweep_launches": tm["launches_sweep"], "sweep_launches_with_events": tm["launches_timed"],
            "ms_levelize": tm["ms_levelize"], "ms_sweep_kernels": ms_launch * tm["launches_sweep"],
            "plan_chunks_in_timed_region": r["chunks"],
            "levels_per_sweep": r["sched"]["levels"] / max(1, r["sched"]["orders"]),
            "min_energy": {"start": float(r["e_start"].min()), "end": float(r["e_end"].min())},
            "tracked_energies_equal_fp64_recomputation": r["tracked_equals_recomputed"],
            "roofline": {"bound": "lds-latency", "unit": "levels/s per workgroup", "achieved": ach_lv, "peak": peak_lv,
                         "frac": ach_lv / peak_lv if peak_lv > 0 else None,
                         "traffic": pmc.get("hbm_bytes_per_launch"),
                         "levels_per_launch": levels_launch, "us_per_level": (sec_launch * 1e6 / levels_launch) if levels_launch else None,
                         "peak_ns_per_level": r["probe"]["conflict_free_ns"],
                         "peak_source": "nlmc_probe_level_round in this run: 256 workgroups x 16 waves, barrier + 8 LDS byte gathers in "
                                        "flight + 1 LDS write per round, bank-conflict-free addresses",
                         "random_gather_round_ns": r["probe"]["random_gather_ns"],
                         "frac_vs_random_gather_round": ach_lv * r["probe"]["random_gather_ns"] * 1e-9,
                         "kernel": kernel, "us_per_launch": ms_launch * 1e3,
                         "launches_timed": tm["launches_timed"], "launches_total": tm["launches_sweep"],
                         "note": "a chain is one workgroup that advances one level of its schedule per barrier round; the level's "
                                 "dependent chain (barrier -> LDS gather -> sum -> decide -> LDS write -> barrier) is the floor of a "
                                 "round; everything else (schedule stream from L2, Philox, energy) must hide behind it.  HBM, L2 "
                                 "and VALU fractions of the same launch: roofline_algorithmic.hbm_measured_frac, roofline_l2, "
                                 "roofline_issue"},
            # contract line: ALGORITHMIC bytes over the launch time measured live with HIP events; `traffic` and
            # everything derived from it are STATIC figures of the committed PMC passes (counters cannot be read live)
            "roofline_algorithmic": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes_per_launch"),
                                     "hbm_measured_frac": (pmc["hbm_bytes_per_launch"] / sec_launch / 1e9 / HBM_PEAK_GBS)
                                     if (pmc.get("hbm_bytes_per_launch") and ms_launch > 0) else None,
                                     "static_from": pmc.get("source"), "bytes_per_update": bpu, "updates_per_launch": upd_launch,
                                     "algorithmic_bytes_per_launch": upd_launch * bpu,
                                     "note": "SURVEY 8d algorithmic bytes per update (63 B with 4-byte J, 91 B with fp64 J; no discount for "
                                             "rows shared by chains) over the HBM peak: > 1 because this design does not move those "
                                             "bytes (spins live in LDS, one level schedule serves all chains of a launch) -- not a bound; "
                                             "hbm_measured_frac = PMC traffic over the same launch time"},
        }
        if r["fused"]:
            # every workgroup (= chain = CU) streams the whole window schedule through its L1 from its XCD's L2
            out["roofline_l2"] = {"bound": "l2", "unit": "GB/s", "peak": L2_PEAK_GBS,
                                  "achieved": upd_launch * sched_bytes / sec_launch / 1e9 if ms_launch > 0 else 0.0,
                                  "frac": upd_launch * sched_bytes / sec_launch / 1e9 / L2_PEAK_GBS if ms_launch > 0 else 0.0,
                                  "bytes_per_update": sched_bytes, "schedule_positions_per_update": positions_per_update}
        if pmc.get("valu_wave_insts_per_launch") and ms_launch > 0:
            lane_insts = pmc["valu_wave_insts_per_launch"] * 64.0
            t_issue = lane_insts / (N_CUS * VALU_LANES_PER_CU_CLK * CLOCK_GHZ * 1e9)
            out["roofline_issue"] = {"bound": "valu-issue", "unit": "lane-instructions/s",
                                     "peak": N_CUS * VALU_LANES_PER_CU_CLK * CLOCK_GHZ * 1e9,
                                     "achieved": lane_insts / sec_launch, "frac": t_issue / sec_launch,
                                     "valu_lane_insts_per_update": lane_insts / upd_launch,
                                     "valu_lane_insts_per_schedule_position": lane_insts / upd_launch / positions_per_update,
                                     "lds_bank_conflict_frac": pmc.get("lds_bank_conflict_frac"),
                                     "wait_any_frac": pmc.get("wait_any_frac"), "static_from": pmc.get("source")}
        return out

    first, second = a.headline, ("f32" if a.headline == "f64" else "f64")
    r = run_leg(first, a.steps, a.warmup)
    r2 = None if a.no_second_leg else run_leg(second, a.steps, a.warmup)

    if rank == 0:
        head = leg_report(r)
        out = {
            "metric": "spin-updates/s (replicas x spins x sweeps / s), NPT sweep + swap rounds, schedule construction included",
            "value": head["value"],
            "unit": "spin-updates/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": r["dt"] / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if a.strong else "weak",
            "vs_baseline": None,
            "dtype": head["dtype"],
            "data": "synthetic",
            "config": {"workload": "NPT heat-bath sweeps + replica exchange, sparse +-J spin glass (C4 per GPU)",
                       "spins": N_SPINS, "edges": 3 * N_SPINS, "replicas_per_gpu": r["count"],
                       "replicas_total": G, "sweeps_per_round": S_SWAP, "rounds_per_step": ROUNDS_PER_STEP,
                       "swap_pairs_per_round": n_pairs, "beta_ladder": [BETA_MIN, BETA_MAX], "rng": "philox4x32-10",
                       "order": "one permutation per sweep", "plan_chunk_rounds": PLAN_CHUNK_ROUNDS},
            "collective": r["collective"],         # the one collective of a round (None: a single process, nothing to gather)
            "ranks_seen": world,
            "plan_in_timed_region": True,
        }
        for k, v in head.items():
            if k not in ("value", "unit", "dtype"):
                out[k] = v
        if r2 is not None:
            out["fixed_point_f32" if second == "f32" else "f64_field"] = leg_report(r2)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(J, h, use_f64=(first == "f64"))
            out["cpu_baseline_numpy_path"] = numpy_path_baseline(J, h)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(use_f64=(first == "f64"))
        else:
            out["cpu_baseline"] = None
        if backend != "nccl":
            out["rehearsal"] = f"process group '{backend}', ranks share {torch.cuda.device_count()} GPU(s): code-path rehearsal, not a measurement"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
