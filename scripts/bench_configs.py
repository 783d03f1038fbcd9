"""Throughput of the other BASELINE.json configurations on one GPU (not the bench.py headline).
 C2  NMC phases, N=1e3, 64 restarts, fixed cluster set (first 5 % of the spins), 1e4 sweeps in total
 C3  NPT, N=1e3, 32 betas x 8 restarts = 256 chains, 1e4 sweeps, 100 swap rounds, 10 pairs per ladder and round
 C5  APT + iso-cluster moves, N=1e4, 32 betas x 8 sub-replicas = 256 chains, 100 rounds x 10 sweeps, 4 ICM pairs per beta
"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_product
from helpers import make_instance, init_spins
P = load_product()


def timed(fn):
    t0 = time.perf_counter(); r = fn(); return time.perf_counter() - t0, r


def c2():
    N, R, S_phase, cycles = 1000, 64, 333, 10             # 3 phases x 10 cycles x 333 = 9990 sweeps (+ 10 anneal)
    J, h = make_instance(N)
    eng = P.Engine(J, h, R)
    m = init_spins(R, N)
    cl = np.arange(N // 20)
    T = P.engine.fused_window(S_phase)
    eng.set_spins(m); eng.sweep_philox(10, 1, beta=3.0)     # warm-up
    def run():
        nonlocal m
        sw = 10
        if T and not os.environ.get("C2_PLAIN"):       # fused windows with running minimum + argmin state, planned at once
            eng.plan_philox_fused(sw, 3 * cycles * (S_phase // T), T, 7)
        for c in range(cycles):
            for kind in ("C", "NC", "ALL"):
                fl = None if kind == "ALL" else np.stack([P.hostlogic.phase_flags(N, m[r], cl, kind) for r in range(R)])
                eng.set_spins(m); eng.set_flags(fl, 20.0)
                o = eng.sweep_philox(S_phase, 7, sweep0=sw, beta=3.0, want_min=True, want_state=True)
                sw += S_phase; m = o["argmin_state"]
        return sw - 10
    dt, sweeps = timed(run)
    eng.close()
    return {"config": "C2 NMC phases N=1e3 x 64 restarts", "sweeps": sweeps, "seconds": dt, "updates_per_s": R * N * sweeps / dt}


def c2_full(lbp):
    """C2 as NMC.run_restarts runs it: anneal, then per cycle backbone inference (LBP + cluster growth) per restart and
    the three sweep phases -- nothing provided, so the Amdahl term of SURVEY.md 8 f-1 is inside the timed region."""
    import contextlib, io
    N, R, S_phase, cycles = 1000, 64, 333, 10
    J, h = make_instance(N)
    obj = P.NMC(J, h, rng="philox", seed=7, lbp=lbp)
    with contextlib.redirect_stdout(io.StringIO()):
        obj.run_restarts(2, 10, 10, 1)                     # warm-up: context, graph, kernels
        obj = P.NMC(J, h, rng="philox", seed=7, lbp=lbp)
        dt, (emin, _, _) = timed(lambda: obj.run_restarts(R, 10, S_phase, cycles, global_beta=3.0))
    sweeps = 10 + 3 * cycles * S_phase
    return {"config": f"C2 NMC.run_restarts N=1e3 x 64 restarts, backbone inference on the {lbp}", "sweeps": sweeps,
            "seconds": dt, "updates_per_s": R * N * sweeps / dt, "min_energy": float(emin.min())}


def c3():
    N, L, NL, S, rounds, pairs = 1000, 32, 8, 100, 100, 10
    J, h = make_instance(N)
    G = L * NL
    eng = P.Engine(J, h, G)
    eng.set_spins(init_spins(G, N)); eng.pt_init(np.geomspace(0.1, 3.0, L))
    planner = P.engine.RoundPlanner(eng, 0, rounds + 1, S, 3); planner._plan(0, True); eng.pt_plan(0, rounds + 1, 3, pairs)
    planner.sweep(0); eng.pt_swap_philox(0, 3, pairs, want_log=False); eng.energy()
    def run():
        for r in range(1, rounds + 1):
            planner.sweep(r)
            eng.pt_swap_philox(r, 3, pairs, want_log=False)
        return eng.energy()
    dt, E = timed(run)
    eng.close()
    return {"config": "C3 NPT N=1e3, 32 betas x 8 restarts", "sweeps": S * rounds, "seconds": dt,
            "updates_per_s": G * N * S * rounds / dt, "min_energy": float(E.min())}


def c5():
    N, R, K, S, rounds = 10_000, 32, 8, 10, 100
    J, h = make_instance(N)
    G = R * K
    eng = P.Engine(J, h, G)
    eng.set_spins(init_spins(G, N)); eng.pt_init(np.geomspace(0.05, 4.0, R))
    planner = P.engine.RoundPlanner(eng, 0, rounds + 1, S, 5, budget_bytes=8 << 30); planner._plan(0, True); eng.pt_plan(0, rounds + 1, 5, 10)
    def one(r):
        planner.sweep(r)
        eng.icm_round_ladders(r, 5, True)
        eng.pt_swap_philox(r, 5, 10, want_log=False)
    one(0); eng.energy()
    def run():
        for r in range(1, rounds + 1):
            one(r)
        return eng.energy()
    dt, E = timed(run)
    eng.close()
    return {"config": "C5 APT+ICM N=1e4, 32 betas x 8 sub-replicas", "sweeps": S * rounds, "seconds": dt,
            "updates_per_s": G * N * S * rounds / dt, "min_energy": float(E.min())}


if __name__ == "__main__":
    want = os.environ.get("CONFIGS", "c2,c3,c5,c2_full").split(",")
    for f in (c2, c3, c5):
        if f.__name__ in want:
            print(json.dumps(f()), flush=True)
    if "c2_full" in want:                # (CONFIGS=...,c2_full_host adds the bit-exact host inference: ~15 s)
        print(json.dumps(c2_full("device")), flush=True)
    if "c2_full_host" in want:
        print(json.dumps(c2_full("host")), flush=True)
