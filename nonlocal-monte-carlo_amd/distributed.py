"""Replica-sharded parallel tempering across the GPUs of one node: one process per GPU (`torch.distributed`,
backend "nccl" = RCCL over xGMI; "gloo" for CPU rehearsal with an injected engine).

Sharding: the global set of chains is cut into contiguous blocks, one per rank; a swap round
needs only the scalar energies, so the single collective is ONE all-gather of `n_local` float64 per rank and round.
Every rank then evaluates the identical Philox-keyed swap decision on the full energy vector and updates its replica of
the (tiny) slot table -- swaps are label exchanges, no configuration ever crosses a link.  Because every random
number is a pure function of (seed, global chain id, sweep, spin), the trajectory is bit-identical for any number
of ranks.
"""
import os
import sys

import numpy as np

from .engine import Engine, RoundPlanner


# Sweep indices (Philox counters) of the NMC phases that marked slots run inside a round: a range of their own, disjoint from the
# plain sweeps' whatever the number of rounds -- so rounds driven with and without a plan draw the same numbers (the
# unplanned fallback used to reuse the next round's plain indices).
NMC_SWEEP_SPACE = 1 << 31


def block_partition(n_global, world, rank):
    """Contiguous block of chains owned by `rank` (first `n_global % world` ranks get one extra)."""
    q, r = divmod(int(n_global), int(world))
    count = q + (1 if rank < r else 0)
    base = rank * q + min(rank, r)
    return base, count


class ShardedTempering:
    """Drives sweep rounds + swap rounds of a sharded ladder set.

    make_engine(inst, n_chains, chain_base, n_chains_global) -> object with the Engine interface used below;
    `dist` is torch.distributed (already initialised) or None for a single process; `torch` the torch module."""

    def __init__(self, make_engine, inst, beta_list, n_chains_global, seed, n_pairs, torch=None, dist=None,
                 device=None, precision="f32"):
        self.torch, self.dist = torch, dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.G = int(n_chains_global)
        self.base, self.count = block_partition(self.G, self.world, self.rank)
        if self.G % self.world != 0 and self.world > 1:
            raise ValueError("all_gather_into_tensor needs equal shards: n_chains_global % world_size != 0")
        self.seed, self.n_pairs, self.precision = int(seed), int(n_pairs), precision
        self.eng = make_engine(inst, self.count, self.base, self.G)
        self.eng_betas = np.asarray(beta_list, dtype=np.float64)
        self.eng.pt_init(self.eng_betas)
        self._lt = None
        self.sweeps_done = 0
        self.rounds_done = 0
        self.collective = dist is not None          # also with a single rank: the launcher path stays exercised
        self.lib_collective = False
        if self.collective:
            # The all-gather of a round is issued by the LIBRARY (ncclAllGather from libnlmc_hip.so on the stream its kernels run on:
            # no hand-over between c10d's collective stream and the kernels' stream, ~10 us per round).  Every rank probes librccl
            # and the ranks agree before anybody enters ncclCommInitRank; the unique id travels through the process group.
            self.lib_collective = agree_on_library_collectives(self.eng, torch, dist, device, want=self.n_pairs > 0)
        self._check_every = int(os.environ.get("NLMC_COMM_CHECK_EVERY", "256"))
        self._timeout_ms = int(float(os.environ.get("NLMC_COMM_TIMEOUT_S", "120")) * 1000)
        # a gloo process group (rehearsals with several ranks on one GPU, which RCCL refuses): collectives on host tensors, the
        # energies staged through the host every round
        self._host_stage = (self.collective and not self.lib_collective and dist.get_backend() == "gloo"
                            and str(device) != "cpu" and device is not None)
        if self.collective and not self.lib_collective:
            self.e_local = torch.empty(self.count, dtype=torch.float64, device=device)
            self.e_all = torch.empty(self.G, dtype=torch.float64, device=device)
            if self._host_stage:
                self.e_local_h = torch.empty(self.count, dtype=torch.float64)
                self.e_all_h = torch.empty(self.G, dtype=torch.float64)
            # the sweep kernels write their chains' energies straight into the all-gather's send buffer
            self._sink = hasattr(self.eng, "set_energy_sink")
            if self._sink:
                self.eng.set_energy_sink(self.e_local.data_ptr())

    def set_spins(self, spins_global):
        self.eng.set_spins(np.asarray(spins_global)[self.base:self.base + self.count])

    def plan(self, n_sweeps, n_rounds=0, chunk_rounds=None, lazy=False, reserve_rounds=0):
        """Level schedules of the next n_sweeps sweeps and pair selections of the next n_rounds swap rounds (both depend
        on the RNG only).  They are built a bounded chunk of rounds at a time (`chunk_rounds`, and a memory budget);
        with `lazy` nothing is built here: every chunk, the first one included, is planned by the round() that first
        needs it, so that the planning work lies inside whatever the caller times (the reference draws its permutation
        inside the sweep loop, NMC/nmc.py:62-71).  `reserve_rounds`: allocate (only allocate) plan buffers for chunks of
        that many rounds now."""
        self._planner = None
        if self._lt is not None:
            self._lt.sweeps_done, self._lt.rounds_done = self.sweeps_done, self.rounds_done
            self._lt.plan(n_sweeps, n_rounds, chunk_rounds=chunk_rounds)
            return
        if n_rounds > 0 and n_sweeps % n_rounds == 0 and hasattr(self.eng, "plan_philox_fused"):
            # rounds of equal length: fused-window level lists where the instance qualifies (same bits, fuller levels)
            self._planner = RoundPlanner(self.eng, self.sweeps_done, n_rounds, n_sweeps // n_rounds, self.seed,
                                         precision=self.precision, budget_bytes=16 << 30, chunk_rounds=chunk_rounds,
                                         pt_pairs=self.n_pairs, pt_round0=self.rounds_done)
            self._planner_round0 = self.rounds_done
            if reserve_rounds and self._planner.window and hasattr(self.eng, "plan_reserve_fused"):
                self.eng.plan_reserve_fused(int(reserve_rounds) * (self._planner.S // self._planner.window), self._planner.window)
            if not lazy:
                self._planner._plan(0, True)
            return
        self.eng.plan_philox(self.sweeps_done, n_sweeps, self.seed, precision=self.precision)
        if n_rounds > 0 and self.n_pairs > 0 and hasattr(self.eng, "pt_plan"):
            self.eng.pt_plan(self.rounds_done, n_rounds, self.seed, self.n_pairs)

    def configure_nmc(self, *args, **kw):
        """NMC_task slots (NPT/npt.py:622-647) under the one-process-per-GPU driver: every rank must own WHOLE ladders (restarts
        sharded over the ranks) -- the chains on marked slots are then a fixed number per rank, each rank decides its own ladders'
        swaps on the device (same Philox keys as one context holding everything) and a round needs no collective at all.  The
        rounds are driven by a LocalTempering over this rank's engine: same bits as LocalTempering over all contexts in one process."""
        L = self.eng.ladder_len
        if self.base % L or self.count % L:
            raise NotImplementedError("NMC slots under a ladder that is cut across ranks: shard whole ladders (restarts % ranks == 0)")
        self._lt = LocalTempering(None, self.eng_betas, self.G, self.seed, self.n_pairs, [0], precision=self.precision,
                                  engines=[self.eng], parts=[(self.base, self.count)])
        self._lt.sweeps_done, self._lt.rounds_done = self.sweeps_done, self.rounds_done
        self._lt.configure_nmc(*args, **kw)

    def round(self, n_sweeps, want_log=False, **outputs):
        """`n_sweeps` sweeps of every local chain at its ladder temperature, then one swap attempt round.  `outputs` (record_stride /
        want_* of sweep_philox): the sweeps' outputs are left in self.last_outputs."""
        if getattr(self, "_lt", None) is not None:           # rounds with NMC slots (whole ladders per rank): no collective
            self.last_outputs = self._lt.round(n_sweeps, **outputs)[0]
            self.sweeps_done, self.rounds_done = self._lt.sweeps_done, self._lt.rounds_done
            return None
        pl = getattr(self, "_planner", None)
        if pl is not None and n_sweeps == pl.S and 0 <= self.rounds_done - self._planner_round0 < pl.R:
            self.last_outputs = pl.sweep(self.rounds_done - self._planner_round0, **outputs)
        else:
            self.last_outputs = self.eng.sweep_philox(n_sweeps, self.seed, sweep0=self.sweeps_done, beta=None, precision=self.precision, **outputs)
        self.sweeps_done += n_sweeps
        log = None
        if self.n_pairs > 0:
            if self.lib_collective:
                log = self.eng.pt_swap_philox_collective(self.rounds_done, self.seed, self.n_pairs, refresh_energies=n_sweeps == 0,
                                                         want_log=want_log)
                if self._check_every and (self.rounds_done + 1) % self._check_every == 0:
                    self.eng.comm_check(self._timeout_ms)            # RCCL async error or a lost rank: abort, RuntimeError (never a hang)
            elif self.collective:
                if not self._sink or n_sweeps == 0:
                    self.eng.energy_dev(self.e_local.data_ptr())       # tracked energies -> device/host buffer
                if self._host_stage:
                    self.e_local_h.copy_(self.e_local)
                    self.dist.all_gather_into_tensor(self.e_all_h, self.e_local_h)
                    self.e_all.copy_(self.e_all_h)
                else:
                    self.dist.all_gather_into_tensor(self.e_all, self.e_local)   # the ONE collective of the round
                log = self.eng.pt_swap_philox(self.rounds_done, self.seed, self.n_pairs,
                                              energies_all_dev=self.e_all.data_ptr(), want_log=want_log)
            else:
                log = self.eng.pt_swap_philox(self.rounds_done, self.seed, self.n_pairs, want_log=want_log)
        self.rounds_done += 1
        return log

    def run_rounds(self, n_rounds, n_sweeps, persistent=None):
        """`n_rounds` rounds of `n_sweeps` sweeps + one swap round each.  `persistent` (default: the environment variable
        NLMC_PERSISTENT): a context that owns whole ladders and needs no collective runs a planned chunk of rounds per launch
        (k_rounds_fused: the chains stay in LDS between the rounds, one grid-wide meeting per round; include/nlmc.h:
        nlmc_pt_rounds_fused) -- same bits as round() called n_rounds times, which is what everything else falls back to.  Off by
        default: measured at the bench shape it costs what a launch per round costs (117.3 vs 117.6 us per round: what it saves in
        launches and spin I/O the grid-wide meeting and the per-round bookkeeping take back)."""
        done = 0
        pl = getattr(self, "_planner", None)
        if persistent is None:
            persistent = bool(os.environ.get("NLMC_PERSISTENT"))
        # default where it applies (one process, whole ladders, fused windows of one round): a sweep launch decides the PREVIOUS
        # round's swap in its prologue (nlmc_pt_rounds_deferred: one launch per round instead of two; NLMC_NO_DEFERRED switches it off)
        eligible = (pl is not None and self._lt is None and not self.collective and self.n_pairs > 0 and pl.window == n_sweeps == pl.S
                    and hasattr(self.eng, "pt_rounds_fused") and not getattr(self, "_persistent_refused", False))
        can = eligible
        batch = self.eng.pt_rounds_fused if persistent else getattr(self.eng, "pt_rounds_deferred", None)
        if batch is None or (not persistent and os.environ.get("NLMC_NO_DEFERRED")):
            can = False
        while done < n_rounds:
            ii = self.rounds_done - (self._planner_round0 if pl is not None else 0)
            if can and 0 <= ii < pl.R:
                if not (pl._fused_from <= ii < pl._fused_to):
                    pl._plan(ii, True)                      # this chunk's schedules and pair selections (inside whatever is timed)
                if pl._fused_from <= ii < pl._fused_to:
                    k = min(pl._fused_to - ii, n_rounds - done)
                    if batch(k, n_sweeps, self.seed, self.sweeps_done, self.rounds_done, self.n_pairs, precision=self.precision):
                        self.sweeps_done += k * n_sweeps
                        self.rounds_done += k
                        done += k
                        if persistent:
                            self.persistent_rounds = getattr(self, "persistent_rounds", 0) + k
                        else:
                            self.deferred_rounds = getattr(self, "deferred_rounds", 0) + k
                        continue
                    self._persistent_refused = True          # stop asking; the reason is in eng.rounds_fused_refusal
                can = False
            self.round(n_sweeps)
            done += 1

    def gather_spins(self):
        """All chains' configurations on every rank (read-out only; not part of a round)."""
        loc = self.eng.get_spins()
        if not self.collective:
            return loc
        dev = self.torch.device("cuda", self.torch.cuda.current_device()) if self.lib_collective else self.e_all.device
        if self._host_stage:
            dev = "cpu"
        t = self.torch.from_numpy(loc.astype(np.int8)).to(dev)
        out = self.torch.empty((self.G, loc.shape[1]), dtype=self.torch.int8, device=dev)
        self.dist.all_gather_into_tensor(out, t)
        return out.cpu().numpy()

    def close(self):
        try:
            if self.lib_collective:
                self.eng.comm_check(self._timeout_ms)
            if self._lt is not None:
                self._lt.check()
            elif self.n_pairs > 0 and hasattr(self.eng, "pt_check"):
                self.eng.pt_check()          # a swap round whose pair selection ran out raises here at the latest
        finally:
            self.eng.close()


class LocalTempering:
    """The same sharded tempering driven by ONE process over several contexts -- one per entry of `device_ids`
    (NPT.run(device_ids=...), APT_ICM.run(device_ids=...); the reference's knob is num_cores, NPT/npt.py:535-539,616).
    Chains are cut into contiguous blocks like ShardedTempering.  Blocks of whole ladders (restarts >= contexts) decide their
    own ladders' swaps on the device and never meet: the contexts run out of step, which is also how `device_ids=[0] * k`
    shortens a round whose cost is its slowest chain (NMC slots: the backbone inference).  Blocks that cut a ladder pass the
    tracked energies through host memory every round (G float64) and every context takes the identical Philox-keyed
    decision.  Either way the trajectory equals the single-context one bit for bit.  For one GPU per process over RCCL use
    ShardedTempering."""

    def __init__(self, inst, beta_list, n_chains_global, seed, n_pairs, device_ids, precision="f32", engine_factory=None,
                 engines=None, parts=None):
        """`parts` (optional): the blocks [(chain_base, count)] this process drives instead of an even cut over device_ids -- one
        rank's share under a launcher; `engines`: contexts that exist already (they must have had pt_init; not closed here... they
        ARE closed by close(): hand over ownership)."""
        self.G, self.seed, self.n_pairs, self.precision = int(n_chains_global), int(seed), int(n_pairs), precision
        devs = list(device_ids)
        self.parts = list(parts) if parts is not None else [block_partition(self.G, len(devs), r) for r in range(len(devs))]
        L = len(np.asarray(beta_list).reshape(-1))
        # every context owns whole ladders: each decides its own ladders' swaps from its own tracked energies (same Philox keys
        # as one context holding everything), nothing passes through the host and the contexts run out of step with each other
        self.whole_ladders = all(base % L == 0 and count % L == 0 for base, count in self.parts)
        self.engs = list(engines) if engines is not None else []
        self.nmc_sweeps_done = 0
        try:
            for d, (base, count) in (zip(devs, self.parts) if engines is None else []):
                if engine_factory is not None:           # (tests: the CPU double of tests/fake_engine.py)
                    e = engine_factory(inst, count, base, self.G)
                else:
                    e = Engine(inst, None, count, device=int(d), chain_base=base, n_chains_global=self.G,
                               own_stream=len(devs) > 1)
                e.pt_init(np.asarray(beta_list, dtype=np.float64))
                self.engs.append(e)
        except Exception:
            self.close()
            raise
        self.sweeps_done = self.rounds_done = 0
        self._planners = None
        self.nmc = None
        self._nmc_planners = None

    def set_spins(self, spins_global):
        for e, (base, count) in zip(self.engs, self.parts):
            e.set_spins(np.asarray(spins_global)[base:base + count])

    def configure_nmc(self, doNMC, phases, sweeps_per_phase, global_beta, temp_x, epsilon, lambdas, tolerance, max_iterations,
                      sat, thresholds, M_skip=1):
        """Temperature slots with doNMC run NMC_task instead of MCMC_task every round (NPT/npt.py:622-647): backbone inference
        seeded with the chain's state, then `phases` ("C" / "NC" / "ALL") of `sweeps_per_phase` sweeps at `global_beta`,
        each starting from the argmin-energy configuration of the one before (NPT/npt.py:357-477).  Everything stays on the
        device (include/nlmc.h: nlmc_pt_mark_slots ... nlmc_set_phase)."""
        marks = np.asarray(doNMC).astype(bool)
        if not marks.any():
            self.nmc = None
            return
        for e in self.engs:
            e.mark_slots(marks)
            e.overlap_subsets(not os.environ.get("NLMC_NO_OVERLAP"))      # plain chains' sweeps beside the NMC chains' work
        self.nmc = dict(phases=list(phases), S=int(sweeps_per_phase), beta=float(global_beta), temp_x=float(temp_x),
                        epsilon=np.asarray(epsilon, dtype=np.float64), lambdas=np.asarray(lambdas, dtype=np.float64),
                        tol=float(tolerance), max_it=int(max_iterations), sat=float(sat),
                        thresholds=np.asarray(thresholds, dtype=np.float64), M_skip=max(1, int(M_skip)))
        # M_skip > 1 (NPT/npt.py:434-437: M = M[:, ::M_skip] before the energies and the argmin): the phases' running minimum looks at
        # every M_skip-th sweep only (nlmc_track_minimum(M_skip)), their recorded traces are strided the same way

    def sweeps_per_round(self, n_sweeps):
        """Plain sweep indices (RNG counters) one round of `n_sweeps` sweeps consumes.  The NMC phases of the marked slots draw from
        a range of their own (NMC_SWEEP_SPACE + nmc_sweeps_done, advanced by phases x sweeps_per_phase per round)."""
        return n_sweeps

    def plan(self, n_sweeps, n_rounds, chunk_rounds=None):
        S = n_sweeps // max(1, n_rounds)
        self._planners = [RoundPlanner(e, self.sweeps_done, n_rounds, S, self.seed,
                                       precision=self.precision, budget_bytes=8 << 30, chunk_rounds=chunk_rounds,
                                       pt_pairs=self.n_pairs, pt_round0=self.rounds_done, slot=0, fused_outputs=True) for e in self.engs]
        self._planner_round0 = self.rounds_done
        self._nmc_planners = None
        if self.nmc:
            # the NMC phases draw from their own range of sweep indices, behind the plain sweeps of all planned rounds: both
            # ranges are contiguous over the rounds, so each is served by one fused-window plan (slots 0 and 1)
            n_ph = len(self.nmc["phases"])
            self._nmc_sweep0 = NMC_SWEEP_SPACE + self.nmc_sweeps_done
            self._nmc_planners = [RoundPlanner(e, self._nmc_sweep0, n_rounds * n_ph, self.nmc["S"], self.seed,
                                               precision=self.precision, budget_bytes=4 << 30,
                                               chunk_rounds=None if chunk_rounds is None else chunk_rounds * n_ph,
                                               slot=1, beta=self.nmc["beta"], fused_outputs=True) for e in self.engs]
            self._planned_rounds = n_rounds

    def log_begin(self, n_rounds):
        for e in self.engs:
            e.pt_log_begin(self.rounds_done, n_rounds, self.n_pairs)

    def round(self, n_sweeps, energy_columns=0, **outputs):
        """Sweeps of every block (launched back to back: the devices work concurrently), then one swap round.  Returns
        the per-context sweep outputs; with NMC slots configured (configure_nmc) every entry is a dict
        {"plain": outputs of the chains on plain slots, "plain_chains": their local ids, "nmc": [outputs per phase],
        "nmc_chains": ids, "plain_energy_columns": fp64 energies of the first `energy_columns` recorded columns}."""
        ii = self.rounds_done - (self._planner_round0 if self._planners else 0)
        outs = []
        wants = any(outputs.get(k) for k in ("record_stride", "want_energy", "want_min", "want_state"))
        for k, e in enumerate(self.engs):
            if self.nmc:
                e.set_phase("ALL")                   # (not set_flags(None): temp_x stays, so do the uploaded temperature tables)
                e.select("unmarked")
            if self._planners and 0 <= ii < self._planners[k].R and n_sweeps == self._planners[k].S:
                o = self._planners[k].sweep(ii, **outputs)
            else:
                o = e.sweep_philox(n_sweeps, self.seed, sweep0=self.sweeps_done, beta=None, precision=self.precision, **outputs)
            if not self.nmc:
                outs.append(o)
                continue
            rec = dict(plain=o, nmc=[])
            if wants:
                rec["plain_chains"] = e.subset()
                if energy_columns and outputs.get("record_stride"):
                    rec["plain_energy_columns"] = e.energy_of_recorded(energy_columns)
            q = self.nmc
            e.select("marked")
            if wants:
                rec["nmc_chains"] = e.subset()
            e.backbone_clusters(q["epsilon"], q["lambdas"], q["beta"], q["tol"], q["max_it"], q["sat"], q["thresholds"])
            e.track_minimum(True, stride=q["M_skip"])
            n_ph = len(q["phases"])
            out_nmc = dict(outputs, record_stride=q["M_skip"]) if outputs.get("record_stride") else outputs
            planned = self._nmc_planners is not None and 0 <= ii < self._planned_rounds
            for p, kind in enumerate(q["phases"]):
                if p > 0:
                    e.adopt_best()                   # NPT/npt.py:436-437,454-455: the next phase starts from the argmin column
                e.set_phase(kind, q["temp_x"])
                if planned:
                    rec["nmc"].append(self._nmc_planners[k].sweep(ii * n_ph + p, **out_nmc))
                else:
                    rec["nmc"].append(e.sweep_philox(q["S"], self.seed, sweep0=NMC_SWEEP_SPACE + self.nmc_sweeps_done + p * q["S"],
                                                     beta=q["beta"], precision=self.precision, **out_nmc))
            e.track_minimum(False)
            e.set_phase("ALL")
            e.select("all")
            outs.append(rec)
        self.sweeps_done += n_sweeps
        if self.nmc:
            self.nmc_sweeps_done += len(self.nmc["phases"]) * self.nmc["S"]
        if self.n_pairs > 0:
            if len(self.engs) == 1 or self.whole_ladders:
                for e in self.engs:
                    e.pt_swap_philox(self.rounds_done, self.seed, self.n_pairs, want_log=False)
            else:
                E = np.concatenate([e.energy_tracked() for e in self.engs])
                for e in self.engs:
                    e.pt_swap_philox_host(self.rounds_done, self.seed, self.n_pairs, E)
        self.rounds_done += 1
        return outs

    def slots(self):
        if len(self.engs) == 1 or not self.whole_ladders:
            return self.engs[0].pt_slots()
        return np.concatenate([e.pt_slots()[base:base + count] for e, (base, count) in zip(self.engs, self.parts)])

    def swap_log(self):
        if len(self.engs) == 1 or not self.whole_ladders:
            return self.engs[0].pt_log_read()
        L = self.engs[0].ladder_len
        pairs, acc = self.engs[0].pt_log_read()
        for e, (base, count) in list(zip(self.engs, self.parts))[1:]:          # every context logged its own ladders' rows
            p, a = e.pt_log_read()
            pairs[:, base // L:(base + count) // L] = p[:, base // L:(base + count) // L]
            acc[:, base // L:(base + count) // L] = a[:, base // L:(base + count) // L]
        return pairs, acc

    def gather_spins(self):
        return np.concatenate([e.get_spins() for e in self.engs])

    def check(self):
        if self.n_pairs > 0:
            for e in self.engs:
                e.pt_check()
        if self.nmc:
            for e in self.engs:
                e.backbone_check()      # ValueError('LBP diverged at initial lambda ...') of NPT/npt.py:178-180

    def close(self):
        for e in self.engs:
            e.close()
        self.engs = []


def agree_on_library_collectives(eng, torch, dist, device, want):
    """Every rank probes librccl (nlmc_comm_probe) and the ranks agree -- an all-reduce(MIN) of the answers over the process group
    that is already up -- BEFORE anybody calls nlmc_comm_init: a rank that cannot load the library must not leave the others
    waiting inside ncclCommInitRank.  Returns True when the library-issued collectives are on for ALL ranks; the
    communicator's unique id then travels through the same process group."""
    ok = bool(want) and hasattr(eng, "comm_init") and dist.get_backend() == "nccl" and not os.environ.get("NLMC_TORCH_COLLECTIVE")
    if ok:
        try:
            ok = bool(type(eng).comm_probe())
        except Exception as ex:  # noqa: BLE001
            print(f"[nlmc] librccl probe failed ({ex})", file=sys.stderr)
            ok = False
    if dist.get_backend() == "gloo":
        device = "cpu"                              # (the flags below are host tensors for a gloo group)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        if want and dist.get_rank() == 0 and dist.get_backend() == "nccl" and not os.environ.get("NLMC_TORCH_COLLECTIVE"):
            print("[nlmc] library-issued collectives unavailable on some rank; using torch.distributed", file=sys.stderr)
        return False
    uid = torch.zeros(129, dtype=torch.uint8, device=device)           # [128] = 1: rank 0 could create an id
    if dist.get_rank() == 0:
        try:
            uid[:128].copy_(torch.from_numpy(type(eng).comm_unique_id()))
            uid[128] = 1
        except (NotImplementedError, RuntimeError) as ex:
            print(f"[nlmc] ncclGetUniqueId failed ({ex}); using torch.distributed", file=sys.stderr)
    dist.broadcast(uid, 0)
    if int(uid[128].item()) != 1:
        return False
    # comm_init itself may still fail on one rank (out of memory, ...): agree once more, and drop the communicator everywhere if so
    err = None
    try:
        eng.comm_init(uid[:128].cpu().numpy(), dist.get_world_size(), dist.get_rank())
    except Exception as ex:  # noqa: BLE001
        err = ex
    flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        raise RuntimeError(f"nlmc_comm_init failed on some rank ({err}); aborting the job") from err
    return True


class SlotShardedAPT:
    """APT + iso-cluster moves (NPT/apt_ICM.py:195-285) with the temperature ladder cut into slot blocks (SURVEY.md section 8e:
    all sub-replicas of a temperature on one GPU; include/nlmc.h: nlmc_apt_shard, csrc/nlmc_apt.h).

    R temperatures x K sub-replicas; shard w of W owns the slots [w R/W, (w+1) R/W) of all K sub-replica ladders.  Per round:
    sweeps at the slots' temperatures -> Houdayer moves between the randomly paired sub-replicas of every local slot (k_icm_round,
    nothing leaves the GPU) -> swap round: ONE all-gather of K R/W tracked energies per shard + ONE exchange of the K boundary
    configurations with each neighbour shard; every shard takes the identical Philox-keyed decision; pairs inside a block are label
    exchanges, accepted pairs across a block boundary move the two configurations.  Random numbers are keyed by (sub-replica,
    global slot): the states on every (sub-replica, slot) are bit-identical for any W.

    Three transports, one protocol:
      * `dist` (torch.distributed, one process per GPU): the library issues the collectives itself over RCCL on the kernels'
        stream (apt_swap_collective) when every rank can; otherwise torch.distributed moves the same buffers (gloo on CPUs);
      * `device_ids` (one process, W = len(device_ids) contexts): the same data through host memory;
      * neither: one context, W = 1.
    make_engine(inst, n_chains, chain_base, n_chains_global) as for ShardedTempering (tests inject the oracle-backed double)."""

    def __init__(self, make_engine, inst, beta_list, n_subreplicas, seed, n_pairs, torch=None, dist=None, device=None,
                 precision="f32", katzgraber=True, device_ids=None):
        self.torch, self.dist, self.device = torch, dist, device
        self.betas = np.asarray(beta_list, dtype=np.float64).reshape(-1)
        self.R, self.K = int(self.betas.shape[0]), int(n_subreplicas)
        self.seed, self.n_pairs, self.precision, self.katz = int(seed), int(n_pairs), precision, bool(katzgraber)
        if dist is not None:
            self.W, ranks = dist.get_world_size(), [dist.get_rank()]
        else:
            self.W = len(device_ids) if device_ids else 1
            ranks = list(range(self.W))
        if self.R % self.W:
            raise ValueError(f"{self.R} temperatures do not split evenly over {self.W} shards")
        self.Rw = self.R // self.W
        self.ranks, self.engs = ranks, []
        try:
            for i, w in enumerate(ranks):
                e = make_engine(inst, self.K * self.Rw, 0, self.K * self.Rw) if dist is not None or not device_ids else \
                    make_engine(inst, self.K * self.Rw, 0, self.K * self.Rw, device_ids[i])
                self.engs.append(e)
                e.pt_init(self.betas[w * self.Rw:(w + 1) * self.Rw])
                e.apt_shard(self.betas, self.W, w)
        except Exception:
            self.close()
            raise
        self.sweeps_done = self.rounds_done = 0
        self._planners = None
        self.lib_collective = False
        if dist is not None and self.W > 1:
            self.lib_collective = agree_on_library_collectives(self.engs[0], torch, dist, device, want=self.n_pairs > 0)
        self._check_every = int(os.environ.get("NLMC_COMM_CHECK_EVERY", "256"))
        self._timeout_ms = int(float(os.environ.get("NLMC_COMM_TIMEOUT_S", "120")) * 1000)

    # -- states by (sub-replica, global slot) -------------------------------------------------------------------------------
    def set_spins_by_slot(self, spins):
        """spins [K, R, N]: the configuration on (sub-replica j, temperature slot r).  Slots are reset to the identity."""
        s = np.asarray(spins, dtype=np.int8).reshape(self.K, self.R, -1)
        for e, w in zip(self.engs, self.ranks):
            e.pt_set_slots((np.arange(self.K * self.Rw) % self.Rw).astype(np.int32))
            e.set_spins(np.ascontiguousarray(s[:, w * self.Rw:(w + 1) * self.Rw]).reshape(self.K * self.Rw, -1))

    def _local_by_slot(self, e):
        cfg, slots = e.get_spins(), e.pt_slots()
        out = np.empty((self.K, self.Rw, cfg.shape[1]), np.int8)
        out[np.arange(self.K * self.Rw) // self.Rw, slots] = cfg
        en = np.empty((self.K, self.Rw))
        en[np.arange(self.K * self.Rw) // self.Rw, slots] = e.energy_tracked()
        return out, en

    def gather_by_slot(self):
        """(configurations [K, R, N], tracked energies [K, R]) by (sub-replica, global slot), on every process (read-out only)."""
        loc = [self._local_by_slot(e) for e in self.engs]
        if self.dist is None or self.W == 1:
            return np.concatenate([x[0] for x in loc], axis=1), np.concatenate([x[1] for x in loc], axis=1)
        t, dev = self.torch, self.device
        cfg = t.from_numpy(loc[0][0]).to(dev)
        en = t.from_numpy(loc[0][1]).to(dev)
        cfgs = [t.empty_like(cfg) for _ in range(self.W)]
        ens = [t.empty_like(en) for _ in range(self.W)]
        self.dist.all_gather(cfgs, cfg)
        self.dist.all_gather(ens, en)
        return np.concatenate([x.cpu().numpy() for x in cfgs], axis=1), np.concatenate([x.cpu().numpy() for x in ens], axis=1)

    # -- rounds -----------------------------------------------------------------------------------------------------------
    def plan(self, n_rounds, sweeps_per_round, chunk_rounds=None, budget_bytes=8 << 30):
        """Level schedules and pair selections of the next rounds (RNG only), built a chunk of rounds at a time by the round that
        first needs them (RoundPlanner)."""
        self._planners = [RoundPlanner(e, self.sweeps_done, n_rounds, sweeps_per_round, self.seed, precision=self.precision,
                                       budget_bytes=budget_bytes, chunk_rounds=chunk_rounds, pt_pairs=self.n_pairs,
                                       pt_round0=self.rounds_done) for e in self.engs]
        self._planner_round0 = self.rounds_done

    def _swap_torch(self, want_log):
        """The round's exchange through torch.distributed (gloo on CPUs, or RCCL when the library cannot issue it itself): the same
        buffers as the library path -- K Rw int64 per rank all-gathered, K configurations to and from each neighbour."""
        t, dist, dev, e, w = self.torch, self.dist, self.device, self.engs[0], self.ranks[0]
        ef, lo, hi = e.apt_pack()
        mine = t.from_numpy(ef.reshape(-1)).to(dev)
        parts = [t.empty_like(mine) for _ in range(self.W)]
        dist.all_gather(parts, mine)
        recv_lo = t.empty((self.K, e.n), dtype=t.int8, device=dev) if w > 0 else None
        recv_hi = t.empty((self.K, e.n), dtype=t.int8, device=dev) if w + 1 < self.W else None
        ops = []
        if w > 0:
            ops += [dist.P2POp(dist.isend, t.from_numpy(lo).to(dev), w - 1), dist.P2POp(dist.irecv, recv_lo, w - 1)]
        if w + 1 < self.W:
            ops += [dist.P2POp(dist.isend, t.from_numpy(hi).to(dev), w + 1), dist.P2POp(dist.irecv, recv_hi, w + 1)]
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        ef_all = np.stack([p.cpu().numpy() for p in parts])
        return e.apt_swap_host(self.rounds_done, self.seed, self.n_pairs, ef_all,
                               None if recv_lo is None else recv_lo.cpu().numpy(), None if recv_hi is None else recv_hi.cpu().numpy(),
                               want_log=want_log)

    def round(self, n_sweeps, want_log=False, want_info=False, **outputs):
        """Sweeps, Houdayer moves, swap round of every local shard.  Returns (swap log or None, cluster info of the local shards or
        None).  `outputs` (record_stride / want_* of sweep_philox): the sweeps' outputs per local shard are left in
        self.last_outputs, with the slots the chains sat on during those sweeps in self.last_slots."""
        ii = self.rounds_done - (self._planner_round0 if self._planners else 0)
        self.last_outputs, self.last_slots = [], []
        for k, e in enumerate(self.engs):
            pl = self._planners[k] if self._planners else None
            if outputs:
                self.last_slots.append(e.pt_slots())
            if pl is not None and 0 <= ii < pl.R and n_sweeps == pl.S:
                o = pl.sweep(ii, **outputs)
            else:
                o = e.sweep_philox(n_sweeps, self.seed, sweep0=self.sweeps_done, beta=None, precision=self.precision, **outputs)
            self.last_outputs.append(o)
        self.sweeps_done += n_sweeps
        info = [e.icm_round_ladders(self.rounds_done, self.seed, self.katz, want_info=want_info) for e in self.engs]
        log = None
        if self.n_pairs > 0:
            if self.lib_collective:
                log = self.engs[0].apt_swap_collective(self.rounds_done, self.seed, self.n_pairs, want_log=want_log)
                if self._check_every and (self.rounds_done + 1) % self._check_every == 0:
                    self.engs[0].comm_check(self._timeout_ms)            # async error / lost rank -> abort, RuntimeError
            elif self.dist is not None and self.W > 1:
                log = self._swap_torch(want_log)
            elif self.W == 1:
                log = self.engs[0].apt_swap_collective(self.rounds_done, self.seed, self.n_pairs, want_log=want_log)
            else:
                packs = [e.apt_pack() for e in self.engs]
                ef_all = np.stack([p[0] for p in packs])
                for w, e in enumerate(self.engs):
                    log = e.apt_swap_host(self.rounds_done, self.seed, self.n_pairs, ef_all, packs[w - 1][2] if w > 0 else None,
                                          packs[w + 1][1] if w + 1 < self.W else None, want_log=want_log)
        self.rounds_done += 1
        return log, (info if want_info else None)

    def check(self):
        for e in self.engs:
            if self.n_pairs > 0 and hasattr(e, "pt_check"):
                e.pt_check()
        if self.lib_collective:
            self.engs[0].comm_check(self._timeout_ms)

    def close(self):
        for e in self.engs:
            e.close()
        self.engs = []


class ShardedAsLocal:
    """The slice of the LocalTempering interface NPT.run drives, over a ShardedTempering (one process per GPU, a ladder cut across
    the ranks: ONE all-gather of the energies per round).  `gather(x)`: rows of every rank's local array, in chain order."""

    def __init__(self, st):
        self.st, self.G, self.engs, self.parts, self.nmc = st, st.G, [st.eng], [(st.base, st.count)], None
        self.nmc_sweeps_done = 0

    sweeps_done = property(lambda self: self.st.sweeps_done, lambda self, v: setattr(self.st, "sweeps_done", v))
    rounds_done = property(lambda self: self.st.rounds_done)

    def configure_nmc(self, *a, **k):
        raise NotImplementedError("NMC slots under a ladder that is cut across ranks: shard whole ladders (num_restarts % ranks == 0)")

    def set_spins(self, spins_global):
        self.st.set_spins(spins_global)

    def sweeps_per_round(self, n_sweeps):
        return n_sweeps

    def plan(self, n_sweeps, n_rounds, chunk_rounds=None):
        self.st.plan(n_sweeps, n_rounds, chunk_rounds=chunk_rounds)

    def log_begin(self, n_rounds):
        self.st.eng.pt_log_begin(self.st.rounds_done, n_rounds, self.st.n_pairs)

    def round(self, n_sweeps, energy_columns=0, **outputs):
        self.st.round(n_sweeps, **outputs)
        return [self.st.last_outputs]

    def slots(self):
        return self.st.eng.pt_slots()              # the replicated table: every rank applied every decision

    def swap_log(self):
        return self.st.eng.pt_log_read()

    def gather(self, x):
        t, dist = self.st.torch, self.st.dist
        dev = "cpu" if dist.get_backend() == "gloo" else t.device("cuda", t.cuda.current_device())
        loc = t.from_numpy(np.ascontiguousarray(x)).to(dev)
        out = t.empty((self.st.world * loc.shape[0],) + tuple(loc.shape[1:]), dtype=loc.dtype, device=dev)
        dist.all_gather_into_tensor(out, loc)
        return out.cpu().numpy()

    def check(self):
        if self.st.n_pairs > 0:
            self.st.eng.pt_check()
        if self.st.lib_collective:
            self.st.eng.comm_check(self.st._timeout_ms)

    def close(self):
        self.st.eng.close()


def launcher_context():
    """(torch, torch.distributed, world, rank, local_rank) when this process is one rank of a `torch.distributed.run` job (RANK /
    WORLD_SIZE in the environment; a job of ONE rank counts: it takes the same path), else None.  The process group (backend "nccl"
    = RCCL) is brought up here if the caller has not done so; call this -- i.e. NPT.run -- before other GPU work of the process
    where possible."""
    try:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    except ValueError:
        return None
    if world < 1 or "RANK" not in os.environ:
        return None
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    return torch, dist, dist.get_world_size(), dist.get_rank(), local_rank


def all_reduce_np(torch, dist, arr):
    """Sum of a NumPy array over the ranks (read-out bookkeeping of NPT.run under a launcher); the tensor lives where the process
    group's backend wants it."""
    dev = "cpu" if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device())
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    dist.all_reduce(t)
    return t.cpu().numpy()
