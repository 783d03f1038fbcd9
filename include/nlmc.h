/*
 * nlmc.h -- C-ABI of the MI355X-native heat-bath sweep + replica-exchange engine (libnlmc_hip.so).
 *
 * The reference (usra-riacs/Nonlocal-Monte-Carlo) is pure Python/NumPy and has NO FFI boundary of its own;
 * its boundary is the Python class surface NMC(J,h).run / NPT(J,h).run / APT_ICM(J,h).run.  The entry points
 * below are what a ctypes binding inside those classes needs to replace the reference's inner loops.  Each
 * one names the reference lines it replaces (paths relative to the reference checkout).  Plain pointers and
 * sizes only; no torch types.  All calls are blocking with respect to their host-pointer outputs and are
 * stream-ordered on the HIP stream given at creation.  One context per host thread.
 *
 * Ownership: the caller owns every host buffer for the duration of the call only; the library owns all
 * device memory until nlmc_destroy.  Errors: 0 = ok, negative = failure (text via nlmc_last_error).
 * There is NO CPU fallback: without a HIP device every compute entry point fails with NLMC_ERR_HIP.
 */
#ifndef NLMC_H
#define NLMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nlmc_ctx nlmc_ctx;

enum {
    NLMC_OK = 0,
    NLMC_ERR_ARG = -1,         /* bad argument (ValueError on the Python side) */
    NLMC_ERR_HIP = -2,         /* HIP runtime failure / no device (RuntimeError) */
    NLMC_ERR_UNSUPPORTED = -3, /* size outside what this build handles */
    NLMC_ERR_STATE = -4        /* call order violated (e.g. PT call before nlmc_pt_init) */
};

/* arithmetic of the local field in philox mode.  NLMC_F32 (4-byte couplings): J and h are held in 24-bit fixed point,
 * Jq = rint(J 2^qs) (nlmc_field_scale), the field is an exact int32 sum and the acceptance is one f32 compare with a
 * logistic threshold drawn from 32 random bits -- the chain samples the Boltzmann law of (Jq, hq) 2^-qs exactly, with
 * |Jq 2^-qs - J| <= 2^-(qs+1) (nothing is lost for +-J / integer instances).  NLMC_F64: fp64 field in the reference's
 * summation order (NMC/nmc.py:86), base-2 logistic test on a 53-bit uniform. */
enum { NLMC_F32 = 0, NLMC_F64 = 1 };
enum { NLMC_ORDER_SHARED = 0, NLMC_ORDER_PER_CHAIN = 1 }; /* one permutation per sweep for all chains | per chain */

/* phase flags, one byte per (chain, spin): caller contract of NMC/nmc.py:377-381,398-401 */
enum { NLMC_SPIN_NORMAL = 0, NLMC_SPIN_SCALED = 1, NLMC_SPIN_FROZEN_UP = 2, NLMC_SPIN_FROZEN_DOWN = 3 };

#define NLMC_LDS_N 24576    /* up to here a chain's spins live in LDS (the fast kernels); longer chains keep them in global memory:
                             * sweeps (all three modes), energies, traces, replica exchange, the Houdayer move and the backbone
                             * inference work at any size with the same results spin for spin (csrc/nlmc_big.h); only the
                             * fused windows stay with the LDS kernels (nlmc_fused_modes reports 0, sweeps run order by order) */
#define NLMC_MAX_N 16777216 /* spins per chain (32-bit positions of the schedules, counters of the level histogram) */

/* Version of this interface; nlmc_abi_version() returns the value the library was built with and the binding refuses a
 * library whose value differs.  2 (round 3): nlmc_timing_total has a fifth out-pointer and nlmc_timing_reset's argument is
 * a sampling period (both since round 2), NLMC_F32 means 24-bit fixed-point couplings + logistic thresholds, chain
 * subsets / plan slots / device-side NMC hand-offs were added. */
#define NLMC_ABI_VERSION 3

int nlmc_abi_version(void);
int nlmc_device_count(void);

/* Build a context: uploads J (CSR, both triangles, sorted columns -- what scipy.sparse.csr_matrix(J) holds at
 * NMC/nmc.py:53) and h, allocates n_chains replicas.  chain_base / n_chains_global describe this rank's shard of
 * a replica set spread over several GPUs (single GPU: 0 / n_chains).  hip_stream may be NULL (default stream).
 * Replaces: the per-call `csr_matrix(J)` + process-pool pickling of self (NMC/nmc.py:53, NPT/npt.py:616-640). */
int nlmc_create(nlmc_ctx **out, int device, void *hip_stream, int n, int64_t nnz, const int32_t *rowptr,
                const int32_t *colidx, const double *vals, const double *h, int n_chains, int chain_base,
                int n_chains_global);
void nlmc_destroy(nlmc_ctx *ctx);
const char *nlmc_last_error(const nlmc_ctx *ctx); /* ctx may be NULL: error of the last failed nlmc_create */

/* Replica state, int8 +-1 (0 allowed), layout [n_chains][n].  Replaces m_start / M[:, -1] hand-offs
 * (NMC/nmc.py:50, NPT/npt.py:612,647). */
int nlmc_set_spins(nlmc_ctx *ctx, const int8_t *spins);
int nlmc_get_spins(nlmc_ctx *ctx, int8_t *spins);

/* Phase parameterisation (NMC/nmc.py:377-381,398-401; NPT/npt.py:406-414,425,441): flags [n_chains][n] or NULL
 * to clear.  SCALED spins see (J_k,: , h_k)/temp_x, FROZEN_UP/DOWN spins see h_k = +-10000. */
int nlmc_set_flags(nlmc_ctx *ctx, const uint8_t *flags, double temp_x);

/* E = -(m^T J m/2 + m^T h) of the current state of every local chain, fp64 (NMC/nmc.py:386, NPT/npt.py:31-45,
 * 657-658).  Also re-synchronises the engine's incremental fixed-point energies. */
int nlmc_energy(nlmc_ctx *ctx, double *out /*[n_chains]*/);
/* Same, result left in device memory (for an RCCL all-gather issued by the caller). */
int nlmc_energy_dev(nlmc_ctx *ctx, double *dev_out /*[n_chains], device pointer*/);
/* The incrementally tracked energies of the current states (no recomputation), to the host. */
int nlmc_energy_tracked(nlmc_ctx *ctx, double *out /*[n_chains]*/);
/* Persistent variant: every later sweep call also stores the tracked energies of its final states there (the send
 * buffer of the per-round all-gather, NPT/npt.py:657-658 energies) -- no extra launch.  NULL switches it off. */
int nlmc_set_energy_sink(nlmc_ctx *ctx, double *dev_out /*[n_chains], device pointer or NULL*/);

/* log2 of the fixed-point scale of the incrementally tracked energies (E_tracked = integer * 2^-scale). */
int nlmc_energy_scale(const nlmc_ctx *ctx);
/* qs of the NLMC_F32 path: couplings are held as rint(J 2^qs), |Jq| < 2^23, row sums < 2^31; qs <= energy scale <= qs+29. */
int nlmc_field_scale(const nlmc_ctx *ctx);

/* Energies of an arbitrary batch of configurations (trace read-out, NPT/npt.py:685-692). */
int nlmc_energy_of(nlmc_ctx *ctx, const int8_t *spins /*[count][n]*/, int64_t count, double *out /*[count]*/);
/* fp64 energies of the configurations the most recent sweep call recorded (out_spins), computed where they still lie on
 * the device: configurations first .. first+count-1 of every chain -> out [n_chains][count].  Replaces the per-replica
 * `replica_energy(M[...], k)` loop of NPT/npt.py:685-692 without sending the trace back.  NLMC_ERR_STATE when the last
 * call recorded fewer. */
int nlmc_energy_of_recorded(nlmc_ctx *ctx, int first, int count, double *out);

/* Outputs shared by both sweep entry points (every pointer nullable):
 *   out_spins  [n_chains][ceil(n_sweeps/record_stride)][n]  state after sweeps 0, record_stride, 2*record_stride, ...
 *                                                     (M[:, ::M_skip], NMC/nmc.py:89,390)
 *   out_energy [n_chains][n_sweeps]                   energy after every sweep (NMC/nmc.py:386-387)
 *   out_min_energy [n_chains], out_argmin [n_chains]  min / first argmin over this call's sweeps (NMC/nmc.py:394)
 *   out_argmin_state [n_chains][n]                    the argmin column (NMC/nmc.py:395)
 */

/* STREAM mode: random-permutation sequential heat-bath sweeps consuming a pre-drawn legacy stream
 * (np.random.permutation(N) + N x np.random.rand() per sweep, NMC/nmc.py:71,87), fp64, update rule
 * m_k = sign(tanh(beta x_k) - 2u + 1).  Replaces MCMC(): NMC/nmc.py:28-91 == NPT/npt.py:47-110,
 * NPT/apt_ICM.py:52-93, NPT/apt_preprocessor.py:33-74.
 *   perm, u   : [n_chains][n_sweeps][n]  (i-th visited spin, i-th uniform)
 *   beta      : table of inverse temperatures, element (c, t) at beta[c*chain_stride + t*sweep_stride] */
int nlmc_sweep_stream(nlmc_ctx *ctx, int n_sweeps, const int32_t *perm, const double *u, const double *beta,
                      int chain_stride, int sweep_stride, int record_stride, int8_t *out_spins, double *out_energy,
                      double *out_min_energy, int32_t *out_argmin, int8_t *out_argmin_state);

/* PHILOX mode (throughput): same Markov kernel, counter-based RNG generated on the device.  Sweep t of the run
 * (global index sweep0 + t) visits spins in ascending order of philox(k, t, group, ORDER) and spin k of chain c
 * draws one word of philox(k>>2, t, c, UNIFORM) (f32: 32 bits -> logistic threshold; f64: its high 27 bits + the high 26 bits of the same word of philox(k>>2, t, c, UNIFORM_LO) -> 53-bit uniform); results are a pure function of (seed, global chain id, sweep index, spin) and
 * therefore independent of how chains are sharded over GPUs.
 *   beta : as above, or NULL to take each chain's beta from the PT ladder (nlmc_pt_init). */
int nlmc_sweep_philox(nlmc_ctx *ctx, int precision, int order_mode, int n_sweeps, uint32_t sweep0, uint64_t seed,
                      const double *beta, int chain_stride, int sweep_stride, int record_stride, int8_t *out_spins,
                      double *out_energy, double *out_min_energy, int32_t *out_argmin, int8_t *out_argmin_state);

/* Optional: build and cache the level schedules of sweeps [sweep0, sweep0+n_sweeps) ahead of time (shared-order
 * philox mode).  Later nlmc_sweep_philox calls inside that range with the same seed and precision skip their
 * schedule pass. */
int nlmc_plan_philox(nlmc_ctx *ctx, int precision, int order_mode, uint32_t sweep0, int n_sweeps, uint64_t seed);
/* Fused-window schedules for n_windows consecutive launches of exactly `window` sweeps each (sweeps sweep0 + w*window
 * ...): all sweeps of a launch share one level list in which the tail of sweep t overlaps the head of sweep t+1
 * (DESIGN.md section 3).  A later nlmc_sweep_philox call with F32, shared order, n_sweeps a multiple of window, a planned
 * sweep0 (a whole number of planned windows) runs on it -- with per-sweep outputs or a temperature per sweep when the
 * snapshot slots of that variant fit in LDS; every other call takes the sweep-by-sweep path.  Results are bit-identical either way.  out_planned: number of windows that got a
 * fused schedule (0 when the instance does not qualify: n < 256 or n > 11264,
 * window < 3 or > 64, or the three threshold tables do not fit in LDS next to the spins). */
int nlmc_plan_philox_fused(nlmc_ctx *ctx, uint32_t sweep0, int n_windows, int window, uint64_t seed, int32_t *out_planned);
/* Which precisions may run on fused windows of `window` sweeps on this context: bit 0 = NLMC_F32, bit 1 = NLMC_F64.  The fp64
 * mode (the reference's arithmetic, NMC/nmc.py:86-87: fp64 field, 53-bit uniform) qualifies when every coupling AND every field
 * is an exact multiple of 2^-qs (+-J, integer and dyadic instances) with max_k (sum|Jq| + |hq|) <= 4095: its field is then the
 * exact integer X times 2^-qs, z = -2 log2(e) beta x takes one value per X, and the test fma(u, 2^z, u) < 1 of the fp64 spec is
 * monotone in the 53-bit integer of u -- one exact integer threshold per (chain, X), built in the kernel's prologue by bisection on
 * the spec's own test.  A plan is shared by both precisions; an fp64 call runs on it while no phase flags are in force and the
 * call has one temperature per chain (otherwise: sweep by sweep).  Bit-identical to the sweep-by-sweep fp64 kernel. */
int nlmc_fused_modes(nlmc_ctx *ctx, int window);
/* Only allocates the plan buffers for up to n_windows windows of `window` sweeps (a later nlmc_plan_philox_fused of at
 * most that size then allocates nothing).  Drops the current fused plan. */
int nlmc_plan_reserve_fused(nlmc_ctx *ctx, int n_windows, int window);

/* Measurement: the floor of ONE level-synchronous step on this device -- workgroup barrier, 8 byte gathers per lane from LDS (all in
 * flight together), sum, one LDS write, barrier: the dependent chain every level of the sweep kernels contains -- in nanoseconds per
 * round, from HIP events around `rounds` rounds of n_workgroups workgroups of `waves` waves (csrc/nlmc_probe.h).  conflict_free = 1:
 * bank-conflict-free gather addresses (the best any placement of row entries could reach), 0: random addresses as in a sweep.
 * bench.py prices the sweep kernel's levels per second against the conflict-free round. */
int nlmc_probe_level_round(nlmc_ctx *ctx, int waves, int conflict_free, int rounds, int n_workgroups, double *out_ns_per_round);

/* Diagnostic: the level list of planned window `window` of the selected plan slot, as chunk offsets (a chunk = 64 schedule
 * positions = one wave's items of one level): level l holds chunks [out[l], out[l+1]); out_n_levels levels (0: the window got
 * no fused schedule).  capacity = entries of the caller's array (>= levels + 1; 1025 always suffices). */
int nlmc_plan_get_levels(nlmc_ctx *ctx, int window, int32_t *out_level_chunk_offsets, int32_t capacity, int32_t *out_n_levels);

/* Is librccl loadable in this process (NLMC_OK), or why not (NLMC_ERR_UNSUPPORTED + nlmc_last_error(NULL))?  Every rank asks
 * BEFORE nlmc_comm_init and the ranks agree on the answer over the process group that is already up, so that no rank waits in
 * ncclCommInitRank for one that could not load the library. */
int nlmc_comm_probe(void);
/* Asynchronous failures of the library-issued collectives: ncclCommGetAsyncError, then (timeout_ms >= 0) a BOUNDED wait for the work
 * queued on the context's stream.  On an error or a timeout (a lost rank leaves the others inside a collective for ever) the
 * communicator is aborted with ncclCommAbort, the context drops it, and the call returns NLMC_ERR_HIP: the caller exits non-zero
 * ("RCCL async error -> abort").  timeout_ms < 0: the error flag only, no wait.  NLMC_OK without a communicator. */
int nlmc_comm_check(nlmc_ctx *ctx, int timeout_ms);

/* ---- APT + iso-cluster moves with the temperature ladder cut into SLOT blocks, one per GPU (NPT/apt_ICM.py:215-285; SURVEY.md
 * section 8e: "keep all sub-replicas of a beta on the same GPU").  Rank w of `world` creates a SELF-CONTAINED context of K ladders
 * (sub-replicas) x Rw local slots (chain_base 0, n_chains_global == n_chains = K Rw, nlmc_pt_init with ITS block
 * beta_global[w Rw .. (w+1) Rw)) and declares it shard w of a global ladder of R_global = world Rw slots:
 *   - random numbers of the sweeps, of the Houdayer pairing and of the cluster pick are keyed by (ladder, GLOBAL slot) instead of by
 *     chain, so exchanging two chains' labels and exchanging their configurations are the same step of the Markov chain;
 *   - nlmc_pt_plan plans the pair selections of the GLOBAL ladder (every rank the same ones);
 *   - nlmc_sweep_philox(beta = NULL), nlmc_icm_round_ladders work on the block as on any context;
 *   - the swap round is nlmc_apt_swap_collective (one process per GPU: the library issues ONE ncclAllGather of K Rw int64 tracked
 *     energies per rank and ONE grouped ncclSend/ncclRecv of the K boundary configurations with each neighbour, on the kernels'
 *     stream) or nlmc_apt_pack + nlmc_apt_swap_host (one process driving several contexts; the same data through host memory):
 *     pairs inside a block are label exchanges, an accepted pair across a block boundary moves the two configurations.
 * The states on every (ladder, global slot) are bit-identical for any `world` (world = 1 included). */
int nlmc_apt_shard(nlmc_ctx *ctx, int R_global, int world, int rank, const double *beta_global /*[R_global]*/);
/* Tracked energies by (ladder, local slot) in units of 2^-escale -> out_slot_efix [K][Rw]; the configurations on local slot 0 /
 * Rw - 1 of every ladder -> out_lo / out_hi [K][n] (any of the three may be NULL). */
int nlmc_apt_pack(nlmc_ctx *ctx, int64_t *out_slot_efix, int8_t *out_lo, int8_t *out_hi);
/* efix_all [world][K][Rw]: every rank's nlmc_apt_pack energies; recv_lo = rank - 1's out_hi, recv_hi = rank + 1's out_lo (NULL at the
 * ends of the ladder).  out_pairs [K][n_pairs][2] GLOBAL slots, out_accepted [K][n_pairs]: the full log, identical on every rank. */
int nlmc_apt_swap_host(nlmc_ctx *ctx, uint32_t round, uint64_t seed, int n_pairs, const int64_t *efix_all, const int8_t *recv_lo,
                       const int8_t *recv_hi, int32_t *out_pairs, uint8_t *out_accepted);
int nlmc_apt_swap_collective(nlmc_ctx *ctx, uint32_t round, uint64_t seed, int n_pairs, int32_t *out_pairs, uint8_t *out_accepted);
/* Rehearsal of the collective path's neighbour exchange with ONE rank: the same grouped ncclSend / ncclRecv pair with this rank as its
 * own lower and upper neighbour; out_recv_lo [K][n] must then equal the packed top-slot configurations (nlmc_apt_pack: out_hi),
 * out_recv_hi the bottom-slot ones.  (N > 1 ranks on one GPU are refused by RCCL: this is what a one-GPU box can execute of it.) */
int nlmc_apt_selftest_exchange(nlmc_ctx *ctx, int8_t *out_recv_lo, int8_t *out_recv_hi);

/* Replica exchange (NPT/npt.py:602-683).  Chains are grouped into ladders of ladder_len consecutive global
 * chain ids; slot r of a ladder runs at beta_list[r].  Accepted swaps exchange the beta slots of two chains
 * (label exchange) -- equivalent to the reference's exchange of the two N-blocks of m_start (NPT/npt.py:677-678). */
int nlmc_pt_init(nlmc_ctx *ctx, int ladder_len, const double *beta_list /*[ladder_len]*/);
int nlmc_pt_get_slots(nlmc_ctx *ctx, int32_t *slot_of_chain /*[n_chains_global]*/);
int nlmc_pt_set_slots(nlmc_ctx *ctx, const int32_t *slot_of_chain /*[n_chains_global]*/);
/* Host-decided swaps (numpy-stream mode: select_non_overlapping_pairs + np.random.rand() stay in Python,
 * NPT/npt.py:514-533,668-671): exchange slots a and b (0-based) of ladder `ladder`. */
int nlmc_pt_apply_swap(nlmc_ctx *ctx, int ladder, int slot_a, int slot_b);
/* Device-decided round: pair selection with the law of NPT/npt.py:514-533 and Metropolis acceptance
 * u < min(1, exp(dBeta*dE)) (NPT/npt.py:668-671), Philox-keyed by (seed, round, ladder).
 *   energies_all_dev: device pointer [n_chains_global] (after the caller's all-gather) or NULL = this context's
 *   own current energies: the single-GPU case, and any context whose block of chains consists of WHOLE ladders -- it then
 *   decides its own ladders only (same keys, same bits as the context that holds every chain), other ladders' rows of the log
 *   read "no pair" (-1, not accepted) and its copy of their slot maps is not advanced.  A block that cuts a ladder needs the
 *   gathered energies.  out_pairs [n_ladders][n_pairs][2] slots, out_accepted [n_ladders][n_pairs]. */
int nlmc_pt_swap_philox(nlmc_ctx *ctx, uint32_t round, uint64_t seed, int n_pairs, const double *energies_all_dev,
                        int32_t *out_pairs, uint8_t *out_accepted);
/* Same round with the all-gathered energies handed over in HOST memory (several contexts driven by one process,
 * NPT.run(device_ids=...): no device-to-device path is assumed between them). */
int nlmc_pt_swap_philox_host(nlmc_ctx *ctx, uint32_t round, uint64_t seed, int n_pairs, const double *energies_all_host,
                             int32_t *out_pairs, uint8_t *out_accepted);
/* Replica-sharded ladders, one process per GPU (NPT/npt.py:616-640 hands one task per replica to a process pool): the ONE
 * collective of a round -- an all-gather of every rank's chain energies -- issued BY THE LIBRARY with RCCL on the stream its
 * kernels run on (librccl is bound at run time; the copy the process already holds is reused).  With a communicator in place
 * the sweep kernels write their chains' energies straight into this rank's block of the gathered vector.
 *   nlmc_comm_unique_id   rank 0 creates the id (128 bytes) and hands it to the others over whatever channel the launcher
 *                         has (torch.distributed broadcast in distributed.ShardedTempering)
 *   nlmc_comm_init        every rank; the context must own block `rank` of `world` equal blocks of chains
 *   nlmc_pt_swap_philox_collective   all-gather + the device-decided swap round of nlmc_pt_swap_philox on the gathered
 *                         energies (identical decision on every rank).  refresh_energies != 0: publish the tracked energies
 *                         first (needed when the states changed other than through a sweep call since the last round). */
int nlmc_comm_unique_id(uint8_t *out_id /*[128]*/);
int nlmc_comm_init(nlmc_ctx *ctx, const uint8_t *id /*[128]*/, int world, int rank);
int nlmc_pt_swap_philox_collective(nlmc_ctx *ctx, uint32_t round, uint64_t seed, int n_pairs, int refresh_energies,
                                   int32_t *out_pairs, uint8_t *out_accepted);
/* Optional: the pair selection depends on the RNG only, so the selections of rounds [round0, round0+n_rounds) can be
 * computed ahead of time (one wave per round and ladder).  Later nlmc_pt_swap_philox calls in that range with the same
 * seed and n_pairs are left with the parallel acceptance test.  Results are identical with or without a plan. */
int nlmc_pt_plan(nlmc_ctx *ctx, uint32_t round0, int n_rounds, uint64_t seed, int n_pairs);
/* n_rounds whole rounds -- sweeps_per_round sweeps of every chain at its ladder temperature, then the swap round of
 * nlmc_pt_swap_philox -- in ONE cooperative launch (k_rounds_fused): the chains stay in LDS from round to round, between two
 * rounds every chain publishes its tracked energy, all workgroups of the launch meet once (bounded wait), and the two chains of a
 * selected pair each evaluate the identical decision.  Bit-identical to nlmc_sweep_philox(beta = NULL) + nlmc_pt_swap_philox round
 * by round (the rounds' decisions go to the device-side swap log when one is open); what a launch per round pays again and again
 * (kernel launches, spins HBM -> LDS -> HBM) is paid once per call.  Needs: a fused-window plan of ONE window per round covering
 * sweeps [sweep0, sweep0 + n_rounds sweeps_per_round) (nlmc_plan_philox_fused, window == sweeps_per_round), the pair selections of
 * rounds [round0, round0 + n_rounds) planned (nlmc_pt_plan, same seed and n_pairs), a context of whole ladders without a
 * communicator, no phase flags / chain subset, one workgroup per chain resident at once (n_chains <= CUs for large n).
 * NLMC_ERR_UNSUPPORTED (nothing was run) when a condition is not met: the caller runs the rounds one by one.  Asynchronous: a grid
 * wait that times out is reported by nlmc_pt_check / nlmc_pt_log_read (NLMC_ERR_HIP). */
int nlmc_pt_rounds_fused(nlmc_ctx *ctx, int precision, int n_rounds, int sweeps_per_round, uint32_t sweep0, uint32_t round0,
                         uint64_t seed, int n_pairs);
/* The same n_rounds rounds as n_rounds sweep launches + ONE swap launch: the sweep launch of round i decides the swap of round i - 1
 * in its prologue (every chain looks up its pair, reads its partner's energy as the previous launch published it, takes k_pt_swap's
 * decision and updates its own entries of the slot maps; wave 0 does it while the other waves load the spins), the last round's
 * swap is the ordinary kernel.  Same conditions and the same NLMC_ERR_UNSUPPORTED convention as nlmc_pt_rounds_fused (n_pairs >= 1);
 * bit-identical to nlmc_sweep_philox + nlmc_pt_swap_philox round by round, one kernel launch per round less. */
int nlmc_pt_rounds_deferred(nlmc_ctx *ctx, int precision, int n_rounds, int sweeps_per_round, uint32_t sweep0, uint32_t round0,
                            uint64_t seed, int n_pairs);
/* Device-side swap log of rounds [round0, round0 + n_rounds): rounds of nlmc_pt_swap_philox(_host) called WITHOUT host
 * output pointers keep their pairs and decisions on the device; nlmc_pt_log_read copies the whole log in one go
 * (out_pairs [n_rounds][n_ladders][n_pairs][2], -1 where a round did not run; out_accepted [n_rounds][n_ladders][n_pairs])
 * and reports pair exhaustion like nlmc_pt_check.  Replaces the reference's per-round prints (NPT/npt.py:662-674). */
int nlmc_pt_log_begin(nlmc_ctx *ctx, uint32_t round0, int n_rounds, int n_pairs);
int nlmc_pt_log_read(nlmc_ctx *ctx, int32_t *out_pairs, uint8_t *out_accepted);
/* NLMC_ERR_ARG ("Cannot find non-overlapping pairs.", the reference's ValueError of NPT/npt.py:526) if the greedy
 * selection of any device-decided round since the last check ran out of pairs; such a round attempts no swap.  Rounds
 * that return their log, and planned rounds (at nlmc_pt_plan time), report it themselves; call this after a run of
 * unplanned rounds without logs.  One stream synchronisation. */
int nlmc_pt_check(nlmc_ctx *ctx);

/* ---- replica-exchange rounds whose marked temperature slots run NMC cycles, device-resident -------------------------
 * NPT.run submits NMC_task (NPT/npt.py:479-512 -> NMC_subroutine :357-477) instead of MCMC_task for the replicas whose
 * doNMC entry is set (NPT/npt.py:622-647); doNMC belongs to the temperature SLOT.  With label-exchange swaps the chains
 * that sit on marked slots change from round to round, so a round is driven on chain SUBSETS:
 *   nlmc_pt_mark_slots(marks)             marks [ladder_len] = doNMC; every ladder must lie inside one context
 *   nlmc_select_chains(UNMARKED | MARKED) later nlmc_sweep_philox / nlmc_adopt_best / nlmc_backbone_clusters /
 *                                         nlmc_set_phase calls act on the local chains currently on such slots (ascending
 *                                         chain id; the list is rebuilt on the device after every swap round); sweep
 *                                         outputs then have one row per chain of the subset (nlmc_subset_count,
 *                                         nlmc_get_subset), per-chain beta tables are refused (ladder or one beta)
 * One NMC_task of all marked chains at once, without a host round trip:
 *   nlmc_backbone_clusters   LBP_convexified (NPT/npt.py:397-403, :129-202) seeded with each chain's current state, then the
 *                            union of find_clusters' clusters (NPT/npt.py:294-355) as a per-chain mask: thresholds[0] =
 *                            threshold_initial selects the seeds, every further entry is one growth step (the host's
 *                            `current_threshold -= threshold_step` values above threshold_cutoff)
 *   nlmc_set_phase(kind, temp_x)   per-spin flags of the subset from its masks: BACKBONE_HOT = cluster rows / temp_x, the
 *                            rest frozen (NPT/npt.py:406-414,425); BACKBONE_FROZEN = cluster spins frozen (:441); ALL =
 *                            plain (:460)
 *   nlmc_track_minimum(1)    sweep calls keep the running minimum + argmin configuration of every chain on the device
 *   nlmc_adopt_best          the argmin configuration becomes the current state, its energy the tracked energy
 *                            (NPT/npt.py:436-437,454-455,469-470: `m_init = M[:, min_energy_idx]`)
 *   nlmc_backbone_check      NLMC_ERR_ARG with the reference's message if an inference since the last check diverged at its
 *                            first lambda (NPT/npt.py:178-180 raises ValueError at once; here it surfaces at the check).
 * nlmc_get_cluster_mask: [n_chains][n] masks of the most recent inference of every chain (tests, diagnostics). */
enum { NLMC_CHAINS_ALL = 0, NLMC_CHAINS_UNMARKED = 1, NLMC_CHAINS_MARKED = 2 };
enum { NLMC_PHASE_ALL = 0, NLMC_PHASE_BACKBONE_HOT = 1, NLMC_PHASE_BACKBONE_FROZEN = 2 };
int nlmc_pt_mark_slots(nlmc_ctx *ctx, const uint8_t *marks /*[ladder_len] or NULL*/);
int nlmc_select_chains(nlmc_ctx *ctx, int which);
/* on != 0: the work of the MARKED subset (inference, phases, hand-offs) is queued on a second HIP stream of the context, forked
 * from the context's stream where a round first selects a subset and joined where it selects all chains again, so that the
 * unmarked chains' sweeps (select UNMARKED first) run beside it -- the reference's pool runs MCMC_task and NMC_task side by
 * side too (NPT/npt.py:616-640).  Same results either way. */
int nlmc_overlap_subsets(nlmc_ctx *ctx, int on);
/* From now on the context queues its work on a non-blocking HIP stream of its own (created here, destroyed with the context)
 * instead of the stream given at creation.  For several contexts on ONE device driven by one process (distributed.LocalTempering
 * with repeated device ids): contexts that were all created on the NULL stream execute one after the other; with a stream each
 * they share the chip -- the reference's knob for this is the pool of num_cores workers, NPT/npt.py:616-640.  Call it right
 * after nlmc_create (pending work of the old stream is waited for), before nlmc_comm_init. */
int nlmc_own_stream(nlmc_ctx *ctx);
int nlmc_subset_count(const nlmc_ctx *ctx);
int nlmc_get_subset(nlmc_ctx *ctx, int32_t *out_chains /*[nlmc_subset_count] local chain ids*/);
/* on = 0: off; 1: the running minimum looks at every sweep of a call; k > 1: at sweeps 0, k, 2k, ... only -- the reference takes its
 * argmin over the recorded columns M[:, ::M_skip] (NMC/nmc.py:390-395 == NPT/npt.py:434-437). */
int nlmc_track_minimum(nlmc_ctx *ctx, int on);
/* mode 1: copy the chains' CURRENT configurations aside; later nlmc_backbone_clusters calls are seeded with that copy instead of the
 * states they find (the m_star of a cycle without a plain phase is the state after the last plain phase, NMC/nmc.py:368-373,433:
 * full_update_frequency != 1).  mode 0: seed with the current states again. */
int nlmc_backbone_seed(nlmc_ctx *ctx, int mode);
int nlmc_adopt_best(nlmc_ctx *ctx);
int nlmc_backbone_clusters(nlmc_ctx *ctx, const double *epsilon /*[n]*/, const double *lambdas, int n_lambdas, double beta,
                           double tolerance, int max_iterations, double sat, const double *thresholds, int n_thresholds);
int nlmc_backbone_check(nlmc_ctx *ctx);
int nlmc_get_cluster_mask(nlmc_ctx *ctx, uint8_t *out /*[n_chains][n]*/);
/* Caller-provided backbones instead of an inference (NMC_subroutine's `all_clusters` argument, NPT/npt.py:357-359): 1 = in a cluster. */
int nlmc_set_cluster_mask(nlmc_ctx *ctx, const uint8_t *mask /*[n_chains][n]*/);
int nlmc_set_phase(nlmc_ctx *ctx, int kind, double temp_x);
/* Fused-window plans live in two slots; nlmc_plan_philox_fused / nlmc_plan_reserve_fused write to the selected one, sweep
 * calls use whichever slot covers their sweeps (a round sweeps its plain chains on windows of num_sweeps_MCMC_per_swap and
 * its NMC phases on windows of num_sweeps_per_NMC_phase_per_swap, NPT/npt.py:577-580). */
int nlmc_plan_slot(nlmc_ctx *ctx, int slot /*0 or 1*/);

/* Houdayer iso-cluster move (NPT/apt_ICM.py:116-143, 215-246) between the current states of local chains a and
 * b: connected components of the disagreement sub-graph, pick component number `pick_index mod n_components`
 * (list ordered by ascending smallest member); if its size > n/2 and katzgraber: state a := -state a, else
 * exchange the component between the two states.  out_info = {n_components, picked size}. */
int nlmc_icm_components(nlmc_ctx *ctx, int chain_a, int chain_b, int32_t *out_n_components);
int nlmc_icm_move(nlmc_ctx *ctx, int chain_a, int chain_b, int64_t pick_index, int katzgraber, int32_t *out_info);
/* Component label (= smallest member index) of every spin for the pair of the most recent nlmc_icm_components /
 * nlmc_icm_move call; -1 where the two states agree.  Backs find_disagreement_clusters (NPT/apt_ICM.py:116-143). */
int nlmc_icm_get_labels(nlmc_ctx *ctx, int32_t *out /*[n]*/);
/* Device-decided batch: pairs [n_pairs][2] local chain ids, pick = philox(pair, round, ., ICM) */
int nlmc_icm_round_philox(nlmc_ctx *ctx, const int32_t *pairs, int n_pairs, uint32_t round, uint64_t seed,
                          int katzgraber, int32_t *out_info /*[n_pairs][2] nullable*/);
/* The Houdayer step of one APT round decided entirely on the device (NPT/apt_ICM.py:216-246 for every temperature):
 * chains are K = n_chains_global / ladder_len ladders (sub-replicas) of ladder_len temperature slots (nlmc_pt_init);
 * per slot the ladders are shuffled (Philox(ladder, round, slot, ICM_PAIR) keys) and paired, each pair gets one
 * iso-cluster move with a Philox-picked cluster, tracked energies are resynchronised.  No host round trip.
 * out_n_pairs = ladder_len * (K / 2); out_info (nullable) [n_pairs][2] = {n_components, picked size}, slot-major. */
int nlmc_icm_round_ladders(nlmc_ctx *ctx, uint32_t round, uint64_t seed, int katzgraber, int32_t *out_n_pairs,
                           int32_t *out_info /*nullable*/);

/* Convexified loopy belief propagation (backbone inference), batched: replaces the lambda loop of LBP_convexified
 * (NMC/nmc.py:126-160) together with LoopyBeliefPropagation (NMC/nmc.py:168-228) for n_problems seeds m_star on the
 * context's (J, h).  Host arrays in, host arrays out; messages stay on the device.
 *   h_lambda = h + lambda * m_star * epsilon; messages start at h_msgs = 0, u_msgs = J * m_star (NMC/nmc.py:128-129) and
 *   are carried from one lambda to the next; each lambda iterates until both relative max-norm changes are below
 *   `tolerance` or max_iterations is reached.  Exhausting the iterations at the first lambda sets status 1 (the
 *   reference raises ValueError there); at a later lambda the previous marginals are kept and the loop stops.
 *   lambdas: the host-computed list (lambda *= factor until < lambda_end or round(lambda, 6) == 0).
 *   sat: clip bound of atanh_saturated (tanh(19.06) - eps, NMC/nmc.py:230-255).
 * out_mag [n_problems][n]: marginals after the last processed lambda; out_mag_all (nullable)
 * [n_problems][n_lambdas][n]: one row per processed lambda; out_n_lambdas [n_problems]; out_iters
 * [n_problems][n_lambdas]: last iteration index per lambda; out_status [n_problems].
 * fp64, row sums in ascending neighbour order: equal to the reference to rounding, not bit for bit.
 * NLMC_ERR_ARG when the sparsity pattern of J is not symmetric. */
int nlmc_lbp_convexified(nlmc_ctx *ctx, int n_problems, const double *m_star, const double *epsilon, const double *lambdas,
                         int n_lambdas, double beta, double tolerance, int max_iterations, double sat,
                         double *out_mag, double *out_mag_all, int32_t *out_n_lambdas, int32_t *out_iters,
                         int32_t *out_status);

/* Cluster growth of find_clusters (NMC/nmc.py:257-318) on the CSR neighbour lists -- host-side graph logic (no context,
 * no device work; irregular and sequential by definition: seeds are served in ascending order and claim exclusively).
 * out_members: clusters concatenated in creation order (capacity members_capacity >= n suffices when J has no
 * diagonal), out_sizes [<= n], out_n_clusters.  Errors are reported through nlmc_last_error(NULL). */
int nlmc_find_clusters(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals, const double *mag,
                       double threshold_initial, double threshold_cutoff, double threshold_step, int32_t *out_members,
                       int64_t members_capacity, int32_t *out_sizes, int32_t *out_n_clusters);

/* Host routine (no device work): lays a recorded trace out as the reference's M blocks.  src [n_blocks][n_sweeps][n]
 * int8 (out_spins of the sweep calls) -> dst [n_dst_blocks][n][row_len]; block b goes to block dst_block[b] (NULL:
 * identity), columns dst_col[b] .. dst_col[b]+n_sweeps-1 (NULL: 0; a multiple of n_sweeps), as int8 (elem_bytes 1) or
 * float64 (elem_bytes 8) -- `M[r*N:(r+1)*N, :] = MCMC(...)` of NPT/npt.py:641 and the sub-replica column groups of
 * NPT/apt_ICM.py:207 for every chain at once, filled by n_threads host threads (<= 0: one per core, at most 32).
 * Elements of dst that no block maps to are left untouched. */
int nlmc_trace_layout(const int8_t *src, int64_t n_blocks, int64_t n_sweeps, int64_t n, const int32_t *dst_block,
                      const int32_t *dst_col, int64_t n_dst_blocks, int64_t row_len, void *dst, int elem_bytes, int n_threads);

/* Host routine: writes one zero byte into every 4 KB page of a freshly allocated buffer from n_threads threads (<= 0: one per
 * core, at most 32) -- the first-touch page faults of the reference-shaped float64 M (205 MB at the C4 shape: 14 of the 47 ms of
 * an NPT.run call) are taken while the GPU is still sweeping instead of after it. */
int nlmc_host_prefault(void *ptr, int64_t bytes, int n_threads);

/* Timing of the most recent sweep call, measured with HIP events on the context's stream -- zero unless event timing
 * was switched on with nlmc_timing_reset (the plain product path records no events). */
int nlmc_last_timing(nlmc_ctx *ctx, float *ms_levelize, float *ms_sweep, int32_t *launches_sweep);
/* Accumulated HIP-event timing of the sweep calls since nlmc_timing_reset (one synchronisation, at read time).
 * enable = 0: off; 1: events around every kernel launch; k > 1: around every k-th fused-window launch only (an event
 * record is a stream command of its own: at one launch per 130 us the two records of a launch are a measurable part of
 * the gap between launches).  While accumulation is on, sweep calls do not recycle their events.
 * nlmc_timing_total: ms_sweep = summed duration of the `launches_timed` launches that had events (of `launches_sweep`
 * launches in all); ms_levelize = summed duration of every schedule construction. */
int nlmc_timing_reset(nlmc_ctx *ctx, int enable);
int nlmc_timing_total(nlmc_ctx *ctx, double *ms_levelize, double *ms_sweep, int64_t *launches_sweep, int64_t *launches_timed);
/* Level-schedule statistics of the most recent sweep call: total levels and total spins over its orders. */
int nlmc_last_schedule_stats(nlmc_ctx *ctx, int64_t *n_orders, int64_t *n_levels);

#ifdef __cplusplus
}
#endif
#endif /* NLMC_H */
