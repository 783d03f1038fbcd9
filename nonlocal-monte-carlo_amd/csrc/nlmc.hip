// nlmc.hip -- context management, kernel launchers and the C-ABI of include/nlmc.h (gfx950 only).
#include "../../include/nlmc.h"
#include "nlmc_kernels.h"
#include "nlmc_pt_icm.h"
#include "nlmc_apt.h"
#include "nlmc_lbp.h"
#include "nlmc_nmc.h"
#include "nlmc_probe.h"
#include "nlmc_big.h"
#include "nlmc_host.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr double LOG2E = 1.4426950408889634;
std::string g_create_error;

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    bool borrowed = false;      // p points into the context's arena (nlmc_create): not freed here
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p && !borrowed) { hipError_t e = hipFree(p); if (e != hipSuccess) { p = nullptr; cap = 0; return e; } }
        p = nullptr; cap = 0; borrowed = false;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void borrow(void *base, size_t &offset, size_t n)     // n elements at the next 256-byte boundary of the arena
    {
        offset = (offset + 255) & ~(size_t)255;
        p = reinterpret_cast<T *>(static_cast<char *>(base) + offset);
        cap = n; borrowed = true;
        offset += std::max<size_t>(n, 1) * sizeof(T);
    }
    static size_t arena_bytes(size_t n) { return ((std::max<size_t>(n, 1) * sizeof(T)) + 255 + 255) & ~(size_t)255; }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
};

}  // namespace

struct nlmc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;             // nlmc_own_stream: `stream` was created by the library
    void *arena = nullptr;                // ONE allocation behind the fixed-size buffers of nlmc_create (13 hipMalloc + 7 blocking
                                          // memsets + 6 blocking copies were 6 of the 29 ms of an NPT.run call at the C4 shape)
    // Work on the MARKED chain subset may run on a second stream beside the unmarked chains' sweeps (nlmc_overlap_subsets):
    // `cur` is the stream the subset-aware launches go to, forked from / joined to `stream` by nlmc_select_chains.
    hipStream_t aux = nullptr, cur = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = false, forked = false;
    int n = 0, n_pad = 0;
    int64_t nnz = 0;
    int n_chains = 0, chain_base = 0, n_chains_global = 0;
    int escale = 32;
    bool compact16 = false;            // every Jq fits 16 bits: fused-window schedules use 4-byte entries
    bool f64_pack16 = false;           // ... and every J equals Jq 2^-qs exactly: the fp64 schedule window holds 16-bit columns / Jq
    bool sign8 = false;                // every Jq is +-1: fused-window schedules may use 2-byte entries (NLMC_FMT_ADDR)
    bool f64_exact = false;            // every J AND every h is an exact multiple of 2^-qs: the fp64 field is the integer field times
                                       // 2^-qs, the fp64 mode may run on fused windows (k_sweep_fused<.., F64>)
    int xmax = 0;                      // max over rows of sum |Jq| + |hq|: the range of the integer field
    unsigned knob_tie_mask = 0xFFFFFFFFu;   // NLMC_F64_TIE_MASK (test knob, read at nlmc_create)
    // diagnostic switches read ONCE, at nlmc_create (NLMC_NO_WARM, NLMC_FUSED_NOPRIO, NLMC_NO_DBUF, NLMC_DBG_FLAGS): not
    // looked up again on the launch path
    bool knob_no_warm = false, knob_no_prio = false, knob_no_dbuf = false;
    int knob_dbg_flags = 0;
    int ev_every = 1;                  // fused-window launches: events around every ev_every-th one (nlmc_timing_reset)
    long long launches_timed = 0;      // launches that contributed to the event sums since nlmc_timing_reset
    int qs = 0;                        // field scale of the fixed-point ("f32") path: Jq = rint(J 2^qs)
    double *energy_sink = nullptr;     // device buffer the sweep kernels also write the tracked energies to
    double temp_x = 1.0;
    bool has_flags = false;
    bool has_diag = false;
    bool has_zero_vals = false;   // a stored entry is 0.0 (or underflows to 0 in fp32)
    size_t lds_opt[96] = {};      // dynamic-LDS opt-in already granted, one slot per kernel: 0 k_levelize, 1 k_icm_components,
                                  // 2..7 sweep-by-sweep kernels, 8 k_stream_scatter, 16 k_levelize_fused, 17..20 k_lbp_lds, 21 k_icm_round, 22..23 packed fp64 sweep kernels, 24..47 k_sweep_fused variants, 48..59 its fp64 variants, 60..71 k_rounds_fused, 72..83 k_sweep_fused with the deferred swap

    DevBuf<int32_t> rowptr, col;
    DevBuf<double> val64, h64;
    DevBuf<EdgeQ> edge32;
    DevBuf<int32_t> hq;
    DevBuf<int8_t> spins, best;
    DevBuf<uint8_t> flags;
    DevBuf<long long> efix, emin, etrace, dbg;
    DevBuf<int32_t> argmin;
    DevBuf<double> energy, tab, ustream, etrace_d;
    DevBuf<uint32_t> keys;
    DevBuf<int32_t> perm_raw;          // stream mode: the caller's permutations / uniforms as drawn, scattered by spin on the device
    DevBuf<double> u_raw;
    DevBuf<int32_t> stream_bad;
    DevBuf<int8_t> strace, cfg, snap_g;
    int strace_nrec = 0;          // recorded configurations per chain held in strace by the last sweep call (0: none)
    int strace_rows = 0;          // chains (rows) of that trace: the subset the call ran on
    // chain subsets (nlmc_pt_mark_slots / nlmc_select_chains): sweeps, hand-offs and backbone inference of a call act on
    // the local chains that currently sit on unmarked / marked temperature slots
    int subset = 0;               // NLMC_CHAINS_ALL / _UNMARKED / _MARKED
    int n_marked_local = 0;       // marked slots x local ladders (static: ladders are whole inside a context)
    bool sub_dirty = true;        // slots moved since the list was built
    DevBuf<uint8_t> slot_mark;    // [ladder_len]
    DevBuf<int32_t> sub_list_buf; // [n_chains]: unmarked chains ascending, then marked chains ascending
    int sub_count() const { return subset == 0 ? n_chains : subset == 1 ? n_chains - n_marked_local : n_marked_local; }
    const int32_t *sub_list() const { return subset == 0 ? nullptr : subset == 1 ? sub_list_buf.p : sub_list_buf.p + (n_chains - n_marked_local); }
    bool track_min = false;       // sweep calls keep running minimum + argmin state on the device (nlmc_track_minimum)
    int track_min_stride = 1;     // ... over the sweeps 0, stride, 2 stride, ... of a call (the reference's M[:, ::M_skip])
    DevBuf<int8_t> seed_snap;     // nlmc_backbone_seed: the configurations later backbone inferences are seeded with
    bool seed_snap_on = false;
    DevBuf<uint8_t> cmask;        // [n_chains][n_pad] backbone mask of the last inference per chain
    DevBuf<uint8_t> cmask_scratch; // chains too long for LDS: the two membership arrays of k_cluster_mask per problem
    DevBuf<int32_t> big_flag;     // ... "changed" words of the levelizer's relaxation passes
    DevBuf<long long> big_esum;   // ... energy deltas of the running sweep per block row
    DevBuf<int32_t> nmc_status;   // sticky: a backbone inference diverged at its first lambda
    DevBuf<double> nmc_thr;       // thresholds of the cluster growth
    std::vector<double> nmc_thr_host, tab_host, lbp_eps_host, lbp_lams_host;   // contents of the device copies (uploads skipped when unchanged)
    double lbp_tJ_beta = 0.0; bool lbp_tJ_valid = false;
    // schedule scratch (per call) and plan cache (persistent)
    struct Sched {            // one set of level-schedule buffers (per-call scratch, or the persistent plan)
        DevBuf<int2> order, head32;
        DevBuf<int32_t> lvl_off, nlev, hi_max, ellc64;
        DevBuf<EdgeQ> ell32;
        DevBuf<double> ellv64, headh64;
        DevBuf<uint32_t> bkey, bcur;      // chains too long for LDS (csrc/nlmc_big.h): the levelizer's keys / cursors
        DevBuf<int32_t> blvl;
        hipError_t reserve_big(size_t orders, size_t n, bool philox_keys)
        {
            hipError_t e;
            if ((e = order.reserve(orders * n)) != hipSuccess) return e;
            if ((e = lvl_off.reserve(orders * (n + 1))) != hipSuccess) return e;
            if ((e = nlev.reserve(orders)) != hipSuccess) return e;
            if ((e = hi_max.reserve(orders)) != hipSuccess) return e;
            if (philox_keys && (e = bkey.reserve(orders * n)) != hipSuccess) return e;
            if ((e = blvl.reserve(orders * n)) != hipSuccess) return e;
            return bcur.reserve(orders * (n + 2));
        }
        hipError_t reserve(size_t orders, size_t n, int mode /*0 none, 1 f32, 2 f64*/)
        {
            hipError_t e;
            if ((e = order.reserve(orders * n)) != hipSuccess) return e;
            if ((e = lvl_off.reserve(orders * (n + 1))) != hipSuccess) return e;
            if ((e = nlev.reserve(orders)) != hipSuccess) return e;
            if ((e = hi_max.reserve(orders)) != hipSuccess) return e;
            if (mode == 1) {
                if ((e = head32.reserve(orders * n)) != hipSuccess) return e;
                if ((e = ell32.reserve(orders * n * NLMC_ELL_W32)) != hipSuccess) return e;
            } else if (mode == 2) {
                if ((e = headh64.reserve(orders * n)) != hipSuccess) return e;
                if ((e = ellc64.reserve(orders * n * NLMC_ELL_W)) != hipSuccess) return e;
                if ((e = ellv64.reserve(orders * n * NLMC_ELL_W)) != hipSuccess) return e;
            }
            return hipSuccess;
        }
        void release() { order.release(); head32.release(); lvl_off.release(); nlev.release(); hi_max.release(); ellc64.release(); ell32.release(); ellv64.release(); headh64.release(); bkey.release(); bcur.release(); blvl.release(); }
    };
    Sched scratch, plan;
    int n_cu = 0;                        // compute units of the device (asked once)
    bool big = false;                    // n > NLMC_LDS_N: every sweep runs the global-memory kernels of csrc/nlmc_big.h
    bool plan_big = false;               // the cached plan was built by k_levelize_big (its items are { k, row start })
    int plan_precision = 0;
    bool plan_valid = false;
    int plan_mode = 0;
    uint32_t plan_sweep0 = 0;
    int plan_count = 0;
    uint64_t plan_seed = 0;
    // fused-window plan (k_levelize_fused / k_sweep_fused)
    int max_deg = 0;
    int n_long = 0;                    // rows with more than NLMC_FZ_W entries (two schedule positions in a fused plan)
    int stat_fused_window = -1;          // >= 0: the most recent sweep call ran this fused window
    // Two plan slots (nlmc_plan_slot): a replica-exchange round whose marked slots run NMC cycles sweeps its plain chains
    // on windows of one length and its NMC phases on windows of another (NPT/npt.py:577-580).
    struct FusedPlan {
        bool valid = false;
        uint32_t sweep0 = 0;
        int windows = 0, T = 0, workers = 13;
        uint64_t seed = 0;
        std::vector<int32_t> nlev_host, npos_host;
        int pstride = 0, gen0 = 0;
        int fmt = 0;                  // entry format of the plan (NLMC_FMT_*)
        DevBuf<int32_t> npos;
        DevBuf<int2> head;
        DevBuf<EdgeQ> ell;
        DevBuf<int32_t> loff, nlev, himax, send;
        void release() { npos.release(); head.release(); ell.release(); loff.release(); nlev.release(); himax.release(); send.release(); }
    };
    FusedPlan fz[2];
    int fz_slot = 0;                  // slot the planning entry points write to
    int stat_fused_slot = 0;
    DevBuf<uint16_t> fz_glv;          // planning scratch, shared by the slots (planning is stream-ordered)
    DevBuf<uint32_t> fz_perm;
    DevBuf<uint16_t> fz_adj;
    DevBuf<long long> fz_stats;
    bool fz_adj_ready = false;
    // PT
    int ladder_len = 0;
    bool pt_tab_valid = false;
    double pt_tab_temp_x = 1.0;
    std::vector<double> beta_list;
    DevBuf<double> pt_tab, pt_beta, pt_energies_all;
    DevBuf<int32_t> slot_of_chain, chain_of_slot, pt_pairs, pt_status, pt_plan_pairs, pt_plan_ok;
    bool pt_plan_valid = false;
    uint32_t pt_plan_round0 = 0;
    int pt_plan_rounds = 0, pt_plan_npairs = 0;
    uint64_t pt_plan_seed = 0;
    DevBuf<uint8_t> pt_acc, pt_log_acc;
    DevBuf<int32_t> pt_log_pairs;
    bool pt_log_on = false;
    uint32_t pt_log_round0 = 0;
    int pt_log_rounds = 0, pt_log_npairs = 0;
    std::vector<uint8_t> stage_in, stage_out;      // padded host staging for row copies
    // ICM
    DevBuf<int32_t> icm_label, icm_info, icm_pairs;
    // loopy BP (edge graph built on first use)
    bool lbp_graph_ready = false;
    DevBuf<int32_t> lbp_src, lbp_rev, lbp_flag, lbp_out_i;
    DevBuf<unsigned int> lbp_bar;
    DevBuf<double> lbp_part;
    DevBuf<double> lbp_tJ, lbp_eps, lbp_ms, lbp_lams, lbp_w0, lbp_w1, lbp_hm, lbp_tot, lbp_mag, lbp_mag_all;
    // timing / stats
    std::vector<hipEvent_t> events;
    size_t ev_used = 0, ev_call_start = 0;
    std::vector<uint8_t> ev_kind;      // per event triple: 0 = {start, after levelize, after sweep}, 1 = {start, -, end} of a sweep
                                       // launch, 2 = {start, -, end} of a planning kernel (the middle event is not recorded)
    bool ev_accumulate = false;
    long long launches_total = 0;
    int launches_sweep = 0;
    int64_t stat_orders = 0, stat_levels = 0;
    bool stats_pending = false;
    const int32_t *stats_nlev_ptr = nullptr;
    int64_t stats_nlev_count = 0;

    // APT run cut into slot blocks over ranks (nlmc_apt_shard, csrc/nlmc_apt.h)
    int apt_R = 0, apt_world = 0, apt_rank = 0;      // global ladder length; 0: not sharded by slot
    int rng_stride = 0, rng_base = 0;                // random numbers keyed by (ladder, global slot): SweepArgs::rng_*
    DevBuf<double> apt_beta;                         // [apt_R] the global ladder
    DevBuf<long long> apt_e_all;                     // [world][K][ladder_len]
    DevBuf<int8_t> apt_send, apt_recv;               // [2][K][n_pad]
    DevBuf<int32_t> apt_bd;                          // [2][K]
    DevBuf<double> rounds_ebuf;                      // k_rounds_fused: [2][n_chains_global] published energies
    DevBuf<unsigned> rounds_bar;                     // its arrival counter
    DevBuf<unsigned char> rounds_args;               // its argument structs (read through constant-memory pointers), 2 slots
    std::vector<unsigned char> rounds_args_host[2];  // ... as uploaded (kept until the next call of the same slot)
    int rounds_args_slot = 0;
    void *comm = nullptr;              // RCCL communicator of the sharded tempering (nlmc_comm_init): the per-round all-gather of
    int comm_world = 0, comm_rank = 0; // the energies is issued by the library on the kernels' own stream

    std::string err;
    CsrDev g{};
};

namespace {

int fail(nlmc_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            return fail((c), NLMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));            \
    } while (0)

// [rows][n] host rows <-> [rows][n_pad] device rows.  Pitched 2-D copies from pageable memory are served row by row
// by the runtime (~80 us per row); one linear copy through a padded staging buffer costs one transfer.
int rows_to_device(nlmc_ctx *c, void *dst_dev, const void *src_host, int rows);
int rows_to_host_begin(nlmc_ctx *c, const void *src_dev, int rows);               // async copy into c->stage
void rows_to_host_finish(nlmc_ctx *c, void *dst_host, int rows);                  // after the stream was synchronised

void tag_triple(nlmc_ctx *c, uint8_t kind)     // call right after taking the three events of a triple
{
    const size_t t = c->ev_used / 3 - 1;
    if (c->ev_kind.size() <= t) c->ev_kind.resize(t + 1, 0);
    c->ev_kind[t] = kind;
}

// elapsed (levelize, sweep) milliseconds of triple t
int triple_ms(nlmc_ctx *c, size_t t, float &lev, float &sw);

hipEvent_t next_event(nlmc_ctx *c)
{
    if (c->ev_used == c->events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->events.push_back(e);
    }
    return c->events[c->ev_used++];
}

int triple_ms(nlmc_ctx *c, size_t t, float &lev, float &sw)
{
    const size_t i = 3 * t;
    const uint8_t kind = t < c->ev_kind.size() ? c->ev_kind[t] : 0;
    lev = sw = 0.f;
    HIP_TRY(c, hipEventSynchronize(c->events[i + 2]));
    if (kind == 0) {
        HIP_TRY(c, hipEventElapsedTime(&lev, c->events[i], c->events[i + 1]));
        HIP_TRY(c, hipEventElapsedTime(&sw, c->events[i + 1], c->events[i + 2]));
    } else if (kind == 1) {
        HIP_TRY(c, hipEventElapsedTime(&sw, c->events[i], c->events[i + 2]));
    } else {
        HIP_TRY(c, hipEventElapsedTime(&lev, c->events[i], c->events[i + 2]));
    }
    return NLMC_OK;
}

int sweep_block(int n)
{
    if (const char *s = getenv("NLMC_SWEEP_NT")) {
        int v = atoi(s);
        if (v >= 64 && v <= 1024 && v % 64 == 0) return v;
    }
    int nt = ((n + 7) / 8 + 63) / 64 * 64;
    return std::min(1024, std::max(64, nt));
}

// fp64-field philox kernels: 8 waves at most (256 registers per lane; with 16 waves the 128-register cap spilled 13-27
// registers per lane to scratch inside the level loop); levels wider than the workgroup are split when the schedule is built
// (fixed-point kernel with self-couplings: 12 waves, 170 registers: at 16 waves it spilled 2)
int sweep_block_for(int n, bool f64_philox, bool f32_diag = false)
{
    return std::min(f64_philox ? 512 : f32_diag ? 768 : 1024, sweep_block(n));
}

// beyond the default dynamic-LDS window a kernel has to opt in (once per size step)
int ensure_lds(nlmc_ctx *c, int slot, const void *func, size_t bytes)
{
    if (bytes > (size_t)158 * 1024) return fail(c, NLMC_ERR_UNSUPPORTED, "instance too large for the LDS-resident kernels of this build");
    if (bytes <= (size_t)60 * 1024 || bytes <= c->lds_opt[slot]) return NLMC_OK;
    HIP_TRY(c, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    c->lds_opt[slot] = bytes;
    return NLMC_OK;
}

int level_block(int n)
{
    int nt = ((n + 3) / 4 + 63) / 64 * 64;
    return std::min(1024, std::max(64, nt));
}

// (chains too long for LDS: read in place, csrc/nlmc_big.h)
void launch_energy(nlmc_ctx *c, unsigned count, hipStream_t st, const EnergyArgs &a)
{
    const dim3 nt(c->n >= 4096 ? 1024 : 256);          // (the partial sums are per thread: one width for both variants)
    if (c->big) hipLaunchKernelGGL(k_energy<true>, dim3(count), nt, 0, st, a);
    else hipLaunchKernelGGL(k_energy<false>, dim3(count), nt, (size_t)c->n_pad, st, a);
}

// energies of the context's own chains -> efix (+ optional double output on device)
int launch_energy_self(nlmc_ctx *c, double *dev_out)
{
    EnergyArgs a{};
    a.g = c->g;
    a.spins = c->spins.p;
    a.stride = c->n_pad;
    a.out = dev_out;
    a.efix = c->efix.p;
    a.escale = c->escale;
    if (c->n_chains > 0) launch_energy(c, (unsigned)c->n_chains, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int run_levelize(nlmc_ctx *c, int n_orders, const uint32_t *keys_in, int per_chain, int n_sweeps, uint32_t sweep0,
                 uint64_t seed, nlmc_ctx::Sched &sc, int ell_mode)
{
    if (n_orders <= 0) return NLMC_OK;
    if (ell_mode < 0) {            // spins in global memory (csrc/nlmc_big.h); sc was reserved with reserve_big
        BigLevelizeArgs b{};
        b.g = c->g;
        b.n_orders = n_orders;
        b.keys_in = keys_in;
        b.seed_lo = (uint32_t)seed; b.seed_hi = (uint32_t)(seed >> 32); b.sweep0 = sweep0;
        b.per_chain = per_chain; b.n_sweeps = n_sweeps; b.chain_base = c->chain_base;
        b.key = sc.bkey.p; b.lvl = sc.blvl.p; b.cur = sc.bcur.p;
        b.ord2 = sc.order.p; b.lvl_off = sc.lvl_off.p; b.nlev = sc.nlev.p; b.hi_max = sc.hi_max.p;
        HIP_TRY(c, c->big_flag.reserve(NLMC_BIG_PASSES));
        b.flag = c->big_flag.p;
        // grid.y carries the order: batches of at most 65535 orders
        const unsigned gx = (unsigned)((c->n + 255) / 256);
        for (int ob = 0; ob < n_orders; ob += 65535) {
            const int no = std::min(65535, n_orders - ob);
            BigLevelizeArgs q = b;
            const size_t n = (size_t)c->n;
            q.n_orders = no;
            if (q.keys_in) q.keys_in += (size_t)ob * n;
            q.key = b.key ? b.key + (size_t)ob * n : nullptr; q.lvl += (size_t)ob * n; q.cur += (size_t)ob * (n + 2);
            q.ord2 += (size_t)ob * n; q.lvl_off += (size_t)ob * (n + 1); q.nlev += ob; q.hi_max += ob;
            // (Philox keys of per-chain orders are numbered over the whole call: the batch's first order id)
            if (!q.keys_in && per_chain && n_orders > 65535) return fail(c, NLMC_ERR_UNSUPPORTED, "more than 65535 per-chain orders of a long chain in one window");
            if (!q.keys_in) q.sweep0 = sweep0 + (uint32_t)ob;
            hipLaunchKernelGGL(k_blv_init, dim3(gx, no), dim3(256), 0, c->cur, q);
            HIP_TRY(c, hipGetLastError());
            // passes until no spin waits for a neighbour's level any more (the depth of the schedule: ~20-40 for sparse graphs, n at worst)
            bool settled = false;
            for (long long done = 0; !settled && done <= (long long)c->n + NLMC_BIG_PASSES; done += NLMC_BIG_PASSES) {
                int32_t fl[NLMC_BIG_PASSES];
                HIP_TRY(c, hipMemsetAsync(q.flag, 0, sizeof(fl), c->cur));
                for (int p = 0; p < NLMC_BIG_PASSES; ++p) hipLaunchKernelGGL(k_blv_pass, dim3(gx, no), dim3(256), 0, c->cur, q, p);
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipMemcpyAsync(fl, q.flag, sizeof(fl), hipMemcpyDeviceToHost, c->cur));
                HIP_TRY(c, hipStreamSynchronize(c->cur));
                for (int p = 0; p < NLMC_BIG_PASSES; ++p) settled = settled || fl[p] == 0;
            }
            // (the order is total: n passes always suffice; never walk on with unset levels)
            if (!settled) return fail(c, NLMC_ERR_STATE, "level schedule of a long chain did not settle");
            const unsigned gt = (unsigned)((c->n + NLMC_BIG_TILE - 1) / NLMC_BIG_TILE);
            hipLaunchKernelGGL(k_blv_hist, dim3(gt, no), dim3(256), 0, c->cur, q);
            hipLaunchKernelGGL(k_blv_scan, dim3(no), dim3(1024), 0, c->cur, q);
            hipLaunchKernelGGL(k_blv_place, dim3(gt, no), dim3(256), 0, c->cur, q);
            HIP_TRY(c, hipGetLastError());
        }
        return NLMC_OK;
    }
    LevelizeArgs a{};
    a.g = c->g;
    a.n_orders = n_orders;
    a.keys_in = keys_in;
    a.seed_lo = (uint32_t)seed;
    a.seed_hi = (uint32_t)(seed >> 32);
    a.sweep0 = sweep0;
    a.per_chain = per_chain;
    a.n_sweeps = n_sweeps;
    a.chain_base = c->chain_base;
    a.level_cap = sweep_block_for(c->n, ell_mode == 2, ell_mode == 1 && c->has_diag);
    a.ord2 = sc.order.p;
    a.lvl_off = sc.lvl_off.p;
    a.nlev = sc.nlev.p;
    a.hi_max = sc.hi_max.p;
    if (ell_mode == 1) { a.ell32 = sc.ell32.p; a.head32 = sc.head32.p; }
    if (ell_mode == 2) { a.ellc64 = sc.ellc64.p; a.ellv64 = sc.ellv64.p; a.headh64 = sc.headh64.p; a.pack16 = c->f64_pack16 ? 1 : 0; }
    const size_t lds_one = (size_t)(c->n + 2) * 4 + (((size_t)c->n * 2 + 3) / 4) * 4;
    const size_t lds_two = lds_one + (size_t)(c->n + 2) * 4;
    a.two_sided = lds_two + 16 <= (size_t)150 * 1024;
    const size_t lds = (a.two_sided ? lds_two : lds_one) + 16;
    { int rc = ensure_lds(c, 0, reinterpret_cast<const void *>(k_levelize), lds); if (rc) return rc; }
    hipLaunchKernelGGL(k_levelize, dim3(n_orders), dim3(level_block(c->n)), lds, c->cur, a);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

// ---- fused-window path ------------------------------------------------------------------------------------
// worker waves of a sweep workgroup of nt threads; the others generate uniforms and warm the cache
// (NLMC_FUSED_WORKERS: tuning knob for the 16-wave case)
// threads of a fused sweep workgroup: at least 4 waves (3 workers + 1 helper) also for small instances
// (3 n / 16 threads: 6-14 % faster than the n / 8 of the sweep-by-sweep kernels between N = 2000 and N = 6000 -- more worker
// waves per level, fewer levels split at the workgroup's width; N = 10^3 is flat from 256 to 768 threads, N >= 5500 has 1024)
int fused_block(int n)
{
    if (getenv("NLMC_SWEEP_NT")) return std::max(256, sweep_block(n));
    const int nt = ((3 * n + 15) / 16 + 63) / 64 * 64;
    return std::min(1024, std::max(256, nt));
}

int fused_workers(int nt)
{
    const int waves = nt / 64;
    if (waves < 16) return waves - 1;
    if (const char *s = getenv("NLMC_FUSED_WORKERS")) { const int v = atoi(s); if (v >= 8 && v <= 16) return v; }
    return 15;      // (one wave left to pull the next window's schedule towards the chip: 16 workers measured 137 vs 124 us
                    // per launch when the plans of 256 windows lie cold in HBM)
}

// LDS of k_sweep_fused: spins (+16: scratch spin of the dummy items, zero bytes) | (address format) negated spins (+16) |
// flags (+16) | 3 threshold tables of n_pad words | (per-sweep outputs only) 3 snapshot slots of n_pad bytes | reduction
// scratch
struct FusedLds { int neg_off, flags_off, u_off, u_bytes, snap_off, red_off, kt_off; size_t total; };
FusedLds fused_lds(int n, int n_pad, bool has_flags, bool with_out, bool with_neg, int kt_entries = 0)
{
    (void)n;
    FusedLds L{};
    L.neg_off = with_neg ? n_pad + 16 : 0;
    L.flags_off = (n_pad + 16) * (with_neg ? 2 : 1);
    int cur = L.flags_off + (has_flags ? n_pad + 16 : 0);
    cur = (cur + 15) / 16 * 16;
    L.u_off = cur;
    L.u_bytes = n_pad * 4;               // threshold word of (slot, k) = slot * n_pad + k = its snapshot byte, too
    cur += 3 * L.u_bytes;
    L.snap_off = cur;
    if (with_out) cur += 3 * n_pad + 16;
    L.red_off = cur;
    L.kt_off = cur + 32;                 // fp64 mode: Khi | Klo, one 4-byte word each per value of the integer field
    L.total = (size_t)cur + 32 + (size_t)kt_entries * 8;
    return L;
}

// the address format needs the negated copy of the spins next to everything else (flags counted in: they may be switched
// on after planning)
bool fused_addr_format(const nlmc_ctx *c)
{
    return c->sign8 && 2 * (c->n_pad + 16) <= 0xFFFF && fused_lds(c->n, c->n_pad, true, false, true).total <= (size_t)150 * 1024;
}

// schedule positions reserved per window: one per update, two for a row longer than NLMC_FZ_W entries (n_long of them),
// plus the padding of every level to a chunk boundary
int fused_pstride(int n, int n_long, int T) { return (int)((((size_t)T * ((size_t)n + n_long) + 63) / 64 + NLMC_LCAP) * 64); }

// first wave that produces thresholds (NLMC_FUSED_GEN0: tuning knob): the trailing half of the workgroup -- the
// waves that rarely or never hold a chunk
int fused_gen0(int nt)
{
    const int waves = nt / 64;
    if (const char *s = getenv("NLMC_FUSED_GEN0")) { const int v = atoi(s); if (v >= 0 && v < waves) return v; }
    return waves / 2;
}

// Instances the fused kernels are built for: a sweep workgroup of at least 4 waves (3 workers + 1 helper), degree in 14 bits, three
// uniform tables next to the spins in LDS (flags counted in: they may be switched on later).
bool fused_supported(const nlmc_ctx *c, int T)
{
    if (getenv("NLMC_NO_FUSED") || c->big) return false;
    if (c->n < 256 || c->n > NLMC_FZ_SPT * 1024 || c->max_deg > 0x3FFF || T < 3 || T > NLMC_FUSED_TMAX) return false;
    if ((size_t)T * ((size_t)c->n + c->n_long) > ((size_t)1 << 22)) return false;   // 32-bit buffer offsets of the packed planes
    if (c->n_pad + 16 > 0x3FFF) return false;                          // spin address in 14 bits of the item head
    (void)T;
    const FusedLds L = fused_lds(c->n, c->n_pad, true, false, false);
    if (3 * L.u_bytes / 4 > 0xFFFF) return false;                      // threshold word index in 16 bits
    return L.total <= (size_t)150 * 1024;
}

// The fp64 mode on fused windows: exact dyadic couplings and fields (the field is an integer), its per-chain threshold tables in
// LDS beside the rest, no phase flags in force (a scaled row's field is a sum of rounded quotients, not an integer).
bool fused_f64_supported(const nlmc_ctx *c, int T)
{
    if (!c->f64_exact || c->xmax > 4095 || getenv("NLMC_NO_FUSED64")) return false;
    if (!fused_supported(c, T)) return false;
    return fused_lds(c->n, c->n_pad, false, false, fused_addr_format(c), 2 * c->xmax + 1).total <= (size_t)156 * 1024;
}

// k_sweep_fused<DIAG, FLAGS, OUT, FMT>: 24 kernels, picked by the instance (self-couplings, entry format of the plan) and the
// call (phase flags on, per-sweep outputs)
const void *fused_kernel(bool diag, bool flags, bool outs, int fmt, bool f64 = false)
{
    if (f64) {           // fp64 mode: plain chains only (12 more kernels)
#define NLMC_K64(D, O) {reinterpret_cast<const void *>(k_sweep_fused<D, false, O, NLMC_FMT_WIDE, true>), reinterpret_cast<const void *>(k_sweep_fused<D, false, O, NLMC_FMT_COMPACT, true>), \
                        reinterpret_cast<const void *>(k_sweep_fused<D, false, O, NLMC_FMT_ADDR, true>)}
        static const void *const t64[2][2][3] = {{NLMC_K64(false, false), NLMC_K64(false, true)}, {NLMC_K64(true, false), NLMC_K64(true, true)}};
#undef NLMC_K64
        return flags ? nullptr : t64[diag][outs][fmt];
    }
#define NLMC_K(D, F, O) {reinterpret_cast<const void *>(k_sweep_fused<D, F, O, NLMC_FMT_WIDE>), reinterpret_cast<const void *>(k_sweep_fused<D, F, O, NLMC_FMT_COMPACT>), \
                         reinterpret_cast<const void *>(k_sweep_fused<D, F, O, NLMC_FMT_ADDR>)}
    static const void *const table[2][2][2][3] = {{{NLMC_K(false, false, false), NLMC_K(false, false, true)}, {NLMC_K(false, true, false), NLMC_K(false, true, true)}},
                                                  {{NLMC_K(true, false, false), NLMC_K(true, false, true)}, {NLMC_K(true, true, false), NLMC_K(true, true, true)}}};
#undef NLMC_K
    static_assert(NLMC_FMT_WIDE == 0 && NLMC_FMT_COMPACT == 1 && NLMC_FMT_ADDR == 2, "table order");
    return table[diag][flags][outs][fmt];
}

struct SweepOut {
    int record_stride;
    int8_t *out_spins;
    double *out_energy;
    double *out_min_energy;
    int32_t *out_argmin;
    int8_t *out_argmin_state;
};

// One fused window.  `outs` (nullable): the launch also produces per-sweep outputs into the context's device buffers
// (etrace / emin / argmin / best / strace, sized by the caller) as sweeps [t0, t0 + T) of a call of n_total sweeps.
int run_fused(nlmc_ctx *c, int slot, int w, uint32_t sweep0, uint64_t seed, const double *tab_dev, int tab_cs, int tab_ss, bool use_slots,
              bool outs, bool want_energy, bool want_min, bool want_state, int rec, int t0, int n_total, bool f64 = false,
              const DeferSwap *defer = nullptr, double *sink_override = nullptr)
{
    const nlmc_ctx::FusedPlan &P = c->fz[slot];
    const int R = c->sub_count(), n = c->n, T = P.T;
    const size_t PS = (size_t)P.pstride;
    // per-sweep outputs: three snapshot slots in LDS when they fit beside the threshold tables, in global memory otherwise
    const int kt = f64 ? 2 * c->xmax + 1 : 0;
    const bool snap_lds = outs && fused_lds(c->n, c->n_pad, c->has_flags, true, P.fmt == NLMC_FMT_ADDR, kt).total <= (size_t)150 * 1024;
    const FusedLds L = fused_lds(c->n, c->n_pad, c->has_flags, snap_lds, P.fmt == NLMC_FMT_ADDR, kt);
    if (outs && !snap_lds) HIP_TRY(c, c->snap_g.reserve((size_t)R * (3 * (size_t)c->n_pad + 16)));
    const int variant = (outs ? 4 : 0) + (c->has_diag ? 2 : 0) + (c->has_flags ? 1 : 0);
    const void *kfun = fused_kernel(c->has_diag, c->has_flags, outs, P.fmt, f64);
    int lds_slot = f64 ? 48 + ((c->has_diag ? 2 : 0) + (outs ? 1 : 0)) * 3 + P.fmt : 24 + variant * 3 + P.fmt;
    if (defer) {             // the previous round's swap decided in this launch's prologue (plain chains, no outputs: checked by the caller)
#define NLMC_KD(D, F64_) {reinterpret_cast<const void *>(k_sweep_fused<D, false, false, NLMC_FMT_WIDE, F64_, true>), reinterpret_cast<const void *>(k_sweep_fused<D, false, false, NLMC_FMT_COMPACT, F64_, true>), \
                          reinterpret_cast<const void *>(k_sweep_fused<D, false, false, NLMC_FMT_ADDR, F64_, true>)}
        static const void *const dtable[2][2][3] = {{NLMC_KD(false, false), NLMC_KD(false, true)}, {NLMC_KD(true, false), NLMC_KD(true, true)}};
#undef NLMC_KD
        kfun = (c->has_flags || outs) ? nullptr : dtable[c->has_diag][f64][P.fmt];
        lds_slot = 72 + ((c->has_diag ? 2 : 0) + (f64 ? 1 : 0)) * 3 + P.fmt;
    }
    if (!kfun) return fail(c, NLMC_ERR_STATE, "run_fused: no kernel for this combination (fp64 mode or deferred swap with phase flags / outputs)");
    { int rc = ensure_lds(c, lds_slot, kfun, L.total); if (rc) return rc; }
    // events around the launch (two stream commands) only while timings accumulate (nlmc_timing_reset): every launch or
    // every ev_every-th one.  An event record costs ~2.5 us of stream time: none on the plain product path.
    const bool timed = c->ev_accumulate && (c->ev_every <= 1 || c->launches_total % c->ev_every == 0);
    hipEvent_t e0 = nullptr, e2 = nullptr;
    if (timed) {
        e0 = next_event(c);
        hipEvent_t e1 = next_event(c);
        e2 = next_event(c);
        if (!e0 || !e1 || !e2) return fail(c, NLMC_ERR_HIP, "hipEventCreate failed");
        tag_triple(c, 1);
        HIP_TRY(c, hipEventRecord(e0, c->cur));
    }
    SweepArgs a{};
    a.g = c->g;
    a.chain_base = c->chain_base;
    a.chain_list = c->sub_list();
    a.spins = c->spins.p;
    a.flags = c->has_flags ? c->flags.p : nullptr;
    a.temp_x = c->temp_x;
    a.lvl_off = P.loff.p + (size_t)w * (NLMC_LCAP + 1);
    a.nlev = P.nlev.p + w;
    a.hi_max = P.himax.p + w;
    a.ell32 = P.ell.p + (size_t)w * PS * NLMC_FZ_W;
    a.head32 = P.head.p + (size_t)w * PS;
    a.fsend = P.send.p + (size_t)w * T;
    a.fz_pstride = P.pstride;
    a.fz_fmt = P.fmt;
    if (w + 1 < P.windows && P.nlev_host[(size_t)w + 1] > 0 && !c->knob_no_warm) {
        a.warm_head = P.head.p + (size_t)(w + 1) * PS;
        a.warm_ell = P.ell.p + (size_t)(w + 1) * PS * NLMC_FZ_W;
        a.fz_npos_next = P.npos_host[(size_t)w + 1];
    }
    a.f_workers = P.workers;
    a.f_gen0 = P.gen0;
    a.f_gen_prio = c->knob_no_prio ? 0 : 1;
#ifdef NLMC_DEBUG_KNOBS
    a.dbg_flags = c->knob_dbg_flags;
#endif
    a.n_sweeps = T;
    a.sweep0 = sweep0;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.tab = tab_dev; a.tab_cs = tab_cs; a.tab_ss = tab_ss;
    a.slot_of_chain = use_slots ? c->slot_of_chain.p : nullptr;
    a.efix = c->efix.p;
    a.energy_sink = sink_override ? sink_override : c->energy_sink;
    if (defer) a.defer = *defer;
    a.escale = c->escale;
    a.eshift = c->escale - c->qs;
    a.qinv = std::ldexp(1.0f, -c->qs);
    a.trace_sweeps = outs ? n_total : T;
    a.t0 = t0;
    a.rec_stride = rec ? rec : 1;
    a.min_stride = c->track_min ? c->track_min_stride : 1;
    a.argmin = c->argmin.p;
    if (outs) {
        a.etrace = want_energy ? c->etrace.p : nullptr;
        a.strace = rec ? c->strace.p : nullptr;
        a.emin = want_min ? c->emin.p : nullptr;
        a.best = (want_min && want_state) ? c->best.p : nullptr;
    }
    a.lds_neg_off = L.neg_off;
    a.lds_flags_off = L.flags_off; a.lds_u_off = L.u_off; a.lds_u_stride = L.u_bytes; a.lds_red_off = L.red_off;
    a.lds_snap_off = L.snap_off;
    a.snap_g = (outs && !snap_lds) ? c->snap_g.p : nullptr;
    a.lds_kt_off = L.kt_off;
    a.f64_xmax = c->xmax;
    a.f64_tie_mask = c->knob_tie_mask;
    a.qinv64 = std::ldexp(1.0, -c->qs);
    a.rng_stride = c->rng_stride; a.rng_base = c->rng_base; a.rng_ladder_len = std::max(1, c->ladder_len);
#ifdef NLMC_STAMPS
    HIP_TRY(c, c->dbg.reserve((size_t)R * 16 * 8 + 96));
    HIP_TRY(c, hipMemsetAsync(c->dbg.p, 0, ((size_t)R * 16 * 8 + 96) * sizeof(long long), c->cur));
    a.dbg = c->dbg.p;
#endif
    const int nt = fused_block(n);
    void *kargs[] = {&a};
    HIP_TRY(c, hipLaunchKernel(kfun, dim3(R), dim3(nt), kargs, L.total, c->cur));
    HIP_TRY(c, hipGetLastError());
    if (timed) { HIP_TRY(c, hipEventRecord(e2, c->cur)); c->launches_timed++; }
    c->launches_sweep++;
    c->launches_total++;
    c->stat_fused_window = w;
    c->stat_fused_slot = slot;
    return NLMC_OK;
}

// device result buffers of a sweep call -> the caller's host arrays.  Rows of the recorded traces are dense in the launch's
// blocks; minima / argmin states live in per-chain rows and are gathered through the subset's chain list.
int read_sweep_outputs(nlmc_ctx *c, const SweepOut &o, int n_sweeps, int rec, int n_rec)
{
    const int R = c->sub_count(), RA = c->n_chains, n = c->n;
    bool need_sync = false;
    std::vector<long long> h_ll;
    if (o.out_energy) {
        h_ll.resize((size_t)R * n_sweeps);
        HIP_TRY(c, hipMemcpyAsync(h_ll.data(), c->etrace.p, sizeof(long long) * h_ll.size(), hipMemcpyDeviceToHost, c->cur));
        need_sync = true;
    }
    std::vector<long long> h_min;
    std::vector<int32_t> h_arg, h_list;
    const bool gather = c->subset != 0 && (o.out_min_energy || o.out_argmin || o.out_argmin_state);
    if (gather) {
        h_list.resize((size_t)R);
        HIP_TRY(c, hipMemcpyAsync(h_list.data(), c->sub_list(), sizeof(int32_t) * R, hipMemcpyDeviceToHost, c->cur));
    }
    if (o.out_min_energy) {
        h_min.resize((size_t)RA);
        HIP_TRY(c, hipMemcpyAsync(h_min.data(), c->emin.p, sizeof(long long) * RA, hipMemcpyDeviceToHost, c->cur));
        need_sync = true;
    }
    if (o.out_argmin) {
        h_arg.resize((size_t)RA);
        HIP_TRY(c, hipMemcpyAsync(h_arg.data(), c->argmin.p, sizeof(int32_t) * RA, hipMemcpyDeviceToHost, c->cur));
        need_sync = true;
    }
    if (o.out_argmin_state) {
        int rc = rows_to_host_begin(c, c->best.p, RA);
        if (rc) return rc;
        need_sync = true;
    }
    if (rec) {
        HIP_TRY(c, hipMemcpyAsync(o.out_spins, c->strace.p, (size_t)R * n_rec * n, hipMemcpyDeviceToHost, c->cur));
        need_sync = true;
    }
    if (need_sync) HIP_TRY(c, hipStreamSynchronize(c->cur));
    auto row = [&](int i) { return gather ? (int)h_list[(size_t)i] : i; };
    if (o.out_argmin_state)
        for (int i = 0; i < R; ++i)
            std::memcpy(o.out_argmin_state + (size_t)i * n, c->stage_out.data() + (size_t)row(i) * c->n_pad, (size_t)n);
    const double inv = std::ldexp(1.0, -c->escale);
    if (o.out_energy) for (size_t i = 0; i < h_ll.size(); ++i) o.out_energy[i] = (double)h_ll[i] * inv;
    if (o.out_min_energy) for (int i = 0; i < R; ++i) o.out_min_energy[i] = (double)h_min[(size_t)row(i)] * inv;
    if (o.out_argmin) for (int i = 0; i < R; ++i) o.out_argmin[i] = h_arg[(size_t)row(i)];
    return NLMC_OK;
}

// chain list of the selected subset, rebuilt (one small launch) after anything that moved chains between slots
int ensure_subset(nlmc_ctx *c)
{
    if (c->subset == 0 || !c->sub_dirty) return NLMC_OK;
    hipLaunchKernelGGL(k_subset_build, dim3(1), dim3(1024), 0, c->stream, c->n_chains, c->chain_base, c->slot_of_chain.p, c->slot_mark.p,
                       c->n_chains - c->n_marked_local, c->sub_list_buf.p);
    HIP_TRY(c, hipGetLastError());
    c->sub_dirty = false;
    return NLMC_OK;
}

// the fused plan slot whose windows cover sweeps [sweep0, sweep0 + n_sweeps) exactly, or -1
int fused_plan_for(const nlmc_ctx *c, uint32_t sweep0, int n_sweeps, uint64_t seed)
{
    for (int k = 0; k < 2; ++k) {
        const nlmc_ctx::FusedPlan &P = c->fz[k];
        if (!P.valid || P.seed != seed || sweep0 < P.sweep0 || (sweep0 - P.sweep0) % (uint32_t)P.T != 0 || n_sweeps % P.T != 0 ||
            (uint64_t)(sweep0 - P.sweep0) + (uint64_t)n_sweeps > (uint64_t)P.T * (uint64_t)P.windows)
            continue;
        const int w0 = (int)((sweep0 - P.sweep0) / (uint32_t)P.T), nw = n_sweeps / P.T;
        bool all_ok = true;
        for (int w = w0; w < w0 + nw; ++w) all_ok = all_ok && P.nlev_host[(size_t)w] > 0;
        if (all_ok) return k;
    }
    return -1;
}

// Chains whose state does not fit in LDS next to what the mode keeps there run the kernels of csrc/nlmc_big.h.  The LDS need is
// taken at its worst (phase flags on, one copy of the uniforms), so that the answer does not change during a context's life: a
// cached plan is in the format of the levelizer that goes with the answer.
bool sweeps_big(const nlmc_ctx *c, bool stream_mode, bool f64)
{
    if (c->big) return true;
    const size_t n4 = ((size_t)c->n + 3) / 4 * 4, u_bytes = stream_mode ? 0 : n4 * (f64 ? 8 : 4);
    return 2 * (size_t)c->n_pad + 32 + u_bytes + (size_t)NLMC_LCAP * 4 + 32 > (size_t)158 * 1024;
}

// Shared driver: windows of sweeps -> (levelize) -> k_sweep.  `stream_mode` selects the kernel flavour.
int run_sweeps(nlmc_ctx *c, bool stream_mode, int precision, int order_mode, int n_sweeps, uint32_t sweep0,
               uint64_t seed, const double *tab_dev, int tab_cs, int tab_ss, bool use_slots, const uint32_t *keys_dev,
               const double *ustream_dev, const SweepOut &o)
{
    const int R = c->sub_count(), n = c->n;
    if (!c->ev_accumulate) c->ev_used = 0;
    c->ev_call_start = c->ev_used;
    c->launches_sweep = 0;
    c->stat_orders = 0;
    c->stat_levels = 0;
    c->stats_pending = false;
    c->strace_nrec = 0;
    c->strace_rows = 0;
    if (R == 0 || n_sweeps == 0) return NLMC_OK;
    { int rc = ensure_subset(c); if (rc) return rc; }
    // running minimum + argmin state: asked for through host outputs, or kept on the device for the hand-off between NMC
    // phases (nlmc_track_minimum; NMC/nmc.py:394-395)
    const bool want_min = o.out_min_energy || o.out_argmin || o.out_argmin_state || c->track_min;
    const bool want_state = o.out_argmin_state || c->track_min;
    c->stat_fused_window = -1;
    const int rec = o.out_spins ? std::max(1, o.record_stride) : 0;
    const int n_rec = rec ? (n_sweeps + rec - 1) / rec : 0;
    const bool any_out = o.out_spins || o.out_energy || want_min;
    // Fused-window schedule: planned ahead (nlmc_plan_philox_fused), same results.  Calls without per-sweep outputs
    // and with one temperature per chain take the plain variant; calls WITH outputs (energy trace, running minimum /
    // argmin state, recorded configurations) or a temperature per sweep take the output variant when its three snapshot
    // slots fit in LDS next to the rest; any whole number of planned windows per call either way.
    int fslot = -1;
    bool fused_out = false;
    // (fp64 mode: on the same windows when the field is an exact integer, no phase flags are in force and the call has one
    // temperature per chain -- fused_f64_supported; the same bits as the sweep-by-sweep fp64 kernel)
    const bool f64_fused = precision == NLMC_F64 && !c->has_flags && tab_ss == 0;
    if (!stream_mode && (precision == NLMC_F32 || f64_fused) && order_mode == NLMC_ORDER_SHARED && !getenv("NLMC_NO_FUSED"))
        fslot = fused_plan_for(c, sweep0, n_sweeps, seed);
    if (fslot >= 0 && f64_fused && !fused_f64_supported(c, c->fz[fslot].T)) fslot = -1;
    if (fslot >= 0) {
        const nlmc_ctx::FusedPlan &P = c->fz[fslot];
        const int w0 = (int)((sweep0 - P.sweep0) / (uint32_t)P.T), nw = n_sweeps / P.T;
        if (!any_out && tab_ss == 0) {
            for (int j = 0; j < nw; ++j) {
                int rc = run_fused(c, fslot, w0 + j, sweep0 + (uint32_t)(j * P.T), seed, tab_dev, tab_cs, 0, use_slots, false, false,
                                   false, false, 0, 0, n_sweeps, f64_fused);
                if (rc) return rc;
            }
            return NLMC_OK;
        }
        if (!getenv("NLMC_NO_FUSED_OUT")) fused_out = true;     // (snapshots in LDS or, for large n, in a global ring: run_fused)
    }
    if (o.out_energy) HIP_TRY(c, c->etrace.reserve((size_t)R * n_sweeps));
    if (rec) HIP_TRY(c, c->strace.reserve((size_t)R * n_rec * n));
    c->strace_nrec = n_rec;
    c->strace_rows = rec ? R : 0;
    if (want_min) {
        hipLaunchKernelGGL(k_fill_min, dim3((R + 255) / 256), dim3(256), 0, c->cur, R, c->sub_list(), c->emin.p, c->argmin.p);
        HIP_TRY(c, hipGetLastError());
    }
    if (fused_out) {
        const nlmc_ctx::FusedPlan &P = c->fz[fslot];
        const int w0 = (int)((sweep0 - P.sweep0) / (uint32_t)P.T), nw = n_sweeps / P.T;
        for (int j = 0; j < nw; ++j) {
            int rc = run_fused(c, fslot, w0 + j, sweep0 + (uint32_t)(j * P.T), seed, tab_dev + (size_t)j * P.T * tab_ss, tab_cs,
                               tab_ss, use_slots, true, o.out_energy != nullptr, want_min, want_state, rec, j * P.T, n_sweeps, f64_fused);
            if (rc) return rc;
        }
        return read_sweep_outputs(c, o, n_sweeps, rec, n_rec);
    }

    // (sweep-by-sweep path from here on: its schedule buffers are shared with whatever the main stream is sweeping)
    if (c->cur != c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int per_chain = (stream_mode || order_mode == NLMC_ORDER_PER_CHAIN) ? 1 : 0;
    if (per_chain && c->subset != 0) return fail(c, NLMC_ERR_UNSUPPORTED, "chain subsets run shared-order philox sweeps only");
    // plan cache hit?
    const bool f64 = stream_mode || precision == NLMC_F64;
    const bool big = sweeps_big(c, stream_mode, f64);      // spins stay in global memory (csrc/nlmc_big.h)
    const int ell_mode = big ? -1 : stream_mode ? 0 : (f64 ? 2 : 1);
    const bool cached = !stream_mode && c->plan_valid && c->plan_mode == order_mode && c->plan_seed == seed && c->plan_big == big &&
                        c->plan_precision == precision && !per_chain && sweep0 >= c->plan_sweep0 &&
                        (uint64_t)sweep0 + (uint64_t)n_sweeps <= (uint64_t)c->plan_sweep0 + (uint64_t)c->plan_count;
    // window size: keep the schedule scratch under ~256 MiB
    const size_t per_sweep_orders = per_chain ? (size_t)R : 1;
    const size_t item_bytes = big ? 8 + 12 : ell_mode == 1 ? 8 + 8 + 8 * NLMC_ELL_W32 : (ell_mode == 2 ? 8 + 8 + 12 * NLMC_ELL_W : 8);
    const size_t bytes_per_sweep = per_sweep_orders * ((size_t)n * item_bytes + (size_t)(n + 1) * 4 + 4);
    int W = n_sweeps;
    if (!cached) {
        // stream mode indexes its uniforms/keys by (chain, sweep) over the WHOLE call -> single window there
        if (!stream_mode) W = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_sweeps, ((size_t)256 << 20) / bytes_per_sweep));
        const size_t orders = per_sweep_orders * (size_t)W;
        if (big) HIP_TRY(c, c->scratch.reserve_big(orders, (size_t)n, !stream_mode));
        else HIP_TRY(c, c->scratch.reserve(orders, (size_t)n, ell_mode));
    }

    const int nt = big ? 1024 : sweep_block_for(n, !stream_mode && f64, !stream_mode && !f64 && c->has_diag);
    // LDS carve-up: spins | flags | uniforms of one sweep (philox) | level offsets (philox) | reduction scratch
    const int lds_flags_off = c->n_pad;
    int cur = c->n_pad * (c->has_flags ? 2 : 1);
    cur = (cur + 15) / 16 * 16;
    const int lds_u_off = cur;
    const int u_bytes = stream_mode ? 0 : ((f64 ? ((n + 3) / 4 * 4) * 8 : ((n + 3) / 4 * 4) * 4) + 15) / 16 * 16;
    // second copy of the per-sweep uniforms / level offsets when it fits: lets the idle waves prepare sweep t+1 while
    // wave 0 runs the narrow tail of sweep t
    const bool dbuf = !stream_mode && (size_t)cur + 2 * (size_t)u_bytes + 2 * NLMC_LCAP * 4 + 32 <= (size_t)150 * 1024 &&
                      !c->knob_no_dbuf;
    cur += u_bytes * (dbuf ? 2 : 1);
    const int lds_loff_off = cur;
    cur += NLMC_LCAP * 4 * (dbuf ? 2 : 1);               // (stream mode: one copy, the level offsets of the running sweep)
    const int lds_red_off = cur;
    const size_t lds = (size_t)cur + 16;
    const void *kfun = stream_mode ? reinterpret_cast<const void *>(k_sweep_stream)
                       : f64 ? (c->f64_pack16 ? (c->has_diag ? reinterpret_cast<const void *>(k_sweep_philox<double, true, true>)
                                                             : reinterpret_cast<const void *>(k_sweep_philox<double, false, true>))
                                              : (c->has_diag ? reinterpret_cast<const void *>(k_sweep_philox<double, true>)
                                                             : reinterpret_cast<const void *>(k_sweep_philox<double, false>)))
                             : (c->has_diag ? reinterpret_cast<const void *>(k_sweep_philox<float, true>)
                                            : reinterpret_cast<const void *>(k_sweep_philox<float, false>));
    const int kslot = stream_mode ? 2 : (f64 ? (c->f64_pack16 ? 22 : 3) : 5) + (c->has_diag ? 1 : 0);
    if (!big) { int rc = ensure_lds(c, kslot, kfun, lds); if (rc) return rc; }

    for (int t0 = 0; t0 < n_sweeps; t0 += W) {
        const int w = std::min(W, n_sweeps - t0);
        const nlmc_ctx::Sched &sc = cached ? c->plan : c->scratch;
        size_t o0 = 0;
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        if (c->ev_accumulate) {
            e0 = next_event(c); e1 = next_event(c); e2 = next_event(c);
            if (!e0 || !e1 || !e2) return fail(c, NLMC_ERR_HIP, "hipEventCreate failed");
            tag_triple(c, 0);
            HIP_TRY(c, hipEventRecord(e0, c->cur));
        }
        if (cached) {
            o0 = (size_t)(sweep0 - c->plan_sweep0) + t0;
        } else {
            const int n_orders = (int)per_sweep_orders * w;
            int rc = run_levelize(c, n_orders, keys_dev, per_chain, w, sweep0 + (uint32_t)t0, seed, c->scratch, ell_mode);
            if (rc) return rc;
            c->stats_nlev_ptr = c->scratch.nlev.p;
            c->stats_nlev_count = n_orders;
            c->stats_pending = true;
        }
        if (e1) HIP_TRY(c, hipEventRecord(e1, c->cur));

        SweepArgs a{};
        a.g = c->g;
        a.chain_base = c->chain_base;
        a.chain_list = c->sub_list();
        a.spins = c->spins.p;
        a.flags = c->has_flags ? c->flags.p : nullptr;
        a.temp_x = c->temp_x;
        a.ord2 = sc.order.p + o0 * n;
        a.lvl_off = sc.lvl_off.p + o0 * (size_t)(n + 1);
        a.nlev = sc.nlev.p + o0;
        a.hi_max = sc.hi_max.p + o0;
        if (ell_mode == 1) { a.ell32 = sc.ell32.p + o0 * (size_t)n * NLMC_ELL_W32; a.head32 = sc.head32.p + o0 * n; }
        if (ell_mode == 2) {
            a.ellc64 = sc.ellc64.p + o0 * (size_t)n * NLMC_ELL_W;
            a.ellv64 = sc.ellv64.p + o0 * (size_t)n * NLMC_ELL_W;
            a.headh64 = sc.headh64.p + o0 * n;
        }
        a.per_chain = per_chain;
        a.n_sweeps = w;
        a.sweep0 = sweep0 + (uint32_t)t0;
        a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
        a.tab = tab_dev + (size_t)t0 * tab_ss;
        a.tab_cs = tab_cs; a.tab_ss = tab_ss;
        a.slot_of_chain = use_slots ? c->slot_of_chain.p : nullptr;
        a.ustream = ustream_dev;
        a.efix = c->efix.p;
        a.energy_sink = c->energy_sink;
        a.escale = c->escale;
        a.eshift = c->escale - c->qs;
        a.qinv = std::ldexp(1.0f, -c->qs);
        a.qinv64 = std::ldexp(1.0, -c->qs);
        a.rng_stride = c->rng_stride; a.rng_base = c->rng_base; a.rng_ladder_len = std::max(1, c->ladder_len);
        a.etrace = o.out_energy ? c->etrace.p : nullptr;
        a.trace_sweeps = n_sweeps;
        a.t0 = t0;
        a.rec_stride = rec ? rec : 1;
        a.min_stride = c->track_min ? c->track_min_stride : 1;
        a.strace = rec ? c->strace.p : nullptr;
        a.emin = want_min ? c->emin.p : nullptr;
        a.argmin = c->argmin.p;
        a.best = (want_min && want_state) ? c->best.p : nullptr;
#ifdef NLMC_STAMPS
        HIP_TRY(c, c->dbg.reserve((size_t)R * 16 * 8 + 96));
        HIP_TRY(c, hipMemsetAsync(c->dbg.p, 0, ((size_t)R * 16 * 8 + 96) * sizeof(long long), c->cur));
        a.dbg = c->dbg.p;
#endif
        a.lds_u_stride = dbuf ? u_bytes : 0;
        a.lds_loff_stride = dbuf ? NLMC_LCAP * 4 : 0;
        a.lds_flags_off = lds_flags_off; a.lds_u_off = lds_u_off; a.lds_loff_off = lds_loff_off; a.lds_red_off = lds_red_off;
        // chains too long for LDS (csrc/nlmc_big.h): enough of them to fill the chip -> one workgroup per chain for the whole
        // window (its spins stay in that CU's L1 from level to level); few -> one launch per level over all chains, so that a
        // single chain of a million spins still uses every CU (NLMC_BIG_PER_LEVEL = 0 / 1: test knob)
        bool per_level = false;
        if (big) {
            if (c->n_cu == 0) HIP_TRY(c, hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
            per_level = (long long)R * 8 <= c->n_cu;
            if (const char *e = getenv("NLMC_BIG_PER_LEVEL")) per_level = atoi(e) != 0;
            if (R > 65535) per_level = false;
        }
        if (big && !per_level) {
            if (stream_mode) hipLaunchKernelGGL(k_sweep_big<NLMC_BIG_STREAM>, dim3(R), dim3(1024), 0, c->cur, a);
            else if (f64) hipLaunchKernelGGL(k_sweep_big<NLMC_BIG_F64>, dim3(R), dim3(1024), 0, c->cur, a);
            else hipLaunchKernelGGL(k_sweep_big<NLMC_BIG_F32>, dim3(R), dim3(1024), 0, c->cur, a);
        } else if (big) {
            // the numbers of levels come back first
            const size_t n_ord = per_sweep_orders * (size_t)w;
            std::vector<int32_t> nlev_h(n_ord);
            HIP_TRY(c, hipMemcpyAsync(nlev_h.data(), a.nlev, sizeof(int32_t) * n_ord, hipMemcpyDeviceToHost, c->cur));
            HIP_TRY(c, hipStreamSynchronize(c->cur));
            HIP_TRY(c, c->big_esum.reserve((size_t)R));
            HIP_TRY(c, hipMemsetAsync(c->big_esum.p, 0, sizeof(long long) * (size_t)R, c->cur));
            const bool per_sweep_out = a.etrace || a.emin || a.strace;
            const unsigned gx = (unsigned)std::max(1, std::min(256, (n / 16 + 255) / 256));
            for (int t = 0; t < w; ++t) {
                int nl = 0;
                if (per_chain) for (size_t q = 0; q < per_sweep_orders; ++q) nl = std::max(nl, nlev_h[q * (size_t)w + t]);
                else nl = nlev_h[(size_t)t];
                for (int l = 0; l < nl; ++l) {
                    if (stream_mode) hipLaunchKernelGGL(k_big_level<NLMC_BIG_STREAM>, dim3(gx, R), dim3(256), 0, c->cur, a, t, l, c->big_esum.p);
                    else if (f64) hipLaunchKernelGGL(k_big_level<NLMC_BIG_F64>, dim3(gx, R), dim3(256), 0, c->cur, a, t, l, c->big_esum.p);
                    else hipLaunchKernelGGL(k_big_level<NLMC_BIG_F32>, dim3(gx, R), dim3(256), 0, c->cur, a, t, l, c->big_esum.p);
                }
                if (per_sweep_out || t == w - 1) hipLaunchKernelGGL(k_big_sweep_end, dim3(R), dim3(1024), 0, c->cur, a, t, c->big_esum.p);
                HIP_TRY(c, hipGetLastError());
            }
        } else if (stream_mode)
            hipLaunchKernelGGL(k_sweep_stream, dim3(R), dim3(nt), lds, c->cur, a);
        else if (f64 && c->f64_pack16 && c->has_diag)
            hipLaunchKernelGGL((k_sweep_philox<double, true, true>), dim3(R), dim3(nt), lds, c->cur, a);
        else if (f64 && c->f64_pack16)
            hipLaunchKernelGGL((k_sweep_philox<double, false, true>), dim3(R), dim3(nt), lds, c->cur, a);
        else if (f64 && c->has_diag)
            hipLaunchKernelGGL((k_sweep_philox<double, true>), dim3(R), dim3(nt), lds, c->cur, a);
        else if (f64)
            hipLaunchKernelGGL((k_sweep_philox<double, false>), dim3(R), dim3(nt), lds, c->cur, a);
        else if (c->has_diag)
            hipLaunchKernelGGL((k_sweep_philox<float, true>), dim3(R), dim3(nt), lds, c->cur, a);
        else
            hipLaunchKernelGGL((k_sweep_philox<float, false>), dim3(R), dim3(nt), lds, c->cur, a);
        HIP_TRY(c, hipGetLastError());
        if (e2) { HIP_TRY(c, hipEventRecord(e2, c->cur)); c->launches_timed++; }
        c->launches_sweep++;
        c->launches_total++;
    }

    return read_sweep_outputs(c, o, n_sweeps, rec, n_rec);
}

}  // namespace

namespace {

// ---- RCCL, bound at run time (dlopen): a process that never shards a ladder over GPUs needs no librccl -----------------
// The library the process already holds (PyTorch ships one under the same soname) is reused.
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;                      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommGetAsyncError)(void *, int *) = nullptr;
    int (*CommAbort)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool tried = false, ok = false;
    std::string why;
};
Rccl g_rccl;

bool rccl_load()
{
    Rccl &r = g_rccl;
    if (r.tried) return r.ok;
    r.tried = true;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { r.why = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?"); return false; }
    r.GetUniqueId = reinterpret_cast<int (*)(Rccl::UniqueId *)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(void **, int, Rccl::UniqueId, int)>(dlsym(h, "ncclCommInitRank"));
    r.AllGather = reinterpret_cast<int (*)(const void *, void *, size_t, int, void *, hipStream_t)>(dlsym(h, "ncclAllGather"));
    r.CommDestroy = reinterpret_cast<int (*)(void *)>(dlsym(h, "ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<const char *(*)(int)>(dlsym(h, "ncclGetErrorString"));
    r.Send = reinterpret_cast<int (*)(const void *, size_t, int, int, void *, hipStream_t)>(dlsym(h, "ncclSend"));
    r.Recv = reinterpret_cast<int (*)(void *, size_t, int, int, void *, hipStream_t)>(dlsym(h, "ncclRecv"));
    r.GroupStart = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
    r.CommGetAsyncError = reinterpret_cast<int (*)(void *, int *)>(dlsym(h, "ncclCommGetAsyncError"));
    r.CommAbort = reinterpret_cast<int (*)(void *)>(dlsym(h, "ncclCommAbort"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd &&
           r.CommGetAsyncError && r.CommAbort;
    if (!r.ok) r.why = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclSend / ncclRecv / ncclGroupStart / "
                       "ncclGroupEnd / ncclCommGetAsyncError / ncclCommAbort / ncclCommDestroy";
    return r.ok;
}

std::string rccl_err(int rc) { return g_rccl.GetErrorString ? std::string(g_rccl.GetErrorString(rc)) : ("ncclResult " + std::to_string(rc)); }

int rows_to_device(nlmc_ctx *c, void *dst_dev, const void *src_host, int rows)
{
    const size_t n = (size_t)c->n, np = (size_t)c->n_pad;
    const void *src = src_host;
    if (n != np) {
        c->stage_in.assign(np * rows, 0);
        for (int r = 0; r < rows; ++r) std::memcpy(c->stage_in.data() + r * np, (const uint8_t *)src_host + r * n, n);
        src = c->stage_in.data();
    }
    HIP_TRY(c, hipMemcpyAsync(dst_dev, src, np * rows, hipMemcpyHostToDevice, c->stream));
    if (n != np) HIP_TRY(c, hipStreamSynchronize(c->stream));      // the staging buffer is reused by the next call
    return NLMC_OK;
}

int rows_to_host_begin(nlmc_ctx *c, const void *src_dev, int rows)
{
    c->stage_out.resize((size_t)c->n_pad * rows);
    HIP_TRY(c, hipMemcpyAsync(c->stage_out.data(), src_dev, c->stage_out.size(), hipMemcpyDeviceToHost, c->cur));
    return NLMC_OK;
}

void rows_to_host_finish(nlmc_ctx *c, void *dst_host, int rows)
{
    const size_t n = (size_t)c->n, np = (size_t)c->n_pad;
    for (int r = 0; r < rows; ++r) std::memcpy((uint8_t *)dst_host + r * n, c->stage_out.data() + r * np, n);
}

}  // namespace

extern "C" {

static int ensure_adjacency(nlmc_ctx *c);
int nlmc_pt_check(nlmc_ctx *c);

int nlmc_abi_version(void) { return NLMC_ABI_VERSION; }

int nlmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *nlmc_last_error(const nlmc_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int nlmc_create(nlmc_ctx **out, int device, void *hip_stream, int n, int64_t nnz, const int32_t *rowptr,
                const int32_t *colidx, const double *vals, const double *h, int n_chains, int chain_base,
                int n_chains_global)
{
    if (!out) return fail(nullptr, NLMC_ERR_ARG, "nlmc_create: out is NULL");
    *out = nullptr;
    if (n < 1 || nnz < 0 || !rowptr || (nnz > 0 && (!colidx || !vals)) || !h || n_chains < 0 || chain_base < 0 ||
        n_chains_global < chain_base + n_chains)
        return fail(nullptr, NLMC_ERR_ARG, "nlmc_create: bad sizes or NULL arrays");
    if (n > NLMC_MAX_N) return fail(nullptr, NLMC_ERR_UNSUPPORTED, "nlmc_create: n exceeds NLMC_MAX_N");
    if (rowptr[0] != 0 || rowptr[n] != nnz) return fail(nullptr, NLMC_ERR_ARG, "nlmc_create: rowptr[0] != 0 or rowptr[n] != nnz");
    bool diag = false, zero_vals = false;
    int max_deg = 0, n_long = 0;
    for (int k = 0; k < n; ++k) {
        if (rowptr[k + 1] < rowptr[k]) return fail(nullptr, NLMC_ERR_ARG, "nlmc_create: rowptr not monotone");
        max_deg = std::max(max_deg, (int)(rowptr[k + 1] - rowptr[k]));
        n_long += (rowptr[k + 1] - rowptr[k]) > NLMC_FZ_W;
        for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) {
            if (colidx[e] < 0 || colidx[e] >= n) return fail(nullptr, NLMC_ERR_ARG, "nlmc_create: column index out of range");
            if (colidx[e] == k && vals[e] != 0.0) diag = true;
            if (vals[e] == 0.0) zero_vals = true;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, NLMC_ERR_HIP, "nlmc_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, NLMC_ERR_ARG, "nlmc_create: device index out of range");

    nlmc_ctx *c = new nlmc_ctx();
    auto bail = [&](int code) { g_create_error = c->err; nlmc_destroy(c); return code; };
#define CT(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e__); return bail(NLMC_ERR_HIP); } } while (0)
    CT(hipSetDevice(device));
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    c->cur = c->stream;
    c->n = n;
    c->big = n > NLMC_LDS_N || getenv("NLMC_FORCE_BIG") != nullptr;      // (the knob: tests compare the two sets of kernels at one size)
    c->n_pad = (n + 15) / 16 * 16;
    c->nnz = nnz;
    c->n_chains = n_chains;
    c->chain_base = chain_base;
    c->n_chains_global = n_chains_global;
    c->has_diag = diag;
    c->has_zero_vals = zero_vals;
    c->max_deg = max_deg;
    c->n_long = n_long;

    // fixed-point scales.  Energies: integers in units of 2^-escale, |E| <= sum|J|/2 + sum|h|.  Couplings of the "f32"
    // throughput path: Jq = rint(J 2^qs), hq = rint(h 2^qs) with qs the largest exponent such that every |Jq| fits a
    // signed 24-bit multiplier and every row sum  sum|Jq| + |hq|  fits int32 (the field is then an exact int32);
    // qs <= escale <= qs + 29, so that an energy delta is the field times +-2^(escale - qs + 1) in one 32 x 32 -> 64
    // bit multiply-add.  (Restated in oracle/nlo.c: nlo_field_scale.)
    double bound = 0.0, maxabs = 0.0, maxh = 0.0;
    for (int64_t e = 0; e < nnz; ++e) { bound += std::fabs(vals[e]) * 0.5; maxabs = std::max(maxabs, std::fabs(vals[e])); }
    for (int k = 0; k < n; ++k) { bound += std::fabs(h[k]); maxh = std::max(maxh, std::fabs(h[k])); }
    int ex = 0;
    std::frexp(std::max(bound, 1.0), &ex);       // bound < 2^ex
    const int escale0 = std::max(0, std::min(52, 60 - ex));
    int qs = 0;
    auto rq = [](double v, int q) { return (int64_t)std::llrint(std::ldexp(v, q)); };
    const double ref = maxabs > 0.0 ? maxabs : maxh;
    if (ref > 0.0) {
        int exr = 0;
        std::frexp(ref, &exr);
        qs = std::min(23 - exr, escale0);
        for (;;) {
            bool ok = true;
            for (int k = 0; k < n && ok; ++k) {
                int64_t row = std::llabs(rq(h[k], qs));
                for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) {
                    const int64_t q = std::llabs(rq(vals[e], qs));
                    if (q > 8388607) ok = false;
                    row += q;
                }
                if (row > 2147483647LL) ok = false;
            }
            if (ok) break;
            --qs;
        }
        // canonical form: drop the power of two common to every Jq and hq (+-J instances: Jq = +-1).  Where every value is an exact
        // multiple of 2^-qs (so that a smaller scale just shifts the integers) the common power is found in ONE pass -- the loop
        // below, one pass per bit, was 22 passes over the couplings for a +-J instance: most of the 6 ms of nlmc_create at N = 10^4.
        {
            bool exact = true, any = false;
            int min_tz = 63;
            auto scan = [&](double v) {
                const int64_t q = rq(v, qs);
                if (std::ldexp((double)q, -qs) != v) exact = false;
                if (q) { any = true; min_tz = std::min(min_tz, __builtin_ctzll((unsigned long long)(q < 0 ? -q : q))); }
            };
            for (int64_t e = 0; e < nnz && exact; ++e) scan(vals[e]);
            for (int k = 0; k < n && exact; ++k) scan(h[k]);
            if (exact && any && min_tz > 0) qs -= min_tz;        // (the loop below then stops at its first pass: some integer is odd)
        }
        for (;;) {
            bool even = true, any = false;
            for (int64_t e = 0; e < nnz && even; ++e) { const int64_t q = rq(vals[e], qs); if (q & 1) even = false; if (q) any = true; }
            for (int k = 0; k < n && even; ++k) { const int64_t q = rq(h[k], qs); if (q & 1) even = false; if (q) any = true; }
            if (!even || !any) break;
            --qs;
        }
    }
    c->qs = qs;
    c->escale = std::min(escale0, qs + 29);

    std::vector<EdgeQ> e32((size_t)std::max<int64_t>(nnz, 1));
    std::vector<int32_t> hq((size_t)n);
    bool fits16 = n <= 65535;        // compact schedule entries of the fused windows: col << 16 | (Jq & 0xFFFF)
    bool pm1 = true;                 // sign format: col | (Jq < 0) << 15
    bool exact_q = true;             // every J is Jq 2^-qs exactly (packed fp64 schedule window)
    for (int64_t e = 0; e < nnz; ++e) {
        e32[e].col = colidx[e]; e32[e].q = (int32_t)rq(vals[e], qs);
        if (e32[e].q > 32767 || e32[e].q < -32768) fits16 = false;
        if (std::ldexp((double)e32[e].q, -qs) != vals[e]) exact_q = false;
        if (e32[e].q != 1 && e32[e].q != -1) pm1 = false;
    }
    c->compact16 = fits16 && !getenv("NLMC_NO_COMPACT");
    c->f64_pack16 = fits16 && exact_q && n <= 65535 && !getenv("NLMC_NO_PACK64");
    c->sign8 = c->compact16 && pm1 && nnz > 0 && !getenv("NLMC_NO_SIGNFMT");
    c->knob_no_warm = getenv("NLMC_NO_WARM") != nullptr;
    c->knob_no_prio = getenv("NLMC_FUSED_NOPRIO") != nullptr;
    c->knob_no_dbuf = getenv("NLMC_NO_DBUF") != nullptr;
    if (const char *e = getenv("NLMC_DBG_FLAGS")) c->knob_dbg_flags = atoi(e);
    bool exact_h = true;
    for (int k = 0; k < n; ++k) {
        hq[k] = (int32_t)rq(h[k], qs);
        if (std::ldexp((double)hq[k], -qs) != h[k]) exact_h = false;
        int64_t row = std::llabs((int64_t)hq[k]);
        for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) row += std::llabs((int64_t)e32[e].q);
        c->xmax = (int)std::min<int64_t>(std::max<int64_t>(c->xmax, row), INT_MAX);
    }
    c->f64_exact = exact_q && exact_h;
    if (const char *e = getenv("NLMC_F64_TIE_MASK")) c->knob_tie_mask = (unsigned)strtoul(e, nullptr, 0);

    // +16 entries of padding behind the row arrays: fixed-width row windows are read unconditionally (never used past the row end)
    const size_t R = (size_t)std::max(n_chains, 1), npad = (size_t)c->n_pad, nz = (size_t)nnz + 16;
    const size_t total = DevBuf<int32_t>::arena_bytes((size_t)n + 1) + DevBuf<int32_t>::arena_bytes(nz) + DevBuf<double>::arena_bytes(nz) +
                         DevBuf<EdgeQ>::arena_bytes(nz) + DevBuf<double>::arena_bytes((size_t)n) + DevBuf<int32_t>::arena_bytes((size_t)n) +
                         3 * DevBuf<int8_t>::arena_bytes(R * npad) + 2 * DevBuf<long long>::arena_bytes(R) +
                         DevBuf<int32_t>::arena_bytes(R) + DevBuf<double>::arena_bytes(R) + 256;
    CT(hipMalloc(&c->arena, total));
    size_t off = 0;
    c->rowptr.borrow(c->arena, off, (size_t)n + 1);
    c->col.borrow(c->arena, off, nz);
    c->val64.borrow(c->arena, off, nz);
    c->edge32.borrow(c->arena, off, nz);
    c->h64.borrow(c->arena, off, (size_t)n);
    c->hq.borrow(c->arena, off, (size_t)n);
    c->spins.borrow(c->arena, off, R * npad);
    c->best.borrow(c->arena, off, R * npad);
    c->flags.borrow(c->arena, off, R * npad);
    c->efix.borrow(c->arena, off, R);
    c->emin.borrow(c->arena, off, R);
    c->argmin.borrow(c->arena, off, R);
    c->energy.borrow(c->arena, off, R);
    if (off > total) { c->err = "nlmc_create: arena accounting"; return bail(NLMC_ERR_HIP); }
    // everything zero (padding, states, flags, tracked energies), then the instance: stream-ordered, ONE wait (the host arrays
    // handed to the copies are the caller's and this function's own: both outlive the wait)
    CT(hipMemsetAsync(c->arena, 0, total, c->stream));
    CT(hipMemcpyAsync(c->rowptr.p, rowptr, sizeof(int32_t) * ((size_t)n + 1), hipMemcpyHostToDevice, c->stream));
    if (nnz > 0) {
        CT(hipMemcpyAsync(c->col.p, colidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, c->stream));
        CT(hipMemcpyAsync(c->val64.p, vals, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, c->stream));
        CT(hipMemcpyAsync(c->edge32.p, e32.data(), sizeof(EdgeQ) * (size_t)nnz, hipMemcpyHostToDevice, c->stream));
    }
    CT(hipMemcpyAsync(c->h64.p, h, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    CT(hipMemcpyAsync(c->hq.p, hq.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    CT(hipStreamSynchronize(c->stream));

    c->g.n = n; c->g.n_pad = c->n_pad;
    c->g.rowptr = c->rowptr.p; c->g.col = c->col.p; c->g.val64 = c->val64.p; c->g.edge32 = c->edge32.p;
    c->g.h64 = c->h64.p; c->g.hq = c->hq.p;

#undef CT
    *out = c;
    return NLMC_OK;
}

void nlmc_destroy(nlmc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
#ifdef NLMC_STAMPS
    if (const char *fn = getenv("NLMC_STAMP_FILE")) {      // diagnostic build: dump the last launch's per-wave cycle sums
        if (c->dbg.p) {
            std::vector<long long> hbuf((size_t)c->n_chains * 16 * 8 + 96);
            if (hipMemcpy(hbuf.data(), c->dbg.p, hbuf.size() * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE *f = fopen(fn, "wb")) { fwrite(hbuf.data(), sizeof(long long), hbuf.size(), f); fclose(f); }
            }
        }
    }
    c->dbg.release();
#endif
    if (c->comm && g_rccl.ok) (void)g_rccl.CommDestroy(c->comm);
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->owns_stream && c->stream) { (void)hipStreamDestroy(c->stream); c->stream = nullptr; }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    c->rowptr.release(); c->col.release(); c->val64.release(); c->h64.release(); c->edge32.release(); c->hq.release();
    // (borrowed buffers: released above / below without a free; the arena goes last)
    c->spins.release(); c->best.release(); c->flags.release(); c->efix.release(); c->emin.release(); c->etrace.release();
    c->argmin.release(); c->energy.release(); c->tab.release(); c->ustream.release(); c->etrace_d.release();
    c->keys.release(); c->perm_raw.release(); c->u_raw.release(); c->stream_bad.release(); c->strace.release(); c->cfg.release(); c->snap_g.release(); c->scratch.release(); c->plan.release();
    c->slot_mark.release(); c->sub_list_buf.release(); c->cmask.release(); c->cmask_scratch.release(); c->big_flag.release(); c->big_esum.release(); c->nmc_status.release(); c->nmc_thr.release();
    c->fz_glv.release(); c->fz_perm.release(); c->fz_adj.release(); c->fz_stats.release(); c->fz[0].release(); c->fz[1].release();
    c->lbp_src.release(); c->lbp_rev.release(); c->lbp_flag.release(); c->lbp_out_i.release(); c->lbp_tJ.release();
    c->lbp_eps.release(); c->lbp_ms.release(); c->lbp_lams.release(); c->lbp_w0.release(); c->lbp_w1.release();
    c->lbp_bar.release(); c->lbp_part.release();
    c->lbp_hm.release(); c->lbp_tot.release(); c->lbp_mag.release(); c->lbp_mag_all.release();
    c->pt_tab.release(); c->pt_beta.release(); c->pt_energies_all.release();
    c->rounds_ebuf.release(); c->rounds_bar.release(); c->rounds_args.release(); c->seed_snap.release();
    c->apt_beta.release(); c->apt_e_all.release(); c->apt_send.release(); c->apt_recv.release(); c->apt_bd.release();
    c->slot_of_chain.release(); c->chain_of_slot.release(); c->pt_pairs.release(); c->pt_status.release();
    c->pt_acc.release(); c->pt_log_acc.release(); c->pt_log_pairs.release(); c->pt_plan_pairs.release(); c->pt_plan_ok.release(); c->icm_label.release(); c->icm_info.release(); c->icm_pairs.release();
    if (c->arena) (void)hipFree(c->arena);
    delete c;
}

int nlmc_set_spins(nlmc_ctx *c, const int8_t *spins)
{
    if (!c || !spins) return fail(c, NLMC_ERR_ARG, "nlmc_set_spins: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    int rc = rows_to_device(c, c->spins.p, spins, c->n_chains);
    if (rc) return rc;
    rc = launch_energy_self(c, nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_get_spins(nlmc_ctx *c, int8_t *spins)
{
    if (!c || !spins) return fail(c, NLMC_ERR_ARG, "nlmc_get_spins: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    int rc = rows_to_host_begin(c, c->spins.p, c->n_chains);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rows_to_host_finish(c, spins, c->n_chains);
    return NLMC_OK;
}

int nlmc_set_flags(nlmc_ctx *c, const uint8_t *flags, double temp_x)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!flags) { c->has_flags = false; c->temp_x = 1.0; return NLMC_OK; }
    if (!(temp_x > 0.0) && !(temp_x < 0.0)) return fail(c, NLMC_ERR_ARG, "nlmc_set_flags: temp_x must be non-zero");
    for (size_t i = 0; i < (size_t)c->n_chains * c->n; ++i)
        if (flags[i] > 3) return fail(c, NLMC_ERR_ARG, "nlmc_set_flags: flag value out of range");
    if (c->n_chains > 0) {
        int rc = rows_to_device(c, c->flags.p, flags, c->n_chains);
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    c->has_flags = true;
    c->temp_x = temp_x;
    return NLMC_OK;
}

int nlmc_energy(nlmc_ctx *c, double *out)
{
    if (!c || !out) return fail(c, NLMC_ERR_ARG, "nlmc_energy: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    int rc = launch_energy_self(c, c->energy.p);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->energy.p, sizeof(double) * c->n_chains, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_energy_dev(nlmc_ctx *c, double *dev_out)
{
    if (!c || !dev_out) return fail(c, NLMC_ERR_ARG, "nlmc_energy_dev: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    hipLaunchKernelGGL(k_efix_to_double, dim3((c->n_chains + 255) / 256), dim3(256), 0, c->stream, c->efix.p, dev_out,
                       c->n_chains, c->escale);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int nlmc_energy_tracked(nlmc_ctx *c, double *out)
{
    if (!c || !out) return fail(c, NLMC_ERR_ARG, "nlmc_energy_tracked: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    std::vector<long long> e((size_t)c->n_chains);
    HIP_TRY(c, hipMemcpyAsync(e.data(), c->efix.p, sizeof(long long) * e.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const double inv = std::ldexp(1.0, -c->escale);
    for (size_t i = 0; i < e.size(); ++i) out[i] = (double)e[i] * inv;
    return NLMC_OK;
}

int nlmc_set_energy_sink(nlmc_ctx *c, double *dev_out)
{
    if (!c) return NLMC_ERR_ARG;
    c->energy_sink = dev_out;
    return NLMC_OK;
}

int nlmc_energy_scale(const nlmc_ctx *c) { return c ? c->escale : 0; }

int nlmc_field_scale(const nlmc_ctx *c) { return c ? c->qs : 0; }

int nlmc_energy_of(nlmc_ctx *c, const int8_t *spins, int64_t count, double *out)
{
    if (!c || !spins || !out || count < 0) return fail(c, NLMC_ERR_ARG, "nlmc_energy_of: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t chunk = std::max<int64_t>(1, ((int64_t)64 << 20) / c->n);
    for (int64_t b = 0; b < count; b += chunk) {
        const int64_t m = std::min(chunk, count - b);
        HIP_TRY(c, c->cfg.reserve((size_t)m * c->n));
        HIP_TRY(c, c->etrace_d.reserve((size_t)m));
        HIP_TRY(c, hipMemcpyAsync(c->cfg.p, spins + b * c->n, (size_t)m * c->n, hipMemcpyHostToDevice, c->stream));
        EnergyArgs a{};
        a.g = c->g; a.spins = c->cfg.p; a.stride = c->n; a.out = c->etrace_d.p; a.efix = nullptr; a.escale = c->escale;
        launch_energy(c, (unsigned)m, c->stream, a);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(out + b, c->etrace_d.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return NLMC_OK;
}

int nlmc_energy_of_recorded(nlmc_ctx *c, int first, int count, double *out)
{
    if (!c || !out || first < 0 || count < 0) return fail(c, NLMC_ERR_ARG, "nlmc_energy_of_recorded: bad argument");
    if (first + count > c->strace_nrec)
        return fail(c, NLMC_ERR_STATE, "nlmc_energy_of_recorded: the last sweep call recorded fewer configurations per chain");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t m = (int64_t)c->strace_rows * count;
    if (m == 0) return NLMC_OK;
    HIP_TRY(c, c->etrace_d.reserve((size_t)m));
    EnergyArgs a{};
    a.g = c->g; a.spins = c->strace.p + (size_t)first * c->n; a.stride = c->n; a.stride_outer = (int64_t)c->strace_nrec * c->n;
    a.inner = count; a.out = c->etrace_d.p; a.efix = nullptr; a.escale = c->escale;
    launch_energy(c, (unsigned)m, c->cur, a);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->etrace_d.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, c->cur));
    HIP_TRY(c, hipStreamSynchronize(c->cur));
    return NLMC_OK;
}

int nlmc_sweep_stream(nlmc_ctx *c, int n_sweeps, const int32_t *perm, const double *u, const double *beta,
                      int chain_stride, int sweep_stride, int record_stride, int8_t *out_spins, double *out_energy,
                      double *out_min_energy, int32_t *out_argmin, int8_t *out_argmin_state)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_sweeps < 0 || !beta || (n_sweeps > 0 && c->n_chains > 0 && (!perm || !u)) || (out_spins && record_stride < 1))
        return fail(c, NLMC_ERR_ARG, "nlmc_sweep_stream: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const int R = c->n_chains, n = c->n;
    if (R == 0 || n_sweeps == 0) return NLMC_OK;
    const size_t tot = (size_t)R * n_sweeps;
    // beta table -> dense [R][n_sweeps]
    std::vector<double> tab(tot);
    for (int r = 0; r < R; ++r)
        for (int t = 0; t < n_sweeps; ++t) tab[(size_t)r * n_sweeps + t] = beta[(size_t)r * chain_stride + (size_t)t * sweep_stride];
    HIP_TRY(c, c->keys.reserve(tot * n));
    HIP_TRY(c, c->ustream.reserve(tot * n));
    HIP_TRY(c, c->perm_raw.reserve(tot * n));
    HIP_TRY(c, c->u_raw.reserve(tot * n));
    HIP_TRY(c, c->stream_bad.reserve(1));
    c->tab_host.clear();
    HIP_TRY(c, c->tab.reserve(tot));
    // the stream goes up as it was drawn and is scattered by spin on the device (rank[perm[i]] = i, u_spin[perm[i]] = u[i]; the
    // host loop + its staging vectors were 9 + 9 ms per 2000 sweeps at N = 2048), which also checks that every row IS a permutation
    HIP_TRY(c, hipMemsetAsync(c->stream_bad.p, 0, sizeof(int32_t), c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->perm_raw.p, perm, sizeof(int32_t) * tot * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->u_raw.p, u, sizeof(double) * tot * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->tab.p, tab.data(), sizeof(double) * tot, hipMemcpyHostToDevice, c->stream));
    if (c->big || (size_t)n * 4 > (size_t)150 * 1024) {       // (the seen-marks of k_stream_scatter live in LDS)
        HIP_TRY(c, hipMemsetAsync(c->keys.p, 0xFF, sizeof(uint32_t) * tot * n, c->stream));
        for (size_t ob = 0; ob < tot; ob += 65535) {
            const size_t no = std::min<size_t>(65535, tot - ob), sh = ob * (size_t)n;
            hipLaunchKernelGGL(k_stream_scatter_big, dim3((unsigned)((n + 255) / 256), (unsigned)no), dim3(256), 0, c->stream, n, c->perm_raw.p + sh,
                               c->u_raw.p + sh, c->keys.p + sh, c->ustream.p + sh, c->stream_bad.p);
        }
    } else {
        { int rc = ensure_lds(c, 8, reinterpret_cast<const void *>(k_stream_scatter), (size_t)n * 4); if (rc) return rc; }
        hipLaunchKernelGGL(k_stream_scatter, dim3((unsigned)tot), dim3(256), (size_t)n * 4, c->stream, n, c->perm_raw.p, c->u_raw.p, c->keys.p,
                           c->ustream.p, c->stream_bad.p);
    }
    HIP_TRY(c, hipGetLastError());
    int32_t bad = 0;
    HIP_TRY(c, hipMemcpyAsync(&bad, c->stream_bad.p, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));          // (also: the caller's buffers and `tab` may go away)
    if (bad) return fail(c, NLMC_ERR_ARG, "nlmc_sweep_stream: perm is not a permutation");
    SweepOut o{record_stride, out_spins, out_energy, out_min_energy, out_argmin, out_argmin_state};
    int rc = run_sweeps(c, true, NLMC_F64, NLMC_ORDER_PER_CHAIN, n_sweeps, 0, 0, c->tab.p, n_sweeps, 1, false, c->keys.p,
                        c->ustream.p, o);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_sweep_philox(nlmc_ctx *c, int precision, int order_mode, int n_sweeps, uint32_t sweep0, uint64_t seed,
                      const double *beta, int chain_stride, int sweep_stride, int record_stride, int8_t *out_spins,
                      double *out_energy, double *out_min_energy, int32_t *out_argmin, int8_t *out_argmin_state)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_sweeps < 0 || (precision != NLMC_F32 && precision != NLMC_F64) ||
        (order_mode != NLMC_ORDER_SHARED && order_mode != NLMC_ORDER_PER_CHAIN) || (out_spins && record_stride < 1))
        return fail(c, NLMC_ERR_ARG, "nlmc_sweep_philox: bad argument");
    if (!beta && c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_sweep_philox: beta == NULL needs nlmc_pt_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    const int R = c->sub_count();
    if (R == 0 || n_sweeps == 0) return NLMC_OK;
    if (beta && chain_stride && c->subset != 0)
        return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_sweep_philox: a per-chain beta table with a chain subset (use the ladder or one beta)");
    const double *tab_dev;
    int tcs, tss;
    bool use_slots = false;
    if (beta) {
        const int rows = chain_stride ? R : 1, T = sweep_stride ? n_sweeps : 1;
        std::vector<double> tab((size_t)rows * T * 2);
        for (int r = 0; r < rows; ++r)
            for (int t = 0; t < T; ++t) {
                const double b = beta[(size_t)r * chain_stride + (size_t)t * sweep_stride];
                tab[((size_t)r * T + t) * 2 + 0] = -2.0 * LOG2E * b;
                tab[((size_t)r * T + t) * 2 + 1] = -2.0 * LOG2E * (b / c->temp_x);
            }
        if (tab != c->tab_host) {          // (the same table as last time -- every NMC phase of a run -- is not uploaded again)
            HIP_TRY(c, c->tab.reserve(tab.size()));
            HIP_TRY(c, hipStreamSynchronize(c->cur));         // launches still reading the previous table
            HIP_TRY(c, hipMemcpyAsync(c->tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, c->cur));
            HIP_TRY(c, hipStreamSynchronize(c->cur));
            c->tab_host.swap(tab);
        }
        tab_dev = c->tab.p;
        tcs = chain_stride ? T * 2 : 0;
        tss = sweep_stride ? 2 : 0;
    } else {
        // ladder table re-uploaded only when temp_x changed since the last upload (no host sync on the hot path)
        if (!c->pt_tab_valid || c->pt_tab_temp_x != c->temp_x) {
            std::vector<double> tab((size_t)c->ladder_len * 2);
            for (int r = 0; r < c->ladder_len; ++r) {
                tab[2 * r + 0] = -2.0 * LOG2E * c->beta_list[r];
                tab[2 * r + 1] = -2.0 * LOG2E * (c->beta_list[r] / c->temp_x);
            }
            HIP_TRY(c, hipMemcpyAsync(c->pt_tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, c->cur));
            HIP_TRY(c, hipStreamSynchronize(c->cur));
            c->pt_tab_valid = true;
            c->pt_tab_temp_x = c->temp_x;
        }
        tab_dev = c->pt_tab.p;
        tcs = 2; tss = 0;
        use_slots = true;
    }
    SweepOut o{record_stride, out_spins, out_energy, out_min_energy, out_argmin, out_argmin_state};
    return run_sweeps(c, false, precision, order_mode, n_sweeps, sweep0, seed, tab_dev, tcs, tss, use_slots, nullptr,
                      nullptr, o);
}

int nlmc_plan_philox(nlmc_ctx *c, int precision, int order_mode, uint32_t sweep0, int n_sweeps, uint64_t seed)
{
    if (!c) return NLMC_ERR_ARG;
    if (order_mode != NLMC_ORDER_SHARED) return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_plan_philox: only shared orders are cached");
    if (n_sweeps < 0 || (precision != NLMC_F32 && precision != NLMC_F64)) return fail(c, NLMC_ERR_ARG, "nlmc_plan_philox: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    c->plan_valid = false;
    if (n_sweeps == 0) return NLMC_OK;
    const bool big = sweeps_big(c, false, precision == NLMC_F64);
    const int ell_mode = big ? -1 : precision == NLMC_F64 ? 2 : 1;
    if (big) HIP_TRY(c, c->plan.reserve_big((size_t)n_sweeps, (size_t)c->n, true));
    else HIP_TRY(c, c->plan.reserve((size_t)n_sweeps, (size_t)c->n, ell_mode));
    hipEvent_t pe0 = nullptr, pe1 = nullptr, pe2 = nullptr;      // planning time -> levelize time of nlmc_timing_total
    if (c->ev_accumulate) {
        pe0 = next_event(c); pe1 = next_event(c); pe2 = next_event(c);
        if (!pe0 || !pe1 || !pe2) return fail(c, NLMC_ERR_HIP, "hipEventCreate failed");
        tag_triple(c, 2);
        HIP_TRY(c, hipEventRecord(pe0, c->stream));
    }
    int rc = run_levelize(c, n_sweeps, nullptr, 0, n_sweeps, sweep0, seed, c->plan, ell_mode);
    if (rc) return rc;
    if (pe0) HIP_TRY(c, hipEventRecord(pe2, c->stream));
    c->plan_valid = true;
    c->plan_big = big;
    c->plan_mode = order_mode;
    c->plan_precision = precision;
    c->plan_sweep0 = sweep0;
    c->plan_count = n_sweeps;
    c->plan_seed = seed;
    return NLMC_OK;
}

static int reserve_fused_plan(nlmc_ctx *c, int n_windows, int T)
{
    nlmc_ctx::FusedPlan &P = c->fz[c->fz_slot];
    const size_t W = (size_t)n_windows, TN = (size_t)T * c->n, PS = (size_t)fused_pstride(c->n, c->n_long, T);
    HIP_TRY(c, c->fz_glv.reserve(W * TN));
    HIP_TRY(c, c->fz_perm.reserve(W * PS));
    HIP_TRY(c, P.head.reserve(W * PS));
    HIP_TRY(c, P.ell.reserve(W * PS * NLMC_FZ_W));
    HIP_TRY(c, P.loff.reserve(W * (NLMC_LCAP + 1)));
    HIP_TRY(c, P.nlev.reserve(W));
    HIP_TRY(c, P.npos.reserve(W));
    HIP_TRY(c, P.himax.reserve(W));
    HIP_TRY(c, P.send.reserve(W * T));
    return NLMC_OK;
}

int nlmc_plan_reserve_fused(nlmc_ctx *c, int n_windows, int window)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_windows < 0 || window < 1) return fail(c, NLMC_ERR_ARG, "nlmc_plan_reserve_fused: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_windows == 0 || !fused_supported(c, window)) return NLMC_OK;
    c->fz[c->fz_slot].valid = false;              // (growing a buffer drops its contents)
    return reserve_fused_plan(c, n_windows, window);
}

int nlmc_plan_philox_fused(nlmc_ctx *c, uint32_t sweep0, int n_windows, int window, uint64_t seed, int32_t *out_planned)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_windows < 0 || window < 1) return fail(c, NLMC_ERR_ARG, "nlmc_plan_philox_fused: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    nlmc_ctx::FusedPlan &P = c->fz[c->fz_slot];
    P.valid = false;
    if (out_planned) *out_planned = 0;
    if (n_windows == 0 || !fused_supported(c, window)) return NLMC_OK;
    const int n = c->n, T = window;
    const size_t W = (size_t)n_windows;
    P.pstride = fused_pstride(n, c->n_long, T);
    if (reserve_fused_plan(c, n_windows, T) != NLMC_OK) {
        // no room for this many windows: not an error -- nothing is planned, the caller's sweeps take the
        // sweep-by-sweep path (or it asks again for fewer windows); what was allocated for the attempt is given back
        (void)hipGetLastError();
        P.release(); c->fz_glv.release(); c->fz_perm.release();
        c->err.clear();
        return NLMC_OK;
    }
    { int rc = ensure_adjacency(c); if (rc) return rc; }
    FusedLevelizeArgs a{};
    a.g = c->g;
    a.T = T;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.sweep0 = sweep0;
    P.workers = fused_workers(fused_block(c->n));
    P.gen0 = fused_gen0(fused_block(c->n));
    a.level_cap = P.workers * 64;
    a.pstride = P.pstride;
    a.tab_words = c->n_pad;
    a.k_dummy = c->n_pad;
    a.fmt = fused_addr_format(c) ? NLMC_FMT_ADDR : c->compact16 ? NLMC_FMT_COMPACT : NLMC_FMT_WIDE;
    a.k_zero = c->n_pad + 8;
    a.neg_off = c->n_pad + 16;
    a.bank_aware = getenv("NLMC_NO_BANK_AWARE") ? 0 : 1;
    P.fmt = a.fmt;
    a.adj = reinterpret_cast<const uint4 *>(c->fz_adj.p); a.glv = c->fz_glv.p; a.perm = c->fz_perm.p; a.head = P.head.p; a.ell = P.ell.p; a.loff = P.loff.p; a.nlev = P.nlev.p;
    a.hi_max = P.himax.p; a.send = P.send.p; a.npos = P.npos.p;
    const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
    const size_t lds = (size_t)n * 8 + n4 * 2 + n4 + n4 * 2 + 2 * (size_t)(NLMC_LCAP + 2) * 4 + 16;
    { int rc = ensure_lds(c, 16, reinterpret_cast<const void *>(k_levelize_fused), lds); if (rc) return rc; }
    const bool fz_diag = getenv("NLMC_FZ_STATS") != nullptr;     // diagnostic: phase cycle counts of window 0 on stderr
    if (fz_diag) { HIP_TRY(c, c->fz_stats.reserve(W * 8)); a.stats = c->fz_stats.p; }
    // planning time counts as levelize time of the accumulating timer (nlmc_timing_total): an event triple whose
    // sweep part is empty
    hipEvent_t pe0 = nullptr, pe1 = nullptr, pe2 = nullptr;
    if (c->ev_accumulate) {
        pe0 = next_event(c); pe1 = next_event(c); pe2 = next_event(c);
        if (!pe0 || !pe1 || !pe2) return fail(c, NLMC_ERR_HIP, "hipEventCreate failed");
        tag_triple(c, 2);
        HIP_TRY(c, hipEventRecord(pe0, c->stream));
    }
    hipLaunchKernelGGL(k_levelize_fused, dim3(n_windows), dim3(1024), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    if (pe0) HIP_TRY(c, hipEventRecord(pe2, c->stream));
    P.nlev_host.assign(W, 0);
    P.npos_host.assign(W, 0);
    HIP_TRY(c, hipMemcpyAsync(P.nlev_host.data(), P.nlev.p, sizeof(int32_t) * W, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(P.npos_host.data(), P.npos.p, sizeof(int32_t) * W, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (fz_diag) {
        long long st[8] = {0};
        HIP_TRY(c, hipMemcpy(st, c->fz_stats.p, sizeof(st), hipMemcpyDeviceToHost));
        fprintf(stderr, "k_levelize_fused window 0 (cycles): keys %lld, init %lld, passes %lld (%lld passes), place-1 %lld, place-2 %lld\n", st[0], st[1], st[2], st[3], st[4], st[5]);
    }
    int ok = 0;
    for (int32_t v : P.nlev_host) ok += v > 0;
    P.valid = true;
    P.sweep0 = sweep0; P.windows = n_windows; P.T = T; P.seed = seed;
    if (out_planned) *out_planned = ok;
    return NLMC_OK;
}

int nlmc_fused_modes(nlmc_ctx *c, int window)
{
    if (!c) return 0;
    return (fused_supported(c, window) ? 1 : 0) | (fused_f64_supported(c, window) ? 2 : 0);
}

int nlmc_probe_level_round(nlmc_ctx *c, int waves, int conflict_free, int rounds, int n_workgroups, double *out_ns_per_round)
{
    if (!c || !out_ns_per_round || waves < 1 || waves > 16 || rounds < 1 || n_workgroups < 1)
        return fail(c, NLMC_ERR_ARG, "nlmc_probe_level_round: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const int lds_bytes = 2 * 16384;
    int *sink = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&sink), sizeof(int)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = NLMC_OK;
    float ms = 0.f;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = fail(c, NLMC_ERR_HIP, "hipEventCreate failed");
    if (!rc) {
        hipLaunchKernelGGL(k_level_round_probe, dim3(n_workgroups), dim3(waves * 64), (size_t)lds_bytes, c->stream, std::min(rounds, 256), conflict_free, lds_bytes, sink);   // warm-up
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL(k_level_round_probe, dim3(n_workgroups), dim3(waves * 64), (size_t)lds_bytes, c->stream, rounds, conflict_free, lds_bytes, sink);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = fail(c, NLMC_ERR_HIP, "nlmc_probe_level_round: launch failed");
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (!rc) *out_ns_per_round = (double)ms * 1e6 / rounds;
    return rc;
}

int nlmc_plan_get_levels(nlmc_ctx *c, int window, int32_t *out_level_chunk_offsets, int32_t capacity, int32_t *out_n_levels)
{
    if (!c || !out_level_chunk_offsets || !out_n_levels) return fail(c, NLMC_ERR_ARG, "nlmc_plan_get_levels: NULL argument");
    const nlmc_ctx::FusedPlan &P = c->fz[c->fz_slot];
    if (!P.valid || window < 0 || window >= P.windows) return fail(c, NLMC_ERR_STATE, "nlmc_plan_get_levels: no such planned window in the selected slot");
    const int nl = P.nlev_host[(size_t)window];
    *out_n_levels = nl;
    if (capacity < nl + 1) return fail(c, NLMC_ERR_ARG, "nlmc_plan_get_levels: capacity < levels + 1");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out_level_chunk_offsets, P.loff.p + (size_t)window * (NLMC_LCAP + 1), sizeof(int32_t) * (size_t)(nl + 1),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_last_timing(nlmc_ctx *c, float *ms_levelize, float *ms_sweep, int32_t *launches_sweep)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    float lev = 0.f, sw = 0.f;
    for (size_t i = c->ev_call_start; i + 3 <= c->ev_used; i += 3) {
        float a = 0.f, b = 0.f;
        int rc = triple_ms(c, i / 3, a, b);
        if (rc) return rc;
        lev += a; sw += b;
    }
    if (ms_levelize) *ms_levelize = lev;
    if (ms_sweep) *ms_sweep = sw;
    if (launches_sweep) *launches_sweep = c->launches_sweep;
    return NLMC_OK;
}

int nlmc_timing_reset(nlmc_ctx *c, int enable)
{
    if (!c || enable < 0) return NLMC_ERR_ARG;
    c->ev_used = 0;
    c->ev_call_start = 0;
    c->launches_total = 0;
    c->launches_timed = 0;
    c->ev_accumulate = enable != 0;
    c->ev_every = enable > 1 ? enable : 1;
    return NLMC_OK;
}

int nlmc_timing_total(nlmc_ctx *c, double *ms_levelize, double *ms_sweep, int64_t *launches_sweep, int64_t *launches_timed)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    double lev = 0.0, sw = 0.0;
    for (size_t i = 0; i + 3 <= c->ev_used; i += 3) {
        float a = 0.f, b = 0.f;
        int rc = triple_ms(c, i / 3, a, b);
        if (rc) return rc;
        lev += a; sw += b;
    }
    if (ms_levelize) *ms_levelize = lev;
    if (ms_sweep) *ms_sweep = sw;
    if (launches_sweep) *launches_sweep = c->launches_total;
    if (launches_timed) *launches_timed = c->launches_timed;
    return NLMC_OK;
}

int nlmc_last_schedule_stats(nlmc_ctx *c, int64_t *n_orders, int64_t *n_levels)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->stat_fused_window >= 0) {       // one merged level list for fz_T sweeps
        if (n_orders) *n_orders = c->fz[c->stat_fused_slot].T;
        if (n_levels) *n_levels = c->fz[c->stat_fused_slot].nlev_host[(size_t)c->stat_fused_window];
        return NLMC_OK;
    }
    const int32_t *p = c->stats_pending ? c->stats_nlev_ptr : (c->plan_valid ? c->plan.nlev.p : nullptr);
    const int64_t cnt = c->stats_pending ? c->stats_nlev_count : (c->plan_valid ? c->plan_count : 0);
    int64_t lv = 0;
    if (p && cnt > 0) {
        std::vector<int32_t> hnl((size_t)cnt);
        HIP_TRY(c, hipMemcpyAsync(hnl.data(), p, sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int32_t v : hnl) lv += v;
    }
    if (n_orders) *n_orders = cnt;
    if (n_levels) *n_levels = lv;
    return NLMC_OK;
}

// ---------------------------------------------------------------------------------------------------
// replica exchange
// ---------------------------------------------------------------------------------------------------
int nlmc_pt_init(nlmc_ctx *c, int ladder_len, const double *beta_list)
{
    if (!c || !beta_list || ladder_len < 1) return fail(c, NLMC_ERR_ARG, "nlmc_pt_init: bad argument");
    if (c->n_chains_global % ladder_len != 0) return fail(c, NLMC_ERR_ARG, "nlmc_pt_init: n_chains_global not a multiple of ladder_len");
    HIP_TRY(c, hipSetDevice(c->device));
    c->ladder_len = ladder_len;
    c->subset = 0; c->n_marked_local = 0; c->sub_dirty = true;
    c->pt_tab_valid = false;
    c->pt_plan_valid = false;
    c->beta_list.assign(beta_list, beta_list + ladder_len);
    const int G = c->n_chains_global;
    HIP_TRY(c, c->pt_tab.reserve((size_t)ladder_len * 2));
    HIP_TRY(c, c->pt_beta.reserve((size_t)ladder_len));
    HIP_TRY(c, c->slot_of_chain.reserve((size_t)G));
    HIP_TRY(c, c->chain_of_slot.reserve((size_t)G));
    HIP_TRY(c, c->pt_status.reserve(1));
    std::vector<int32_t> ident((size_t)G);
    for (int i = 0; i < G; ++i) ident[i] = i % ladder_len;
    // (stream-ordered: the context's stream may be a non-blocking one, which legacy NULL-stream copies are not ordered against)
    std::vector<int32_t> cos((size_t)G);
    for (int i = 0; i < G; ++i) cos[i] = i;     // chain_of_slot[ladder*L + slot] = global chain id
    HIP_TRY(c, hipMemcpyAsync(c->slot_of_chain.p, ident.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->chain_of_slot.p, cos.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->pt_beta.p, beta_list, sizeof(double) * ladder_len, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->pt_status.p, 0, sizeof(int32_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_pt_plan(nlmc_ctx *c, uint32_t round0, int n_rounds, uint64_t seed, int n_pairs)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_plan: call nlmc_pt_init first");
    // (a context cut by temperature slot plans the selections of the GLOBAL ladder of its K sub-replica ladders: every rank the same)
    const int nl = c->n_chains_global / c->ladder_len, L = c->apt_R > 0 ? c->apt_R : c->ladder_len;
    if (n_rounds < 0 || n_pairs < 0 || n_pairs > std::max(0, L - 1)) return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    if (L > 4096) return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_pt_plan: ladder_len > 4096");
    HIP_TRY(c, hipSetDevice(c->device));
    c->pt_plan_valid = false;
    if (n_rounds == 0 || n_pairs == 0) return NLMC_OK;
    HIP_TRY(c, c->pt_plan_pairs.reserve((size_t)n_rounds * nl * n_pairs * 2));
    HIP_TRY(c, c->pt_plan_ok.reserve((size_t)n_rounds * nl));
    PtSelectArgs a{};
    a.ladder_len = L; a.n_pairs = n_pairs; a.n_ladders = nl; a.round0 = round0;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.plan_pairs = c->pt_plan_pairs.p; a.plan_ok = c->pt_plan_ok.p;
    hipLaunchKernelGGL(k_pt_select, dim3(n_rounds * nl), dim3(64), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    // the selection depends on the RNG only: an exhausted greedy selection (NPT/npt.py:526 raises ValueError in that
    // round) is known now -- report it here instead of silently skipping that round's swaps later
    std::vector<int32_t> ok((size_t)n_rounds * nl);
    HIP_TRY(c, hipMemcpyAsync(ok.data(), c->pt_plan_ok.p, sizeof(int32_t) * ok.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int32_t v : ok)
        if (!v) return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    c->pt_plan_valid = true;
    c->pt_plan_round0 = round0; c->pt_plan_rounds = n_rounds; c->pt_plan_npairs = n_pairs; c->pt_plan_seed = seed;
    return NLMC_OK;
}

int nlmc_pt_get_slots(nlmc_ctx *c, int32_t *slot_of_chain)
{
    if (!c || !slot_of_chain) return fail(c, NLMC_ERR_ARG, "nlmc_pt_get_slots: NULL argument");
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_get_slots: call nlmc_pt_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(slot_of_chain, c->slot_of_chain.p, sizeof(int32_t) * c->n_chains_global, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_pt_set_slots(nlmc_ctx *c, const int32_t *slot_of_chain)
{
    if (!c || !slot_of_chain) return fail(c, NLMC_ERR_ARG, "nlmc_pt_set_slots: NULL argument");
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_set_slots: call nlmc_pt_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    const int G = c->n_chains_global, L = c->ladder_len;
    std::vector<int32_t> cos((size_t)G, -1);
    for (int i = 0; i < G; ++i) {
        const int s = slot_of_chain[i];
        if (s < 0 || s >= L || cos[(size_t)(i / L) * L + s] != -1) return fail(c, NLMC_ERR_ARG, "nlmc_pt_set_slots: not a permutation per ladder");
        cos[(size_t)(i / L) * L + s] = i;
    }
    c->sub_dirty = true;
    HIP_TRY(c, hipMemcpyAsync(c->slot_of_chain.p, slot_of_chain, sizeof(int32_t) * G, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->chain_of_slot.p, cos.data(), sizeof(int32_t) * G, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_pt_apply_swap(nlmc_ctx *c, int ladder, int slot_a, int slot_b)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_apply_swap: call nlmc_pt_init first");
    const int L = c->ladder_len;
    if (ladder < 0 || ladder >= c->n_chains_global / L || slot_a < 0 || slot_a >= L || slot_b < 0 || slot_b >= L || slot_a == slot_b)
        return fail(c, NLMC_ERR_ARG, "nlmc_pt_apply_swap: bad ladder/slot");
    HIP_TRY(c, hipSetDevice(c->device));
    c->sub_dirty = true;
    hipLaunchKernelGGL(k_pt_apply_swap, dim3(1), dim3(1), 0, c->stream, c->slot_of_chain.p, c->chain_of_slot.p, L, ladder, slot_a, slot_b);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int nlmc_pt_swap_philox(nlmc_ctx *c, uint32_t round, uint64_t seed, int n_pairs, const double *energies_all_dev,
                        int32_t *out_pairs, uint8_t *out_accepted)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_swap_philox: call nlmc_pt_init first");
    const int L = c->ladder_len, G = c->n_chains_global, nl = G / L;
    if (n_pairs < 0 || n_pairs > std::max(0, L - 1)) return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    // a context that owns whole ladders needs nobody else's energies: it decides its own ladders from its tracked energies
    // (same Philox keys -- round, GLOBAL ladder index, pair -- as the context that holds all chains)
    const bool whole_ladders = c->chain_base % L == 0 && c->n_chains % L == 0;
    if (!energies_all_dev && !whole_ladders)
        return fail(c, NLMC_ERR_ARG, "nlmc_pt_swap_philox: a context whose block cuts a ladder needs the all-gathered energies");
    if (L > 4096) return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_pt_swap_philox: ladder_len > 4096");
    if (c->apt_R > 0 && c->apt_world > 1) return fail(c, NLMC_ERR_STATE, "nlmc_pt_swap_philox: this context is one slot block of a cut ladder (nlmc_apt_shard): use nlmc_apt_swap_*");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_pairs == 0) return NLMC_OK;
    HIP_TRY(c, c->pt_pairs.reserve((size_t)nl * n_pairs * 2));
    HIP_TRY(c, c->pt_acc.reserve((size_t)nl * n_pairs));
    PtSwapArgs a{};
    a.ladder_len = L; a.n_pairs = n_pairs; a.round = round;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.beta = c->pt_beta.p;
    a.energies = energies_all_dev;
    a.efix = energies_all_dev ? nullptr : c->efix.p;
    a.escale = c->escale;
    a.slot_of_chain = c->slot_of_chain.p; a.chain_of_slot = c->chain_of_slot.p;
    a.out_pairs = c->pt_pairs.p; a.out_acc = c->pt_acc.p; a.status = c->pt_status.p;
    a.ladder0 = energies_all_dev ? 0 : c->chain_base / L;
    a.chain_base = c->chain_base;
    const int n_decide = energies_all_dev ? nl : c->n_chains / L;            // ladders this launch decides
    if (c->pt_log_on && c->pt_log_npairs == n_pairs && round >= c->pt_log_round0 &&
        round < c->pt_log_round0 + (uint32_t)c->pt_log_rounds && !out_pairs && !out_accepted) {
        const size_t r = round - c->pt_log_round0;       // device-side log: read back once (nlmc_pt_log_read)
        a.out_pairs = c->pt_log_pairs.p + r * (size_t)nl * n_pairs * 2;
        a.out_acc = c->pt_log_acc.p + r * (size_t)nl * n_pairs;
    }
    if (c->pt_plan_valid && c->pt_plan_seed == seed && c->pt_plan_npairs == n_pairs && round >= c->pt_plan_round0 &&
        round < c->pt_plan_round0 + (uint32_t)c->pt_plan_rounds) {
        const size_t r = round - c->pt_plan_round0;
        a.plan_pairs = c->pt_plan_pairs.p + r * (size_t)nl * n_pairs * 2;
        a.plan_ok = c->pt_plan_ok.p + r * (size_t)nl;
    }
    // one lane per selected pair when the selection is planned (the in-kernel selection is written for ONE wave)
    const int swap_nt = a.plan_pairs ? std::min(256, (n_pairs + 63) / 64 * 64) : 64;
    c->sub_dirty = true;
    if ((out_pairs || out_accepted) && n_decide < nl) {          // rows of the ladders other contexts decide: "no pair"
        HIP_TRY(c, hipMemsetAsync(c->pt_pairs.p, 0xFF, sizeof(int32_t) * (size_t)nl * n_pairs * 2, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->pt_acc.p, 0, (size_t)nl * n_pairs, c->stream));
    }
    hipLaunchKernelGGL(k_pt_swap, dim3(n_decide), dim3(std::max(64, swap_nt)), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    if (out_pairs || out_accepted) {
        int32_t st = 0;
        if (out_pairs) HIP_TRY(c, hipMemcpyAsync(out_pairs, c->pt_pairs.p, sizeof(int32_t) * (size_t)nl * n_pairs * 2, hipMemcpyDeviceToHost, c->stream));
        if (out_accepted) HIP_TRY(c, hipMemcpyAsync(out_accepted, c->pt_acc.p, (size_t)nl * n_pairs, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&st, c->pt_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (st != 0) {
            HIP_TRY(c, hipMemsetAsync(c->pt_status.p, 0, sizeof(int32_t), c->stream));
            return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
        }
    }
    return NLMC_OK;
}

int nlmc_pt_swap_philox_host(nlmc_ctx *c, uint32_t round, uint64_t seed, int n_pairs, const double *energies_all_host,
                             int32_t *out_pairs, uint8_t *out_accepted)
{
    if (!c || !energies_all_host) return fail(c, NLMC_ERR_ARG, "nlmc_pt_swap_philox_host: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->pt_energies_all.reserve((size_t)c->n_chains_global));
    HIP_TRY(c, hipMemcpyAsync(c->pt_energies_all.p, energies_all_host, sizeof(double) * (size_t)c->n_chains_global,
                              hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // the caller's buffer may go away after the call
    return nlmc_pt_swap_philox(c, round, seed, n_pairs, c->pt_energies_all.p, out_pairs, out_accepted);
}

// ---- replica-sharded ladders: the ONE collective of a round, issued by the library on the kernels' stream -----------------
int nlmc_comm_unique_id(uint8_t *out_id)
{
    if (!out_id) return fail(nullptr, NLMC_ERR_ARG, "nlmc_comm_unique_id: NULL argument");
    if (!rccl_load()) return fail(nullptr, NLMC_ERR_UNSUPPORTED, g_rccl.why);
    Rccl::UniqueId id;
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) return fail(nullptr, NLMC_ERR_HIP, "ncclGetUniqueId: " + rccl_err(rc));
    std::memcpy(out_id, id.internal, sizeof(id.internal));
    return NLMC_OK;
}

int nlmc_comm_init(nlmc_ctx *c, const uint8_t *id, int world, int rank)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(c, NLMC_ERR_ARG, "nlmc_comm_init: bad argument");
    const bool apt = c->apt_R > 0;         // cut by temperature slot (nlmc_apt_shard): a self-contained block of chains per rank
    if (apt ? (c->apt_world != world || c->apt_rank != rank)
            : (c->n_chains * world != c->n_chains_global || c->chain_base != rank * c->n_chains))
        return fail(c, NLMC_ERR_ARG, "nlmc_comm_init: the context must own block `rank` of `world` equal blocks of chains (or be shard `rank` of `world` of nlmc_apt_shard)");
    if (!rccl_load()) return fail(c, NLMC_ERR_UNSUPPORTED, g_rccl.why);
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    Rccl::UniqueId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    const int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (rc != 0) { c->comm = nullptr; return fail(c, NLMC_ERR_HIP, "ncclCommInitRank: " + rccl_err(rc)); }
    c->comm_world = world; c->comm_rank = rank;
    if (apt) return NLMC_OK;               // (its gathered vector is apt_e_all, filled by k_apt_pack)
    HIP_TRY(c, c->pt_energies_all.reserve((size_t)c->n_chains_global));
    HIP_TRY(c, hipMemsetAsync(c->pt_energies_all.p, 0, sizeof(double) * (size_t)c->n_chains_global, c->stream));
    // the sweep kernels store their chains' tracked energies straight into this rank's block of the gathered vector
    c->energy_sink = c->pt_energies_all.p + c->chain_base;
    return NLMC_OK;
}

int nlmc_pt_swap_philox_collective(nlmc_ctx *c, uint32_t round, uint64_t seed, int n_pairs, int refresh_energies,
                                   int32_t *out_pairs, uint8_t *out_accepted)
{
    if (!c) return NLMC_ERR_ARG;
    if (!c->comm) return fail(c, NLMC_ERR_STATE, "nlmc_pt_swap_philox_collective: call nlmc_comm_init first");
    if (c->apt_R > 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_swap_philox_collective: this context is a slot block of nlmc_apt_shard: use nlmc_apt_swap_collective");
    HIP_TRY(c, hipSetDevice(c->device));
    double *block = c->pt_energies_all.p + c->chain_base;
    if (refresh_energies || c->energy_sink != block) {        // (no sweep since the last state change wrote the block)
        hipLaunchKernelGGL(k_efix_to_double, dim3((c->n_chains + 255) / 256), dim3(256), 0, c->stream, c->efix.p, block, c->n_chains, c->escale);
        HIP_TRY(c, hipGetLastError());
    }
    // in-place all-gather (send = this rank's block of the receive buffer), on the stream the sweep and swap kernels use:
    // stream order is the only synchronisation a round needs
    const int rc = g_rccl.AllGather(block, c->pt_energies_all.p, (size_t)c->n_chains, /*ncclDouble*/ 8, c->comm, c->stream);
    if (rc != 0) return fail(c, NLMC_ERR_HIP, "ncclAllGather: " + rccl_err(rc));
    return nlmc_pt_swap_philox(c, round, seed, n_pairs, c->pt_energies_all.p, out_pairs, out_accepted);
}

int nlmc_comm_probe(void)
{
    return rccl_load() ? NLMC_OK : fail(nullptr, NLMC_ERR_UNSUPPORTED, g_rccl.why);
}

// Asynchronous failures of the library-issued collectives: ncclCommGetAsyncError, and a BOUNDED wait for the stream the collectives
// were queued on (a lost rank leaves the others spinning inside a collective kernel for ever).  On either, the communicator is
// aborted (ncclCommAbort ends the kernels that wait for the lost peer) and the call fails; the context keeps no communicator.
int nlmc_comm_check(nlmc_ctx *c, int timeout_ms)
{
    if (!c) return NLMC_ERR_ARG;
    if (!c->comm) return NLMC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    auto abort_comm = [&](const std::string &why) {
        (void)g_rccl.CommAbort(c->comm);
        c->comm = nullptr;
        return fail(c, NLMC_ERR_HIP, why + "; the RCCL communicator was aborted (ncclCommAbort)");
    };
    int async = 0;
    int rc = g_rccl.CommGetAsyncError(c->comm, &async);
    if (rc != 0) return abort_comm("ncclCommGetAsyncError: " + rccl_err(rc));
    if (async != 0) return abort_comm("RCCL asynchronous error: " + rccl_err(async));
    if (timeout_ms < 0) return NLMC_OK;                      // error flag only, no wait
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return abort_comm(std::string("hipStreamQuery: ") + hipGetErrorString(q));
        rc = g_rccl.CommGetAsyncError(c->comm, &async);
        if (rc != 0 || async != 0) return abort_comm("RCCL asynchronous error: " + rccl_err(rc != 0 ? rc : async));
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
            return abort_comm("the collectives queued on the context's stream did not finish within " + std::to_string(timeout_ms) + " ms (a rank lost?)");
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    return NLMC_OK;
}

// ---- APT run cut into temperature-slot blocks over ranks (csrc/nlmc_apt.h) ---------------------------------------------------
int nlmc_apt_shard(nlmc_ctx *c, int R_global, int world, int rank, const double *beta_global)
{
    if (!c || !beta_global) return fail(c, NLMC_ERR_ARG, "nlmc_apt_shard: NULL argument");
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_apt_shard: call nlmc_pt_init with this rank's block of the ladder first");
    const int L = c->ladder_len;
    if (world < 1 || rank < 0 || rank >= world || R_global != world * L)
        return fail(c, NLMC_ERR_ARG, "nlmc_apt_shard: R_global must be world x the local ladder length");
    if (c->chain_base != 0 || c->n_chains != c->n_chains_global)
        return fail(c, NLMC_ERR_ARG, "nlmc_apt_shard: the context must be self-contained (K ladders of its own slots: chain_base 0, n_chains_global == n_chains)");
    for (int i = 0; i < L; ++i)
        if (beta_global[(size_t)rank * L + i] != c->beta_list[(size_t)i])
            return fail(c, NLMC_ERR_ARG, "nlmc_apt_shard: nlmc_pt_init was not given block `rank` of beta_global");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t K = (size_t)(c->n_chains / L);
    HIP_TRY(c, c->apt_beta.reserve((size_t)R_global));
    HIP_TRY(c, c->apt_e_all.reserve((size_t)world * K * L));
    HIP_TRY(c, c->apt_send.reserve(2 * K * (size_t)c->n_pad));
    HIP_TRY(c, c->apt_recv.reserve(2 * K * (size_t)c->n_pad));
    HIP_TRY(c, c->apt_bd.reserve(2 * K));
    HIP_TRY(c, hipMemcpyAsync(c->apt_beta.p, beta_global, sizeof(double) * (size_t)R_global, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->apt_e_all.p, 0, sizeof(long long) * (size_t)world * K * L, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->apt_recv.p, 0, 2 * K * (size_t)c->n_pad, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->apt_bd.p, 0, sizeof(int32_t) * 2 * K, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->apt_R = R_global; c->apt_world = world; c->apt_rank = rank;
    c->rng_stride = R_global; c->rng_base = rank * L;
    c->pt_plan_valid = false;                        // (a selection planned over the local ladder is not this mode's)
    return NLMC_OK;
}

static int apt_require(nlmc_ctx *c, const char *who)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->apt_R == 0) return fail(c, NLMC_ERR_STATE, std::string(who) + ": call nlmc_apt_shard first");
    return NLMC_OK;
}

static int apt_launch_pack(nlmc_ctx *c)
{
    const int L = c->ladder_len, K = c->n_chains / L;
    AptPackArgs a{};
    a.L = L; a.K = K; a.n_pad = c->n_pad;
    a.chain_of_slot = c->chain_of_slot.p; a.efix = c->efix.p; a.spins = c->spins.p;
    a.e_block = c->apt_e_all.p + (size_t)c->apt_rank * K * L;
    a.send = c->apt_send.p;
    hipLaunchKernelGGL(k_apt_pack, dim3(2 * K + 1), dim3(256), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

// decision of every selected pair + label exchanges inside the block + adoption of the accepted boundary pairs
static int apt_launch_swap(nlmc_ctx *c, uint32_t round, uint64_t seed, int n_pairs, int32_t *out_pairs, uint8_t *out_accepted)
{
    const int L = c->ladder_len, K = c->n_chains / L;
    HIP_TRY(c, c->pt_pairs.reserve((size_t)K * n_pairs * 2));
    HIP_TRY(c, c->pt_acc.reserve((size_t)K * n_pairs));
    AptSwapArgs a{};
    a.L = L; a.R = c->apt_R; a.K = K; a.n_pairs = n_pairs; a.world = c->apt_world; a.rank = c->apt_rank;
    a.round = round; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.beta = c->apt_beta.p; a.e_all = c->apt_e_all.p; a.escale = c->escale;
    a.slot_of_chain = c->slot_of_chain.p; a.chain_of_slot = c->chain_of_slot.p;
    a.out_pairs = c->pt_pairs.p; a.out_acc = c->pt_acc.p; a.status = c->pt_status.p; a.bd = c->apt_bd.p;
    if (c->pt_log_on && c->pt_log_npairs == n_pairs && round >= c->pt_log_round0 &&
        round < c->pt_log_round0 + (uint32_t)c->pt_log_rounds && !out_pairs && !out_accepted) {
        const size_t r = round - c->pt_log_round0;
        a.out_pairs = c->pt_log_pairs.p + r * (size_t)K * n_pairs * 2;
        a.out_acc = c->pt_log_acc.p + r * (size_t)K * n_pairs;
    }
    if (c->pt_plan_valid && c->pt_plan_seed == seed && c->pt_plan_npairs == n_pairs && round >= c->pt_plan_round0 &&
        round < c->pt_plan_round0 + (uint32_t)c->pt_plan_rounds) {
        const size_t r = round - c->pt_plan_round0;
        a.plan_pairs = c->pt_plan_pairs.p + r * (size_t)K * n_pairs * 2;
        a.plan_ok = c->pt_plan_ok.p + r * (size_t)K;
    }
    const int swap_nt = a.plan_pairs ? std::min(256, (n_pairs + 63) / 64 * 64) : 64;
    c->sub_dirty = true;
    hipLaunchKernelGGL(k_apt_swap, dim3(K), dim3(std::max(64, swap_nt)), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    if (c->apt_world > 1) {
        AptAdoptArgs d{};
        d.L = L; d.K = K; d.n_pad = c->n_pad; d.world = c->apt_world; d.rank = c->apt_rank; d.escale = c->escale;
        d.bd = c->apt_bd.p; d.chain_of_slot = c->chain_of_slot.p; d.recv = c->apt_recv.p; d.e_all = c->apt_e_all.p;
        d.spins = c->spins.p; d.efix = c->efix.p; d.energy_sink = c->energy_sink;
        hipLaunchKernelGGL(k_apt_adopt, dim3(2 * K), dim3(256), 0, c->stream, d);
        HIP_TRY(c, hipGetLastError());
    }
    if (out_pairs || out_accepted) {
        int32_t st = 0;
        if (out_pairs) HIP_TRY(c, hipMemcpyAsync(out_pairs, c->pt_pairs.p, sizeof(int32_t) * (size_t)K * n_pairs * 2, hipMemcpyDeviceToHost, c->stream));
        if (out_accepted) HIP_TRY(c, hipMemcpyAsync(out_accepted, c->pt_acc.p, (size_t)K * n_pairs, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&st, c->pt_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (st != 0) {
            HIP_TRY(c, hipMemsetAsync(c->pt_status.p, 0, sizeof(int32_t), c->stream));
            return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
        }
    }
    return NLMC_OK;
}

static int apt_check_pairs(nlmc_ctx *c, int n_pairs)
{
    if (n_pairs < 0 || n_pairs > std::max(0, c->apt_R - 1)) return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    if (c->apt_R > 4096) return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_apt_swap: ladder_len > 4096");
    return NLMC_OK;
}

int nlmc_apt_pack(nlmc_ctx *c, int64_t *out_slot_efix, int8_t *out_lo, int8_t *out_hi)
{
    { int rc = apt_require(c, "nlmc_apt_pack"); if (rc) return rc; }
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = apt_launch_pack(c); if (rc) return rc; }
    const int L = c->ladder_len, K = c->n_chains / L;
    if (out_slot_efix)
        HIP_TRY(c, hipMemcpyAsync(out_slot_efix, c->apt_e_all.p + (size_t)c->apt_rank * K * L, sizeof(long long) * (size_t)K * L, hipMemcpyDeviceToHost, c->stream));
    for (int side = 0; side < 2; ++side) {
        int8_t *dst = side ? out_hi : out_lo;
        if (!dst) continue;
        c->stage_out.resize((size_t)c->n_pad * K);
        HIP_TRY(c, hipMemcpyAsync(c->stage_out.data(), c->apt_send.p + (size_t)side * K * c->n_pad, c->stage_out.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        rows_to_host_finish(c, dst, K);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_apt_swap_host(nlmc_ctx *c, uint32_t round, uint64_t seed, int n_pairs, const int64_t *efix_all, const int8_t *recv_lo,
                       const int8_t *recv_hi, int32_t *out_pairs, uint8_t *out_accepted)
{
    { int rc = apt_require(c, "nlmc_apt_swap_host"); if (rc) return rc; }
    { int rc = apt_check_pairs(c, n_pairs); if (rc) return rc; }
    if (!efix_all || (c->apt_rank > 0 && !recv_lo) || (c->apt_rank + 1 < c->apt_world && !recv_hi))
        return fail(c, NLMC_ERR_ARG, "nlmc_apt_swap_host: the gathered energies and the neighbours' boundary configurations are required");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_pairs == 0) return NLMC_OK;
    const int L = c->ladder_len, K = c->n_chains / L;
    HIP_TRY(c, hipMemcpyAsync(c->apt_e_all.p, efix_all, sizeof(long long) * (size_t)c->apt_world * K * L, hipMemcpyHostToDevice, c->stream));
    if (c->apt_rank > 0) { int rc = rows_to_device(c, c->apt_recv.p, recv_lo, K); if (rc) return rc; }
    if (c->apt_rank + 1 < c->apt_world) { int rc = rows_to_device(c, c->apt_recv.p + (size_t)K * c->n_pad, recv_hi, K); if (rc) return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));          // the caller's buffers may go away after the call
    return apt_launch_swap(c, round, seed, n_pairs, out_pairs, out_accepted);
}

// Rehearsal of the neighbour exchange on ONE rank: the grouped ncclSend / ncclRecv pair of nlmc_apt_swap_collective with this rank
// as its own neighbour on both sides (a send to oneself inside a group is legal): what rank - 1 would receive from this rank's
// bottom-slot chains comes back as `from the upper neighbour`, and the other way round.
int nlmc_apt_selftest_exchange(nlmc_ctx *c, int8_t *out_recv_lo, int8_t *out_recv_hi)
{
    { int rc = apt_require(c, "nlmc_apt_selftest_exchange"); if (rc) return rc; }
    if (!c->comm) return fail(c, NLMC_ERR_STATE, "nlmc_apt_selftest_exchange: call nlmc_comm_init first");
    if (!out_recv_lo || !out_recv_hi) return fail(c, NLMC_ERR_ARG, "nlmc_apt_selftest_exchange: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    { int rc = apt_launch_pack(c); if (rc) return rc; }
    const int L = c->ladder_len, K = c->n_chains / L, me = c->comm_rank;
    const size_t rowbytes = (size_t)K * c->n_pad;
    int rc = g_rccl.GroupStart();
    if (rc == 0) rc = g_rccl.Send(c->apt_send.p, rowbytes, /*ncclInt8*/ 0, me, c->comm, c->stream);                 // my bottom-slot chains ...
    if (rc == 0) rc = g_rccl.Recv(c->apt_recv.p + rowbytes, rowbytes, 0, me, c->comm, c->stream);                   // ... arrive as the upper neighbour's
    if (rc == 0) rc = g_rccl.Send(c->apt_send.p + rowbytes, rowbytes, 0, me, c->comm, c->stream);                   // my top-slot chains ...
    if (rc == 0) rc = g_rccl.Recv(c->apt_recv.p, rowbytes, 0, me, c->comm, c->stream);                              // ... as the lower neighbour's
    const int rc2 = g_rccl.GroupEnd();
    if (rc != 0 || rc2 != 0) return fail(c, NLMC_ERR_HIP, "ncclSend / ncclRecv: " + rccl_err(rc != 0 ? rc : rc2));
    for (int side = 0; side < 2; ++side) {
        c->stage_out.resize(rowbytes);
        HIP_TRY(c, hipMemcpyAsync(c->stage_out.data(), c->apt_recv.p + (size_t)side * rowbytes, rowbytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        rows_to_host_finish(c, side ? out_recv_hi : out_recv_lo, K);
    }
    return nlmc_comm_check(c, 10000);
}

int nlmc_apt_swap_collective(nlmc_ctx *c, uint32_t round, uint64_t seed, int n_pairs, int32_t *out_pairs, uint8_t *out_accepted)
{
    { int rc = apt_require(c, "nlmc_apt_swap_collective"); if (rc) return rc; }
    { int rc = apt_check_pairs(c, n_pairs); if (rc) return rc; }
    if (c->apt_world > 1 && !c->comm) return fail(c, NLMC_ERR_STATE, "nlmc_apt_swap_collective: call nlmc_comm_init first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n_pairs == 0) return NLMC_OK;
    { int rc = apt_launch_pack(c); if (rc) return rc; }
    if (c->comm) {                          // (also with ONE rank: the in-place all-gather then rehearses the path on one GPU)
        const int L = c->ladder_len, K = c->n_chains / L, W = c->apt_world, me = c->apt_rank;
        const size_t blk = (size_t)K * L, rowbytes = (size_t)K * c->n_pad;
        // all on the stream the kernels run on: stream order is the only synchronisation.  The all-gather is in place (send = this
        // rank's block of the receive buffer); the two neighbour exchanges are one group (no deadlock whatever the order).
        int rc = g_rccl.AllGather(c->apt_e_all.p + (size_t)me * blk, c->apt_e_all.p, blk, /*ncclInt64*/ 4, c->comm, c->stream);
        if (rc != 0) return fail(c, NLMC_ERR_HIP, "ncclAllGather: " + rccl_err(rc));
        rc = g_rccl.GroupStart();
        if (rc == 0 && me > 0) rc = g_rccl.Send(c->apt_send.p, rowbytes, /*ncclInt8*/ 0, me - 1, c->comm, c->stream);
        if (rc == 0 && me > 0) rc = g_rccl.Recv(c->apt_recv.p, rowbytes, 0, me - 1, c->comm, c->stream);
        if (rc == 0 && me + 1 < W) rc = g_rccl.Send(c->apt_send.p + rowbytes, rowbytes, 0, me + 1, c->comm, c->stream);
        if (rc == 0 && me + 1 < W) rc = g_rccl.Recv(c->apt_recv.p + rowbytes, rowbytes, 0, me + 1, c->comm, c->stream);
        const int rc2 = g_rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) return fail(c, NLMC_ERR_HIP, "ncclSend / ncclRecv: " + rccl_err(rc != 0 ? rc : rc2));
    }
    return apt_launch_swap(c, round, seed, n_pairs, out_pairs, out_accepted);
}

// n_rounds rounds (sweeps_per_round sweeps at the ladder temperatures + the swap round) in ONE cooperative launch (k_rounds_fused).
int nlmc_pt_rounds_fused(nlmc_ctx *c, int precision, int n_rounds, int sweeps_per_round, uint32_t sweep0, uint32_t round0, uint64_t seed,
                         int n_pairs)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_rounds < 0 || sweeps_per_round < 1 || (precision != NLMC_F32 && precision != NLMC_F64) || n_pairs < 0)
        return fail(c, NLMC_ERR_ARG, "nlmc_pt_rounds_fused: bad argument");
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_rounds_fused: call nlmc_pt_init first");
    if (n_rounds == 0 || c->n_chains == 0) return NLMC_OK;
    const int L = c->ladder_len, T = sweeps_per_round;
    auto no = [&](const char *why) { return fail(c, NLMC_ERR_UNSUPPORTED, std::string("nlmc_pt_rounds_fused: ") + why); };
    if (getenv("NLMC_NO_PERSISTENT")) return no("switched off (NLMC_NO_PERSISTENT)");
    if (c->chain_base % L != 0 || c->n_chains % L != 0) return no("the context's block cuts a ladder (the swap needs other contexts' energies)");
    if (c->comm || (c->apt_R > 0 && c->apt_world > 1)) return no("the context takes part in a collective swap round");
    if (c->has_flags || c->subset != 0 || c->cur != c->stream) return no("phase flags or a chain subset are in force");
    if (n_pairs > std::max(0, L - 1)) return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    const int fslot = fused_plan_for(c, sweep0, n_rounds * T, seed);
    if (fslot < 0 || c->fz[fslot].T != T) return no("no fused-window plan of one window per round covers these sweeps");
    if (precision == NLMC_F64 && !fused_f64_supported(c, T)) return no("the fp64 mode does not run on fused windows for this instance");
    if (n_pairs > 0 && !(c->pt_plan_valid && c->pt_plan_seed == seed && c->pt_plan_npairs == n_pairs && round0 >= c->pt_plan_round0 &&
                         (uint64_t)round0 + (uint64_t)n_rounds <= (uint64_t)c->pt_plan_round0 + (uint64_t)c->pt_plan_rounds))
        return no("the pair selections of these rounds are not planned (nlmc_pt_plan)");
    HIP_TRY(c, hipSetDevice(c->device));
    const nlmc_ctx::FusedPlan &P = c->fz[fslot];
    const bool f64 = precision == NLMC_F64;
    const int kt = f64 ? 2 * c->xmax + 1 : 0;
    const FusedLds Lds = fused_lds(c->n, c->n_pad, false, false, P.fmt == NLMC_FMT_ADDR, kt);
#define NLMC_KR(D, F64_) {reinterpret_cast<const void *>(k_rounds_fused<D, NLMC_FMT_WIDE, F64_>), reinterpret_cast<const void *>(k_rounds_fused<D, NLMC_FMT_COMPACT, F64_>), \
                          reinterpret_cast<const void *>(k_rounds_fused<D, NLMC_FMT_ADDR, F64_>)}
    static const void *const table[2][2][3] = {{NLMC_KR(false, false), NLMC_KR(false, true)}, {NLMC_KR(true, false), NLMC_KR(true, true)}};
#undef NLMC_KR
    const void *kfun = table[c->has_diag][f64][P.fmt];
    { int rc = ensure_lds(c, 60 + ((c->has_diag ? 2 : 0) + (f64 ? 1 : 0)) * 3 + P.fmt, kfun, Lds.total); if (rc) return rc; }
    const int nt = fused_block(c->n);
    // every workgroup must be resident at once (they wait for each other): asked of the runtime, which also refuses the launch
    int per_cu = 0, n_cu = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfun, nt, Lds.total));
    HIP_TRY(c, hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
    if ((long long)per_cu * n_cu < c->n_chains) return no("more chains than workgroups the device holds at once");
    const int w0 = (int)((sweep0 - P.sweep0) / (uint32_t)T), nl = c->n_chains_global / L;
    const size_t G = (size_t)c->n_chains_global, PS = (size_t)P.pstride;
    HIP_TRY(c, c->rounds_ebuf.reserve(2 * G));
    HIP_TRY(c, c->rounds_bar.reserve(1));
    HIP_TRY(c, hipMemsetAsync(c->rounds_bar.p, 0, sizeof(unsigned), c->stream));
    if (!c->pt_tab_valid || c->pt_tab_temp_x != c->temp_x) {
        std::vector<double> tab((size_t)L * 2);
        for (int r = 0; r < L; ++r) { tab[2 * r] = -2.0 * LOG2E * c->beta_list[r]; tab[2 * r + 1] = -2.0 * LOG2E * (c->beta_list[r] / c->temp_x); }
        HIP_TRY(c, hipMemcpyAsync(c->pt_tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->pt_tab_valid = true; c->pt_tab_temp_x = c->temp_x;
    }
    SweepArgs a{};
    a.g = c->g; a.chain_base = c->chain_base; a.spins = c->spins.p; a.temp_x = c->temp_x;
    a.fz_pstride = P.pstride; a.fz_fmt = P.fmt;
    a.f_workers = P.workers; a.f_gen0 = P.gen0; a.f_gen_prio = c->knob_no_prio ? 0 : 1;
#ifdef NLMC_DEBUG_KNOBS
    a.dbg_flags = c->knob_dbg_flags;
#endif
    a.n_sweeps = T; a.sweep0 = sweep0; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    a.tab = c->pt_tab.p; a.tab_cs = 2; a.tab_ss = 0; a.slot_of_chain = c->slot_of_chain.p;
    a.efix = c->efix.p; a.escale = c->escale; a.eshift = c->escale - c->qs; a.qinv = std::ldexp(1.0f, -c->qs);
    a.trace_sweeps = T; a.rec_stride = 1; a.min_stride = 1; a.argmin = c->argmin.p;
    a.lds_neg_off = Lds.neg_off; a.lds_flags_off = Lds.flags_off; a.lds_u_off = Lds.u_off; a.lds_u_stride = Lds.u_bytes; a.lds_red_off = Lds.red_off;
    a.lds_snap_off = Lds.snap_off; a.lds_kt_off = Lds.kt_off; a.f64_xmax = c->xmax; a.f64_tie_mask = c->knob_tie_mask;
    a.qinv64 = std::ldexp(1.0, -c->qs);
    a.rng_stride = c->rng_stride; a.rng_base = c->rng_base; a.rng_ladder_len = std::max(1, L);
    RoundsArgs q{};
    q.n_rounds = n_rounds; q.n_windows_avail = c->knob_no_warm ? n_rounds : P.windows - w0;
    q.loff = P.loff.p + (size_t)w0 * (NLMC_LCAP + 1); q.nlev = P.nlev.p + w0; q.himax = P.himax.p + w0; q.send = P.send.p + (size_t)w0 * T;
    q.npos = P.npos.p + w0; q.head = P.head.p + (size_t)w0 * PS; q.ell = P.ell.p + (size_t)w0 * PS * NLMC_FZ_W;
    q.ladder_len = L; q.n_pairs = n_pairs; q.n_ladders = nl; q.round0 = round0;
    q.plan_pairs = n_pairs > 0 ? c->pt_plan_pairs.p + (size_t)(round0 - c->pt_plan_round0) * nl * n_pairs * 2 : nullptr;
    q.beta = c->pt_beta.p; q.slot_of_chain = c->slot_of_chain.p; q.chain_of_slot = c->chain_of_slot.p;
    q.ebuf = c->rounds_ebuf.p; q.bar = c->rounds_bar.p; q.status = c->pt_status.p;
    q.timeout_ticks = 100000000ll * 20;                   // 20 s of the 100 MHz wall clock
    if (c->pt_log_on && c->pt_log_npairs == n_pairs && n_pairs > 0 && round0 >= c->pt_log_round0 &&
        (uint64_t)round0 + (uint64_t)n_rounds <= (uint64_t)c->pt_log_round0 + (uint64_t)c->pt_log_rounds) {
        const size_t r = round0 - c->pt_log_round0;
        q.log_pairs = c->pt_log_pairs.p + r * (size_t)nl * n_pairs * 2;
        q.log_acc = c->pt_log_acc.p + r * (size_t)nl * n_pairs;
    }
    const bool timed = c->ev_accumulate;
    hipEvent_t e0 = nullptr, e2 = nullptr;
    if (timed) {
        e0 = next_event(c);
        hipEvent_t e1 = next_event(c);
        e2 = next_event(c);
        if (!e0 || !e1 || !e2) return fail(c, NLMC_ERR_HIP, "hipEventCreate failed");
        tag_triple(c, 1);
        HIP_TRY(c, hipEventRecord(e0, c->stream));
    }
    // the two argument structs travel through device memory (k_rounds_fused reads them through laundered constant-memory pointers);
    // two slots, so that a launch still running never sees the next one's arguments
    const size_t a_bytes = (sizeof(SweepArgs) + 255) & ~(size_t)255, slot_bytes = a_bytes + ((sizeof(RoundsArgs) + 255) & ~(size_t)255);
    HIP_TRY(c, c->rounds_args.reserve(2 * slot_bytes));
    const int as = c->rounds_args_slot ^= 1;
    std::vector<unsigned char> &hb = c->rounds_args_host[as];
    hb.assign(slot_bytes, 0);
    std::memcpy(hb.data(), &a, sizeof(SweepArgs));
    std::memcpy(hb.data() + a_bytes, &q, sizeof(RoundsArgs));
    unsigned char *dargs = c->rounds_args.p + (size_t)as * slot_bytes;
    HIP_TRY(c, hipMemcpyAsync(dargs, hb.data(), slot_bytes, hipMemcpyHostToDevice, c->stream));
    const SweepArgs *ap_dev = reinterpret_cast<const SweepArgs *>(dargs);
    const RoundsArgs *qp_dev = reinterpret_cast<const RoundsArgs *>(dargs + a_bytes);
    void *kargs[] = {&ap_dev, &qp_dev};
    HIP_TRY(c, hipLaunchCooperativeKernel(kfun, dim3(c->n_chains), dim3(nt), kargs, (unsigned)Lds.total, c->stream));
    HIP_TRY(c, hipGetLastError());
    if (timed) { HIP_TRY(c, hipEventRecord(e2, c->stream)); c->launches_timed += n_rounds; }
    c->launches_sweep += n_rounds;
    c->launches_total += n_rounds;
    c->stat_fused_window = w0 + n_rounds - 1;
    c->stat_fused_slot = fslot;
    c->sub_dirty = true;
    return NLMC_OK;
}

// n_rounds rounds as n_rounds sweep launches + ONE swap launch: launch i decides the swap of round i - 1 in its prologue
// (k_sweep_fused<.., DEFER>), the last round's swap is the ordinary k_pt_swap.
int nlmc_pt_rounds_deferred(nlmc_ctx *c, int precision, int n_rounds, int sweeps_per_round, uint32_t sweep0, uint32_t round0, uint64_t seed,
                            int n_pairs)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_rounds < 0 || sweeps_per_round < 1 || (precision != NLMC_F32 && precision != NLMC_F64) || n_pairs < 0)
        return fail(c, NLMC_ERR_ARG, "nlmc_pt_rounds_deferred: bad argument");
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_rounds_deferred: call nlmc_pt_init first");
    if (n_rounds == 0 || c->n_chains == 0) return NLMC_OK;
    const int L = c->ladder_len, T = sweeps_per_round;
    auto no = [&](const char *why) { return fail(c, NLMC_ERR_UNSUPPORTED, std::string("nlmc_pt_rounds_deferred: ") + why); };
    if (getenv("NLMC_NO_DEFERRED")) return no("switched off (NLMC_NO_DEFERRED)");
    if (c->chain_base % L != 0 || c->n_chains % L != 0) return no("the context's block cuts a ladder (the swap needs other contexts' energies)");
    if (c->comm || (c->apt_R > 0 && c->apt_world > 1)) return no("the context takes part in a collective swap round");
    if (c->has_flags || c->subset != 0 || c->cur != c->stream || c->track_min) return no("phase flags, a chain subset or a tracked minimum are in force");
    if (n_pairs < 1 || n_pairs > std::max(0, L - 1)) return n_pairs < 1 ? no("no swap pairs") : fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    const int fslot = fused_plan_for(c, sweep0, n_rounds * T, seed);
    if (fslot < 0 || c->fz[fslot].T != T) return no("no fused-window plan of one window per round covers these sweeps");
    if (precision == NLMC_F64 && !fused_f64_supported(c, T)) return no("the fp64 mode does not run on fused windows for this instance");
    if (!(c->pt_plan_valid && c->pt_plan_seed == seed && c->pt_plan_npairs == n_pairs && round0 >= c->pt_plan_round0 &&
          (uint64_t)round0 + (uint64_t)n_rounds <= (uint64_t)c->pt_plan_round0 + (uint64_t)c->pt_plan_rounds))
        return no("the pair selections of these rounds are not planned (nlmc_pt_plan)");
    HIP_TRY(c, hipSetDevice(c->device));
    const nlmc_ctx::FusedPlan &P = c->fz[fslot];
    const int w0 = (int)((sweep0 - P.sweep0) / (uint32_t)T), nl = c->n_chains_global / L;
    const size_t G = (size_t)c->n_chains_global;
    HIP_TRY(c, c->rounds_ebuf.reserve(2 * G));
    if (!c->pt_tab_valid || c->pt_tab_temp_x != c->temp_x) {
        std::vector<double> tab((size_t)L * 2);
        for (int r = 0; r < L; ++r) { tab[2 * r] = -2.0 * LOG2E * c->beta_list[r]; tab[2 * r + 1] = -2.0 * LOG2E * (c->beta_list[r] / c->temp_x); }
        HIP_TRY(c, hipMemcpyAsync(c->pt_tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->pt_tab_valid = true; c->pt_tab_temp_x = c->temp_x;
    }
    if (!c->ev_accumulate) c->ev_used = 0;
    c->ev_call_start = c->ev_used;
    c->launches_sweep = 0;
    const bool log = c->pt_log_on && c->pt_log_npairs == n_pairs && round0 >= c->pt_log_round0 &&
                     (uint64_t)round0 + (uint64_t)n_rounds <= (uint64_t)c->pt_log_round0 + (uint64_t)c->pt_log_rounds;
    for (int r = 0; r < n_rounds; ++r) {
        DeferSwap d{};
        d.ladder_len = L; d.n_ladders = nl; d.beta = c->pt_beta.p;
        d.slot_of_chain = c->slot_of_chain.p; d.chain_of_slot = c->chain_of_slot.p;
        if (r > 0) {                          // the swap of round round0 + r - 1, on the energies the previous launch published
            const uint32_t rr = round0 + (uint32_t)(r - 1);
            d.n_pairs = n_pairs; d.round = rr;
            d.plan_pairs = c->pt_plan_pairs.p + (size_t)(rr - c->pt_plan_round0) * nl * n_pairs * 2;
            d.e_prev = c->rounds_ebuf.p + (size_t)((r - 1) & 1) * G;
            if (log) {
                const size_t lr = rr - c->pt_log_round0;
                d.log_pairs = c->pt_log_pairs.p + lr * (size_t)nl * n_pairs * 2;
                d.log_acc = c->pt_log_acc.p + lr * (size_t)nl * n_pairs;
            }
        }
        // (this launch publishes its chains' final energies for the next one: rows of the local chains inside the global vector)
        double *sink = c->rounds_ebuf.p + (size_t)(r & 1) * G + c->chain_base;
        int rc = run_fused(c, fslot, w0 + r, sweep0 + (uint32_t)(r * T), seed, c->pt_tab.p, 2, 0, true, false, false, false, false, 0, 0, T,
                           precision == NLMC_F64, &d, sink);
        if (rc) return rc;
    }
    c->sub_dirty = true;
    // the last round's swap: the ordinary kernel on the tracked energies
    return nlmc_pt_swap_philox(c, round0 + (uint32_t)(n_rounds - 1), seed, n_pairs, nullptr, nullptr, nullptr);
}

int nlmc_pt_log_begin(nlmc_ctx *c, uint32_t round0, int n_rounds, int n_pairs)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_log_begin: call nlmc_pt_init first");
    if (n_rounds < 0 || n_pairs < 0) return fail(c, NLMC_ERR_ARG, "nlmc_pt_log_begin: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nl = (size_t)(c->n_chains_global / c->ladder_len), tot = (size_t)n_rounds * nl * (size_t)n_pairs;
    c->pt_log_on = false;
    if (tot == 0) return NLMC_OK;
    HIP_TRY(c, c->pt_log_pairs.reserve(tot * 2));
    HIP_TRY(c, c->pt_log_acc.reserve(tot));
    HIP_TRY(c, hipMemsetAsync(c->pt_log_pairs.p, 0xFF, sizeof(int32_t) * tot * 2, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->pt_log_acc.p, 0, tot, c->stream));
    c->pt_log_on = true;
    c->pt_log_round0 = round0; c->pt_log_rounds = n_rounds; c->pt_log_npairs = n_pairs;
    return NLMC_OK;
}

int nlmc_pt_log_read(nlmc_ctx *c, int32_t *out_pairs, uint8_t *out_accepted)
{
    if (!c || !out_pairs || !out_accepted) return fail(c, NLMC_ERR_ARG, "nlmc_pt_log_read: NULL argument");
    if (!c->pt_log_on) return fail(c, NLMC_ERR_STATE, "nlmc_pt_log_read: no log was begun");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t nl = (size_t)(c->n_chains_global / c->ladder_len), tot = (size_t)c->pt_log_rounds * nl * (size_t)c->pt_log_npairs;
    HIP_TRY(c, hipMemcpyAsync(out_pairs, c->pt_log_pairs.p, sizeof(int32_t) * tot * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out_accepted, c->pt_log_acc.p, tot, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return nlmc_pt_check(c);
}

int nlmc_pt_check(nlmc_ctx *c)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return NLMC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int32_t st = 0;
    HIP_TRY(c, hipMemcpyAsync(&st, c->pt_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (st != 0) {
        HIP_TRY(c, hipMemsetAsync(c->pt_status.p, 0, sizeof(int32_t), c->stream));
        if (st == 3) return fail(c, NLMC_ERR_HIP, "nlmc_pt_rounds_fused: the workgroups of a launch lost each other (grid wait timed out); the chains' states are not valid");
        return fail(c, NLMC_ERR_ARG, "Cannot find non-overlapping pairs.");
    }
    return NLMC_OK;
}

// ---------------------------------------------------------------------------------------------------
// iso-cluster move
// ---------------------------------------------------------------------------------------------------
// 16-bit neighbour table shared by k_levelize_fused and k_icm_components (built once per context)
static int ensure_adjacency(nlmc_ctx *c)
{
    if (c->fz_adj_ready) return NLMC_OK;
    HIP_TRY(c, c->fz_adj.reserve((size_t)c->n * NLMC_FZ_ADJ));
    hipLaunchKernelGGL(k_fused_adjacency, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, c->n, c->rowptr.p, c->col.p, c->fz_adj.p);
    HIP_TRY(c, hipGetLastError());
    c->fz_adj_ready = true;
    return NLMC_OK;
}

static int icm_launch_components(nlmc_ctx *c, const int32_t *pairs_dev, int n_pairs)
{
    HIP_TRY(c, c->icm_label.reserve((size_t)n_pairs * c->n));
    HIP_TRY(c, c->icm_info.reserve((size_t)n_pairs * 2));
    IcmArgs a{};
    a.g = c->g; a.spins = c->spins.p; a.pairs = pairs_dev; a.label = c->icm_label.p; a.info = c->icm_info.p;
    if (c->big) {                   // the forest lives in the labels themselves (csrc/nlmc_big.h)
        hipLaunchKernelGGL(k_icm_components_big, dim3(n_pairs), dim3(1024), 0, c->stream, a);
        HIP_TRY(c, hipGetLastError());
        return NLMC_OK;
    }
    if (!c->has_zero_vals && c->n <= 65535) {
        int rc = ensure_adjacency(c);
        if (rc) return rc;
        a.adj = reinterpret_cast<const uint4 *>(c->fz_adj.p);
    }
    const size_t lds = (size_t)c->n * 4 + (((size_t)c->n + 1) & ~(size_t)1) * 2 + 16;
    { int rc = ensure_lds(c, 1, reinterpret_cast<const void *>(k_icm_components), lds); if (rc) return rc; }
    hipLaunchKernelGGL(k_icm_components, dim3(n_pairs), dim3(c->n >= 4096 ? 1024 : 256), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

// components + pick + move + incremental energies of a batch of pairs in one launch (k_icm_round)
static int icm_launch_round(nlmc_ctx *c, const int32_t *pairs_dev, int n_pairs, uint32_t round, uint64_t seed, int katz,
                            int pair_R = 0, int pair_K = 0)
{
    HIP_TRY(c, c->icm_info.reserve((size_t)n_pairs * 2));
    IcmRoundArgs a{};
    a.g = c->g; a.spins = c->spins.p; a.pairs = pairs_dev; a.info = c->icm_info.p;
    a.pair_R = pair_R; a.pair_K = pair_K; a.chain_of_slot = c->chain_of_slot.p;
    a.slot0 = c->rng_base; a.rng_stride = c->rng_stride; a.rng_base = c->rng_base;
    if (!c->has_zero_vals && c->n <= 65535) {
        int rc = ensure_adjacency(c);
        if (rc) return rc;
        a.adj = reinterpret_cast<const uint4 *>(c->fz_adj.p);
    }
    a.round = round; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.katz = katz; a.chain_base = c->chain_base;
    a.efix = c->efix.p; a.energy_sink = c->energy_sink; a.eshift = c->escale - c->qs; a.escale = c->escale;
    if (c->big) {
        if (!pairs_dev && pair_K > 1024) return fail(c, NLMC_ERR_UNSUPPORTED, "icm: more sub-replicas than threads of a workgroup");
        HIP_TRY(c, c->icm_label.reserve((size_t)n_pairs * c->n));
        a.adj = nullptr;
        hipLaunchKernelGGL(k_icm_round_big, dim3(n_pairs), dim3(1024), 0, c->stream, a, c->icm_label.p);
        HIP_TRY(c, hipGetLastError());
        return NLMC_OK;
    }
    size_t cur = (size_t)c->n * 4;
    a.lds_cand_off = (int)cur; cur += (((size_t)c->n + 7) & ~(size_t)7) * 2;
    cur = (cur + 15) & ~(size_t)15;
    a.lds_sa_off = (int)cur; cur += (size_t)c->n_pad;
    a.lds_sb_off = (int)cur; cur += (size_t)c->n_pad;
    const size_t lds = cur + 16;
    { int rc = ensure_lds(c, 21, reinterpret_cast<const void *>(k_icm_round), lds); if (rc) return rc; }
    hipLaunchKernelGGL(k_icm_round, dim3(n_pairs), dim3(c->n >= 4096 ? 1024 : 256), lds, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

static int icm_check_pair(nlmc_ctx *c, int a, int b)
{
    if (a < 0 || b < 0 || a >= c->n_chains || b >= c->n_chains || a == b) return fail(c, NLMC_ERR_ARG, "icm: bad chain pair");
    return NLMC_OK;
}

int nlmc_icm_components(nlmc_ctx *c, int chain_a, int chain_b, int32_t *out_n_components)
{
    if (!c || !out_n_components) return fail(c, NLMC_ERR_ARG, "nlmc_icm_components: NULL argument");
    int rc = icm_check_pair(c, chain_a, chain_b);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->icm_pairs.reserve(2));
    const int32_t pr[2] = {chain_a, chain_b};
    HIP_TRY(c, hipMemcpyAsync(c->icm_pairs.p, pr, sizeof(pr), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = icm_launch_components(c, c->icm_pairs.p, 1);
    if (rc) return rc;
    int32_t info[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(info, c->icm_info.p, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (info[0] < 0) return fail(c, NLMC_ERR_STATE, "icm: the component search did not converge");
    *out_n_components = info[0];
    return NLMC_OK;
}

int nlmc_icm_get_labels(nlmc_ctx *c, int32_t *out)
{
    if (!c || !out) return fail(c, NLMC_ERR_ARG, "nlmc_icm_get_labels: NULL argument");
    if (c->icm_label.cap < (size_t)c->n) return fail(c, NLMC_ERR_STATE, "nlmc_icm_get_labels: no component search has run yet");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->icm_label.p, sizeof(int32_t) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->n; ++k) if (out[k] == INT_MAX) out[k] = -1;
    return NLMC_OK;
}

static int icm_apply(nlmc_ctx *c, int n_pairs, const int64_t *pick_dev_or_null, int64_t pick_host, uint32_t round,
                     uint64_t seed, int katz, int philox)
{
    IcmMoveArgs m{};
    m.g = c->g; m.spins = c->spins.p; m.pairs = c->icm_pairs.p; m.label = c->icm_label.p; m.info = c->icm_info.p;
    m.pick_host = pick_host; m.use_philox = philox; m.round = round;
    m.seed_lo = (uint32_t)seed; m.seed_hi = (uint32_t)(seed >> 32); m.katz = katz; m.chain_base = c->chain_base;
    (void)pick_dev_or_null;
    hipLaunchKernelGGL(k_icm_move, dim3(n_pairs), dim3(c->n >= 4096 ? 1024 : 256), 0, c->stream, m);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int nlmc_icm_move(nlmc_ctx *c, int chain_a, int chain_b, int64_t pick_index, int katzgraber, int32_t *out_info)
{
    if (!c) return NLMC_ERR_ARG;
    int rc = icm_check_pair(c, chain_a, chain_b);
    if (rc) return rc;
    if (pick_index < 0) return fail(c, NLMC_ERR_ARG, "nlmc_icm_move: pick_index < 0");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->icm_pairs.reserve(2));
    const int32_t pr[2] = {chain_a, chain_b};
    HIP_TRY(c, hipMemcpyAsync(c->icm_pairs.p, pr, sizeof(pr), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = icm_launch_components(c, c->icm_pairs.p, 1);
    if (rc) return rc;
    rc = icm_apply(c, 1, nullptr, pick_index, 0, 0, katzgraber, 0);
    if (rc) return rc;
    rc = launch_energy_self(c, nullptr);   // states changed non-incrementally: resync tracked energies
    if (rc) return rc;
    int32_t info[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(info, c->icm_info.p, sizeof(info), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (info[0] < 0) return fail(c, NLMC_ERR_STATE, "icm: the component search did not converge");
    if (out_info) { out_info[0] = info[0]; out_info[1] = info[1]; }
    return NLMC_OK;
}

int nlmc_icm_round_philox(nlmc_ctx *c, const int32_t *pairs, int n_pairs, uint32_t round, uint64_t seed, int katzgraber,
                          int32_t *out_info)
{
    if (!c || (n_pairs > 0 && !pairs) || n_pairs < 0) return fail(c, NLMC_ERR_ARG, "nlmc_icm_round_philox: bad argument");
    if (n_pairs == 0) return NLMC_OK;
    std::vector<uint8_t> used((size_t)c->n_chains, 0);
    for (int p = 0; p < n_pairs; ++p) {
        int rc = icm_check_pair(c, pairs[2 * p], pairs[2 * p + 1]);
        if (rc) return rc;
        if (used[pairs[2 * p]] || used[pairs[2 * p + 1]]) return fail(c, NLMC_ERR_ARG, "nlmc_icm_round_philox: a chain appears in two pairs");
        used[pairs[2 * p]] = used[pairs[2 * p + 1]] = 1;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->icm_pairs.reserve((size_t)n_pairs * 2));
    HIP_TRY(c, hipMemcpyAsync(c->icm_pairs.p, pairs, sizeof(int32_t) * 2 * n_pairs, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the tracked energies must be the energies of the current states in the fixed-point model: true after sweeps and
    // moves of this kind; a caller that overwrote the states gets them re-synchronised by nlmc_set_spins / nlmc_energy
    int rc = icm_launch_round(c, c->icm_pairs.p, n_pairs, round, seed, katzgraber);
    if (rc) return rc;
    if (out_info) {
        HIP_TRY(c, hipMemcpyAsync(out_info, c->icm_info.p, sizeof(int32_t) * 2 * n_pairs, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int p = 0; p < n_pairs; ++p)
            if (out_info[2 * p] < 0) return fail(c, NLMC_ERR_STATE, "icm: the component search did not converge");
    }
    return NLMC_OK;
}

int nlmc_icm_round_ladders(nlmc_ctx *c, uint32_t round, uint64_t seed, int katzgraber, int32_t *out_n_pairs,
                           int32_t *out_info)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_icm_round_ladders: nlmc_pt_init first");
    if (c->chain_base != 0 || c->n_chains != c->n_chains_global)
        return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_icm_round_ladders: the sub-replicas of a temperature must live in one context");
    const int R = c->ladder_len, K = c->n_chains_global / R, n_pairs = R * (K / 2);
    if (out_n_pairs) *out_n_pairs = n_pairs;
    if (n_pairs == 0) return NLMC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int nt_round = c->n >= 4096 ? 1024 : 256;
    const bool pair_in_kernel = K <= nt_round;          // the pairing (a sort of K Philox keys per slot) is made by the move kernel itself
    if (!pair_in_kernel && c->rng_stride) return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_icm_round_ladders: more sub-replicas than threads with slot-keyed random numbers");
    if (!pair_in_kernel) {
        HIP_TRY(c, c->icm_pairs.reserve((size_t)n_pairs * 2));
        hipLaunchKernelGGL(k_icm_pair_ladders, dim3((R * K + 63) / 64), dim3(64), 0, c->stream, R, K, round, (uint32_t)seed,
                           (uint32_t)(seed >> 32), c->chain_of_slot.p, c->icm_pairs.p);
        HIP_TRY(c, hipGetLastError());
    }
    // one launch: components, pick, move, incremental fixed-point energies (k_icm_round; round 2's separate row walk of the
    // cluster ran one thread per cluster spin from a cold kernel: 41 us -- inside the components workgroup, with both
    // states in LDS, it is a few us)
    int rc = icm_launch_round(c, pair_in_kernel ? nullptr : c->icm_pairs.p, n_pairs, round, seed, katzgraber, R, K);
    if (rc) return rc;
    if (out_info) {
        HIP_TRY(c, hipMemcpyAsync(out_info, c->icm_info.p, sizeof(int32_t) * 2 * n_pairs, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int p = 0; p < n_pairs; ++p)
            if (out_info[2 * p] < 0) return fail(c, NLMC_ERR_STATE, "icm: the component search did not converge");
    }
    return NLMC_OK;
}

// ------------------------------------------------------------------------------------------------------
// convexified loopy belief propagation (backbone inference), batched over problems
// ------------------------------------------------------------------------------------------------------
// graph, buffers, constants of a batch of n_problems inferences; seeds (lbp_ms) are the caller's business
static int lbp_setup(nlmc_ctx *c, int n_problems, const double *epsilon, const double *lambdas, int n_lambdas, double beta,
                     bool want_all)
{
    if (c->nnz > (int64_t)INT_MAX / 2) return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_lbp_convexified: nnz too large");
    const int n = c->n, nnz = (int)c->nnz;
    const size_t P = (size_t)n_problems, E = (size_t)std::max(nnz, 1);
    if (!c->lbp_graph_ready) {
        HIP_TRY(c, c->lbp_src.reserve(E));
        HIP_TRY(c, c->lbp_rev.reserve(E));
        HIP_TRY(c, c->lbp_flag.reserve(1));
        HIP_TRY(c, hipMemsetAsync(c->lbp_flag.p, 0, sizeof(int32_t), c->cur));
        hipLaunchKernelGGL(k_lbp_src, dim3((n + 255) / 256), dim3(256), 0, c->cur, n, c->rowptr.p, c->lbp_src.p);
        HIP_TRY(c, hipGetLastError());
        if (nnz > 0) {
            hipLaunchKernelGGL(k_lbp_rev, dim3((nnz + 255) / 256), dim3(256), 0, c->cur, nnz, c->rowptr.p, c->col.p,
                               c->lbp_src.p, c->lbp_rev.p, c->lbp_flag.p);
            HIP_TRY(c, hipGetLastError());
        }
        int32_t flag = 0;
        HIP_TRY(c, hipMemcpyAsync(&flag, c->lbp_flag.p, sizeof(flag), hipMemcpyDeviceToHost, c->cur));
        HIP_TRY(c, hipStreamSynchronize(c->cur));
        if (flag) return fail(c, NLMC_ERR_ARG, "LBP needs a structurally symmetric J");
        c->lbp_graph_ready = true;
    }
    HIP_TRY(c, c->lbp_tJ.reserve(E));
    HIP_TRY(c, c->lbp_eps.reserve((size_t)n));
    HIP_TRY(c, c->lbp_ms.reserve(P * n));
    HIP_TRY(c, c->lbp_lams.reserve((size_t)n_lambdas));
    HIP_TRY(c, c->lbp_w0.reserve(P * E));
    HIP_TRY(c, c->lbp_w1.reserve(P * E));
    HIP_TRY(c, c->lbp_hm.reserve(P * E));
    HIP_TRY(c, c->lbp_tot.reserve(P * n));
    HIP_TRY(c, c->lbp_mag.reserve(P * n));
    HIP_TRY(c, c->lbp_out_i.reserve(P * (2 + (size_t)n_lambdas)));
    if (want_all) HIP_TRY(c, c->lbp_mag_all.reserve(P * (size_t)n_lambdas * n));
    // constants of the run: uploaded when they change only (a replica-exchange run infers backbones once per round with the
    // same epsilon, lambda list and beta: no host synchronisation on that path)
    const std::vector<double> eps_h(epsilon, epsilon + n), lam_h(lambdas, lambdas + n_lambdas);
    if (eps_h != c->lbp_eps_host || lam_h != c->lbp_lams_host) {
        HIP_TRY(c, hipStreamSynchronize(c->cur));
        HIP_TRY(c, hipMemcpyAsync(c->lbp_eps.p, eps_h.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->cur));
        HIP_TRY(c, hipMemcpyAsync(c->lbp_lams.p, lam_h.data(), sizeof(double) * n_lambdas, hipMemcpyHostToDevice, c->cur));
        HIP_TRY(c, hipStreamSynchronize(c->cur));
        c->lbp_eps_host = eps_h;
        c->lbp_lams_host = lam_h;
    }
    if (nnz > 0 && (!c->lbp_tJ_valid || c->lbp_tJ_beta != beta)) {
        hipLaunchKernelGGL(k_lbp_tanhJ, dim3((nnz + 255) / 256), dim3(256), 0, c->cur, nnz, c->val64.p, beta, c->lbp_tJ.p);
        HIP_TRY(c, hipGetLastError());
        c->lbp_tJ_valid = true;
        c->lbp_tJ_beta = beta;
    }
    return NLMC_OK;
}

// the batched lambda loop on the seeds in lbp_ms (stream-ordered, no synchronisation)
static int lbp_launch(nlmc_ctx *c, int n_problems, int n_lambdas, double beta, double tolerance, int max_iterations, double sat,
                      bool want_all, bool single_workgroup = false)
{
    const int n = c->n, nnz = (int)c->nnz;
    const size_t P = (size_t)n_problems;
    HIP_TRY(c, hipMemsetAsync(c->lbp_mag.p, 0, sizeof(double) * P * n, c->cur));
    HIP_TRY(c, hipMemsetAsync(c->lbp_out_i.p, 0, sizeof(int32_t) * P * (2 + (size_t)n_lambdas), c->cur));
    LbpArgs a{};
    a.n = n; a.nnz = nnz; a.n_lams = n_lambdas; a.max_iter = max_iterations;
    a.rowptr = c->rowptr.p; a.col = c->col.p; a.src = c->lbp_src.p; a.rev = c->lbp_rev.p;
    a.val = c->val64.p; a.tJ = c->lbp_tJ.p; a.h = c->h64.p; a.eps = c->lbp_eps.p; a.m_star = c->lbp_ms.p; a.lams = c->lbp_lams.p;
    a.beta = beta; a.inv_beta = 1.0 / beta; a.tol = tolerance; a.sat = sat; a.usat = std::atanh(sat) / beta;
    a.w0 = c->lbp_w0.p; a.w1 = c->lbp_w1.p; a.hm = c->lbp_hm.p; a.tot = c->lbp_tot.p; a.mag = c->lbp_mag.p;
    a.mag_all = want_all ? c->lbp_mag_all.p : nullptr;
    a.out_nlam = c->lbp_out_i.p; a.out_status = c->lbp_out_i.p + P; a.out_iters = c->lbp_out_i.p + 2 * P;
    // Workgroups per problem (a problem is bound by the fp64 VALU of the CUs it runs on): up to 8 when FEW problems
    // are in flight and each has work for them.  The group barrier costs two agent-scope fences per iteration, and L2
    // write-backs from many workgroups at once are expensive -- measured: 1 problem of 10^4 spins 23 -> 6.6 ms with 8
    // workgroups, 4 problems 23 -> 8.9 ms, but 64 problems 41 -> 67 ms with 4 each (and 8x slower at 10^3 spins): so
    // at most 32 workgroups take part in barriers, every launch stays resident (<= one workgroup per CU) in any case.
    int cus = 0;
    HIP_TRY(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    int group = 1;
    while (group < 8 && (long long)n_problems * group * 2 <= std::min(32, cus) && n / (group * 2) >= 1024) group *= 2;
    if (const char *e = getenv("NLMC_LBP_GROUP")) { const int v = atoi(e); if (v >= 1 && v <= 8 && (long long)n_problems * v <= cus) group = v; }
    if (single_workgroup) group = 1;
    a.group = group;
    if (const char *e = getenv("NLMC_LBP_POLL_BUDGET")) a.poll_budget = atoi(e);      // (test knob: provoke the timeout path)
    HIP_TRY(c, c->lbp_bar.reserve(P));
    HIP_TRY(c, c->lbp_part.reserve(P * 2 * (size_t)group * 4));
    HIP_TRY(c, hipMemsetAsync(c->lbp_bar.p, 0, sizeof(unsigned int) * P, c->cur));
    a.bar = c->lbp_bar.p; a.part = c->lbp_part.p;
    // small instances: messages in LDS, a thread's edges in registers (k_lbp_lds; same bits): 8 waves x 12 edges per thread, two
    // messages in lock step (other shapes: see the kernel's header)
    if (group == 1 && n <= 2048 && nnz <= 6144 && !getenv("NLMC_LBP_GLOBAL")) {
        const size_t lds_small = ((size_t)2 * (6144 + 1) + 2048 + 1 + 64) * sizeof(double);
        const void *kf = c->has_diag ? reinterpret_cast<const void *>(k_lbp_lds<512, 12, 4, true, 2>)
                                     : reinterpret_cast<const void *>(k_lbp_lds<512, 12, 4, false, 2>);
        { int rc = ensure_lds(c, 17 + (c->has_diag ? 1 : 0), kf, lds_small); if (rc) return rc; }
        void *kargs[] = {&a};
        HIP_TRY(c, hipLaunchKernel(kf, dim3(n_problems), dim3(512), kargs, lds_small, c->cur));
        return NLMC_OK;
    }
    hipLaunchKernelGGL(k_lbp, dim3(n_problems * group), dim3(NLMC_LBP_THREADS), 0, c->cur, a);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int nlmc_lbp_convexified(nlmc_ctx *c, int n_problems, const double *m_star, const double *epsilon, const double *lambdas,
                         int n_lambdas, double beta, double tolerance, int max_iterations, double sat,
                         double *out_mag, double *out_mag_all, int32_t *out_n_lambdas, int32_t *out_iters,
                         int32_t *out_status)
{
    if (!c) return NLMC_ERR_ARG;
    if (n_problems < 1 || n_lambdas < 1 || !m_star || !epsilon || !lambdas || !out_mag || !out_n_lambdas || !out_iters ||
        !out_status || max_iterations < 0 || !(beta != 0.0))
        return fail(c, NLMC_ERR_ARG, "nlmc_lbp_convexified: bad sizes or NULL arrays");
    HIP_TRY(c, hipSetDevice(c->device));
    const int n = c->n;
    const size_t P = (size_t)n_problems;
    { int rc = lbp_setup(c, n_problems, epsilon, lambdas, n_lambdas, beta, out_mag_all != nullptr); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(c->lbp_ms.p, m_star, sizeof(double) * P * n, hipMemcpyHostToDevice, c->cur));
    { int rc = lbp_launch(c, n_problems, n_lambdas, beta, tolerance, max_iterations, sat, out_mag_all != nullptr); if (rc) return rc; }
    std::vector<int32_t> oi(P * (2 + (size_t)n_lambdas));
    HIP_TRY(c, hipMemcpyAsync(oi.data(), c->lbp_out_i.p, sizeof(int32_t) * oi.size(), hipMemcpyDeviceToHost, c->cur));
    HIP_TRY(c, hipMemcpyAsync(out_mag, c->lbp_mag.p, sizeof(double) * P * n, hipMemcpyDeviceToHost, c->cur));
    if (out_mag_all)
        HIP_TRY(c, hipMemcpyAsync(out_mag_all, c->lbp_mag_all.p, sizeof(double) * P * (size_t)n_lambdas * n, hipMemcpyDeviceToHost, c->cur));
    HIP_TRY(c, hipStreamSynchronize(c->cur));
    bool lost = false;
    for (size_t q = 0; q < P; ++q) lost = lost || oi[P + q] == 2;
    if (lost) {
        // The workgroups that share a problem poll each other with a bounded budget; they only find each other when all of
        // them are resident, which another context on the same GPU can prevent.  Once more with ONE workgroup per
        // problem -- slower, no barrier between workgroups, same bits -- before giving up.
        HIP_TRY(c, hipMemcpyAsync(c->lbp_ms.p, m_star, sizeof(double) * P * n, hipMemcpyHostToDevice, c->cur));
        { int rc = lbp_launch(c, n_problems, n_lambdas, beta, tolerance, max_iterations, sat, out_mag_all != nullptr, true); if (rc) return rc; }
        HIP_TRY(c, hipMemcpyAsync(oi.data(), c->lbp_out_i.p, sizeof(int32_t) * oi.size(), hipMemcpyDeviceToHost, c->cur));
        HIP_TRY(c, hipMemcpyAsync(out_mag, c->lbp_mag.p, sizeof(double) * P * n, hipMemcpyDeviceToHost, c->cur));
        if (out_mag_all)
            HIP_TRY(c, hipMemcpyAsync(out_mag_all, c->lbp_mag_all.p, sizeof(double) * P * (size_t)n_lambdas * n, hipMemcpyDeviceToHost, c->cur));
        HIP_TRY(c, hipStreamSynchronize(c->cur));
    }
    for (size_t q = 0; q < P; ++q)
        if (oi[P + q] == 2) return fail(c, NLMC_ERR_HIP, "nlmc_lbp_convexified: the workgroups of a problem lost each other (group barrier timed out)");
    std::memcpy(out_n_lambdas, oi.data(), sizeof(int32_t) * P);
    std::memcpy(out_status, oi.data() + P, sizeof(int32_t) * P);
    std::memcpy(out_iters, oi.data() + 2 * P, sizeof(int32_t) * P * (size_t)n_lambdas);
    return NLMC_OK;
}

// ------------------------------------------------------------------------------------------------------
// replica-exchange rounds whose marked temperature slots run NMC cycles: chain subsets, device-side hand-offs
// ------------------------------------------------------------------------------------------------------
int nlmc_pt_mark_slots(nlmc_ctx *c, const uint8_t *marks)
{
    if (!c) return NLMC_ERR_ARG;
    if (c->ladder_len == 0) return fail(c, NLMC_ERR_STATE, "nlmc_pt_mark_slots: call nlmc_pt_init first");
    const int L = c->ladder_len;
    if (c->chain_base % L != 0 || c->n_chains % L != 0)
        return fail(c, NLMC_ERR_UNSUPPORTED, "nlmc_pt_mark_slots: every ladder must lie inside one context");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<uint8_t> m((size_t)L, 0);
    int k = 0;
    if (marks) for (int r = 0; r < L; ++r) { m[(size_t)r] = marks[r] ? 1 : 0; k += m[(size_t)r]; }
    HIP_TRY(c, c->slot_mark.reserve((size_t)L));
    HIP_TRY(c, c->sub_list_buf.reserve((size_t)std::max(c->n_chains, 1)));
    HIP_TRY(c, c->cmask.reserve((size_t)std::max(c->n_chains, 1) * c->n_pad));
    HIP_TRY(c, c->nmc_status.reserve(1));
    HIP_TRY(c, hipMemcpyAsync(c->slot_mark.p, m.data(), (size_t)L, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->nmc_status.p, 0, sizeof(int32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->cmask.p, 0, (size_t)std::max(c->n_chains, 1) * c->n_pad, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));          // (`m` goes away)
    c->n_marked_local = k * (c->n_chains / L);
    c->subset = 0;
    c->sub_dirty = true;
    return NLMC_OK;
}

int nlmc_select_chains(nlmc_ctx *c, int which)
{
    if (!c) return NLMC_ERR_ARG;
    if (which < NLMC_CHAINS_ALL || which > NLMC_CHAINS_MARKED) return fail(c, NLMC_ERR_ARG, "nlmc_select_chains: bad selector");
    if (which != NLMC_CHAINS_ALL && !c->slot_mark.p) return fail(c, NLMC_ERR_STATE, "nlmc_select_chains: call nlmc_pt_mark_slots first");
    HIP_TRY(c, hipSetDevice(c->device));
    // the subset-aware launches of the marked chains go to the second stream when overlap is on: forked from the main stream
    // where the first subset of a round is selected (after the chain lists are up to date, before any subset work is
    // queued), joined again where all chains are selected
    if (which != NLMC_CHAINS_ALL) {
        c->subset = which;
        if (c->cur == c->stream) { int rc = ensure_subset(c); if (rc) return rc; }
        if (c->overlap && !c->forked) {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            c->forked = true;
        }
        if (c->overlap && which == NLMC_CHAINS_MARKED && c->cur == c->stream) {
            HIP_TRY(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
            c->cur = c->aux;
        } else if (which == NLMC_CHAINS_UNMARKED && c->cur != c->stream) {
            return fail(c, NLMC_ERR_STATE, "nlmc_select_chains: select all chains between the marked and the unmarked subset of a round");
        }
        return NLMC_OK;
    }
    if (c->cur != c->stream) {
        HIP_TRY(c, hipEventRecord(c->ev_join, c->aux));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->cur = c->stream;
    }
    c->forked = false;
    c->subset = which;
    return NLMC_OK;
}

int nlmc_own_stream(nlmc_ctx *c)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->owns_stream) return NLMC_OK;
    if (c->cur != c->stream) return fail(c, NLMC_ERR_STATE, "nlmc_own_stream: select all chains first");
    if (c->comm) return fail(c, NLMC_ERR_STATE, "nlmc_own_stream: call it before nlmc_comm_init");
    // whatever was queued on the stream given at creation is finished first: from here on the two streams are unrelated
    if (c->stream) HIP_TRY(c, hipStreamSynchronize(c->stream)); else HIP_TRY(c, hipDeviceSynchronize());
    hipStream_t s = nullptr;
    HIP_TRY(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    c->stream = s;
    c->cur = s;
    c->owns_stream = true;
    return NLMC_OK;
}

int nlmc_overlap_subsets(nlmc_ctx *c, int on)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->cur != c->stream) return fail(c, NLMC_ERR_STATE, "nlmc_overlap_subsets: select all chains first");
    if (on && !c->aux) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    }
    c->overlap = on != 0;
    return NLMC_OK;
}

int nlmc_subset_count(const nlmc_ctx *c) { return c ? c->sub_count() : 0; }

int nlmc_get_subset(nlmc_ctx *c, int32_t *out_chains)
{
    if (!c || !out_chains) return fail(c, NLMC_ERR_ARG, "nlmc_get_subset: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const int R = c->sub_count();
    if (c->subset == 0) { for (int i = 0; i < R; ++i) out_chains[i] = i; return NLMC_OK; }
    { int rc = ensure_subset(c); if (rc) return rc; }
    HIP_TRY(c, hipMemcpyAsync(out_chains, c->sub_list(), sizeof(int32_t) * (size_t)R, hipMemcpyDeviceToHost, c->cur));
    HIP_TRY(c, hipStreamSynchronize(c->cur));
    return NLMC_OK;
}

int nlmc_track_minimum(nlmc_ctx *c, int on)
{
    if (!c) return NLMC_ERR_ARG;
    c->track_min = on != 0;
    c->track_min_stride = on > 1 ? on : 1;
    return NLMC_OK;
}

int nlmc_backbone_seed(nlmc_ctx *c, int mode)
{
    if (!c) return NLMC_ERR_ARG;
    if (mode < 0 || mode > 1) return fail(c, NLMC_ERR_ARG, "nlmc_backbone_seed: mode must be 0 or 1");
    HIP_TRY(c, hipSetDevice(c->device));
    if (mode == 0) { c->seed_snap_on = false; return NLMC_OK; }
    const size_t bytes = (size_t)std::max(c->n_chains, 1) * c->n_pad;
    HIP_TRY(c, c->seed_snap.reserve(bytes));
    HIP_TRY(c, hipMemcpyAsync(c->seed_snap.p, c->spins.p, bytes, hipMemcpyDeviceToDevice, c->cur));
    c->seed_snap_on = true;
    return NLMC_OK;
}

int nlmc_adopt_best(nlmc_ctx *c)
{
    if (!c) return NLMC_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    const int R = c->sub_count();
    if (R == 0) return NLMC_OK;
    { int rc = ensure_subset(c); if (rc) return rc; }
    hipLaunchKernelGGL(k_adopt_best, dim3(R), dim3(256), 0, c->cur, c->n_pad, c->sub_list(), c->spins.p, c->best.p, c->efix.p, c->emin.p);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int nlmc_backbone_clusters(nlmc_ctx *c, const double *epsilon, const double *lambdas, int n_lambdas, double beta, double tolerance,
                           int max_iterations, double sat, const double *thresholds, int n_thresholds)
{
    if (!c) return NLMC_ERR_ARG;
    if (!epsilon || !lambdas || n_lambdas < 1 || !thresholds || n_thresholds < 1 || max_iterations < 0 || !(beta != 0.0))
        return fail(c, NLMC_ERR_ARG, "nlmc_backbone_clusters: bad sizes or NULL arrays");
    HIP_TRY(c, hipSetDevice(c->device));
    const int P = c->sub_count();
    if (P == 0) return NLMC_OK;
    { int rc = ensure_subset(c); if (rc) return rc; }
    { int rc = lbp_setup(c, P, epsilon, lambdas, n_lambdas, beta, false); if (rc) return rc; }
    HIP_TRY(c, c->cmask.reserve((size_t)c->n_chains * c->n_pad));
    if (!c->nmc_status.p) {
        HIP_TRY(c, c->nmc_status.reserve(1));
        HIP_TRY(c, hipMemsetAsync(c->nmc_status.p, 0, sizeof(int32_t), c->cur));
    }
    const std::vector<double> thr(thresholds, thresholds + n_thresholds);
    if (thr != c->nmc_thr_host) {
        HIP_TRY(c, c->nmc_thr.reserve((size_t)n_thresholds));
        HIP_TRY(c, hipStreamSynchronize(c->cur));
        HIP_TRY(c, hipMemcpyAsync(c->nmc_thr.p, thr.data(), sizeof(double) * thr.size(), hipMemcpyHostToDevice, c->cur));
        HIP_TRY(c, hipStreamSynchronize(c->cur));
        c->nmc_thr_host = thr;
    }
    hipLaunchKernelGGL(k_lbp_seeds, dim3(P), dim3(256), 0, c->cur, c->n, c->n_pad, c->sub_list(), c->seed_snap_on ? c->seed_snap.p : c->spins.p, c->lbp_ms.p);
    HIP_TRY(c, hipGetLastError());
    { int rc = lbp_launch(c, P, n_lambdas, beta, tolerance, max_iterations, sat, false); if (rc) return rc; }
    if (c->big || (size_t)2 * c->n_pad > (size_t)60 * 1024) {
        HIP_TRY(c, c->cmask_scratch.reserve((size_t)P * 2 * c->n_pad));
        hipLaunchKernelGGL(k_cluster_mask<true>, dim3(P), dim3(1024), 0, c->cur, c->g, c->sub_list(), c->lbp_mag.p,
                           c->lbp_out_i.p + P, c->nmc_thr.p, n_thresholds, c->cmask.p, c->nmc_status.p, c->cmask_scratch.p);
    } else
        hipLaunchKernelGGL(k_cluster_mask<false>, dim3(P), dim3(256), (size_t)2 * c->n_pad, c->cur, c->g, c->sub_list(), c->lbp_mag.p,
                           c->lbp_out_i.p + P, c->nmc_thr.p, n_thresholds, c->cmask.p, c->nmc_status.p, (uint8_t *)nullptr);
    HIP_TRY(c, hipGetLastError());
    return NLMC_OK;
}

int nlmc_backbone_check(nlmc_ctx *c)
{
    if (!c) return NLMC_ERR_ARG;
    if (!c->nmc_status.p) return NLMC_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int32_t st = 0;
    HIP_TRY(c, hipMemcpyAsync(&st, c->nmc_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (st != 0) {
        HIP_TRY(c, hipMemsetAsync(c->nmc_status.p, 0, sizeof(int32_t), c->stream));
        if (st & 2) return fail(c, NLMC_ERR_HIP, "nlmc_backbone_clusters: the workgroups of a problem lost each other (group barrier timed out)");
        return fail(c, NLMC_ERR_ARG, "LBP diverged at initial lambda, please try a larger lambda_start or increase max_iterations or beta");
    }
    return NLMC_OK;
}

int nlmc_get_cluster_mask(nlmc_ctx *c, uint8_t *out)
{
    if (!c || !out) return fail(c, NLMC_ERR_ARG, "nlmc_get_cluster_mask: NULL argument");
    if (!c->cmask.p) return fail(c, NLMC_ERR_STATE, "nlmc_get_cluster_mask: no backbone inference has run");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    int rc = rows_to_host_begin(c, c->cmask.p, c->n_chains);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rows_to_host_finish(c, out, c->n_chains);
    return NLMC_OK;
}

int nlmc_set_cluster_mask(nlmc_ctx *c, const uint8_t *mask)
{
    if (!c || !mask) return fail(c, NLMC_ERR_ARG, "nlmc_set_cluster_mask: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->n_chains == 0) return NLMC_OK;
    HIP_TRY(c, c->cmask.reserve((size_t)c->n_chains * c->n_pad));
    for (size_t i = 0; i < (size_t)c->n_chains * c->n; ++i)
        if (mask[i] > 1) return fail(c, NLMC_ERR_ARG, "nlmc_set_cluster_mask: entries must be 0 or 1");
    int rc = rows_to_device(c, c->cmask.p, mask, c->n_chains);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return NLMC_OK;
}

int nlmc_set_phase(nlmc_ctx *c, int kind, double temp_x)
{
    if (!c) return NLMC_ERR_ARG;
    if (kind < NLMC_PHASE_ALL || kind > NLMC_PHASE_BACKBONE_FROZEN) return fail(c, NLMC_ERR_ARG, "nlmc_set_phase: bad phase");
    HIP_TRY(c, hipSetDevice(c->device));
    if (kind == NLMC_PHASE_ALL) { c->has_flags = false; return NLMC_OK; }     // (temp_x stays: the uploaded temperature tables stay valid)
    if (!(temp_x > 0.0) && !(temp_x < 0.0)) return fail(c, NLMC_ERR_ARG, "nlmc_set_phase: temp_x must be non-zero");
    if (!c->cmask.p) return fail(c, NLMC_ERR_STATE, "nlmc_set_phase: no backbone inference has run");
    const int R = c->sub_count();
    if (R > 0) {
        { int rc = ensure_subset(c); if (rc) return rc; }
        hipLaunchKernelGGL(k_phase_flags, dim3(R), dim3(256), 0, c->cur, c->n_pad, c->sub_list(), c->cmask.p, kind, c->flags.p);
        HIP_TRY(c, hipGetLastError());
    }
    c->has_flags = true;
    c->temp_x = temp_x;
    return NLMC_OK;
}

int nlmc_plan_slot(nlmc_ctx *c, int slot)
{
    if (!c) return NLMC_ERR_ARG;
    if (slot < 0 || slot > 1) return fail(c, NLMC_ERR_ARG, "nlmc_plan_slot: slot must be 0 or 1");
    c->fz_slot = slot;
    return NLMC_OK;
}

int nlmc_find_clusters(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals, const double *mag,
                       double threshold_initial, double threshold_cutoff, double threshold_step, int32_t *out_members,
                       int64_t members_capacity, int32_t *out_sizes, int32_t *out_n_clusters)
{
    if (n < 1 || !rowptr || !mag || !out_members || !out_sizes || !out_n_clusters || (rowptr[n] > 0 && (!colidx || !vals)))
        return fail(nullptr, NLMC_ERR_ARG, "nlmc_find_clusters: bad sizes or NULL arrays");
    if (!(threshold_step > 0.0)) return fail(nullptr, NLMC_ERR_ARG, "nlmc_find_clusters: threshold_step must be > 0");
    if (host_find_clusters(n, rowptr, colidx, vals, mag, threshold_initial, threshold_cutoff, threshold_step, out_members,
                           members_capacity, out_sizes, out_n_clusters) != 0)
        return fail(nullptr, NLMC_ERR_ARG, "nlmc_find_clusters: out_members too small");
    return NLMC_OK;
}

int nlmc_host_prefault(void *ptr, int64_t bytes, int n_threads)
{
    if (!ptr || bytes < 0) return fail(nullptr, NLMC_ERR_ARG, "nlmc_host_prefault: bad argument");
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int64_t page = 4096, pages = (bytes + page - 1) / page;
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(n_threads, pages / 256 + 1));
    auto part = [&](int64_t p0, int64_t p1) {
        volatile unsigned char *b = static_cast<volatile unsigned char *>(ptr);
        for (int64_t q = p0; q < p1; ++q) b[q * page] = 0;
    };
    if (nt == 1) { part(0, pages); return NLMC_OK; }
    std::vector<std::thread> th;
    for (int64_t i = 0; i < nt; ++i) th.emplace_back(part, pages * i / nt, pages * (i + 1) / nt);
    for (auto &t : th) t.join();
    return NLMC_OK;
}

int nlmc_trace_layout(const int8_t *src, int64_t n_blocks, int64_t n_sweeps, int64_t n, const int32_t *dst_block,
                      const int32_t *dst_col, int64_t n_dst_blocks, int64_t row_len, void *dst, int elem_bytes, int n_threads)
{
    if (n_blocks < 0 || n_sweeps < 0 || n < 1 || (n_blocks * n_sweeps > 0 && (!src || !dst)) || row_len < n_sweeps ||
        (!dst_block && n_dst_blocks < n_blocks))
        return fail(nullptr, NLMC_ERR_ARG, "nlmc_trace_layout: bad sizes or NULL arrays");
    if (elem_bytes != 1 && elem_bytes != 8) return fail(nullptr, NLMC_ERR_ARG, "nlmc_trace_layout: elem_bytes must be 1 or 8");
    if ((dst_block || dst_col) && n_sweeps > 0) {    // destinations: in range, whole column groups, distinct (threads write disjointly)
        const int64_t groups = row_len / n_sweeps;
        std::vector<uint8_t> seen((size_t)(n_dst_blocks * groups), 0);
        for (int64_t b = 0; b < n_blocks; ++b) {
            const int64_t r = dst_block ? dst_block[b] : b, c = dst_col ? dst_col[b] : 0;
            if (r < 0 || r >= n_dst_blocks || c < 0 || c % n_sweeps || c + n_sweeps > row_len || seen[r * groups + c / n_sweeps])
                return fail(nullptr, NLMC_ERR_ARG, "nlmc_trace_layout: destinations must be distinct (block, column group) pairs in range");
            seen[r * groups + c / n_sweeps] = 1;
        }
    }
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (elem_bytes == 8) host_trace_layout<double>(src, n_blocks, n_sweeps, n, dst_block, dst_col, row_len, (double *)dst, n_threads);
    else host_trace_layout<int8_t>(src, n_blocks, n_sweeps, n, dst_block, dst_col, row_len, (int8_t *)dst, n_threads);
    return NLMC_OK;
}

}  // extern "C"
