// nlmc_big.h -- chains that do not fit in LDS (gfx950): the same sweeps with the spins left in global memory.
//
// The reference takes any N (NMC/nmc.py:49-53).  The kernels of nlmc_kernels.h keep a chain's spins (and, per mode, its uniforms /
// thresholds and the levelizer's keys) in the 160 KB of LDS of one CU, which ends at N = 24 576 (fp64 uniforms: earlier).  Past
// that the same level schedule is walked with the spins left in their row of `spins` in global memory and the rows read from
// the CSR arrays.  Same arithmetic as the LDS kernels, term by term: stream mode = the
// reference's (NMC/nmc.py:86-87), "f64" / "f32" = the Philox modes' spec (nlmc_device.h, oracle/nlo.c) -- the parity tests
// compare them with the same oracles, and at sizes both paths take with each other (NLMC_FORCE_BIG).
//
// Spin indices are 32 bits here (the LDS kernels pack k and the degree into one word).  A level is one kernel launch over all
// chains of the call (the launch boundary is the barrier), the level schedule a handful of grid-wide kernels: a single chain of a
// million spins uses the whole chip.  Throughput is still not the point of this path: a level costs a launch and a few dependent
// L2 round trips instead of LDS ones.
#pragma once
#include "nlmc_kernels.h"

// ---- level schedule: a handful of grid-wide kernels per batch of orders ---------------------------------------------------------
// (one workgroup per order -- the LDS levelizer's shape -- leaves the chip idle when a call has few orders of many spins: 52 ms
// per order at N = 10^6)
struct BigLevelizeArgs {
    CsrDev g;
    int n_orders;
    const uint32_t *keys_in;    // [n_orders][n] ranks (stream mode), or nullptr: Philox ORDER keys, written to `key`
    uint32_t seed_lo, seed_hi, sweep0;
    int per_chain, n_sweeps, chain_base;
    uint32_t *key;              // scratch [n_orders][n] (Philox keys)
    int32_t *lvl;               // scratch [n_orders][n]
    uint32_t *cur;              // scratch [n_orders][n + 2]: level histogram, then the placement cursors
    int32_t *flag;              // [NLMC_BIG_PASSES] one "something changed" word per relaxation pass of a batch
    int2 *ord2;                 // [n_orders][n]  { k, row start }
    int32_t *lvl_off;           // [n_orders][n + 1]
    int32_t *nlev, *hi_max;     // [n_orders]
};
#define NLMC_BIG_PASSES 8       // relaxation passes between two looks at the flags

// grid (ceil(n / 256), n_orders)
__global__ __launch_bounds__(256) void k_blv_init(BigLevelizeArgs a)
{
    const int n = a.g.n, k = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t o = blockIdx.y;
    if (k == 0) { a.nlev[o] = 0; a.hi_max[o] = n; }
    for (int l = k; l < n + 2; l += gridDim.x * blockDim.x) a.cur[o * ((size_t)n + 2) + l] = 0u;
    if (k >= n) return;
    a.lvl[o * (size_t)n + k] = -1;                 // not known yet
    if (!a.keys_in) {
        const uint32_t t = a.sweep0 + (uint32_t)(a.per_chain ? ((int)o % a.n_sweeps) : (int)o);
        const uint32_t grp = a.per_chain ? (uint32_t)(a.chain_base + (int)o / a.n_sweeps + 1) : 0u;
        a.key[o * (size_t)n + k] = philox4x32_10((uint32_t)k, t, grp, NLMC_TAG_ORDER, a.seed_lo, a.seed_hi).x;
    }
}

// One pass of the level assignment: level of k = 1 + the largest level among the neighbours that precede k in the order (0 without
// one).  A spin takes its level once every such neighbour has one -- only final values are ever written, so reading a level another
// thread sets during the same pass is as good as reading it in the next; pass p + 1 at the latest settles level p.  Spins that
// are done leave after one load, waiting ones at their first unsettled neighbour.  flag[pass] != 0 iff some spin still waits.
__global__ __launch_bounds__(256) void k_blv_pass(BigLevelizeArgs a, int pass)
{
    const int n = a.g.n, k = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t o = blockIdx.y;
    int waiting = 0;
    if (k < n) {
        int32_t *lvl = a.lvl + o * (size_t)n;
        if (lvl[k] < 0) {
            const uint32_t *key = (a.keys_in ? a.keys_in : a.key) + o * (size_t)n;
            const uint32_t kk = key[k];
            const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
            int m = 0;
            for (int e = rs; e < re; ++e) {
                const int j = a.g.col[e];
                if (j == k || !precedes(key[j], j, kk, k)) continue;
                const int lj = lvl[j];
                if (lj < 0) { waiting = 1; break; }
                m = max(m, lj + 1);
            }
            if (!waiting) lvl[k] = m;
        }
    }
    if (__any(waiting) && (threadIdx.x & 63) == 0) atomicOr(&a.flag[pass], 1);
}

// histogram of the levels (cur[o][l]) and their number (nlev[o] = 1 + the largest level).  A workgroup counts a tile of
// NLMC_BIG_TILE spins in LDS first and adds its non-empty bins to the global histogram (a sparse graph has a few dozen levels: a
// global atomic per spin would queue a million updates on each of them); levels past the LDS bins go straight to global memory.
#define NLMC_BIG_TILE 4096
#define NLMC_BIG_LBINS 2048
// grid (ceil(n / NLMC_BIG_TILE), n_orders) x 256 threads
__global__ __launch_bounds__(256) void k_blv_hist(BigLevelizeArgs a)
{
    __shared__ uint32_t h[NLMC_BIG_LBINS];
    const int n = a.g.n, tid = threadIdx.x;
    const size_t o = blockIdx.y;
    const int32_t *lvl = a.lvl + o * (size_t)n;
    uint32_t *cnt = a.cur + o * ((size_t)n + 2);
    for (int l = tid; l < NLMC_BIG_LBINS; l += 256) h[l] = 0u;
    __syncthreads();
    int m = -1;
    for (int q = 0; q < NLMC_BIG_TILE / 256; ++q) {
        const int k = blockIdx.x * NLMC_BIG_TILE + q * 256 + tid;
        if (k >= n) break;
        const int l = lvl[k];
        m = max(m, l);
        if (l < NLMC_BIG_LBINS) atomicAdd(&h[l], 1u); else atomicAdd(&cnt[l], 1u);
    }
    __syncthreads();
    for (int l = tid; l < NLMC_BIG_LBINS; l += 256) if (h[l]) atomicAdd(&cnt[l], h[l]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if ((tid & 63) == 0 && m >= 0) atomicMax(&a.nlev[o], m + 1);
}

// exclusive scan of the histogram -> level offsets and placement cursors; one workgroup per order
__global__ __launch_bounds__(1024) void k_blv_scan(BigLevelizeArgs a)
{
    __shared__ int sh_scan[16];
    const int n = a.g.n, tid = threadIdx.x, nt = blockDim.x;
    const size_t o = blockIdx.x;
    const int nl = a.nlev[o];
    uint32_t *cnt = a.cur + o * ((size_t)n + 2);
    int32_t *off = a.lvl_off + o * ((size_t)n + 1);
    const int chunk = (nl + nt - 1) / nt;
    const int b = min(tid * chunk, nl), e = min(b + chunk, nl);
    int s = 0;
    for (int l = b; l < e; ++l) s += (int)cnt[l];
    const int lane = tid & 63, wv = tid >> 6;
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) sh_scan[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += sh_scan[w];
    int run = base + incl - s;
    for (int l = b; l < e; ++l) { const int c = (int)cnt[l]; off[l] = run; cnt[l] = (uint32_t)run; run += c; }
    if (tid == 0) off[nl] = n;
}

// placement (the order inside a level does not matter: its spins are independent): a workgroup ranks the spins of its tile
// inside their levels in LDS, reserves one range per non-empty level from the global cursors and writes its spins there
__global__ __launch_bounds__(256) void k_blv_place(BigLevelizeArgs a)
{
    __shared__ uint32_t h[NLMC_BIG_LBINS], base[NLMC_BIG_LBINS];
    const int n = a.g.n, tid = threadIdx.x;
    const size_t o = blockIdx.y;
    const int32_t *lvl = a.lvl + o * (size_t)n;
    uint32_t *cur = a.cur + o * ((size_t)n + 2);
    int2 *ord = a.ord2 + o * (size_t)n;
    for (int l = tid; l < NLMC_BIG_LBINS; l += 256) h[l] = 0u;
    __syncthreads();
    int lv[NLMC_BIG_TILE / 256];
    uint32_t rk[NLMC_BIG_TILE / 256];
#pragma unroll
    for (int q = 0; q < NLMC_BIG_TILE / 256; ++q) {
        const int k = blockIdx.x * NLMC_BIG_TILE + q * 256 + tid;
        lv[q] = k < n ? lvl[k] : -1;
        rk[q] = (lv[q] >= 0 && lv[q] < NLMC_BIG_LBINS) ? atomicAdd(&h[lv[q]], 1u) : 0u;
    }
    __syncthreads();
    for (int l = tid; l < NLMC_BIG_LBINS; l += 256) if (h[l]) base[l] = atomicAdd(&cur[l], h[l]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NLMC_BIG_TILE / 256; ++q) {
        const int k = blockIdx.x * NLMC_BIG_TILE + q * 256 + tid;
        if (lv[q] < 0) continue;
        const uint32_t pos = lv[q] < NLMC_BIG_LBINS ? base[lv[q]] + rk[q] : atomicAdd(&cur[lv[q]], 1u);
        ord[pos] = make_int2(k, a.g.rowptr[k]);
    }
}

// rank[perm[i]] = i, u_spin[perm[i]] = u[i] (k_stream_scatter); `rank` arrives filled with 0xFFFFFFFF, a second hit of a spin
// or an index outside [0, n) sets *bad
__global__ void k_stream_scatter_big(int n, const int32_t *perm, const double *u, uint32_t *rank, double *us, int32_t *bad)
{
    const size_t o = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = perm[o * n + i];
    if (k < 0 || k >= n || atomicExch(&rank[o * n + k], (uint32_t)i) != 0xFFFFFFFFu) { atomicOr(bad, 1); return; }
    us[o * n + k] = u[o * n + i];
}

// ---- sweeps: one launch per level ----------------------------------------------------------------------------------------------
// Every chain of the call takes level l of its order of sweep t in the same launch (grid.y = chain, grid.x workgroups share the
// level's spins): the kernel boundary is the barrier between levels, a single long chain uses the whole chip, and many chains
// balance better than one workgroup each.  Energy deltas are summed into esum[block row] (exact integers: any order), folded into
// the tracked energy by k_big_sweep_end -- after every sweep when the call wants per-sweep outputs, once at the end otherwise.
#define NLMC_BIG_STREAM 0
#define NLMC_BIG_F32 1
#define NLMC_BIG_F64 2

// One heat-bath update of spin k (row start rs) of the chain described by b, in the arithmetic of MODE; returns the change of the
// tracked fixed-point energy.
struct BigChain { int8_t *s; const uint8_t *fl; const double *ut; double esc, tb0, tb1; uint32_t gc, tt; };

template <int MODE>
__device__ __forceinline__ long long big_update(const SweepArgs &a, const BigChain &b, int k, int rs)
{
    long long dE = 0;
    const int re = a.g.rowptr[k + 1];
    const unsigned f = b.fl ? (unsigned)b.fl[k] : 0u;
    const int so = (int)b.s[k];
    if constexpr (MODE == NLMC_BIG_STREAM) {
        // NMC/nmc.py:86-87 on the unmodified / scaled / frozen row (k_sweep_stream)
        double xs = 0.0, xd = 0.0;
        for (int e = rs; e < re; ++e) {
            const int j = a.g.col[e];
            const double tm = a.g.val64[e] * (double)b.s[j];
            xs += tm;
            if (j == k) xd += tm;
        }
        const double hk = a.g.h64[k];
        const double x_true = (xs - xd) + hk;
        double xp;
        if (f == 0u) xp = xs + hk;
        else if (f == 1u) {
            double y = 0.0;
            for (int e = rs; e < re; ++e) y += (a.g.val64[e] / a.temp_x) * (double)b.s[a.g.col[e]];
            xp = y + hk / a.temp_x;
        } else xp = xs + ((f == 2u) ? 10000.0 : -10000.0);
        const double v = tanh(b.tb0 * xp) - 2.0 * b.ut[k] + 1.0;
        const int sn = (v > 0.0) - (v < 0.0);
        if (sn != so) {
            dE = __double2ll_rn(-(double)(sn - so) * x_true * b.esc);
            b.s[k] = (int8_t)sn;
        }
    } else if constexpr (MODE == NLMC_BIG_F64) {
        if (f >= 2u) return 0;                                  // frozen
        const u32x4 rh = philox4x32_10((uint32_t)(k >> 2), b.tt, b.gc, NLMC_TAG_UNIFORM, a.seed_lo, a.seed_hi);
        const u32x4 rl = philox4x32_10((uint32_t)(k >> 2), b.tt, b.gc, NLMC_TAG_UNIFORM_LO, a.seed_lo, a.seed_hi);
        const int w = k & 3;
        const uint32_t hw = w == 0 ? rh.x : w == 1 ? rh.y : w == 2 ? rh.z : rh.w;
        const uint32_t lw = w == 0 ? rl.x : w == 1 ? rl.y : w == 2 ? rl.z : rl.w;
        const double uk = uniform53_spec(hw, lw);
        double xs = 0.0, xd = 0.0;
        for (int e = rs; e < re; ++e) {
            const int j = a.g.col[e];
            const double v = a.g.val64[e], sv = (double)b.s[j];
            xs = __fma_rn(v, sv, xs);
            if (j == k) xd = __fma_rn(v, sv, xd);
        }
        const double hk = a.g.h64[k];
        const double x_true = (xs - xd) + hk;                   // (xd == 0 without a self-coupling: the same bits as xs + hk)
        const double z = (f == 1u ? b.tb1 : b.tb0) * (xs + hk);
        const int sn = accept_up(uk, z) ? 1 : -1;
        const int ds = sn - so;
        if (ds != 0) dE = fixed_delta_slow(x_true, ds, b.esc);
        b.s[k] = (int8_t)sn;
    } else {
        if (f >= 2u) return 0;
        const u32x4 r = philox4x32_10((uint32_t)(k >> 2), b.tt, b.gc, NLMC_TAG_UNIFORM, a.seed_lo, a.seed_hi);
        const int w = k & 3;
        const float wk = threshold_spec(w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w);
        int X = a.g.hq[k], Xd = 0;
        for (int e = rs; e < re; ++e) {
            const EdgeQ q = a.g.edge32[e];
            const int tm = q.q * (int)b.s[q.col];
            X += tm;
            if (q.col == k) Xd += tm;
        }
        const float cq = (float)(f == 1u ? b.tb1 : b.tb0) * a.qinv;
        const float z = cq * (float)X;
        const int sn = (z < wk) ? 1 : -1;
        const int cc = (so - sn) << a.eshift;
        dE = (long long)(X - Xd) * (long long)cc;
        b.s[k] = (int8_t)sn;
    }
    return dE;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_big_level(SweepArgs a, int t, int l, long long *esum)
{
    const int ob = blockIdx.y, n = a.g.n;
    const int c = a.chain_list ? a.chain_list[ob] : ob;
    // (order ids as in the LDS kernels: the stream mode's by chain id, the Philox modes' by block)
    const int oid = a.per_chain ? ((MODE == NLMC_BIG_STREAM ? c : ob) * a.n_sweeps + t) : t;
    if (l >= a.nlev[oid]) return;
    const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * ((size_t)n + 1);
    const int lo = off[l], hi = off[l + 1];
    const int2 *__restrict__ ord = a.ord2 + (size_t)oid * n;
    const uint32_t gc_chain = (uint32_t)(a.chain_base + c);
    const int row = (MODE != NLMC_BIG_STREAM && a.slot_of_chain) ? a.slot_of_chain[gc_chain] : c;
    BigChain b;
    b.s = a.spins + (size_t)c * a.g.n_pad;
    b.fl = a.flags ? a.flags + (size_t)c * a.g.n_pad : nullptr;
    b.esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale
    b.gc = (a.rng_stride && a.slot_of_chain) ? (uint32_t)((c / a.rng_ladder_len) * a.rng_stride + a.rng_base + row) : gc_chain;
    b.tt = a.sweep0 + (uint32_t)t;
    b.tb0 = a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss];
    b.tb1 = MODE == NLMC_BIG_STREAM ? 0.0 : a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss + 1];
    b.ut = MODE == NLMC_BIG_STREAM ? a.ustream + ((size_t)c * a.n_sweeps + t) * n : nullptr;
    long long e_loc = 0;
    for (int i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += gridDim.x * blockDim.x) {
        const int2 en = ord[i];
        e_loc += big_update<MODE>(a, b, en.x, en.y);
    }
    e_loc = wave_sum_i64(e_loc);
    if ((threadIdx.x & 63) == 0 && e_loc != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&esum[ob]), (unsigned long long)e_loc);
}

// End of sweep t of every chain (one workgroup each): the sweep's energy delta into the tracked energy, then the bookkeeping of
// the LDS kernels (sweep_epilogue: energy trace, running minimum + its state, recorded configurations) on the spins in place.
__global__ __launch_bounds__(1024) void k_big_sweep_end(SweepArgs a, int t, long long *esum)
{
    __shared__ long long red_s[2];
    ChainCtx x;
    x.st = nullptr; x.lvl_t = nullptr; x.ustride = 0;
    x.n = a.g.n; x.n_pad = a.g.n_pad;
    x.tid = threadIdx.x; x.nt = blockDim.x; x.ob = blockIdx.x;
    x.c = a.chain_list ? a.chain_list[blockIdx.x] : (int)blockIdx.x;
    x.s = a.spins + (size_t)x.c * x.n_pad;
    x.fl = nullptr;
    x.red = red_s;
    if (x.tid == 0) { red_s[0] = 0; red_s[1] = 0; }
    x.e_loc = 0;
    if (x.tid == 0) { x.e_loc = esum[x.ob]; esum[x.ob] = 0; }
    x.E = uniform64(a.efix[x.c]);
    x.Emin = a.emin ? uniform64(a.emin[x.c]) : 0;
    x.amin = a.emin ? __builtin_amdgcn_readfirstlane(a.argmin[x.c]) : 0;
    x.per_sweep = true;
    __syncthreads();
    sweep_epilogue(a, x, t);
    if (x.tid == 0) {
        a.efix[x.c] = x.E;
        if (a.energy_sink) a.energy_sink[x.c] = (double)x.E * __longlong_as_double((long long)(1023 - a.escale) << 52);
        if (a.emin) { a.emin[x.c] = x.Emin; a.argmin[x.c] = x.amin; }
    }
}

// ---- sweeps: one workgroup per chain -------------------------------------------------------------------------------------------
// The shape of the LDS kernels with the spins in global memory: a workgroup walks the levels of its chain's order, a thread loops
// over its share of a level, __syncthreads() between levels (all waves of a workgroup share their CU's write-through vector L1, so
// one level's stores are seen by the next level's loads; a chain's spins stay in that L1 / the XCD's L2 from level to level, which
// the launch-per-level kernels above give up at every launch).  Taken when the call has enough chains to fill the chip.
template <int MODE>
__global__ __launch_bounds__(1024) void k_sweep_big(SweepArgs a)
{
    __shared__ long long red_s[2];
    ChainCtx x;
    x.st = nullptr; x.lvl_t = nullptr; x.ustride = 0;
    x.n = a.g.n; x.n_pad = a.g.n_pad;
    x.tid = threadIdx.x; x.nt = blockDim.x; x.ob = blockIdx.x;
    x.c = a.chain_list ? a.chain_list[blockIdx.x] : (int)blockIdx.x;
    x.s = a.spins + (size_t)x.c * x.n_pad;
    x.fl = a.flags ? const_cast<uint8_t *>(a.flags) + (size_t)x.c * x.n_pad : nullptr;
    x.red = red_s;
    if (x.tid == 0) { red_s[0] = 0; red_s[1] = 0; }
    x.e_loc = 0;
    x.E = uniform64(a.efix[x.c]);
    x.Emin = a.emin ? uniform64(a.emin[x.c]) : 0;
    x.amin = a.emin ? __builtin_amdgcn_readfirstlane(a.argmin[x.c]) : 0;
    x.per_sweep = (a.etrace != nullptr) || (a.emin != nullptr);
    __syncthreads();

    const int n = x.n, tid = x.tid, nt = x.nt, c = x.c;
    const uint32_t gc_chain = (uint32_t)(a.chain_base + c);
    const int row = (MODE != NLMC_BIG_STREAM && a.slot_of_chain) ? a.slot_of_chain[gc_chain] : c;
    BigChain b;
    b.s = x.s;
    b.fl = x.fl;
    b.esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale
    b.gc = (a.rng_stride && a.slot_of_chain) ? (uint32_t)((c / a.rng_ladder_len) * a.rng_stride + a.rng_base + row) : gc_chain;

    for (int t = 0; t < a.n_sweeps; ++t) {
        // (order ids as in the LDS kernels: the stream mode's by chain id, the Philox modes' by block)
        const int oid = a.per_chain ? ((MODE == NLMC_BIG_STREAM ? c : x.ob) * a.n_sweeps + t) : t;
        const int2 *__restrict__ ord = a.ord2 + (size_t)oid * n;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * ((size_t)n + 1);
        const int nl = a.nlev[oid];
        b.tt = a.sweep0 + (uint32_t)t;
        b.tb0 = a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss];
        b.tb1 = MODE == NLMC_BIG_STREAM ? 0.0 : a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss + 1];
        b.ut = MODE == NLMC_BIG_STREAM ? a.ustream + ((size_t)c * a.n_sweeps + t) * n : nullptr;
        for (int l = 0; l < nl; ++l) {
            const int lo = off[l], hi = off[l + 1];
            for (int i = lo + tid; i < hi; i += nt) {
                const int2 en = ord[i];
                x.e_loc += big_update<MODE>(a, b, en.x, en.y);
            }
            __syncthreads();
        }
        sweep_epilogue(a, x, t);
    }
    // (chain_store without the copy of the spins: they are where they belong)
    if (!x.per_sweep) {
        const long long w = wave_sum_i64(x.e_loc);
        if ((x.tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&x.red[0]), (unsigned long long)w);
        __syncthreads();
        x.E += uniform64(x.red[0]);
    }
    if (x.tid == 0) {
        a.efix[x.c] = x.E;
        if (a.energy_sink) a.energy_sink[x.c] = (double)x.E * __longlong_as_double((long long)(1023 - a.escale) << 52);
        if (a.emin) { a.emin[x.c] = x.Emin; a.argmin[x.c] = x.amin; }
    }
}

// ---- iso-cluster move (csrc/nlmc_pt_icm.h) on chains too long for LDS ------------------------------------------------------------
#include "nlmc_pt_icm.h"

// k_icm_components with the union-find forest in the output labels themselves (global memory).  Same result: the label of a
// disagreeing spin is the smallest member of its component, INT_MAX elsewhere; info = {number of components, 0}.
__global__ __launch_bounds__(1024) void k_icm_components_big(IcmArgs a)
{
    __shared__ int nroots;
    const int n = a.g.n, tid = threadIdx.x, nt = blockDim.x, p = blockIdx.x;
    const int8_t *sa = a.spins + (size_t)a.pairs[2 * p] * a.g.n_pad;
    const int8_t *sb = a.spins + (size_t)a.pairs[2 * p + 1] * a.g.n_pad;
    int32_t *lab = a.label + (size_t)p * n;
    if (tid == 0) nroots = 0;
    for (int k = tid; k < n; k += nt) lab[k] = ((int)sa[k] * (int)sb[k] == -1) ? k : INT_MAX;
    __syncthreads();
    bool converged = false;
    for (int it = 0; it <= n; ++it) {
        int changed = 0;
        for (int k = tid; k < n; k += nt) {
            if (lab[k] == INT_MAX) continue;
            int rk = icm_find(lab, k);
            const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
            for (int e = rs; e < re; ++e) {
                const EdgeQ t = a.g.edge32[e];
                const int j = t.col;
                if ((t.q == 0 && a.g.val64[e] == 0.0) || j == k || lab[j] == INT_MAX) continue;   // (`val != 0`, NPT/apt_ICM.py:129)
                const int rj = icm_find(lab, j);
                if (rj < rk) { atomicMin(&lab[rk], rj); rk = rj; changed = 1; }
                else if (rk < rj) { atomicMin(&lab[rj], rk); changed = 1; }
            }
        }
        if (!__syncthreads_or(changed)) { converged = true; break; }
    }
    int cnt = 0;
    for (int k = tid; k < n; k += nt) {
        int l = lab[k];
        if (l != INT_MAX) l = icm_find(lab, k);      // (points every spin at its root: a pointer is only ever replaced by an ancestor)
        cnt += (l == k);
    }
    if (cnt) atomicAdd(&nroots, cnt);
    __syncthreads();
    for (int k = tid; k < n; k += nt) if (lab[k] != INT_MAX) lab[k] = icm_find(lab, k);
    if (tid == 0) { a.info[2 * p] = converged ? nroots : -1; a.info[2 * p + 1] = 0; }
}

// k_icm_round (components, pick, move, energy bookkeeping of one pair in one workgroup) with the labels in global scratch
// `lab_g` [n_pairs][n] and both configurations read and exchanged in place.  Same pairing keys, same pick, same integers.
__global__ __launch_bounds__(1024) void k_icm_round_big(IcmRoundArgs a, int32_t *lab_g)
{
    __shared__ int nroots, sh_root, sh_size;
    __shared__ int sh_scan[17];
    __shared__ long long sh_dE[2];
    __shared__ int sh_pair[2], sh_lad[2];
    const int n = a.g.n, n_pad = a.g.n_pad, tid = threadIdx.x, nt = blockDim.x, p = blockIdx.x;
    if (!a.pairs) {
        const int half = a.pair_K / 2, r = p / half, i = p % half;
        const uint32_t rg = (uint32_t)(r + a.slot0);
        if (tid < a.pair_K) {
            const uint32_t kj = philox4x32_10((uint32_t)tid, a.round, rg, NLMC_TAG_ICM_PAIR, a.seed_lo, a.seed_hi).x;
            int rank = 0;
            for (int q = 0; q < a.pair_K; ++q) {
                if (q == tid) continue;
                const uint32_t kq = philox4x32_10((uint32_t)q, a.round, rg, NLMC_TAG_ICM_PAIR, a.seed_lo, a.seed_hi).x;
                rank += (kq < kj) || (kq == kj && q < tid);
            }
            if (rank == 2 * i || rank == 2 * i + 1) { sh_pair[rank & 1] = a.chain_of_slot[(size_t)tid * a.pair_R + r]; sh_lad[rank & 1] = tid; }
        }
        __syncthreads();
    }
    const int ca = a.pairs ? a.pairs[2 * p] : sh_pair[0], cb = a.pairs ? a.pairs[2 * p + 1] : sh_pair[1];
    int8_t *sa = a.spins + (size_t)ca * n_pad, *sb = a.spins + (size_t)cb * n_pad;
    int32_t *lab = lab_g + (size_t)p * n;
    if (tid == 0) { nroots = 0; sh_size = 0; sh_dE[0] = 0; sh_dE[1] = 0; sh_root = -1; }
    for (int k = tid; k < n; k += nt) lab[k] = ((int)sa[k] * (int)sb[k] == -1) ? k : INT_MAX;
    __syncthreads();
    for (int k = tid; k < n; k += nt) {
        if (lab[k] == INT_MAX) continue;
        const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
        for (int e = rs; e < re; ++e) {
            const EdgeQ t = a.g.edge32[e];
            const int j = t.col;
            if ((t.q == 0 && a.g.val64[e] == 0.0) || j >= k || lab[j] == INT_MAX) continue;       // every edge from its larger end
            int ra = icm_find_halving(lab, k), rb = icm_find_halving(lab, j);
            while (ra != rb) {
                if (ra < rb) { const int x = ra; ra = rb; rb = x; }
                const int old = atomicCAS(&lab[ra], ra, rb);
                if (old == ra) break;
                ra = icm_find_halving(lab, old);
                rb = icm_find_halving(lab, rb);
            }
        }
    }
    __syncthreads();
    int cnt = 0;
    for (int k = tid; k < n; k += nt) if (lab[k] != INT_MAX) cnt += (icm_find_halving(lab, k) == k);
    if (cnt) atomicAdd(&nroots, cnt);
    __syncthreads();
    for (int k = tid; k < n; k += nt) if (lab[k] != INT_MAX) lab[k] = icm_find_halving(lab, k);
    __syncthreads();
    const int ncomp = nroots;
    if (ncomp <= 0) {
        if (tid == 0) { a.info[2 * p] = ncomp; a.info[2 * p + 1] = 0; }
        return;
    }
    uint32_t ida = (uint32_t)(a.chain_base + ca), idb = (uint32_t)(a.chain_base + cb);
    if (a.rng_stride && !a.pairs) {
        const int rl = p / (a.pair_K / 2);
        ida = (uint32_t)(sh_lad[0] * a.rng_stride + a.rng_base + rl);
        idb = (uint32_t)(sh_lad[1] * a.rng_stride + a.rng_base + rl);
    }
    const uint32_t r = philox4x32_10(ida, a.round, idb, NLMC_TAG_ICM, a.seed_lo, a.seed_hi).x;
    const int pick = (int)(((unsigned long long)r * (unsigned long long)ncomp) >> 32);
    const int chunk = (n + nt - 1) / nt;
    const int b = min(tid * chunk, n), e = min(b + chunk, n);
    int mine = 0;
    for (int k = b; k < e; ++k) mine += (lab[k] == k);
    const int lane = tid & 63, wv = tid >> 6;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) sh_scan[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += sh_scan[w];
    const int before = base + incl - mine;
    if (pick >= before && pick < before + mine) {
        int seen = before;
        for (int k = b; k < e; ++k)
            if (lab[k] == k) { if (seen == pick) { sh_root = k; break; } ++seen; }
    }
    __syncthreads();
    const int root = sh_root;
    int csz = 0;
    for (int k = tid; k < n; k += nt) csz += (lab[k] == root);
    if (csz) atomicAdd(&sh_size, csz);
    __syncthreads();
    const int size = sh_size;
    long long dEa = 0, dEb = 0;
    if (a.katz && size > n / 2) {
        for (int k = tid; k < n; k += nt) { const int8_t v = sa[k]; dEa += 2ll * (long long)a.g.hq[k] * (long long)v; sa[k] = (int8_t)(-v); }
    } else {
        // (a cluster member reads its own two spins and spins OUTSIDE the cluster only; nothing outside is written)
        for (int k = tid; k < n; k += nt) {
            if (lab[k] != root) continue;
            const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
            long long fa = a.g.hq[k], fb = fa;
            for (int q = rs; q < re; ++q) {
                const EdgeQ t = a.g.edge32[q];
                if (lab[t.col] == root) continue;
                fa += (long long)t.q * (long long)sa[t.col];
                fb += (long long)t.q * (long long)sb[t.col];
            }
            const int8_t va = sa[k], vb = sb[k];
            dEa += 2ll * (long long)va * fa;
            dEb += 2ll * (long long)vb * fb;
            sa[k] = vb;
            sb[k] = va;
        }
    }
    dEa = wave_sum_i64(dEa);
    dEb = wave_sum_i64(dEb);
    if (lane == 0) {
        if (dEa) atomicAdd(reinterpret_cast<unsigned long long *>(&sh_dE[0]), (unsigned long long)dEa);
        if (dEb) atomicAdd(reinterpret_cast<unsigned long long *>(&sh_dE[1]), (unsigned long long)dEb);
    }
    __syncthreads();
    if (tid == 0) {
        const long long ea = a.efix[ca] + sh_dE[0] * (1ll << a.eshift), eb = a.efix[cb] + sh_dE[1] * (1ll << a.eshift);
        a.efix[ca] = ea;
        a.efix[cb] = eb;
        if (a.energy_sink) {
            const double inv = __longlong_as_double((long long)(1023 - a.escale) << 52);
            a.energy_sink[ca] = (double)ea * inv;
            a.energy_sink[cb] = (double)eb * inv;
        }
        a.info[2 * p] = ncomp;
        a.info[2 * p + 1] = size;
    }
}
