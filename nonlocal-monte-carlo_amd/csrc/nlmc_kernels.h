// nlmc_kernels.h -- HIP kernels of the sweep path for gfx950.
//
//   k_levelize : one workgroup per sweep-order.  Turns a visiting order (Philox keys, or ranks of a host-drawn
//                permutation) into a LEVEL SCHEDULE: level(k) = 1 + max level(j) over neighbours j visited before k.
//                Spins of one level are mutually non-adjacent, so updating levels in ascending order, each level
//                in parallel, is bit-identical to the reference's sequential pass (NMC/nmc.py:71-87).
//   k_sweep_*  : one workgroup per chain; the chain's spins (and phase flags, and one sweep's uniforms) live in LDS
//                for the whole launch; loops sweeps x levels with one s_barrier per level; CSR rows come from L2
//                and are prefetched one level ahead (they do not depend on spin values); energy is tracked
//                incrementally in 64-bit fixed point (associative -> reduction order cannot change a bit).
//   k_energy   : E = -(m^T J m/2 + m^T h) in fp64, one workgroup per configuration.
#pragma once
#include "nlmc_device.h"

// 8-byte packed CSR entry of the "f32" throughput path (one dwordx2 load): column and the coupling in 24-bit fixed
// point, q = rint(J 2^qs) (nlmc_create).  The field of a spin is then an exact int32, independent of summation order.
struct EdgeQ { int32_t col; int32_t q; };

struct CsrDev {
    int n, n_pad;
    const int32_t *rowptr;
    const int32_t *col;      // [nnz]
    const double *val64;     // [nnz]
    const EdgeQ *edge32;     // [nnz]
    const double *h64;       // [n]
    const int32_t *hq;       // [n]  rint(h 2^qs)
};

// ------------------------------------------------------------------------------------------------------
// level schedule
// ------------------------------------------------------------------------------------------------------
struct LevelizeArgs {
    CsrDev g;
    int n_orders;
    // key source: keys_in != nullptr -> ranks [n_orders][n] (stream mode); else Philox ORDER keys
    const uint32_t *keys_in;
    uint32_t seed_lo, seed_hi, sweep0;
    int per_chain;        // philox: order id o = c * n_sweeps + t  (group = chain_base + c + 1) else o = t (group 0)
    int n_sweeps;
    int chain_base;
    int level_cap;        // levels wider than this are split (any subset of an independent set is independent)
    int two_sided;        // LDS holds a second cursor array: rows longer than 8 entries are placed at the FRONT of their
                          // level so that only the first wave(s) of a level pay for the second half of the row window
    int2 *ord2;           // [n_orders][n]  { k | deg << 16, row start }
    int32_t *lvl_off;     // [n_orders][n+1]
    int32_t *nlev;        // [n_orders]
    int32_t *hi_max;      // [n_orders] largest number of long rows (> 8 entries) at the front of any level
    // optional packed per-order schedule in ELL form (slot-major, position-minor) so that the sweep kernel's
    // loads are fully coalesced: lane i of a level reads slot q at [(o*8+q)*n + i]
    EdgeQ *ell32;         // [n_orders][8][n][2]  first 16 entries of row k(i) as 8 planes of 2 entries (16 B),
                          //                      zero-padded (col 0, q 0)
    int2 *head32;         // [n_orders][n]     { k | deg << 16, hq_k }
    int32_t *ellc64;      // [n_orders][NLMC_ELL_W / 4][n] planes of 4 columns (16 B)
    double *ellv64;       // [n_orders][NLMC_ELL_W / 2][n] planes of 2 couplings (16 B)
    double *headh64;      // [n_orders][n]
    int pack16;           // fp64 path, couplings exactly Jq 2^-qs with |Jq| < 2^15: 16-bit columns and Jq, 4 planes in ellc64
};
#define NLMC_ELL_W 16         // packed row window, fp64 path (second half read by the waves that may hold a long row only)
#ifndef NLMC_ELL_W32
#define NLMC_ELL_W32 16       // packed row window, fp32 path: covers every row of a degree-6 random graph (max ~16)
#endif

__device__ __forceinline__ bool precedes(uint32_t kj, int j, uint32_t kk, int k) { return kj < kk || (kj == kk && j < k); }

// LDS: keys u32[n+2] (reused as level histogram / cursors), lvl u16[n]
__global__ void k_levelize(LevelizeArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = a.g.n;
    uint32_t *key = reinterpret_cast<uint32_t *>(lds_raw);
    uint16_t *lvl = reinterpret_cast<uint16_t *>(lds_raw + (size_t)(n + 2) * 4);
    __shared__ int sh_scan[64];
    __shared__ int sh_max;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int o = blockIdx.x;

    if (a.keys_in) {
        const uint32_t *src = a.keys_in + (size_t)o * n;
        for (int k = tid; k < n; k += nt) { key[k] = src[k]; lvl[k] = 0; }
    } else {
        const uint32_t t = a.sweep0 + (uint32_t)(a.per_chain ? (o % a.n_sweeps) : o);
        const uint32_t grp = a.per_chain ? (uint32_t)(a.chain_base + o / a.n_sweeps + 1) : 0u;
        for (int k = tid; k < n; k += nt) {
            key[k] = philox4x32_10((uint32_t)k, t, grp, NLMC_TAG_ORDER, a.seed_lo, a.seed_hi).x;
            lvl[k] = 0;
        }
    }
    if (tid == 0) sh_max = 0;
    __syncthreads();

    // chaotic relaxation to the unique fixed point (levels only ever increase towards their final value)
    for (int it = 0; it <= n; ++it) {
        int changed = 0;
        for (int k = tid; k < n; k += nt) {
            const uint32_t kk = key[k];
            const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
            int m = 0;
            for (int e = rs; e < re; ++e) {
                const int j = a.g.col[e];
                if (j != k && precedes(key[j], j, kk, k)) m = max(m, (int)lvl[j] + 1);
            }
            if (m != (int)lvl[k]) { lvl[k] = (uint16_t)m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }

    // number of levels
    int lmax = 0;
    for (int k = tid; k < n; k += nt) lmax = max(lmax, (int)lvl[k]);
    atomicMax(&sh_max, lmax);
    __syncthreads();
    const int nl = (n > 0) ? sh_max + 1 : 0;

    // histogram (keys are dead now: reuse their LDS as cnt[nl+1])
    uint32_t *cnt = key;
    for (int l = tid; l <= nl; l += nt) cnt[l] = 0;
    __syncthreads();
    for (int k = tid; k < n; k += nt) atomicAdd(&cnt[lvl[k]], 1u);
    __syncthreads();

    // exclusive scan of cnt[0..nl) : chunk per thread + scan of chunk sums over waves
    const int chunk = (nl + nt - 1) / nt;
    const int b = min(tid * chunk, nl), e = min(b + chunk, nl);
    int s = 0;
    for (int l = b; l < e; ++l) s += (int)cnt[l];
    // inclusive scan of s across the block
    const int lane = tid & 63, wv = tid >> 6;
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) sh_scan[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += sh_scan[w];
    int run = base + incl - s;
    int32_t *off = a.lvl_off + (size_t)o * (n + 1);
    // cursors for the placement; the published offsets additionally split levels wider than level_cap, so that
    // the sweep kernel sees at most one spin per thread and level (sub-levels run back to back, barrier between)
    for (int l = b; l < e; ++l) { const int c = (int)cnt[l]; cnt[l] = (uint32_t)run; run += c; }
    __syncthreads();
    if (tid == 0) {
        const int cap = a.level_cap > 0 ? a.level_cap : n;
        int m = 0;
        for (int l = 0; l < nl; ++l) {
            const int lo = (int)cnt[l], hi = (l + 1 < nl) ? (int)cnt[l + 1] : n;
            for (int p = lo; p < hi; p += cap) off[m++] = p;
        }
        off[m] = n;
        a.nlev[o] = m;      // m <= n because every (sub-)level holds at least one spin
    }
    __syncthreads();

    // placement (intra-level order is irrelevant: same-level spins are independent)
    int2 *ord = a.ord2 + (size_t)o * n;
    uint32_t *back = reinterpret_cast<uint32_t *>(lds_raw + (size_t)(n + 2) * 4 + (((size_t)n * 2 + 3) / 4) * 4);
    __shared__ int sh_himax;
    if (tid == 0) sh_himax = 0;
    if (a.two_sided) {
        // long rows per level (they go to the FRONT of their level): the sweep kernel lets only the waves that can
        // hold one issue the loads of the second half of the row window
        for (int l = tid; l < nl; l += nt) back[l] = 0u;
        __syncthreads();
        for (int k = tid; k < n; k += nt)
            if (a.g.rowptr[k + 1] - a.g.rowptr[k] > 8) atomicAdd(&back[lvl[k]], 1u);
        __syncthreads();
        int hm = 0;
        for (int l = tid; l < nl; l += nt) hm = max(hm, (int)back[l]);
        if (hm) atomicMax(&sh_himax, hm);
        __syncthreads();
        for (int l = tid; l < nl; l += nt) back[l] = (l + 1 < nl) ? cnt[l + 1] : (uint32_t)n;   // end of level l
        __syncthreads();
    }
    if (tid == 0) a.hi_max[o] = a.two_sided ? sh_himax : n;
    for (int k = tid; k < n; k += nt) {
        const int rs = a.g.rowptr[k], deg = a.g.rowptr[k + 1] - rs;
        uint32_t pos;
        if (a.two_sided && deg <= 8) pos = atomicSub(&back[lvl[k]], 1u) - 1u;     // short rows fill from the back
        else pos = atomicAdd(&cnt[lvl[k]], 1u);                                  // long rows from the front
        const int kd = k | (deg << 16);
        ord[pos] = make_int2(kd, rs);
        if (a.ell32) {
            a.head32[(size_t)o * n + pos] = make_int2(kd, a.g.hq[k]);
#pragma unroll
            for (int q = 0; q < NLMC_ELL_W32; q += 2) {
                EdgeQ e0{0, 0}, e1{0, 0};
                if (q < deg) e0 = a.g.edge32[rs + q];
                if (q + 1 < deg) e1 = a.g.edge32[rs + q + 1];
                int4 pk = make_int4(e0.col, e0.q, e1.col, e1.q);
                reinterpret_cast<int4 *>(a.ell32)[((size_t)o * (NLMC_ELL_W32 / 2) + q / 2) * n + pos] = pk;
            }
        }
        if (a.ellc64 && a.pack16) {
            // packed fp64 window (Pf<double, true>): plane 0 = columns 0-7, plane 1 = Jq 0-7, planes 2 / 3 = entries 8-15, 16 bits
            // each -- 48 instead of 112 bytes for a row of <= 8 entries.  Every chain reads the whole schedule: at 112 bytes
            // the 32 CUs of an XCD pulled 1.8 MB per level through its L2, which is what a level then waited for.
            a.headh64[(size_t)o * n + pos] = a.g.h64[k];
            unsigned cw[NLMC_ELL_W / 2], qw[NLMC_ELL_W / 2];
#pragma unroll
            for (int q = 0; q < NLMC_ELL_W; q += 2) {
                EdgeQ e0{0, 0}, e1{0, 0};
                if (q < deg) e0 = a.g.edge32[rs + q];
                if (q + 1 < deg) e1 = a.g.edge32[rs + q + 1];
                cw[q / 2] = ((unsigned)e0.col & 0xFFFFu) | ((unsigned)e1.col << 16);
                qw[q / 2] = ((unsigned)e0.q & 0xFFFFu) | ((unsigned)e1.q << 16);
            }
            uint4 *dst = reinterpret_cast<uint4 *>(a.ellc64);
#pragma unroll
            for (int hf = 0; hf < NLMC_ELL_W / 8; ++hf) {
                dst[((size_t)o * (NLMC_ELL_W / 4) + 2 * hf) * n + pos] = make_uint4(cw[4 * hf], cw[4 * hf + 1], cw[4 * hf + 2], cw[4 * hf + 3]);
                dst[((size_t)o * (NLMC_ELL_W / 4) + 2 * hf + 1) * n + pos] = make_uint4(qw[4 * hf], qw[4 * hf + 1], qw[4 * hf + 2], qw[4 * hf + 3]);
            }
        } else if (a.ellc64) {
            // 16-byte planes like the fixed-point window: columns 4 to a plane, couplings 2 to a plane -- the sweep kernel reads an
            // item with 2 + 4 wide loads (it was 8 + 8 narrow ones: the fp64 kernel was bound by its vector-memory instructions)
            a.headh64[(size_t)o * n + pos] = a.g.h64[k];
            static_assert(NLMC_ELL_W % 4 == 0, "columns 4 to a plane, couplings 2 to a plane");
            int cj[NLMC_ELL_W]; double vj[NLMC_ELL_W];
#pragma unroll
            for (int q = 0; q < NLMC_ELL_W; ++q) {
                cj[q] = 0; vj[q] = 0.0;
                if (q < deg) { cj[q] = a.g.col[rs + q]; vj[q] = a.g.val64[rs + q]; }
            }
#pragma unroll
            for (int pl = 0; pl < NLMC_ELL_W / 4; ++pl)
                reinterpret_cast<int4 *>(a.ellc64)[((size_t)o * (NLMC_ELL_W / 4) + pl) * n + pos] = make_int4(cj[4 * pl], cj[4 * pl + 1], cj[4 * pl + 2], cj[4 * pl + 3]);
#pragma unroll
            for (int pl = 0; pl < NLMC_ELL_W / 2; ++pl)
                reinterpret_cast<double2 *>(a.ellv64)[((size_t)o * (NLMC_ELL_W / 2) + pl) * n + pos] = make_double2(vj[2 * pl], vj[2 * pl + 1]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// fused-window level schedule: ONE level list for T consecutive sweeps (see k_sweep_fused)
// ------------------------------------------------------------------------------------------------------
#define NLMC_LCAP 1024          // level offsets of one schedule kept in LDS by the sweep kernels
#define NLMC_FUSED_TMAX 64
#define NLMC_FZ_W 8              // row entries per schedule position of a fused plan; longer rows take two positions
#define NLMC_FMT_WIDE 0
#define NLMC_FMT_COMPACT 1
#define NLMC_FMT_ADDR 2
struct FusedLevelizeArgs {
    CsrDev g;
    int T;                    // sweeps per window
    uint32_t seed_lo, seed_hi, sweep0;     // window w covers sweeps sweep0 + w T ... + T - 1 (shared order, group 0)
    int level_cap;            // 64 x worker waves of k_sweep_fused
    int pstride;              // schedule positions reserved per window (multiple of 64, >= T n + 64 NLMC_LCAP)
    int tab_words;            // 4-byte words per threshold table of k_sweep_fused (its LDS stride / 4)
    int k_dummy;              // LDS address of the scratch spin that dummy (padding) items update: n_pad
    int fmt;                  // entry format of the planes: NLMC_FMT_WIDE / _COMPACT / _SIGN (see FusedItem)
    int k_zero;               // address format: LDS address of a byte that is always 0 (padding entries point at it)
    int neg_off;              // address format: LDS offset of the negated copy of the spins
    int bank_aware;           // order every item's row entries so that the 32 lanes of a half-wave gather from different LDS banks
    const uint4 *adj;         // [n][2]: the first 16 neighbours of every spin as 16-bit indices (k_fused_adjacency)
    uint16_t *glv;            // scratch [n_windows][T][n]: level of update (t, k), 1-based
    uint32_t *perm;           // scratch [n_windows][pstride]: item id k | t << 16 at its position, ~0 = padding
    long long *stats;         // diagnostic (NLMC_FZ_STATS): [n_windows][8] cycles keys / init / passes, pass count, place 1 / 2
    int2 *head;               // [n_windows][pstride]   { k | LONG << 14 | threshold word << 16, hq_k }
    EdgeQ *ell;               // [n_windows][NLMC_FZ_W / 2][pstride][2]   row window planes (8 entries per position), position-minor
    int32_t *loff;            // [n_windows][NLMC_LCAP + 1] published level offsets, in chunks of 64 positions
    int32_t *nlev;            // [n_windows] published levels; 0 = deeper than NLMC_LCAP - 1 (caller falls back)
    int32_t *hi_max;          // [n_windows] leading chunks of a level that may hold PAIRS (rows longer than 8 entries)
    int32_t *send;            // [n_windows][T] published index of the last level that holds an item of sweep t
    int32_t *npos;            // [n_windows] schedule positions in use (multiple of 64)
};

// 16-bit neighbour lists, 32 B per spin (absent slots hold the spin's own index, which every consumer skips): one
// pair of 16-byte loads instead of a chain of dependent 4-byte CSR reads per spin in k_levelize_fused.
#define NLMC_FZ_ADJ 16
__global__ void k_fused_adjacency(int n, const int32_t *rowptr, const int32_t *col, uint16_t *adj)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int rs = rowptr[k], deg = rowptr[k + 1] - rs;
    for (int q = 0; q < NLMC_FZ_ADJ; ++q) adj[(size_t)k * NLMC_FZ_ADJ + q] = (uint16_t)(q < deg ? col[rs + q] : k);
}

// Level of update (t, k) = 1 + max over: its own update of sweep t-1; every neighbour's update of sweep t that comes
// earlier in the order of sweep t; every other neighbour's update of sweep t-1 (its value is read, and it must not be
// overwritten before the neighbours of sweep t-1 that read the old one are done -- they precede by the same rule);
// and the end of sweep t-2 (at most two sweeps live per level: three uniform tables suffice).
// Per sweep: (1) every spin gets base = the part of the maximum known up front and cnt = number of neighbours that
// precede it; (2) topological passes: a spin with cnt == 0 is final at level m + 1 and pushes that level to its later
// neighbours (LDS atomicMax on m, then decrement of their cnt) -- every edge is handled once per sweep, a pass costs a
// look at the thread's own <= NLMC_FZ_SPT counters plus the pushes of the spins that became final.
// LDS: key u32[n] | m u32[n] | g u16[n] | cnt u8[n] | queue u16[n] | hist u32[LCAP + 2] | histL u32[LCAP + 2]
#define NLMC_FZ_SPT 11         // spins per thread: n <= 11 * 1024
__global__ __launch_bounds__(1024) void k_levelize_fused(FusedLevelizeArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = a.g.n, T = a.T, w = blockIdx.x;
    const int n4 = (n + 3) & ~3;
    uint32_t *key = reinterpret_cast<uint32_t *>(lds_raw);
    uint32_t *mx = key + n;
    uint16_t *g = reinterpret_cast<uint16_t *>(mx + n);
    uint8_t *cnt8 = reinterpret_cast<uint8_t *>(g + n4);
    uint32_t *cnt32 = reinterpret_cast<uint32_t *>(cnt8);
    uint16_t *queue = reinterpret_cast<uint16_t *>(cnt8 + n4);
    uint32_t *hist = reinterpret_cast<uint32_t *>(queue + n4);
    uint32_t *histL = hist + NLMC_LCAP + 2;
    __shared__ int sh_lmax[NLMC_FUSED_TMAX];
    __shared__ int sh_max, sh_fail, sh_nlev, sh_qn, sh_npos;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int k = tid; k < n; k += nt) g[k] = 0;
    for (int l = tid; l < 2 * (NLMC_LCAP + 2); l += nt) hist[l] = 0u;      // hist and histL are adjacent
    if (tid == 0) { sh_fail = 0; sh_qn = 0; }
    __syncthreads();
    uint16_t *glv = a.glv + (size_t)w * T * n;

    long long st[6] = {0, 0, 0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
        const uint32_t tt = a.sweep0 + (uint32_t)(w * T + t);
        long long c0 = (long long)__builtin_readcyclecounter();
        for (int k = tid; k < n; k += nt) key[k] = philox4x32_10((uint32_t)k, tt, 0u, NLMC_TAG_ORDER, a.seed_lo, a.seed_hi).x;
        if (tid == 0) sh_max = 0;
        __syncthreads();
        long long c1 = (long long)__builtin_readcyclecounter();
        st[0] += c1 - c0;
        const int floor_lv = t >= 2 ? sh_lmax[t - 2] : 0;
        for (int i = 0; i < NLMC_FZ_SPT; ++i) {        // this thread's spins: k = tid + i nt
            const int k = tid + i * nt;
            if (k < n) {
                const int re = a.g.rowptr[k + 1], rs = a.g.rowptr[k];
                const uint32_t kk = key[k];
                int base = max((int)g[k], floor_lv), c = 0;
                const uint4 a0 = a.adj[2 * k], a1 = a.adj[2 * k + 1];
                const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int q = 0; q < NLMC_FZ_ADJ; ++q) {
                    const int j = (int)((aw[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu);
                    if (j != k) { if (precedes(key[j], j, kk, k)) ++c; else base = max(base, (int)g[j]); }
                }
                for (int e = rs + NLMC_FZ_ADJ; e < re; ++e) {      // rows longer than 16 entries
                    const int j = a.g.col[e];
                    if (j != k) { if (precedes(key[j], j, kk, k)) ++c; else base = max(base, (int)g[j]); }
                }
                mx[k] = (uint32_t)base;
                cnt8[k] = (uint8_t)min(c, 255);
                if (c > 255) sh_fail = 1;          // more than 255 preceding neighbours: no fused schedule
                if (c == 0) queue[atomicAdd(&sh_qn, 1)] = (uint16_t)k;     // no earlier neighbour: ready at once
            }
        }
        __syncthreads();                           // every old level has been read: g is rewritten below
        c0 = (long long)__builtin_readcyclecounter();
        st[1] += c0 - c1;
        int lmax = 0;
        // Topological passes over a work queue that every spin enters exactly once per sweep: the thread whose decrement
        // takes a spin's counter of unfinished earlier neighbours to zero appends it (its level is final then: every
        // earlier neighbour has pushed its level with an atomicMax BEFORE its decrement, and LDS operations are served in
        // order).  A pass processes the entries appended during the previous one, one spin per lane; nobody scans
        // counters.
        int q_lo = 0;
        for (int pass = 0; pass <= n; ++pass) {
            const int q_hi = sh_qn;                // stable: appended to only between the two barriers below
            __syncthreads();
            if (q_lo == q_hi) break;               // nothing became ready: done (a DAG always has a ready node otherwise)
            for (int idx = q_lo + tid; idx < q_hi; idx += nt) {
                const int k = (int)queue[idx];
                const int re = a.g.rowptr[k + 1], rs = a.g.rowptr[k];
                const uint4 a0 = a.adj[2 * k], a1 = a.adj[2 * k + 1];
                const int lv = min((int)mx[k] + 1, 65535);
                g[k] = (uint16_t)lv;
                glv[(size_t)t * n + k] = (uint16_t)lv;
                lmax = max(lmax, lv);
                if (lv <= NLMC_LCAP) {
                    atomicAdd(&hist[lv], 1u);
                    if (re - rs > 8) atomicAdd(&histL[lv], 1u);
                }
                const uint32_t kk = key[k];
                const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                uint32_t kj[NLMC_FZ_ADJ];
#pragma unroll
                for (int q = 0; q < NLMC_FZ_ADJ; ++q) kj[q] = key[(aw[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu];   // one batch
#pragma unroll
                for (int q = 0; q < NLMC_FZ_ADJ; ++q) {
                    const int j = (int)((aw[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu);
                    if (j != k && precedes(kk, k, kj[q], j)) {                // k comes before j: j waits for k
                        atomicMax(&mx[j], (uint32_t)lv);
                        const uint32_t old = atomicSub(&cnt32[j >> 2], 1u << ((j & 3) * 8));      // after the max
                        if (((old >> ((j & 3) * 8)) & 0xFFu) == 1u) queue[atomicAdd(&sh_qn, 1)] = (uint16_t)j;
                    }
                }
                for (int e = rs + NLMC_FZ_ADJ; e < re; ++e) {
                    const int j = a.g.col[e];
                    if (j != k && precedes(kk, k, key[j], j)) {
                        atomicMax(&mx[j], (uint32_t)lv);
                        const uint32_t old = atomicSub(&cnt32[j >> 2], 1u << ((j & 3) * 8));
                        if (((old >> ((j & 3) * 8)) & 0xFFu) == 1u) queue[atomicAdd(&sh_qn, 1)] = (uint16_t)j;
                    }
                }
            }
            st[3] += 1;
            q_lo = q_hi;
            __syncthreads();
        }
        __syncthreads();
        if (tid == 0) sh_qn = 0;                   // (read as q_hi by everybody before the barrier above)
        st[2] += (long long)__builtin_readcyclecounter() - c0;
        atomicMax(&sh_max, lmax);
        __syncthreads();
        if (tid == 0) { sh_lmax[t] = sh_max; if (sh_max > NLMC_LCAP) sh_fail = 1; }
        __syncthreads();
    }

    // publish offsets.  Positions are handed out in CHUNKS of 64 (one wave's items of one level): every level starts on
    // a chunk boundary and its last chunk is padded with dummy items, so that the sweep kernel never deals with a
    // partly filled wave (no per-lane validity: a wave either holds a chunk of a level or it does not).  A row longer
    // than NLMC_FZ_W entries takes TWO neighbouring positions (an even / odd lane pair: entries 0-7 and 8-15, the two
    // partial fields are added with one cross-lane move), so that every wave of a level does the same amount of work;
    // the pairs come first in their level.  Levels wider than level_cap positions (a multiple of 64) are split.
    // off[] counts chunks; hist[lv] becomes the level's first position, histL[lv] (was: its number of long rows) the
    // position behind its last real item.
    if (tid == 0) {
        const int L = sh_lmax[T - 1];
        int32_t *off = a.loff + (size_t)w * (NLMC_LCAP + 1);
        int m = 0, run = 0, t_next = 0, himax = 0;
        if (!sh_fail) {
            for (int lv = 1; lv <= L; ++lv) {
                const int cl = (int)histL[lv], width = (int)hist[lv] + cl;
                hist[lv] = (uint32_t)run;
                himax = max(himax, (2 * cl + 63) >> 6);
                histL[lv] = (uint32_t)(run + width);
                for (int p = 0; p < width; p += a.level_cap) { if (m < NLMC_LCAP) { off[m] = (run + p) >> 6; mx[m] = (uint32_t)((run + p) >> 6); } ++m; }
                run += (width + 63) & ~63;
                while (t_next < T && sh_lmax[t_next] == lv) a.send[(size_t)w * T + t_next++] = m - 1;
            }
            if (m >= NLMC_LCAP || run > a.pstride) sh_fail = 1; else { off[m] = run >> 6; mx[m] = (uint32_t)(run >> 6); }
        }
        sh_nlev = sh_fail ? 0 : m;
        sh_npos = run;
        a.nlev[w] = sh_nlev;
        a.npos[w] = run;
        a.hi_max[w] = min(himax, a.level_cap >> 6);
    }
    __syncthreads();
    if (sh_nlev == 0) return;

    // placement: the pairs (rows longer than 8 entries) from the front of their level, two positions each, the others
    // from the back of its real items.  Two steps so that the 40-72 B per position are written with coalesced stores:
    // (1) scatter the 4-byte item ids to their positions, (2) position-major: lane p gathers its half of row k(p) (CSR is
    // cache resident) and writes head[p] and the planes [q][p] next to its neighbours' -- a direct scatter of 16-byte
    // pieces ran at a tenth of the bandwidth.
    const int npos = sh_npos;
    const size_t PS = (size_t)a.pstride;
    long long p0 = (long long)__builtin_readcyclecounter();
    uint32_t *perm = a.perm + (size_t)w * PS;
    for (int pos = tid; pos < npos; pos += nt) perm[pos] = 0xFFFFFFFFu;       // padding = dummy items
    __threadfence_block();
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        for (int k = tid; k < n; k += nt) {
            const int lv = (int)glv[(size_t)t * n + k];
            const int deg = a.g.rowptr[k + 1] - a.g.rowptr[k];
            const uint32_t id = (uint32_t)k | ((uint32_t)t << 16);
            if (deg > NLMC_FZ_W) {
                const uint32_t pos = atomicAdd(&hist[lv], 2u);                 // even: a level starts on a multiple of 64
                perm[pos] = id;
                perm[pos + 1] = id | 0x80000000u;                              // second half of the row
            } else {
                perm[atomicSub(&histL[lv], 1u) - 1u] = id;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    long long p1 = (long long)__builtin_readcyclecounter();
    st[4] = p1 - p0;
    // head.x = k | LONG << 14 | PAIR << 15 | thr << 16: k = LDS address of the spin, LONG = row longer than two windows
    // (the rest is read from the CSR arrays by the first lane of the pair), PAIR = this lane holds half of a row, thr =
    // word index of the update's threshold in the three LDS tables (slot t mod 3); head.y = hq_k (first half only).
    // A dummy item updates the scratch spin behind the real ones from an all-zero row: harmless by construction.
    int2 *head = a.head + (size_t)w * PS;
    int4 *ell = reinterpret_cast<int4 *>(a.ell) + (size_t)w * (NLMC_FZ_W / 2) * PS;
    // Bank-aware entry order (round 4).  The sweep kernel gathers entry q of all 64 items of a chunk with ONE ds_read_u8; the LDS
    // serves it per half-wave of 32 lanes over 32 banks of 4 bytes (scripts/probes/lds_conflict_probe.hip: same bank + different dword
    // inside a half = one more pass: 205 / 280 / 486 ns per round of 8 gathers at 1 / 2 / 4 lanes per bank; random addresses 303).  The
    // field is a sum, so a row's entries may stand in ANY order: slot by slot, the lanes of a half-wave claim banks (atomicOr on a
    // 32-bit mask per half and slot; up to three candidates per lane and slot, then whatever is left).  Early slots -- many
    // candidates per lane -- come out conflict-free, the last real entries of a row take what remains.
    // One claim word per (wave, half, bank): a lane claims bank b of the slot in hand with an exchange on word b -- different banks
    // are different LDS banks, so the 32 lanes of a half proceed side by side and only true rivals queue (one mask word per half
    // and slot, OR-ed by all 32 lanes, serialised every claim: 1.2 of the 1.85 M cycles of this loop at N = 10^4).  The word holds
    // the number of the (iteration, slot) it was last claimed in, so nothing has to be cleared between slots.
    __shared__ unsigned sh_claim[16][2][32];
    sh_claim[tid >> 6][(tid >> 5) & 1][tid & 31] = 0u;
    unsigned claim_gen = 0u;
    // Software pipeline over a thread's positions: item id -> row bounds -> row entries are three dependent reads (L2 hits, ~1 us
    // each); the id of the position after next and the row bounds of the next one are fetched while the current one is packed, so
    // that an iteration waits for its row entries only.
    uint32_t it_a = tid < npos ? perm[tid] : 0xFFFFFFFFu, it_b = tid + nt < npos ? perm[tid + nt] : 0xFFFFFFFFu;
    int rs_a = 0, re_a = 0;
    if (it_a != 0xFFFFFFFFu) { rs_a = a.g.rowptr[it_a & 0xFFFFu]; re_a = a.g.rowptr[(it_a & 0xFFFFu) + 1]; }
    for (int pos = tid; pos < npos; pos += nt) {
        const uint32_t it = it_a;
        const int rs_cur = rs_a, deg_cur = re_a - rs_a;
        claim_gen += NLMC_FZ_W;
        it_a = it_b;
        if (it_a != 0xFFFFFFFFu) { rs_a = a.g.rowptr[it_a & 0xFFFFu]; re_a = a.g.rowptr[(it_a & 0xFFFFu) + 1]; }
        it_b = pos + 2 * nt < npos ? perm[pos + 2 * nt] : 0xFFFFFFFFu;
        const uint32_t dpack = (uint32_t)a.k_dummy << 16;
        if (it == 0xFFFFFFFFu) {
            // (threshold word 3 tab_words: behind the three tables / snapshot slots, so that a dummy's threshold read and
            // snapshot write touch nothing that belongs to a spin)
            head[pos] = make_int2(a.k_dummy | ((3 * a.tab_words) << 16), 0);
            if (a.fmt == NLMC_FMT_ADDR) {
                const int z2 = a.k_zero | (a.k_zero << 16);
                ell[pos] = make_int4(z2, z2, z2, z2);
            } else if (a.fmt == NLMC_FMT_COMPACT) {
#pragma unroll
                for (int q = 0; q < NLMC_FZ_W / 4; ++q) ell[(size_t)q * PS + pos] = make_int4((int)dpack, (int)dpack, (int)dpack, (int)dpack);
            } else {
#pragma unroll
                for (int q = 0; q < NLMC_FZ_W; q += 2) ell[(size_t)(q / 2) * PS + pos] = make_int4(a.k_dummy, 0, a.k_dummy, 0);
            }
            continue;
        }
        const int second = (int)(it >> 31);
        const int k = (int)(it & 0xFFFFu), t = (int)((it >> 16) & 0x7FFFu);
        const int rs = rs_cur, deg = deg_cur;
        const int thr = ((t % 3) * a.tab_words + k) << 16;
        const int e0 = second ? NLMC_FZ_W : 0, left = deg - e0;                 // this lane's entries: e0 .. e0 + 7
        EdgeQ ed[NLMC_FZ_W];
#pragma unroll
        for (int q = 0; q < NLMC_FZ_W; ++q) ed[q] = a.g.edge32[rs + e0 + q];    // unconditional (the array is padded by a
                                                                                // full window): independent loads
        const int hy = second ? 0 : a.g.hq[k];
        if (a.bank_aware) {
            const int nreal = min(max(left, 0), NLMC_FZ_W);
            unsigned long long banks = 0ull;                   // 5 bits per entry
            unsigned long long adr_lo = 0ull, adr_hi = 0ull;   // address format: the 16-bit LDS addresses of entries 0-3 | 4-7
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; ++q) {
                const unsigned ad = (unsigned)ed[q].col + ((a.fmt == NLMC_FMT_ADDR && ed[q].q < 0) ? (unsigned)a.neg_off : 0u);
                banks |= (unsigned long long)((ad >> 2) & 31u) << (5 * q);
                if (q < 4) adr_lo |= (unsigned long long)(ad & 0xFFFFu) << (16 * q);
                else adr_hi |= (unsigned long long)(ad & 0xFFFFu) << (16 * (q - 4));
            }
            unsigned *claim = &sh_claim[tid >> 6][(tid >> 5) & 1][0];          // (npos is a multiple of 64: whole waves iterate)
            unsigned rem = (1u << nreal) - 1u, ordpk = 0u;     // entries not placed yet; 4 bits per slot: entry index, 8 = padding
            int spare = NLMC_FZ_W - nreal;                     // padding entries (they gather a byte that is always 0: one address for
                                                               // every lane, no conflict) -- a lane may spend one on a contested slot
            const unsigned rot = (unsigned)tid * 3u;
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; ++q) {
                unsigned chosen = 8u;
                const unsigned gen = claim_gen + (unsigned)q + 1u;
                if (rem) {
                    unsigned cand = rem;
                    bool got = false;
#pragma unroll
                    for (int at = 0; at < 4; ++at) {
                        if (!got && cand) {
                            // a candidate among the remaining ones, starting at a lane-dependent position
                            const unsigned r8 = ((cand | (cand << 8)) >> ((rot + (unsigned)q) & 7u)) & 0xFFu;
                            const unsigned e = ((unsigned)__builtin_ctz(r8) + ((rot + (unsigned)q) & 7u)) & 7u;
                            const unsigned old = atomicExch(&claim[(unsigned)((banks >> (5 * e)) & 31ull)], gen);
                            if (old != gen) { got = true; chosen = e; }
                            cand &= ~(1u << e);
                        }
                    }
                    if (!got && spare > 0) { --spare; }               // every candidate's bank is taken: padding here, the entries later
                    else {
                        if (!got) chosen = (unsigned)__builtin_ctz(rem);
                        rem &= ~(1u << chosen);
                    }
                }
                ordpk |= chosen << (4 * q);
            }
            if (a.fmt == NLMC_FMT_ADDR) {
                // (the address format -- what a +-J instance gets -- needs the 16-bit address of the chosen entry only: a 64-bit
                // select and a shift per slot instead of the chain of selects over whole entries below, which was a third of this
                // loop's instructions)
                uint32_t pk[NLMC_FZ_W / 2] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int q = 0; q < NLMC_FZ_W; ++q) {
                    const unsigned e = (ordpk >> (4 * q)) & 15u;
                    const unsigned long long wsel = (e & 4u) ? adr_hi : adr_lo;
                    const uint32_t v16 = e == 8u ? (uint32_t)a.k_zero : (uint32_t)(wsel >> (16u * (e & 3u))) & 0xFFFFu;
                    pk[q / 2] |= v16 << (16 * (q & 1));
                }
                ell[pos] = make_int4((int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]);
                head[pos] = second ? make_int2(k | 0x8000 | thr, hy)
                                   : make_int2(k | (deg > 2 * NLMC_FZ_W ? 0x4000 : 0) | (deg > NLMC_FZ_W ? 0x8000 : 0) | thr, hy);
                continue;
            }
            EdgeQ od[NLMC_FZ_W];
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; ++q) {
                const unsigned e = (ordpk >> (4 * q)) & 15u;
                EdgeQ sel = ed[0];
#pragma unroll
                for (int j = 1; j < NLMC_FZ_W; ++j) { sel.col = e == (unsigned)j ? ed[j].col : sel.col; sel.q = e == (unsigned)j ? ed[j].q : sel.q; }
                od[q] = sel;
            }
            // real entries first in `od` is no longer true: mark padding by a zero coupling and the dummy column
            int nr = 0;
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; ++q) {
                const bool pad = ((ordpk >> (4 * q)) & 15u) == 8u;
                ed[q] = pad ? EdgeQ{-1, 0} : od[q];
                nr += pad ? 0 : 1;
            }
            (void)nr;
        }
        auto is_real = [&](int q) { return a.bank_aware ? ed[q].col >= 0 : q < left; };
        if (a.fmt == NLMC_FMT_ADDR) {
            // 16-bit LDS addresses: s_j for Jq = +1, the negated copy for Jq = -1, the zero byte for padding
            uint32_t pk[NLMC_FZ_W / 2];
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; q += 2) {
                const uint32_t lo = is_real(q) ? (uint32_t)(ed[q].col + (ed[q].q < 0 ? a.neg_off : 0)) : (uint32_t)a.k_zero;
                const uint32_t hi = is_real(q + 1) ? (uint32_t)(ed[q + 1].col + (ed[q + 1].q < 0 ? a.neg_off : 0)) : (uint32_t)a.k_zero;
                pk[q / 2] = lo | (hi << 16);
            }
            ell[pos] = make_int4((int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3]);
        }
        head[pos] = second ? make_int2(k | 0x8000 | thr, hy)
                           : make_int2(k | (deg > 2 * NLMC_FZ_W ? 0x4000 : 0) | (deg > NLMC_FZ_W ? 0x8000 : 0) | thr, hy);
        if (a.fmt == NLMC_FMT_ADDR) continue;
        if (a.fmt == NLMC_FMT_COMPACT) {
            uint32_t pk[NLMC_FZ_W];
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; ++q) pk[q] = is_real(q) ? ((uint32_t)ed[q].col << 16) | ((uint32_t)ed[q].q & 0xFFFFu) : dpack;
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; q += 4)
                ell[(size_t)(q / 4) * PS + pos] = make_int4((int)pk[q], (int)pk[q + 1], (int)pk[q + 2], (int)pk[q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < NLMC_FZ_W; q += 2) {
                const EdgeQ z{a.k_dummy, 0};
                const EdgeQ f0 = is_real(q) ? ed[q] : z, f1 = is_real(q + 1) ? ed[q + 1] : z;
                ell[(size_t)(q / 2) * PS + pos] = make_int4(f0.col, f0.q, f1.col, f1.q);
            }
        }
    }
    if (a.stats && tid == 0) {
        st[5] = (long long)__builtin_readcyclecounter() - p1;
        for (int i = 0; i < 6; ++i) a.stats[(size_t)w * 8 + i] = st[i];
    }
}

// ------------------------------------------------------------------------------------------------------
// sweeps
// ------------------------------------------------------------------------------------------------------
#ifdef NLMC_STAMPS
#define NLMC_CLK(v) { __builtin_amdgcn_sched_barrier(0); v = (long long)__builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define NLMC_CLK(v)
#endif

#ifdef NLMC_DEBUG_KNOBS
#define NLMC_DBG_NOLOAD && !(a.dbg_flags & 4)
#define NLMC_DBG_BARRIER if (!(a.dbg_flags & 64)) __syncthreads();
#define NLMC_DBG_NOPROLOGUE && !(a.dbg_flags & 256)
#else
#define NLMC_DBG_NOPROLOGUE
#define NLMC_DBG_NOLOAD
#define NLMC_DBG_BARRIER __syncthreads();
#endif
#ifdef NLMC_STAMPS
#define NLMC_FW0 NLMC_CLK(sw0)
#define NLMC_FW1 NLMC_CLK(sw1)
#define NLMC_FW2 NLMC_CLK(sw2) st_work += sw1 - sw0; st_bar += sw2 - sw1;
#define NLMC_FCALL st_calls += 1;
#else
#define NLMC_FW0
#define NLMC_FW1
#define NLMC_FW2
#define NLMC_FCALL
#endif

// The swap round of the PREVIOUS replica-exchange round, decided in the prologue of this round's sweep launch (k_sweep_fused<..,
// DEFER>, nlmc_pt_rounds_deferred): every chain looks up the selected pair its slot belongs to, reads its partner's energy as the
// previous launch published it, takes the decision k_pt_swap would take (same keys, same arithmetic; both chains of a pair compute
// it) and updates its OWN entries of the slot maps -- chain_of_slot[new slot] is written by the chain that moves there and read,
// before that, by the same chain only; nothing one chain writes is read by another inside the launch, so there is no fence, no
// atomic and no second copy of the maps.  Wave 0 does it while the other waves load the spins.
struct DeferSwap {
    int ladder_len, n_pairs, n_ladders;
    uint32_t round;
    const int32_t *plan_pairs;       // the round's planned selection [n_ladders][n_pairs][2]
    const double *beta;              // [ladder_len]
    int32_t *slot_of_chain, *chain_of_slot;
    const double *e_prev;            // [n_chains_global] energies after the previous round's sweeps
    int32_t *log_pairs;              // the round's rows of the device-side swap log, or nullptr
    uint8_t *log_acc;
};

struct SweepArgs {
    CsrDev g;
    DeferSwap defer;          // (k_sweep_fused<.., DEFER = true> only)
    int chain_base;
    const int32_t *chain_list; // launch over a SUBSET of the context's chains: block b runs local chain chain_list[b] (state rows
                              // and RNG use the chain id, recorded traces / energy traces are dense in b); nullptr: block b = chain b
    int8_t *spins;            // [n_chains][n_pad]
    const uint8_t *flags;     // [n_chains][n_pad] or nullptr
    double temp_x;
    // schedule: ord2[o][i] = { k | deg << 16, row start }, lvl_off[o][0..nlev], nlev[o]
    const int2 *ord2;
    const int32_t *lvl_off, *nlev, *hi_max;
    const EdgeQ *ell32;       // packed schedule (philox kernels), see LevelizeArgs: [o][4][n] int4 planes
    const int2 *head32;
    const int32_t *ellc64;
    const double *ellv64, *headh64;
    double qinv64;                // 2^-qs (packed fp64 window: J = Jq 2^-qs exactly)
    int per_chain;            // order id = c * n_sweeps + t, else t
    int n_sweeps;             // sweeps in this launch
    uint32_t sweep0;          // global index of sweep 0 of this launch
    uint32_t seed_lo, seed_hi;
    // temperature table: element (row, t, j) at tab[row*tab_cs + t*tab_ss + j], row = slot or local chain
    const double *tab;
    int tab_cs, tab_ss;
    const int32_t *slot_of_chain;   // [n_chains_global] or nullptr
    // stream mode
    const double *ustream;    // [n_chains*n_sweeps][n] uniform to be consumed by spin k
    // energies
    long long *efix;          // [n_chains] in/out
    double *energy_sink;      // [n_chains] or nullptr: tracked energy as a double, written with the final state
    int escale;
    int eshift;               // escale - qs: an energy delta of the fixed-point path is  (s - s') X << eshift
    float qinv;               // 2^-qs
    long long *etrace;        // [n_chains][trace_sweeps] or nullptr
    int trace_sweeps, t0;     // sweeps of the whole call / index of this launch's first sweep inside the call
    int rec_stride;
    int min_stride;           // the running minimum looks at sweeps tg = 0, min_stride, 2 min_stride, ... of the call only (>= 1): the
                              // reference takes its argmin over the RECORDED columns M[:, ::M_skip] (NMC/nmc.py:390-395)
    int8_t *strace;           // [n_chains][ceil(trace_sweeps/rec_stride)][n] or nullptr
    long long *emin;          // [n_chains] (in/out) or nullptr
    int32_t *argmin;          // [n_chains]
    int8_t *best;             // [n_chains][n_pad] or nullptr
    // LDS carve-up (bytes from the dynamic base)
    int lds_flags_off, lds_u_off, lds_loff_off, lds_red_off;
    int lds_u_stride, lds_loff_stride;   // bytes between the two copies (0: single-buffered)
    long long *dbg;           // diagnostic build (-DNLMC_STAMPS) only: per-wave cycle sums [chains][16][4]
    // fused-window schedule (k_sweep_fused): one merged level list for all n_sweeps of the launch
    const int32_t *fsend;     // [n_sweeps] published index of the last level holding an item of sweep t
    const int2 *warm_head;    // head / plane arrays of the NEXT planned window (or nullptr): pulled towards the chip by
    const EdgeQ *warm_ell;    // the helper waves while this window runs
    int f_workers;            // waves that take schedule items
    int f_gen_prio;           // raise the issue priority of the producing waves
    int f_gen0;               // first wave that produces thresholds (waves [f_gen0, nt/64) share that work)
    int fz_pstride;           // schedule positions reserved per window (plane stride of ell32, length of head32)
    int fz_npos_next;         // positions the NEXT window actually uses (for the warm-up touches)
    int fz_fmt;               // entry format of the plan: NLMC_FMT_WIDE / _COMPACT / _ADDR, see FusedItem
    int lds_neg_off;          // NLMC_FMT_ADDR: LDS offset of the negated copy of the spins (n_pad + 16); 0: none
    int lds_send_off;
    int lds_snap_off;         // k_sweep_fused<.., OUT>: three snapshot slots of n_pad bytes
    int8_t *snap_g;           // ... or, when they do not fit in LDS beside the threshold tables (n > ~9000): the same three slots
                              // per block in global memory, [blocks][3 n_pad + 16]; nullptr: LDS
    // fused windows of the fp64 mode (k_sweep_fused<.., F64 = true>): the field is the exact integer X (J = Jq 2^-qs, h = hq 2^-qs
    // exactly), the acceptance an integer threshold per value of X (accept_count_spec), built per chain in the kernel's prologue
    // random numbers keyed by (ladder, GLOBAL temperature slot) instead of by chain (nlmc_apt_shard; 0: by chain): the id of the
    // chain on slot s of local ladder j is j rng_stride + rng_base + s, whichever chain currently sits there
    int rng_stride, rng_base, rng_ladder_len;
    int lds_kt_off;           // LDS offset of the threshold tables: Khi u32[2 xmax + 1] | Klo u32[2 xmax + 1]
    int f64_xmax;             // largest |X| any row can reach: max_k (sum |Jq| + |hq|)
    unsigned f64_tie_mask;    // 0xFFFFFFFF; a test knob (NLMC_F64_TIE_MASK) clears low bits so that the rare exact path runs often
    int dbg_flags;            // -DNLMC_DEBUG_KNOBS builds only (NLMC_DBG_FLAGS): 1 = no threshold production, 2 = no updates, 4 = no item loads,
                              // 512 / 1024 = bank-conflict-free addresses for the neighbour gather / the spin's own accesses (wrong results)
};

// a wave-uniform 64-bit value pinned into scalar registers (a uniform value in a vector register costs 64 lanes, a spilled
// scalar one lane of a spill register)
__device__ __forceinline__ long long uniform64(long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// ---- pieces shared by the two sweep kernels ------------------------------------------------------------
struct ChainCtx {
    long long *st;            // diagnostic build: per-thread stamp sums inside update_spin
    long long *lvl_t;         // diagnostic build: per-level barrier-to-barrier cycles [48] + widths [48] (thread 0)
    int8_t *s;
    uint8_t *fl;
    long long *red;
    int tid, nt, c, ob, n, n_pad;   // c: local chain id (state rows, RNG), ob: block index (rows of the recorded outputs)
    int ustride;              // fused schedule: bytes between the uniform tables of consecutive table slots
    long long e_loc, E, Emin;
    int amin;
    bool per_sweep;
};

__device__ __forceinline__ void chain_load(const SweepArgs &a, unsigned char *lds_raw, ChainCtx &x)
{
    x.st = nullptr;
    x.lvl_t = nullptr;
    x.ustride = a.lds_u_stride;
    x.n = a.g.n; x.n_pad = a.g.n_pad;
    x.tid = threadIdx.x; x.nt = blockDim.x; x.ob = blockIdx.x;
    x.c = a.chain_list ? a.chain_list[blockIdx.x] : (int)blockIdx.x;
    x.s = reinterpret_cast<int8_t *>(lds_raw);
    x.fl = a.flags ? (lds_raw + a.lds_flags_off) : nullptr;
    x.red = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);   // [0] sweep sum, [1] broadcast flag
    const int4 *src = reinterpret_cast<const int4 *>(a.spins + (size_t)x.c * x.n_pad);
    int4 *dst = reinterpret_cast<int4 *>(x.s);
    for (int i = x.tid; i < x.n_pad / 16; i += x.nt) dst[i] = src[i];
    if (x.fl) {
        const int4 *fsrc = reinterpret_cast<const int4 *>(a.flags + (size_t)x.c * x.n_pad);
        int4 *fdst = reinterpret_cast<int4 *>(x.fl);
        for (int i = x.tid; i < x.n_pad / 16; i += x.nt) fdst[i] = fsrc[i];
    }
    if (x.tid == 0) { x.red[0] = 0; x.red[1] = 0; }
    x.e_loc = 0;
    x.E = uniform64(a.efix[x.c]);          // E, Emin, amin: the same value in every thread, kept in scalar registers
    x.Emin = a.emin ? uniform64(a.emin[x.c]) : 0;
    x.amin = a.emin ? __builtin_amdgcn_readfirstlane(a.argmin[x.c]) : 0;
    x.per_sweep = (a.etrace != nullptr) || (a.emin != nullptr);
    __syncthreads();
}

// end-of-sweep bookkeeping: energy trace, running minimum + argmin state, recorded configurations
__device__ __forceinline__ void sweep_epilogue(const SweepArgs &a, ChainCtx &x, int t)
{
    const int tg = a.t0 + t;                                 // sweep index inside the call
    const bool rec = a.strace && (tg % a.rec_stride == 0);   // M[:, ::M_skip]  (NMC/nmc.py:390)
    bool read_spins = rec;                                   // the LDS spins are copied out below (workgroup-uniform)
    if (x.per_sweep) {
        const long long w = wave_sum_i64(x.e_loc);
        x.e_loc = 0;
        if ((x.tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&x.red[0]), (unsigned long long)w);
        __syncthreads();
        x.E += uniform64(x.red[0]);              // every thread: the sum stays wave-uniform (no broadcast of the decision)
        const bool better = a.emin && x.E < x.Emin && (tg % a.min_stride == 0);      // strict <: first argmin (np.argmin)
        if (better) { x.Emin = x.E; x.amin = tg; }
        if (x.tid == 0 && a.etrace) a.etrace[(size_t)x.ob * a.trace_sweeps + tg] = x.E;
        __syncthreads();                         // everybody has read the sum
        if (x.tid == 0) x.red[0] = 0;            // (next added to at the end of the next sweep, many barriers from here)
        if (a.best && better) {
            read_spins = true;
            int4 *dst = reinterpret_cast<int4 *>(a.best + (size_t)x.c * x.n_pad);
            const int4 *src = reinterpret_cast<const int4 *>(x.s);
            for (int i = x.tid; i < x.n_pad / 16; i += x.nt) dst[i] = src[i];
        }
    }
    if (rec) {
        const int n_rec = (a.trace_sweeps + a.rec_stride - 1) / a.rec_stride;
        int8_t *dst = a.strace + ((size_t)x.ob * n_rec + (size_t)(tg / a.rec_stride)) * x.n;
        for (int i = x.tid; i < x.n; i += x.nt) dst[i] = x.s[i];
    }
    if (read_spins) __syncthreads();           // LDS spins are rewritten next sweep
}

__device__ __forceinline__ void chain_store(const SweepArgs &a, ChainCtx &x)
{
    if (!x.per_sweep) {
        const long long w = wave_sum_i64(x.e_loc);
        if ((x.tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&x.red[0]), (unsigned long long)w);
        __syncthreads();
        x.E += uniform64(x.red[0]);
    }
    int4 *dst = reinterpret_cast<int4 *>(a.spins + (size_t)x.c * x.n_pad);
    const int4 *src = reinterpret_cast<const int4 *>(x.s);
    for (int i = x.tid; i < x.n_pad / 16; i += x.nt) dst[i] = src[i];
    if (x.tid == 0) {
        a.efix[x.c] = x.E;
        if (a.energy_sink) a.energy_sink[x.c] = (double)x.E * __longlong_as_double((long long)(1023 - a.escale) << 52);
        if (a.emin) { a.emin[x.c] = x.Emin; a.argmin[x.c] = x.amin; }
    }
}

// ---- STREAM mode: the reference's arithmetic, fp64, externally drawn uniforms --------------------------
// m_k = sign(tanh(beta x_k) - 2u + 1),  x = (sum_e J_e m_e) + h_k in CSR order  (NMC/nmc.py:86-87)
// The reference stream of one (chain, sweep) as drawn -- perm[i] = spin visited i-th, u[i] = its uniform -- scattered by spin:
// rank[perm[i]] = i (the sort key of k_levelize), u_spin[perm[i]] = u[i].  *bad is set when a row is not a permutation.
__global__ void k_stream_scatter(int n, const int32_t *perm, const double *u, uint32_t *rank, double *us, int32_t *bad)
{
    extern __shared__ unsigned int seen_lds[];
    const size_t o = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += blockDim.x) seen_lds[i] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int k = perm[o * n + i];
        if (k < 0 || k >= n || atomicExch(&seen_lds[k], 1u) != 0u) { atomicOr(bad, 1); continue; }
        rank[o * n + k] = (uint32_t)i;
        us[o * n + k] = u[o * n + i];
    }
}

__global__ void k_sweep_stream(SweepArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ChainCtx x;
    chain_load(a, lds_raw, x);
    const int n = x.n, tid = x.tid, c = x.c;
    int8_t *s = x.s;
    const uint8_t *fl = x.fl;
    const double esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale

    // One item per thread and level (k_levelize splits levels at the workgroup's width).  Software pipeline over the levels of a
    // sweep (round 3; the loop used to chase ord -> col / val -> spin one dependent global load after the other, 1 + degree
    // round trips per level: 52 us per sweep of one chain at N = 10^3): while level l is computed, the first 8 entries of the
    // rows of level l+1 (+ their uniform and field term) and the schedule entries of level l+2 are in flight.  The sums keep
    // the reference's association: entries in CSR order, one rounding per term.  (col / val64 are allocated with 16 padding
    // entries, so the unconditional 8-entry reads of a short last row stay inside the arrays.)
    struct Row { int cj[8]; double vj[8]; double u, h; };
    for (int t = 0; t < a.n_sweeps; ++t) {
        const int oid = a.per_chain ? (c * a.n_sweeps + t) : t;
        const int2 *__restrict__ ord = a.ord2 + (size_t)oid * n;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * (n + 1);
        const int nl = a.nlev[oid];
        const double beta = a.tab[(size_t)c * a.tab_cs + (size_t)t * a.tab_ss];
        const double *__restrict__ ut = a.ustream + ((size_t)c * a.n_sweeps + t) * n;
        // level offsets of the sweep in LDS (a scalar load per level would sit in the same counter as the LDS reads of the
        // update: every wait for a spin would wait for it, too)
        int *loff = reinterpret_cast<int *>(lds_raw + a.lds_loff_off);
        const bool in_lds = nl < NLMC_LCAP;
        if (in_lds) {
            __syncthreads();                               // (the previous sweep's readers are done)
            for (int l = tid; l <= nl; l += x.nt) loff[l] = off[l];
            __syncthreads();
        }
        auto item_at = [&](int l, int2 &en, bool &valid) __attribute__((always_inline)) {
            const int lc = min(l, max(nl - 1, 0));
            const int lo = in_lds ? loff[lc] : off[lc], hi = in_lds ? loff[lc + 1] : off[lc + 1];
            valid = (l < nl) && (lo + tid < hi);
            en = ord[valid ? lo + tid : 0];
        };
        auto row_of = [&](const int2 &en, Row &r) __attribute__((always_inline)) {
            const int k = en.x & 0xFFFF, rs = en.y;
#pragma unroll
            for (int q = 0; q < 8; ++q) { r.cj[q] = a.g.col[rs + q]; r.vj[q] = a.g.val64[rs + q]; }
            r.u = ut[k];
            r.h = a.g.h64[k];
        };
        int2 enA, enB;
        bool vA, vB;
        Row rA;
        item_at(0, enA, vA);
        row_of(enA, rA);
        item_at(1, enB, vB);
        for (int l = 0; l < nl; ++l) {
            int2 enC;
            bool vC;
            Row rB;
            item_at(l + 2, enC, vC);
            row_of(enB, rB);
            if (vA) {
                const int k = enA.x & 0xFFFF, rs = enA.y, deg = (int)((unsigned)enA.x >> 16), re = rs + deg;
                const unsigned f = fl ? (unsigned)fl[k] : 0u;
                const int so = (int)s[k];
                double sj[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) sj[q] = (double)s[rA.cj[q]];      // (entries past the row: some valid column, unused)
                double xs = 0.0, xd = 0.0;   // xd: diagonal term, excluded from the energy delta
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double tm = rA.vj[q] * sj[q];
                    if (q < deg) { xs += tm; if (rA.cj[q] == k) xd += tm; }
                }
                for (int e = rs + 8; e < re; ++e) {
                    const int j = a.g.col[e];
                    const double tm = a.g.val64[e] * (double)s[j];
                    xs += tm;
                    if (j == k) xd += tm;
                }
                const double hk = rA.h;
                const double x_true = (xs - xd) + hk;   // field of the UNMODIFIED (J,h)
                double xp;
                if (f == 0u) xp = xs + hk;
                else if (f == 1u) {   // cluster rows divided element-wise by temp_x (NMC/nmc.py:379-380)
                    double y = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const double tm = (rA.vj[q] / a.temp_x) * sj[q]; if (q < deg) y += tm; }
                    for (int e = rs + 8; e < re; ++e) y += (a.g.val64[e] / a.temp_x) * (double)s[a.g.col[e]];
                    xp = y + hk / a.temp_x;
                } else xp = xs + ((f == 2u) ? 10000.0 : -10000.0);   // NMC/nmc.py:381,401
                const double v = tanh(beta * xp) - 2.0 * rA.u + 1.0;
                const int sn = (v > 0.0) - (v < 0.0);                // np.sign
                if (sn != so) {
                    x.e_loc += __double2ll_rn(-(double)(sn - so) * x_true * esc);
                    s[k] = (int8_t)sn;
                }
            }
            __syncthreads();
            enA = enB; vA = vB; rA = rB;
            enB = enC; vB = vC;
        }
        sweep_epilogue(a, x, t);
    }
    chain_store(a, x);
}

// ---- PHILOX mode (throughput) ---------------------------------------------------------------------------
template <typename T, bool PK = false> struct Pf;   // one schedule item: k, degree, h_k and the packed window of row k
                                                    // (PK: fp64 path with 16-bit columns / couplings)
typedef int nlmc_i4 __attribute__((ext_vector_type(4)));
typedef int nlmc_i2 __attribute__((ext_vector_type(2)));
typedef float nlmc_f4 __attribute__((ext_vector_type(4)));
template <> struct Pf<float, false> {
    static constexpr int W = NLMC_ELL_W32;
    nlmc_i4 pk[W / 2];            // plane q: { col(2q), Jq(2q), col(2q+1), Jq(2q+1) }   (fixed-point couplings)
    nlmc_i2 hd;                   // { k | deg << 16, hq_k }
    // The loads of one item (head + 4 planes, or + 8 planes in waves that may hold rows longer than 8 entries) are
    // issued unconditionally inside a pipeline stage: a load behind a branch makes hipcc's vmcnt model
    // path-dependent, and it then drains vmcnt to 0 -- i.e. waits for the prefetch it has just issued -- before every
    // update; with a static count per loop copy it emits the counted wait.  Different load counts therefore live in
    // separate copies of the level loop (run_levels<TAIL>), and a wave without further work leaves the loop instead
    // of branching around its loads.  (Inline-asm loads with hand-placed waits were tried: hipcc copies the
    // still-pending destination registers around the long-row tail loop, which corrupts them.)
    // The schedule is read through two raw-buffer descriptors (SRSRC): 32-bit offsets instead of 64-bit address
    // arithmetic, the plane offset in the scalar-offset operand, and the hardware range check turns the loads of idle
    // lanes (offset >= size) into no-ops that return 0 without touching memory.  Coalesced: consecutive lanes read
    // consecutive positions of every plane.
    struct View {
        __amdgpu_buffer_rsrc_t ell, head;
        int plane_bytes;
        __device__ __forceinline__ void bind(const SweepArgs &a, size_t oid, int n)
        {
            const char *pe = reinterpret_cast<const char *>(a.ell32) + oid * (W / 2) * (size_t)n * 16;
            const char *ph = reinterpret_cast<const char *>(a.head32) + oid * (size_t)n * 8;
            plane_bytes = n * 16;
            ell = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(pe), 0, (W / 2) * n * 16, 0x00020000);
            head = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(ph), 0, n * 8, 0x00020000);
        }
    };
    template <bool TAIL>
    __device__ __forceinline__ void issue(const View &v, int i, bool valid)
    {
        const int oob = 0x7FF00000;                      // past both buffers even after adding the plane offsets
        const int o16 = valid ? i * 16 : oob;
        hd = __builtin_amdgcn_raw_buffer_load_b64(v.head, valid ? i * 8 : oob, 0, 0);
#pragma unroll
        for (int q = 0; q < (TAIL ? W / 2 : 4); ++q)   // plane offset rides in the scalar offset operand: no VALU add
            pk[q] = __builtin_amdgcn_raw_buffer_load_b128(v.ell, o16, q * v.plane_bytes, 0);
    }
    __device__ __forceinline__ int kd() const { return hd.x; }
    __device__ __forceinline__ int h() const { return hd.y; }
    __device__ __forceinline__ int col(int q) const { return (q & 1) ? pk[q >> 1].z : pk[q >> 1].x; }
    __device__ __forceinline__ int val(int q) const { return (q & 1) ? pk[q >> 1].w : pk[q >> 1].y; }
    static __device__ __forceinline__ void tail(const CsrDev &g, int e, int &cj, int &vj) { const EdgeQ t = g.edge32[e]; cj = t.col; vj = t.q; }
};
template <> struct Pf<double, false> {
    static constexpr int W = NLMC_ELL_W;
    int cj[NLMC_ELL_W];
    double vj[NLMC_ELL_W];
    int kd_;
    double h_;
    struct View {
        const int32_t *pc;
        const double *pv, *ph;
        const int2 *po;
        int n;
        __device__ __forceinline__ void bind(const SweepArgs &a, size_t oid, int n_)
        {
            n = n_;
            pc = a.ellc64 + oid * NLMC_ELL_W * n;
            pv = a.ellv64 + oid * NLMC_ELL_W * n;
            ph = a.headh64 + oid * n;
            po = a.ord2 + oid * n;
        }
    };
    template <bool TAIL>
    __device__ __forceinline__ void issue(const View &v, int i, bool valid)
    {
        const int ic = valid ? i : 0;
        const nlmc_i4 *pc4 = reinterpret_cast<const nlmc_i4 *>(v.pc);
        typedef double nlmc_d2 __attribute__((ext_vector_type(2)));
        const nlmc_d2 *pv2 = reinterpret_cast<const nlmc_d2 *>(v.pv);
        constexpr int NQ = TAIL ? NLMC_ELL_W : 8;          // entries this wave reads: the first 8, or the whole window
#pragma unroll
        for (int pl = 0; pl < NQ / 4; ++pl) {
            const nlmc_i4 c4 = pc4[(size_t)pl * v.n + ic];
            cj[4 * pl] = c4.x; cj[4 * pl + 1] = c4.y; cj[4 * pl + 2] = c4.z; cj[4 * pl + 3] = c4.w;
        }
#pragma unroll
        for (int pl = 0; pl < NQ / 2; ++pl) {
            const nlmc_d2 v2 = pv2[(size_t)pl * v.n + ic];
            vj[2 * pl] = v2.x; vj[2 * pl + 1] = v2.y;
        }
        kd_ = v.po[ic].x;
        h_ = v.ph[ic];
    }
    __device__ __forceinline__ int kd() const { return kd_; }
    __device__ __forceinline__ double h() const { return h_; }
    __device__ __forceinline__ int col(int q) const { return cj[q]; }
    __device__ __forceinline__ double val(int q) const { return vj[q]; }
    static __device__ __forceinline__ void tail(const CsrDev &g, int e, int &c, double &v) { c = g.col[e]; v = g.val64[e]; }
};

template <> struct Pf<double, true> {
    static constexpr int W = NLMC_ELL_W;
    nlmc_i4 pc[NLMC_ELL_W / 8], pq[NLMC_ELL_W / 8];   // 8 columns / 8 couplings per 16-byte plane
    int kd_;
    double h_, qi;
    struct View {
        const nlmc_i4 *pp;
        const double *ph;
        const int2 *po;
        int n;
        double qi;
        __device__ __forceinline__ void bind(const SweepArgs &a, size_t oid, int n_)
        {
            n = n_;
            pp = reinterpret_cast<const nlmc_i4 *>(a.ellc64 + oid * NLMC_ELL_W * n);
            ph = a.headh64 + oid * n;
            po = a.ord2 + oid * n;
            qi = a.qinv64;
        }
    };
    template <bool TAIL>
    __device__ __forceinline__ void issue(const View &v, int i, bool valid)
    {
        const int ic = valid ? i : 0;
#pragma unroll
        for (int hf = 0; hf < (TAIL ? NLMC_ELL_W / 8 : 1); ++hf) {
            pc[hf] = v.pp[(size_t)(2 * hf) * v.n + ic];
            pq[hf] = v.pp[(size_t)(2 * hf + 1) * v.n + ic];
        }
        kd_ = v.po[ic].x;
        h_ = v.ph[ic];
        qi = v.qi;
    }
    static __device__ __forceinline__ int word(const nlmc_i4 &w, int j) { return j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w; }
    __device__ __forceinline__ int kd() const { return kd_; }
    __device__ __forceinline__ double h() const { return h_; }
    __device__ __forceinline__ int col(int q) const { return (int)(((unsigned)word(pc[q >> 3], (q & 7) >> 1) >> ((q & 1) * 16)) & 0xFFFFu); }
    __device__ __forceinline__ double val(int q) const     // Jq 2^-qs: exact (a power-of-two scale)
    {
        const int w = word(pq[q >> 3], (q & 7) >> 1);
        return (double)((q & 1) ? (w >> 16) : (int)(short)(w & 0xFFFF)) * qi;
    }
    static __device__ __forceinline__ void tail(const CsrDev &g, int e, int &c, double &v) { c = g.col[e]; v = g.val64[e]; }
};

// -(ds) * x * 2^escale rounded to the nearest integer (fp64 paths)
__device__ __forceinline__ long long fixed_delta_slow(double xt, int ds, double esc) { return __double2ll_rn(-(double)ds * xt * esc); }

__device__ __forceinline__ float scale_cb(float cb, float qinv) { return cb * qinv; }
__device__ __forceinline__ double scale_cb(double cb, float) { return cb; }
__device__ __forceinline__ double fma_rn(double a, double b, double c) { return __fma_rn(a, b, c); }

// Heat-bath update of one spin from a prefetched schedule item.  J*s is exact (s = +-1), so fma(J, s, x) == x + J*s;
// the zero-padded slots add +-0 and leave x unchanged, which keeps the oracle's row-order sum bit for bit.
template <bool DIAG, bool TAIL, bool FUSED>
__device__ __forceinline__ void update_spin_q(const SweepArgs &a, ChainCtx &x, const float *wt, const Pf<float> &pf, size_t oid, int i,
                                              float cq0, float cq1);

template <typename T, bool DIAG, bool TAIL, bool FUSED = false, bool PK = false>
__device__ __forceinline__ void update_spin(const SweepArgs &a, ChainCtx &x, const T *ur, const Pf<T, PK> &pf, size_t oid, int i,
                                            T cb0, T cb1, double esc)
{
    if constexpr (sizeof(T) == 4) {            // "f32" throughput mode: fixed-point field, threshold test
        update_spin_q<DIAG, TAIL, FUSED>(a, x, ur, pf, oid, i, cb0, cb1);
        return;
    } else {
    // fused schedule items carry the uniform-table slot of their sweep (t mod 3) in the two top bits
    const int k = pf.kd() & 0xFFFF, deg = FUSED ? (int)(((unsigned)pf.kd() >> 16) & 0x3FFFu) : (int)((unsigned)pf.kd() >> 16);
    const unsigned f = x.fl ? (unsigned)x.fl[k] : 0u;
    if (f >= 2u) return;                       // frozen
    int8_t *s = x.s;
    const T *urs = FUSED ? reinterpret_cast<const T *>(reinterpret_cast<const unsigned char *>(ur) +
                                                        ((unsigned)pf.kd() >> 30) * (unsigned)x.ustride) : ur;
    const T uk = urs[k];                       // issued up front with the gathers: off the dependent chain
    const int so = (int)s[k];
#ifdef NLMC_STAMPS
    long long u0, u1, u2, u3, u4;
    NLMC_CLK(u0)
#endif
    constexpr int W = Pf<T, PK>::W;
    // The spins sit at LDS offset 0 (first region of the dynamic LDS, no static LDS in this kernel -- checked at
    // kernel entry), so a column index IS the LDS address: no per-read base add.
    typedef const int8_t __attribute__((address_space(3))) *lds_i8;
#define NLMC_SPIN(c) ((T)(*(lds_i8)(uintptr_t)(unsigned)(c)))
    T sj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sj[q] = NLMC_SPIN(pf.col(q));         // all LDS reads in flight together
#ifdef NLMC_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NLMC_CLK(u1)
#endif
    T xs = T(0), xd = T(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        xs = fma_rn(pf.val(q), sj[q], xs);
        if (DIAG) { const T nd = fma_rn(pf.val(q), sj[q], xd); xd = (q < deg && pf.col(q) == k) ? nd : xd; }
    }
    if (TAIL && W > 8 && __ballot(deg > 8) != 0ull) {  // second half of the packed window: only waves holding a long row
        T sk[8];
#pragma unroll
        for (int q = 8; q < W; ++q) sk[q - 8] = NLMC_SPIN(pf.col(q));
#pragma unroll
        for (int q = 8; q < W; ++q) {
            xs = fma_rn(pf.val(q), sk[q - 8], xs);
            if (DIAG) { const T nd = fma_rn(pf.val(q), sk[q - 8], xd); xd = (q < deg && pf.col(q) == k) ? nd : xd; }
        }
    }
    if (deg > W) {                             // rows longer than the packed window: rest from the CSR arrays
        const int rs = FUSED ? a.g.rowptr[k] : a.ord2[oid * x.n + i].y;
        for (int e = W; e < deg; ++e) {
            int j; T v;
            Pf<T, PK>::tail(a.g, rs + e, j, v);
            const T sv = (T)s[j];
            xs = fma_rn(v, sv, xs);
            if (DIAG && j == k) xd = fma_rn(v, sv, xd);
        }
    }
    const T hk = pf.h();
    const T x_true = DIAG ? ((xs - xd) + hk) : (xs + hk);
    const T xf = xs + hk;
#ifdef NLMC_STAMPS
    asm volatile("" :: "v"(xf));
    NLMC_CLK(u2)
#endif
    const T z = (f == 1u ? cb1 : cb0) * xf;
    const int sn = accept_up(uk, z) ? 1 : -1;
#ifdef NLMC_STAMPS
    asm volatile("" :: "v"(sn));
    NLMC_CLK(u3)
#endif
    // straight-line tail: branches cost a wave far more than the handful of integer ops they would skip
    const int ds = sn - so;                    // 0 or +-2 (spins are +-1 in this mode)
    if (ds != 0) x.e_loc += fixed_delta_slow((double)x_true, ds, esc);
    s[k] = (int8_t)sn;
#ifdef NLMC_STAMPS
    NLMC_CLK(u4)
    if (x.st) { x.st[0] += u1 - u0; x.st[1] += u2 - u1; x.st[2] += u3 - u2; x.st[3] += u4 - u3; x.st[4] += 1; }
#endif
    }
}

// Heat-bath update of one spin in the "f32" throughput mode (fixed-point couplings): the field X = hq_k + sum Jq s is
// an exact int32 whatever the order of the sum, so it is accumulated in two independent multiply-add chains
// (v_mad_i32_i24: the spins come sign-extended from LDS, |Jq| < 2^23); the acceptance is ONE compare against the
// threshold that was computed with the random number (threshold_spec); the energy delta is one 32 x 32 -> 64 bit
// multiply-add, exact.
template <bool DIAG, bool TAIL, bool FUSED>
__device__ __forceinline__ void update_spin_q(const SweepArgs &a, ChainCtx &x, const float *wt, const Pf<float> &pf, size_t oid, int i,
                                              float cq0, float cq1)
{
    const int k = pf.kd() & 0xFFFF, deg = FUSED ? (int)(((unsigned)pf.kd() >> 16) & 0x3FFFu) : (int)((unsigned)pf.kd() >> 16);
    const unsigned f = x.fl ? (unsigned)x.fl[k] : 0u;
    if (f >= 2u) return;                       // frozen
    int8_t *s = x.s;
    // fused schedule items carry the threshold-table slot of their sweep (t mod 3) in the two top bits
    const float *wts = FUSED ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(wt) +
                                                                ((unsigned)pf.kd() >> 30) * (unsigned)x.ustride) : wt;
    const float wk = wts[k];                   // issued up front with the gathers
    const int so = (int)s[k];
    constexpr int W = Pf<float>::W;
    typedef const int8_t __attribute__((address_space(3))) *lds_i8;   // spins sit at LDS offset 0: column == address
#define NLMC_SPIN_I(c) ((int)(*(lds_i8)(uintptr_t)(unsigned)(c)))
#ifdef NLMC_STAMPS
    long long u0, u1, u4;
    NLMC_CLK(u0)
#endif
    int sj[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) sj[q] = NLMC_SPIN_I(pf.col(q));       // all LDS reads in flight together
#ifdef NLMC_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    NLMC_CLK(u1)
#endif
    int X0 = pf.h(), X1 = 0, Xd = 0;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        X0 += __mul24(pf.val(q), sj[q]);
        X1 += __mul24(pf.val(q + 1), sj[q + 1]);
    }
    if (DIAG) {
#pragma unroll
        for (int q = 0; q < 8; ++q) Xd += (q < deg && pf.col(q) == k) ? __mul24(pf.val(q), sj[q]) : 0;
    }
    if (TAIL && __ballot(deg > 8) != 0ull) {   // second half of the packed window: only waves holding a long row
        int sk[8];
#pragma unroll
        for (int q = 8; q < W; ++q) sk[q - 8] = NLMC_SPIN_I(pf.col(q));
#pragma unroll
        for (int q = 8; q < W; q += 2) {
            X0 += __mul24(pf.val(q), sk[q - 8]);
            X1 += __mul24(pf.val(q + 1), sk[q - 7]);
        }
        if (DIAG) {
#pragma unroll
            for (int q = 8; q < W; ++q) Xd += (q < deg && pf.col(q) == k) ? __mul24(pf.val(q), sk[q - 8]) : 0;
        }
    }
    if (deg > W) {                             // rows longer than the packed window: rest from the CSR arrays
        const int rs = FUSED ? a.g.rowptr[k] : a.ord2[oid * x.n + i].y;
        for (int e = W; e < deg; ++e) {
            int j, v;
            Pf<float>::tail(a.g, rs + e, j, v);
            const int t = __mul24(v, (int)s[j]);
            X0 += t;
            if (DIAG && j == k) Xd += t;
        }
    }
    const int X = X0 + X1;
    const float z = (f == 1u ? cq1 : cq0) * (float)X;
    const int sn = (z < wk) ? 1 : -1;
    const int c = (so - sn) << a.eshift;       // 0 or +-2^(eshift+1) <= 2^30
    x.e_loc += (long long)(DIAG ? X - Xd : X) * (long long)c;
    s[k] = (int8_t)sn;
#ifdef NLMC_STAMPS
    asm volatile("" :: "v"(sn));
    NLMC_CLK(u4)
    if (x.st) { x.st[0] += u1 - u0; x.st[1] += u4 - u1; x.st[4] += 1; }
#endif
}

__device__ __forceinline__ float4 thresholds4(const u32x4 &r);

// uniforms of one sweep for every spin of this chain -> LDS.  One block of 4 spins = one Philox4x32-10 call (f32) / two (f64):
//   f32: W(k) = threshold_spec(word (k & 3) of philox(k >> 2, t, chain, UNIFORM))
//   f64: u(k) = uniform53_spec(word (k & 3) of philox(k >> 2, t, chain, UNIFORM), word (k & 3) of philox(k >> 2, t, chain, UNIFORM_LO))
__device__ __forceinline__ void fill_uniforms(float *ur, int n, uint32_t tt, uint32_t gc, uint32_t k0, uint32_t k1, int tid, int nt)
{
    const int nblk = (n + 3) / 4;                    // "f32" mode: the table holds the logistic thresholds W(r)
    for (int b = tid; b < nblk; b += 2 * nt) {       // two blocks per step: interleaved Philox calls
        u32x4 r0, r1;
        philox4x32_10_x2((uint32_t)b, (uint32_t)(b + nt), tt, gc, NLMC_TAG_UNIFORM, k0, k1, r0, r1);
        reinterpret_cast<float4 *>(ur)[b] = thresholds4(r0);     // ur has (n+3)/4*4 entries
        if (b + nt < nblk) reinterpret_cast<float4 *>(ur)[b + nt] = thresholds4(r1);
    }
}
__device__ __forceinline__ void fill_uniforms(double *ur, int n, uint32_t tt, uint32_t gc, uint32_t k0, uint32_t k1, int tid, int nt)
{
    for (int b = tid; b < (n + 3) / 4; b += nt) {
        const u32x4 r = philox4x32_10((uint32_t)b, tt, gc, NLMC_TAG_UNIFORM, k0, k1);
        const u32x4 q = philox4x32_10((uint32_t)b, tt, gc, NLMC_TAG_UNIFORM_LO, k0, k1);
        double2 v0, v1;
        v0.x = uniform53_spec(r.x, q.x); v0.y = uniform53_spec(r.y, q.y);
        v1.x = uniform53_spec(r.z, q.z); v1.y = uniform53_spec(r.w, q.w);
        reinterpret_cast<double2 *>(ur)[2 * b] = v0;    // ur has (n+3)/4*4 entries
        reinterpret_cast<double2 *>(ur)[2 * b + 1] = v1;
    }
}

// fused fp64 windows: the table of a sweep holds the 27 HIGH bits of every spin's 53-bit uniform (word k & 3 of the UNIFORM call
// k >> 2, shifted: one call per four spins); the 26 low bits (UNIFORM_LO call) are made only in the rare update whose high bits do
// not decide (k_sweep_fused<.., F64>)
__device__ __forceinline__ void fill_uniform_words(unsigned *tab, int n, uint32_t tt, uint32_t gc, uint32_t k0, uint32_t k1, int tid, int nt)
{
    const int nblk = (n + 3) / 4;
    for (int b = tid; b < nblk; b += 2 * nt) {
        u32x4 r0, r1;
        philox4x32_10_x2((uint32_t)b, (uint32_t)(b + nt), tt, gc, NLMC_TAG_UNIFORM, k0, k1, r0, r1);
        reinterpret_cast<uint4 *>(tab)[b] = make_uint4(r0.x >> 5, r0.y >> 5, r0.z >> 5, r0.w >> 5);
        if (b + nt < nblk) reinterpret_cast<uint4 *>(tab)[b + nt] = make_uint4(r1.x >> 5, r1.y >> 5, r1.z >> 5, r1.w >> 5);
    }
}

// The pipelined level loop of one sweep (see k_sweep_philox).  TAIL: this wave may hold rows longer than 8 entries and
// therefore also loads / folds the second half of the 16-entry row window.
struct NoGen { __device__ __forceinline__ void operator()(int) const {} };

template <typename T, bool DIAG, bool TAIL, bool FUSED = false, typename Gen = NoGen, bool PK = false>
__device__ __forceinline__ void run_levels(const SweepArgs &a, ChainCtx &x, const T *ur, const int *loff, size_t so, int nl,
                                           int n_bar, T cb0, T cb1, double esc, int n_items = 0, Gen gen = Gen())
{
    // gen(l): per-level side job of this wave, run before the level's update (k_sweep_fused: every wave produces its
    // share of the threshold table of the sweep after next while the schedule items of the next level are in flight)
    // n_bar: levels [0, n_bar) end with a workgroup barrier; levels [n_bar, nl) are at most one wave wide and belong
    // to wave 0 alone, which runs them back to back (LDS executes one wave's accesses in order, so its own writes are
    // visible to its own later reads) while the other waves already prepare the next sweep (k_sweep_philox).
    // software pipeline over levels: while level l is computed, the schedule items of level l+1 are in flight
    // (fp64: level l+2 as well was measured -- three register sets, 170 registers -- and changed nothing: 4.9 vs 5.0e10).
    // Their addresses depend only on the level offsets (LDS), never on spin values, and the schedule was built with
    // level_cap == blockDim.x: at most one spin per thread and level.
    const int n = FUSED ? n_items : x.n, tid = x.tid;     // positions per plane of the packed schedule
    Pf<T, PK> pfa, pfb;                // ping-pong (manual 2x unroll: no register rotation)
    bool va, vb;                   // lane has an item in the level held by pfa / pfb
    int ia, ib;
    typename Pf<T, PK>::View view;
    view.bind(a, so, n);
    int lo_next = loff[0], hi_next = loff[min(1, nl)];      // offsets of the level the next fetch will load
    auto fetch = [&](int l, Pf<T, PK> &p, bool &valid, int &ic) {
        const int i = lo_next + tid;
        valid = (l < nl) && (i < hi_next);
        ic = valid ? i : 0;
        p.template issue<TAIL>(view, i, valid);      // idle lanes: out-of-range no-ops
        lo_next = hi_next;                           // roll: the LDS read for level l+1 is consumed a stage later
        hi_next = loff[min(l + 2, nl)];
    };
    // Items fill the lanes of a level from 0, so wave w is idle in every level narrower than 64 w + 1; level widths
    // shrink with depth, hence each wave has a LAST level with work.  Past it the wave only keeps the barrier count
    // (second loop): no loads, no instructions -- a vector-memory instruction occupies the CU's address unit for
    // ~11 cycles per wave even when every lane is out of range, and half of all wave-levels are idle ones.
    int n_live = 0;
    {
        const int wbase = tid & ~63;
        for (int l0 = 0; l0 < nl; l0 += 64) {
            const int l = l0 + (tid & 63);
            const bool has = (l < nl) && (loff[l + 1] - loff[l] > wbase);
            const unsigned long long m = __ballot(has);
            if (m) n_live = l0 + 64 - __clzll((long long)m);
        }
        n_live = (wbase == 0) ? nl : min(n_live, n_bar);
    }
    fetch(0, pfa, va, ia);
    int l = 0;
#ifdef NLMC_STAMPS
    long long lt0; NLMC_CLK(lt0)
#define NLMC_LVL_STAMP(lv) if (x.lvl_t && tid == 0 && (lv) < 48) { long long lt1; NLMC_CLK(lt1) x.lvl_t[lv] += lt1 - lt0; x.lvl_t[48 + (lv)] = loff[(lv) + 1] - loff[lv]; lt0 = lt1; }
#else
#define NLMC_LVL_STAMP(lv)
#endif
#if defined(NLMC_V_FETCH_AFTER)
#define NLMC_STAGE(lv, pnext, vnext, inext, pcur, vcur, icur) \
    if (vcur) update_spin<T, DIAG, TAIL, FUSED, PK>(a, x, ur, pcur, so, icur, cb0, cb1, esc); \
    fetch((lv) + 1, pnext, vnext, inext); gen(lv);
#elif defined(NLMC_V_GEN_AFTER)
#define NLMC_STAGE(lv, pnext, vnext, inext, pcur, vcur, icur) \
    fetch((lv) + 1, pnext, vnext, inext); \
    if (vcur) update_spin<T, DIAG, TAIL, FUSED, PK>(a, x, ur, pcur, so, icur, cb0, cb1, esc); \
    gen(lv);
#else
#define NLMC_STAGE(lv, pnext, vnext, inext, pcur, vcur, icur) \
    fetch((lv) + 1, pnext, vnext, inext); gen(lv); \
    if (vcur) update_spin<T, DIAG, TAIL, FUSED, PK>(a, x, ur, pcur, so, icur, cb0, cb1, esc);
#endif
#ifdef NLMC_STAMPS
    // diagnostic build: per-wave cycle split of a level -- work (fetch issue + side job + update) vs barrier wait
    long long sw0, sw1, sw2, st_work = 0, st_bar = 0;
#define NLMC_W0 NLMC_CLK(sw0)
#define NLMC_W1 NLMC_CLK(sw1)
#define NLMC_W2 NLMC_CLK(sw2) st_work += sw1 - sw0; st_bar += sw2 - sw1;
#else
#define NLMC_W0
#define NLMC_W1
#define NLMC_W2
#endif
    for (; l < n_live; l += 2) {
        NLMC_W0
        NLMC_STAGE(l, pfb, vb, ib, pfa, va, ia)
        NLMC_W1
        if (l < n_bar) __syncthreads();
        NLMC_W2
        NLMC_LVL_STAMP(l)
        if (l + 1 < nl) {
            NLMC_W0
            NLMC_STAGE(l + 1, pfa, va, ia, pfb, vb, ib)
            NLMC_W1
            if (l + 1 < n_bar) __syncthreads();
            NLMC_W2
            NLMC_LVL_STAMP(l + 1)
        }
    }
#ifdef NLMC_STAMPS
    if (x.st) { x.st[5] += st_work; x.st[6] += st_bar; }
#endif
#undef NLMC_STAGE
    l = min(l, n_bar);
    for (; l < n_bar; ++l) { gen(l); __syncthreads(); }   // retired: this wave has no item in any remaining barrier level
}

template <typename T, bool DIAG, bool PK = false>
__global__ __launch_bounds__(sizeof(T) == 8 ? 512 : DIAG ? 768 : 1024) void k_sweep_philox(SweepArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ChainCtx x;
    chain_load(a, lds_raw, x);
    if ((unsigned)reinterpret_cast<size_t>(lds_raw) != 0u)   // low word of a generic LDS address = LDS offset
        __builtin_trap();      // update_spin addresses spins as LDS offset == column index
    const int n = x.n, tid = x.tid, nt = x.nt, c = x.c;
    T *ur = reinterpret_cast<T *>(lds_raw + a.lds_u_off);
    int *loff = reinterpret_cast<int *>(lds_raw + a.lds_loff_off);
    const uint32_t gc_chain = (uint32_t)(a.chain_base + c);
    const int row = a.slot_of_chain ? a.slot_of_chain[gc_chain] : c;
    const uint32_t gc = (a.rng_stride && a.slot_of_chain) ? (uint32_t)((c / a.rng_ladder_len) * a.rng_stride + a.rng_base + row) : gc_chain;
    const double esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale

#ifdef NLMC_STAMPS
    long long st_fill = 0, st_epi = 0;
    long long st_u[5] = {0, 0, 0, 0, 0};      // update_spin: lds-gather, field, decide, energy+write, calls
    x.st = st_u;
    x.lvl_t = (a.dbg && c == 0) ? a.dbg + (size_t)gridDim.x * 16 * 8 : nullptr;
    const long long st_begin = (long long)__builtin_readcyclecounter();
#endif
    T *const ur0 = ur;
    int *const loff0 = loff;
    const bool dbuf = a.lds_u_stride != 0;
    bool prefilled = false;
    for (int t = 0; t < a.n_sweeps; ++t) {
        const int oid = a.per_chain ? (x.ob * a.n_sweeps + t) : t;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * (n + 1);
        const int nl = a.nlev[oid];
        // "f32" mode: z = cb * (X 2^-qs) with the int32 field X -> fold the exact power of two into the coefficient
        const T cb0 = scale_cb((T)a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss], a.qinv);
        const T cb1 = scale_cb((T)a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss + 1], a.qinv);
        const uint32_t tt = a.sweep0 + (uint32_t)t;
        const int pb = dbuf ? (t & 1) : 0;
        ur = reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(ur0) + (size_t)pb * a.lds_u_stride);
        loff = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(loff0) + (size_t)pb * a.lds_loff_stride);
        const bool fast = nl < NLMC_LCAP;

        // sweep prologue (first sweep of a launch, or whenever the previous sweep could not prepare it):
        // uniforms for every spin (all lanes busy), level offsets into LDS
#ifdef NLMC_STAMPS
        const long long f0 = (long long)__builtin_readcyclecounter();
#endif
        if (!prefilled) {
            fill_uniforms(ur, n, tt, gc, a.seed_lo, a.seed_hi, tid, nt);
            if (fast) for (int l = tid; l <= nl; l += nt) loff[l] = off[l];
            __syncthreads();
        }
        prefilled = false;
#ifdef NLMC_STAMPS
        st_fill += (long long)__builtin_readcyclecounter() - f0;
#endif

        const size_t so = (size_t)oid;
        if (fast) {
            // levels that need a barrier: everything up to the last level wider than one wave
            int n_bar = 0;
            for (int l0 = 0; l0 < nl; l0 += 64) {
                const int l = l0 + (tid & 63);
                const unsigned long long m = __ballot((l < nl) && (loff[l + 1] - loff[l] > 64));
                if (m) n_bar = l0 + 64 - __clzll((long long)m);
            }
            // Rows longer than 8 entries sit at the front of their level (k_levelize), at most hi_max of them: only the
            // waves whose lanes can hold one run the variant that also prefetches the second half of the row window.
            // Two copies of the level loop, chosen per wave and sweep, keep the number of loads per stage STATIC in
            // each copy (so hipcc emits counted vmcnt waits); both copies execute the same n_bar barriers.
            const bool role_long = (Pf<T, PK>::W > 8) && ((tid & ~63) < a.hi_max[oid]);
            if (role_long) run_levels<T, DIAG, true, false, NoGen, PK>(a, x, ur, loff, so, nl, n_bar, cb0, cb1, esc);
            else run_levels<T, DIAG, false, false, NoGen, PK>(a, x, ur, loff, so, nl, n_bar, cb0, cb1, esc);

            // While wave 0 finishes the narrow tail of this sweep, the other waves prepare the next one in the second
            // set of LDS buffers: its uniforms (the Philox work of a whole sweep) and its level offsets.
            if (dbuf && t + 1 < a.n_sweeps) {
                const int oid_n = a.per_chain ? (x.ob * a.n_sweeps + t + 1) : (t + 1);
                const int nl_n = a.nlev[oid_n];
                const bool wave0_busy = n_bar < nl;
                if (nl_n < NLMC_LCAP && (!wave0_busy || nt > 64)) {
                    const int hid = wave0_busy ? tid - 64 : tid, hcnt = wave0_busy ? nt - 64 : nt;
                    if (hid >= 0) {
                        T *ur_n = reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(ur0) + (size_t)(pb ^ 1) * a.lds_u_stride);
                        int *loff_n = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(loff0) + (size_t)(pb ^ 1) * a.lds_loff_stride);
                        const int32_t *__restrict__ off_n = a.lvl_off + (size_t)oid_n * (n + 1);
                        fill_uniforms(ur_n, n, tt + 1u, gc, a.seed_lo, a.seed_hi, hid, hcnt);
                        for (int l = hid; l <= nl_n; l += hcnt) loff_n[l] = off_n[l];
                    }
                    prefilled = true;
                }
            }
        } else {
            // very deep schedules (dense graphs): plain level loop
            for (int l = 0; l < nl; ++l) {
                const int lo = off[l], hi = off[l + 1];
                for (int i = lo + tid; i < hi; i += nt) {
                    Pf<T, PK> pe;
                    typename Pf<T, PK>::View vw;
                    vw.bind(a, so, n);
                    pe.template issue<true>(vw, i, true);
                    update_spin<T, DIAG, true, false, PK>(a, x, ur, pe, so, i, cb0, cb1, esc);
                }
                __syncthreads();
            }
        }
#ifdef NLMC_STAMPS
        const long long e0 = (long long)__builtin_readcyclecounter();
#endif
        __syncthreads();        // the solo tail of wave 0 and the next sweep's preparation are complete
        sweep_epilogue(a, x, t);
#ifdef NLMC_STAMPS
        st_epi += (long long)__builtin_readcyclecounter() - e0;
#endif
    }
#ifdef NLMC_STAMPS
    if (a.dbg && (tid & 63) == 0) {
        long long *d = a.dbg + ((size_t)c * 16 + (tid >> 6)) * 8;
        d[0] = st_u[0]; d[1] = st_u[1]; d[2] = st_u[2]; d[3] = st_u[3]; d[4] = st_u[4]; d[5] = st_fill;
        d[6] = (long long)__builtin_readcyclecounter() - st_begin; d[7] = st_epi;
    }
#endif
    chain_store(a, x);
}

// ---- fused-window sweeps ----------------------------------------------------------------------------------
// All sweeps of a launch share ONE level list (k_levelize_fused): an update of sweep t+1 is scheduled as soon as the
// updates it depends on -- earlier neighbours of sweep t+1, every neighbour's and its own update of sweep t -- are
// done, so the narrow tail of sweep t overlaps the wide head of sweep t+1 and every level is ~0.07 n wide.  The
// sequential semantics (and therefore every result bit) are those of the sweep-by-sweep schedule.  At most two sweeps
// are live in any level (k_levelize_fused floors sweep t+2 behind the end of sweep t); thresholds live in three LDS
// tables (slot = t mod 3): while sweeps t, t+1 run, the table of sweep t+2 is produced (its slot was freed when sweep
// t-1 ended).
//
// The level loop is the hot loop of the whole path and is written for instruction count: a level is a list of CHUNKS
// of 64 items (padded with dummy items), wave w of the f_workers worker waves takes chunk w of the level if there is
// one -- no per-lane validity, no divergent branches; the chunk's position rides in the scalar offset of the buffer
// loads (one 8-byte head + 2 or 4 16-byte planes, fully coalesced, issued two levels ahead into rotating registers);
// the update is straight-line: 10 LDS reads, 8 multiply-adds on the int32 field, one compare with the prepared
// threshold, one 64-bit multiply-add for the energy, one LDS write, one s_barrier.  Waves [f_gen0, nt/64) produce the
// thresholds (Philox + logit), the waves behind the workers also pull the next window's schedule towards the chip.
// No per-sweep epilogue: used when the caller wants neither per-sweep energies, nor recorded configurations, nor the
// running minimum, and beta is constant.
// Schedule entry formats of a fused plan (8 row entries per position):
//   NLMC_FMT_WIDE    { col, Jq } 8 bytes per entry, 4 planes of 16 B   (any couplings)
//   NLMC_FMT_COMPACT col << 16 | (Jq & 0xFFFF), 2 planes               (every |Jq| < 2^15)
//   NLMC_FMT_ADDR    a 16-bit LDS ADDRESS per entry, 1 plane           (every Jq = +-1): the workgroup keeps a negated copy of
//                    its spins behind the spins; an entry with Jq = +1 holds the address of s_j, one with Jq = -1 the
//                    address of -s_j, a padding entry the address of a byte that is always 0 -- the local field is the plain
//                    sum of the gathered bytes (no unpacking, no multiply), every update writes s and -s.
// The vector-memory path of the CU is what a level's loads queue on (24 instead of 40 bytes per position measured 9 %
// off the launch), and the update of the address format is 11 instructions shorter.
template <int FMT> struct FusedItem {
    static constexpr int NE = NLMC_FZ_W;
    static constexpr int NP = FMT == NLMC_FMT_WIDE ? NE / 2 : FMT == NLMC_FMT_COMPACT ? NE / 4 : NE / 8;
    nlmc_i2 hd;                   // { k | LONG << 14 | PAIR << 15 | threshold word << 16, hq_k (+ sign-format constant) }
    nlmc_i4 pk[NP];
    __device__ __forceinline__ int word(int i) const { const int p = i >> 2, j = i & 3; return j == 0 ? pk[p].x : j == 1 ? pk[p].y : j == 2 ? pk[p].z : pk[p].w; }
    // LDS address of the byte entry q gathers (wide / compact: the neighbour's spin)
    __device__ __forceinline__ unsigned col(int q) const
    {
        if (FMT == NLMC_FMT_WIDE) return (unsigned)word(2 * q);
        if (FMT == NLMC_FMT_COMPACT) return (unsigned)word(q) >> 16;
        return (q & 1) ? (unsigned)word(q >> 1) >> 16 : (unsigned)word(q >> 1) & 0xFFFFu;
    }
    __device__ __forceinline__ int val(int q) const       // (wide / compact only)
    {
        if (FMT == NLMC_FMT_WIDE) return word(2 * q + 1);
        return (int)(short)(word(q) & 0xFFFF);
    }
};

__device__ __forceinline__ float4 thresholds4(const u32x4 &r)
{
    float4 v;
    v.x = threshold_spec(r.x);
    v.y = threshold_spec(r.y);
    v.z = threshold_spec(r.z);
    v.w = threshold_spec(r.w);
    return v;
}

// Threshold producer of one wave.  Block b of sweep u (4 thresholds from one Philox call) belongs to lane b mod gnt of
// the producing lanes.  The table slot u mod 3 is free once sweep u-3 has ended (level send[u-3]; sweep 2: from the
// start) and must be complete before the first item of sweep u, which k_levelize_fused places after send[u-2].  A call
// is cut into TWO steps -- the Philox rounds, then the four logits and the store -- and the 2 nj steps of a lane are
// spread evenly over that range of levels: integer multiplies are slow on this chip and a level lasts as long as its
// slowest wave.  The state is a set of LOCAL variables of the function that runs the level loop (a struct handed
// around by reference ended up in scratch memory: two global-memory round trips per level).
struct FusedGenParams { uint32_t gc; int gtid, gnt, nblk, nj, Tn; };
// What changes from one planned window to the next (everything else of a launch is in SweepArgs): k_sweep_fused runs one window,
// k_rounds_fused a run of consecutive ones.
struct FusedWin {
    const int32_t *lvl_off;       // [nl + 1] chunk offsets of the levels
    const int32_t *fsend;         // [T] last level of every sweep
    const EdgeQ *ell;             // row-entry planes
    const int2 *head;             // item heads
    const int2 *warm_head;        // the NEXT window's arrays (or nullptr)
    const EdgeQ *warm_ell;
    int npos_next;
    int nl, hi_max;
    uint32_t sweep0;              // global index of the window's first sweep
};
#ifdef NLMC_DEBUG_KNOBS
#define NLMC_GEN_DBG_OFF_COND (a.dbg_flags & 1)
#define NLMC_GEN_DBG_PHILOX if (a.dbg_flags & 8) g_r = u32x4{(uint32_t)b * 2654435761u, (uint32_t)b ^ gp.gc, (uint32_t)g_u * 40503u + (uint32_t)b, ~(uint32_t)b}; else
#define NLMC_GEN_DBG_LOGIT if (a.dbg_flags & 16) dst[b] = nlmc_f4{__uint_as_float(g_r.x & 0x3FFFFFFFu), __uint_as_float(g_r.y & 0x3FFFFFFFu), __uint_as_float(g_r.z & 0x3FFFFFFFu), __uint_as_float(g_r.w & 0x3FFFFFFFu)}; else
#else
#define NLMC_GEN_DBG_OFF_COND false
#define NLMC_GEN_DBG_PHILOX
#define NLMC_GEN_DBG_LOGIT
#endif
// (plain macros over local variables: with lambdas the captured state was kept in scratch memory)
/* (the launch constants the producer needs are read ONCE, up front: inside the level loop they sit behind conditions, and a load
   through the persistent kernel's argument pointer is not hoisted out of a condition -- 27 instructions and a scalar-memory
   wait per stage) */
#define NLMC_GEN_STATE int g_u = 2, g_slot = 2, g_w0 = 0, g_wend = 0, g_wlen = 1, g_acc = 0, g_sidx = 0; u32x4 g_r{0u, 0u, 0u, 0u}; \
    const uint32_t g_seed_lo = a.seed_lo, g_seed_hi = a.seed_hi, g_sweep0 = W.sweep0; const int g_u_off = a.lds_u_off, g_u_stride = a.lds_u_stride; \
    const int32_t *const g_fsend = W.fsend;
/* steps per call: Philox rounds 0-4 | rounds 5-9 | four logits + store  (fp64 mode: rounds 0-4 | rounds 5-9 + store of the four
   27-bit high words: one call serves 4 spins in both modes, the fp64 mode has no logit to evaluate) */
#define NLMC_GEN_NSTEP (g_f64 ? 2 : 3)
#define NLMC_GEN_ARM(a, gp)                                                                                             \
    {                                                                                                                   \
        typedef const int32_t __attribute__((address_space(4))) *const_i32_;                                            \
        g_sidx = 0; g_acc = 0;                                                                                          \
        if (g_u < gp.Tn) {                                                                                              \
            const const_i32_ send_ = (const_i32_)(uintptr_t)g_fsend;                                                    \
            g_w0 = g_u >= 3 ? __builtin_amdgcn_readfirstlane(send_[g_u - 3]) : -1;                                      \
            g_wend = __builtin_amdgcn_readfirstlane(send_[g_u - 2]);                                                    \
            g_wlen = max(1, g_wend - g_w0);              /* levels (w0, wend] are the production window of sweep u */   \
        } else g_w0 = 0x7FFFFFFF;                                                                                       \
    }
#define NLMC_GEN_STEP(a, gp)                                                                                            \
    {                                                                                                                   \
        const int call_ = g_sidx / NLMC_GEN_NSTEP, ph_ = g_sidx - call_ * NLMC_GEN_NSTEP;                               \
        const int b = gp.gtid + call_ * gp.gnt;                                                                         \
        if (ph_ == 0) {                                                                                                 \
            NLMC_GEN_DBG_PHILOX                                                                                         \
            g_r = philox4x32_rounds(u32x4{(uint32_t)b, g_sweep0 + (uint32_t)g_u, gp.gc, NLMC_TAG_UNIFORM}, g_seed_lo, g_seed_hi, 0, 5); \
        } else if (g_f64) {                                                                                             \
            typedef nlmc_i4 __attribute__((address_space(3))) *lds_i4_;                                                 \
            g_r = philox4x32_rounds(g_r, g_seed_lo, g_seed_hi, 5, 5);                                                   \
            if (b < gp.nblk) ((lds_i4_)(uintptr_t)(unsigned)(g_u_off + g_slot * g_u_stride))[b] = nlmc_i4{(int)(g_r.x >> 5), (int)(g_r.y >> 5), (int)(g_r.z >> 5), (int)(g_r.w >> 5)}; \
        } else if (ph_ == 1) {                                                                                          \
            NLMC_GEN_DBG_PHILOX                                                                                         \
            g_r = philox4x32_rounds(g_r, g_seed_lo, g_seed_hi, 5, 5);                                                   \
        } else if (b < gp.nblk) {                                                                                       \
            typedef nlmc_f4 __attribute__((address_space(3))) *lds_f4_;    /* LDS offsets, no generic pointers */         \
            const lds_f4_ dst = (lds_f4_)(uintptr_t)(unsigned)(g_u_off + g_slot * g_u_stride);                  \
            NLMC_GEN_DBG_LOGIT                                                                                          \
            { const float4 t4 = thresholds4(g_r); dst[b] = nlmc_f4{t4.x, t4.y, t4.z, t4.w}; }                           \
        }                                                                                                               \
        ++g_sidx;                                                                                                       \
    }
/* the NLMC_GEN_NSTEP * nj steps of a lane are spread EVENLY over the levels of the production window (an accumulator in
   the manner of a line-drawing algorithm): a level lasts as long as its slowest wave */
#define NLMC_GEN(a, gp, l)                                                                                              \
    if (!(NLMC_GEN_DBG_OFF_COND) && (l) > g_w0) {                                                                       \
        g_acc += NLMC_GEN_NSTEP * gp.nj;                                                                                \
        while (g_acc >= g_wlen) { g_acc -= g_wlen; if (g_sidx < NLMC_GEN_NSTEP * gp.nj) NLMC_GEN_STEP(a, gp) }          \
        if (__builtin_expect((l) == g_wend, 0)) {                                                                       \
            while (g_sidx < NLMC_GEN_NSTEP * gp.nj) NLMC_GEN_STEP(a, gp)                                                \
            ++g_u; g_slot = g_slot == 2 ? 0 : g_slot + 1; NLMC_GEN_ARM(a, gp)                                           \
        }                                                                                                               \
    }

// OUT: per-sweep outputs (energy trace, running minimum + argmin state, recorded configurations) and a temperature per
// sweep.  In a fused window at most two sweeps are live in a level, the OLDER one (index o_t) and the one after it; an
// item belongs to the older one iff its threshold word lies in that sweep's table.  Every update also writes its new
// spin to a snapshot slot of its sweep (three slots like the threshold tables: the snapshot of sweep t is complete at
// level send[t] and not overwritten before sweep t+3 starts), energy deltas go to the accumulator of their sweep, and
// when a sweep ends its sum is reduced over the workgroup (one LDS atomic per wave, read after the level's barrier):
// that gives E after every sweep, the strict running minimum (first argmin, like np.argmin) and, from the snapshot,
// the argmin / recorded states -- NMC/nmc.py:386-395 -- without giving up the overlap of consecutive sweeps.
template <bool DIAG, bool FLAGS, bool PAIR, bool GEN, int FMT, bool OUT = false, bool F64 = false>
__device__ __forceinline__ void fused_levels(const SweepArgs &a, const FusedWin &W, unsigned char *lds_raw, int wv, int lane, int nl, float cq0,
                                             float cq1, long long &e_loc, const FusedGenParams gp)
{
    constexpr bool g_f64 = F64;
    static_assert(!(F64 && FLAGS), "fused fp64 windows: plain chains only (a scaled row's field is not an exact integer)");
    NLMC_GEN_STATE
    NLMC_GEN_ARM(a, gp)
    typedef const int32_t __attribute__((address_space(4))) *const_i32o;
    typedef const double __attribute__((address_space(4))) *const_f64o;
    const int o_npad = a.g.n_pad, o_b = blockIdx.x, o_c = a.chain_list ? a.chain_list[o_b] : o_b, o_tid = wv * 64 + lane;
    int o_t = 0, o_end = 0;                                // older live sweep and its last level
    unsigned o_lo = 0u;                                    // its threshold-word range is [o_lo, o_lo + n_pad)
    long long e_new = 0, E_run = 0, E_min = 0;
    int a_min = 0;
    float cqo0 = cq0, cqo1 = cq1, cqn0 = cq0, cqn1 = cq1;  // coefficients of the older / the newer live sweep
    const int o_row = a.slot_of_chain ? a.slot_of_chain[a.chain_base + o_c] : o_c;
    // Snapshot ring in global memory (wave-uniform pointer): every update stores its new spin there (a scattered byte store,
    // fire and forget); at the end of a sweep every storing wave drains its stores before the level's barrier, and the copy
    // to `best` / the recorded trace reads the ring past the CU's vector L1 (lines of the slot's previous use may linger there).
    int8_t *const snapg = (OUT && a.snap_g) ? a.snap_g + (size_t)o_b * (3 * (size_t)o_npad + 16) : nullptr;
#define NLMC_OCQ(t, j) ((float)((const_f64o)(uintptr_t)a.tab)[(size_t)o_row * a.tab_cs + (size_t)min((t), a.n_sweeps - 1) * a.tab_ss + (j)] * a.qinv)
    // (wave-uniform 64-bit values are pinned into scalar registers: a uniform value in a vector register costs 64 lanes, a
    // spilled scalar one lane)
    if (OUT) {
        o_end = ((const_i32o)(uintptr_t)W.fsend)[0];
        E_run = uniform64(a.efix[o_c]);
        E_min = a.emin ? uniform64(a.emin[o_c]) : 0x7FFFFFFFFFFFFFFFll;
        a_min = a.emin ? __builtin_amdgcn_readfirstlane(a.argmin[o_c]) : 0;
        cqn0 = NLMC_OCQ(1, 0); cqn1 = NLMC_OCQ(1, 1);
    }
    typedef FusedItem<FMT> Item;
    // launch constants of the update kept in VECTOR registers (there are spare ones; the scalar file is what spills here:
    // every spilled scalar is re-read with a v_readlane in every level)
    unsigned v_u_off = (unsigned)a.lds_u_off, v_neg_off = (unsigned)a.lds_neg_off;
    int v_eshift = a.eshift;
    asm volatile("" : "+v"(v_u_off), "+v"(v_neg_off), "+v"(v_eshift));
    // fp64 mode: LDS address of Khi[X = 0] (entry X sits 4 X bytes from it), bytes from a Khi entry to its Klo entry
    unsigned v_kt0 = (unsigned)(a.lds_kt_off + 4 * a.f64_xmax), v_tie = a.f64_tie_mask;
    if (F64) asm volatile("" : "+v"(v_kt0), "+v"(v_tie));
    typedef const unsigned __attribute__((address_space(3))) *lds_u32;
    constexpr int NP = Item::NP, NE = Item::NE;
    const int plane_bytes = a.fz_pstride * 16;
    const __amdgpu_buffer_rsrc_t r_ell = __builtin_amdgcn_make_buffer_rsrc(const_cast<EdgeQ *>(W.ell), 0, (NLMC_FZ_W / 2) * plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_head = __builtin_amdgcn_make_buffer_rsrc(const_cast<int2 *>(W.head), 0, a.fz_pstride * 8, 0x00020000);
    // level offsets in chunks, read with SCALAR loads (uniform index): no VGPR, no VALU, no vector-memory slot
    // (constant address space: the plan is read-only while sweep kernels run, and a uniform index then gives s_load_dword;
    // a register-resident window of offsets picked with v_readlane measured slower)
    typedef const int32_t __attribute__((address_space(4))) *const_i32;
    const const_i32 loffp = (const_i32)(uintptr_t)W.lvl_off;
    auto lo = [&](int i) __attribute__((always_inline)) { return loffp[min(i, nl)]; };
    const int lane16 = lane * 16, oob = 0x7FF00000;        // oob: past both buffers (the hardware range check drops the load)
    typedef const int8_t __attribute__((address_space(3))) *lds_i8;
    typedef int8_t __attribute__((address_space(3))) *lds_i8w;
    typedef const uint8_t __attribute__((address_space(3))) *lds_u8;
    typedef const float __attribute__((address_space(3))) *lds_f32;

    // chunk `c` of the plan -> registers; has == false: every lane out of range (no memory traffic, registers = 0)
    auto issue = [&](Item &it, int c, bool has) __attribute__((always_inline)) {
#ifdef NLMC_DEBUG_KNOBS
        if (a.dbg_flags & 32) return;                // timing experiment: no vector-memory instructions at all
#endif
        const int v16 = has ? lane16 : oob;
        it.hd = __builtin_amdgcn_raw_buffer_load_b64(r_head, v16 >> 1, c * 512, 0);
#pragma unroll
        for (int q = 0; q < NP; ++q) it.pk[q] = __builtin_amdgcn_raw_buffer_load_b128(r_ell, v16, c * 1024 + q * plane_bytes, 0);
    };
    auto update = [&](const Item &it, int lv) __attribute__((always_inline)) {
        (void)lv;
#ifdef NLMC_DEBUG_KNOBS
        if (a.dbg_flags & 2) { asm volatile("" :: "v"(it.hd.x), "v"(it.pk[0].x), "v"(it.pk[Item::NP - 1].x)); return; }   // timing experiment: loads only
#endif
        const int hx = it.hd.x;
#ifdef NLMC_DEBUG_KNOBS
        const unsigned ka = (a.dbg_flags & 1024) ? (unsigned)((lane & 31) * 4 + (lane >> 5)) : (unsigned)hx & 0x3FFFu;
        const float wk = *(lds_f32)(uintptr_t)(((a.dbg_flags & 1024) ? (unsigned)(lane * 4) : (((unsigned)hx >> 16) << 2)) + (unsigned)a.lds_u_off);
#else
        const unsigned ka = (unsigned)hx & 0x3FFFu;                               // LDS address of the spin
        const float wk = *(lds_f32)(uintptr_t)((((unsigned)hx >> 16) << 2) + v_u_off);
#endif
        int so = (int)*(lds_i8)(uintptr_t)ka;
        unsigned f = 0u;
        if (FLAGS) f = (unsigned)*(lds_u8)(uintptr_t)(ka + (unsigned)a.lds_flags_off);
        int sj[NE];
#ifdef NLMC_DEBUG_KNOBS
        if (a.dbg_flags & 512) {                     // timing experiment: conflict-free gathers (wrong results)
#pragma unroll
            for (int q = 0; q < NE; ++q) sj[q] = (int)*(lds_i8)(uintptr_t)(unsigned)((lane & 31) * 4 + q * 132 + (lane >> 5));
        } else
#endif
        {
#pragma unroll
            for (int q = 0; q < NE; ++q) sj[q] = (int)*(lds_i8)(uintptr_t)it.col(q);   // all reads in flight
        }
        // (pins the read of the spin's own value into the batch of the gathers: left alone, the scheduler sinks it behind
        // the field sum -- one more LDS round trip on the dependent chain of every level)
        asm volatile("" : "+v"(so));
        int X0 = it.hd.y, X1 = 0, Xd = 0;
#pragma unroll
        for (int q = 0; q < NE; q += 2) {
            if (FMT == NLMC_FMT_ADDR) {             // the gathered byte is Jq s already
                X0 += sj[q];
                X1 += sj[q + 1];
            } else {
                X0 += __mul24(it.val(q), sj[q]);
                X1 += __mul24(it.val(q + 1), sj[q + 1]);
            }
        }
        if (DIAG) {
#pragma unroll
            for (int q = 0; q < NE; ++q) {
                if (FMT == NLMC_FMT_ADDR) Xd += (it.col(q) == ka || it.col(q) == ka + v_neg_off) ? sj[q] : 0;
                else Xd += (it.col(q) == ka) ? __mul24(it.val(q), sj[q]) : 0;
            }
        }
        int X = X0 + X1;
        bool second = false;
        if (PAIR) {
            // chunks at the head of a level may hold PAIRS: a row of 9-16 entries on an even / odd lane pair, eight entries
            // each; both lanes end up with the whole field and take the same decision, the energy counts once.  A row longer
            // than that: the rest comes from the CSR arrays, in the first lane of the pair.
            if (__builtin_amdgcn_ballot_w64((hx & 0x4000) != 0) != 0ull) {
                if (hx & 0x4000) {
                    const int rs = a.g.rowptr[ka], re = a.g.rowptr[ka + 1];
#pragma clang loop vectorize(disable) unroll(disable)
                    for (int e = rs + 2 * NLMC_FZ_W; e < re; ++e) {
                        const EdgeQ t = a.g.edge32[e];
                        const int pr = __mul24(t.q, (int)*(lds_i8)(uintptr_t)(unsigned)t.col);
                        X += pr;
                        if (DIAG && (unsigned)t.col == ka) Xd += pr;
                    }
                }
            }
            const bool pair = (hx & 0x8000) != 0;
            const int Xp = __builtin_amdgcn_mov_dpp(X, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]: the neighbour lane
            X += pair ? Xp : 0;
            if (DIAG) { const int Xdp = __builtin_amdgcn_mov_dpp(Xd, 0xB1, 0xF, 0xF, true); Xd += pair ? Xdp : 0; }
            second = pair && (lane & 1);
        }
        // fp64 mode: s' = +1 iff the 53-bit integer of the update's uniform lies below K(X) (accept_count_spec).  The table of
        // the sweep holds the 27 high bits; they decide unless they EQUAL the high word of K(X) (probability 2^-27 per
        // update: ~0.2 updates per launch of the bench shape) -- then the low 26 bits are made again with one Philox call.
        bool up64 = false;
        if (F64) {
            const unsigned hk = __float_as_uint(wk);
            const unsigned kaddr = ((unsigned)X << 2) + v_kt0;
#ifdef NLMC_DEBUG_KNOBS
            // 16384: timing experiment -- no dependent table read behind the field sum (wrong results)
            const unsigned kh = (a.dbg_flags & 16384) ? (unsigned)(X * 3000000 + 0x4000000) : *(lds_u32)(uintptr_t)kaddr;
#else
            const unsigned kh = *(lds_u32)(uintptr_t)kaddr;
#endif
            up64 = hk < kh;
            const bool tie = ((hk ^ kh) & v_tie) == 0u;
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(tie) != 0ull, 0)) {
                if (tie) {
                    // sweep of the item: the first t = slot (mod 3) whose last level is not before this one (sweep t + 3
                    // starts behind the end of sweep t + 1); a dummy item (slot 3) gets an arbitrary draw for its scratch spin
                    typedef const int32_t __attribute__((address_space(4))) *const_i32s;
                    const unsigned tw_ = (unsigned)hx >> 16, np_ = (unsigned)a.g.n_pad;
                    int t_ = tw_ >= 3u * np_ ? 3 : tw_ >= 2u * np_ ? 2 : tw_ >= np_ ? 1 : 0;
                    while (t_ + 3 < a.n_sweeps && lv > ((const_i32s)(uintptr_t)W.fsend)[t_]) t_ += 3;
                    const u32x4 r_ = philox4x32_10(ka >> 2, W.sweep0 + (uint32_t)t_, gp.gc, NLMC_TAG_UNIFORM_LO, a.seed_lo, a.seed_hi);
                    const unsigned lw_ = (ka & 2u) ? ((ka & 1u) ? r_.w : r_.z) : ((ka & 1u) ? r_.y : r_.x);
                    const unsigned lo_ = lw_ >> 6;
                    const unsigned kl = *(lds_u32)(uintptr_t)(kaddr + 4u * (unsigned)(2 * a.f64_xmax + 1));
                    up64 = hk < kh || (hk == kh && lo_ < kl);
                }
            }
        }
        if (!OUT) {
            const float z = ((FLAGS && f == 1u) ? cq1 : cq0) * (float)X;
            int sn = F64 ? (up64 ? 1 : -1) : (z < wk) ? 1 : -1;
            if (FLAGS) sn = (f >= 2u) ? so : sn;                                    // frozen: unchanged
            const int cv = (PAIR && second) ? 0 : (so - sn) << v_eshift;
            e_loc += (long long)(DIAG ? X - Xd : X) * (long long)cv;
            *(lds_i8w)(uintptr_t)ka = (int8_t)sn;
            if (FMT == NLMC_FMT_ADDR) *(lds_i8w)(uintptr_t)(ka + v_neg_off) = (int8_t)-sn;
        } else {
            const unsigned tw = (unsigned)hx >> 16;                                 // threshold word = slot * n_pad + k
            const bool is_old = tw - o_lo < (unsigned)o_npad;
            const float cqa = is_old ? cqo0 : cqn0, cqb = is_old ? cqo1 : cqn1;
            const float z = ((FLAGS && f == 1u) ? cqb : cqa) * (float)X;
            int sn = F64 ? (up64 ? 1 : -1) : (z < wk) ? 1 : -1;
            if (FLAGS) sn = (f >= 2u) ? so : sn;
            const int cv = (PAIR && second) ? 0 : (so - sn) << v_eshift, cvo = is_old ? cv : 0;
            const long long Xt = (long long)(DIAG ? X - Xd : X);
            e_loc += Xt * (long long)cvo;                                           // e_loc: the older sweep's deltas
            e_new += Xt * (long long)(cv - cvo);
            *(lds_i8w)(uintptr_t)ka = (int8_t)sn;
            if (FMT == NLMC_FMT_ADDR) *(lds_i8w)(uintptr_t)(ka + v_neg_off) = (int8_t)-sn;
            if (snapg) snapg[tw] = (int8_t)sn;                                      // snapshot slot of the update's sweep:
            else *(lds_i8w)(uintptr_t)(tw + (unsigned)a.lds_snap_off) = (int8_t)sn; // global ring (large n) or LDS
        }
    };
    // End of the older live sweep: before the level's barrier every wave adds its share to the sweep's LDS accumulator
    // (NLMC_OUT_PRE); after it every worker thread knows the energy after that sweep (NLMC_OUT_POST).  Plain macros over
    // local variables: lambdas that capture this much state end up in scratch memory.
#define NLMC_OUT_PRE                                                                                                    \
    {                                                                                                                   \
        if (snapg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* this wave's snapshot stores have landed */     \
        const long long w_ = wave_sum_i64(e_loc);                                                                       \
        long long *red_ = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);                                       \
        if (lane == 0 && w_ != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red_[o_t % 3]), (unsigned long long)w_); \
    }
#define NLMC_OUT_POST                                                                                                   \
    {                                                                                                                   \
        long long *red_ = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);                                       \
        const int slot_ = o_t % 3, tg_ = a.t0 + o_t;                                                                    \
        E_run += uniform64(red_[slot_]);                                                                                \
        const bool better_ = a.emin && E_run < E_min && (tg_ % a.min_stride == 0);   /* strict <: first argmin (np.argmin, NMC/nmc.py:394) */ \
        if (better_) { E_min = E_run; a_min = tg_; }                                                                    \
        if (o_tid == 0) {                                                                                               \
            red_[(o_t + 2) % 3] = 0;                          /* read a sweep ago, next used two sweeps from now */      \
            if (a.etrace) a.etrace[(size_t)o_b * a.trace_sweeps + tg_] = E_run;                                         \
        }                                                                                                               \
        const unsigned char *snap_ = snapg ? reinterpret_cast<const unsigned char *>(snapg) + (size_t)slot_ * o_npad              \
                                           : lds_raw + a.lds_snap_off + (size_t)slot_ * o_npad;                         \
        if (better_ && a.best) {                                                                                        \
            int4 *dst_ = reinterpret_cast<int4 *>(a.best + (size_t)o_c * o_npad);                                       \
            const int4 *src_ = reinterpret_cast<const int4 *>(snap_);                                                   \
            if (snapg) for (int i_ = o_tid; i_ < o_npad / 16; i_ += a.f_workers * 64)                                  \
                reinterpret_cast<nlmc_i4 *>(dst_)[i_] = __builtin_nontemporal_load(&reinterpret_cast<const nlmc_i4 *>(src_)[i_]);   \
            else for (int i_ = o_tid; i_ < o_npad / 16; i_ += a.f_workers * 64) dst_[i_] = src_[i_];                    \
        }                                                                                                               \
        if (a.strace && tg_ % a.rec_stride == 0) {            /* M[:, ::M_skip]  (NMC/nmc.py:390) */                     \
            const int n_rec_ = (a.trace_sweeps + a.rec_stride - 1) / a.rec_stride;                                      \
            int8_t *dst_ = a.strace + ((size_t)o_b * n_rec_ + (size_t)(tg_ / a.rec_stride)) * a.g.n;                   \
            if (snapg) for (int i_ = o_tid; i_ < a.g.n; i_ += a.f_workers * 64) dst_[i_] = (int8_t)__builtin_nontemporal_load(&snap_[i_]); \
            else for (int i_ = o_tid; i_ < a.g.n; i_ += a.f_workers * 64) dst_[i_] = (int8_t)snap_[i_];                 \
        }                                                                                                               \
        e_loc = e_new; e_new = 0;                                                                                       \
        ++o_t;                                                                                                          \
        o_lo = (unsigned)((o_t % 3) * o_npad);                                                                          \
        o_end = o_t < a.n_sweeps ? ((const_i32o)(uintptr_t)W.fsend)[o_t] : 0x7FFFFFFF;                                   \
        cqo0 = cqn0; cqo1 = cqn1;                                                                                       \
        cqn0 = NLMC_OCQ(o_t + 1, 0); cqn1 = NLMC_OCQ(o_t + 1, 1);                                                       \
    }

    // Order inside a level: update first, THEN loads.  The vector-memory path of the CU is the scarcest resource of
    // this loop (a 1 KB wave load occupies it ~16 cycles): a wave that issues its loads first blocks on the full queue
    // with its LDS reads still behind them (measured: 245 vs 184 us per launch).  The loads issued in level l are
    // those of level l+2 (three register sets, rotated by a 3x unrolled loop), so that they have the whole of level
    // l+1 to arrive; every wave issues the same number of loads in every level (all lanes out of range when it has no
    // chunk: the hardware drops them), which keeps the counted vmcnt waits static.
#ifdef NLMC_STAMPS
    long long sw0, sw1, sw2, st_work = 0, st_bar = 0, st_calls = 0;
#endif
    {
        Item I0, I1, I2;
        int r0 = lo(0), r1 = lo(1), r2 = lo(2), r3 = lo(3), r4 = lo(4), r5 = lo(5);    // offsets of levels l .. l+5
        bool h0 = r0 + wv < r1, h1 = (1 < nl) && (r1 + wv < r2), h2;
        issue(I0, r0 + wv, h0);
        issue(I1, r1 + wv, h1);
#define NLMC_FSTAGE(lv, cur, hcur, nxt, hnxt, b2, b3)                                       \
        {                                                                                   \
            NLMC_FW0                                                                        \
            hnxt = ((lv) + 2 < nl) && ((b2) + wv < (b3));                                   \
            if (GEN) NLMC_GEN(a, gp, lv)                                                    \
            if (hcur) { update(cur, lv); NLMC_FCALL }                                       \
            issue(nxt, (b2) + wv, hnxt NLMC_DBG_NOLOAD);                                    \
            const bool end_ = OUT && __builtin_expect((lv) == o_end, 0);   /* rare: once per sweep */ \
            if (end_) NLMC_OUT_PRE                                                          \
            NLMC_FW1                                                                        \
            NLMC_DBG_BARRIER                                                                \
            NLMC_FW2                                                                        \
            if (end_) NLMC_OUT_POST                                                         \
        }
        for (int l = 0; l < nl; l += 3) {
            const int n3 = lo(l + 6), n4 = lo(l + 7), n5 = lo(l + 8);
            NLMC_FSTAGE(l, I0, h0, I2, h2, r2, r3)
            if (l + 1 < nl) NLMC_FSTAGE(l + 1, I1, h1, I0, h0, r3, r4)
            if (l + 2 < nl) NLMC_FSTAGE(l + 2, I2, h2, I1, h1, r4, r5)
            r0 = r3; r1 = r4; r2 = r5; r3 = n3; r4 = n4; r5 = n5;
        }
#undef NLMC_FSTAGE
    }
#ifdef NLMC_STAMPS
    if (a.dbg && lane == 0) {
        long long *d = a.dbg + ((size_t)blockIdx.x * 16 + wv) * 8;
        d[4] = st_calls; d[5] = st_work; d[6] = st_bar;
    }
#endif
    if (OUT && o_tid == 0) {                       // every sweep has ended: E_run is the energy of the final state
        a.efix[o_c] = E_run;
        if (a.energy_sink) a.energy_sink[o_c] = (double)E_run * __longlong_as_double((long long)(1023 - a.escale) << 52);
        if (a.emin) { a.emin[o_c] = E_min; a.argmin[o_c] = a_min; }
    }
}

// FMT is a parameter of the KERNEL (round 3; it was a run-time switch over three inlined copies of every level-loop
// variant: 12 copies per kernel, whose spilled scalars together took 19 vector registers of the per-sweep-output kernels)
// ---- the pieces of a fused-window launch: state in, one window, state out ---------------------------------------------------
template <bool FLAGS>
__device__ __forceinline__ void fused_state_load(const SweepArgs &a, unsigned char *lds_raw, int c)
{
    const int n_pad = a.g.n_pad, tid = threadIdx.x, nt = blockDim.x;
    long long *red = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);
    // spins (+ the scratch spin of the dummy items behind them), phase flags
    {
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (size_t)c * n_pad);
        int4 *dst = reinterpret_cast<int4 *>(lds_raw);
        for (int i = tid; i < n_pad / 16; i += nt) dst[i] = src[i];
        if (tid < 4) reinterpret_cast<int *>(lds_raw + n_pad)[tid] = tid < 2 ? 0x01010101 : 0;   // dummy spin | zero bytes (padding entries)
        if (a.lds_neg_off) {                                       // address format: -s behind s (0x01 <-> 0xFF)
            int4 *ndst = reinterpret_cast<int4 *>(lds_raw + a.lds_neg_off);
            for (int i = tid; i < n_pad / 16; i += nt) {
                const int4 v = src[i];
                ndst[i] = make_int4(v.x ^ (int)0xFEFEFEFE, v.y ^ (int)0xFEFEFEFE, v.z ^ (int)0xFEFEFEFE, v.w ^ (int)0xFEFEFEFE);
            }
            if (tid < 4) reinterpret_cast<int *>(lds_raw + a.lds_neg_off + n_pad)[tid] = (int)0xFFFFFFFF;
        }
        if (FLAGS) {
            const int4 *fsrc = reinterpret_cast<const int4 *>(a.flags + (size_t)c * n_pad);
            int4 *fdst = reinterpret_cast<int4 *>(lds_raw + a.lds_flags_off);
            for (int i = tid; i < n_pad / 16; i += nt) fdst[i] = fsrc[i];
            if (tid < 4) reinterpret_cast<int *>(lds_raw + a.lds_flags_off + n_pad)[tid] = 0;
        }
        if (tid < 4) red[tid] = 0;
    }
}

// One planned window on the chain whose state sits in LDS: threshold tables of its first two sweeps (+ the fp64 mode's K tables at
// the chain's temperature `row`), then the level loop.  Energy deltas of this thread come back in e_loc (plain variant).
template <bool DIAG, bool FLAGS, bool OUT, int FMT, bool F64>
__device__ __forceinline__ void fused_window(const SweepArgs &a, const FusedWin &W, unsigned char *lds_raw, int row, uint32_t gc, long long &e_loc)
{
    constexpr bool g_f64 = F64;
    const int n = a.g.n, tid = threadIdx.x, nt = blockDim.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float *ur = reinterpret_cast<float *>(lds_raw + a.lds_u_off);
    const int Tn = a.n_sweeps;
#ifdef NLMC_DEBUG_KNOBS
    const int nl = (a.dbg_flags & 128) ? 0 : W.nl;      // timing experiment: prologue + epilogue only
#else
    const int nl = W.nl;
#endif
    // "f32" mode: z = cb * (X 2^-qs) with the int32 field X -> the exact power of two is folded into the coefficient
    const float cq0 = (float)a.tab[(size_t)row * a.tab_cs] * a.qinv, cq1 = (float)a.tab[(size_t)row * a.tab_cs + 1] * a.qinv;
    // thresholds of the first two sweeps (the third table is produced inside the level loop like all later ones)
#ifdef NLMC_DEBUG_KNOBS
    // (timing experiments that switch the in-loop production off or down: every table holds valid words, so that stale entries
    // do not send the fp64 mode into its exact path)
    for (int t = 0; t < ((a.dbg_flags & 1) ? 3 : min(2, Tn)) NLMC_DBG_NOPROLOGUE; ++t) {
#else
    for (int t = 0; t < min(2, Tn) NLMC_DBG_NOPROLOGUE; ++t) {
#endif
        float *tab_t = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ur) + (size_t)t * a.lds_u_stride);
        if (F64) fill_uniform_words(reinterpret_cast<unsigned *>(tab_t), n, W.sweep0 + (uint32_t)t, gc, a.seed_lo, a.seed_hi, tid, nt);
        else fill_uniforms(tab_t, n, W.sweep0 + (uint32_t)t, gc, a.seed_lo, a.seed_hi, tid, nt);
    }
    if (F64) {
        // K(X) for every field value a row can reach, at this chain's temperature: z = cb (X 2^-qs) as in update_spin<double>.
        // The trailing lanes of the workgroup take it (they have one Philox call less than the others above).
        const int ne = 2 * a.f64_xmax + 1;
        unsigned *kt = reinterpret_cast<unsigned *>(lds_raw + a.lds_kt_off);
        const double cb = a.tab[(size_t)row * a.tab_cs];
        for (int i = nt - 1 - tid; i < ne; i += nt) {
            const double xf = (double)(i - a.f64_xmax) * a.qinv64;
            const unsigned long long K = accept_count_spec(cb * xf);
            kt[i] = (unsigned)(K >> 26);
            kt[ne + i] = (unsigned)(K & 0x3FFFFFFull);
        }
    }
    __syncthreads();

    const int g0 = a.f_gen0 * 64, nblk = (n + 3) / 4, gnt = nt - g0;
    // calls per sweep of THIS wave: block b = (tid - g0) + call * gnt must lie below nblk for at least one of its lanes
    // (wave-uniform; the waves at the end of the producing range do one call less when gnt does not divide nblk)
    const int gwave0 = __builtin_amdgcn_readfirstlane((tid - g0) & ~63);
    const FusedGenParams gp{gc, tid - g0, gnt, nblk, max(0, (nblk - gwave0 + gnt - 1) / gnt), Tn};
    const bool is_gen = tid >= g0;
    // The producing waves are the youngest of their SIMDs and would get the issue slots the older worker waves leave
    // over (measured: a 150-instruction call stretched to ~3000 cycles while the workers waited at the barrier).
    if (is_gen && a.f_gen_prio) __builtin_amdgcn_s_setprio(2);
    if (wv < a.f_workers) {
        const bool role_long = wv < W.hi_max;           // chunks that may hold lane PAIRS (rows longer than 8 entries) come first
        const int variant = (role_long ? 2 : 0) + (is_gen ? 1 : 0);
#define NLMC_FL(P, G) fused_levels<DIAG, FLAGS, P, G, FMT, OUT, F64>(a, W, lds_raw, wv, lane, nl, cq0, cq1, e_loc, gp)
        switch (variant) { case 0: NLMC_FL(false, false); break; case 1: NLMC_FL(false, true); break;
                           case 2: NLMC_FL(true, false); break; default: NLMC_FL(true, true); break; }
#undef NLMC_FL
    } else {
        // Waves without schedule items: their share of the thresholds, and they pull the NEXT window's schedule towards
        // the chip.  A window's schedule (72 B per update actually touched) is read once per launch and sits in HBM
        // when many windows were planned ahead; the workers prefetch one level ahead, which does not cover a cold miss
        // that every chain of an XCD then waits on.  The chains of a launch share the work: one dword per 128-byte line
        // of the head array and of planes 0-3, striped over chains and helper lanes, one load per level, retired a
        // level later.
        NLMC_GEN_STATE
        NLMC_GEN_ARM(a, gp)
        const int hid = tid - a.f_workers * 64, hcnt = nt - a.f_workers * 64;
        const unsigned warm_lines = W.warm_head ? (unsigned)(((size_t)W.npos_next * 8 + 127) / 128) : 0u;       // head
        const unsigned warm_lines_p = W.warm_head ? (unsigned)(((size_t)a.fz_pstride * 16 + 127) / 128) : 0u;     // lines per plane
        const unsigned warm_used_p = W.warm_head ? (unsigned)(((size_t)W.npos_next * 16 + 127) / 128) : 0u;     // touched lines per plane
        const unsigned warm_total = warm_lines + (FMT == NLMC_FMT_ADDR ? 1u : FMT == NLMC_FMT_COMPACT ? 2u : 4u) * warm_used_p;
        unsigned warm_at = (unsigned)blockIdx.x * (unsigned)hcnt + (unsigned)hid;
        const unsigned warm_step = gridDim.x * (unsigned)hcnt;
        // four touches in flight (a touch of a cold line takes longer than a level: waiting for the previous one every
        // level made these waves the last to reach the barrier)
        unsigned wv0 = 0u, wv1 = 0u, wv2 = 0u, wv3 = 0u;
        auto warm_addr = [&]() __attribute__((always_inline)) -> const unsigned * {
            // (always a load, so that the number of loads in flight is static: past the end, or with nothing to warm, the
            // lane re-reads a line of this window's own level offsets)
            const char *p = reinterpret_cast<const char *>(W.lvl_off);
            if (warm_at < warm_lines) p = reinterpret_cast<const char *>(W.warm_head) + (size_t)warm_at * 128;
            else if (warm_at < warm_total) {
                const unsigned r = warm_at - warm_lines, q = r / warm_used_p, i = r - q * warm_used_p;
                p = reinterpret_cast<const char *>(W.warm_ell) + ((size_t)q * warm_lines_p + i) * 128;
            }
            warm_at = warm_at < warm_total ? warm_at + warm_step : warm_at;
            return reinterpret_cast<const unsigned *>(p);
        };
#define NLMC_WARM(reg) { asm volatile("" :: "v"(reg) : "memory"); reg = *warm_addr(); }
#ifdef NLMC_STAMPS
        long long sw0, sw1, sw2, st_work = 0, st_bar = 0;
#endif
        for (int l = 0; l < nl; l += 4) {
            NLMC_FW0
            NLMC_WARM(wv0)
            if (is_gen) NLMC_GEN(a, gp, l)
            NLMC_FW1
            __syncthreads();
            NLMC_FW2
            if (l + 1 < nl) { NLMC_FW0 NLMC_WARM(wv1) if (is_gen) NLMC_GEN(a, gp, l + 1) NLMC_FW1 __syncthreads(); NLMC_FW2 }
            if (l + 2 < nl) { NLMC_FW0 NLMC_WARM(wv2) if (is_gen) NLMC_GEN(a, gp, l + 2) NLMC_FW1 __syncthreads(); NLMC_FW2 }
            if (l + 3 < nl) { NLMC_FW0 NLMC_WARM(wv3) if (is_gen) NLMC_GEN(a, gp, l + 3) NLMC_FW1 __syncthreads(); NLMC_FW2 }
        }
#ifdef NLMC_STAMPS
        if (a.dbg && lane == 0) {
            long long *d = a.dbg + ((size_t)blockIdx.x * 16 + wv) * 8;
            d[5] = st_work; d[6] = st_bar;
        }
#endif
        auto warm_next = [&]() __attribute__((always_inline)) { NLMC_WARM(wv0) };
        while (warm_at < warm_total) warm_next();     // few chains: the rest of this chain's share
        asm volatile("" :: "v"(wv0), "v"(wv1), "v"(wv2), "v"(wv3) : "memory");
#undef NLMC_WARM
    }

}

template <bool OUT>
__device__ __forceinline__ void fused_state_store(const SweepArgs &a, unsigned char *lds_raw, int c, long long e_loc)
{
    const int n_pad = a.g.n_pad, tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    long long *red = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);
    // energy of the final state, spins back to HBM
    {
        const long long w = OUT ? 0ll : wave_sum_i64(e_loc);
        if (lane == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red[3]), (unsigned long long)w);
        __syncthreads();
        int4 *dst = reinterpret_cast<int4 *>(a.spins + (size_t)c * n_pad);
        const int4 *src = reinterpret_cast<const int4 *>(lds_raw);
        for (int i = tid; i < n_pad / 16; i += nt) dst[i] = src[i];
        if (tid == 0 && !OUT) {                  // (OUT: the level loop has written the per-chain results)
            const long long E = a.efix[c] + red[3];
            a.efix[c] = E;
            if (a.energy_sink) a.energy_sink[c] = (double)E * __longlong_as_double((long long)(1023 - a.escale) << 52);
        }
    }
}

// F64: the fp64 mode (k_sweep_philox<double>) on the same windows, for instances whose couplings AND fields are exact multiples
// of 2^-qs.  The fp64 field of the spec is then the exact integer X times 2^-qs whatever the order of the sum, z = cb x takes
// one value per X, and the spec's test fma(u, 2^z, u) < 1 is a threshold on the 53-bit integer of u (accept_count_spec): the
// update is the fixed-point one with `k_u < K[X]` in place of `z < W(r)`.  Same bits as the sweep-by-sweep fp64 kernel.
// The swap step of one chain (k_rounds_fused after its grid-wide meeting, k_sweep_fused<.., DEFER> in its prologue): wave 0, `slot` =
// the chain's slot before the swap, Ed / e_all = its own / everybody's energy after the round's sweeps.  Returns the slot after it.
template <bool ATOMIC>
__device__ __forceinline__ int pt_swap_step_of_chain(int slot, uint32_t gc, int lane, int L, int n_pairs, int n_ladders, uint32_t round,
                                                     const int32_t *sel_round, const double *beta, int32_t *slot_of_chain, int32_t *chain_of_slot,
                                                     double Ed, const double *e_all, int32_t *log_pairs, uint8_t *log_acc, uint32_t seed_lo,
                                                     uint32_t seed_hi)
{
    (void)n_ladders;
    const int g = (int)gc / L;
    const int32_t *sel = sel_round + (size_t)g * n_pairs * 2;
    int fp = -1, fi = 0, out = slot;
    for (int p0 = 0; p0 < n_pairs; p0 += 64) {             // the pair this chain's slot belongs to, if any: one lane per selected pair
        const int p = p0 + lane;
        const int i = p < n_pairs ? sel[2 * p] : -5;
        const unsigned long long m = __ballot(i == slot || i + 1 == slot);
        if (m) { const int src = __ffsll((long long)m) - 1; fp = p0 + src; fi = __shfl(i, src, 64); break; }
    }
    if (fp >= 0) {                                         // (every lane of the wave computes it: the result is wave-uniform)
        const int i = fi, ps = slot == i ? i + 1 : i;
        int partner;
        double Ep;
        if (ATOMIC) {
            partner = __hip_atomic_load(&chain_of_slot[(size_t)g * L + ps], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            Ep = __hip_atomic_load(&e_all[partner], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            partner = chain_of_slot[(size_t)g * L + ps];
            Ep = e_all[partner];
        }
        const double Ea = slot == i ? Ed : Ep, Eb = slot == i ? Ep : Ed;
        const double dE = Eb - Ea, dB = beta[i + 1] - beta[i];
        const u32x4 rr = philox4x32_10((uint32_t)fp, round, (uint32_t)g, NLMC_TAG_SWAP, seed_lo, seed_hi);
        const double u = uniform_from(rr, 0.0);
        const double z = (dB * dE) * 1.4426950408889634;
        const bool acc = u < exp2_spec(z);
        if (lane == 0) {
            if (slot == i && log_pairs) {                  // the chain on the lower slot keeps the round's log entry
                const size_t at = (size_t)g * n_pairs + fp;
                log_pairs[2 * at] = i; log_pairs[2 * at + 1] = i + 1;
                log_acc[at] = acc ? 1 : 0;
            }
            if (acc) {
                if (ATOMIC) {
                    __hip_atomic_store(&slot_of_chain[gc], ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&chain_of_slot[(size_t)g * L + ps], (int)gc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    slot_of_chain[gc] = ps;
                    chain_of_slot[(size_t)g * L + ps] = (int)gc;
                }
            }
        }
        if (acc) out = ps;
    }
    return out;
}

template <bool DIAG, bool FLAGS, bool OUT, int FMT, bool F64 = false, bool DEFER = false>
__global__ __launch_bounds__(1024) void k_sweep_fused(SweepArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if ((unsigned)reinterpret_cast<size_t>(lds_raw) != 0u) __builtin_trap();   // spins at LDS offset 0: column == address
    static_assert(!(DEFER && OUT), "the deferred swap rides on the plain variant");
    const int c = a.chain_list ? a.chain_list[blockIdx.x] : (int)blockIdx.x;      // local chain id (state rows, RNG)
    const uint32_t gc_chain = (uint32_t)(a.chain_base + c);
    int row = a.slot_of_chain ? a.slot_of_chain[gc_chain] : c;
    int new_row = row;
    if (DEFER && a.defer.n_pairs > 0 && threadIdx.x < 64)          // wave 0, while the other waves load the spins
        new_row = pt_swap_step_of_chain<false>(row, gc_chain, (int)threadIdx.x, a.defer.ladder_len, a.defer.n_pairs, a.defer.n_ladders, a.defer.round,
                                               a.defer.plan_pairs, a.defer.beta, a.defer.slot_of_chain, a.defer.chain_of_slot,
                                               a.defer.e_prev[gc_chain], a.defer.e_prev, a.defer.log_pairs, a.defer.log_acc, a.seed_lo, a.seed_hi);
#ifdef NLMC_STAMPS
    const long long st_begin = (long long)__builtin_readcyclecounter();
#endif
    const FusedWin W{a.lvl_off, a.fsend, a.ell32, a.head32, a.warm_head, a.warm_ell, a.fz_npos_next, a.nlev[0], a.hi_max[0], a.sweep0};
    fused_state_load<FLAGS>(a, lds_raw, c);
    if (DEFER) {                                                   // the chain's slot after the swap, to every wave
        volatile int *sh = reinterpret_cast<volatile int *>(lds_raw + a.lds_red_off);
        __syncthreads();
        if (threadIdx.x == 0) sh[0] = new_row;
        __syncthreads();
        row = sh[0];
        __syncthreads();
        if (threadIdx.x == 0) sh[0] = 0;
    }
    const uint32_t gc = (a.rng_stride && a.slot_of_chain) ? (uint32_t)((c / a.rng_ladder_len) * a.rng_stride + a.rng_base + row) : gc_chain;
    long long e_loc = 0;
    fused_window<DIAG, FLAGS, OUT, FMT, F64>(a, W, lds_raw, row, gc, e_loc);
    fused_state_store<OUT>(a, lds_raw, c, e_loc);
#ifdef NLMC_STAMPS
    if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + 7] = (long long)__builtin_readcyclecounter() - st_begin;
#endif
}

// ---- persistent rounds: sweeps AND replica exchange of many rounds in one launch --------------------------------------------
// A context that owns whole ladders needs nothing from outside between two rounds: the chains stay in LDS, a round's swap
// decision needs only the energies of the two chains of a pair.  k_rounds_fused runs n_rounds consecutive planned windows (one
// window = the sweeps of one round) and, between them, the swap round of k_pt_swap -- same selection (planned), same Philox keys,
// same arithmetic, same label exchange -- with ONE grid-wide arrive-and-wait per round: every chain publishes its tracked energy
// (double-buffered by round parity), waits until all chains of the launch have, then the two chains of a selected pair each
// evaluate the identical decision and each updates its OWN entries of the slot maps (chain_of_slot[new slot] is written by the
// chain that moves there and read, before that, only by the same chain).  What a launch per round pays again and again -- kernel
// launch, spins HBM -> LDS -> HBM, the swap kernel's launch -- is paid once per chunk of rounds.  Launched cooperatively (all
// workgroups resident); the wait is bounded all the same (status 3 on a timeout, every workgroup leaves).  Bit-identical to
// k_sweep_fused + k_pt_swap round by round.
struct RoundsArgs {
    int n_rounds, n_windows_avail;   // rounds of this launch; planned windows from the first one on (>= n_rounds: the one behind the last is warmed)
    const int32_t *loff, *nlev, *himax, *send, *npos;      // plan arrays AT the first window; window r lies r strides further
    const int2 *head;
    const EdgeQ *ell;
    int ladder_len, n_pairs, n_ladders;
    uint32_t round0;
    const int32_t *plan_pairs;       // [n_rounds][n_ladders][n_pairs][2] at round0
    const double *beta;              // [ladder_len]
    int32_t *slot_of_chain, *chain_of_slot;
    double *ebuf;                    // [2][n_chains_global]
    int32_t *log_pairs;              // [n_rounds][n_ladders][n_pairs][2] at round0, or nullptr
    uint8_t *log_acc;
    unsigned *bar;                   // arrival counter, zero at launch
    int32_t *status;                 // sticky pt status: 3 = the grid wait timed out
    long long timeout_ticks;         // of the 100 MHz wall clock
};

// (no static LDS in these kernels: the spins sit at LDS offset 0; `flag` is a word of the reduction scratch)
// Everything the workgroups tell each other -- energies, slot maps, the counter -- is written and read by THREAD 0 with agent-scope
// atomics (coherent across the XCDs' L2s by themselves): no fence.  A __threadfence() here would write back and INVALIDATE the L2
// of every XCD once per wave and round -- and with it the window schedules every level streams from L2 (measured: 236 instead of
// 117 us per round).
__device__ __forceinline__ bool grid_arrive_and_wait(unsigned *bar, unsigned target, int32_t *status, long long t_end, volatile int *flag)
{
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's published stores have been acknowledged
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3 || (long long)wall_clock64() > t_end) {
                __hip_atomic_store(status, 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        *flag = good;
    }
    __syncthreads();
    return *flag != 0;
}

// The arguments come through POINTERS into constant memory, laundered once per round: as by-value kernel arguments the window
// body's loop-invariant scalar loads were hoisted out of the loop over the rounds and kept alive across it (176 spilled scalars
// instead of 31, a level loop ~15 % slower than the one-window kernel's); behind an opaque pointer they stay where they are used.
typedef const SweepArgs __attribute__((address_space(4))) *sweep_args_cptr;
typedef const RoundsArgs __attribute__((address_space(4))) *rounds_args_cptr;

template <bool DIAG, int FMT, bool F64>
__global__ __launch_bounds__(1024) void k_rounds_fused(const SweepArgs *ap_g, const RoundsArgs *qp_g)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if ((unsigned)reinterpret_cast<size_t>(lds_raw) != 0u) __builtin_trap();
    const sweep_args_cptr ap = (sweep_args_cptr)(uintptr_t)ap_g;
    const rounds_args_cptr qp = (rounds_args_cptr)(uintptr_t)qp_g;
    // State that lives from round to round sits in LDS (reduction scratch: [0] = {slot, barrier flag}, [1] = tracked energy, [3] =
    // energy sum of the round), NOT in registers: the level loop has no scalar register to spare (every value kept alive across it is
    // one more spill or reload inside it).
    {
        const SweepArgs &a0 = *(const SweepArgs *)ap;
        fused_state_load<false>(a0, lds_raw, (int)blockIdx.x);
        long long *red0 = reinterpret_cast<long long *>(lds_raw + a0.lds_red_off);
        if (threadIdx.x == 0) {
            reinterpret_cast<volatile int *>(red0)[0] = a0.slot_of_chain[a0.chain_base + (int)blockIdx.x];
            red0[1] = a0.efix[blockIdx.x];
        }
        __syncthreads();
    }
    const int n_rounds = qp->n_rounds;
    for (int r = 0; r < n_rounds; ++r) {
        sweep_args_cptr a_r = ap;
        rounds_args_cptr q_r = qp;
        asm volatile("" : "+s"(a_r), "+s"(q_r));                  // opaque per round: nothing of the body moves out of the loop
        const SweepArgs &a = *(const SweepArgs *)a_r;
        const RoundsArgs &q = *(const RoundsArgs *)q_r;
        long long *red = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);
        volatile int *sh = reinterpret_cast<volatile int *>(red);
        {
            const int c = (int)blockIdx.x, slot = sh[0];
            const size_t PS = (size_t)a.fz_pstride;
            const bool has_next = r + 1 < q.n_windows_avail && q.nlev[r + 1] > 0;
            const FusedWin W{q.loff + (size_t)r * (NLMC_LCAP + 1), q.send + (size_t)r * a.n_sweeps, q.ell + (size_t)r * PS * NLMC_FZ_W, q.head + (size_t)r * PS,
                             has_next ? q.head + (size_t)(r + 1) * PS : nullptr, has_next ? q.ell + (size_t)(r + 1) * PS * NLMC_FZ_W : nullptr,
                             has_next ? q.npos[r + 1] : 0, q.nlev[r], q.himax[r], a.sweep0 + (uint32_t)(r * a.n_sweeps)};
            const uint32_t gcr = a.rng_stride ? (uint32_t)((c / a.rng_ladder_len) * a.rng_stride + a.rng_base + slot) : (uint32_t)(a.chain_base + c);
            long long e_loc = 0;
            fused_window<DIAG, false, false, FMT, F64>(a, W, lds_raw, slot, gcr, e_loc);
            const long long w = wave_sum_i64(e_loc);
            if ((threadIdx.x & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red[3]), (unsigned long long)w);
            __syncthreads();
        }
        // ---- the swap round: thread 0 publishes, everybody meets, wave 0 decides
        const int tid = threadIdx.x, lane = tid & 63;
        const uint32_t gc = (uint32_t)(a.chain_base + (int)blockIdx.x);
        const int L = q.ladder_len, g = (int)gc / L;
        const size_t G = (size_t)q.n_ladders * L;
        double *eb = q.ebuf + (size_t)(r & 1) * G;
        double Ed = 0.0;
        if (tid < 64) {
            const long long E = red[1] + red[3];
            Ed = (double)E * __longlong_as_double((long long)(1023 - a.escale) << 52);
            if (tid == 0) __hip_atomic_store(&eb[gc], Ed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const long long t_end = (long long)wall_clock64() + q.timeout_ticks;       // (bounded per round)
#ifdef NLMC_DEBUG_KNOBS
        const bool ok = (a.dbg_flags & 4096) ? true : grid_arrive_and_wait(q.bar, (unsigned)(r + 1) * gridDim.x, q.status, t_end, sh + 1);   // 4096: timing experiment, nobody waits (wrong results)
#else
        const bool ok = grid_arrive_and_wait(q.bar, (unsigned)(r + 1) * gridDim.x, q.status, t_end, sh + 1);
#endif
        if (tid == 0) { red[1] += red[3]; red[3] = 0; }          // (wave 0 has read both; the sum is next added to a whole window from here)
        if (!ok) break;
#ifdef NLMC_DEBUG_KNOBS
        if (tid < 64 && !(a.dbg_flags & 8192)) {     // 8192: timing experiment, no swap step
#else
        if (tid < 64) {
#endif
            const int slot = sh[0];
            const int ns = pt_swap_step_of_chain<true>(slot, gc, lane, L, q.n_pairs, q.n_ladders, q.round0 + (uint32_t)r,
                                                       q.plan_pairs + (size_t)r * q.n_ladders * q.n_pairs * 2, q.beta, q.slot_of_chain, q.chain_of_slot, Ed, eb,
                                                       q.log_pairs ? q.log_pairs + (size_t)r * q.n_ladders * q.n_pairs * 2 : nullptr,
                                                       q.log_acc ? q.log_acc + (size_t)r * q.n_ladders * q.n_pairs : nullptr, a.seed_lo, a.seed_hi);
            if (lane == 0 && ns != slot) sh[0] = ns;
        }
        __syncthreads();
    }
    // state out: spins, tracked energy
    {
        const int n_pad = ap->g.n_pad, nt = blockDim.x, tid = threadIdx.x, c = (int)blockIdx.x;
        const long long *red = reinterpret_cast<const long long *>(lds_raw + ap->lds_red_off);
        __syncthreads();
        int4 *dst = reinterpret_cast<int4 *>(ap->spins + (size_t)c * n_pad);
        const int4 *src = reinterpret_cast<const int4 *>(lds_raw);
        for (int i = tid; i < n_pad / 16; i += nt) dst[i] = src[i];
        if (tid == 0) ap->efix[c] = red[1];
    }
}

// ------------------------------------------------------------------------------------------------------
// energy of a batch of configurations: E = -(m^T J m / 2 + m^T h), fp64, fixed reduction tree
// ------------------------------------------------------------------------------------------------------
struct EnergyArgs {
    CsrDev g;
    const int8_t *spins;   // [count][stride], or [count / inner][stride_outer] of [inner][stride] when inner > 0
    int64_t stride;
    int64_t stride_outer;
    int inner;
    double *out;           // [count] or nullptr
    long long *efix;       // [count] or nullptr
    int escale;
};

// IN_PLACE: the configuration is read where it lies in global memory (chains too long for LDS, csrc/nlmc_big.h) -- the same sums
// in the same order.
template <bool IN_PLACE = false>
__global__ void k_energy(EnergyArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double part[16];
    const int n = a.g.n;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int8_t *src = a.inner > 0 ? a.spins + (size_t)(blockIdx.x / a.inner) * a.stride_outer + (size_t)(blockIdx.x % a.inner) * a.stride
                                    : a.spins + (size_t)blockIdx.x * a.stride;
    const int8_t *s = src;
    if constexpr (!IN_PLACE) {
        int8_t *sl = reinterpret_cast<int8_t *>(lds_raw);
        for (int i = tid; i < n; i += nt) sl[i] = src[i];
        __syncthreads();
        s = sl;
    }
    // Four rows per thread at a time, the first 8 entries of each fetched unconditionally (padded arrays): 64
    // independent loads in flight instead of a rowptr -> entry -> entry chain per row.  The association of every sum
    // is the plain row-by-row, entry-by-entry one (a skipped slot leaves x untouched).
    double acc = 0.0;
    for (int k0 = tid; k0 < n; k0 += 4 * nt) {
        int rs[4], dg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * nt;
            rs[u] = k < n ? a.g.rowptr[k] : 0;
            dg[u] = k < n ? a.g.rowptr[k + 1] - rs[u] : 0;
        }
        int cj[4][8];
        double vj[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 8; ++q) { cj[u][q] = a.g.col[rs[u] + q]; vj[u][q] = a.g.val64[rs[u] + q]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * nt;
            if (k < n) {
                double x = 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q) x = q < dg[u] ? x + vj[u][q] * (double)s[cj[u][q]] : x;
                for (int e = rs[u] + 8; e < rs[u] + dg[u]; ++e) x += a.g.val64[e] * (double)s[a.g.col[e]];
                acc += (double)s[k] * (0.5 * x + a.g.h64[k]);
            }
        }
    }
    acc = wave_sum_f64_tree(acc);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < (nt + 63) / 64; ++w) tot += part[w];
        const double E = -tot;
        if (a.out) a.out[blockIdx.x] = E;
        if (a.efix) a.efix[blockIdx.x] = __double2ll_rn(E * __longlong_as_double((long long)(1023 + a.escale) << 52));
    }
}

__global__ void k_efix_to_double(const long long *efix, double *out, int count, int escale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)efix[i] * __longlong_as_double((long long)(1023 - escale) << 52);
}
