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

struct EdgeF { int32_t col; float val; };   // 8-byte packed CSR entry for the fp32 path (one dwordx2 load)

struct CsrDev {
    int n, n_pad;
    const int32_t *rowptr;
    const int32_t *col;      // [nnz]
    const double *val64;     // [nnz]
    const EdgeF *edge32;     // [nnz]
    const double *h64;       // [n]
    const float *h32;        // [n]
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
    int2 *ord2;           // [n_orders][n]  { k | deg << 16, row start }
    int32_t *lvl_off;     // [n_orders][n+1]
    int32_t *nlev;        // [n_orders]
    // optional packed per-order schedule in ELL form (slot-major, position-minor) so that the sweep kernel's
    // loads are fully coalesced: lane i of a level reads slot q at [(o*8+q)*n + i]
    EdgeF *ell32;         // [n_orders][8][n]  first 8 entries of row k(i), zero-padded (col 0, val 0)
    int2 *head32;         // [n_orders][n]     { k | deg << 16, bits of (float)h_k }
    int32_t *ellc64;      // [n_orders][8][n]
    double *ellv64;       // [n_orders][8][n]
    double *headh64;      // [n_orders][n]
};
#define NLMC_ELL_W 8

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
    for (int k = tid; k < n; k += nt) {
        const uint32_t pos = atomicAdd(&cnt[lvl[k]], 1u);
        const int rs = a.g.rowptr[k], deg = a.g.rowptr[k + 1] - rs;
        const int kd = k | (deg << 16);
        ord[pos] = make_int2(kd, rs);
        if (a.ell32) {
            a.head32[(size_t)o * n + pos] = make_int2(kd, __float_as_int(a.g.h32[k]));
#pragma unroll
            for (int q = 0; q < NLMC_ELL_W; ++q) {
                EdgeF ed{0, 0.0f};
                if (q < deg) ed = a.g.edge32[rs + q];
                a.ell32[((size_t)o * NLMC_ELL_W + q) * n + pos] = ed;
            }
        }
        if (a.ellc64) {
            a.headh64[(size_t)o * n + pos] = a.g.h64[k];
#pragma unroll
            for (int q = 0; q < NLMC_ELL_W; ++q) {
                int cj = 0; double vj = 0.0;
                if (q < deg) { cj = a.g.col[rs + q]; vj = a.g.val64[rs + q]; }
                a.ellc64[((size_t)o * NLMC_ELL_W + q) * n + pos] = cj;
                a.ellv64[((size_t)o * NLMC_ELL_W + q) * n + pos] = vj;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// sweeps
// ------------------------------------------------------------------------------------------------------
#define NLMC_LCAP 1024          // level offsets of one sweep kept in LDS by the pipelined path

struct SweepArgs {
    CsrDev g;
    int chain_base;
    int8_t *spins;            // [n_chains][n_pad]
    const uint8_t *flags;     // [n_chains][n_pad] or nullptr
    double temp_x;
    // schedule: ord2[o][i] = { k | deg << 16, row start }, lvl_off[o][0..nlev], nlev[o]
    const int2 *ord2;
    const int32_t *lvl_off, *nlev;
    const EdgeF *ell32;       // packed schedule (philox kernels), see LevelizeArgs
    const int2 *head32;
    const int32_t *ellc64;
    const double *ellv64, *headh64;
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
    int escale;
    long long *etrace;        // [n_chains][trace_sweeps] or nullptr
    int trace_sweeps, t0;     // sweeps of the whole call / index of this launch's first sweep inside the call
    int rec_stride;
    int8_t *strace;           // [n_chains][ceil(trace_sweeps/rec_stride)][n] or nullptr
    long long *emin;          // [n_chains] (in/out) or nullptr
    int32_t *argmin;          // [n_chains]
    int8_t *best;             // [n_chains][n_pad] or nullptr
    // LDS carve-up (bytes from the dynamic base)
    int lds_flags_off, lds_u_off, lds_loff_off, lds_red_off;
};

// ---- pieces shared by the two sweep kernels ------------------------------------------------------------
struct ChainCtx {
    int8_t *s;
    uint8_t *fl;
    long long *red;
    int tid, nt, c, n, n_pad;
    long long e_loc, E, Emin;
    int amin;
    bool per_sweep;
};

__device__ __forceinline__ void chain_load(const SweepArgs &a, unsigned char *lds_raw, ChainCtx &x)
{
    x.n = a.g.n; x.n_pad = a.g.n_pad;
    x.tid = threadIdx.x; x.nt = blockDim.x; x.c = blockIdx.x;
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
    x.E = a.efix[x.c];
    x.Emin = a.emin ? a.emin[x.c] : 0;
    x.amin = a.emin ? a.argmin[x.c] : 0;
    x.per_sweep = (a.etrace != nullptr) || (a.emin != nullptr);
    __syncthreads();
}

// end-of-sweep bookkeeping: energy trace, running minimum + argmin state, recorded configurations
__device__ __forceinline__ void sweep_epilogue(const SweepArgs &a, ChainCtx &x, int t)
{
    const int tg = a.t0 + t;                                 // sweep index inside the call
    const bool rec = a.strace && (tg % a.rec_stride == 0);   // M[:, ::M_skip]  (NMC/nmc.py:390)
    if (x.per_sweep) {
        const long long w = wave_sum_i64(x.e_loc);
        x.e_loc = 0;
        if ((x.tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&x.red[0]), (unsigned long long)w);
        __syncthreads();
        if (x.tid == 0) {
            x.E += x.red[0];
            x.red[0] = 0;
            if (a.etrace) a.etrace[(size_t)x.c * a.trace_sweeps + tg] = x.E;
            int better = 0;
            if (a.emin && x.E < x.Emin) { x.Emin = x.E; x.amin = tg; better = 1; }   // strict <: first argmin (np.argmin)
            x.red[1] = better;
        }
        __syncthreads();
        if (a.best && x.red[1]) {
            int4 *dst = reinterpret_cast<int4 *>(a.best + (size_t)x.c * x.n_pad);
            const int4 *src = reinterpret_cast<const int4 *>(x.s);
            for (int i = x.tid; i < x.n_pad / 16; i += x.nt) dst[i] = src[i];
        }
    }
    if (rec) {
        const int n_rec = (a.trace_sweeps + a.rec_stride - 1) / a.rec_stride;
        int8_t *dst = a.strace + ((size_t)x.c * n_rec + (size_t)(tg / a.rec_stride)) * x.n;
        for (int i = x.tid; i < x.n; i += x.nt) dst[i] = x.s[i];
    }
    if (x.per_sweep || rec) __syncthreads();   // LDS spins / red[1] are rewritten next sweep
}

__device__ __forceinline__ void chain_store(const SweepArgs &a, ChainCtx &x)
{
    if (!x.per_sweep) {
        const long long w = wave_sum_i64(x.e_loc);
        if ((x.tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&x.red[0]), (unsigned long long)w);
        __syncthreads();
        if (x.tid == 0) x.E += x.red[0];
    }
    int4 *dst = reinterpret_cast<int4 *>(a.spins + (size_t)x.c * x.n_pad);
    const int4 *src = reinterpret_cast<const int4 *>(x.s);
    for (int i = x.tid; i < x.n_pad / 16; i += x.nt) dst[i] = src[i];
    if (x.tid == 0) {
        a.efix[x.c] = x.E;
        if (a.emin) { a.emin[x.c] = x.Emin; a.argmin[x.c] = x.amin; }
    }
}

// ---- STREAM mode: the reference's arithmetic, fp64, externally drawn uniforms --------------------------
// m_k = sign(tanh(beta x_k) - 2u + 1),  x = (sum_e J_e m_e) + h_k in CSR order  (NMC/nmc.py:86-87)
__global__ void k_sweep_stream(SweepArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ChainCtx x;
    chain_load(a, lds_raw, x);
    const int n = x.n, tid = x.tid, nt = x.nt, c = x.c;
    int8_t *s = x.s;
    const uint8_t *fl = x.fl;
    const double esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale

    for (int t = 0; t < a.n_sweeps; ++t) {
        const int oid = a.per_chain ? (c * a.n_sweeps + t) : t;
        const int2 *__restrict__ ord = a.ord2 + (size_t)oid * n;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * (n + 1);
        const int nl = a.nlev[oid];
        const double beta = a.tab[(size_t)c * a.tab_cs + (size_t)t * a.tab_ss];
        const double *__restrict__ ut = a.ustream + ((size_t)c * a.n_sweeps + t) * n;
        for (int l = 0; l < nl; ++l) {
            const int lo = off[l], hi = off[l + 1];
            for (int i = lo + tid; i < hi; i += nt) {
                const int2 en = ord[i];
                const int k = en.x & 0xFFFF, rs = en.y, re = en.y + (int)((unsigned)en.x >> 16);
                const unsigned f = fl ? (unsigned)fl[k] : 0u;
                const int so = (int)s[k];
                double xs = 0.0, xd = 0.0;   // xd: diagonal term, excluded from the energy delta
                for (int e = rs; e < re; ++e) {
                    const int j = a.g.col[e];
                    const double tm = a.g.val64[e] * (double)s[j];
                    xs += tm;
                    if (j == k) xd += tm;
                }
                const double hk = a.g.h64[k];
                const double x_true = (xs - xd) + hk;   // field of the UNMODIFIED (J,h)
                double xp;
                if (f == 0u) xp = xs + hk;
                else if (f == 1u) {   // cluster rows divided element-wise by temp_x (NMC/nmc.py:379-380)
                    double y = 0.0;
                    for (int e = rs; e < re; ++e) y += (a.g.val64[e] / a.temp_x) * (double)s[a.g.col[e]];
                    xp = y + hk / a.temp_x;
                } else xp = xs + ((f == 2u) ? 10000.0 : -10000.0);   // NMC/nmc.py:381,401
                const double v = tanh(beta * xp) - 2.0 * ut[k] + 1.0;
                const int sn = (v > 0.0) - (v < 0.0);                // np.sign
                if (sn != so) {
                    x.e_loc += __double2ll_rn(-(double)(sn - so) * x_true * esc);
                    s[k] = (int8_t)sn;
                }
            }
            __syncthreads();
        }
        sweep_epilogue(a, x, t);
    }
    chain_store(a, x);
}

// ---- PHILOX mode (throughput) ---------------------------------------------------------------------------
template <typename T> struct Pf;            // one schedule item: k, degree, h_k and the first 8 entries of row k
template <> struct Pf<float> {
    EdgeF ed[NLMC_ELL_W];
    int kd;
    float h;
    // coalesced: consecutive lanes read consecutive positions of every slot plane
    __device__ __forceinline__ void load(const SweepArgs &a, size_t oid, int n, int i)
    {
        const int2 hd = a.head32[oid * n + i];
        const EdgeF *__restrict__ p = a.ell32 + oid * NLMC_ELL_W * n + i;
#pragma unroll
        for (int q = 0; q < NLMC_ELL_W; ++q) ed[q] = p[(size_t)q * n];
        kd = hd.x;
        h = __int_as_float(hd.y);
    }
    // Declare every register of the item dead (no instruction): lets a lane-masked load land directly in these
    // registers without a merge copy that would wait on the load right after issuing it.
    __device__ __forceinline__ void kill()
    {
#pragma unroll
        for (int q = 0; q < NLMC_ELL_W; ++q) { asm volatile("" : "=v"(ed[q].col)); asm volatile("" : "=v"(ed[q].val)); }
        asm volatile("" : "=v"(kd));
        asm volatile("" : "=v"(h));
    }
    __device__ __forceinline__ int col(int q) const { return ed[q].col; }
    __device__ __forceinline__ float val(int q) const { return ed[q].val; }
    static __device__ __forceinline__ void tail(const CsrDev &g, int e, int &cj, float &vj) { const EdgeF t = g.edge32[e]; cj = t.col; vj = t.val; }
};
template <> struct Pf<double> {
    int cj[NLMC_ELL_W];
    double vj[NLMC_ELL_W];
    int kd;
    double h;
    __device__ __forceinline__ void load(const SweepArgs &a, size_t oid, int n, int i)
    {
        const int32_t *__restrict__ pc = a.ellc64 + oid * NLMC_ELL_W * n + i;
        const double *__restrict__ pv = a.ellv64 + oid * NLMC_ELL_W * n + i;
#pragma unroll
        for (int q = 0; q < NLMC_ELL_W; ++q) { cj[q] = pc[(size_t)q * n]; vj[q] = pv[(size_t)q * n]; }
        kd = a.ord2[oid * n + i].x;
        h = a.headh64[oid * n + i];
    }
    __device__ __forceinline__ void kill()
    {
#pragma unroll
        for (int q = 0; q < NLMC_ELL_W; ++q) { asm volatile("" : "=v"(cj[q])); asm volatile("" : "=v"(vj[q])); }
        asm volatile("" : "=v"(kd));
        asm volatile("" : "=v"(h));
    }
    __device__ __forceinline__ int col(int q) const { return cj[q]; }
    __device__ __forceinline__ double val(int q) const { return vj[q]; }
    static __device__ __forceinline__ void tail(const CsrDev &g, int e, int &c, double &v) { c = g.col[e]; v = g.val64[e]; }
};

__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __fmaf_rn(a, b, c); }
__device__ __forceinline__ double fma_rn(double a, double b, double c) { return __fma_rn(a, b, c); }

// Heat-bath update of one spin from a prefetched schedule item.  J*s is exact (s = +-1), so fma(J, s, x) == x + J*s;
// the zero-padded slots add +-0 and leave x unchanged, which keeps the oracle's row-order sum bit for bit.
template <typename T, bool DIAG>
__device__ __forceinline__ void update_spin(const SweepArgs &a, ChainCtx &x, const T *ur, const Pf<T> &pf, size_t oid, int i,
                                            T cb0, T cb1, double esc)
{
    const int k = pf.kd & 0xFFFF, deg = (int)((unsigned)pf.kd >> 16);
    const unsigned f = x.fl ? (unsigned)x.fl[k] : 0u;
    if (f >= 2u) return;                       // frozen
    int8_t *s = x.s;
    T sj[NLMC_ELL_W];
#pragma unroll
    for (int q = 0; q < NLMC_ELL_W; ++q) sj[q] = (T)s[pf.col(q)];     // all LDS reads in flight together
    T xs = T(0), xd = T(0);
#pragma unroll
    for (int q = 0; q < NLMC_ELL_W; ++q) {
        xs = fma_rn(pf.val(q), sj[q], xs);
        if (DIAG) { const T nd = fma_rn(pf.val(q), sj[q], xd); xd = (q < deg && pf.col(q) == k) ? nd : xd; }
    }
    if (deg > NLMC_ELL_W) {                    // rows longer than the packed window: rest from the CSR arrays
        const int rs = a.ord2[oid * x.n + i].y;
        for (int e = NLMC_ELL_W; e < deg; ++e) {
            int j; T v;
            Pf<T>::tail(a.g, rs + e, j, v);
            const T sv = (T)s[j];
            xs = fma_rn(v, sv, xs);
            if (DIAG && j == k) xd = fma_rn(v, sv, xd);
        }
    }
    const T x_true = DIAG ? ((xs - xd) + pf.h) : (xs + pf.h);
    const T xf = xs + pf.h;
    const T z = (f == 1u ? cb1 : cb0) * xf;
    const int so = (int)s[k];
    const int sn = accept_up(ur[k], z) ? 1 : -1;
    if (sn != so) {
        x.e_loc += __double2ll_rn(-(double)(sn - so) * (double)x_true * esc);
        s[k] = (int8_t)sn;
    }
}

// uniforms of one sweep for every spin of this chain -> LDS.  One Philox4x32-10 call serves 4 (f32) / 2 (f64) spins:
//   f32: u(k) = 24 high bits of word (k & 3) of philox(k >> 2, t, chain, UNIFORM)
//   f64: u(k) = 53 bits from words (2(k&1), 2(k&1)+1) of philox(k >> 1, t, chain, UNIFORM)
__device__ __forceinline__ void fill_uniforms(float *ur, int n, uint32_t tt, uint32_t gc, uint32_t k0, uint32_t k1, int tid, int nt)
{
    for (int b = tid; b < (n + 3) / 4; b += nt) {
        const u32x4 r = philox4x32_10((uint32_t)b, tt, gc, NLMC_TAG_UNIFORM, k0, k1);
        float4 v;
        v.x = (float)(r.x >> 8) * 5.9604644775390625e-08f;
        v.y = (float)(r.y >> 8) * 5.9604644775390625e-08f;
        v.z = (float)(r.z >> 8) * 5.9604644775390625e-08f;
        v.w = (float)(r.w >> 8) * 5.9604644775390625e-08f;
        reinterpret_cast<float4 *>(ur)[b] = v;     // ur has (n+3)/4*4 entries
    }
}
__device__ __forceinline__ void fill_uniforms(double *ur, int n, uint32_t tt, uint32_t gc, uint32_t k0, uint32_t k1, int tid, int nt)
{
    for (int b = tid; b < (n + 1) / 2; b += nt) {
        const u32x4 r = philox4x32_10((uint32_t)b, tt, gc, NLMC_TAG_UNIFORM, k0, k1);
        double2 v;
        v.x = ((double)(r.x >> 5) * 67108864.0 + (double)(r.y >> 6)) / 9007199254740992.0;
        v.y = ((double)(r.z >> 5) * 67108864.0 + (double)(r.w >> 6)) / 9007199254740992.0;
        reinterpret_cast<double2 *>(ur)[b] = v;    // ur has (n+1)/2*2 entries
    }
}

template <typename T, bool DIAG>
__global__ void k_sweep_philox(SweepArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ChainCtx x;
    chain_load(a, lds_raw, x);
    const int n = x.n, tid = x.tid, nt = x.nt, c = x.c;
    T *ur = reinterpret_cast<T *>(lds_raw + a.lds_u_off);
    int *loff = reinterpret_cast<int *>(lds_raw + a.lds_loff_off);
    const uint32_t gc = (uint32_t)(a.chain_base + c);
    const int row = a.slot_of_chain ? a.slot_of_chain[gc] : c;
    const double esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale

    for (int t = 0; t < a.n_sweeps; ++t) {
        const int oid = a.per_chain ? (c * a.n_sweeps + t) : t;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * (n + 1);
        const int nl = a.nlev[oid];
        const T cb0 = (T)a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss];
        const T cb1 = (T)a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss + 1];
        const uint32_t tt = a.sweep0 + (uint32_t)t;

        // sweep prologue: all lanes busy -- uniforms for every spin, level offsets into LDS
        fill_uniforms(ur, n, tt, gc, a.seed_lo, a.seed_hi, tid, nt);
        const bool fast = nl < NLMC_LCAP;
        if (fast) for (int l = tid; l <= nl; l += nt) loff[l] = off[l];
        __syncthreads();

        const size_t so = (size_t)oid;
        if (fast) {
            // software pipeline over levels: while level l is computed, the schedule items of level l+1 are in
            // flight.  Their addresses depend only on the level offsets (LDS), never on spin values, and the
            // schedule was built with level_cap == blockDim.x: at most one spin per thread and level.
            Pf<T> pfa, pfb;                // ping-pong (manual 2x unroll: no register rotation)
            bool va, vb;
            int ia, ib;
            auto fetch = [&](int l, Pf<T> &p, bool &valid, int &ic) {
                const int lc = min(l, nl - 1);
                const int i = loff[lc] + tid;
                valid = (l < nl) && (i < loff[lc + 1]);
                ic = valid ? i : 0;        // idle lanes of a busy wave all read item 0 (one extra cache line)
                // a vector-memory instruction costs the CU's address unit ~16 cycles per wave whatever its lanes
                // do, so waves without any item in this level skip the loads altogether (wave-uniform branch)
                if (__ballot(valid) != 0ull) p.load(a, so, n, ic);
            };
            fetch(0, pfa, va, ia);
            for (int l = 0; l < nl; l += 2) {
                fetch(l + 1, pfb, vb, ib);
                if (va) update_spin<T, DIAG>(a, x, ur, pfa, so, ia, cb0, cb1, esc);
                __syncthreads();
                if (l + 1 < nl) {
                    fetch(l + 2, pfa, va, ia);
                    if (vb) update_spin<T, DIAG>(a, x, ur, pfb, so, ib, cb0, cb1, esc);
                    __syncthreads();
                }
            }
        } else {
            // very deep schedules (dense graphs): plain level loop
            for (int l = 0; l < nl; ++l) {
                const int lo = off[l], hi = off[l + 1];
                for (int i = lo + tid; i < hi; i += nt) {
                    Pf<T> pe;
                    pe.load(a, so, n, i);
                    update_spin<T, DIAG>(a, x, ur, pe, so, i, cb0, cb1, esc);
                }
                __syncthreads();
            }
        }
        sweep_epilogue(a, x, t);
    }
    chain_store(a, x);
}

// ------------------------------------------------------------------------------------------------------
// energy of a batch of configurations: E = -(m^T J m / 2 + m^T h), fp64, fixed reduction tree
// ------------------------------------------------------------------------------------------------------
struct EnergyArgs {
    CsrDev g;
    const int8_t *spins;   // [count][stride]
    int64_t stride;
    double *out;           // [count] or nullptr
    long long *efix;       // [count] or nullptr
    int escale;
};

__global__ void k_energy(EnergyArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double part[16];
    const int n = a.g.n;
    int8_t *s = reinterpret_cast<int8_t *>(lds_raw);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int8_t *src = a.spins + (size_t)blockIdx.x * a.stride;
    for (int i = tid; i < n; i += nt) s[i] = src[i];
    __syncthreads();
    double acc = 0.0;
    for (int k = tid; k < n; k += nt) {
        double x = 0.0;
        for (int e = a.g.rowptr[k]; e < a.g.rowptr[k + 1]; ++e) x += a.g.val64[e] * (double)s[a.g.col[e]];
        acc += (double)s[k] * (0.5 * x + a.g.h64[k]);
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
