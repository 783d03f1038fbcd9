// nlmc_kernels.h -- HIP kernels of the sweep path for gfx950.
//
//   k_levelize : one workgroup per sweep-order.  Turns a visiting order (Philox keys, or ranks of a host-drawn
//                permutation) into a LEVEL SCHEDULE: level(k) = 1 + max level(j) over neighbours j visited before k.
//                Spins of one level are mutually non-adjacent, so updating levels in ascending order, each level
//                in parallel, is bit-identical to the reference's sequential pass (NMC/nmc.py:71-87).
//   k_sweep    : one workgroup per chain; the chain's spins (and phase flags) live in LDS for the whole launch;
//                loops sweeps x levels with one s_barrier per level; CSR rows come from L2; energy is tracked
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
    int32_t *order;       // [n_orders][n]
    int32_t *lvl_off;     // [n_orders][n+1]
    int32_t *nlev;        // [n_orders]
};

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
    for (int l = b; l < e; ++l) { const int c = (int)cnt[l]; off[l] = run; cnt[l] = (uint32_t)run; run += c; }
    if (tid == 0) { off[nl] = n; a.nlev[o] = nl; }
    __syncthreads();

    // placement (intra-level order is irrelevant: same-level spins are independent)
    int32_t *ord = a.order + (size_t)o * n;
    for (int k = tid; k < n; k += nt) {
        const uint32_t pos = atomicAdd(&cnt[lvl[k]], 1u);
        ord[pos] = k;
    }
}

// ------------------------------------------------------------------------------------------------------
// sweeps
// ------------------------------------------------------------------------------------------------------
struct SweepArgs {
    CsrDev g;
    int chain_base;
    int8_t *spins;            // [n_chains][n_pad]
    const uint8_t *flags;     // [n_chains][n_pad] or nullptr
    double temp_x;
    // schedule
    const int32_t *order, *lvl_off, *nlev;
    int per_chain;            // order id = c * sched_sweeps + (t - 0) else t
    int n_sweeps;             // sweeps in this launch (== orders per chain in the schedule window)
    uint32_t sweep0;          // global index of sweep 0 of this launch
    uint32_t seed_lo, seed_hi;
    // temperature table: element (row, t, j) at tab[row*tab_cs + t*tab_ss + j], row = slot or local chain
    const double *tab;
    int tab_cs, tab_ss;
    const int32_t *slot_of_chain;   // [n_chains_global] or nullptr
    int ladder_len;
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
    int lds_red_off;
};

template <typename T> struct RowAcc;

template <> struct RowAcc<float> {
    // x = ((0 + v0 s0) + v1 s1) + ...  in fp32; xd = diagonal term (excluded from the energy delta)
    static __device__ __forceinline__ void field(const CsrDev &g, const int8_t *s, int k, int rs, int re, float &x,
                                                 float &xd)
    {
        x = 0.0f; xd = 0.0f;
        for (int e = rs; e < re; e += 8) {
            EdgeF ed[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ed[q] = g.edge32[min(e + q, re - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float t = ed[q].val * (float)s[ed[q].col];
                if (e + q < re) { x += t; if (ed[q].col == k) xd += t; }
            }
        }
    }
    static __device__ __forceinline__ float h(const CsrDev &g, int k) { return g.h32[k]; }
};

template <> struct RowAcc<double> {
    static __device__ __forceinline__ void field(const CsrDev &g, const int8_t *s, int k, int rs, int re, double &x,
                                                 double &xd)
    {
        x = 0.0; xd = 0.0;
        for (int e = rs; e < re; e += 4) {
            int cj[4]; double vj[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int ee = min(e + q, re - 1); cj[q] = g.col[ee]; vj[q] = g.val64[ee]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double t = vj[q] * (double)s[cj[q]];
                if (e + q < re) { x += t; if (cj[q] == k) xd += t; }
            }
        }
    }
    static __device__ __forceinline__ double h(const CsrDev &g, int k) { return g.h64[k]; }
};

template <typename T, bool STREAM>
__global__ void k_sweep(SweepArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = a.g.n, n_pad = a.g.n_pad;
    int8_t *s = reinterpret_cast<int8_t *>(lds_raw);
    uint8_t *fl = a.flags ? (lds_raw + n_pad) : nullptr;
    long long *red = reinterpret_cast<long long *>(lds_raw + a.lds_red_off);   // [0] sweep sum, [1] broadcast flag
    const int tid = threadIdx.x, nt = blockDim.x;
    const int c = blockIdx.x;
    const uint32_t gc = (uint32_t)(a.chain_base + c);

    {   // chain state -> LDS (16-byte vectors; rows are padded to 16)
        const int4 *src = reinterpret_cast<const int4 *>(a.spins + (size_t)c * n_pad);
        int4 *dst = reinterpret_cast<int4 *>(s);
        for (int i = tid; i < n_pad / 16; i += nt) dst[i] = src[i];
        if (fl) {
            const int4 *fsrc = reinterpret_cast<const int4 *>(a.flags + (size_t)c * n_pad);
            int4 *fdst = reinterpret_cast<int4 *>(fl);
            for (int i = tid; i < n_pad / 16; i += nt) fdst[i] = fsrc[i];
        }
    }
    if (tid == 0) { red[0] = 0; red[1] = 0; }
    __syncthreads();

    const int row = a.slot_of_chain ? a.slot_of_chain[gc] : c;
    const double esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale
    long long e_loc = 0;                       // this thread's share of the running energy delta
    long long E = a.efix[c];                   // meaningful in thread 0
    long long Emin = a.emin ? a.emin[c] : 0;
    int amin = a.emin ? a.argmin[c] : 0;
    const bool per_sweep = (a.etrace != nullptr) || (a.emin != nullptr);

    for (int t = 0; t < a.n_sweeps; ++t) {
        const int oid = a.per_chain ? (c * a.n_sweeps + t) : t;
        const int32_t *__restrict__ ord = a.order + (size_t)oid * n;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * (n + 1);
        const int nl = a.nlev[oid];
        const double tb0 = a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss];
        const double tb1 = STREAM ? tb0 : a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss + 1];
        const uint32_t tt = a.sweep0 + (uint32_t)t;

        for (int l = 0; l < nl; ++l) {
            const int lo = off[l], hi = off[l + 1];
            for (int i = lo + tid; i < hi; i += nt) {
                const int k = ord[i];
                const unsigned f = fl ? (unsigned)fl[k] : 0u;
                const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
                const int so = (int)s[k];
                int sn;
                double x_true;   // field of the UNMODIFIED (J,h) without the diagonal term, for the energy delta
                if constexpr (STREAM) {
                    // reference arithmetic: x = (sum_e J_e m_e) + h_k on the phase matrices, fp64, CSR order
                    double x, xd;
                    RowAcc<double>::field(a.g, s, k, rs, re, x, xd);
                    x_true = (x - xd) + a.g.h64[k];
                    double xp;
                    if (f == 0u) xp = x + a.g.h64[k];
                    else if (f == 1u) {   // rows of cluster spins divided element-wise by temp_x (NMC/nmc.py:379-380)
                        double y = 0.0;
                        for (int e = rs; e < re; ++e) y += (a.g.val64[e] / a.temp_x) * (double)s[a.g.col[e]];
                        xp = y + a.g.h64[k] / a.temp_x;
                    } else xp = x + ((f == 2u) ? 10000.0 : -10000.0);   // NMC/nmc.py:381,401
                    const double u = a.ustream[((size_t)c * a.n_sweeps + t) * n + k];
                    const double v = tanh(tb0 * xp) - 2.0 * u + 1.0;
                    sn = (v > 0.0) - (v < 0.0);                          // np.sign
                } else {
                    if (f >= 2u) continue;                               // frozen
                    T x, xd;
                    RowAcc<T>::field(a.g, s, k, rs, re, x, xd);
                    const T hk = RowAcc<T>::h(a.g, k);
                    x_true = (double)((x - xd) + hk);
                    x = x + hk;
                    const u32x4 r = philox4x32_10((uint32_t)k, tt, gc, NLMC_TAG_UNIFORM, a.seed_lo, a.seed_hi);
                    const T u = uniform_from(r, T(0));
                    const T z = (T)(f == 1u ? tb1 : tb0) * x;
                    sn = accept_up(u, z) ? 1 : -1;
                }
                if (sn != so) {
                    e_loc += __double2ll_rn(-(double)(sn - so) * x_true * esc);
                    s[k] = (int8_t)sn;
                }
            }
            __syncthreads();
        }

        const int tg = a.t0 + t;                                 // sweep index inside the call
        const bool rec = a.strace && (tg % a.rec_stride == 0);   // M[:, ::M_skip]  (NMC/nmc.py:390)
        if (per_sweep) {
            const long long w = wave_sum_i64(e_loc);
            e_loc = 0;
            if ((tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red[0]), (unsigned long long)w);
            __syncthreads();
            if (tid == 0) {
                E += red[0];
                red[0] = 0;
                if (a.etrace) a.etrace[(size_t)c * a.trace_sweeps + tg] = E;
                int better = 0;
                if (a.emin && E < Emin) { Emin = E; amin = tg; better = 1; }   // strict <: first argmin (np.argmin)
                red[1] = better;
            }
            __syncthreads();
            if (a.best && red[1]) {
                int4 *dst = reinterpret_cast<int4 *>(a.best + (size_t)c * n_pad);
                const int4 *src = reinterpret_cast<const int4 *>(s);
                for (int i = tid; i < n_pad / 16; i += nt) dst[i] = src[i];
            }
        }
        if (rec) {
            const int n_rec = (a.trace_sweeps + a.rec_stride - 1) / a.rec_stride;
            int8_t *dst = a.strace + ((size_t)c * n_rec + (size_t)(tg / a.rec_stride)) * n;
            for (int i = tid; i < n; i += nt) dst[i] = s[i];
        }
        if (per_sweep || rec) __syncthreads();   // LDS spins / red[1] are rewritten next sweep
    }

    if (!per_sweep) {
        const long long w = wave_sum_i64(e_loc);
        if ((tid & 63) == 0 && w != 0) atomicAdd(reinterpret_cast<unsigned long long *>(&red[0]), (unsigned long long)w);
        __syncthreads();
        if (tid == 0) E += red[0];
    }
    {
        int4 *dst = reinterpret_cast<int4 *>(a.spins + (size_t)c * n_pad);
        const int4 *src = reinterpret_cast<const int4 *>(s);
        for (int i = tid; i < n_pad / 16; i += nt) dst[i] = src[i];
    }
    if (tid == 0) {
        a.efix[c] = E;
        if (a.emin) { a.emin[c] = Emin; a.argmin[c] = amin; }
    }
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
