// nlmc_big.h -- chains that do not fit in LDS (gfx950): the same sweeps with the spins left in global memory.
//
// The reference takes any N (NMC/nmc.py:49-53).  The kernels of nlmc_kernels.h keep a chain's spins (and, per mode, its uniforms /
// thresholds and the levelizer's keys) in the 160 KB of LDS of one CU, which ends at N = 24 576 (fp64 uniforms: earlier).  Past
// that a chain is still one workgroup that walks the level schedule of its order, but reads and writes the spins in its row of
// `spins` in global memory (n bytes per chain: they live in the L2 of the workgroup's XCD) and reads the rows from the CSR
// arrays; the barrier between two levels orders the stores of one level before the loads of the next (all waves of a workgroup
// share the CU's vector L1, which writes through).  Same arithmetic as the LDS kernels, term by term: stream mode = the
// reference's (NMC/nmc.py:86-87), "f64" / "f32" = the Philox modes' spec (nlmc_device.h, oracle/nlo.c) -- the parity tests
// compare them with the same oracles, and at sizes both paths take with each other (NLMC_FORCE_BIG).
//
// Spin indices are 32 bits here (the LDS kernels pack k and the degree into one word); levels are not split at the workgroup's
// width (a thread loops over its share of a level).  Throughput is not the point of this path: a level costs a few dependent L2
// round trips instead of LDS ones.
#pragma once
#include "nlmc_kernels.h"

#define NLMC_BIG_HCAP 8192      // level histogram / cursors in LDS up to this many levels, in global scratch beyond

struct BigLevelizeArgs {
    CsrDev g;
    const uint32_t *keys_in;    // [n_orders][n] ranks (stream mode), or nullptr: Philox ORDER keys, written to `key`
    uint32_t seed_lo, seed_hi, sweep0;
    int per_chain, n_sweeps, chain_base;
    uint32_t *key;              // scratch [n_orders][n] (Philox keys)
    int32_t *lvl;               // scratch [n_orders][n]
    uint32_t *cur;              // scratch [n_orders][n + 2]: histogram / cursors of schedules deeper than NLMC_BIG_HCAP
    int2 *ord2;                 // [n_orders][n]  { k, row start }
    int32_t *lvl_off;           // [n_orders][n + 1]
    int32_t *nlev, *hi_max;     // [n_orders]
};

// The same levels as k_levelize (level of k = 1 + the largest level among the neighbours that precede k in the order; chaotic
// relaxation to the unique fixed point), one workgroup per order, the arrays in global memory.
__global__ __launch_bounds__(1024) void k_levelize_big(BigLevelizeArgs a)
{
    __shared__ uint32_t hist_lds[NLMC_BIG_HCAP + 2];
    __shared__ int sh_scan[16];
    __shared__ int sh_max;
    const int n = a.g.n, tid = threadIdx.x, nt = blockDim.x;
    const size_t o = blockIdx.x;
    int32_t *lvl = a.lvl + o * (size_t)n;
    const uint32_t *key;
    if (a.keys_in) {
        key = a.keys_in + o * (size_t)n;
        for (int k = tid; k < n; k += nt) lvl[k] = 0;
    } else {
        uint32_t *kw = a.key + o * (size_t)n;
        const uint32_t t = a.sweep0 + (uint32_t)(a.per_chain ? ((int)o % a.n_sweeps) : (int)o);
        const uint32_t grp = a.per_chain ? (uint32_t)(a.chain_base + (int)o / a.n_sweeps + 1) : 0u;
        for (int k = tid; k < n; k += nt) {
            kw[k] = philox4x32_10((uint32_t)k, t, grp, NLMC_TAG_ORDER, a.seed_lo, a.seed_hi).x;
            lvl[k] = 0;
        }
        key = kw;
    }
    if (tid == 0) sh_max = 0;
    __syncthreads();

    for (int it = 0; it <= n; ++it) {
        int changed = 0;
        for (int k = tid; k < n; k += nt) {
            const uint32_t kk = key[k];
            const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
            int m = 0;
            for (int e = rs; e < re; ++e) {
                const int j = a.g.col[e];
                if (j != k && precedes(key[j], j, kk, k)) m = max(m, lvl[j] + 1);
            }
            if (m != lvl[k]) { lvl[k] = m; changed = 1; }
        }
        if (!__syncthreads_or(changed)) break;
    }

    int lmax = 0;
    for (int k = tid; k < n; k += nt) lmax = max(lmax, lvl[k]);
    atomicMax(&sh_max, lmax);
    __syncthreads();
    const int nl = sh_max + 1;
    uint32_t *cnt = (nl + 1 <= NLMC_BIG_HCAP) ? hist_lds : a.cur + o * ((size_t)n + 2);
    for (int l = tid; l <= nl; l += nt) cnt[l] = 0u;
    __syncthreads();
    for (int k = tid; k < n; k += nt) atomicAdd(&cnt[lvl[k]], 1u);
    __syncthreads();
    // exclusive scan of cnt[0 .. nl): a chunk per thread, then the chunk sums over the workgroup
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
    int32_t *off = a.lvl_off + o * ((size_t)n + 1);
    for (int l = b; l < e; ++l) { const int c = (int)cnt[l]; off[l] = run; cnt[l] = (uint32_t)run; run += c; }
    if (tid == 0) { off[nl] = n; a.nlev[o] = nl; a.hi_max[o] = n; }
    __syncthreads();
    int2 *ord = a.ord2 + o * (size_t)n;
    for (int k = tid; k < n; k += nt) {
        const uint32_t pos = atomicAdd(&cnt[lvl[k]], 1u);      // (the order inside a level does not matter: its spins are independent)
        ord[pos] = make_int2(k, a.g.rowptr[k]);
    }
}

// rank[perm[i]] = i, u_spin[perm[i]] = u[i] (k_stream_scatter); `rank` arrives filled with 0xFFFFFFFF, a second hit of a spin
// or an index outside [0, n) sets *bad
__global__ void k_stream_scatter_big(int n, const int32_t *perm, const double *u, uint32_t *rank, double *us, int32_t *bad)
{
    const size_t o = blockIdx.x;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int k = perm[o * n + i];
        if (k < 0 || k >= n || atomicExch(&rank[o * n + k], (uint32_t)i) != 0xFFFFFFFFu) { atomicOr(bad, 1); continue; }
        us[o * n + k] = u[o * n + i];
    }
}

#define NLMC_BIG_STREAM 0
#define NLMC_BIG_F32 1
#define NLMC_BIG_F64 2

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
    int8_t *s = x.s;
    const uint8_t *fl = x.fl;
    const double esc = __longlong_as_double((long long)(1023 + a.escale) << 52);   // 2^escale
    const uint32_t gc_chain = (uint32_t)(a.chain_base + c);
    const int row = (MODE != NLMC_BIG_STREAM && a.slot_of_chain) ? a.slot_of_chain[gc_chain] : c;
    const uint32_t gc = (a.rng_stride && a.slot_of_chain) ? (uint32_t)((c / a.rng_ladder_len) * a.rng_stride + a.rng_base + row) : gc_chain;

    for (int t = 0; t < a.n_sweeps; ++t) {
        // (order ids as in the LDS kernels: the stream mode's by chain id, the Philox modes' by block)
        const int oid = a.per_chain ? ((MODE == NLMC_BIG_STREAM ? c : x.ob) * a.n_sweeps + t) : t;
        const int2 *__restrict__ ord = a.ord2 + (size_t)oid * n;
        const int32_t *__restrict__ off = a.lvl_off + (size_t)oid * ((size_t)n + 1);
        const int nl = a.nlev[oid];
        const uint32_t tt = a.sweep0 + (uint32_t)t;
        const double tb0 = a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss];
        const double tb1 = MODE == NLMC_BIG_STREAM ? 0.0 : a.tab[(size_t)row * a.tab_cs + (size_t)t * a.tab_ss + 1];
        const double *__restrict__ ut = MODE == NLMC_BIG_STREAM ? a.ustream + ((size_t)c * a.n_sweeps + t) * n : nullptr;
        for (int l = 0; l < nl; ++l) {
            const int lo = off[l], hi = off[l + 1];
            for (int i = lo + tid; i < hi; i += nt) {
                const int2 en = ord[i];
                const int k = en.x, rs = en.y, re = a.g.rowptr[k + 1];
                const unsigned f = fl ? (unsigned)fl[k] : 0u;
                const int so = (int)s[k];
                if constexpr (MODE == NLMC_BIG_STREAM) {
                    // NMC/nmc.py:86-87 on the unmodified / scaled / frozen row (k_sweep_stream)
                    double xs = 0.0, xd = 0.0;
                    for (int e = rs; e < re; ++e) {
                        const int j = a.g.col[e];
                        const double tm = a.g.val64[e] * (double)s[j];
                        xs += tm;
                        if (j == k) xd += tm;
                    }
                    const double hk = a.g.h64[k];
                    const double x_true = (xs - xd) + hk;
                    double xp;
                    if (f == 0u) xp = xs + hk;
                    else if (f == 1u) {
                        double y = 0.0;
                        for (int e = rs; e < re; ++e) y += (a.g.val64[e] / a.temp_x) * (double)s[a.g.col[e]];
                        xp = y + hk / a.temp_x;
                    } else xp = xs + ((f == 2u) ? 10000.0 : -10000.0);
                    const double v = tanh(tb0 * xp) - 2.0 * ut[k] + 1.0;
                    const int sn = (v > 0.0) - (v < 0.0);
                    if (sn != so) {
                        x.e_loc += __double2ll_rn(-(double)(sn - so) * x_true * esc);
                        s[k] = (int8_t)sn;
                    }
                } else if constexpr (MODE == NLMC_BIG_F64) {
                    if (f >= 2u) continue;                                  // frozen
                    const u32x4 rh = philox4x32_10((uint32_t)(k >> 2), tt, gc, NLMC_TAG_UNIFORM, a.seed_lo, a.seed_hi);
                    const u32x4 rl = philox4x32_10((uint32_t)(k >> 2), tt, gc, NLMC_TAG_UNIFORM_LO, a.seed_lo, a.seed_hi);
                    const int w = k & 3;
                    const uint32_t hw = w == 0 ? rh.x : w == 1 ? rh.y : w == 2 ? rh.z : rh.w;
                    const uint32_t lw = w == 0 ? rl.x : w == 1 ? rl.y : w == 2 ? rl.z : rl.w;
                    const double uk = uniform53_spec(hw, lw);
                    double xs = 0.0, xd = 0.0;
                    for (int e = rs; e < re; ++e) {
                        const int j = a.g.col[e];
                        const double v = a.g.val64[e], sv = (double)s[j];
                        xs = __fma_rn(v, sv, xs);
                        if (j == k) xd = __fma_rn(v, sv, xd);
                    }
                    const double hk = a.g.h64[k];
                    const double x_true = (xs - xd) + hk;                   // (xd == 0 without a self-coupling: the same bits as xs + hk)
                    const double z = (f == 1u ? tb1 : tb0) * (xs + hk);
                    const int sn = accept_up(uk, z) ? 1 : -1;
                    const int ds = sn - so;
                    if (ds != 0) x.e_loc += fixed_delta_slow(x_true, ds, esc);
                    s[k] = (int8_t)sn;
                } else {
                    if (f >= 2u) continue;
                    const u32x4 r = philox4x32_10((uint32_t)(k >> 2), tt, gc, NLMC_TAG_UNIFORM, a.seed_lo, a.seed_hi);
                    const int w = k & 3;
                    const float wk = threshold_spec(w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w);
                    int X = a.g.hq[k], Xd = 0;
                    for (int e = rs; e < re; ++e) {
                        const EdgeQ q = a.g.edge32[e];
                        const int tm = q.q * (int)s[q.col];
                        X += tm;
                        if (q.col == k) Xd += tm;
                    }
                    const float cq = (float)(f == 1u ? tb1 : tb0) * a.qinv;
                    const float z = cq * (float)X;
                    const int sn = (z < wk) ? 1 : -1;
                    const int cc = (so - sn) << a.eshift;
                    x.e_loc += (long long)(X - Xd) * (long long)cc;
                    s[k] = (int8_t)sn;
                }
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
