// nlmc_nmc.h -- device-side bookkeeping of replica-exchange rounds whose marked temperature slots run NMC cycles
// (NPT/npt.py:622-647 submits NMC_task, :479-512, for the doNMC slots): chain subsets, backbone seeds and cluster masks,
// phase flags, argmin hand-off.  Nothing here touches the host: a whole round is a chain of stream-ordered launches.
#pragma once
#include "nlmc_kernels.h"

// Stable partition of the local chains by the mark of the temperature slot they currently sit on:
//   list[0 .. n_unmarked)            local chain ids on unmarked slots, ascending
//   list[n_unmarked .. n_chains)     local chain ids on marked slots, ascending
// One workgroup (n_chains is a few thousand at most); ladders are whole inside a context, so both counts are static.
__global__ __launch_bounds__(1024) void k_subset_build(int n_chains, int chain_base, const int32_t *slot_of_chain, const uint8_t *slot_mark,
                                                       int n_unmarked, int32_t *list)
{
    __shared__ int wave_tot[16];
    __shared__ int base_m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_m = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_chains; c0 += 1024) {
        const int c = c0 + tid;
        const bool in = c < n_chains;
        const bool marked = in && slot_mark[slot_of_chain[chain_base + c]] != 0;
        const unsigned long long bm = __ballot(marked);
        const int before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bm);
        __syncthreads();
        int wbase = base_m;
        for (int w = 0; w < wv; ++w) wbase += wave_tot[w];
        const int m_before = wbase + before;                     // marked chains with a smaller id
        if (in) {
            if (marked) list[n_unmarked + m_before] = c;
            else list[c - m_before] = c;
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_tot[w]; base_m += t; }
        __syncthreads();
    }
}

// The argmin-energy configuration of the previous phase becomes the start state of the next one (NMC/nmc.py:394-395,
// 415-416, 430-431 == NPT/npt.py:436-437,454-455,469-470): spins <- best, tracked energy <- the minimum.
__global__ void k_adopt_best(int n_pad, const int32_t *list, int8_t *spins, const int8_t *best, long long *efix, const long long *emin)
{
    const int c = list ? list[blockIdx.x] : (int)blockIdx.x;
    const int4 *src = reinterpret_cast<const int4 *>(best + (size_t)c * n_pad);
    int4 *dst = reinterpret_cast<int4 *>(spins + (size_t)c * n_pad);
    for (int i = threadIdx.x; i < n_pad / 16; i += blockDim.x) dst[i] = src[i];
    if (threadIdx.x == 0) efix[c] = emin[c];
}

__global__ void k_fill_min(int count, const int32_t *list, long long *emin, int32_t *argmin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int c = list ? list[i] : i;
    emin[c] = 0x7FFFFFFFFFFFFFFFll;
    argmin[c] = 0;
}

// m_star of LBP_convexified for problem p = the current configuration of chain list[p] (NPT/npt.py:630-633 hands
// m_start[replica] to NMC_task)
__global__ void k_lbp_seeds(int n, int n_pad, const int32_t *list, const int8_t *spins, double *ms)
{
    const int p = blockIdx.x, c = list ? list[p] : p;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ms[(size_t)p * n + i] = (double)spins[(size_t)c * n_pad + i];
}

// Union of the clusters find_clusters (NMC/nmc.py:257-318) grows from the marginals of problem p -- all NMC_subroutine
// consumes (`np.concatenate(clusters)`, NPT/npt.py:403): the seeds {|mag| >= thresholds[0]} (every seed ends up in some
// cluster), then per further threshold the spins outside every cluster with |mag| >= threshold that have a neighbour
// (J != 0) inside the union as it stood BEFORE that threshold step (a cluster's neighbours are taken from its members
// at the start of its turn and spins claimed earlier in the same step are only excluded, so the union does not depend on
// the order the reference serves the clusters in: one synchronous expansion per threshold).
// -> mask[list[p]][n_pad] (1 = backbone spin).  A diverged inference (status 1: "LBP diverged at initial lambda") sets the
// sticky flag; its marginals are all zero, i.e. the mask is empty.
// (BIG: chains too long for LDS keep the two membership arrays in global scratch [problems][2 n_pad], csrc/nlmc_big.h)
template <bool BIG = false>
__global__ void k_cluster_mask(CsrDev g, const int32_t *list, const double *mag, const int32_t *lbp_status, const double *thresholds,
                               int n_thresholds, uint8_t *mask, int32_t *sticky, uint8_t *scratch_g)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = g.n, p = blockIdx.x, c = list ? list[p] : p, tid = threadIdx.x, nt = blockDim.x;
    uint8_t *cur = BIG ? scratch_g + (size_t)p * 2 * g.n_pad : lds_raw, *nxt = cur + g.n_pad;
    const double *m = mag + (size_t)p * n;
    if (tid == 0 && lbp_status[p] != 0) atomicOr(sticky, lbp_status[p]);
    const double t0 = thresholds[0];
    for (int i = tid; i < n; i += nt) cur[i] = fabs(m[i]) >= t0 ? 1 : 0;
    __syncthreads();
    for (int s = 1; s < n_thresholds; ++s) {
        const double th = thresholds[s];
        for (int i = tid; i < n; i += nt) {
            uint8_t v = cur[i];
            if (!v && fabs(m[i]) >= th) {
                for (int e = g.rowptr[i]; e < g.rowptr[i + 1] && !v; ++e)
                    if (g.val64[e] != 0.0 && cur[g.col[e]]) v = 1;       // J is symmetric: J[w, i] != 0 <=> J[i, w] != 0
            }
            nxt[i] = v;
        }
        __syncthreads();
        uint8_t *t = cur; cur = nxt; nxt = t;
    }
    uint8_t *dst = mask + (size_t)c * g.n_pad;
    for (int i = tid; i < g.n_pad; i += nt) dst[i] = i < n ? cur[i] : 0;
}

// Phase flags of the three NMC phases from the backbone mask (NPT/npt.py:406-414,425,441 == NMC/nmc.py:377-381,398-401):
//   kind 1 (clusters hot):   backbone spins SCALED (row / temp_x), every other spin frozen at its start value
//   kind 2 (clusters frozen): backbone spins frozen at their start value, the others plain
// A spin frozen by h = +-10000 m_init keeps the value it starts the phase with (tanh saturates to +-1 exactly), and the
// device-RNG kernels treat both frozen codes as "unchanged": the code written is FROZEN_UP whatever the spin.
__global__ void k_phase_flags(int n_pad, const int32_t *list, const uint8_t *mask, int kind, uint8_t *flags)
{
    const int c = list ? list[blockIdx.x] : (int)blockIdx.x;
    const uint8_t *m = mask + (size_t)c * n_pad;
    uint8_t *f = flags + (size_t)c * n_pad;
    for (int i = threadIdx.x; i < n_pad; i += blockDim.x)
        f[i] = kind == 1 ? (m[i] ? 1 : 2) : (m[i] ? 2 : 0);
}
