// nlmc_pt_icm.h -- replica-exchange round and Houdayer iso-cluster move kernels (gfx950).
#pragma once
#include "nlmc_kernels.h"

// ------------------------------------------------------------------------------------------------------
// parallel tempering: label exchange
// ------------------------------------------------------------------------------------------------------
__global__ void k_pt_apply_swap(int32_t *slot_of_chain, int32_t *chain_of_slot, int L, int ladder, int sa, int sb)
{
    const int ca = chain_of_slot[(size_t)ladder * L + sa], cb = chain_of_slot[(size_t)ladder * L + sb];
    slot_of_chain[ca] = sb;
    slot_of_chain[cb] = sa;
    chain_of_slot[(size_t)ladder * L + sa] = cb;
    chain_of_slot[(size_t)ladder * L + sb] = ca;
}

struct PtSwapArgs {
    int ladder_len, n_pairs;
    uint32_t round, seed_lo, seed_hi;
    const double *beta;         // [ladder_len]
    const double *energies;     // [n_chains_global] or nullptr
    const long long *efix;      // [n_chains_global] (single-GPU) or nullptr
    int escale;
    int32_t *slot_of_chain, *chain_of_slot;
    int32_t *out_pairs;         // [n_ladders][n_pairs][2]
    uint8_t *out_acc;           // [n_ladders][n_pairs]
    int32_t *status;            // sticky: 1 = "Cannot find non-overlapping pairs."
    const int32_t *plan_pairs;  // this round's planned selection [n_ladders][n_pairs][2] or nullptr
    const int32_t *plan_ok;     // [n_ladders]
    int ladder0, chain_base;    // a context that owns WHOLE ladders decides those alone: block b = ladder ladder0 + b, efix by local chain id
};

// position of the r-th (0-based) set bit of w (r < popcount(w)): binary search on popcounts
__device__ __forceinline__ int nth_set_bit(unsigned long long w, int r)
{
    int pos = 0;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const int c = __popcll((w >> pos) & ((1ull << sft) - 1ull));
        if (r >= c) { r -= c; pos += sft; }
    }
    return pos;
}

// One wave per ladder.  Pair selection restates select_non_overlapping_pairs (NPT/npt.py:514-533): repeatedly pick
// uniformly among the still-available adjacent pairs (ascending list), then drop the pairs that share a replica with
// the pick.  The availability set lives in registers: lane l holds pairs [64 l, 64 l + 64) as a bit mask, the pick is
// located with a wave prefix sum of popcounts (ladder_len <= 4096).  Acceptance: u < min(1, exp(dBeta dE))
// (NPT/npt.py:668-671), one lane per selected pair.
// Pair selection of one (round, ladder) by one wave -> pairs[n_pairs][2] (slot indices), returns 0 on exhaustion.
// The selection depends on the RNG only (never on energies or states), so whole runs can be planned ahead
// (k_pt_select, one wave per round and ladder) and the per-round kernel is left with the parallel acceptance test.
__device__ __forceinline__ int pt_select_pairs(int L, int n_pairs, uint32_t round, uint32_t g, uint32_t seed_lo,
                                               uint32_t seed_hi, int32_t *pairs, int lane)
{
    const int npairs_all = L - 1;
    const int rem = npairs_all - lane * 64;
    unsigned long long word = rem >= 64 ? ~0ull : (rem > 0 ? ((1ull << rem) - 1ull) : 0ull);
    int cnt = npairs_all;
    for (int p = 0; p < n_pairs; ++p) {
        if (cnt == 0) return 0;
        const uint32_t r = philox4x32_10((uint32_t)p, round, g, NLMC_TAG_PAIR, seed_lo, seed_hi).x;
        const int idx = (int)(((unsigned long long)r * (unsigned long long)cnt) >> 32);
        const int pc = __popcll(word);
        int incl = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
        const int before = incl - pc;
        const bool mine = idx >= before && idx < before + pc;
        const int owner = __ffsll((long long)__ballot(mine)) - 1;
        const int bit = nth_set_bit(word, mine ? idx - before : 0);
        const int i = __shfl(lane * 64 + bit, owner, 64);
        if (lane == 0) { pairs[2 * p] = i; pairs[2 * p + 1] = i + 1; }
        int cleared = 0;
#pragma unroll
        for (int q = -1; q <= 1; ++q) {
            const int b = i + q;
            if (b >= 0 && b < npairs_all && (b >> 6) == lane && ((word >> (b & 63)) & 1ull)) { word &= ~(1ull << (b & 63)); ++cleared; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cleared += __shfl_xor(cleared, o, 64);
        cnt -= cleared;
    }
    return 1;
}

struct PtSelectArgs {
    int ladder_len, n_pairs, n_ladders;
    uint32_t round0, seed_lo, seed_hi;
    int32_t *plan_pairs;        // [n_rounds][n_ladders][n_pairs][2]
    int32_t *plan_ok;           // [n_rounds][n_ladders]
};

__global__ void k_pt_select(PtSelectArgs a)      // grid = n_rounds * n_ladders, block = 64
{
    const int r = blockIdx.x / a.n_ladders, g = blockIdx.x % a.n_ladders, lane = threadIdx.x;
    int32_t *pairs = a.plan_pairs + (size_t)blockIdx.x * a.n_pairs * 2;
    const int ok = pt_select_pairs(a.ladder_len, a.n_pairs, a.round0 + (uint32_t)r, (uint32_t)g, a.seed_lo, a.seed_hi, pairs, lane);
    if (lane == 0) a.plan_ok[blockIdx.x] = ok;
}

// One wave per ladder.  Pair selection restates select_non_overlapping_pairs (NPT/npt.py:514-533): repeatedly pick
// uniformly among the still-available adjacent pairs (ascending list), then drop the pairs that share a replica with
// the pick.  The availability set lives in registers: lane l holds pairs [64 l, 64 l + 64) as a bit mask, the pick is
// located with a wave prefix sum of popcounts (ladder_len <= 4096).  Acceptance: u < min(1, exp(dBeta dE))
// (NPT/npt.py:668-671), one lane per selected pair.
__global__ void k_pt_swap(PtSwapArgs a)
{
    const int L = a.ladder_len, g = a.ladder0 + blockIdx.x, lane = threadIdx.x, nt = blockDim.x;     // nt == 64 unless the selection is planned
    int32_t *pairs = a.out_pairs + (size_t)g * a.n_pairs * 2;
    const int32_t *sel = pairs;          // where the acceptance step reads this round's selection from
    int good;
    if (a.plan_pairs) {                                                  // selection planned ahead: read in place (no copy +
        sel = a.plan_pairs + (size_t)g * a.n_pairs * 2;                  // barrier + re-read on the latency path of a round)
        good = a.plan_ok[g];
    } else {
        good = pt_select_pairs(L, a.n_pairs, a.round, (uint32_t)g, a.seed_lo, a.seed_hi, pairs, lane);
    }
    if (!good) {
        if (lane == 0) atomicExch(a.status, 1);
        for (int p = lane; p < a.n_pairs; p += nt) { a.out_acc[(size_t)g * a.n_pairs + p] = 0; pairs[2 * p] = pairs[2 * p + 1] = -1; }
        return;
    }
    if (!a.plan_pairs) __syncthreads();      // pairs[] written by the selection are read by all lanes below
    const double inv = __longlong_as_double((long long)(1023 - a.escale) << 52);
    for (int p = lane; p < a.n_pairs; p += nt) {
        const int i = sel[2 * p];
        const int ca = a.chain_of_slot[(size_t)g * L + i], cb = a.chain_of_slot[(size_t)g * L + i + 1];
        const double Ea = a.energies ? a.energies[ca] : (double)a.efix[ca - a.chain_base] * inv;
        const double Eb = a.energies ? a.energies[cb] : (double)a.efix[cb - a.chain_base] * inv;
        const double dE = Eb - Ea, dB = a.beta[i + 1] - a.beta[i];
        const u32x4 r = philox4x32_10((uint32_t)p, a.round, (uint32_t)g, NLMC_TAG_SWAP, a.seed_lo, a.seed_hi);
        const double u = uniform_from(r, 0.0);
        const double z = (dB * dE) * 1.4426950408889634;
        const bool acc = u < exp2_spec(z);
        if (acc) {   // selected pairs are disjoint -> no two lanes touch the same entries
            a.slot_of_chain[ca] = i + 1;
            a.slot_of_chain[cb] = i;
            a.chain_of_slot[(size_t)g * L + i] = cb;
            a.chain_of_slot[(size_t)g * L + i + 1] = ca;
        }
        a.out_acc[(size_t)g * a.n_pairs + p] = acc ? 1 : 0;
        if (a.plan_pairs) { pairs[2 * p] = i; pairs[2 * p + 1] = sel[2 * p + 1]; }      // the round's log
    }
}

// ------------------------------------------------------------------------------------------------------
// iso-cluster move
// ------------------------------------------------------------------------------------------------------
struct IcmArgs {
    CsrDev g;
    const int8_t *spins;      // [n_chains][n_pad]
    const int32_t *pairs;     // [n_pairs][2] local chain ids
    int32_t *label;           // [n_pairs][n]  min member index of the component, or INT_MAX if the spins agree
    int32_t *info;            // [n_pairs][2]  {n_components, picked size}
    const uint4 *adj;         // 16-bit adjacency table [n][2] (k_fused_adjacency) or nullptr: use the CSR entries and
                              // test `val != 0` like the reference (needed when a stored coupling is 0)
};

// Connected components of the sub-graph induced by {k : s_a[k] s_b[k] = -1} (NPT/apt_ICM.py:116-143): union-find in
// LDS.  lab[k] always points to a member of k's component with an index <= k; roots are hooked onto SMALLER roots
// (atomicMin), so the final root of a component is its smallest member and ascending labels reproduce the
// reference's list order.  A handful of hook rounds with path compression replace the O(diameter) rounds of plain
// label propagation.
__device__ __forceinline__ int icm_find(int32_t *lab, int k)
{
    int r = lab[k];
    while (true) { const int up = lab[r]; if (up == r) break; r = up; }
    if (r != k) atomicMin(&lab[k], r);                 // path compression; atomicMin: a concurrent hook of k onto a smaller
    return r;                                          // root (atomicMin on lab[k] as well) can never be overwritten
}

__device__ __forceinline__ int icm_find_halving(int32_t *lab, int k)
{
    // (k_icm_round only: its hooks are compare-and-swaps that succeed on true roots alone, so a node that has a parent is never
    // written by a hook and the plain stores below cannot lose one.)  Root of k with path halving on the way (every node visited is re-pointed at its grandparent, as in ECL-CC): parents always
    // have SMALLER indices than their children and a node that has a parent never becomes a root again, so a racing store can
    // only replace one ancestor by another -- the forest stays a forest -- and the long chains that "hook the larger root under
    // the smaller" leaves behind in a giant component are halved by every find that walks them (k_icm_round spent 53 of 83
    // thousand cycles in its merge pass with compression of the start node only).
    int prev = k, cur = lab[k];
    if (cur == k) return k;
    int next;
    while (cur > (next = lab[cur])) {
        lab[prev] = next;
        prev = cur;
        cur = next;
    }
    return cur;
}

__global__ void k_icm_components(IcmArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    int32_t *lab = reinterpret_cast<int32_t *>(lds_raw);
    __shared__ int nroots, ncand;
    const int n = a.g.n, tid = threadIdx.x, nt = blockDim.x, p = blockIdx.x;
    uint16_t *cand = reinterpret_cast<uint16_t *>(lab + n);          // the disagreeing spins, compacted (any order)
    const int8_t *sa = a.spins + (size_t)a.pairs[2 * p] * a.g.n_pad;
    const int8_t *sb = a.spins + (size_t)a.pairs[2 * p + 1] * a.g.n_pad;
    if (tid == 0) { nroots = 0; ncand = 0; }
    __syncthreads();
    for (int k = tid; k < n; k += nt) {
        const bool d = (int)sa[k] * (int)sb[k] == -1;
        lab[k] = d ? k : INT_MAX;
        if (d) cand[atomicAdd(&ncand, 1)] = (uint16_t)k;
    }
    __syncthreads();
    const int nc = ncand;
    // Hook rounds over the candidates only.  Neighbour lists come from the 16-bit adjacency table (two 16-byte loads
    // per spin, absent slots hold the spin itself) when every stored coupling is non-zero -- the common case --
    // otherwise from the CSR entries with the reference's `val != 0` test (NPT/apt_ICM.py:129).
    bool converged = false;                // (workgroup-uniform)
    for (int it = 0; it <= n; ++it) {
        int changed = 0;
        for (int idx = tid; idx < nc; idx += nt) {
            const int k = (int)cand[idx];
            int rk;
            auto hook = [&](int j) {
                if (j == k || lab[j] == INT_MAX) return;
                const int rj = icm_find(lab, j);
                if (rj < rk) { atomicMin(&lab[rk], rj); rk = rj; changed = 1; }
                else if (rk < rj) { atomicMin(&lab[rj], rk); changed = 1; }
            };
            if (a.adj) {
                const uint4 a0 = a.adj[2 * k], a1 = a.adj[2 * k + 1];
                const int rs = a.g.rowptr[k], deg = a.g.rowptr[k + 1] - rs;
                rk = icm_find(lab, k);
                const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int q = 0; q < NLMC_FZ_ADJ; ++q) hook((int)((aw[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu));
                for (int e = rs + NLMC_FZ_ADJ; e < rs + deg; ++e) hook(a.g.col[e]);
            } else {
                const int rs = a.g.rowptr[k], deg = a.g.rowptr[k + 1] - rs;
                EdgeQ ed[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) ed[q] = a.g.edge32[rs + q];
                rk = icm_find(lab, k);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < deg && (ed[q].q != 0 || a.g.val64[rs + q] != 0.0)) hook(ed[q].col);
                for (int e = rs + 8; e < rs + deg; ++e) {
                    const EdgeQ t = a.g.edge32[e];
                    if (t.q != 0 || a.g.val64[e] != 0.0) hook(t.col);
                }
            }
        }
        if (!__syncthreads_or(changed)) { converged = true; break; }
    }
    int cnt = 0;
    int32_t *out = a.label + (size_t)p * n;
    for (int k = tid; k < n; k += nt) {
        int l = lab[k];
        if (l != INT_MAX) l = icm_find(lab, k);
        out[k] = l;
        cnt += (l == k);
    }
    if (cnt) atomicAdd(&nroots, cnt);
    __syncthreads();
    // n_components = -1: the hook rounds hit their iteration cap (cannot happen for a finite graph; the host reports it)
    if (tid == 0) { a.info[2 * p] = converged ? nroots : -1; a.info[2 * p + 1] = 0; }
}

// ---- one fused kernel per batch of device-decided moves (round 3) ---------------------------------------------------
// Components, pick, move and the energy bookkeeping of one pair in ONE workgroup: the labels never leave LDS, both
// configurations are staged in LDS, and the tracked energies of the two chains are updated INCREMENTALLY in the fixed-point
// model of the sweep kernels (exact integers: dE = 2 sum_{i in C} s_i (sum_{j not in C} Jq_ij s_j + hq_i) for the exchanged
// cluster C -- only bonds that leave the cluster count --, dE = 2 hq.s for the global flip of the Katzgraber variant):
// no k_energy pass over all chains afterwards.  Round 2 ran k_icm_components (57 us) + k_icm_move (12 us) + k_energy over
// all 256 chains (30 us) per APT round at N = 10^4.
struct IcmRoundArgs {
    CsrDev g;
    int8_t *spins;            // [n_chains][n_pad]
    const int32_t *pairs;     // [n_pairs][2] local chain ids, or nullptr: the Houdayer pairing of the APT round is made here --
    int pair_R, pair_K;       // block p pairs the ladders ranked 2i, 2i+1 among the K ladders of slot r = p / (K/2), i = p % (K/2),
    const int32_t *chain_of_slot;   // ranks by the keys philox(j, round, r, ICM_PAIR) (k_icm_pair_ladders' law, NPT/apt_ICM.py:216-222)
    int32_t *info;            // [n_pairs][2]  {n_components (-1: hook rounds did not converge), picked size}
    const uint4 *adj;         // 16-bit adjacency table or nullptr (see IcmArgs)
    uint32_t round, seed_lo, seed_hi;
    int katz, chain_base;
    long long *efix;          // [n_chains] tracked energies, units 2^-escale
    double *energy_sink;      // [n_chains] or nullptr
    int eshift, escale;       // escale - qs
    int lds_cand_off, lds_sa_off, lds_sb_off;
    int slot0;                // pairing on the fly: global index of local slot 0 (the pairing keys carry the GLOBAL slot)
    int rng_stride, rng_base; // != 0: the pick is keyed by the (ladder, global slot) ids of the two chains instead of their chain ids
};

__global__ __launch_bounds__(1024) void k_icm_round(IcmRoundArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    int32_t *lab = reinterpret_cast<int32_t *>(lds_raw);
    uint16_t *cand = reinterpret_cast<uint16_t *>(lds_raw + a.lds_cand_off);
    int8_t *sa = reinterpret_cast<int8_t *>(lds_raw + a.lds_sa_off), *sb = reinterpret_cast<int8_t *>(lds_raw + a.lds_sb_off);
    __shared__ int nroots, ncand, sh_root, sh_size;
    __shared__ int sh_scan[17];
    __shared__ long long sh_dE[2];
    const int n = a.g.n, n_pad = a.g.n_pad, tid = threadIdx.x, nt = blockDim.x, p = blockIdx.x;
    __shared__ int sh_pair[2], sh_lad[2];
    if (!a.pairs) {                        // pairing on the fly (pair_K <= blockDim.x: checked by the host)
        const int half = a.pair_K / 2, r = p / half, i = p % half;
        const uint32_t rg = (uint32_t)(r + a.slot0);
        if (tid < a.pair_K) {
            const uint32_t kj = philox4x32_10((uint32_t)tid, a.round, rg, 6u /*NLMC_TAG_ICM_PAIR*/, a.seed_lo, a.seed_hi).x;
            int rank = 0;
            for (int q = 0; q < a.pair_K; ++q) {
                if (q == tid) continue;
                const uint32_t kq = philox4x32_10((uint32_t)q, a.round, rg, 6u, a.seed_lo, a.seed_hi).x;
                rank += (kq < kj) || (kq == kj && q < tid);
            }
            if (rank == 2 * i || rank == 2 * i + 1) { sh_pair[rank & 1] = a.chain_of_slot[(size_t)tid * a.pair_R + r]; sh_lad[rank & 1] = tid; }
        }
        __syncthreads();
    }
    const int ca = a.pairs ? a.pairs[2 * p] : sh_pair[0], cb = a.pairs ? a.pairs[2 * p + 1] : sh_pair[1];
    int8_t *ga = a.spins + (size_t)ca * n_pad, *gb = a.spins + (size_t)cb * n_pad;
    if (tid == 0) { nroots = 0; ncand = 0; sh_size = 0; sh_dE[0] = 0; sh_dE[1] = 0; sh_root = -1; }
    for (int i = tid; i < n_pad / 16; i += nt) {
        reinterpret_cast<int4 *>(sa)[i] = reinterpret_cast<const int4 *>(ga)[i];
        reinterpret_cast<int4 *>(sb)[i] = reinterpret_cast<const int4 *>(gb)[i];
    }
    __syncthreads();
    for (int k = tid; k < n; k += nt) {
        const bool d = (int)sa[k] * (int)sb[k] == -1;
        lab[k] = d ? k : INT_MAX;
        if (d) cand[atomicAdd(&ncand, 1)] = (uint16_t)k;
    }
    __syncthreads();
    const int nc = ncand;
    // ONE pass over the edges of the disagreement sub-graph (round 3; it was "hook, then repeat until nothing changes": ~5 rounds
    // over all candidates).  Every edge (k, j) is merged to completion where it is met: find both roots, hook the LARGER root
    // under the smaller one with a compare-and-swap that only succeeds while the larger one still IS a root, and on failure
    // (somebody else hooked it first) look the roots up again -- the loop ends when both ends share a root, so no edge is ever
    // left for a later round.  Roots only ever move to smaller indices: the final root of a component is its smallest member
    // (the order find_disagreement_clusters lists the clusters in, NPT/apt_ICM.py:120-141).  Each edge is taken from its
    // larger end only.
    const bool converged = true;
    auto merge = [&](int k, int j) {
        if (j >= k || lab[j] == INT_MAX) return;           // (j == k: the padding of the adjacency table / the diagonal)
        int ra = icm_find_halving(lab, k), rb = icm_find_halving(lab, j);
        while (ra != rb) {
            if (ra < rb) { const int t = ra; ra = rb; rb = t; }            // ra > rb
            const int old = atomicCAS(&lab[ra], ra, rb);
            if (old == ra) break;                                          // ra was a root and now hangs under rb
            ra = icm_find_halving(lab, old);
            rb = icm_find_halving(lab, rb);
        }
    };
    if (a.adj) {
        // neighbour lists of the NEXT candidate of this thread are in flight while the current one is merged
        int idx = tid, k = 0, rs = 0, deg = 0;
        uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
        if (idx < nc) { k = (int)cand[idx]; a0 = a.adj[2 * k]; a1 = a.adj[2 * k + 1]; rs = a.g.rowptr[k]; deg = a.g.rowptr[k + 1] - rs; }
        while (idx < nc) {
            const int idn = idx + nt;
            int kn = 0, rsn = 0, degn = 0;
            uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0;
            if (idn < nc) { kn = (int)cand[idn]; b0 = a.adj[2 * kn]; b1 = a.adj[2 * kn + 1]; rsn = a.g.rowptr[kn]; degn = a.g.rowptr[kn + 1] - rsn; }
            const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int q = 0; q < NLMC_FZ_ADJ; ++q) merge(k, (int)((aw[q >> 1] >> ((q & 1) * 16)) & 0xFFFFu));
            for (int e = rs + NLMC_FZ_ADJ; e < rs + deg; ++e) merge(k, a.g.col[e]);
            idx = idn; k = kn; rs = rsn; deg = degn; a0 = b0; a1 = b1;
        }
    } else {
        for (int idx = tid; idx < nc; idx += nt) {
            const int k = (int)cand[idx];
            const int rs = a.g.rowptr[k], deg = a.g.rowptr[k + 1] - rs;
            EdgeQ ed[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ed[q] = a.g.edge32[rs + q];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < deg && (ed[q].q != 0 || a.g.val64[rs + q] != 0.0)) merge(k, ed[q].col);
            for (int e = rs + 8; e < rs + deg; ++e) {
                const EdgeQ t = a.g.edge32[e];
                if (t.q != 0 || a.g.val64[e] != 0.0) merge(k, t.col);
            }
        }
    }
    __syncthreads();
    // final labels (every candidate points at its root), number of components
    int cnt = 0;
    for (int idx = tid; idx < nc; idx += nt) {
        const int k = (int)cand[idx];
        const int l = icm_find_halving(lab, k);
        cnt += (l == k);
    }
    if (cnt) atomicAdd(&nroots, cnt);
    __syncthreads();
    for (int idx = tid; idx < nc; idx += nt) { const int k = (int)cand[idx]; lab[k] = icm_find_halving(lab, k); }
    __syncthreads();
    const int ncomp = converged ? nroots : -1;
    if (ncomp <= 0) {                       // nothing to move (identical or opposite... no disagreement), or not converged
        if (tid == 0) { a.info[2 * p] = ncomp; a.info[2 * p + 1] = 0; }
        return;
    }
    // pick component number floor(r ncomp / 2^32) in ascending-label order (NPT/apt_ICM.py:232-233)
    uint32_t ida = (uint32_t)(a.chain_base + ca), idb = (uint32_t)(a.chain_base + cb);
    if (a.rng_stride && !a.pairs) {        // keyed by (ladder, global slot): the same pick whichever chains sit on the two places
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
    for (int idx = tid; idx < nc; idx += nt) csz += (lab[(int)cand[idx]] == root);
    if (csz) atomicAdd(&sh_size, csz);
    __syncthreads();
    const int size = sh_size;
    long long dEa = 0, dEb = 0;
    if (a.katz && size > n / 2) {
        // state_1 = -state_1 (NPT/apt_ICM.py:236-237): the coupling term is even in s, the field term changes sign
        for (int k = tid; k < n; k += nt) { dEa += 2ll * (long long)a.g.hq[k] * (long long)sa[k]; ga[k] = (int8_t)(-sa[k]); }
    } else {
        // exchange the cluster between the two states (both flip on it: they disagree there)
        for (int idx = tid; idx < nc; idx += nt) {
            const int k = (int)cand[idx];
            if (lab[k] != root) continue;
            const int rs = a.g.rowptr[k], re = a.g.rowptr[k + 1];
            long long fa = a.g.hq[k], fb = fa;
            EdgeQ ed[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ed[q] = a.g.edge32[rs + q];     // unconditional (padded array): eight loads in flight
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (rs + q < re && lab[ed[q].col] != root) {            // bonds inside the cluster (and the diagonal) keep their energy
                    fa += (long long)ed[q].q * (long long)sa[ed[q].col];
                    fb += (long long)ed[q].q * (long long)sb[ed[q].col];
                }
            }
            for (int q = rs + 8; q < re; ++q) {
                const EdgeQ t = a.g.edge32[q];
                if (lab[t.col] == root) continue;
                fa += (long long)t.q * (long long)sa[t.col];
                fb += (long long)t.q * (long long)sb[t.col];
            }
            dEa += 2ll * (long long)sa[k] * fa;
            dEb += 2ll * (long long)sb[k] * fb;
            ga[k] = sb[k];
            gb[k] = sa[k];
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

// Houdayer pairing on the device (NPT/apt_ICM.py:216-222): for every temperature slot r the K ladders (sub-replicas)
// are shuffled -- order = sort of the keys philox(j, round, r, ICM_PAIR) -- and paired (sh[0], sh[1]), (sh[2], sh[3]), ...;
// a pair is written as the two LOCAL chains that currently hold slot r in those ladders.
#define NLMC_TAG_ICM_PAIR 6u
__global__ void k_icm_pair_ladders(int R, int K, uint32_t round, uint32_t seed_lo, uint32_t seed_hi,
                                   const int32_t *chain_of_slot, int32_t *pairs)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (slot r, ladder j)
    if (id >= R * K) return;
    const int r = id / K, j = id % K, half = K / 2;
    const uint32_t kj = philox4x32_10((uint32_t)j, round, (uint32_t)r, NLMC_TAG_ICM_PAIR, seed_lo, seed_hi).x;
    int rank = 0;                              // position of ladder j in the shuffled order of slot r
    for (int i = 0; i < K; ++i) {
        if (i == j) continue;
        const uint32_t ki = philox4x32_10((uint32_t)i, round, (uint32_t)r, NLMC_TAG_ICM_PAIR, seed_lo, seed_hi).x;
        rank += (ki < kj) || (ki == kj && i < j);
    }
    if (rank < 2 * half) pairs[((size_t)r * half + rank / 2) * 2 + (rank & 1)] = chain_of_slot[(size_t)j * R + r];
}

struct IcmMoveArgs {
    CsrDev g;
    int8_t *spins;
    const int32_t *pairs;
    const int32_t *label;
    int32_t *info;
    long long pick_host;
    int use_philox;
    uint32_t round, seed_lo, seed_hi;
    int katz;
    int chain_base;
};

// Pick component number (pick mod n_components) in ascending-label order and apply the move
// (NPT/apt_ICM.py:232-246).
__global__ void k_icm_move(IcmMoveArgs a)
{
    __shared__ int sh_scan[17];
    __shared__ int sh_root, sh_size;
    const int n = a.g.n, tid = threadIdx.x, nt = blockDim.x, p = blockIdx.x;
    const int ncomp = a.info[2 * p];
    if (ncomp <= 0) return;
    const int ca = a.pairs[2 * p], cb = a.pairs[2 * p + 1];
    int8_t *sa = a.spins + (size_t)ca * a.g.n_pad;
    int8_t *sb = a.spins + (size_t)cb * a.g.n_pad;
    const int32_t *lab = a.label + (size_t)p * n;
    int idx;
    if (a.use_philox) {
        const uint32_t r = philox4x32_10((uint32_t)(a.chain_base + ca), a.round, (uint32_t)(a.chain_base + cb), NLMC_TAG_ICM,
                                         a.seed_lo, a.seed_hi).x;
        idx = (int)(((unsigned long long)r * (unsigned long long)ncomp) >> 32);
    } else {
        idx = (int)(a.pick_host % (long long)ncomp);
    }
    // locate the idx-th root (label[k] == k) in ascending k: contiguous chunk per thread + block scan
    const int chunk = (n + nt - 1) / nt;
    const int b = min(tid * chunk, n), e = min(b + chunk, n);
    int mine = 0;
    for (int k = b; k < e; ++k) mine += (lab[k] == k);
    const int lane = tid & 63, wv = tid >> 6;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) sh_scan[wv] = incl;
    if (tid == 0) sh_size = 0;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += sh_scan[w];
    const int before = base + incl - mine;
    if (idx >= before && idx < before + mine) {
        int seen = before;
        for (int k = b; k < e; ++k)
            if (lab[k] == k) { if (seen == idx) { sh_root = k; break; } ++seen; }
    }
    __syncthreads();
    const int root = sh_root;
    int cnt = 0;
    for (int k = tid; k < n; k += nt) cnt += (lab[k] == root);
    if (cnt) atomicAdd(&sh_size, cnt);
    __syncthreads();
    const int size = sh_size;
    if (a.katz && size > n / 2) {
        for (int k = tid; k < n; k += nt) sa[k] = (int8_t)(-sa[k]);
    } else {
        for (int k = tid; k < n; k += nt)
            if (lab[k] == root) { const int8_t t = sa[k]; sa[k] = sb[k]; sb[k] = t; }
    }
    if (tid == 0) a.info[2 * p + 1] = size;
}
