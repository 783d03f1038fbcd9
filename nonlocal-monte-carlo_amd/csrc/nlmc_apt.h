// nlmc_apt.h -- replica exchange of an APT run whose temperature ladder is cut into SLOT blocks, one per GPU (gfx950).
//
// NPT/apt_ICM.py:215-246 pairs the sub-replicas of ONE temperature for the iso-cluster move, :248-285 swaps neighbouring
// temperatures of one sub-replica.  So that every temperature's K sub-replicas stay on one GPU, rank w of W
// owns the slots [w Rw, (w+1) Rw) of every one of the K sub-replica ladders (R = W Rw): K local ladders of Rw slots.  Inside a
// rank a swap is a label exchange as everywhere else; a swap across the boundary between rank w's top slot and rank w+1's bottom
// slot moves the two CONFIGURATIONS (n_pad int8 each + the tracked energy) -- one neighbour exchange per round.  Random numbers
// are keyed by (ladder, GLOBAL slot) instead of by chain (nlmc_apt_shard), so "exchange the labels" and "exchange the
// configurations" are the same Markov step and the run is bit-identical for any W.
//
// A round's swap step on every rank:  k_apt_pack  (tracked energies by slot -> this rank's block of the gathered vector; the
// chains on the first / last local slot of every ladder -> send buffers)  ->  all-gather of K Rw int64 per rank + one
// send/receive with each neighbour (K n_pad bytes)  ->  k_apt_swap  (EVERY rank evaluates the identical Philox-keyed decision for
// every selected pair of the global ladder from the gathered energies, applies the pairs inside its block as label exchanges and
// notes the accepted boundary pairs)  ->  k_apt_adopt  (an accepted boundary pair: the local chain on that slot takes the
// neighbour's configuration and energy).
#pragma once
#include "nlmc_pt_icm.h"

struct AptPackArgs {
    int L, K, n_pad;                 // local slots per ladder, ladders
    const int32_t *chain_of_slot;    // [K][L] local chain ids
    const long long *efix;           // [K L]
    const int8_t *spins;             // [K L][n_pad]
    long long *e_block;              // [K][L] this rank's block of the gathered vector: tracked energy of the chain on (ladder, slot)
    int8_t *send;                    // [2][K][n_pad]: the chains on slot 0 (towards rank - 1) | on slot L - 1 (towards rank + 1)
};

__global__ void k_apt_pack(AptPackArgs a)     // grid = 2 K + 1
{
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (b == 2 * a.K) {
        for (int i = tid; i < a.K * a.L; i += nt) a.e_block[i] = a.efix[a.chain_of_slot[i]];
        return;
    }
    const int side = b / a.K, j = b % a.K;
    const int ch = a.chain_of_slot[(size_t)j * a.L + (side ? a.L - 1 : 0)];
    const int4 *src = reinterpret_cast<const int4 *>(a.spins + (size_t)ch * a.n_pad);
    int4 *dst = reinterpret_cast<int4 *>(a.send + ((size_t)side * a.K + j) * a.n_pad);
    for (int i = tid; i < a.n_pad / 16; i += nt) dst[i] = src[i];
}

struct AptSwapArgs {
    int L, R, K, n_pairs, world, rank;
    uint32_t round, seed_lo, seed_hi;
    const double *beta;              // [R] the GLOBAL ladder
    const long long *e_all;          // [world][K][L] gathered tracked energies, by (rank, ladder, local slot)
    int escale;
    int32_t *slot_of_chain, *chain_of_slot;   // local maps [K L]
    int32_t *out_pairs;              // [K][n_pairs][2] GLOBAL slots
    uint8_t *out_acc;                // [K][n_pairs]
    int32_t *status;
    const int32_t *plan_pairs;       // this round's planned selection over the global ladder [K][n_pairs][2], or nullptr
    const int32_t *plan_ok;          // [K]
    int32_t *bd;                     // [2][K] out: boundary pair accepted (side 0: with rank - 1, side 1: with rank + 1)
};

// One workgroup per ladder; the same selection, keys and arithmetic as k_pt_swap on a ladder of R slots (so that W ranks of
// R / W slots reproduce one rank of R slots bit for bit).
__global__ void k_apt_swap(AptSwapArgs a)
{
    const int L = a.L, j = blockIdx.x, lane = threadIdx.x, nt = blockDim.x;
    int32_t *pairs = a.out_pairs + (size_t)j * a.n_pairs * 2;
    const int32_t *sel = pairs;
    int good;
    if (lane == 0) { a.bd[j] = 0; a.bd[a.K + j] = 0; }
    if (a.plan_pairs) {
        sel = a.plan_pairs + (size_t)j * a.n_pairs * 2;
        good = a.plan_ok[j];
    } else {
        good = pt_select_pairs(a.R, a.n_pairs, a.round, (uint32_t)j, a.seed_lo, a.seed_hi, pairs, lane);      // (one wave: nt == 64)
    }
    __syncthreads();
    if (!good) {
        if (lane == 0) atomicExch(a.status, 1);
        for (int p = lane; p < a.n_pairs; p += nt) { a.out_acc[(size_t)j * a.n_pairs + p] = 0; pairs[2 * p] = pairs[2 * p + 1] = -1; }
        return;
    }
    const double inv = __longlong_as_double((long long)(1023 - a.escale) << 52);
    for (int p = lane; p < a.n_pairs; p += nt) {
        const int i = sel[2 * p];                                        // global slots i, i + 1
        const int wa = i / L, wb = (i + 1) / L, la = i - wa * L, lb = i + 1 - wb * L;
        const double Ea = (double)a.e_all[((size_t)wa * a.K + j) * L + la] * inv;
        const double Eb = (double)a.e_all[((size_t)wb * a.K + j) * L + lb] * inv;
        const double dE = Eb - Ea, dB = a.beta[i + 1] - a.beta[i];
        const u32x4 r = philox4x32_10((uint32_t)p, a.round, (uint32_t)j, NLMC_TAG_SWAP, a.seed_lo, a.seed_hi);
        const double u = uniform_from(r, 0.0);
        const double z = (dB * dE) * 1.4426950408889634;
        const bool acc = u < exp2_spec(z);
        if (acc) {
            if (wa == a.rank && wb == a.rank) {                          // inside this rank's block: label exchange
                const int ca = a.chain_of_slot[(size_t)j * L + la], cb = a.chain_of_slot[(size_t)j * L + lb];
                a.slot_of_chain[ca] = lb;
                a.slot_of_chain[cb] = la;
                a.chain_of_slot[(size_t)j * L + la] = cb;
                a.chain_of_slot[(size_t)j * L + lb] = ca;
            } else if (wa == a.rank) a.bd[a.K + j] = 1;                  // my top slot <-> the next rank's bottom slot
            else if (wb == a.rank) a.bd[j] = 1;                          // my bottom slot <-> the previous rank's top slot
        }
        a.out_acc[(size_t)j * a.n_pairs + p] = acc ? 1 : 0;
        if (a.plan_pairs) { pairs[2 * p] = i; pairs[2 * p + 1] = i + 1; }
    }
}

struct AptAdoptArgs {
    int L, K, n_pad, world, rank, escale;
    const int32_t *bd;               // [2][K]
    const int32_t *chain_of_slot;    // [K][L]
    const int8_t *recv;              // [2][K][n_pad]: from rank - 1 (its top-slot chains) | from rank + 1 (its bottom-slot chains)
    const long long *e_all;          // [world][K][L]
    int8_t *spins;
    long long *efix;
    double *energy_sink;             // or nullptr
};

__global__ void k_apt_adopt(AptAdoptArgs a)   // grid = 2 K
{
    const int side = blockIdx.x / a.K, j = blockIdx.x % a.K, tid = threadIdx.x, nt = blockDim.x;
    if (!a.bd[(size_t)side * a.K + j]) return;
    // (the pairs of a round share no slot: the chain on a boundary slot of an accepted boundary pair took part in no other swap
    // of this round, it is still the chain k_apt_pack sent)
    const int ch = a.chain_of_slot[(size_t)j * a.L + (side ? a.L - 1 : 0)];
    const int4 *src = reinterpret_cast<const int4 *>(a.recv + ((size_t)side * a.K + j) * a.n_pad);
    int4 *dst = reinterpret_cast<int4 *>(a.spins + (size_t)ch * a.n_pad);
    for (int i = tid; i < a.n_pad / 16; i += nt) dst[i] = src[i];
    if (tid == 0) {
        const long long e = side ? a.e_all[((size_t)(a.rank + 1) * a.K + j) * a.L] : a.e_all[((size_t)(a.rank - 1) * a.K + j) * a.L + a.L - 1];
        a.efix[ch] = e;
        if (a.energy_sink) a.energy_sink[ch] = (double)e * __longlong_as_double((long long)(1023 - a.escale) << 52);
    }
}
