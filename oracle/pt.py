"""oracle/pt.py -- TEST INFRASTRUCTURE ONLY.  Sequential restatement of the device-decided swap round
(k_pt_swap, include/nlmc.h: nlmc_pt_swap_philox): pair selection with the law of NPT/npt.py:514-533, Metropolis
test u < min(1, exp(dBeta dE)) of NPT/npt.py:668-671, Philox-keyed by (seed, round, ladder)."""
import numpy as np

from . import philox, lib

LOG2E = 1.4426950408889634
TAG_SWAP, TAG_PAIR, TAG_ICM = 3, 4, 5


def swap_round(E, slot_of_chain, betas, ladder_len, n_pairs, rnd, seed, ladder0=0):
    """ladder0: global index of the first ladder of the block handed in (the Philox key of a ladder is its GLOBAL index: a
    context that owns whole ladders decides them alone, csrc/nlmc_pt_icm.h: k_pt_swap)."""
    lo, hi = int(seed) & 0xFFFFFFFF, int(seed) >> 32
    G = len(E)
    slots = np.array(slot_of_chain, dtype=np.int32).copy()
    pairs_all, acc_all = [], []
    for g in range(G // ladder_len):
        chain_of_slot = np.empty(ladder_len, dtype=int)
        for c in range(g * ladder_len, (g + 1) * ladder_len):
            chain_of_slot[slots[c]] = c
        avail = list(range(ladder_len - 1))
        sel = []
        for p in range(n_pairs):
            if not avail:
                raise ValueError("Cannot find non-overlapping pairs.")
            r = int(philox(p, rnd, ladder0 + g, TAG_PAIR, lo, hi)[0])
            i = avail[(r * len(avail)) >> 32]
            sel.append(i)
            avail = [q for q in avail if abs(q - i) > 1]
        for p, i in enumerate(sel):
            ca, cb = chain_of_slot[i], chain_of_slot[i + 1]
            w = philox(p, rnd, ladder0 + g, TAG_SWAP, lo, hi)
            u = ((int(w[0]) >> 5) * 67108864.0 + (int(w[1]) >> 6)) / 9007199254740992.0
            z = ((betas[i + 1] - betas[i]) * (E[cb] - E[ca])) * LOG2E
            acc = u < lib().nlo_exp2_f64(z)
            if acc:
                slots[ca], slots[cb] = i + 1, i
                chain_of_slot[i], chain_of_slot[i + 1] = cb, ca
            pairs_all.append((i, i + 1))
            acc_all.append(int(acc))
    nl = G // ladder_len
    return slots, np.array(pairs_all, dtype=np.int32).reshape(nl, n_pairs, 2), np.array(acc_all, dtype=np.uint8).reshape(nl, n_pairs)
