"""Bit-exact emulation of NumPy's pairwise summation for SPARSE vectors.

The reference sums dense length-N columns/rows that are mostly zeros (`np.sum(u_msgs[:, i])`, NMC/nmc.py:201;
`np.sum(np.abs(self.J), axis=1)`, NMC/nmc.py:353).  NumPy reduces a 1-D (possibly strided) double vector with
`pairwise_sum`: 8 running accumulators over blocks of <= 128 elements, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
the <8 leftover elements added one by one, and blocks combined by a binary recursion that halves n (rounded down to
a multiple of 8).  Its convergence test (tolerance = machine epsilon) makes loopy BP sensitive to the last bit of
these sums, so the host restates the exact association on the non-zero entries only: adding 0.0 never changes a
partial sum, hence only the POSITION of each non-zero decides which accumulator / block it lands in.

The plan depends on (N, sparsity pattern) only and is built once per graph; a sum then costs a few np.bincount calls.
"""
import numpy as np


def _leaves(n):
    """Leaf blocks of numpy's pairwise recursion for length n: list of (start, length, node_id, depth) with heap ids."""
    out = []

    def rec(start, m, node, depth):
        if m <= 128:
            out.append((start, m, node, depth))
            return
        n2 = m // 2
        n2 -= n2 % 8
        rec(start, n2, 2 * node, depth + 1)
        rec(start + n2, m - n2, 2 * node + 1, depth + 1)

    rec(0, n, 1, 0)
    return out


class SparsePairwiseSum:
    """sum over `pos` of values grouped by `grp`, each group emulating np.sum of a dense length-n vector.

    grp, pos: int arrays of equal length, sorted by (grp, pos) ascending, pos unique within a group."""

    def __init__(self, n, n_groups, grp, pos):
        grp = np.asarray(grp, dtype=np.int64)
        pos = np.asarray(pos, dtype=np.int64)
        self.n_groups = int(n_groups)
        self.nnz = grp.shape[0]
        lv = _leaves(int(n))
        starts = np.array([l[0] for l in lv], dtype=np.int64)
        lens = np.array([l[1] for l in lv], dtype=np.int64)
        nodes = np.array([l[2] for l in lv], dtype=np.int64)
        depths = np.array([l[3] for l in lv], dtype=np.int64)
        leaf = np.searchsorted(starts, pos, side="right") - 1
        off = pos - starts[leaf]
        m = lens[leaf]
        main = m - (m % 8)
        is_acc = (m >= 8) & (off < main)
        # compact (group, leaf) slots, in (grp, leaf) order
        key = grp * len(lv) + leaf
        uniq, inv = np.unique(key, return_inverse=True)
        self.n_slots = uniq.shape[0]
        self.acc_sel = np.nonzero(is_acc)[0]
        self.acc_idx = inv[self.acc_sel] * 8 + (off[self.acc_sel] % 8)
        self.rest_sel = np.nonzero(~is_acc)[0]
        rest_off = np.where(m >= 8, off - main, off)
        self.rest_idx = inv[self.rest_sel] * 7 + rest_off[self.rest_sel]
        # tree combination plan: items = (group, node) with a depth; merge children into parents level by level
        item_grp = uniq // len(lv)
        item_node = nodes[uniq % len(lv)]
        item_depth = depths[uniq % len(lv)]
        self.stages = []
        maxd = int(item_depth.max()) if uniq.size else 0
        for d in range(maxd, 0, -1):
            up = item_depth == d
            new_node = np.where(up, item_node >> 1, item_node)
            new_depth = np.where(up, d - 1, item_depth)
            k2 = item_grp * (1 << (maxd + 2)) + new_node
            # order: by destination, children left (even id) before right (odd id)
            order = np.lexsort((item_node, k2))
            u2, inv2 = np.unique(k2[order], return_inverse=True)
            self.stages.append((order, inv2, u2.shape[0]))
            first = np.concatenate([[True], u2[1:] != u2[:-1]]) if u2.size else np.zeros(0, bool)
            del first
            # representative attributes of the merged items
            rep = np.zeros(u2.shape[0], dtype=np.int64)
            rep[inv2] = order            # any member works for grp/new_node/new_depth
            item_grp, item_node, item_depth = item_grp[rep], new_node[rep], new_depth[rep]
        self.final_grp = item_grp      # one item per group that has any entry (node 1)

    def __call__(self, values):
        v = np.asarray(values, dtype=np.float64)
        acc = np.bincount(self.acc_idx, weights=v[self.acc_sel], minlength=self.n_slots * 8).reshape(-1, 8)
        res = ((acc[:, 0] + acc[:, 1]) + (acc[:, 2] + acc[:, 3])) + ((acc[:, 4] + acc[:, 5]) + (acc[:, 6] + acc[:, 7]))
        if self.rest_sel.size:
            rest = np.bincount(self.rest_idx, weights=v[self.rest_sel], minlength=self.n_slots * 7).reshape(-1, 7)
            for r in range(7):
                res = res + rest[:, r]
        cur = res
        for order, inv2, cnt in self.stages:
            cur = np.bincount(inv2, weights=cur[order], minlength=cnt)
        out = np.zeros(self.n_groups)
        out[self.final_grp] = cur
        return out
