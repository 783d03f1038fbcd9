"""Instance text loaders of the reference's example drivers, building CSR directly (the reference fills a dense
N x N array first: NMC/examples/wishart_example.py:8-47, chimera_example.py:8-40, DCL_example.py:8-47,
contrived_wishart_example.py:8-57).  Same file formats (`i j value` per line, `#` comments), same conventions and
return types: (scipy.sparse.csr_matrix W, h as an [N,1] column).  As in the reference the caller flips the sign
(J = -W, h = -h) to match the Hamiltonian E = -(m^T J m / 2 + m^T h)."""
import numpy as np
import scipy.sparse as sp


def _parse(txtfile, one_based, diag_to_h):
    W, hv = {}, {}
    with open(txtfile, "r") as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            x = list(map(float, line.split()))
            i, j = int(x[0]) - one_based, int(x[1]) - one_based
            if i == j:
                if diag_to_h:
                    hv[i] = x[2]
                continue
            W[(i, j)] = x[2]          # dict semantics of the reference: a later line overwrites an earlier one
            W[(j, i)] = x[2]
    N = max(max(W.keys())) + 1
    keys = np.array(list(W.keys()), dtype=np.int64).reshape(-1, 2)
    vals = np.array(list(W.values()), dtype=np.float64)
    A = sp.csr_matrix((vals, (keys[:, 0], keys[:, 1])), shape=(N, N))
    A.eliminate_zeros()               # csr_matrix(dense) of the reference holds no explicit zeros
    A.sort_indices()
    h = np.zeros((N, 1))
    for i, v in hv.items():
        if i < N:
            h[i] = v
    return A, h


def txt_to_A_wishart(txtfile):
    """0-based indices, diagonal lines ignored, h = 0 (NMC/examples/wishart_example.py:8-47)."""
    return _parse(txtfile, 0, False)


def txt_to_A_DCL(txtfile):
    """Same format as the Wishart files (NMC/examples/DCL_example.py:8-47)."""
    return _parse(txtfile, 0, False)


def txt_to_A_droplet(txtfile):
    """1-based indices, `i i value` lines are the biases (NMC/examples/chimera_example.py:8-40)."""
    return _parse(txtfile, 1, True)


def txt_to_A_wishart_contrived_tree(txtfile):
    """0-based indices, `i i value` lines are the biases (NMC/examples/contrived_wishart_example.py:8-57)."""
    return _parse(txtfile, 0, True)
