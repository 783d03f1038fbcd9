"""Shared test helpers (synthetic instances of SURVEY.md section 8d, fixed-point scale, stream drawing)."""

import numpy as np
import scipy.sparse as sp


def make_instance(N, seed=20250225, with_h=False, gaussian=False):
    """Undirected random graph with exactly 3N distinct edges (mean degree 6), J=+-1 (or |gaussian|-weighted)."""
    r = np.random.default_rng(seed)
    edges = set()
    while len(edges) < 3 * N:
        need = 3 * N - len(edges)
        ij = r.integers(0, N, size=(need + 16, 2))
        for i, j in ij:
            if i != j:
                edges.add((min(int(i), int(j)), max(int(i), int(j))))
                if len(edges) == 3 * N:
                    break
    e = np.array(sorted(edges), dtype=np.int64)
    w = r.choice([-1.0, 1.0], size=len(e))
    if gaussian:
        w = w * np.abs(r.standard_normal(len(e)))
        w /= np.max(np.abs(w))
    J = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]]))),
                      shape=(N, N)).tocsr()
    J.sort_indices()
    h = (r.standard_normal(N) * 0.3) if with_h else np.zeros(N)
    return J, h


def init_spins(R, N, base=1000):
    """m0 = sign(U - 0.5) from default_rng(1000 + chain)  (SURVEY.md section 8d)."""
    out = np.empty((R, N), dtype=np.int8)
    for c in range(R):
        out[c] = np.where(np.random.default_rng(base + c).random(N) < 0.5, -1, 1)
    return out


def energy_scale(csr, h):
    """log2 of the engine's fixed-point energy scale for oracle.Csr `csr` (nlmc_create's rule, restated in
    oracle/nlo.c:nlo_field_scale)."""
    import oracle
    return oracle.field_scale(csr, h)[1]


def draw_stream(R, S, N):
    """Legacy-stream draws in the reference's program order: chain by chain, sweep by sweep: permutation(N), N x rand()."""
    perm = np.empty((R, S, N), dtype=np.int32)
    u = np.empty((R, S, N), dtype=np.float64)
    for c in range(R):
        for t in range(S):
            perm[c, t] = np.random.permutation(N)
            u[c, t] = np.random.rand(N)
    return perm, u


class DeviceBuffer:
    """A float64 device buffer allocated through the HIP runtime the engine already loaded (ctypes), so that tests can
    hand a device pointer to the C-ABI without importing torch (a second HIP runtime in the same process)."""

    def __init__(self, host_array):
        import ctypes
        self._hip = ctypes.CDLL("libamdhip64.so")
        self._ct = ctypes
        a = np.ascontiguousarray(host_array, dtype=np.float64)
        self.ptr = ctypes.c_void_p()
        assert self._hip.hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(a.nbytes)) == 0
        assert self._hip.hipMemcpy(self.ptr, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes), 1) == 0   # H2D

    def read(self, count):
        out = np.empty(count, dtype=np.float64)
        assert self._hip.hipDeviceSynchronize() == 0
        assert self._hip.hipMemcpy(out.ctypes.data_as(self._ct.c_void_p), self.ptr, self._ct.c_size_t(out.nbytes), 2) == 0  # D2H
        return out

    def free(self):
        self._hip.hipFree(self.ptr)
