"""oracle/ -- TEST INFRASTRUCTURE ONLY (CPU checker for the HIP hot path).

Allowed callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product package
(`nonlocal-monte-carlo_amd/`) must never import this module.

Parity status: PINNED against golden vectors captured from the reference itself
(tools/make_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).

`oracle/_ref/`: the reference is pure Python/NumPy (no C sources), so there is nothing to compile from
/root/reference; the reference-backed leg of the oracle is the golden fixtures.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnlo.so")
_LIB = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    """gcc -> oracle/_build/libnlo.so (git-ignored, travels with the gpurun snapshot)."""
    src = os.path.join(_HERE, "nlo.c")
    if not force and os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-o", _SO, src, "-lm"])
    return _SO


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.nlo_sweeps_stream.restype = ctypes.c_int
        L.nlo_sweeps_stream.argtypes = [ctypes.c_int, _i32p, _i32p, _f64p, _f64p, ctypes.c_int, _i32p, _f64p, _f64p,
                                        _f64p, ctypes.c_void_p]
        L.nlo_energy.restype = ctypes.c_double
        L.nlo_energy.argtypes = [ctypes.c_int, _i32p, _i32p, _f64p, _f64p, _i8p]
        L.nlo_clusters.restype = ctypes.c_int
        L.nlo_clusters.argtypes = [ctypes.c_int, _i32p, _i32p, _f64p, _i8p, _i8p, _i32p]
        L.nlo_philox.restype = None
        L.nlo_philox.argtypes = [ctypes.c_uint32] * 6 + [_u32p]
        L.nlo_exp2_f32.restype = ctypes.c_float
        L.nlo_exp2_f32.argtypes = [ctypes.c_float]
        L.nlo_exp2_f64.restype = ctypes.c_double
        L.nlo_exp2_f64.argtypes = [ctypes.c_double]
        L.nlo_field_scale.restype = ctypes.c_int
        L.nlo_field_scale.argtypes = [ctypes.c_int, _i32p, _f64p, _f64p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.nlo_threshold_f32.restype = ctypes.c_float
        L.nlo_threshold_f32.argtypes = [ctypes.c_uint32]
        L.nlo_threshold_count.restype = ctypes.c_uint64
        L.nlo_threshold_count.argtypes = [ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
        L.nlo_sweeps_philox.restype = ctypes.c_int
        L.nlo_sweeps_philox.argtypes = [ctypes.c_int, _i32p, _i32p, _f64p, _f64p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_uint32, _f64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                        ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, _i8p, _i64p, ctypes.c_void_p,
                                        ctypes.c_void_p]
        _LIB = L
    return _LIB


class Csr:
    """CSR view of J with sorted column indices and explicit zeros dropped (== scipy.sparse.csr_matrix(dense))."""

    def __init__(self, J):
        import scipy.sparse as sp
        A = sp.csr_matrix(J)
        A = A.copy()
        A.eliminate_zeros()
        A.sort_indices()
        self.n = int(A.shape[0])
        self.indptr = np.ascontiguousarray(A.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        self.data = np.ascontiguousarray(A.data, dtype=np.float64)

    @classmethod
    def from_parts(cls, n, indptr, indices, data):
        self = cls.__new__(cls)
        self.n = int(n)
        self.indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        return self

    def toarray(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.data, self.indices, self.indptr), shape=(self.n, self.n)).toarray()


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def sweeps_stream(csr, h, m_start, beta_run, perm, u, want_M=True):
    """a-1 with a pre-drawn legacy stream.  Returns (M int8 [S,N] or None, m_final float64 [N])."""
    n = csr.n
    S = len(beta_run)
    perm = np.ascontiguousarray(perm, dtype=np.int32).reshape(S, n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(S, n)
    m = np.ascontiguousarray(np.asarray(m_start, dtype=np.float64).reshape(-1).copy())
    M = np.empty((S, n), dtype=np.int8) if want_M else None
    lib().nlo_sweeps_stream(n, csr.indptr, csr.indices, csr.data, np.ascontiguousarray(h, dtype=np.float64).reshape(-1),
                            S, perm, u, np.ascontiguousarray(beta_run, dtype=np.float64), m, _ptr(M))
    return M, m


def energy(csr, h, m):
    return float(lib().nlo_energy(csr.n, csr.indptr, csr.indices, csr.data,
                                  np.ascontiguousarray(h, dtype=np.float64).reshape(-1),
                                  np.ascontiguousarray(m, dtype=np.int8).reshape(-1)))


def clusters(csr, s1, s2):
    """a-9: list of sorted member arrays, in the reference's list order."""
    lab = np.empty(csr.n, dtype=np.int32)
    nc = lib().nlo_clusters(csr.n, csr.indptr, csr.indices, csr.data, np.ascontiguousarray(s1, dtype=np.int8),
                            np.ascontiguousarray(s2, dtype=np.int8), lab)
    return [np.nonzero(lab == c)[0] for c in range(nc)]


def philox(c0, c1, c2, c3, k0, k1):
    out = np.empty(4, dtype=np.uint32)
    lib().nlo_philox(c0, c1, c2, c3, k0, k1, out)
    return out


LOG2E = 1.4426950408889634


def field_scale(csr, h):
    """(qs, escale) of the "f32" throughput mode: couplings are held as rint(J 2^qs) (24-bit fixed point), energies as
    integers in units of 2^-escale (restated from nlmc_create)."""
    qs, es = ctypes.c_int(0), ctypes.c_int(0)
    lib().nlo_field_scale(csr.n, csr.indptr, csr.data, np.ascontiguousarray(h, dtype=np.float64).reshape(-1),
                          ctypes.byref(qs), ctypes.byref(es))
    return int(qs.value), int(es.value)


def threshold(r):
    """W(r) ~= log2((1-u)/u), u = (r + 1/2) / 2^32: the logistic threshold of the "f32" throughput mode."""
    return float(lib().nlo_threshold_f32(int(r) & 0xFFFFFFFF))


def threshold_count(z, r0=0, r1=1 << 32, stride=1):
    """Number of r on the lattice r0, r0 + stride, ... < r1 with z < W(r)  (s' = +1 at that z)."""
    return int(lib().nlo_threshold_count(float(z), int(r0), int(r1), int(stride)))


def threshold_cdf(z, stride=1):
    """P(s' = +1 | z) estimated on every `stride`-th of the 2^32 values of r."""
    return threshold_count(z, 0, 1 << 32, stride) / float(-(-(1 << 32) // stride))


def cb_pair(beta, temp_x=1.0, use_f64=False):
    """(T)(-2 log2(e) beta) for normal and `scaled` spins, rounded like the product does at upload."""
    a = -2.0 * LOG2E * float(beta)
    b = -2.0 * LOG2E * (float(beta) / float(temp_x))
    if use_f64:
        return a, b
    return float(np.float32(a)), float(np.float32(b))


def sweeps_philox(csr, h, s_start, cb_run, seed, chain_id, order_group=0, sweep0=0, flags=None, escale=None,
                  use_f64=False, efix0=0, want_M=True):
    """Sequential spec of the throughput mode for ONE chain.  cb_run: [S,2] (normal, scaled).  escale: log2 of the
    fixed-point energy unit (default: the instance's own, field_scale)."""
    n = csr.n
    if escale is None:
        escale = field_scale(csr, h)[1]
    cb_run = np.ascontiguousarray(cb_run, dtype=np.float64).reshape(-1, 2)
    S = cb_run.shape[0]
    s = np.ascontiguousarray(np.asarray(s_start, dtype=np.int8).reshape(-1).copy())
    ef = np.array([efix0], dtype=np.int64)
    M = np.empty((S, n), dtype=np.int8) if want_M else None
    tr = np.empty(S, dtype=np.int64)
    fl = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint8)
    rc = lib().nlo_sweeps_philox(n, csr.indptr, csr.indices, csr.data, np.ascontiguousarray(h, dtype=np.float64).reshape(-1),
                                 int(bool(use_f64)), S, int(sweep0), cb_run, int(seed) & 0xFFFFFFFF,
                                 (int(seed) >> 32) & 0xFFFFFFFF, int(chain_id), int(order_group), _ptr(fl), int(escale),
                                 s, ef, _ptr(M), _ptr(tr))
    if rc != 0:
        raise ValueError("escale outside [qs, qs + 29] of this instance (oracle.field_scale)")
    return M, s, tr
