"""Bridge between the golden-pinned kernel and the timed one ("next" item 2a).

`k_sweep_stream` (fp64, tanh, externally drawn permutation + uniforms) is pinned to the reference's golden vectors
(tests/test_gpu_sweep.py).  The throughput kernels draw their own numbers from Philox.  Here the host derives, from
the Philox SPEC alone (oracle.philox + the documented counter layout), the permutation and the uniforms the
throughput mode uses, feeds them to the stream kernel, and requires the same spins:
  * f64 mode: identical, every sweep of every chain;
  * "f32" mode (fixed-point couplings, logistic threshold): identical on a +-J instance; on Gaussian couplings the
    two sides see fields that differ by the stated quantisation (<= deg 2^-(qs+1)), so single updates with the uniform
    within ~1e-6 of the acceptance probability may differ -- they are COUNTED sweep by sweep from identical start states
    and must stay a handful.
The reference's rule under test: m_k = sign(tanh(beta x_k) - 2u + 1), NMC/nmc.py:86-87.
"""
import numpy as np
import pytest

import oracle
from helpers import make_instance, init_spins

TAG_UNIFORM, TAG_ORDER = 1, 2


def philox_stream(n, R, S, seed, f64, sweep0=0):
    """(perm [S, n], u [R, S, n] indexed by visiting position) of the throughput mode, shared order."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    perm = np.empty((S, n), np.int32)
    u_spin = np.empty((R, S, n), np.float64)
    for t in range(S):
        keys = np.array([int(oracle.philox(k, sweep0 + t, 0, TAG_ORDER, lo, hi)[0]) for k in range(n)], dtype=np.uint64)
        perm[t] = np.lexsort((np.arange(n), keys))
        for c in range(R):
            if f64:
                for b in range((n + 3) // 4):          # high 27 bits: the UNIFORM call; low 26 bits: the same word of the UNIFORM_LO call (tag 7)
                    r = oracle.philox(b, sweep0 + t, c, TAG_UNIFORM, lo, hi).astype(np.uint64)
                    q = oracle.philox(b, sweep0 + t, c, 7, lo, hi).astype(np.uint64)
                    for j in range(4):
                        k = 4 * b + j
                        if k < n:
                            u_spin[c, t, k] = (float(r[j] >> np.uint64(5)) * 67108864.0 + float(q[j] >> np.uint64(6))) / 9007199254740992.0
            else:
                for b in range((n + 3) // 4):
                    r = oracle.philox(b, sweep0 + t, c, TAG_UNIFORM, lo, hi)
                    for j in range(4):
                        k = 4 * b + j
                        if k < n:
                            u_spin[c, t, k] = (float(r[j]) + 0.5) / 4294967296.0
    u = np.stack([np.stack([u_spin[c, t][perm[t]] for t in range(S)]) for c in range(R)])
    return perm, u


@pytest.mark.gpu
@pytest.mark.parametrize("gaussian", [False, True])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_stream_kernel_fed_with_the_philox_spec_equals_the_philox_kernel(product, precision, gaussian):
    N, R, S, seed = 240, 6, 12, 0x1234ABCD5
    J, h = make_instance(N, seed=31, with_h=gaussian, gaussian=gaussian)
    betas = np.geomspace(0.1, 3.5, R)
    btab = np.repeat(betas[:, None], S, axis=1)
    m0 = init_spins(R, N)
    perm, u = philox_stream(N, R, S, seed, precision == "f64")
    with product.Engine(J, h, R) as eng:
        eng.set_spins(m0)
        ph = eng.sweep_philox(S, seed, beta=btab, precision=precision, record_stride=1)["spins"]       # [R, S, N]
        if precision == "f64" or not gaussian:
            eng.set_spins(m0)
            st = eng.sweep_stream(np.broadcast_to(perm, (R, S, N)), u, btab, record_stride=1)["spins"]
            assert np.array_equal(ph, st)
            return
        # Gaussian couplings, fixed-point field vs fp64 field: count single-sweep differences from identical states
        differing = 0
        for t in range(S):
            eng.set_spins(m0 if t == 0 else ph[:, t - 1])
            st = eng.sweep_stream(np.broadcast_to(perm[t:t + 1], (R, 1, N)), u[:, t:t + 1], btab[:, t:t + 1], record_stride=1)["spins"]
            differing += int(np.count_nonzero(st[:, 0] != ph[:, t]))
        assert differing <= 3, f"{differing} of {R * S * N} updates differ"
