// nlmc_lbp.h -- convexified loopy belief propagation on the edge list (gfx950), SURVEY.md section 8 row f-1.
//
// Reference: NMC/nmc.py:93-228 (== NPT/npt.py:129-265).  The reference keeps dense N x N message matrices; here every
// stored CSR entry e = (i -> j) of row i carries
//     w[e]  = u_msgs[j, i]   the message INCOMING to i from j (so a node's total is a contiguous row sum)
//     hm[e] = h_msgs[i, j]   the cavity field of i without j
// 1-8 workgroups of 1024 threads per problem (one m_star), see k_lbp; all lambdas of LBP_convexified run inside one launch; Jacobi
// iteration with ping-pong message buffers in global memory (L2 resident: 2 * nnz doubles per problem).
// Arithmetic: fp64 in the reference's operation order, row sums sequential in ascending neighbour index (the
// reference's `total` uses NumPy's pairwise association and NumPy's own tanh/arctanh, so results agree to rounding,
// not bit for bit -- the bit-exact path is the host restatement in lbp.py; tolerance stated in tests/test_gpu_lbp.py).
#pragma once
#include "nlmc_kernels.h"

#define NLMC_LBP_THREADS 1024

struct LbpArgs {
    int n, nnz, n_lams, max_iter;
    const int32_t *rowptr, *col, *src, *rev;
    const double *val;          // [nnz] normalised J
    const double *tJ;           // [nnz] tanh(beta J)
    const double *h;            // [n]
    const double *eps;          // [n]  |h| + sum_j |J_ij|            (NMC/nmc.py:353)
    const double *m_star;       // [P][n]
    const double *lams;         // [n_lams]
    double beta, inv_beta, tol, sat;   // sat = tanh(19.06) - eps: clip bound of atanh_saturated (NMC/nmc.py:230-255)
    double usat;                       // atanh(sat) / beta
    double *w0, *w1, *hm;       // [P][nnz]
    double *tot;                // [P][n]
    double *mag;                // [P][n]   final marginals
    double *mag_all;            // [P][n_lams][n] or nullptr
    int32_t *out_nlam;          // [P] number of lambdas processed (entries of the reference's marginals dict)
    int32_t *out_iters;         // [P][n_lams] last iteration index per lambda
    int32_t *out_status;        // [P] 0 ok, 1 = "LBP diverged at initial lambda", 2 = group barrier timed out
    // a problem may be spread over `group` workgroups (k_lbp: blockIdx = p * group + g), see lbp_group_barrier
    int group;
    unsigned int *bar;          // [P] arrival counters (zeroed before the launch)
    double *part;               // [P][2][group][4] partial maxima of the convergence test, double-buffered
};

__global__ void k_lbp_src(int n, const int32_t *rowptr, int32_t *src)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) src[e] = i;
}

// rev[e] = position of (j -> i) in row j; flag[0] = 1 when the pattern is not symmetric
__global__ void k_lbp_rev(int nnz, const int32_t *rowptr, const int32_t *col, const int32_t *src, int32_t *rev, int32_t *flag)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nnz) return;
    const int i = src[e], j = col[e];
    int lo = rowptr[j], hi = rowptr[j + 1] - 1, found = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c == i) { found = mid; break; }
        if (c < i) lo = mid + 1; else hi = mid - 1;
    }
    if (found < 0) { flag[0] = 1; found = e; }
    rev[e] = found;
}

__global__ void k_lbp_tanhJ(int nnz, const double *val, double beta, double *tJ)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nnz) tJ[e] = tanh(beta * val[e]);
}

// u = atanh(clip(a tanh(y), +-sat)) / beta with one exponential, one logarithm and two divisions instead of
// tanh + atanh (330 fp64 instructions with the library functions, 135 like this):
//   E = exp(-2|y|), a' = a sign(y):  a tanh(y) = a'(1-E)/(1+E),
//   atanh(x) = log((1+x)/(1-x))/2 = log1p( 2a'(1-E) / ((1-a') + E(1+a')) )/2     (no cancellation: 1-E from expm1)
// |x| > sat  <=>  |a'|(1-E) > sat(1+E): the saturated value usat = atanh(sat)/beta comes from the host.
// expm1 / log1p are plain polynomial kernels (Taylor to r^13 after the usual 2^n split; fdlibm's log series with the
// rounding error of 1+z fed back): measured on 2*10^7 random (a, y) with |atanh a| <= 3 against long double, the
// message is within 1.6e-14 relative (tanh + atanh of the C library: 3.6e-15; both are limited by the conditioning
// 1/(1-x^2) of atanh near the saturated messages).
__device__ __forceinline__ double lbp_expm1_neg(double x, double &E)      // x <= 0; returns e^x - 1, E = e^x
{
    const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    x = fmax(x, -80.0);
    const double nf = rint(x * LOG2E);
    double r = fma(-nf, LN2_HI, x);
    r = fma(-nf, LN2_LO, r);                                               // |r| <= ln2 / 2
    double q = 1.0 / 87178291200.0;                                        // (e^r - 1)/r = sum r^i / (i+1)!
    q = fma(q, r, 1.0 / 6227020800.0);
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    const double em1r = r * q;
    const double s = ldexp(1.0, (int)nf);
    E = fma(s, em1r, s);
    return nf == 0.0 ? em1r : E - 1.0;
}

__device__ __forceinline__ double lbp_log1p(double z)                      // z > -1
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double u = 1.0 + z;
    const double c = z - (u - 1.0);                                        // u + c == 1 + z
    int k;
    double m = frexp(u, &k);                                               // u = m 2^k, m in [1/2, 1)
    const bool low = m < 0.70710678118654752440;
    m = low ? m * 2.0 : m;
    k = low ? k - 1 : k;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z2 = s * s, w = z2 * z2;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z2 * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    const double lo = dk * LN2_LO + c / u;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + lo)) - f);
}

__device__ __forceinline__ double lbp_message(double a, double y, double sat, double usat, double inv_beta)
{
    const double ap = copysign(a, a * y);
    double E;
    const double em1 = lbp_expm1_neg(-2.0 * fabs(y), E);                   // E - 1 in (-1, 0]
    const double u = 0.5 * inv_beta * lbp_log1p((-2.0 * ap * em1) / ((1.0 - ap) + E * (1.0 + ap)));
    return (fabs(ap) * (-em1) > sat * (1.0 + E)) ? copysign(usat, ap) : u;
}

__device__ __forceinline__ double lbp_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Barrier between the `group` workgroups that share one problem (they run on different CUs, possibly on different
// XCDs whose L2s are not coherent): every thread releases its stores at agent scope, one lane arrives on a monotonic
// counter and polls it with relaxed agent-scope loads, then every thread acquires (invalidates its CU's L1).  The
// launch keeps problems * group <= number of CUs, so all workgroups are resident; the poll is BOUNDED all the same --
// on a timeout the problem is flagged (status 2) and its workgroups leave.  Returns false on timeout.
__device__ __forceinline__ bool lbp_group_barrier(unsigned int *counter, unsigned int target, int tid)
{
    __shared__ int ok;
    __threadfence();                                            // release (agent): write back this thread's stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int budget = 1 << 22;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && --budget > 0)
            __builtin_amdgcn_s_sleep(2);
        ok = budget > 0;
    }
    __syncthreads();
    __threadfence();                                            // acquire (agent): drop stale L1 lines
    return ok != 0;
}

// blockIdx.x = p * group + g: workgroup g of problem p owns the nodes [i0, i1) and their CSR rows [e0, e1).  Per
// iteration: node totals of the own nodes (they read messages other workgroups scattered into the own rows during the
// previous iteration), messages of the own edges (scattered into the neighbours' rows of the other buffer), ONE group
// barrier that also carries the four partial maxima of the convergence test.  Row sums are sequential and maxima are
// order-independent, so the result does not depend on `group`.
__global__ __launch_bounds__(NLMC_LBP_THREADS) void k_lbp(LbpArgs a)
{
    __shared__ double red[4][NLMC_LBP_THREADS / 64];
    __shared__ double res[4];
    const int G = a.group, p = blockIdx.x / G, g = blockIdx.x % G;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n, nnz = a.nnz;
    const int i0 = (int)((long long)n * g / G), i1 = (int)((long long)n * (g + 1) / G);
    const int e0 = a.rowptr[i0], e1 = a.rowptr[i1];
    double *wc = a.w0 + (size_t)p * nnz, *wn = a.w1 + (size_t)p * nnz;
    double *hm = a.hm + (size_t)p * nnz, *tot = a.tot + (size_t)p * n, *mag = a.mag + (size_t)p * n;
    const double *ms = a.m_star + (size_t)p * n;
    unsigned int seq = 0;                                   // group barriers passed so far
    bool dead = false;

    // h_msgs = 0, u_msgs = J * m_star.reshape(1, -1)  (NMC/nmc.py:128-129): u_msgs[j, i] = J[j, i] * m_star[i]
    for (int e = e0 + tid; e < e1; e += NLMC_LBP_THREADS) {
        wc[e] = a.val[a.rev[e]] * ms[a.src[e]];
        hm[e] = 0.0;
    }
    for (int i = i0 + tid; i < i1; i += NLMC_LBP_THREADS) tot[i] = 0.0;
    if (G > 1) dead = !lbp_group_barrier(a.bar + p, ++seq * (unsigned)G, tid);
    else __syncthreads();

    int n_done = 0, status = 0;
    for (int l = 0; l < a.n_lams && !dead; ++l) {
        const double lam = a.lams[l];
        int it = 0;
        for (int iter = 0; iter < a.max_iter; ++iter) {
            it = iter;
            double dh_n = 0.0, dh_d = 0.0, du_n = 0.0, du_d = 0.0;
            // ---- node totals: total_i = h_lam[i] + sum_k u_msgs[k, i]   (NMC/nmc.py:199-201)
            for (int i = i0 + tid; i < i1; i += NLMC_LBP_THREADS) {
                const int r0 = a.rowptr[i], r1 = a.rowptr[i + 1];
                double s = 0.0;
                int offdiag = 0;
                for (int e = r0; e < r1; ++e) { s += wc[e]; offdiag += a.col[e] != i; }
                const double hl = a.h[i] + lam * ms[i] * a.eps[i];
                const double t_new = hl + s, t_old = tot[i];
                tot[i] = t_new;
                if (offdiag < n - 1) {         // row i of the dense h_msgs has non-edge entries, all equal to total_i
                    dh_n = fmax(dh_n, fabs(t_new - t_old));
                    dh_d = fmax(dh_d, fabs(t_new) + fabs(t_old));
                }
            }
            __syncthreads();
            // ---- messages: h_msgs[i, j] = total_i - u_msgs[j, i];  u_msgs[i, j] = atanh_sat(tanh(bJ) tanh(b h_msgs)) / b
            // (VALU-bound: ~220 fp64 instructions per message; batching the gathers four messages deep changed nothing)
            for (int e = e0 + tid; e < e1; e += NLMC_LBP_THREADS) {
                const int i = a.src[e], r = a.rev[e];
                const double h_old = hm[e];
                const double h_new = (a.col[e] != i) ? tot[i] - wc[e] : 0.0;
                const double u_new = lbp_message(a.tJ[e], a.beta * h_new, a.sat, a.usat, a.inv_beta);
                const double u_old = wc[r];
                hm[e] = h_new;
                wn[r] = u_new;
                dh_n = fmax(dh_n, fabs(h_new - h_old));
                dh_d = fmax(dh_d, fabs(h_new) + fabs(h_old));
                du_n = fmax(du_n, fabs(u_new - u_old));
                du_d = fmax(du_d, fabs(u_new) + fabs(u_old));
            }
            dh_n = lbp_wave_max(dh_n); dh_d = lbp_wave_max(dh_d); du_n = lbp_wave_max(du_n); du_d = lbp_wave_max(du_d);
            if (lane == 0) { red[0][wv] = dh_n; red[1][wv] = dh_d; red[2][wv] = du_n; red[3][wv] = du_d; }
            __syncthreads();
            if (tid < 4) {
                double m = red[tid][0];
                for (int k = 1; k < NLMC_LBP_THREADS / 64; ++k) m = fmax(m, red[tid][k]);
                res[tid] = m;
                if (G > 1) a.part[(((size_t)p * 2 + (seq & 1u)) * G + g) * 4 + tid] = m;
            }
            if (G > 1) {
                const unsigned int slot = seq & 1u;
                if (!lbp_group_barrier(a.bar + p, ++seq * (unsigned)G, tid)) { dead = true; break; }
                if (tid < 4) {
                    double m = 0.0;
                    for (int k = 0; k < G; ++k) m = fmax(m, a.part[(((size_t)p * 2 + slot) * G + k) * 4 + tid]);
                    res[tid] = m;
                }
            }
            __syncthreads();
            { double *t = wc; wc = wn; wn = t; }
            // 0/0 = NaN compares false, like the reference's `du < tolerance and dh < tolerance` (NMC/nmc.py:212-213)
            const double dh = res[0] / res[1], du = res[2] / res[3];
            __syncthreads();                   // res is rewritten in the next iteration
            if (du < a.tol && dh < a.tol) break;
        }
        if (dead) break;
        if (tid == 0 && g == 0) a.out_iters[(size_t)p * a.n_lams + l] = it;
        const bool exhausted = (it == a.max_iter - 1);
        if (exhausted && l == 0) { status = 1; break; }          // NMC/nmc.py:142-144
        if (!exhausted) {
            // magnetizations = tanh(beta (h_lam + sum_k u_msgs[k, :]))  (NMC/nmc.py:216-217), rows added in ascending k
            for (int i = i0 + tid; i < i1; i += NLMC_LBP_THREADS) {
                double s = 0.0;
                for (int e = a.rowptr[i]; e < a.rowptr[i + 1]; ++e) s += wc[e];
                mag[i] = tanh(a.beta * ((a.h[i] + lam * ms[i] * a.eps[i]) + s));
            }
        }
        // exhausted at a later lambda: keep the previous marginals and stop (NMC/nmc.py:145-148)
        if (a.mag_all) {
            double *dst = a.mag_all + ((size_t)p * a.n_lams + l) * n;
            __syncthreads();
            for (int i = i0 + tid; i < i1; i += NLMC_LBP_THREADS) dst[i] = mag[i];
        }
        n_done = l + 1;
        if (exhausted) break;
        __syncthreads();
    }
    if (tid == 0 && (g == 0 || dead)) { a.out_nlam[p] = dead ? 0 : n_done; a.out_status[p] = dead ? 2 : status; }
}
