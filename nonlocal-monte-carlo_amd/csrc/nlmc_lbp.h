// nlmc_lbp.h -- convexified loopy belief propagation on the edge list (gfx950), SURVEY.md section 8 row f-1.
//
// Reference: NMC/nmc.py:93-228 (== NPT/npt.py:129-265).  The reference keeps dense N x N message matrices; here every
// stored CSR entry e = (i -> j) of row i carries
//     w[e]  = u_msgs[j, i]   the message INCOMING to i from j (so a node's total is a contiguous row sum)
//     hm[e] = h_msgs[i, j]   the cavity field of i without j
// 1-8 workgroups of NLMC_LBP_THREADS threads per problem (one m_star), see k_lbp (small instances: k_lbp_lds); all lambdas of LBP_convexified run inside one launch; Jacobi
// iteration with ping-pong message buffers in global memory (L2 resident: 2 * nnz doubles per problem).
// Arithmetic: fp64 in the reference's operation order, row sums sequential in ascending neighbour index (the
// reference's `total` uses NumPy's pairwise association and NumPy's own tanh/arctanh, so results agree to rounding,
// not bit for bit -- the bit-exact path is the host restatement in lbp.py; tolerance stated in tests/test_gpu_lbp.py).
#pragma once
#include "nlmc_kernels.h"

#define NLMC_LBP_THREADS 768     // 12 waves: 170 registers per lane (at 1024 threads the 128-register cap spilled 26 of them)

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
    int poll_budget;            // polls a workgroup spends waiting for its group (0: the default, ~4M)
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

// u = atanh(clip(a tanh(y), +-sat)) / beta with one exponential, one logarithm and two divisions instead of tanh + atanh (330
// fp64 instructions with the library functions; 211 in round 2's formulation as the compiler emitted it; ~95 now -- the kernels
// are fp64-issue bound, scripts/probes/fp64_issue_probe.hip):
//   every multiply-add is an explicit fma (the compiler's two-address v_fmac_f64 form copied each polynomial coefficient into
//   the accumulator first: 54 moves per message), a division is reciprocal + two Newton steps + one correction without the
//   scaling / fix-up instructions of the general-purpose expansion; the saturated value usat = atanh(sat)/beta comes from the host.
// Agreement with the reference's tanh / arctanh: marginals within 1e-10 on every lambda both sides converge on
// (tests/test_gpu_lbp.py, golden vectors of the reference itself).
__device__ __forceinline__ double lbp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// W messages in lock step (W = 1, 2): every step is written for all W elements before the next one, so that a wave carries W
// independent dependent chains (the fp64 pipe of a SIMD with 2 waves wants them).  Per element:
//   E = exp(-2|y|) by 2^n * (1 + r q(r)), q a degree-13 Taylor kernel after the usual ln2 split; em1 = E - 1 without cancellation
//   t = tanh(|y|) = -em1 / (1 + E),  x = clip(a sign(y) t, +-sat): both rounded to double, as in the reference
//   atanh(x) = log(P / D) / 2,  P = 1 + x,  D = 1 - x  (D is exact for |x| >= 1/2)
//   log(P / D) = k ln2 + log((1 + s) / (1 - s)),  s = (P 2^-k - D) / (P 2^-k + D) with k chosen from the exponents so that the
//   ratio lies in [sqrt(1/2), sqrt(2)] (|s| <= 0.1716: fdlibm's log kernel 2s + s R(s^2)); where k = 0 the numerator is taken as
//   2x = P - D exactly -- small messages keep their relative accuracy.
// Two divisions per message (round 2: two divisions and a reciprocal).
// WHY x IS ROUNDED.  An algebraically equivalent form without the intermediate x -- P = (1 + a') + E (1 - a'), D = (1 - a') +
// E (1 + a'), one division, 9 instructions fewer -- was built first and is MORE accurate than the reference near |x| -> 1, where
// arctanh amplifies the rounding of x by x / ((1 - x^2) atanh x).  That matters: the reference iterates to a relative change
// below MACHINE EPSILON and ends the lambda continuation at the first lambda that does not get there in max_iterations, i.e.
// where its own rounding noise stops the messages from becoming bit-stable.  On 16 seeds of the C3 shape (N = 10^3, beta =
// lambda_start = 3, chains after 10^3 sweeps) the NumPy restatement converges on 33.3 lambdas on average, the no-intermediate
// form on 36.1 (+2.8: it reaches bit-stable fixed points NumPy does not, and hands marginals of a weaker bias to the cluster
// search), this form on 32.9 (-0.4; per seed both differ from NumPy by ~3 lambdas either way -- the last bit decides).  The
// product keeps the reference's roundings.
template <int W>
__device__ __forceinline__ void lbp_message_w(const double (&a)[W], const double (&y)[W], double (&out)[W], double sat, double usat,
                                              double inv_beta)
{
    const double LOG2E = 1.4426950408889634, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    const double SQRT2 = 1.41421356237309504880;
    double ap[W], x[W], nf[W], r[W], q[W], em1r[W], sc[W], E[W], em1[W];
#define LBP_ALL for (int j = 0; j < W; ++j)
#pragma unroll
    LBP_ALL ap[j] = copysign(a[j], a[j] * y[j]);
#pragma unroll
    LBP_ALL x[j] = fmax(-2.0 * fabs(y[j]), -80.0);
#pragma unroll
    LBP_ALL nf[j] = rint(x[j] * LOG2E);
#pragma unroll
    LBP_ALL r[j] = lbp_fma(-nf[j], LN2_HI, x[j]);
#pragma unroll
    LBP_ALL r[j] = lbp_fma(-nf[j], LN2_LO, r[j]);                          // |r| <= ln2 / 2
#pragma unroll
    LBP_ALL q[j] = 1.0 / 87178291200.0;                                    // (e^r - 1)/r = sum r^i / (i+1)!
#define LBP_HORNER(C) _Pragma("unroll") LBP_ALL q[j] = lbp_fma(q[j], r[j], C);
    LBP_HORNER(1.0 / 6227020800.0) LBP_HORNER(1.0 / 479001600.0) LBP_HORNER(1.0 / 39916800.0) LBP_HORNER(1.0 / 3628800.0)
    LBP_HORNER(1.0 / 362880.0) LBP_HORNER(1.0 / 40320.0) LBP_HORNER(1.0 / 5040.0) LBP_HORNER(1.0 / 720.0) LBP_HORNER(1.0 / 120.0)
    LBP_HORNER(1.0 / 24.0) LBP_HORNER(1.0 / 6.0) LBP_HORNER(0.5) LBP_HORNER(1.0)
#undef LBP_HORNER
#pragma unroll
    LBP_ALL em1r[j] = r[j] * q[j];
#pragma unroll
    LBP_ALL sc[j] = ldexp(1.0, (int)nf[j]);
#pragma unroll
    LBP_ALL E[j] = lbp_fma(sc[j], em1r[j], sc[j]);
#pragma unroll
    LBP_ALL em1[j] = lbp_fma(sc[j], em1r[j], sc[j] - 1.0);                 // E - 1 in (-1, 0]: sc - 1 is exact, one rounding
    double N[W], P[W], D[W], mP[W], mD[W], num[W], den[W], s[W], dk[W];
    int eP[W], eD[W], k[W];
    // t = tanh(|y|) = (1 - E) / (1 + E) and x = a' t as DOUBLES, like the reference's intermediates (see above)
    double t[W], xx[W];
    {
        double dn[W], rc[W], e[W], qq[W], nm[W];
#pragma unroll
        LBP_ALL { dn[j] = 1.0 + E[j]; nm[j] = -em1[j]; }
#pragma unroll
        LBP_ALL rc[j] = __builtin_amdgcn_rcp(dn[j]);
#pragma unroll
        LBP_ALL e[j] = lbp_fma(-dn[j], rc[j], 1.0);
#pragma unroll
        LBP_ALL rc[j] = lbp_fma(rc[j], e[j], rc[j]);
#pragma unroll
        LBP_ALL e[j] = lbp_fma(-dn[j], rc[j], 1.0);
#pragma unroll
        LBP_ALL rc[j] = lbp_fma(rc[j], e[j], rc[j]);
#pragma unroll
        LBP_ALL qq[j] = nm[j] * rc[j];
#pragma unroll
        LBP_ALL t[j] = lbp_fma(lbp_fma(-dn[j], qq[j], nm[j]), rc[j], qq[j]);
    }
#pragma unroll
    LBP_ALL xx[j] = fmin(fmax(ap[j] * t[j], -sat), sat);
#pragma unroll
    LBP_ALL { P[j] = 1.0 + xx[j]; D[j] = 1.0 - xx[j]; N[j] = xx[j] + xx[j]; }
#pragma unroll
    LBP_ALL { mP[j] = frexp(P[j], &eP[j]); mD[j] = frexp(D[j], &eD[j]); }  // mantissas in [1/2, 1)
#pragma unroll
    LBP_ALL {
        const int adj = (mP[j] > mD[j] * SQRT2 ? 1 : 0) - (mP[j] * SQRT2 < mD[j] ? 1 : 0);
        k[j] = eP[j] - eD[j] + adj;
        mP[j] = ldexp(mP[j], -adj);
    }
#pragma unroll
    LBP_ALL N[j] = ldexp(N[j], -eD[j]);
#pragma unroll
    LBP_ALL num[j] = mP[j] - mD[j];
#pragma unroll
    LBP_ALL num[j] = k[j] == 0 ? N[j] : num[j];                            // (both arms are values: a select, not control flow)
#pragma unroll
    LBP_ALL den[j] = mP[j] + mD[j];
    {   // s = num / den: reciprocal + two Newton steps + one correction (operands far from the ends of the exponent range)
        double rc[W], e[W], qq[W];
#pragma unroll
        LBP_ALL rc[j] = __builtin_amdgcn_rcp(den[j]);
#pragma unroll
        LBP_ALL e[j] = lbp_fma(-den[j], rc[j], 1.0);
#pragma unroll
        LBP_ALL rc[j] = lbp_fma(rc[j], e[j], rc[j]);
#pragma unroll
        LBP_ALL e[j] = lbp_fma(-den[j], rc[j], 1.0);
#pragma unroll
        LBP_ALL rc[j] = lbp_fma(rc[j], e[j], rc[j]);
#pragma unroll
        LBP_ALL qq[j] = num[j] * rc[j];
#pragma unroll
        LBP_ALL s[j] = lbp_fma(lbp_fma(-den[j], qq[j], num[j]), rc[j], qq[j]);
    }
    double z2[W], w[W], t1[W], t2[W], lg[W];
#pragma unroll
    LBP_ALL { z2[j] = s[j] * s[j]; dk[j] = (double)k[j]; }
#pragma unroll
    LBP_ALL w[j] = z2[j] * z2[j];
#pragma unroll
    LBP_ALL t1[j] = w[j] * lbp_fma(w[j], lbp_fma(w[j], Lg6, Lg4), Lg2);
#pragma unroll
    LBP_ALL t2[j] = z2[j] * lbp_fma(w[j], lbp_fma(w[j], lbp_fma(w[j], Lg7, Lg5), Lg3), Lg1);
#pragma unroll
    LBP_ALL lg[j] = lbp_fma(dk[j], LN2_HI, lbp_fma(2.0, s[j], lbp_fma(s[j], t2[j] + t1[j], dk[j] * LN2_LO)));
#pragma unroll
    LBP_ALL out[j] = fmin(fmax(0.5 * inv_beta * lg[j], -usat), usat);      // atanh is monotone: clipping its argument to +-sat == clamping it
#undef LBP_ALL
}

__device__ __forceinline__ double lbp_message(double a, double y, double sat, double usat, double inv_beta)
{
    const double aa[1] = {a}, yy[1] = {y};
    double o[1];
    lbp_message_w<1>(aa, yy, o, sat, usat, inv_beta);
    return o[0];
}

__device__ __forceinline__ double lbp_wave_max(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// max over the 16 lanes of a DPP row, in registers (quad swaps, then the two mirrors): every lane ends up with the row maximum
template <int CTRL>
__device__ __forceinline__ double lbp_dpp_max_step(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return fmax(v, __hiloint2double(hi2, lo2));
}
__device__ __forceinline__ double lbp_row_max(double v)
{
    v = lbp_dpp_max_step<0xB1>(v);      // quad_perm [1,0,3,2]
    v = lbp_dpp_max_step<0x4E>(v);      // quad_perm [2,3,0,1]
    v = lbp_dpp_max_step<0x141>(v);     // row_half_mirror
    v = lbp_dpp_max_step<0x140>(v);     // row_mirror
    return v;
}
// four non-negative per-lane quantities -> row q (lanes [16 q, 16 q + 16)) of the wave holds the wave maximum of quantity q:
// two exchange steps that halve the number of quantities a lane carries (3 cross-row shuffles), then the in-row maximum
// (the plain butterfly was 6 shuffles + 6 maxima per quantity: 48 ds_bpermute per iteration)
__device__ __forceinline__ double lbp_wave_max4(double q0, double q1, double q2, double q3, int lane)
{
    const bool up = (lane & 32) != 0, odd = (lane & 16) != 0;
    double k0 = up ? q2 : q0, k1 = up ? q3 : q1;
    const double s0 = up ? q0 : q2, s1 = up ? q1 : q3;
    k0 = fmax(k0, __shfl_xor(s0, 32, 64));
    k1 = fmax(k1, __shfl_xor(s1, 32, 64));
    double kk = odd ? k1 : k0;
    const double ss = odd ? k0 : k1;
    kk = fmax(kk, __shfl_xor(ss, 16, 64));
    return lbp_row_max(kk);
}
// sum of a 32-bit value over the wave (every lane gets it): in-row DPP steps, then the two cross-row exchanges
__device__ __forceinline__ unsigned lbp_wave_sum_u32(unsigned v)
{
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);
    v += (unsigned)__shfl_xor((int)v, 16, 64);
    v += (unsigned)__shfl_xor((int)v, 32, 64);
    return v;
}
__device__ __forceinline__ double lbp_read_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Barrier between the `group` workgroups that share one problem (they run on different CUs, possibly on different
// XCDs whose L2s are not coherent): every thread releases its stores at agent scope, one lane arrives on a monotonic
// counter and polls it with relaxed agent-scope loads, then every thread acquires (invalidates its CU's L1).  The
// launch keeps problems * group <= number of CUs, so all workgroups are resident; the poll is BOUNDED all the same --
// on a timeout the problem is flagged (status 2) and its workgroups leave.  Returns false on timeout.
__device__ __forceinline__ bool lbp_group_barrier(unsigned int *counter, unsigned int target, int tid, int poll_budget)
{
    __shared__ int ok;
    __threadfence();                                            // release (agent): write back this thread's stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int budget = poll_budget > 0 ? poll_budget : 1 << 22;
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
    if (G > 1) dead = !lbp_group_barrier(a.bar + p, ++seq * (unsigned)G, tid, a.poll_budget);
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
                if (!lbp_group_barrier(a.bar + p, ++seq * (unsigned)G, tid, a.poll_budget)) { dead = true; break; }
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

// ---- small instances: messages in LDS, edge constants in registers -----------------------------------------------
// k_lbp walks global-memory arrays (L2-resident) with two dependent loads per message: at n = 10^3 (6 messages per
// thread) an iteration took 17 us against a ~5.4 us fp64 VALU floor (scripts/lbp_c3_probe.py).  When both message buffers
// and the node totals fit in LDS (16 nnz + 8 n bytes) and a thread's share of the edges (MPT) and nodes (<= 2) fits in
// registers, everything an iteration touches is LDS or registers: thread t owns edges t, t + 1024, ... for the whole
// launch (source, reverse position, tanh(beta J), its cavity field h_msgs), the iteration is node totals -> barrier ->
// messages -> ONE barrier (the four convergence maxima: lbp_wave_max4 per wave, one LDS row per quantity, then every wave
// repeats the in-row maximum over the waves' partials, so the decision is uniform without a broadcast).  Same operations in the
// same order as k_lbp: results are bit-identical (tests/test_gpu_lbp.py).
// No per-lane predicates in the loop (a bool per edge or per neighbour term is a 64-bit scalar mask each, hoisted out of the
// iteration loop: they pushed the polynomial coefficients out of the scalar file and were read back lane by lane): edge slots
// past nnz are harmless dummies (source = a node slot that always holds 0, tanh(beta J) = 0, reads from message slots that stay
// 0, writes to a scratch slot: their message is exactly 0 and every maximum they feed is 0), node slots past n likewise, a
// node's neighbour terms past its degree read the zero message slot; diagonal entries only in the HAS_DIAG variant.
// NT threads, MPT edges and NPT nodes per thread (NT MPT = 6144 edge slots, NT NPT = 2048 node slots), ILP message
// computations in lock step per wave.  Instantiated as 8 waves x 12 edges x 2 (250 registers, no scratch).  Per iteration and
// wave at n = 10^3: ~1380 instructions in the message phase (115 per message, 91 of them fp64), ~180 in the node phase (two of
// four node slots populated), ~60 for the maxima: 7.0 us (round 3 started at 8.8: a reciprocal more per message, saturation
// as a compare + select, 165 instructions per node slot instead of 80, 48 ds_bpermute for the maxima instead of 6).  More
// independent streams do not help (3 or 4 messages in lock step: no gain; 16 waves x 6 edges x 1 and 12 waves x 8 x 1 spill).
#define NLMC_LBP_HASH_FROM 10       // iterations of a lambda before the cycle detector starts hashing (most lambdas converge earlier)
// The message phase of one iteration (a macro: it is instantiated with and without the state hash of the cycle detector, and a
// lambda over this much captured state ends up in scratch memory).  Two messages at a time, the LDS reads of the next pair in
// flight meanwhile (left to itself the scheduler interleaves all MPT message computations and spills).  Hash: the low mantissa
// word (a change by a few ulp always shows there) mixed with sign, exponent and high mantissa bits; the multiplier is not affine
// in (thread, k): +1 ulp at two edges and -1 ulp at the two "crossed" ones must not cancel.
#define NLMC_LBP_MESSAGE_PHASE(HASH) \
        { \
            double t_n[ILP], w_n[ILP], u_n[ILP];                                                                               \
_Pragma("unroll")                                                                                                              \
            for (int j = 0; j < ILP; ++j) { t_n[j] = tot[E_SRC(j)]; w_n[j] = wc[tid + j * NT]; u_n[j] = wc[E_REV(j)]; }        \
_Pragma("unroll")                                                                                                              \
            for (int k = 0; k < MPT; k += ILP) {                                                                               \
                double t_i[ILP], w_e[ILP], u_o[ILP], h_new[ILP], u_new[ILP];                                                   \
_Pragma("unroll")                                                                                                              \
                for (int j = 0; j < ILP; ++j) { t_i[j] = t_n[j]; w_e[j] = w_n[j]; u_o[j] = u_n[j]; }                           \
                if (k + ILP < MPT) {                                                                                           \
_Pragma("unroll")                                                                                                              \
                    for (int j = 0; j < ILP; ++j) {                                                                            \
                        const int kn = k + ILP + j < MPT ? k + ILP + j : 0;                                                    \
                        t_n[j] = tot[E_SRC(kn)];                                                                               \
                        w_n[j] = wc[tid + kn * NT];                                                                            \
                        u_n[j] = wc[E_REV(kn)];                                                                                \
                    }                                                                                                          \
                }                                                                                                              \
_Pragma("unroll")                                                                                                              \
                for (int j = 0; j < ILP; ++j) {                                                                                \
                    h_new[j] = t_i[j] - w_e[j];                                                                                \
                    if (HAS_DIAG) h_new[j] = (bits & (1u << (8 + k + j))) ? 0.0 : h_new[j];                                    \
                }                                                                                                              \
                {                                                                                                              \
                    double tj[ILP], yy[ILP];                                                                                   \
_Pragma("unroll")                                                                                                              \
                    for (int j = 0; j < ILP; ++j) { tj[j] = e_tJ[k + j]; yy[j] = a.beta * h_new[j]; }                          \
                    lbp_message_w<ILP>(tj, yy, u_new, a.sat, a.usat, a.inv_beta);                                              \
                }                                                                                                              \
_Pragma("unroll")                                                                                                              \
                for (int j = 0; j < ILP; ++j) {                                                                                \
                    const double h_old = e_hm[k + j];                                                                          \
                    e_hm[k + j] = h_new[j];                                                                                    \
                    wn[E_REV(k + j)] = u_new[j];                                                                               \
                    if (HASH) hs1 += __umul24((unsigned)__double2loint(u_new[j]) ^ __builtin_rotateleft32((unsigned)__double2hiint(u_new[j]), 13), __umul24(c1, 2u * (unsigned)(k + j) + 1u)); \
                    dh_n = fmax(dh_n, fabs(h_new[j] - h_old));                                                                 \
                    dh_d = fmax(dh_d, fabs(h_new[j]) + fabs(h_old));                                                           \
                    du_n = fmax(du_n, fabs(u_new[j] - u_o[j]));                                                                \
                    du_d = fmax(du_d, fabs(u_new[j]) + fabs(u_o[j]));                                                          \
                }                                                                                                              \
                __builtin_amdgcn_sched_barrier(0);                                                                             \
            }                                                                                                                  \
        }

template <int NT, int MPT, int NPT, bool HAS_DIAG, int ILP>
__global__ __launch_bounds__(NT) void k_lbp_lds(LbpArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NW = NT / 64, ESLOTS = MPT * NT;
    static_assert(ESLOTS < 65536 && NPT * NT < 65536, "source / reverse slot are packed into 16 bits each");
    static_assert(NW <= 16, "red rows hold 16 waves");
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n, nnz = a.nnz;
    // message buffers [ESLOTS + 1] (slot ESLOTS: scratch), node totals [NPT NT + 1] (slot NPT NT: always 0), red [4][16]
    double *wc = reinterpret_cast<double *>(lds_raw), *wn = wc + ESLOTS + 1, *tot = wn + ESLOTS + 1, *red = tot + NPT * NT + 1;
    double *mag = a.mag + (size_t)p * n;
    const double *ms = a.m_star + (size_t)p * n;

    // this thread's edges
    unsigned e_sr[MPT];                 // source node slot | reverse message slot << 16
    double e_tJ[MPT], e_hm[MPT];
#ifdef NLMC_LBP_DBG_NOCONFLICT          /* timing experiment (wrong results): every LDS access of the message phase coalesced */
#define E_SRC(k) (int)((tid + (k) * 7) & 1023)
#define E_REV(k) (int)(tid + (k) * NT)
#else
#define E_SRC(k) (int)(e_sr[k] & 0xFFFFu)
#define E_REV(k) (int)(e_sr[k] >> 16)
#endif
    unsigned bits = 0u;                 // bit k: node slot k exists; 4 + k: its dense h_msgs row has non-edge entries; 8 + k: edge k is diagonal
#pragma unroll
    for (int k = 0; k < MPT; ++k) {
        const int e = tid + k * NT;
        const bool on = e < nnz;
        const int ec = on ? e : 0;
        const int si = a.src[ec], ri = a.rev[ec];
        e_sr[k] = on ? ((unsigned)si | ((unsigned)ri << 16)) : ((unsigned)(NPT * NT) | ((unsigned)ESLOTS << 16));
        e_tJ[k] = on ? a.tJ[ec] : 0.0;
        e_hm[k] = 0.0;
        if (HAS_DIAG && on && a.col[ec] == si) bits |= 1u << (8 + k);
        // h_msgs = 0, u_msgs = J * m_star.reshape(1, -1)  (NMC/nmc.py:128-129): u_msgs[j, i] = J[j, i] * m_star[i]
        wc[e] = on ? a.val[ri] * ms[si] : 0.0;
        wn[e] = 0.0;
    }
    // this thread's nodes
    int v_r0[NPT], v_deg[NPT];
    double v_hl[NPT];                   // h_lam of the running lambda (set where the lambda starts)
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int i = tid + k * NT;
        const bool on = i < n;
        const int ic = on ? i : 0;
        v_r0[k] = a.rowptr[ic]; v_deg[k] = on ? a.rowptr[ic + 1] - v_r0[k] : 0;
        int offdiag = 0;
        for (int e = v_r0[k]; e < v_r0[k] + v_deg[k]; ++e) offdiag += a.col[e] != ic;
        if (on) bits |= 1u << k;
        if (on && offdiag < n - 1) bits |= 1u << (4 + k);      // row i of the dense h_msgs has non-edge entries, all equal to total_i
        tot[i] = 0.0;
    }
    if (tid == 0) { wc[ESLOTS] = 0.0; wn[ESLOTS] = 0.0; tot[NPT * NT] = 0.0; }
    __syncthreads();

    // row sum of the current messages of node slot k, sequential in ascending neighbour index.  Terms past the degree read the
    // message slot that always holds 0 (sum + 0 is exact: a sum that starts from +0 is never -0), i.e. one address select per
    // term and no predicated loads; `deg` is made opaque per iteration so that the 8 compares stay in the loop -- hoisted they
    // are 8 lane masks per node slot (64 scalar registers at NPT = 4, spilled to lanes of a vector register and read back with
    // two v_readlane per use: the node phase was 165 instructions per slot).
    auto row_sum = [&](int k) __attribute__((always_inline)) {
        int deg = v_deg[k], r0 = v_r0[k];
        asm volatile("" : "+v"(deg), "+v"(r0));                                    // (r0: or 8 hoisted addresses per slot)
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = wc[q < deg ? r0 + q : ESLOTS];             // all reads in flight
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += v[q];
        for (int q = 8; q < deg; ++q) sum += wc[r0 + q];
        return sum;
    };

    // Exact limit cycles.  The reference iterates until both relative changes fall below machine epsilon or max_iterations is
    // reached; the lambda where that fails is, in practice, one where the messages settle into a cycle of period 2 or 4 in
    // their last bits (every exhausted lambda of 16 C3 seeds did, by iteration 9-39 of 100).  The map u(t) -> u(t+1) is a
    // deterministic function of the message array alone, so once u(t) = u(t-p), u(t-1) = u(t-1-p) and u(t-2) = u(t-2-p) hold
    // exactly, every later convergence test (a function of u(s), u(s-1), u(s-2)) repeats one that has already failed: the
    // remaining iterations cannot change the outcome and are skipped -- same marginals, same lambda count, iteration count
    // reported as max_iterations - 1.  States are compared through a 32-bit sum over all messages of (24 bits mixed from both words x a
    // per-edge odd multiplier) (two v_mul_u32_u24 per message); three consecutive matches at the same period are required, so a chance
    // collision (2^-32 each) cannot end a lambda that would have converged.  Ring of the last 64 hashes: lane j of every wave holds
    // the hash of the latest iteration == j (mod 64); periods up to 63 are seen.  On 64 C3 seeds: median iterations per seed
    // 309 -> 229; the slowest seed of a batch rarely gains (one seed in 64 finds no short cycle), batches that fill the chip do.
    static_assert(NW < 16, "red[15] / red[31] hold the hash sum of the running / the next iteration");
    unsigned *const hsum = reinterpret_cast<unsigned *>(red);          // words 30, 31 (red[15]) and 62, 63 (red[31]): two slots
    const unsigned hc1 = (((unsigned)tid * 0x9E3779B1u) >> 8) | 1u;       // 24 bits, odd
    unsigned hr1 = 0u;
    int hpar = 0;                                                     // slot of the running iteration (alternates across lambdas too)
    if (tid == 0) { hsum[30] = 0u; hsum[62] = 0u; }
    int n_done = 0, status = 0;
    for (int l = 0; l < a.n_lams; ++l) {
        const double lam = a.lams[l];
        // h_lam = h + lam * (m_star * epsilon)  (NMC/nmc.py:137), slots past n: 0
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int i = tid + k * NT;
            int ic = i < n ? i : 0;
            asm volatile("" : "+v"(ic));             // (re-read per lambda: hoisted, the three inputs take 6 registers per slot)
            const double me = ms[ic] * a.eps[ic];
            v_hl[k] = i < n ? a.h[ic] + lam * me : 0.0;
        }
        int it = 0, cyc_run = 0, cyc_p = 0;
        for (int iter = 0; iter < a.max_iter; ++iter) {
            it = iter;
            double dh_n = 0.0, dh_d = 0.0, du_n = 0.0, du_d = 0.0;
            unsigned hs1 = 0u, c1 = hc1;
            asm volatile("" : "+v"(c1));                // (keeps the 12 per-edge multipliers out of registers: one multiply each, in the loop)
            // ---- node totals: total_i = h_lam[i] + sum_k u_msgs[k, i]   (NMC/nmc.py:199-201)
#pragma unroll
            for (int k = 0; k < NPT; ++k) {
                if (__ballot((bits >> k) & 1u) == 0ull) continue;      // no lane of this wave owns a node in slot k (n < NPT NT)
                const int i = tid + k * NT;
                const double hl = v_hl[k];
                const double t_new = hl + row_sum(k), t_old = tot[i];
                tot[i] = t_new;                                         // (slots past n: nobody reads them; their value is 0 anyway)
                if (bits & (1u << (4 + k))) {
                    dh_n = fmax(dh_n, fabs(t_new - t_old));
                    dh_d = fmax(dh_d, fabs(t_new) + fabs(t_old));
                }
            }
            __syncthreads();
            // ---- messages: h_msgs[i, j] = total_i - u_msgs[j, i];  u_msgs[i, j] = atanh_sat(tanh(bJ) tanh(b h_msgs)) / b
            // (two messages at a time, the LDS reads of the next pair in flight meanwhile: left to itself the scheduler
            // interleaves all MPT message computations and spills)
            static_assert(MPT % ILP == 0, "messages are processed ILP at a time");
            const bool hashing = iter >= NLMC_LBP_HASH_FROM;             // (uniform)
            if (hashing) NLMC_LBP_MESSAGE_PHASE(true) else NLMC_LBP_MESSAGE_PHASE(false)
            const double wm = lbp_wave_max4(dh_n, dh_d, du_n, du_d, lane);      // row q of the wave: quantity q
            if ((lane & 15) == 0) red[(lane >> 4) * 16 + wv] = wm;
            const int hslot = 30 + 32 * hpar;
            if (hashing) {
                const unsigned w1 = lbp_wave_sum_u32(hs1);
                if (lane == 0) atomicAdd(&hsum[hslot], w1);
                hpar ^= 1;
            }
            __syncthreads();
            bool cycling = false;
            if (hashing) {
                const unsigned h1 = hsum[hslot];
                if (tid == 0) hsum[hslot ^ 32] = 0u;                                      // the other slot: the next hashed iteration's sum
                // lane j holds the hash of the latest iteration == j (mod 64) (valid if that iteration was hashed in this lambda);
                // the nearest earlier equal state is p iterations back: rotate the match mask so that the entry of iteration
                // iter - q sits at bit 63 - q, count leading zeros
                const int r = iter & 63;
                const int latest = iter - 1 - ((iter - 1 - lane) & 63);
                unsigned long long m64 = __ballot(latest >= NLMC_LBP_HASH_FROM && hr1 == h1) & ~(1ull << r);
                const int sh = 63 - r;
                m64 = sh ? ((m64 << sh) | (m64 >> (64 - sh))) : m64;
                const int pm = m64 ? __clzll((long long)m64) : 0;
                cyc_run = (pm != 0 && pm == cyc_p) ? cyc_run + 1 : (pm != 0 ? 1 : 0);
                cyc_p = pm;
                if (lane == r) hr1 = h1;
                cycling = cyc_run >= 3;
            }
            // every wave reduces the 4 x NW partial maxima itself: lanes [16 q, 16 q + 16) hold quantity q
            const double m = lbp_row_max((lane & 15) < NW ? red[lane] : 0.0);
            const double r0 = lbp_read_lane(m, 0), r1 = lbp_read_lane(m, 16), r2 = lbp_read_lane(m, 32), r3 = lbp_read_lane(m, 48);
            { double *t = wc; wc = wn; wn = t; }
            // 0/0 = NaN compares false, like the reference's `du < tolerance and dh < tolerance` (NMC/nmc.py:212-213)
            const double dh = r0 / r1, du = r2 / r3;
            if (du < a.tol && dh < a.tol) break;
            if (cycling) { it = a.max_iter - 1; break; }          // an exact cycle: no later iteration can pass the test
        }
        if (tid == 0) a.out_iters[(size_t)p * a.n_lams + l] = it;
        const bool exhausted = (it == a.max_iter - 1);
        if (exhausted && l == 0) { status = 1; break; }          // NMC/nmc.py:142-144
        if (!exhausted) {
            // magnetizations = tanh(beta (h_lam + sum_k u_msgs[k, :]))  (NMC/nmc.py:216-217), rows added in ascending k
#pragma unroll
            for (int k = 0; k < NPT; ++k)
                if (bits & (1u << k)) mag[tid + k * NT] = tanh(a.beta * (v_hl[k] + row_sum(k)));
        }
        // exhausted at a later lambda: keep the previous marginals and stop (NMC/nmc.py:145-148)
        if (a.mag_all) {
            double *dst = a.mag_all + ((size_t)p * a.n_lams + l) * n;
            __syncthreads();
            for (int i = tid; i < n; i += NT) dst[i] = mag[i];
        }
        n_done = l + 1;
        if (exhausted) break;
        __syncthreads();               // red / the message buffers are rewritten by the next lambda's first iteration
    }
    if (tid == 0) { a.out_nlam[p] = n_done; a.out_status[p] = status; }
#undef E_SRC
#undef E_REV
}
