/*
 * oracle/nlo.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, scalar, one thread) of the
 * reference hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it;
 * the product (nonlocal-monte-carlo_amd/) never imports, links or executes anything under oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against golden vectors
 * captured by running the reference itself in the build container (tools/make_golden.py -> tests/golden/).
 *
 * Reference lines restated here (paths relative to the reference checkout):
 *   nlo_sweeps_stream   NMC/nmc.py:28-91  (== NPT/npt.py:47-110, NPT/apt_ICM.py:52-93,
 *                       NPT/apt_preprocessor.py:33-74): random-permutation sequential heat-bath sweeps.
 *                       The field uses scipy's csr_matvec order (third-party, scipy==1.11.2 pinned in
 *                       requirements.txt:14; algorithm: y_i = 0; for jj in row i ascending: y_i += A_jj*x_col),
 *                       then `+ h`  (NMC/nmc.py:86); update rule NMC/nmc.py:87.
 *   nlo_energy          NMC/nmc.py:386 / NPT/npt.py:31-45:  E = -(m^T J m / 2 + m^T h).
 *   nlo_clusters        NPT/apt_ICM.py:116-143: connected components of the disagreement sub-graph,
 *                       ordered by ascending smallest member.
 *   nlo_sweeps_philox   NOT a reference function: the sequential *specification* of the product's throughput
 *                       mode (Philox4x32-10 order keys + uniforms; "f32": 24-bit fixed-point couplings, exact int32
 *                       field, logistic threshold drawn from 32 random bits; "f64": fp64 field, base-2 logistic test
 *                       on a 53-bit uniform).  The HIP kernels run a level-parallel schedule that must reproduce it
 *                       bit for bit.  Its LAW is tied to the reference by tests/test_law_cpu.py (exhaustive check of
 *                       the threshold distribution) and by the reference-derived statistics under tests/golden/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------ */
/* a-1  heat-bath sweeps consuming an externally drawn legacy stream (perm + uniforms per sweep)      */
/* ------------------------------------------------------------------------------------------------ */
/* perm[t*n + i]  : i-th spin visited in sweep t            (np.random.permutation(N), NMC/nmc.py:71)
 * u[t*n + i]     : the i-th np.random.rand() of sweep t     (NMC/nmc.py:87)
 * beta_run[t]    : inverse temperature of sweep t           (NMC/nmc.py:56-69, built by the caller)
 * m              : in: start state (+-1, may hold 0), out: final state, as double like the reference
 * M_out[t*n + k] : state after sweep t (int8), nullable
 */
int nlo_sweeps_stream(int n, const int32_t *rowptr, const int32_t *col, const double *val, const double *h,
                      int num_sweeps, const int32_t *perm, const double *u, const double *beta_run,
                      double *m, int8_t *M_out)
{
    for (int t = 0; t < num_sweeps; ++t) {
        const int32_t *p = perm + (size_t)t * n;
        const double *ut = u + (size_t)t * n;
        const double b = beta_run[t];
        for (int i = 0; i < n; ++i) {
            const int k = p[i];
            double x = 0.0;
            for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) x += val[e] * m[col[e]];
            x = x + h[k];
            const double v = tanh(b * x) - 2.0 * ut[i] + 1.0; /* left-to-right like the Python expression */
            m[k] = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0); /* np.sign */
        }
        if (M_out) {
            int8_t *o = M_out + (size_t)t * n;
            for (int k = 0; k < n; ++k) o[k] = (int8_t)m[k];
        }
    }
    return 0;
}

/* a-3  E = -(m^T J m / 2 + m^T h), J in CSR with both triangles stored */
double nlo_energy(int n, const int32_t *rowptr, const int32_t *col, const double *val, const double *h,
                  const int8_t *m)
{
    double q = 0.0, l = 0.0;
    for (int k = 0; k < n; ++k) {
        double x = 0.0;
        for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) x += val[e] * (double)m[col[e]];
        q += (double)m[k] * x;
        l += (double)m[k] * h[k];
    }
    return -(q / 2.0 + l);
}

/* ------------------------------------------------------------------------------------------------ */
/* a-9  disagreement clusters                                                                        */
/* ------------------------------------------------------------------------------------------------ */
/* label_out[k] = index (0-based, in the reference's list order) of the cluster holding k, or -1.
 * returns the number of clusters. */
int nlo_clusters(int n, const int32_t *rowptr, const int32_t *col, const double *val, const int8_t *s1,
                 const int8_t *s2, int32_t *label_out)
{
    int32_t *queue = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int nc = 0;
    for (int k = 0; k < n; ++k) label_out[k] = -1;
    for (int s = 0; s < n; ++s) {
        if ((int)s1[s] * (int)s2[s] != -1 || label_out[s] >= 0) continue;
        int head = 0, tail = 0;
        queue[tail++] = s;
        label_out[s] = nc;
        while (head < tail) {
            const int c = queue[head++];
            for (int e = rowptr[c]; e < rowptr[c + 1]; ++e) {
                const int j = col[e];
                if (val[e] == 0.0) continue; /* `val != 0` test, NPT/apt_ICM.py:129 */
                if ((int)s1[j] * (int)s2[j] != -1 || label_out[j] >= 0) continue;
                label_out[j] = nc;
                queue[tail++] = j;
            }
        }
        ++nc;
    }
    free(queue);
    return nc;
}

/* ------------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) -- public algorithm, restated                                */
/* ------------------------------------------------------------------------------------------------ */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void nlo_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

/* Stream tags (counter word 3).  Must match nonlocal-monte-carlo_amd/csrc/nlmc_device.h */
#define NLMC_TAG_UNIFORM 1u
#define NLMC_TAG_UNIFORM_LO 7u
#define NLMC_TAG_ORDER 2u
#define NLMC_TAG_SWAP 3u
#define NLMC_TAG_PAIR 4u
#define NLMC_TAG_ICM 5u

/* 2^z for |z| <= 100 from IEEE basic operations only (identical bits on CPU and GPU). */
static inline float exp2_spec_f32(float z)
{
    if (z > 100.0f) z = 100.0f;
    if (z < -100.0f) z = -100.0f;
    const float nf = rintf(z);
    const float f = z - nf; /* exact, |f| <= 0.5 */
    float p = 1.540352969e-04f;     /* ln2^6/6! ; degree-6 Taylor of 2^f, rel. err ~1e-7 */
    p = fmaf(p, f, 1.333355787e-03f);
    p = fmaf(p, f, 9.618128650e-03f);
    p = fmaf(p, f, 5.550410971e-02f);
    p = fmaf(p, f, 2.402265072e-01f);
    p = fmaf(p, f, 6.931471825e-01f);
    p = fmaf(p, f, 1.0f);
    union { float f; uint32_t u; } v;
    v.f = p;
    v.u += ((uint32_t)(int32_t)nf) << 23;
    return v.f;
}

static inline double exp2_spec_f64(double z)
{
    if (z > 1000.0) z = 1000.0;
    if (z < -1000.0) z = -1000.0;
    const double nf = rint(z);
    const double f = z - nf; /* exact, |f| <= 0.5 */
    double p = 1.36914888539041241e-12;     /* ln2^13/13! ; degree-13 Taylor, rel. err < 1e-16 */
    p = fma(p, f, 2.56784359934881958e-11);
    p = fma(p, f, 4.44553827187081007e-10);
    p = fma(p, f, 7.05491162080112088e-09);
    p = fma(p, f, 1.01780860092396960e-07);
    p = fma(p, f, 1.32154867901443053e-06);
    p = fma(p, f, 1.52527338040598377e-05);
    p = fma(p, f, 1.54035303933816061e-04);
    p = fma(p, f, 1.33335581464284411e-03);
    p = fma(p, f, 9.61812910762847688e-03);
    p = fma(p, f, 5.55041086648215762e-02);
    p = fma(p, f, 2.40226506959100694e-01);
    p = fma(p, f, 6.93147180559945286e-01);
    p = fma(p, f, 1.0);
    union { double f; uint64_t u; } v;
    v.f = p;
    v.u += ((uint64_t)(int64_t)nf) << 52;
    return v.f;
}

float nlo_exp2_f32(float z) { return exp2_spec_f32(z); }
double nlo_exp2_f64(double z) { return exp2_spec_f64(z); }

/* ------------------------------------------------------------------------------------------------ */
/* "f32" throughput mode: fixed-point couplings and the logistic threshold                            */
/* ------------------------------------------------------------------------------------------------ */
/* Field scale qs and energy scale escale of an instance (restated from nlmc_create, csrc/nlmc.hip):
 *   Jq_e = rint(J_e 2^qs), hq_k = rint(h_k 2^qs) with qs the largest exponent such that every |Jq_e| <= 2^23 - 1
 *   (signed 24-bit multiplier) and every row sum  sum_e |Jq_e| + |hq_k| <= 2^31 - 1  (the field is an exact int32),
 *   then lowered by the number of trailing zero bits common to every Jq and hq (canonical form; no value changes);
 *   escale0 = clamp(60 - ceil_log2(sum|J|/2 + sum|h|), 0, 52);  qs <= escale0;  escale = min(escale0, qs + 29)
 *   (an energy delta is the int32 field times +-2^(escale - qs + 1), one 32 x 32 -> 64 bit multiply-add). */
static int64_t rint_scaled(double v, int qs) { return (int64_t)llrint(ldexp(v, qs)); }

int nlo_field_scale(int n, const int32_t *rowptr, const double *val, const double *h, int *qs_out, int *escale_out)
{
    const int64_t nnz = rowptr[n];
    double maxabs = 0.0, bound = 0.0;
    for (int64_t e = 0; e < nnz; ++e) { maxabs = fmax(maxabs, fabs(val[e])); bound += fabs(val[e]) * 0.5; }
    double maxh = 0.0;
    for (int k = 0; k < n; ++k) { maxh = fmax(maxh, fabs(h[k])); bound += fabs(h[k]); }
    int exb = 0;
    frexp(fmax(bound, 1.0), &exb);
    int escale0 = 60 - exb;
    if (escale0 < 0) escale0 = 0;
    if (escale0 > 52) escale0 = 52;
    int qs = 0;
    const double ref = maxabs > 0.0 ? maxabs : maxh;
    if (ref > 0.0) {
        int ex = 0;
        frexp(ref, &ex);
        qs = 23 - ex;
        if (qs > escale0) qs = escale0;
        for (;;) {
            int ok = 1;
            for (int k = 0; k < n && ok; ++k) {
                int64_t row = llabs(rint_scaled(h[k], qs));
                for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) {
                    const int64_t q = llabs(rint_scaled(val[e], qs));
                    if (q > 8388607) ok = 0;
                    row += q;
                }
                if (row > 2147483647LL) ok = 0;
            }
            if (ok) break;
            --qs;
        }
        /* canonical form: drop the power of two common to every Jq and hq (+-J instances: Jq = +-1, qs = 0) */
        for (;;) {
            int even = 1, any = 0;
            for (int64_t e = 0; e < nnz && even; ++e) { const int64_t q = rint_scaled(val[e], qs); if (q & 1) even = 0; if (q) any = 1; }
            for (int k = 0; k < n && even; ++k) { const int64_t q = rint_scaled(h[k], qs); if (q & 1) even = 0; if (q) any = 1; }
            if (!even || !any) break;
            --qs;
        }
    }
    int escale = escale0 < qs + 29 ? escale0 : qs + 29;
    if (qs_out) *qs_out = qs;
    if (escale_out) *escale_out = escale;
    return 0;
}

/* log2(1.5 + t) on |t| <= 0.5: degree-7 minimax fit, fmaf Horner (max error 3.8e-7) */
static inline float log2_15_spec(float t)
{
    float p = 0x1.e444e6p-7f;
    p = fmaf(p, t, -0x1.9b9e5ap-6f);
    p = fmaf(p, t, 0x1.32e57ap-5f);
    p = fmaf(p, t, -0x1.2122a4p-4f);
    p = fmaf(p, t, 0x1.23e4dep-3f);
    p = fmaf(p, t, -0x1.4853d8p-2f);
    p = fmaf(p, t, 0x1.ec7086p-1f);
    p = fmaf(p, t, 0x1.2b803ep-1f);
    return p;
}

/* Logistic threshold of one update from 32 random bits r:  W(r) ~= log2((1 - u) / u),  u = (r + 1/2) / 2^32.
 * The heat-bath rule  s' = +1  iff  u < 1 / (1 + exp(-2 beta x))  (NMC/nmc.py:87 with the centred uniform) reads
 * s' = +1  iff  z < W(r),  z = -2 log2(e) beta x.  Built from v = min(u, 1 - u) so that W(~r) == -W(r) exactly
 * (the law is symmetric under a global spin flip); IEEE single operations only, identical bits on CPU and GPU. */
static inline float threshold_spec_f32(uint32_t r)
{
    const uint32_t m = (uint32_t)((int32_t)r >> 31);      /* all ones iff u > 1/2 */
    const uint32_t a = r ^ m;                             /* v = (a + 1/2) / 2^32 in (0, 1/2) */
    const float v = fmaf((float)a, 0x1p-32f, 0x1p-33f);
    union { float f; uint32_t u; } b;
    b.f = v;
    const float ef = (float)((int32_t)(b.u >> 23) - 126); /* v = mant 2^(ef - 1), mant in [1, 2) */
    b.u = (b.u & 0x7FFFFFu) | 0x3F800000u;
    const float t1 = b.f - 1.5f;                          /* log2 v       = (ef - 1) + log2(1.5 + t1) */
    const float t2 = fmaf(v, -2.0f, 0.5f);                /* log2 (1 - v) = -1 + log2(1.5 + t2)       */
    const float w = (log2_15_spec(t2) - log2_15_spec(t1)) - ef;
    union { float f; uint32_t u; } o;
    o.f = w;
    o.u = (o.u & 0x7FFFFFFFu) | (r & 0x80000000u);        /* |w| with the sign of (u - 1/2) */
    return o.f;
}

float nlo_threshold_f32(uint32_t r) { return threshold_spec_f32(r); }

/* number of r in [r0, r1) on the lattice r0 + i*stride with z < W(r)  (tests/test_law_cpu.py) */
uint64_t nlo_threshold_count(float z, uint64_t r0, uint64_t r1, uint32_t stride)
{
    uint64_t cnt = 0;
    for (uint64_t r = r0; r < r1; r += stride) cnt += (z < threshold_spec_f32((uint32_t)r));
    return cnt;
}

typedef struct { uint32_t key; int32_t idx; } keyed_t;
static int keyed_cmp(const void *a, const void *b)
{
    const keyed_t *x = (const keyed_t *)a, *y = (const keyed_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

/* Sequential specification of the product's throughput mode, one chain.
 *   order of sweep t   : spins sorted by (philox(k, t, order_group, ORDER)[0], k)
 *   flags[k]           : 0 normal, 1 scaled (cluster spin at beta/temp_x), 2/3 frozen (never updated)
 *   "f32" (use_f64 = 0): couplings in 24-bit fixed point (nlo_field_scale), field X = hq_k + sum_e Jq_e s_c(e) exact
 *                        in int32 (any summation order);  z = cbq * (float)X,  cbq = (float)cb * 2^-qs;
 *                        s' = +1 iff z < W(r),  r = word (k&3) of philox(k>>2, t, chain_id, UNIFORM) (threshold_spec_f32);
 *                        energy: E_fix += (s - s') * X_offdiag * 2^(escale - qs) on every flip (exact).
 *   "f64" (use_f64 = 1): field x = ((0 + J_e0 s_c0) + J_e1 s_c1 ...) + h_k in double;  u = 53 bits from words
 *                        (2(k&1), 2(k&1)+1) of philox(k>>1, t, chain_id, UNIFORM);  z = cb * x;
 *                        s' = (fma(u, exp2_spec(z), u) < 1) ? +1 : -1;  E_fix += llrint(-(s'-s) * x * 2^escale).
 *   cb = (T)(-2 log2(e) beta), beta = flags==1 ? beta_scaled : beta  (rounded by the caller)
 * Returns -1 when escale is not in [qs, qs + 29] in f32 mode.
 */
int nlo_sweeps_philox(int n, const int32_t *rowptr, const int32_t *col, const double *val, const double *h,
                      int use_f64, int num_sweeps, uint32_t sweep0, const double *cb_run /*[num_sweeps][2]*/,
                      uint32_t seed_lo, uint32_t seed_hi, uint32_t chain_id, uint32_t order_group,
                      const uint8_t *flags /*nullable*/, int escale, int8_t *s, int64_t *efix_io,
                      int8_t *M_out /*nullable [num_sweeps][n]*/, int64_t *efix_trace /*nullable [num_sweeps]*/)
{
    keyed_t *ord = (keyed_t *)malloc(sizeof(keyed_t) * (size_t)(n > 0 ? n : 1));
    int32_t *valq = NULL, *hq = NULL;
    const int64_t nnz = rowptr[n];
    int qs = 0;
    if (!use_f64) {
        nlo_field_scale(n, rowptr, val, h, &qs, NULL);
        if (escale < qs || escale > qs + 29) { free(ord); return -1; }
        valq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
        hq = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        for (int64_t e = 0; e < nnz; ++e) valq[e] = (int32_t)rint_scaled(val[e], qs);
        for (int k = 0; k < n; ++k) hq[k] = (int32_t)rint_scaled(h[k], qs);
    }
    int64_t efix = efix_io ? *efix_io : 0;
    const double esc = ldexp(1.0, escale);
    for (int t = 0; t < num_sweeps; ++t) {
        const uint32_t tt = sweep0 + (uint32_t)t;
        uint32_t r[4];
        for (int k = 0; k < n; ++k) {
            philox4x32_10((uint32_t)k, tt, order_group, NLMC_TAG_ORDER, seed_lo, seed_hi, r);
            ord[k].key = r[0];
            ord[k].idx = k;
        }
        qsort(ord, (size_t)n, sizeof(keyed_t), keyed_cmp);
        for (int i = 0; i < n; ++i) {
            const int k = ord[i].idx;
            const unsigned fl = flags ? flags[k] : 0u;
            if (fl >= 2u) continue;
            /* one Philox call serves 4 consecutive spins; the fp64 mode takes the 26 low bits of its 53-bit uniform from the same
             * word of a second call (tag UNIFORM_LO) */
            uint32_t r2[4] = {0u, 0u, 0u, 0u};
            philox4x32_10((uint32_t)(k >> 2), tt, chain_id, NLMC_TAG_UNIFORM, seed_lo, seed_hi, r);
            if (use_f64) philox4x32_10((uint32_t)(k >> 2), tt, chain_id, NLMC_TAG_UNIFORM_LO, seed_lo, seed_hi, r2);
            int accept;
            double xd = 0.0;
            int32_t Xq = 0;
            if (use_f64) {
                double x = 0.0, xdg = 0.0; /* xdg: diagonal term, excluded from the energy delta */
                for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) {
                    const double tm = val[e] * (double)s[col[e]];
                    x += tm;
                    if (col[e] == k) xdg += tm;
                }
                xd = (x - xdg) + h[k];
                x = x + h[k];
                const double u = ((double)(r[k & 3] >> 5) * 67108864.0 + (double)(r2[k & 3] >> 6)) / 9007199254740992.0;
                const double z = cb_run[2 * t + (fl == 1u)] * x;
                const double ee = exp2_spec_f64(z);
                accept = fma(u, ee, u) < 1.0;
            } else {
                int32_t X = hq[k], Xdg = 0;
                for (int e = rowptr[k]; e < rowptr[k + 1]; ++e) {
                    const int32_t tm = valq[e] * (int32_t)s[col[e]];
                    X += tm;
                    if (col[e] == k) Xdg += tm;
                }
                Xq = X - Xdg;
                const float cbq = ldexpf((float)cb_run[2 * t + (fl == 1u)], -qs);
                const float z = cbq * (float)X;
                accept = z < threshold_spec_f32(r[k & 3]);
            }
            const int8_t sn = accept ? 1 : -1;
            if (sn != s[k]) {
                if (use_f64) efix += llrint(-(double)(sn - s[k]) * xd * esc);
                else efix += (int64_t)(s[k] - sn) * (int64_t)Xq * ((int64_t)1 << (escale - qs));
                s[k] = sn;
            }
        }
        if (M_out) memcpy(M_out + (size_t)t * n, s, (size_t)n);
        if (efix_trace) efix_trace[t] = efix;
    }
    if (efix_io) *efix_io = efix;
    free(ord);
    free(valq);
    free(hq);
    return 0;
}
