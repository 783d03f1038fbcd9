// nlmc_device.h -- device-side building blocks for gfx950 (wave64): Philox4x32-10, the base-2 logistic test.
// The arithmetic here is the product's throughput-mode SPEC; oracle/nlo.c restates it independently in C and
// the -m gpu tests require bit-identical spins between the two.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NLMC_TAG_UNIFORM 1u
#define NLMC_TAG_ORDER 2u
#define NLMC_TAG_SWAP 3u
#define NLMC_TAG_PAIR 4u
#define NLMC_TAG_ICM 5u
#define NLMC_TAG_UNIFORM_LO 7u        // (6: the Houdayer pairing keys, csrc/nlmc_pt_icm.h)

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 bit multiply (v_mad_u64_u32) yields both halves
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// rounds [r0, r0 + nr) of philox4x32-10 on a state (the key schedule is a function of the round index): lets a caller
// cut one call into pieces
__device__ __forceinline__ u32x4 philox4x32_rounds(u32x4 c, uint32_t k0, uint32_t k1, int r0, int nr)
{
    k0 += (uint32_t)r0 * 0x9E3779B9u;
    k1 += (uint32_t)r0 * 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < nr; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        c = u32x4{n0, (uint32_t)p1, n2, (uint32_t)p0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// two independent calls side by side: their dependent multiply chains interleave in one wave's instruction stream
// (a lone producing wave gets no latency hiding from its SIMD neighbours, which sit in barriers most of the time)
__device__ __forceinline__ void philox4x32_10_x2(uint32_t a0, uint32_t b0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1, u32x4 &ra, u32x4 &rb)
{
    uint32_t x0 = a0, x1 = c1, x2 = c2, x3 = c3, y0 = b0, y1 = c1, y2 = c2, y3 = c3;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
        const unsigned long long q0 = (unsigned long long)0xD2511F53u * y0, q1 = (unsigned long long)0xCD9E8D57u * y2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1;
        const uint32_t m0 = (uint32_t)(q1 >> 32) ^ y1 ^ k0, m2 = (uint32_t)(q0 >> 32) ^ y3 ^ k1;
        x0 = n0; x1 = (uint32_t)p1; x2 = n2; x3 = (uint32_t)p0;
        y0 = m0; y1 = (uint32_t)q1; y2 = m2; y3 = (uint32_t)q0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    ra = u32x4{x0, x1, x2, x3};
    rb = u32x4{y0, y1, y2, y3};
}

// 2^z from IEEE basic operations only (clamp, rint, fma chain, exponent add): same bits on host and device.
__device__ __forceinline__ float exp2_spec(float z)
{
    z = fminf(fmaxf(z, -100.0f), 100.0f);
    const float nf = rintf(z);
    const float f = z - nf;
    float p = 1.540352969e-04f;
    p = __fmaf_rn(p, f, 1.333355787e-03f);
    p = __fmaf_rn(p, f, 9.618128650e-03f);
    p = __fmaf_rn(p, f, 5.550410971e-02f);
    p = __fmaf_rn(p, f, 2.402265072e-01f);
    p = __fmaf_rn(p, f, 6.931471825e-01f);
    p = __fmaf_rn(p, f, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (((uint32_t)(int32_t)nf) << 23));
}

__device__ __forceinline__ double exp2_spec(double z)
{
    z = fmin(fmax(z, -1000.0), 1000.0);
    const double nf = rint(z);
    const double f = z - nf;
    double p = 1.36914888539041241e-12;
    p = __fma_rn(p, f, 2.56784359934881958e-11);
    p = __fma_rn(p, f, 4.44553827187081007e-10);
    p = __fma_rn(p, f, 7.05491162080112088e-09);
    p = __fma_rn(p, f, 1.01780860092396960e-07);
    p = __fma_rn(p, f, 1.32154867901443053e-06);
    p = __fma_rn(p, f, 1.52527338040598377e-05);
    p = __fma_rn(p, f, 1.54035303933816061e-04);
    p = __fma_rn(p, f, 1.33335581464284411e-03);
    p = __fma_rn(p, f, 9.61812910762847688e-03);
    p = __fma_rn(p, f, 5.55041086648215762e-02);
    p = __fma_rn(p, f, 2.40226506959100694e-01);
    p = __fma_rn(p, f, 6.93147180559945286e-01);
    p = __fma_rn(p, f, 1.0);
    return __longlong_as_double(__double_as_longlong(p) + (long long)(((uint64_t)(int64_t)nf) << 52));
}

__device__ __forceinline__ double uniform_from(const u32x4 &r, double)
{
    return ((double)(r.x >> 5) * 67108864.0 + (double)(r.y >> 6)) / 9007199254740992.0;
}

// heat-bath acceptance of s=+1 in the fp64 mode:  u < 1/(1+2^z)  <=>  fma(u, 2^z, u) < 1,   z = -2 log2(e) beta x
__device__ __forceinline__ bool accept_up(double u, double z) { return __fma_rn(u, exp2_spec(z), u) < 1.0; }

// The 53-bit uniform of spin k in the fp64 mode (round 4): the 27 high bits come from word k & 3 of the call (k >> 2, sweep, chain,
// UNIFORM) -- the call the "f32" mode takes its 32 random bits from --, the 26 low bits from the same word of the call with the tag
// UNIFORM_LO.  High and low half are separate calls because the fused-window kernel decides an update from the high bits alone
// (2^-27 of the updates excepted): it then makes ONE call per four spins instead of one per two.
__device__ __forceinline__ double uniform53_spec(uint32_t hi_word, uint32_t lo_word)
{
    return ((double)(hi_word >> 5) * 67108864.0 + (double)(lo_word >> 6)) / 9007199254740992.0;
}

// The same test as an integer threshold (fused windows of the fp64 mode).  u = k 2^-53 with the 53-bit integer
// k = (hi >> 5) << 26 | (lo >> 6) of uniform53_spec; for a fixed E = 2^z > 0 the correctly rounded fma(u, E, u) is monotone
// non-decreasing in u, so  { k : fma(k 2^-53, E, k 2^-53) < 1 }  is an initial segment [0, K) and accept_up(u, z) <=> k < K.
// K in [1, 2^53] is found by bisection on the test itself (bracketed by an estimate first: a handful of steps instead
// of 53) -- no second arithmetic, the same bits as accept_up by construction.
__device__ __forceinline__ unsigned long long accept_count_spec(double z)
{
    const double E = exp2_spec(z);
    const unsigned long long top = 1ull << 53;
    auto acc = [&](unsigned long long k) { const double u = (double)k * 0x1p-53; return __fma_rn(u, E, u) < 1.0; };
    unsigned long long lo = 0ull, hi = top;                  // acc(lo) holds (fma(0, E, 0) = 0 < 1); hi: first k known to fail, or 2^53
    const double est_d = 0x1p53 / (1.0 + E);
    const unsigned long long est = est_d >= 0x1p53 ? top : (unsigned long long)est_d;
    const unsigned long long a = est > 8ull ? est - 8ull : 0ull, b = est + 8ull;
    if (acc(a)) lo = a;
    if (b < top && !acc(b)) hi = b;
    for (int it = 0; it < 54 && hi - lo > 1ull; ++it) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        if (acc(mid)) lo = mid; else hi = mid;
    }
    return hi;
}

// ---- "f32" throughput mode: logistic threshold from 32 random bits --------------------------------------------
// log2(1.5 + t) on |t| <= 0.5: degree-7 minimax fit, fma Horner (max error 3.8e-7)
__device__ __forceinline__ float log2_15_spec(float t)
{
    float p = 0x1.e444e6p-7f;
    p = __fmaf_rn(p, t, -0x1.9b9e5ap-6f);
    p = __fmaf_rn(p, t, 0x1.32e57ap-5f);
    p = __fmaf_rn(p, t, -0x1.2122a4p-4f);
    p = __fmaf_rn(p, t, 0x1.23e4dep-3f);
    p = __fmaf_rn(p, t, -0x1.4853d8p-2f);
    p = __fmaf_rn(p, t, 0x1.ec7086p-1f);
    p = __fmaf_rn(p, t, 0x1.2b803ep-1f);
    return p;
}

// W(r) ~= log2((1 - u) / u), u = (r + 1/2) / 2^32.  The heat-bath rule s' = +1 iff u < 1 / (1 + exp(-2 beta x))
// (NMC/nmc.py:87) becomes  s' = +1 iff z < W(r),  z = -2 log2(e) beta x: the transcendental depends on the random
// number only, so it is evaluated where the random number is made (off the spin update's dependent chain).  Built
// from v = min(u, 1 - u): W(~r) == -W(r) exactly.  Restated in oracle/nlo.c (threshold_spec_f32), same bits.
__device__ __forceinline__ float threshold_spec(uint32_t r)
{
    const uint32_t m = (uint32_t)((int32_t)r >> 31);
    const uint32_t a = r ^ m;
    const float v = __fmaf_rn((float)a, 0x1p-32f, 0x1p-33f);
    const uint32_t b = __float_as_uint(v);
    const float ef = (float)((int32_t)(b >> 23) - 126);
    const float mant = __uint_as_float((b & 0x7FFFFFu) | 0x3F800000u);
    const float t1 = mant - 1.5f;
    const float t2 = __fmaf_rn(v, -2.0f, 0.5f);
    const float w = (log2_15_spec(t2) - log2_15_spec(t1)) - ef;
    return __uint_as_float((__float_as_uint(w) & 0x7FFFFFFFu) | (r & 0x80000000u));
}

// 64-bit wave reduction (wave64) by shuffles
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64_tree(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
