// vgl_common.hip.h -- device helpers shared by the gfx950 (MI355X, CDNA4) kernels of the vcfgl
// genotype-likelihood simulation hot path: rand48 arithmetic, the reference's samplers (rng.h) and the
// bounded float32 decisions that replace double transcendentals.  Wave64 only.
//
// Work decomposition (all kernels): one LANE owns one evaluation = one (site, sample); one WAVEFRONT
// owns 64 consecutive samples of one site, so every per-evaluation global access of a wave is one
// contiguous segment (structure-of-arrays tiles, sample index fastest).
//
//   vgl_sample.hip   k_sample<EQS>: Poisson depth -> per-read haplotype / base-call error / strand /
//                    quality-score draws -> staged reads + per-evaluation ACGT depth (tile mode)
//   vgl_serial.hip   k_scout, k_scout_wave, k_sample_serial: VGL_RNG_SERIAL, the reference's own
//                    draw order (sequential stream-state scout + parallel evaluation from its states)
//   vgl_gl.hip       k_site (allele order / status), k_gl<A> (likelihoods, PL / GP / AD),
//                    k_siteagg (order-dependent per-site float sums: QS, I16)
//
// The float32 accumulation order of the reference is kept exactly (double add rounded to
// float per genotype per read, float max, float subtract): build with -ffp-contract=off.
// Decisions that the reference takes on double transcendentals (rejection tests, floor of a
// scaled tan, (int)(-10 log10 p)) are taken from float32 hardware transcendentals with measured
// error bounds and fall back to the exact double expression inside the error band, behind
// wave-uniform branches: results equal the exact evaluation, the common path has no f64 log/tan/exp.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "vgl_device.h"


#define VGL_PI 3.141592654            // shared.h:37 (not M_PI)
#define CAP_BASEQ 63                  // shared.h:241
#define MAXPL 255                     // shared.h:208
#define F32_MISSING_BITS 0x7F800001u  // bcf_float_missing
#define I32_MISSING ((int32_t)0x80000000)

#define SITE_OK 0
#define SITE_SKIP_INVAR (-3)
#define SITE_SKIP_EMPTY (-4)
#define SITE_NO_READS 1

// ------------------------------------------------------------------------------------
// rand48: X <- (A X + C) mod 2^48, u = X 2^-48   (glibc drand48/erand48; rng.h:8-10)
__device__ __forceinline__ uint64_t lcg_next(uint64_t x) { return (x * VGL_LCG_A + VGL_LCG_C) & VGL_MASK48; }
__device__ __forceinline__ uint64_t aff(const VglAffine m, uint64_t x) { return (m.a * x + m.c) & VGL_MASK48; }
// state after n more steps (square-and-multiply on the affine map); rare paths only
static __device__ uint64_t rand48_jump(uint64_t st, uint64_t n) {
    uint64_t a = VGL_LCG_A, c = VGL_LCG_C, ra = 1, rc = 0;
    while (n) {
        if (n & 1) { ra = (ra * a) & VGL_MASK48; rc = (rc * a + c) & VGL_MASK48; }
        c = ((a + 1) * c) & VGL_MASK48;
        a = (a * a) & VGL_MASK48;
        n >>= 1;
    }
    return (ra * st + rc) & VGL_MASK48;
}
__device__ __forceinline__ double u01(uint64_t x) {
    // exact for x < 2^48: the 48 bits become the top of the mantissa of 1.xxx (what glibc's erand48 does)
    return __longlong_as_double((long long)(0x3FF0000000000000ULL | (x << 4))) - 1.0;
}
// u01(x) - 0.5 in one subtraction: (1 + f) - 1.5 = f - 0.5 is a multiple of 2^-48 in [-0.5, 0.5), so it is exact, like the
// two exact steps (1 + f) - 1.0 and f - 0.5 of the reference's `sample_uniform() - 0.5` (rng.h:74)
__device__ __forceinline__ double u01_minus_half(uint64_t x) {
    return __longlong_as_double((long long)(0x3FF0000000000000ULL | (x << 4))) - 1.5;
}
// The pool loop of k_sample<2> carries its generator states shifted left by 4 (x' = 16 x < 2^52): the 48 state bits then
// already sit where the mantissa of 1.xxx wants them, and every uniform saves the 64-bit shift of u01().  Same generator:
// x' <- (A x' + 16 C) mod 2^52.
#define VGL_MASK52 0xFFFFFFFFFFFFFULL
__device__ __forceinline__ uint64_t lcg_next52(uint64_t x) { return (x * VGL_LCG_A + (VGL_LCG_C << 4)) & VGL_MASK52; }
// ... and without the reduction mod 2^52: bits 52-63 of the result are garbage that no later step looks at (a product's low 52
// bits depend on the factors' low 52 bits only); u01_52r() masks while it builds the double
__device__ __forceinline__ uint64_t lcg_next52r(uint64_t x) { return x * VGL_LCG_A + (VGL_LCG_C << 4); }
// 1.xxx from a raw scaled state: the high word is (hi & 0xFFFFF) | 0x3FF00000 in ONE v_and_or_b32 (the compiler emits an and and
// an or: it will not spend a register on the second constant, which the ISA needs there); `k3ff` = 0x3FF00000 in a VGPR
__device__ __forceinline__ double bits_1xxx_52r(uint64_t x, uint32_t k3ff) {
    uint32_t hi;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(hi) : "v"((uint32_t)(x >> 32)), "s"(0xFFFFFu), "v"(k3ff));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | (uint32_t)x));
}
// m = a jump of the generator with its constant already scaled (c' = 16 c): the per-read table of k_sample<2> is stored that way
__device__ __forceinline__ uint64_t aff52(const VglAffine m, uint64_t x) { return (m.a * x + m.c) & VGL_MASK52; }
__device__ __forceinline__ double u01_52(uint64_t x) { return __longlong_as_double((long long)(0x3FF0000000000000ULL | x)) - 1.0; }
__device__ __forceinline__ double u01_52_minus_half(uint64_t x) { return __longlong_as_double((long long)(0x3FF0000000000000ULL | x)) - 1.5; }
__device__ __forceinline__ double next_u(uint64_t& st) { st = lcg_next(st); return u01(st); }

// gamma_ln, rng.h:38-43,60-64
static __device__ double gamma_ln_dev(const double xx) {
    const double cof[6] = {76.18009172947146, -86.50532032941677, 24.01409824083091,
                           -1.231739572450155, 0.1208650973866179e-2, -0.5395239384953e-5};
    double x, tmp, y, ser;
    y = x = xx;
    tmp = x + 5.5;
    tmp -= (x + 0.5) * log(tmp);
    ser = 1.000000000190015;
#pragma unroll
    for (int j = 0; j <= 5; j++) ser += cof[j] / ++y;
    return -tmp + log(2.5066282746310005 * ser / x);
}

// one depth draw, rng.h:289-312.  gamma_ln(em + 1) of the integer em comes from a table the host
// fills with the same formula (rng.h:60-64); larger arguments are evaluated here.
static __device__ int poisson_draw(const VglPois& p, uint64_t& st, const double* __restrict__ glt, const int glt_n) {
    double em, t;
    if (p.st12) {
        em = -1.0; t = 1.0;
        do { ++em; t *= next_u(st); } while (t > p.g);
    } else {
        double y;
        do {
            do {
                y = tan(VGL_PI * next_u(st));
                em = p.sq * y + p.lm;
            } while (em < 0.0);
            em = floor(em);
            const double gl = (em < (double)(glt_n - 1)) ? glt[(int)em + 1] : gamma_ln_dev(em + 1.0);
            t = 0.9 * (1.0 + y * y) * exp(em * p.alxm - gl - p.g);
        } while (next_u(st) > t);
    }
    return (int)em;
}

// The same draw with the transcendental work in float32 (rng.h:300-312 otherwise unchanged).
// tan() only feeds (1) em = floor(sq*y + lm) and (2) the acceptance bound t, exp() only (2): both are
// decisions, so float32 values with explicit error bounds decide them and the exact double
// expressions are evaluated only inside the error band.  Bounds measured on MI355X
// (tools/vlogcheck.py): |tanf(x) - tan(x)| <= 1.2 ulp, v_exp_f32 <= 0.71 ulp; 4x margins used.
// Flat loop: one attempt per iteration for every lane that has not accepted yet.
// The bounds themselves are the helpers below; tests/test_gpu_bounds.py sweeps every float32 argument of each
// hardware function on the device and asserts them (vgl_bounds.hip), so a toolchain change that moves one fails the suite.
// tan(a) for a float a in (0, pi] (the rejection sampler's tan(PI u)), in float32: quadrant n = rint(a 2/pi) in {0, 1, 2}, r = a - n pi/2 by a
// two-constant reduction (a - n hi is exact for these n; the smallest |r| next to the pole is 4.4e-8, kept to 2^-24 relative), tan r =
// r + r^3 P(r^2) on [-pi/4, pi/4] (the classic degree-6 minimax in r^2), -1 / tan r (v_rcp_f32, 1 ulp) in the odd quadrant.  About 20
// instructions; the general-argument library tanf spends 80 on its branch-free large-argument reduction, which no argument here needs.
// Error against tan(a): asserted below 4 ulp + the argument term by tests/test_gpu_bounds.py over EVERY float32 of the range.
__device__ __forceinline__ float vgl_tanf_0pi(const float a) {
    const float n = __builtin_rintf(a * 0.63661977f);
    float r = __builtin_fmaf(n, -1.57079637f, a);                     // (float)(pi/2)
    r = __builtin_fmaf(n, 4.37113883e-8f, r);                          // pi/2 = 1.57079637 - 4.37113883e-8
    const float z = r * r;
    float p = 9.38540185543e-3f;
    p = __builtin_fmaf(p, z, 3.11992232697e-3f); p = __builtin_fmaf(p, z, 2.44301354525e-2f); p = __builtin_fmaf(p, z, 5.34112807005e-2f);
    p = __builtin_fmaf(p, z, 1.33387994085e-1f); p = __builtin_fmaf(p, z, 3.33331568548e-1f);
    const float t = __builtin_fmaf(p * z, r, r);
    return (n == 1.0f) ? -__builtin_amdgcn_rcpf(t) : t;
}
// One rejection attempt (rng.h:302-309) decided in float32 (round 4: no float64 arithmetic but the three operations of the exponent).
//   a = PI u1, u1 = st1 2^-48.  af = (float)(st1 >> 16) * (float)(PI 2^-32): |af - a| <= a 2.5 x 2^-24 + PI 2^-32 (the truncated 16 bits,
//   the conversion, the constant -- 2^-25.1 -- and the product).
//   yf = vgl_tanf_0pi(af):  |yf - tan(af)| <= |yf| 2^-21 + (1 + yf^2) af 2^-24 for EVERY float of the range (VGL_BOUND_TANF), and
//   tan(af + d) - tan(af) = tan d (1 + Y^2) / (1 - Y tan d), Y = tan(af): at most 2 |d| (1 + Y^2) while |Y d| <= 1/2.  Hence
//       dy = |yf| 2^-21 + (1 + yf^2) (af 7 x 2^-24 + 2e-9)  >=  |yf - tan(a)|          [7 > 1 + 2 x 2.5 with room for yf^2 against Y^2]
//   in that regime; outside it (next to the pole at PI/2) dy > |yf|, which (a) puts sq dy > 1 over every test of the floor below and
//   (b) fails the explicit guard dy <= |yf| / 2 in front of the two shortcuts -- nothing is decided from an invalid bound.
//   e0f = fma(sqf, yf, lmf) against e0 = sq tan(a) + lm:  derr = sqf dy + 2^-22 (|sqf yf| + lmf) + 1e-7 (the float parameters, the fma, the
//   subtraction 1 - fr below, with a factor 2 in hand).
//   Decisions: e0 < 0 (`neg`; nothing else of the attempt is then used) when e0f < -derr; e0 >= 0 when e0f > derr, and then either
//     * the attempt is rejected whatever floor(e0) is: e0f - derr >= p.e_hi, the host's bound beyond which 0.9 (1 + y^2) exp(..) < 2^-60
//       for every em (pois_init), against an acceptance draw u2 >= 2^-32 -- the Lorentzian's tail: |y| > 300 in one attempt of 500, which
//       the float bounds of the floor cannot resolve and the gamma_ln table does not reach;
//     * or floor(e0) = floorf(e0f) (both distances to the neighbouring integers exceed derr), inside the table, and the acceptance test
//       u2 > t is decided from tt = 0.9f (1 + yf^2) v_exp_f32(z log2 e) with relative error dy + |z| 2^-22 + 2^-19 (VGL_BOUND_EXP2; the
//       factor (1 + y^2) contributes 2 |y| dy / (1 + y^2) <= dy) against u2f = (float)(st2 >> 16) 2^-32 (2^-24 relative + 2^-32).
//   Anything else is `amb`: the caller evaluates the attempt exactly (poisson_exact), behind a wave-uniform branch.
// tests/test_gpu_bounds.py (VGL_BOUND_POISSON) runs both on 2^32 attempts per mean depth -- pseudo-random, next to the pole, next to
// integer e0 -- and requires every decision poisson_fast() does not call ambiguous to equal the exact one; it prints the ambiguous share.
// ZT: the exponent (em alxm - gamma_ln(em + 1) - g) log2 e comes from `zt`, tabulated in float32 for THIS p (VglDevParams::pois_zt: one mean
// depth for all samples) -- its rounding, 2^-24 |zt|, is the rounding of (float)(z log2 e) in the other branch, with the same allowance.
template <bool ZT>
__device__ __forceinline__ void poisson_fast(const VglPois& p, const uint64_t st1, const uint64_t st2, const double* __restrict__ glt, const int glt_n,
                                             const float* __restrict__ zt, bool& neg, bool& rej, int& em, bool& amb) {
    const float af = (float)(uint32_t)(st1 >> 16) * ((float)VGL_PI * 0x1p-32f);    // (the constant: 0x1.921fb6p-31)
    const float yf = vgl_tanf_0pi(af);
    const float y2 = yf * yf, ay = fabsf(yf);
    const float dy = __builtin_fmaf(ay, 0x1p-21f, (1.0f + y2) * __builtin_fmaf(af, 7.0f * 0x1p-24f, 2e-9f));
    const float e0f = __builtin_fmaf(p.sqf, yf, p.lmf);
    const float derr = __builtin_fmaf(p.sqf, dy, 0x1p-22f * (fabsf(p.sqf * yf) + p.lmf)) + 1e-7f;
    const bool guard = dy <= 0.5f * ay;
    const float emf = __builtin_floorf(e0f);
    const float fr = e0f - emf;                                              // exact
    const bool floor_ok = (fr > derr) & (1.0f - fr > derr) & (fabsf(e0f) < 1.0e6f);
    const bool sure_neg = guard & (e0f < -derr);
    const bool sure_pos = e0f > derr;
    const uint32_t hi2 = (uint32_t)(st2 >> 16);
    const bool sure_rej = guard & (e0f - derr >= p.e_hi) & (hi2 != 0u);
    em = (int)emf;                                                           // (saturates; used only where floor_ok)
    const bool in_tab = (em >= 0) & (em < glt_n - 1);
    float zl, za;                                                            // z log2 e, and |z| (or more)
    if (ZT) { zl = zt[in_tab ? em : 0]; za = fabsf(zl); }
    else {
        const double gl = glt[in_tab ? em + 1 : 1];
        const double z = (double)emf * p.alxm - gl - p.g;
        zl = (float)(z * 1.4426950408889634); za = fabsf((float)z);
    }
    const float ex = __builtin_amdgcn_exp2f(zl);
    const float tt = 0.9f * (1.0f + y2) * ex;
    const float rel_t = dy + za * 0x1p-22f + 0x1p-19f;
    const float u2f = (float)hi2 * 0x1p-32f;
    const bool t_ok = fabsf(u2f - tt) > __builtin_fmaf(tt, rel_t, __builtin_fmaf(u2f, 0x1p-22f, 0x1p-30f));
    neg = sure_neg;
    rej = sure_rej | (u2f > tt);
    amb = !(sure_neg | (sure_pos & (sure_rej | (floor_ok & in_tab & t_ok))));
}
// the attempt as the reference evaluates it (rng.h:302-309)
__device__ __forceinline__ void poisson_exact(const VglPois& p, const uint64_t st1, const uint64_t st2, const double* __restrict__ glt, const int glt_n,
                                              bool& neg, bool& rej, int& em) {
    const double y = tan(VGL_PI * u01(st1));
    double eme = p.sq * y + p.lm;
    neg = eme < 0.0;
    eme = floor(eme);
    const double t = 0.9 * (1.0 + y * y) * exp(eme * p.alxm - ((eme >= 0.0 && eme < (double)(glt_n - 1)) ? glt[(int)eme + 1] : gamma_ln_dev(eme + 1.0)) - p.g);
    rej = u01(st2) > t;
    em = (int)eme;
}
// one rejection attempt from the two generator states it would consume: `neg`: em < 0
// (only st1 is consumed, no acceptance draw), else `rej` = the acceptance draw u(st2) exceeds t, else the draw is `em`.
template <bool ZT>
__device__ __forceinline__ void poisson_attempt(const VglPois& p, const uint64_t st1, const uint64_t st2, const bool need,
                                                const double* __restrict__ glt, const int glt_n, const float* __restrict__ zt, bool& neg, bool& rej, int& em) {
    bool amb;
    poisson_fast<ZT>(p, st1, st2, glt, glt_n, zt, neg, rej, em, amb);
    amb &= need;
    if (__builtin_expect(__ballot(amb) != 0, 0)) {               // exact evaluation (rare, wave-uniform branch)
        asm volatile("" ::: "memory");
        bool nege, reje; int eme;
        poisson_exact(p, st1, st2, glt, glt_n, nege, reje, eme);
        neg = amb ? nege : neg; em = amb ? eme : em; rej = amb ? reje : rej;
    }
}

static __device__ int poisson_draw_fast(const VglPois& p, uint64_t& st, const double* __restrict__ glt, const int glt_n) {
    if (p.st12) {
        double em = -1.0, t = 1.0;
        do { ++em; t *= next_u(st); } while (t > p.g);
        return (int)em;
    }
    bool done = false;
    int em_res = 0;
    while (__ballot(!done)) {
        const uint64_t st1 = lcg_next(st);
        const uint64_t st2 = lcg_next(st1);
        bool neg, reject; int em;
        poisson_attempt<false>(p, st1, st2, !done, glt, glt_n, nullptr, neg, reject, em);
        const bool acc = !done & !neg & !reject;
        st = done ? st : (neg ? st1 : st2);                          // em < 0 consumes one draw, an attempt two
        em_res = acc ? em : em_res;
        done = done | acc;
    }
    return em_res;
}

// sample_NormalSampler_0_1_0, rng.h:70-80
static __device__ double normal_rou(uint64_t& st) {
    double u, v, x, y, q;
    do {
        u = next_u(st);
        v = 1.7156 * (next_u(st) - 0.5);
        x = u - 0.449871;
        y = fabs(v) + 0.386595;
        q = (x * x) + y * (0.19600 * y - 0.25472 * x);
    } while ((q > 0.27597) && (q > 0.27846 || (v * v) > -4.0 * log(u) * (u * u)));
    return v / u;
}

// Gamma1Sampler::sample, rng.h:133-152
static __device__ double gamma1_draw(const VglGamma1& g, uint64_t& st) {
    double u, v, x, xsq;
    do {
        do {
            x = normal_rou(st);
            v = 1.0 + g.a2 * x;
        } while (v <= 0.0);
        v = v * v * v;
        u = next_u(st);
        xsq = x * x;
    } while (u > 1.0 - 0.0331 * (xsq * xsq) && log(u) > 0.5 * xsq + g.a1 * (1.0 - v + log(v)));
    if (g.changed) {
        while ((u = next_u(st)) == 0.0);
        return pow(u, 1.0 / g.alpha0) * g.a1 * v;
    }
    return g.a1 * v;
}

// BetaSampler::sample, rng.h:433-444
__device__ __forceinline__ double beta_draw(const VglDevParams& P, uint64_t& st) {
    double x = gamma1_draw(P.gx, st);
    double y = gamma1_draw(P.gy, st);
    return x / (x + y);
}

// apply_qs_bins, vcfgl.cpp:57-64
static __device__ int apply_bins(const VglDevParams& P, int q, uint32_t* errflag) {
    for (int i = 0; i < P.n_qs_bins; ++i)
        if (q >= P.qs_bins[3 * i] && q <= P.qs_bins[3 * i + 1]) return P.qs_bins[3 * i + 2];
    atomicOr(errflag, VGL_DEVERR_QSBIN);
    return 0;
}

// IEEE-754 quotient n / d for operands far from the exponent limits (here d = u in [2^-48, 1),
// |n| < 1): v_rcp_f64, two Newton steps and the residual correction -- the compiler's own division
// sequence without v_div_scale / v_div_fixup, which only act on extreme exponents, NaN and infinity.
__device__ __forceinline__ double div_inrange(const double n, const double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0); y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0); y = __builtin_fma(y, e, y);
    const double q = n * y;
    const double r = __builtin_fma(-d, q, n);
    return __builtin_fma(r, y, q);
}

// (float)(q / sum) for integer-valued floats 0 <= q <= sum <= 2^24 (INFO/QS: a base's share of a sample's quality sum, vcfgl.cpp:893)
// through ONE double reciprocal per sum: the exact quotient of two such integers is never a rounding tie of float32 (a dyadic q / sum
// has at most 24 significant bits, i.e. is itself a float32) and lies at least 2^-49 (relative) from the nearest one, while
// q * (1 / sum) in double -- v_rcp_f64 and two Newton steps, then one multiplication -- is within 2^-51 of it: rounding that double
// to float32 gives the correctly rounded quotient, what the IEEE division sequence (ten instructions per quotient) returns.
// tests/test_gpu_bounds.py compares the two over every pair up to 4096 and 4e9 pseudo-random larger pairs (vgl_bounds.hip).
__device__ __forceinline__ double recip_int24(const float sum) {
    const double ds = (double)sum;
    double r = __builtin_amdgcn_rcp(ds);
    r = __builtin_fma(__builtin_fma(-ds, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-ds, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float quot_int24(const float q, const double recip_sum) { return (float)((double)q * recip_sum); }

// x / 10 in float32, correctly rounded, for finite x >= 0 (GL model 1: GL = -cost / 10, gl_methods.cpp:271-290 -- the reference divides in
// double and rounds to float, which for a float numerator IS the float quotient): q0 = x RN(1/10), the exact residual, one correction
// (Markstein) -- three instructions where the IEEE sequence takes ten.  Compared with the division over EVERY non-negative finite
// float32 by tests/test_gpu_bounds.py (vgl_bounds.hip).
__device__ __forceinline__ float div10_f32(const float x) {
    const float y = 0.1f;                                            // RN(1/10)
    const float q0 = x * y;
    const float r = __builtin_fmaf(-10.0f, q0, x);
    float q = __builtin_fmaf(r, y, q0);
    if (__builtin_expect(__ballot(x != 0.0f && x < 0x1p-100f) != 0, 0)) {    // next to the subnormal range the residual is no longer exact: the division itself (never seen on real costs)
        asm volatile("" ::: "memory");
        q = (x < 0x1p-100f) ? x / 10.0f : q;
    }
    return q;
}

// error probability -> qScore / adjusted qScore, vcfgl.cpp:500-523
static __device__ void errprob_to_qs(const VglDevParams& P, double ep, int& q, int& aq, uint32_t* errflag) {
    q = -1; aq = -1;
    if (0.0 == ep) q = CAP_BASEQ;
    else if (1.0 == ep) q = 0;
    else {
        double tmp = -10.0 * log10(ep);
        q = (int)tmp;
        if (P.adjust_qs) aq = (int)(tmp + P.adjust_by);
    }
    if (P.n_qs_bins != 0) {
        q = apply_bins(P, q, errflag);
        if (P.adjust_qs) aq = apply_bins(P, aq, errflag);
    } else {
        q = (q > CAP_BASEQ) ? CAP_BASEQ : q;
        if (P.adjust_qs) aq = (aq > CAP_BASEQ) ? CAP_BASEQ : aq;
    }
}

__device__ __forceinline__ int qs_to_qssq(int q) { return (0 == q) ? 0 : ((q < CAP_BASEQ) ? q * q : 3969); }  // shared.h:459

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Workgroups are handed to the 8 XCDs of the part round-robin by index, and each XCD has its own L2.  xcd_block() maps
// the hardware index to a logical one so that every XCD walks ONE contiguous range of the tile (sites in order): the
// pieces of a plane that neighbouring workgroups write then meet in one L2 and leave it as long runs (a bijection of
// [0, n) for any n; only an affinity -- nothing depends on where a workgroup really runs).
__device__ __forceinline__ uint32_t xcd_block(const uint32_t b, const uint32_t n) {
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
}

// wave -> (local site, 64-sample chunk); everything here is wave-uniform (SGPRs)
struct WavePos { int ls; int chunk; int wib; bool valid; };
__device__ __forceinline__ WavePos wave_pos_of(const VglDevParams& P, const VglTilePtrs& T, const int64_t w) {
    WavePos r;
    r.wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    r.valid = w < (int64_t)T.n_sites * P.chunks;
    r.ls = (int)(w / P.chunks);
    r.chunk = (int)(w - (int64_t)r.ls * P.chunks);
    return r;
}
__device__ __forceinline__ WavePos wave_pos(const VglDevParams& P, const VglTilePtrs& T) {
    return wave_pos_of(P, T, (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
}

// ---- decisions of the rejection samplers without a float64 logarithm ---------------------
// The reference compares expressions that contain log() in double.  The hardware float32
// log2 (v_log_f32) with an explicit error bound decides the comparison unless the two sides
// are closer than that bound; only then is the exact double expression evaluated, so the
// decision is always the one the exact expression gives.
// Bound: for normal float x, |v_log_f32(x) - log2(x)| <= 2^-22.9 |log2 x| (measured on MI355X
// over 1.2e7 inputs incl. a dense set around 1: tools/vlogcheck.py; relative to |x-1| near 1);
// with the double->float rounding of the argument and the multiplication by ln 2:
//     |fast_ln(x) - ln(x)| <= |fast_ln(x)| 2^-21 + 2^-22
#define VGL_LN2 0.6931471805599453
__device__ __forceinline__ double fast_ln(const float xf) { return (double)__builtin_amdgcn_logf(xf) * VGL_LN2; }
__device__ __forceinline__ double fast_ln_err(const double l) { return fabs(l) * 0x1p-21 + 0x1p-22; }

// (v*v) > -4.0*log(u)*(u*u)                                              rng.h:78
// `need` = lanes whose result is used; only those can force the exact evaluation.
// DEFER: the lanes the bound cannot decide are reported in `undecided` instead of being settled by the double log here (k_sample<2>
// hands their reads to k_redo: the exact code then costs this kernel neither registers nor scratch)
template <bool DEFER>
__device__ __forceinline__ bool normal_slow_test_t(const double v, const double u, const bool need, bool& undecided) {
    const double lhs = v * v;
    const float uf = (float)u;
    const double l = fast_ln(uf);
    const double uu = u * u;
    const double rhs = -4.0 * l * uu;
    const double m = 4.0 * uu * fast_ln_err(l);
    const bool hi = lhs > rhs + m, lo = lhs < rhs - m;
    bool res = hi;
    const bool amb = need && !((hi || lo) && (uf > 0.0f));
    undecided = DEFER && amb;
    if (!DEFER) {
        if (__builtin_expect(__ballot(amb) != 0, 0)) {      // wave-uniform and rare: a real branch
            asm volatile("" ::: "memory");                  // (keeps the compiler from speculating the double log)
            const bool ex = lhs > -4.0 * log(u) * (u * u);
            res = amb ? ex : res;
        }
    }
    return res;
}
__device__ __forceinline__ bool normal_slow_test(const double v, const double u, const bool need) {
    bool und;
    return normal_slow_test_t<false>(v, u, need, und);
}

// pieces of gamma_slow_test (asserted by tests/test_gpu_bounds.py over every float32 argument):
// ln(u) from the float32 log2 and a float multiplication: |lu - ln(u)| <= |lu| 2^-20 + 2^-21 (argument rounding included)
__device__ __forceinline__ float gamma_test_lu(const float uf) { return __builtin_amdgcn_logf(uf) * 0.69314718f; }
// g = 3 a1 s^4 (1/4 - s/5 + ... + s^10/14) = -(0.5 x^2 + a1 (1 - v + log v)) for |s| <= 1/3, within 4e-6 g + 1e-10.
// Horner steps as explicit fused multiply-adds: this is the bounded estimate, not the reference's arithmetic
// (the translation unit is built with -ffp-contract=off), and a fused step only tightens the bound
__device__ __forceinline__ float gamma_rhs_series(const float sf, const float a1f) {
    const float ns = -sf;
    float p = 1.0f / 14.0f;
    p = __builtin_fmaf(ns, p, 1.0f / 13.0f); p = __builtin_fmaf(ns, p, 1.0f / 12.0f); p = __builtin_fmaf(ns, p, 1.0f / 11.0f);
    p = __builtin_fmaf(ns, p, 1.0f / 10.0f); p = __builtin_fmaf(ns, p, 1.0f / 9.0f); p = __builtin_fmaf(ns, p, 1.0f / 8.0f);
    p = __builtin_fmaf(ns, p, 1.0f / 7.0f); p = __builtin_fmaf(ns, p, 1.0f / 6.0f); p = __builtin_fmaf(ns, p, 1.0f / 5.0f);
    p = __builtin_fmaf(ns, p, 1.0f / 4.0f);
    const float s2 = sf * sf;
    return 3.0f * a1f * (s2 * s2) * p;
}
__device__ __forceinline__ float gamma_test_margin(const float lu, const float g) {
    return fabsf(lu) * 0x1p-20f + 0x1p-21f + g * 4e-6f + 1e-10f;
}

// log(u) > 0.5*xsq + a1*(1.0 - v + log(v)),  v = fl(fl(w*w)*w), w = fl(1 + a2 x)    rng.h:139-145
// With s = a2 x and a2^2 = 1/(9 a1) the quadratic terms cancel analytically:
//     0.5 x^2 + a1 (1 - (1+s)^3 + 3 ln(1+s)) = -3 a1 s^4 (1/4 - s/5 + s^2/6 - s^3/7 + ...)
// a well-conditioned series (no cancellation), so float32 is enough for the bounded decision:
// |s| <= 1/3 => 11 terms leave < 2e-6 relative truncation (first omitted term s^11/15 against P ~ 0.2); float32
// evaluation < 1e-6 relative: together inside the 4e-6 g of the margin (swept: tests/test_gpu_bounds.py);
// the reference's own double rounding of its expression is < 1e-12.  Outside the band (or for
// |s| > 1/3) the exact double expression is evaluated.
template <bool DEFER>
__device__ __forceinline__ bool gamma_slow_test_t(const double u, const double xsq, const double a1, const double v,
                                                  const double s, const bool need, bool& undecided) {
    const float uf = (float)u, sf = (float)s, a1f = (float)a1;
    const float lu = gamma_test_lu(uf);
    const float g = gamma_rhs_series(sf, a1f);                  // = -(rhs of the reference), >= 0
    const float d = lu + g;                                     // log(u) - rhs
    const float m = gamma_test_margin(lu, g);
    const bool ok = (fabsf(d) > m) && (fabsf(sf) <= 0.3333f) && (uf > 0.0f);
    bool res = d > 0.0f;
    const bool amb = need && !ok;
    undecided = DEFER && amb;
    if (!DEFER) {
        if (__builtin_expect(__ballot(amb) != 0, 0)) {      // wave-uniform and rare: a real branch
            asm volatile("" ::: "memory");
            const bool ex = log(u) > 0.5 * xsq + a1 * (1.0 - v + log(v));
            res = amb ? ex : res;
        }
    }
    return res;
}
__device__ __forceinline__ bool gamma_slow_test(const double u, const double xsq, const double a1, const double v,
                                                const double s, const bool need) {
    bool und;
    return gamma_slow_test_t<false>(u, xsq, a1, v, s, need, und);
}

// ---- the pool loop of k_sample<2> in float32 (round 5; the deferred builds without --precise-gl 1) ------------------------------------
// What a read's quality score needs from the beta sampler (rng.h:433-444: two gamma deviates, each a chain of ratio-of-uniforms normal
// attempts, rng.h:72-78 / :133-145) is (a) the DECISIONS of the rejection loops, exactly as the reference's doubles take them, and (b) the
// quotient X / (X + Y) closely enough for floor(-10 log10 p).  Neither needs float64 arithmetic on every attempt: the loop below forms the
// uniforms as float32 straight from the generator's integer state (no conversion from double anywhere), evaluates every expression in
// float32 (two-operand float32 forms issue in 2.6 cycles per wavefront on this part, float64 in 4.3: DESIGN.md section 5) and compares
// against thresholds moved by an explicit bound of the float32 error; an attempt whose comparison falls inside a bound -- and every
// attempt outside the range the value bound below is stated for (u < 2^-9, w < 0.5) -- sends its READ to k_redo, which draws it in
// double like the reads the dense pass cannot settle.  Every decision that is taken here is therefore the reference's (the bounds are
// derived below and swept by tests/test_gpu_bounds.py, mode VGL_BOUND_POOL32, against the float64 expressions).
//
// Units: a uniform u = X / 2^48 is carried as uf = (float)T, T = the top 32 bits of X, i.e. u 2^32 within |T| 2^-24 + 1; the normal
// attempt's v = 1.7156 (u' - 0.5) as sv = 1.7156f (float)(int32)(T' - 2^31), i.e. v 2^32 within |sv| 3 x 2^-24 + 1.8 (the integer
// subtraction is exact, so the relative precision holds next to v = 0).  eps = 2^-24 below.
//
// q = x^2 + y (0.196 y - 0.25472 x), x = u - 0.449871, y = |v| + 0.386595 (|x| <= 0.55, y <= 1.2444, |0.196 y - 0.25472 x| <= 0.38):
// |dx| <= 1.1e-7, |dy| <= 2.3e-7, |dt| <= 1.3e-7, |d(x^2)| <= 1.3e-7, |d(y t)| <= 2.4e-7, last rounding 6e-8: |dq| <= 4.4e-7 < VGL_P32_QBAND.
#define VGL_P32_QBAND 1.0e-6f
#define VGL_P32_QLO 0.27597f
#define VGL_P32_QHI 0.27846f
// gamma step, stated for u >= 2^-9 (T >= 2^23: the 16 state bits below T are then <= 2^-23 of u) and w >= 0.5:
// rel(xn = v / u) <= 9 eps (+ 2e-7 absolute, from the bits below T'), rel(w = 1 + a2 xn) <= 12 eps (|a2 xn| / w <= 1 for w >= 0.5),
// rel(w^3) <= 38 eps, rel(a1 w^3) <= 40 eps = 2.4e-6; p = X / (X + Y) through v_rcp_f32: <= 84 eps = 5.0e-6, i.e. 2.2e-5 in
// tf = -10 log10 p (VGL_P32_TF_EXTRA, added to qs_tf_margin by the dense pass of these builds).
#define VGL_P32_TF_EXTRA 2.2e-5f
#define VGL_P32_UMIN 0x1p23f
// squeeze u2 > 1 - 0.0331 x^4 (rng.h:143), in units of 2^-32: |d u2| <= 6e-8, |d(0.0331 x^4)| <= 0.0331 (34 eps x^4 + 1.7e-6 |xn|^3)
// <= 1.2e-7 (1 + x^4) -> a band of 3e-7 (1 + x^4) on either side, folded into the constants of one fused multiply-add each:
// u2f > R_lo: the squeeze MAY fail (everything else: it surely holds); u2f > R_hi: it surely fails.
#define VGL_P32_SQ_K_LO (-(0.0331f + 3.0e-7f) * 0x1p32f)
#define VGL_P32_SQ_C_LO ((1.0f - 3.0e-7f) * 0x1p32f)
#define VGL_P32_SQ_K_HI (-(0.0331f - 3.0e-7f) * 0x1p32f)
#define VGL_P32_SQ_C_HI ((1.0f + 3.0e-7f) * 0x1p32f)

// one step of x' <- A x' + 16 C on a state carried as (lo, hi) of the raw 52-bit form (lcg_next52r): the low product and the constant
// in one v_mad_u64_u32; of the high word only bits 0-19 are ever looked at, so hi * A_lo and lo * A_hi (A_hi = 5) go through the 24-bit
// multiply-add (bits 20 and up of either operand reach bits 20 and up of the result only): three instructions (four with v_mul_lo_u32)
__device__ __forceinline__ void lcg52_step(const uint32_t lo, const uint32_t hi, uint32_t& nlo, uint32_t& nhi) {
    const uint64_t p = (uint64_t)lo * (uint32_t)(VGL_LCG_A & 0xFFFFFFFFULL) + (uint64_t)(VGL_LCG_C << 4);
    uint32_t h = (uint32_t)(p >> 32);
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(h) : "v"(hi), "s"((uint32_t)(VGL_LCG_A & 0xFFFFFFULL)));
    asm("v_mad_u32_u24 %0, %1, 5, %0" : "+v"(h) : "v"(lo));
    nlo = (uint32_t)p; nhi = h;
}
// a x + c for a jump (a, c) of the generator on the same (lo, hi) form: three instructions likewise
__device__ __forceinline__ void aff52_step(const VglAffine m, const uint32_t lo, const uint32_t hi, uint32_t& nlo, uint32_t& nhi) {
    const uint64_t p = (uint64_t)lo * (uint32_t)m.a + m.c;
    uint32_t h = (uint32_t)(p >> 32);
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(h) : "v"(hi), "v"((uint32_t)m.a));
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(h) : "v"(lo), "v"((uint32_t)(m.a >> 32)));
    nlo = (uint32_t)p; nhi = h;
}
// T = the top 32 bits of the 48-bit state = bits 20-51 of the scaled form
__device__ __forceinline__ uint32_t lcg52_top32(const uint32_t lo, const uint32_t hi) { return __builtin_amdgcn_alignbit(hi, lo, 20); }
__device__ __forceinline__ float pool32_u(const uint32_t T) { return (float)T; }                                   // u 2^32
__device__ __forceinline__ float pool32_sv(const uint32_t T) { return (float)(int32_t)(T ^ 0x80000000u) * 1.7156f; }   // v 2^32
__device__ __forceinline__ float pool32_q(const float uf, const float sv) {
    const float x = __builtin_fmaf(uf, 0x1p-32f, -0.449871f);
    const float y = __builtin_fmaf(fabsf(sv), 0x1p-32f, 0.386595f);
    float t = 0.19600f * y;
    t = __builtin_fmaf(-0.25472f, x, t);
    return __builtin_fmaf(y, t, x * x);
}
// (v*v) > -4.0*log(u)*(u*u) (rng.h:78) for an attempt whose q_f lies in (QLO - band, QHI + band]: the reference looks at this test
// only when q > 0.27597 and q <= 0.27846.  Returns the loop's `reject`; `undecided`: the float32 values cannot tell (the read goes to k_redo).
// Divided by u^2 the test reads xn^2 > -4 ln u with xn = v / u, the quotient the gamma step needs anyway.  For uf >= 2^23 (asked for):
// xn within 9 eps |xn| + 2.1e-7, so xn^2 within 18 eps xn^2 + 4.2e-7 |xn|; ln u = ln 2 log2(uf 2^-32) with v_log_f32 on [2^-9, 1)
// (|error| <= 2^-22.9 |log2| <= 1.2e-6: VGL_BOUND_FAST_LN's sweep), the argument within 3 eps and the float32 constant and product
// within 2^-23 of the result (<= 25): -4 ln u within 7e-6.  Margin: 2^-19 xn^2 + 1.5e-5.
__device__ __forceinline__ bool pool32_normal_slow(const float xn, const float uf, const float q, const bool need, bool& undecided) {
    const float lhs = xn * xn;
    const float rhs = __builtin_amdgcn_logf(uf * 0x1p-32f) * -2.7725887f;     // -4 ln 2 log2 u
    const float m = __builtin_fmaf(lhs, 0x1p-19f, 1.5e-5f);
    const bool hi = lhs > rhs + m, lo = lhs < rhs - m;
    const bool near_lo = q < VGL_P32_QLO + VGL_P32_QBAND;      // q may be <= 0.27597: accepted without the test
    const bool near_hi = q > VGL_P32_QHI - VGL_P32_QBAND;      // q may be > 0.27846: rejected without the test
    undecided = need && (!(hi || lo) || (near_lo && hi) || (near_hi && lo) || !(uf >= VGL_P32_UMIN));
    return hi;
}
// The gamma step's acceptance (rng.h:143-144: reject while u2 > 1 - 0.0331 x^4 && log(u2) > 0.5 xsq + a1 (1 - v + log v)) for a lane whose
// sure-accept bound did not hold.  (The loop's common path tests only that bound: for x^4 >= 1e-5 it is implied by the squeeze --
// 0.15 a2^2 < 0.0331 -- and below that the squeeze fails for 3e-7 of the draws; so the squeeze is looked at here, with its band.)
// s = a2 xn within |s| 10 eps + 2.4e-8, so g = 3 a1 s^4 P(s) (gamma_rhs_series) within g (40 eps + 1e-7 / |s|) on top of the series' own
// 4e-6 g; ln(u2) from v_log_f32 of u2f 2^-32 (u2f >= 2^23 asked for; the argument within 3 eps < 2^-21).  Returns the loop's `reject`;
// decided when the squeeze surely holds (accept), the logarithm test is outside its margin on the accepting side (accept), or on the
// rejecting side with the squeeze surely failing (reject); undecided otherwise.
__device__ __forceinline__ bool pool32_gamma_slow(const float u2f, const float sf, const float a1f, const float x4, const bool need, bool& undecided) {
    const bool sq_holds = !(u2f > __builtin_fmaf(x4, VGL_P32_SQ_K_LO, VGL_P32_SQ_C_LO));
    const bool sq_fails = u2f > __builtin_fmaf(x4, VGL_P32_SQ_K_HI, VGL_P32_SQ_C_HI);
    const float lu = gamma_test_lu(u2f * 0x1p-32f);
    const float g = gamma_rhs_series(sf, a1f);
    const float d = lu + g;
    const float m = gamma_test_margin(lu, g) + 0x1p-21f + g * (4.0e-6f + 1.0e-7f * __builtin_amdgcn_rcpf(fabsf(sf)));
    const bool rej_log = d > m, acc_log = d < -m;
    const bool log_ok = (fabsf(sf) <= 0.3333f) && (u2f >= VGL_P32_UMIN);
    undecided = need && !(sq_holds || (log_ok && (acc_log || (rej_log && sq_fails))));
    return !sq_holds && rej_log;
}

// error probability -> qScore / adjusted qScore (vcfgl.cpp:500-523) in two steps.  k_sample<2> leaves
// p = gx / (gx + gy) (rng.h:438) of every finished read in LDS as a float32 (qs_stage_pf), and a dense pass over
// the wave's reads -- 64 useful lanes per instruction, where the pool loop would spend the same instructions
// on every iteration for the few lanes that finish a read in it -- takes (int)(-10*log10(p)) from the float32
// log2 unless p sits within its error bound of an integer boundary (qs_decide_pf); those reads (about 1 in
// 10^4) are drawn again in double by the lane that owns them (beta_draw + errprob_raw).
// pf: v_rcp_f32 is 1 ulp, so its relative error is <= 2^-24 (gx) + 2^-24 (sum) + 2^-23 (rcp) + 2^-24 (product)
// = 1.25 x 2^-22 < 3.0e-7, i.e. < 1.3e-6 in tf; tf = -10 log10(p) within |tf| 2^-19 + 4e-6 of the exact value with that
// argument error included (qs_tf_margin; v_rcp_f32 and this bound are swept over every float32 by tests/test_gpu_bounds.py).
__device__ __forceinline__ float qs_stage_pf(const double gx, const double gy) {
    return (float)gx * __builtin_amdgcn_rcpf((float)(gx + gy));
}
__device__ __forceinline__ float qs_tf(const float pf) { return -3.0103f * __builtin_amdgcn_logf(pf); }
__device__ __forceinline__ float qs_tf_margin(const float tf) { return tf * 0x1p-19f + 4e-6f; }
// returns false where the float32 value cannot decide (the caller then needs the exact evaluation)
// adj: P.adjust_qs, or 0 where the caller knows it at compile time (the lean builds of k_sample<2>)
// extra: what the caller's p carries beyond qs_stage_pf's own error (the float32 pool loop: VGL_P32_TF_EXTRA)
__device__ __forceinline__ bool qs_decide_pf(const VglDevParams& P, const float pf, int& q, int& aq, const int adj, const float extra = 0.0f) {
    const float tf = qs_tf(pf);
    const float m = qs_tf_margin(tf) + extra;
    const float fl = floorf(tf);
    bool ok = (pf > 1.0e-37f) && (pf < 1.0f) && (tf - fl > m) && (fl + 1.0f - tf > m);
    q = (int)fl; aq = -1;
    if (adj) {
        const float t2 = tf + (float)P.adjust_by;
        const float fl2 = floorf(t2);
        ok = ok && (t2 > m) && (t2 - fl2 > m) && (fl2 + 1.0f - t2 > m);
        aq = (int)fl2;
    }
    return ok;
}
// The same decision in fixed point (round 5: the finish block of the two-byte-item pool loop runs it once per iteration for the few
// lanes that finish a read, so its instruction count is the loop's): tfs = tf x 2^16 (the scaling is exact: one float32 rounding as in
// qs_tf), i = (int)tfs truncated, q = i >> 16, frac = i & 0xFFFF.  Decided iff 0 < tf < 64 and VGL_QS_FIX_M <= frac <= 2^16 - VGL_QS_FIX_M:
// then frac(tf) >= M 2^-16 and 1 - frac(tf) > (M - 1) 2^-16, and (M - 1) 2^-16 = 1.53e-4 exceeds qs_tf_margin(63) + VGL_P32_TF_EXTRA =
// 63 x 2^-19 + 4e-6 + 2.2e-5 = 1.46e-4.  tf >= 64 (p < 4e-7) is left to k_redo.  adj: the adjusted score (int)(tf + adjust_by) likewise.
#define VGL_QS_FIX_M 11
__device__ __forceinline__ bool qs_decide_fix(const VglDevParams& P, const float pf, int& q, int& aq, const int adj) {
    const float tfs = __builtin_amdgcn_logf(pf) * (-3.0103f * 65536.0f);
    const int i = (int)tfs;                                               // (NaN -> 0, +inf -> INT_MAX: both undecided below)
    bool ok = ((uint32_t)i < (64u << 16)) && (((uint32_t)i & 0xFFFFu) - (uint32_t)VGL_QS_FIX_M <= 65536u - 2u * (uint32_t)VGL_QS_FIX_M);
    q = i >> 16; aq = -1;
    if (adj) {
        const int i2 = (int)(tfs + (float)P.adjust_by * 65536.0f);        // (one more float32 rounding of a value below 2^23: within the slack of M)
        ok = ok && ((uint32_t)i2 < (64u << 16)) && (((uint32_t)i2 & 0xFFFFu) - (uint32_t)VGL_QS_FIX_M <= 65536u - 2u * (uint32_t)VGL_QS_FIX_M);
        aq = i2 >> 16;
    }
    return ok;
}
// vcfgl.cpp:500-507, exact
static __device__ void errprob_raw(const VglDevParams& P, const double ep, int& q, int& aq) {
    q = -1; aq = -1;
    if (0.0 == ep) q = CAP_BASEQ;
    else if (1.0 == ep) q = 0;
    else {
        const double tmp = -10.0 * log10(ep);
        q = (int)tmp;
        if (P.adjust_qs) aq = (int)(tmp + P.adjust_by);
    }
}
// vcfgl.cpp:509-523: bins or the cap
__device__ __forceinline__ void qs_finish(const VglDevParams& P, int& q, int& aq, uint32_t* errflag, const bool need) {
    if (P.n_qs_bins != 0) {
        if (need) {
            q = apply_bins(P, q, errflag);
            if (P.adjust_qs) aq = apply_bins(P, aq, errflag);
        }
    } else {
        q = (q > CAP_BASEQ) ? CAP_BASEQ : q;
        if (P.adjust_qs) aq = (aq > CAP_BASEQ) ? CAP_BASEQ : aq;
    }
}

// inclusive prefix sum over the 64 lanes of a wavefront in seven DPP adds (row_shr 1/2/3 of the input, row_shr 4 and 8 of the
// running sums on the upper banks, row_bcast 15 and 31 across the rows) instead of six ds_bpermute round trips
__device__ __forceinline__ uint32_t wave_incl_scan_u32(const uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, true);      // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, true);      // row_shr:4, banks 1-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, true);      // row_shr:8, banks 2-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

// the four 16-bit per-base depths of the wavefront's 64 evaluations (ad4: A | C << 16 | G << 32 | T << 48, each at most read_cap <= 1023, so
// that a field's sum over the wavefront, at most 65 472, stays inside its 16 bits) summed by TWO 32-bit DPP scans (wave_incl_scan_u32:
// seven adds each, the total in lane 63) -- the nine separate ds_bpermute reductions of rounds 1-2 were 160 vector instructions per
// wavefront, a tenth of the fixed-score k_sample.  Wave-uniform results.
__device__ __forceinline__ void wave_sum_ad4(const uint64_t ad4, int out[4]) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32((uint32_t)ad4), 63);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32((uint32_t)(ad4 >> 32)), 63);
    out[0] = (int)(lo & 0xFFFFu); out[1] = (int)(lo >> 16); out[2] = (int)(hi & 0xFFFFu); out[3] = (int)(hi >> 16);
}

// per-site integer totals of the per-base quality sums (-addQS / -addI16): the wavefront's eight sums by DPP scans, one atomic each into
// acc[VGL_ACC_QSUM ..] (k_siteagg: INFO/I16 fields 5-8 are the reference's float32 running sums of these integers over the samples,
// vcfgl.cpp:997-1000,1052-1056 -- exact, hence order-free, while they stay below 2^24).  Lanes without an evaluation pass zeros.
__device__ __forceinline__ void wave_add_qsum_totals(int32_t* acc, const int lane, const uint32_t q[4], const uint32_t qq[4], const bool with_sq) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(q[b]), 63);
        if (lane == 0 && t) atomicAdd((unsigned int*)&acc[VGL_ACC_QSUM + b], t);
    }
    if (with_sq) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(qq[b]), 63);
            if (lane == 0 && t) atomicAdd((unsigned int*)&acc[VGL_ACC_QSUMSQ + b], t);
        }
    }
}

// sample_read_base() on states carried shifted left by 16 bits: 64-bit wraparound is then the generator's mod 2^48 (no masking
// per step), u < 0.5 is the sign bit, floor(4u) the top two bits.  err_thresh16 = err_thresh << 16, saturated (sample_thresh16).
__device__ __forceinline__ uint64_t lcg_next16(const uint64_t x) { return x * VGL_LCG_A + (VGL_LCG_C << 16); }
__device__ __forceinline__ uint64_t sample_thresh16(const uint64_t t) { return t >= (1ULL << 48) ? ~0ULL : (t << 16); }
// HOM (wave-uniform: every evaluation of the wavefront is homozygous): the haplotype stream is not stepped -- both alleles are the same
// base, and in VGL_RNG_TILE the stream is the evaluation's own window, which nobody else reads
template <bool HOM = false>
__device__ __forceinline__ int sample_read_base16(uint64_t& st_hap, uint64_t& st_base, const int a0, const int a1,
                                                  const uint64_t err_thresh16, const bool sample_strand, bool& fwd) {
    if (!HOM) st_hap = lcg_next16(st_hap);
    const int true_base = HOM ? a0 : (((int64_t)st_hap >= 0) ? a0 : a1);
    int r_base = true_base;
    st_base = lcg_next16(st_base);
    if (st_base < err_thresh16) {
        do { st_base = lcg_next16(st_base); r_base = (int)(st_base >> 62); } while (r_base == true_base);
    }
    fwd = true;
    if (sample_strand) { st_base = lcg_next16(st_base); fwd = (int64_t)st_base >= 0; }
    return r_base;
}

// The reads of ONE evaluation with one fixed quality score and no strand draws (vcfgl.cpp:469-613 without :494-531, :581-605), four per
// trip, on states carried shifted by 16 (sample_read_base16).  Per read the common path is the two generator steps, the error compare
// and one bit: the haplotype picks of a trip are collected as four bits and turned into the trip's bases at once (read j shows
// a0 or a1), a base-call error (1 % of the reads) patches its read inside the rare branch, and the per-base depths come from the
// number of a1 picks plus the errors' corrections instead of a 64-bit shift-and-add per read.  HOM (wave-uniform: every evaluation of
// the wavefront is homozygous): the haplotype stream is not even stepped -- both alleles are the same base, and in VGL_RNG_TILE the
// stream is the evaluation's own window, read by nobody else.
// emit(trip, bases): bases = the trip's reads, two bits each (read 4 trip + j at bits 2j); entries beyond dp are a0's bits.
// Returns the per-base depths (A | C << 16 | G << 32 | T << 48).
// (the 64-bit-state form with 16-bit depth fields: evaluations of up to 65535 reads -- what a run with a staging capacity beyond 255 reads uses)
template <bool HOM, class Emit>
__device__ __forceinline__ uint64_t sample_reads_fixed_wide(uint64_t st_hap16, uint64_t st_base16, const int a0, const int a1, const int dp,
                                                       const uint64_t err_thresh16, Emit&& emit) {
    const uint32_t dx = (uint32_t)(a0 ^ a1);
    const uint32_t rep0 = (uint32_t)a0 * 0x55u;                       // a0 in four 2-bit fields
    uint32_t n1 = 0;                                                  // reads that picked the second haplotype
    uint64_t adfix = 0;                                               // the errors' corrections to the per-base depths (modulo 2^64)
    for (int r0 = 0; r0 < dp; r0 += 4) {
        uint32_t hb = 0, fix = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (r0 + j < dp) {
                uint32_t h = 0;
                if (!HOM) {
                    st_hap16 = lcg_next16(st_hap16);
                    h = (uint32_t)(st_hap16 >> 63);                   // u >= 0.5: the second allele (vcfgl.cpp:473)
                    hb |= h << j;
                }
                st_base16 = lcg_next16(st_base16);
                if (st_base16 < err_thresh16) {                       // vcfgl.cpp:486-488
                    const uint32_t tb = HOM ? (uint32_t)a0 : (h ? (uint32_t)a1 : (uint32_t)a0);
                    uint32_t rb;
                    do { st_base16 = lcg_next16(st_base16); rb = (uint32_t)(st_base16 >> 62); } while (rb == tb);
                    fix |= (tb ^ rb) << (2 * j);
                    adfix += (1ULL << (16 * rb)) - (1ULL << (16 * tb));
                }
            }
        }
        uint32_t bases = rep0 ^ fix;
        if (!HOM) {
            uint32_t sp = (hb | (hb << 2)) & 0x33u;                   // bit j -> bit 2j
            sp = (sp | (sp << 1)) & 0x55u;
            bases ^= sp * dx;                                         // dx <= 3: no carries between the fields
            n1 += (uint32_t)__builtin_popcount(hb);
        }
        emit(r0 >> 2, bases);
    }
    return (((uint64_t)((uint32_t)dp - n1)) << (16 * (a0 & 3))) + (((uint64_t)n1) << (16 * (a1 & 3))) + adfix;   // (dp = 0 for a missing genotype, alleles 0xF)
}
// Round 6: the same reads on the raw 52-bit form of the two states (x 16, as lcg52_step carries them: the low product and the constant in one
// v_mad_u64_u32, the two cross terms through the 24-bit multiply-add -- three instructions a step instead of four, and no 64-bit pair to shuffle).
// Of the high word only bits 0-19 are valid: the haplotype pick is bit 19 (u >= 0.5), a wrong base bits 18-19 (floor(4 u)), and the base-call error test
// u < e is taken in two steps -- high 20 bits <= the threshold's (one 32-bit compare on the common path; true for e + 2^-20 of the reads), then the exact
// 52-bit comparison inside the rare block.  The errors' corrections to the per-base depths are kept in four 8-bit fields of one register (modulo 2^32;
// every final depth is below 256: the caller's read_cap <= 255, else sample_reads_fixed_wide) and spread to 16-bit fields once at the end.
template <bool HOM, class Emit>
__device__ __forceinline__ uint64_t sample_reads_fixed(const uint64_t st_hap16, const uint64_t st_base16, const int a0, const int a1, const int dp,
                                                       const uint64_t err_thresh16, Emit&& emit) {
    uint32_t hl = (uint32_t)(st_hap16 >> 12), hh = (uint32_t)(st_hap16 >> 44);          // state x 16 = (state x 2^16) >> 12
    uint32_t bl = (uint32_t)(st_base16 >> 12), bh = (uint32_t)(st_base16 >> 44);
    // threshold x 16; sample_thresh16 saturates e >= 1 to 2^64 - 1, whose top 52 bits compare above every state as well
    const uint32_t t_lo = (uint32_t)(err_thresh16 >> 12), t_hi = (uint32_t)(err_thresh16 >> 44);
    const uint32_t dx = (uint32_t)(a0 ^ a1);
    const uint32_t rep0 = (uint32_t)a0 * 0x55u;                       // a0 in four 2-bit fields
    uint32_t n1 = 0;                                                  // reads that picked the second haplotype
    uint32_t adfix8 = 0;                                              // the errors' corrections to the per-base depths, 8 bits per base (modulo 2^32)
    for (int r0 = 0; r0 < dp; r0 += 4) {
        uint32_t hb = 0, fix = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (r0 + j < dp) {
                if (!HOM) {
                    lcg52_step(hl, hh, hl, hh);
                    hb = ((hh >> (19 - j)) & (1u << j)) | hb;             // u >= 0.5: the second allele (vcfgl.cpp:473)
                }
                lcg52_step(bl, bh, bl, bh);
                if (__builtin_expect((bh & 0xFFFFFu) <= t_hi, 0)) {       // may be a base-call error (vcfgl.cpp:486-488): settle it on all 52 bits
                    if ((bh & 0xFFFFFu) < t_hi || bl < t_lo) {
                        const uint32_t tb = HOM ? (uint32_t)a0 : (((hb >> j) & 1u) ? (uint32_t)a1 : (uint32_t)a0);
                        uint32_t rb;
                        do { lcg52_step(bl, bh, bl, bh); rb = (bh >> 18) & 3u; } while (rb == tb);
                        fix |= (tb ^ rb) << (2 * j);
                        adfix8 += (1u << (8 * rb)) - (1u << (8 * tb));
                    }
                }
            }
        }
        uint32_t bases = rep0 ^ fix;
        if (!HOM) {
            uint32_t sp = (hb | (hb << 2)) & 0x33u;                   // bit j -> bit 2j
            sp = (sp | (sp << 1)) & 0x55u;
            bases ^= sp * dx;                                         // dx <= 3: no carries between the fields
            n1 += (uint32_t)__builtin_popcount(hb);
        }
        emit(r0 >> 2, bases);
    }
    // per-base depths in 8-bit fields: picks + corrections (each final field is the count of reads that show the base: 0 ... dp < 256), then 16 bits a field
    const uint32_t d8 = (((uint32_t)dp - n1) << (8 * (a0 & 3))) + (n1 << (8 * (a1 & 3))) + adfix8;      // (dp = 0 for a missing genotype, alleles 0xF)
    return (uint64_t)(d8 & 0xFFu) | ((uint64_t)(d8 & 0xFF00u) << 8) | ((uint64_t)(d8 & 0xFF0000u) << 16) | ((uint64_t)(d8 & 0xFF000000u) << 24);
}
// the trip's staged word, one byte per read (score << 2 | base), from its 2-bit bases; qrep = (score << 2) in every byte
__device__ __forceinline__ uint32_t staged_word_of(const uint32_t bases, const uint32_t qrep) {
    uint32_t w = (bases | (bases << 12)) & 0x000F000Fu;               // fields 0,1 | fields 2,3
    w = (w | (w << 6)) & 0x03030303u;                                 // field j at byte j
    return w | qrep;
}

// one read: haplotype pick, base-call error, strand (vcfgl.cpp:473,486-488,581-586); all compares
// are exact integer restatements on the 48-bit state: u<0.5 <=> X<2^47, u<e <=> X<ceil(e 2^48),
// floor(4u) = X>>46
__device__ __forceinline__ int sample_read_base(uint64_t& st_hap, uint64_t& st_base, const int a0, const int a1,
                                                const uint64_t err_thresh, const bool sample_strand, bool& fwd) {
    st_hap = lcg_next(st_hap);
    const int true_base = (st_hap < (1ULL << 47)) ? a0 : a1;
    int r_base = true_base;
    st_base = lcg_next(st_base);
    if (st_base < err_thresh) {
        do { st_base = lcg_next(st_base); r_base = (int)(st_base >> 46); } while (r_base == true_base);
    }
    fwd = true;
    if (sample_strand) { st_base = lcg_next(st_base); fwd = st_base < (1ULL << 47); }
    return r_base;
}

