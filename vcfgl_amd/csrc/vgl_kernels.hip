// vgl_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the vcfgl genotype-likelihood
// simulation hot path.  Written for wave64 only; no other target is supported.
//
// Work decomposition: one LANE owns one evaluation = one (site, sample); one WAVEFRONT
// owns 64 consecutive samples of one site, so every per-evaluation global access of a
// wave is one contiguous segment (structure-of-arrays tiles, sample index fastest).
//
//   k_sample<EQS>    Poisson depth -> per-read haplotype / base-call error / strand draws ->
//                    per-evaluation ACGT depth; reads staged as 1 byte each in
//                    [read][site][sample] planes; per-site depth sums by wave reduction + one
//                    integer atomic per wave and counter.  EQS=2 (a beta deviate per read):
//                    the quality-score sampling of the wave's reads is an LDS-staged pool dealt
//                    round-robin to the lanes, each lane a select-only state machine over
//                    normal-deviate attempts.            (vcfgl.cpp:364-389, 441-640; rng.h)
//   k_scout, k_sample_serial   VGL_RNG_SERIAL: sequential stream-state scout + parallel
//                    evaluation from the recorded states (reference draw order, bit for bit).
//   k_site           per-site allele order / status.     (vcfgl.cpp:396-404, 665-766)
//   k_gl<A>          genotype likelihoods from the staged reads in the site's allele order
//                    (lanes re-dealt in depth order inside each workgroup), accumulators in
//                    VGPRs, then PL / GP / AD epilogue.   (gl_methods.cpp:4-369, vcfgl.cpp:806-970)
//   k_siteagg        order-dependent per-site float sums (QS, I16).  (vcfgl.cpp:845-898, 982-1074)
//
// The float32 accumulation order of the reference is kept exactly (double add rounded to
// float per genotype per read, float max, float subtract): build with -ffp-contract=off.
// Decisions that the reference takes on double transcendentals (rejection tests, floor of a
// scaled tan, (int)(-10 log10 p)) are taken from float32 hardware transcendentals with measured
// error bounds and fall back to the exact double expression inside the error band, behind
// wave-uniform branches: results equal the exact evaluation, the common path has no f64 log/tan/exp.
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
__device__ __forceinline__ double u01(uint64_t x) {
    // exact for x < 2^48: the 48 bits become the top of the mantissa of 1.xxx (what glibc's erand48 does)
    return __longlong_as_double((long long)(0x3FF0000000000000ULL | (x << 4))) - 1.0;
}
__device__ __forceinline__ double next_u(uint64_t& st) { st = lcg_next(st); return u01(st); }

// gamma_ln, rng.h:38-43,60-64
__device__ double gamma_ln_dev(const double xx) {
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
__device__ int poisson_draw(const VglPois& p, uint64_t& st, const double* __restrict__ glt, const int glt_n) {
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
// one rejection attempt (rng.h:302-309) from the two generator states it would consume: `neg`: em < 0
// (only st1 is consumed, no acceptance draw), else `rej` = the acceptance draw u(st2) exceeds t.
__device__ __forceinline__ void poisson_attempt(const VglPois& p, const uint64_t st1, const uint64_t st2, const bool need,
                                                const double* __restrict__ glt, const int glt_n, bool& neg, bool& rej, double& em) {
    const double a = VGL_PI * u01(st1);
    const float af = (float)a;
    const float yf = tanf(af);
    const float y2 = yf * yf;
    const float dy = fabsf(yf) * 0x1p-21f + (1.0f + y2) * af * 0x1p-23f;      // |yf - tan(a)|
    const double e0 = p.sq * (double)yf + p.lm;
    const double derr = p.sq * (double)dy + 1e-9;
    em = floor(e0);
    const bool amb_em = (e0 - em < derr) | (em + 1.0 - e0 < derr) | !(fabs(e0) < 1.0e6);
    neg = e0 < 0.0;
    const bool in_tab = (em >= 0.0) & (em < (double)(glt_n - 1));
    const double gl = glt[in_tab ? (int)em + 1 : 1];
    const double z = em * p.alxm - gl - p.g;
    const float ex = __builtin_amdgcn_exp2f((float)(z * 1.4426950408889634));
    const float tt = 0.9f * (1.0f + y2) * ex;
    const float rel_t = 2.0f * fabsf(yf) * dy / (1.0f + y2) + fabsf((float)z) * 0x1p-22f + 0x1p-19f;
    const double u2 = u01(st2);
    rej = u2 > (double)tt;
    const bool amb_t = fabs(u2 - (double)tt) <= (double)(tt * rel_t) + 1e-30;
    const bool amb = need & (amb_em | (!neg & (amb_t | !in_tab)));
    if (__builtin_expect(__ballot(amb) != 0, 0)) {               // exact evaluation (rare, wave-uniform branch)
        asm volatile("" ::: "memory");
        const double y = tan(a);
        double eme = p.sq * y + p.lm;
        const bool nege = eme < 0.0;
        eme = floor(eme);
        const double t = 0.9 * (1.0 + y * y) * exp(eme * p.alxm - ((eme >= 0.0 && eme < (double)(glt_n - 1)) ? glt[(int)eme + 1] : gamma_ln_dev(eme + 1.0)) - p.g);
        const bool reje = u2 > t;
        neg = amb ? nege : neg; em = amb ? eme : em; rej = amb ? reje : rej;
    }
}

__device__ int poisson_draw_fast(const VglPois& p, uint64_t& st, const double* __restrict__ glt, const int glt_n) {
    if (p.st12) {
        double em = -1.0, t = 1.0;
        do { ++em; t *= next_u(st); } while (t > p.g);
        return (int)em;
    }
    bool done = false;
    double em_res = 0.0;
    while (__ballot(!done)) {
        const uint64_t st1 = lcg_next(st);
        const uint64_t st2 = lcg_next(st1);
        bool neg, reject; double em;
        poisson_attempt(p, st1, st2, !done, glt, glt_n, neg, reject, em);
        const bool acc = !done & !neg & !reject;
        st = done ? st : (neg ? st1 : st2);                          // em < 0 consumes one draw, an attempt two
        em_res = acc ? em : em_res;
        done = done | acc;
    }
    return (int)em_res;
}

// sample_NormalSampler_0_1_0, rng.h:70-80
__device__ double normal_rou(uint64_t& st) {
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
__device__ double gamma1_draw(const VglGamma1& g, uint64_t& st) {
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
__device__ int apply_bins(const VglDevParams& P, int q, uint32_t* errflag) {
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

// error probability -> qScore / adjusted qScore, vcfgl.cpp:500-523
__device__ void errprob_to_qs(const VglDevParams& P, double ep, int& q, int& aq, uint32_t* errflag) {
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

// wave -> (local site, 64-sample chunk); everything here is wave-uniform (SGPRs)
struct WavePos { int ls; int chunk; int wib; bool valid; };
__device__ __forceinline__ WavePos wave_pos(const VglDevParams& P, const VglTilePtrs& T) {
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    WavePos r;
    r.wib = wib;
    r.valid = w < (int64_t)T.n_sites * P.chunks;
    r.ls = (int)(w / P.chunks);
    r.chunk = (int)(w - (int64_t)r.ls * P.chunks);
    return r;
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
__device__ __forceinline__ bool normal_slow_test(const double v, const double u, const bool need) {
    const double lhs = v * v;
    const float uf = (float)u;
    const double l = fast_ln(uf);
    const double uu = u * u;
    const double rhs = -4.0 * l * uu;
    const double m = 4.0 * uu * fast_ln_err(l);
    const bool hi = lhs > rhs + m, lo = lhs < rhs - m;
    bool res = hi;
    const bool amb = need && !((hi || lo) && (uf > 0.0f));
    if (__builtin_expect(__ballot(amb) != 0, 0)) {          // wave-uniform and rare: a real branch
        asm volatile("" ::: "memory");                      // (keeps the compiler from speculating the double log)
        const bool ex = lhs > -4.0 * log(u) * (u * u);
        res = amb ? ex : res;
    }
    return res;
}

// log(u) > 0.5*xsq + a1*(1.0 - v + log(v)),  v = fl(fl(w*w)*w), w = fl(1 + a2 x)    rng.h:139-145
// With s = a2 x and a2^2 = 1/(9 a1) the quadratic terms cancel analytically:
//     0.5 x^2 + a1 (1 - (1+s)^3 + 3 ln(1+s)) = -3 a1 s^4 (1/4 - s/5 + s^2/6 - s^3/7 + ...)
// a well-conditioned series (no cancellation), so float32 is enough for the bounded decision:
// |s| <= 1/3 => 11 terms leave < 2e-7 relative truncation; float32 evaluation < 1e-6 relative;
// the reference's own double rounding of its expression is < 1e-12.  Outside the band (or for
// |s| > 1/3) the exact double expression is evaluated.
__device__ __forceinline__ bool gamma_slow_test(const double u, const double xsq, const double a1, const double v,
                                                const double s, const bool need) {
    const float uf = (float)u, sf = (float)s, a1f = (float)a1;
    const float lu = __builtin_amdgcn_logf(uf) * 0.69314718f;
    float p = 1.0f / 14.0f;
    p = 1.0f / 13.0f - sf * p; p = 1.0f / 12.0f - sf * p; p = 1.0f / 11.0f - sf * p; p = 1.0f / 10.0f - sf * p;
    p = 1.0f / 9.0f - sf * p; p = 1.0f / 8.0f - sf * p; p = 1.0f / 7.0f - sf * p; p = 1.0f / 6.0f - sf * p;
    p = 1.0f / 5.0f - sf * p; p = 1.0f / 4.0f - sf * p;
    const float s2 = sf * sf;
    const float g = 3.0f * a1f * (s2 * s2) * p;                 // = -(rhs of the reference), >= 0
    const float d = lu + g;                                     // log(u) - rhs
    const float m = fabsf(lu) * 0x1p-20f + 0x1p-21f + g * 4e-6f + 1e-10f;
    const bool ok = (fabsf(d) > m) && (fabsf(sf) <= 0.3333f) && (uf > 0.0f);
    bool res = d > 0.0f;
    const bool amb = need && !ok;
    if (__builtin_expect(__ballot(amb) != 0, 0)) {          // wave-uniform and rare: a real branch
        asm volatile("" ::: "memory");
        const bool ex = log(u) > 0.5 * xsq + a1 * (1.0 - v + log(v));
        res = amb ? ex : res;
    }
    return res;
}

// error probability -> qScore / adjusted qScore (vcfgl.cpp:500-523); (int)(-10*log10(p)) is
// taken from the float32 log2 unless p sits within its error bound of an integer boundary
// p = gx / (gx + gy) (rng.h:438) is itself taken in float32 (v_rcp_f32, 1 ulp): relative error of pf
// <= 2^-24 (gx) + 2^-24 (sum) + 2^-23 (rcp) + 2^-24 (product) < 2^-22, i.e. < 1.1e-6 in tf.
__device__ __forceinline__ void errprob_to_qs_fast(const VglDevParams& P, const double gx, const double gy, int& q, int& aq, uint32_t* errflag, const bool need) {
    // float32: tf = -10 log10(p) within |tf| 2^-20 + 2.2e-6 (v_log_f32 bound + argument error)
    const float pf = (float)gx * __builtin_amdgcn_rcpf((float)(gx + gy));
    const float tf = -3.0103f * __builtin_amdgcn_logf(pf);
    const float m = tf * 0x1p-19f + 4e-6f;
    const float fl = floorf(tf);
    const float t2 = tf + (float)P.adjust_by;
    const float fl2 = floorf(t2);
    bool ok = (pf > 1.0e-37f) && (pf < 1.0f) && (tf - fl > m) && (fl + 1.0f - tf > m);
    if (P.adjust_qs) ok = ok && (t2 > m) && (t2 - fl2 > m) && (fl2 + 1.0f - t2 > m);
    q = (int)fl; aq = P.adjust_qs ? (int)fl2 : -1;
    ok = ok || !need;
    if (__builtin_expect(__ballot(!ok) != 0, 0)) {     // exact: vcfgl.cpp:500-507 (rare: a real branch)
        asm volatile("" ::: "memory");
        const double ep = gx / (gx + gy);
        int qe = -1, aqe = -1;
        if (0.0 == ep) qe = CAP_BASEQ;
        else if (1.0 == ep) qe = 0;
        else {
            const double tmp = -10.0 * log10(ep);
            qe = (int)tmp;
            if (P.adjust_qs) aqe = (int)(tmp + P.adjust_by);
        }
        q = ok ? q : qe; aq = ok ? aq : aqe;
    }
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

// ------------------------------------------------------------------------------------
// EQS = --error-qs.  EQS 2 (a beta deviate per read) runs the quality-score sampling as a
// wavefront-wide pool: the reads of the wave's 64 evaluations are independent work items
// (stream 3 is addressed per read), staged in LDS and dealt round-robin to the lanes, so a
// lane's work does not depend on its own evaluation's depth; each lane runs the nested
// rejection loops of the gamma sampler as one flat state machine (one normal-deviate attempt
// per iteration) so that lanes at different stages share every iteration.
// DBG: diagnostic instantiation (VGL_DEBUG_STAMPS / VGL_DEBUG_PHASE), never used in a timed run
template <int EQS, bool DBG>
__global__ __launch_bounds__(256) void k_sample(const VglDevParams P, const VglTilePtrs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const WavePos wp = wave_pos(P, T);
    if (!wp.valid) return;
    const int lane = threadIdx.x & 63;
    const int N = P.n_samples;
    const int ls = wp.ls;
    const int s = wp.chunk * 64 + lane;
    const bool active = s < N;
    const size_t ev0 = (size_t)ls * N + (size_t)wp.chunk * 64;      // evaluation of lane 0
    const size_t ev = ev0 + (active ? lane : 0);
    const size_t plane = (size_t)T.n_sites * N;

    int dp = 0, a0 = 0, a1 = 0;
    uint64_t ad4 = 0, adf4 = 0;
    uint32_t qs0 = 0, qs1 = 0, qs2 = 0, qs3 = 0, qq0 = 0, qq1 = 0, qq2 = 0, qq3 = 0;
    uint64_t st_hap = 0, st_base = 0, st_qs = 0;
    // DBG only: per-phase cycle stamps
    unsigned long long c_t0 = 0, c_pois = 0, c_owner = 0, c_pool = 0, c_flush = 0, c_iter = 0, c_items = 0, c_tmp = 0;
    if (DBG) c_t0 = clock64();
    uint64_t err_thresh = P.err_thresh;

    // ---- stream states of this evaluation: J^(off_k) . J^(block*s) . J^(block*N*site) (x0)
    const uint64_t site_abs = (uint64_t)(T.site0 + ls);
    uint64_t xb = P.x0;
#pragma unroll 1
    for (int b = 0; b < 40; ++b)
        if ((site_abs >> b) & 1) xb = aff(P.site_pow[b], xb);

    if (active) {
        const VglAffine ms = P.samp_tab[s];
        const uint64_t xe = aff(ms, xb);
        uint64_t st_depth = aff(P.off[0], xe);
        st_hap = aff(P.off[1], xe);
        st_base = aff(P.off[2], xe);
        st_qs = aff(P.off[3], xe);

        // ---- depth (vcfgl.cpp:364-389): drawn even when the genotype is missing
        int n;
        if (P.per_sample_depth) { const VglPois pc = P.pois[s]; n = poisson_draw_fast(pc, st_depth, P.gamma_ln_tab, P.gamma_ln_n); }
        else n = poisson_draw_fast(P.pois0, st_depth, P.gamma_ln_tab, P.gamma_ln_n);
        const uint32_t g = T.gt[ev];
        a0 = g & 0xF; a1 = (g >> 4) & 0xF;
        dp = (a0 == 0xF || a1 == 0xF) ? 0 : n;
        if (dp > P.read_cap) { atomicOr(T.errflag, VGL_DEVERR_CAPACITY); dp = P.read_cap; }
    }

    if (DBG) c_pois = clock64() - c_t0;
    if (DBG && P.dbg_phase == 1) return;
    if (EQS == 1) {
        // one beta deviate per site: stream 3 of sample 0, read 0 (vcfgl.cpp:425-437); lane 0 draws it
        uint32_t lo = 0, hi = 0;
        if (lane == 0) {
            uint64_t st_site = aff(P.off[3], aff(P.samp_tab[0], xb));
            const double pe = beta_draw(P, st_site);
            const uint64_t th = (uint64_t)ceil(ldexp(pe, 48));
            lo = (uint32_t)th; hi = (uint32_t)(th >> 32);
        }
        lo = __shfl(lo, 0, 64); hi = __shfl(hi, 0, 64);
        err_thresh = ((uint64_t)hi << 32) | lo;
    }

    if (EQS != 2) {
        // ---- read loop (vcfgl.cpp:469-613), fixed quality score
        const int q_i = P.pre_q, aq_i = P.pre_adjq;
        const uint32_t q_gl = (uint32_t)((P.adjust_qs & 1) ? aq_i : q_i);
        const uint32_t qq = (uint32_t)((P.adjust_qs & 2) ? aq_i : q_i);
        const uint32_t q2 = (uint32_t)qs_to_qssq((int)qq);
        const bool stage = (P.gl_model != 1);      // GL model 1 with one fixed qScore needs only the per-base depths
        for (int r = 0; r < dp; ++r) {
            bool fwd;
            const int r_base = sample_read_base(st_hap, st_base, a0, a1, err_thresh, P.sample_strand != 0, fwd);
            if (stage) T.reads[(size_t)r * plane + ev] = (uint8_t)((q_gl << 2) | r_base);
            if (T.reads_out && r < T.reads_out_cap) T.reads_out[(size_t)r * plane + ev] = (uint8_t)((q_i << 2) | r_base);
            const uint64_t one = 1ULL << (16 * r_base);
            ad4 += one;
            if (fwd) adf4 += one;
        }
        if (P.need_qsum) {
            qs0 = qq * (uint32_t)(ad4 & 0xFFFF); qs1 = qq * (uint32_t)((ad4 >> 16) & 0xFFFF);
            qs2 = qq * (uint32_t)((ad4 >> 32) & 0xFFFF); qs3 = qq * (uint32_t)((ad4 >> 48) & 0xFFFF);
            qq0 = q2 * (uint32_t)(ad4 & 0xFFFF); qq1 = q2 * (uint32_t)((ad4 >> 16) & 0xFFFF);
            qq2 = q2 * (uint32_t)((ad4 >> 32) & 0xFFFF); qq3 = q2 * (uint32_t)((ad4 >> 48) & 0xFFFF);
        }
    } else {
        // ---- LDS of this wave: [64] u64 qscore-stream bases | [cap] u16 item->(read,owner) |
        //      [cap] u8 base | [cap] u8 qScore | [cap] u8 adjusted qScore
        const int cap = P.pool_cap;
        uint8_t* wl = lds_raw + (size_t)wp.wib * P.pool_lds_bytes;
        uint64_t* l_stq = (uint64_t*)wl;
        uint16_t* l_map = (uint16_t*)(wl + 512);
        uint8_t* l_pb = wl + 512 + 2 * (size_t)cap;
        uint8_t* l_pq = l_pb + cap;
        uint8_t* l_paq = l_pq + cap;

        // exclusive prefix sum of the depths = first pool index of each owner
        int incl = dp;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        const int offs = incl - dp;
        const int total = __shfl(incl, 63, 64);
        l_stq[lane] = st_qs;
        int rdone = 0;
        // Kernel arguments arrive in 16-dword scalar tuples that the register allocator spills and
        // reloads as a whole (v_readlane, VALU work) inside the loops below; the three values the flush
        // loop needs are therefore pinned to vector registers.
        uint8_t* reads_v = T.reads;
        uint8_t* reads_out_v = T.reads_out;
        int reads_out_cap_v = T.reads_out ? T.reads_out_cap : 0;
        asm volatile("" : "+v"(reads_v), "+v"(reads_out_v), "+v"(reads_out_cap_v));

        for (int seg0 = 0; seg0 < total; seg0 += cap) {                 // normally one segment
            const int segT = (total - seg0 < cap) ? (total - seg0) : cap;
            // -- owners: bases of their reads that fall into this segment
            if (DBG) c_tmp = clock64();
            int r_end = seg0 + segT - offs; r_end = r_end > dp ? dp : r_end; r_end = r_end < rdone ? rdone : r_end;
            for (int r = rdone; r < r_end; ++r) {
                bool fwd;
                const int r_base = sample_read_base(st_hap, st_base, a0, a1, err_thresh, P.sample_strand != 0, fwd);
                const uint64_t one = 1ULL << (16 * r_base);
                ad4 += one;
                if (fwd) adf4 += one;
                const int k = offs + r - seg0;
                l_map[k] = (uint16_t)((r << 6) | lane);
                l_pb[k] = (uint8_t)r_base;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // -- pool: item k -> lane k % 64.  One iteration = one normal-deviate attempt
            //    (rng.h:72-78) + the gamma step it feeds (rng.h:139-145), computed for every lane
            //    without branches; lane state (stream, stage, first gamma) advances by selects, so
            //    lanes in different stages of different reads share every instruction.  Real
            //    branches remain only around the rare bounded-log tests and the per-read epilogue.
            if (DBG) { const unsigned long long c = clock64(); c_owner += c - c_tmp; c_tmp = c; c_items += segT; }
            if (DBG && P.dbg_phase == 2) return;
            {
                // Items are dealt dynamically: a lane owns its current item k and an item kn claimed one
                // item ahead (so that kn's operands are in flight while k is worked on); a lane that
                // finishes an item adopts kn and claims the next unclaimed one.  Each item's stream is
                // addressed by its read, so the result does not depend on who works on it.
                int k = lane, kn = lane + 64;
                int next_free = 128;                         // wave-uniform: first unclaimed item
                bool have = k < segT;
                bool stage1 = false;                         // false: first gamma deviate (x), true: second (y)
                uint64_t st = 0; double gx = 0.0; int it_o = 0, it_r = 0;
                if (have) { const uint32_t m = l_map[k]; it_o = m & 63; it_r = m >> 6; st = aff(P.qs_read_tab[it_r], l_stq[it_o]); }
                const bool any_changed = (P.gx.changed | P.gy.changed) != 0;
                // The bounded-log tests are needed by a few lanes per iteration but cost every lane of the
                // wave; they run only every P.slow_period-th iteration.  In between, a lane that needs one
                // holds: its state is left untouched, so the later iteration recomputes the same attempt.
                int slow_cnt = P.slow_period;
                while (__ballot(have)) {
                    if (DBG) c_iter++;
                    const bool full = (--slow_cnt == 0);
                    if (full) slow_cnt = P.slow_period;
                    // operands of this lane's next item, fetched at the top of the iteration and consumed at
                    // the bottom (unconditional, clamped index: no divergent control flow in the loop)
                    const bool hn = kn < segT;
                    const uint32_t m_n = l_map[hn ? kn : 0];
                    const int o_n = m_n & 63, r_n = m_n >> 6;
                    const VglAffine tab_n = P.qs_read_tab[r_n];
                    const uint64_t base_n = l_stq[o_n];

                    const double ga1 = stage1 ? P.gy.a1 : P.gx.a1;
                    const double ga2 = stage1 ? P.gy.a2 : P.gx.a2;
                    // normal attempt
                    const uint64_t st1 = lcg_next(st);
                    const uint64_t st2 = lcg_next(st1);
                    const uint64_t st3 = lcg_next(st2);
                    const double u = u01(st1);
                    const double v = 1.7156 * (u01(st2) - 0.5);
                    const double x = u - 0.449871;
                    const double y = fabs(v) + 0.386595;
                    const double q = (x * x) + y * (0.19600 * y - 0.25472 * x);
                    const bool q_lo = q > 0.27597, q_hi = q > 0.27846;
                    bool slow_n = false;
                    const bool n_amb = have && q_lo && !q_hi;
                    bool hold = n_amb && !full;
                    if (full && __ballot(n_amb)) slow_n = normal_slow_test(v, u, n_amb);
                    const bool acc_n = !(q_lo && (q_hi || slow_n));
                    // gamma step on the accepted deviate
                    const double xn = div_inrange(v, u);
                    const double w = 1.0 + ga2 * xn;
                    const bool w_pos = w > 0.0;
                    const double vv = w * w * w;
                    const double u2 = u01(st3);
                    const double xsq = xn * xn;
                    const bool sq_fail = u2 > 1.0 - 0.0331 * (xsq * xsq);
                    const bool g_try = have && acc_n && w_pos && !hold;
                    const bool g_amb = g_try && sq_fail;
                    hold = hold || (g_amb && !full);
                    bool slow_g = false;
                    if (full && __ballot(g_amb)) slow_g = gamma_slow_test(u2, xsq, ga1, vv, ga2 * xn, g_amb);
                    const bool acc_g = g_try && !(sq_fail && slow_g) && !hold;
                    st = hold ? st : (g_try ? st3 : st2);    // u2 is drawn only when w > 0 (rng.h:140-142)
                    double val = ga1 * vv;
                    if (any_changed) {                       // alpha < 1 (rng.h:146-148); wave-uniform guard
                        if (acc_g && (stage1 ? P.gy.changed : P.gx.changed)) {
                            double u3;
                            do { st = lcg_next(st); u3 = u01(st); } while (u3 == 0.0);
                            val = pow(u3, 1.0 / (stage1 ? P.gy.alpha0 : P.gx.alpha0)) * ga1 * vv;
                        }
                    }
                    const bool fin = acc_g && stage1;
                    // per-read epilogue, computed for every lane, committed where fin (rng.h:438, vcfgl.cpp:500-531)
                    int q_i, aq_i;
                    errprob_to_qs_fast(P, gx, val, q_i, aq_i, T.errflag, fin);
                    if (fin) {
                        l_pq[k] = (uint8_t)q_i;
                        l_paq[k] = (uint8_t)aq_i;
                    }
                    if (P.precise_gl) { if (fin) T.errp[(size_t)it_r * plane + ev0 + it_o] = gx / (gx + val); }
                    gx = (acc_g && !stage1) ? val : gx;
                    stage1 = stage1 != acc_g;
                    // a lane that finished its item adopts kn and claims the next unclaimed item
                    const uint64_t fin_m = __ballot(fin);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fin_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fin_m, 0u));
                    const uint64_t st_n = aff(tab_n, base_n);
                    st = fin ? st_n : st;
                    it_o = fin ? o_n : it_o; it_r = fin ? r_n : it_r;
                    k = fin ? kn : k;
                    kn = fin ? next_free + rank : kn;
                    next_free += __popcll(fin_m);
                    have = have && (!fin || hn);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

            // -- owners: combine base + quality score, stage the read, quality sums (vcfgl.cpp:525-564)
            if (DBG) { const unsigned long long c = clock64(); c_pool += c - c_tmp; c_tmp = c; }
            if (DBG && P.dbg_phase == 3) return;
            for (int r = rdone; r < r_end; ++r) {
                const int k = offs + r - seg0;
                const int r_base = l_pb[k];
                const int q_i = l_pq[k];
                const int aq_i = P.adjust_qs ? (int)l_paq[k] : -1;
                const int q_gl = (P.adjust_qs & 1) ? aq_i : q_i;
                reads_v[(size_t)r * plane + ev] = (uint8_t)((q_gl << 2) | r_base);
                if (r < reads_out_cap_v) reads_out_v[(size_t)r * plane + ev] = (uint8_t)((q_i << 2) | r_base);
                if (P.need_qsum) {
                    const uint32_t qq = (uint32_t)((P.adjust_qs & 2) ? aq_i : q_i);
                    const uint32_t q2 = (uint32_t)qs_to_qssq((int)qq);
                    qs0 += (r_base == 0) ? qq : 0u; qs1 += (r_base == 1) ? qq : 0u;
                    qs2 += (r_base == 2) ? qq : 0u; qs3 += (r_base == 3) ? qq : 0u;
                    qq0 += (r_base == 0) ? q2 : 0u; qq1 += (r_base == 1) ? q2 : 0u;
                    qq2 += (r_base == 2) ? q2 : 0u; qq3 += (r_base == 3) ? q2 : 0u;
                }
            }
            rdone = r_end;
            __builtin_amdgcn_wave_barrier();
            if (DBG) c_flush += clock64() - c_tmp;
            if (DBG && P.dbg_phase == 4) return;
        }
    }

    if (active) {
        if (!P.sample_strand) adf4 = ad4;
        if (T.fmt_dp) T.fmt_dp[ev] = dp;
        T.ad4[ev] = ad4;
        if (P.need_adf) T.adf4[ev] = adf4;
        if (P.need_qsum) {
            uint32_t* q = T.qsum + (size_t)ls * 4 * N + s;
            q[0] = qs0; q[(size_t)N] = qs1; q[(size_t)2 * N] = qs2; q[(size_t)3 * N] = qs3;
            if (P.need_qsumsq) {
                uint32_t* qq = T.qsumsq + (size_t)ls * 4 * N + s;
                qq[0] = qq0; qq[(size_t)N] = qq1; qq[(size_t)2 * N] = qq2; qq[(size_t)3 * N] = qq3;
            }
        }
        if (T.reads_out) for (int r = dp; r < T.reads_out_cap; ++r) T.reads_out[(size_t)r * plane + ev] = 0xFF;
    }

    // ---- per-site sums: wave reduction, one atomic per wave and counter
    int v[9];
    v[0] = dp;
#pragma unroll
    for (int b = 0; b < 4; ++b) { v[1 + b] = (int)((ad4 >> (16 * b)) & 0xFFFF); v[5 + b] = (int)((adf4 >> (16 * b)) & 0xFFFF); }
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
        int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) if (v[k]) atomicAdd(&acc[k], v[k]);
        if (DBG) {
            atomicAdd(&T.dbg[0], 1ULL); atomicAdd(&T.dbg[1], clock64() - c_t0); atomicAdd(&T.dbg[2], c_pois);
            atomicAdd(&T.dbg[3], c_owner); atomicAdd(&T.dbg[4], c_pool); atomicAdd(&T.dbg[5], c_flush);
            atomicAdd(&T.dbg[6], c_iter); atomicAdd(&T.dbg[7], c_items);
        }
    }
}

// ------------------------------------------------------------------------------------
// VGL_RNG_SERIAL: the reference's own stream order.  The three rand48 streams
// and the mt19937 of the default beta sampler are consumed in a data-dependent serial order, so a
// sequential scout (one lane) walks the tile once and records, per evaluation, the state of each
// stream at the point the reference reaches it; the per-read beta deviates (one global mt19937 /
// rng2 stream) are recorded too.  All the remaining work (Poisson evaluation, reads, likelihoods)
// then runs in parallel from those recorded states and reproduces the serial program exactly.

// std::mt19937 (libstdc++) -- default BetaSampler, rng.h:353-421
__device__ uint32_t mt_next(VglSerialState* S) {
    if (S->mt_idx >= 624) {
        for (int i = 0; i < 624; i++) {
            const uint32_t y = (S->mt[i] & 0x80000000u) | (S->mt[(i + 1) % 624] & 0x7fffffffu);
            S->mt[i] = S->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        S->mt_idx = 0;
    }
    uint32_t y = S->mt[S->mt_idx++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
// std::generate_canonical<double,53>
__device__ double mt_canonical(VglSerialState* S) {
    double sum = 0.0, tmp = 1.0;
    sum += (double)mt_next(S) * tmp; tmp *= 4294967296.0;
    sum += (double)mt_next(S) * tmp; tmp *= 4294967296.0;
    double r = sum / tmp;
    if (r >= 1.0) r = 0x1.fffffffffffffp-1;
    return r;
}
// std::gamma_distribution<double>(alpha,1)(gen) on a fresh distribution object (rng.h:409-412)
__device__ double std_gamma_fresh(VglSerialState* S, const double alpha) {
    const double malpha = alpha < 1.0 ? alpha + 1.0 : alpha;
    const double a1 = malpha - 1.0 / 3.0;
    const double a2 = 1.0 / sqrt(9.0 * a1);
    bool saved_avail = false; double saved = 0.0;
    double u, v, n;
    do {
        do {
            if (saved_avail) { saved_avail = false; n = saved; }
            else {
                double x, y, r2;
                do {
                    x = 2.0 * mt_canonical(S) - 1.0;
                    y = 2.0 * mt_canonical(S) - 1.0;
                    r2 = x * x + y * y;
                } while (r2 > 1.0 || r2 == 0.0);
                const double mult = sqrt(-2 * log(r2) / r2);
                saved = x * mult; saved_avail = true;
                n = y * mult;
            }
            v = 1.0 + a2 * n;
        } while (v <= 0.0);
        v = v * v * v;
        u = mt_canonical(S);
    } while (u > 1.0 - 0.0331 * n * n * n * n && (log(u) > (0.5 * n * n + a1 * (1.0 - v + log(v)))));
    if (alpha == malpha) return a1 * v;
    do u = mt_canonical(S); while (u == 0.0);
    return pow(u, 1.0 / alpha) * a1 * v;
}
__device__ double serial_beta(const VglDevParams& P, VglSerialState* S) {
    if (P.beta_std) {
        const double x = std_gamma_fresh(S, P.beta_a);
        const double y = std_gamma_fresh(S, P.beta_b);
        return x / (x + y);
    }
    return beta_draw(P, S->st2);
}

// glibc rand(): random_r() TYPE_3
__device__ int glibc_rand(VglSerialState* S) {
    const uint32_t val = (S->rand_state[S->rand_f] += S->rand_state[S->rand_r]);
    if (++S->rand_f >= 31) { S->rand_f = 0; ++S->rand_r; }
    else if (++S->rand_r >= 31) S->rand_r = 0;
    return (int)(val >> 1);
}

__global__ __launch_bounds__(64) void k_scout(const VglDevParams P, const VglTilePtrs T, VglSerialState* S) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int N = P.n_samples;
    const size_t plane = (size_t)T.n_sites * N;
    uint64_t st0 = S->st0, st1 = S->st1;
    for (int ls = 0; ls < T.n_sites; ++ls) {
        const size_t e0 = (size_t)ls * N;
        // depths of all samples first (vcfgl.cpp:364-389)
        long long info_dp = 0;
        for (int s = 0; s < N; ++s) {
            int n;
            if (P.per_sample_depth) { const VglPois pc = P.pois[s]; n = poisson_draw_fast(pc, st1, P.gamma_ln_tab, P.gamma_ln_n); }
            else n = poisson_draw_fast(P.pois0, st1, P.gamma_ln_tab, P.gamma_ln_n);
            const uint32_t g = T.gt[e0 + s];
            if ((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF) n = 0;
            if (n > P.read_cap) { atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
            T.sdp[e0 + s] = n;
            info_dp += n;
        }
        uint64_t thresh = P.err_thresh;
        int last_base = -1;
        if (T.site_tail) { VglSiteTail z; z.sum = 0.0f; z.sumsq = 0.0f; z.base = -1; z.pad = 0; T.site_tail[ls] = z; }
        if (info_dp == 0) { if (P.error_qs == 1) T.site_thresh[ls] = thresh; continue; }   // nothing else is drawn (vcfgl.cpp:396-404)
        if (P.error_qs == 1) {                                     // vcfgl.cpp:425-437
            const double pe = serial_beta(P, S);
            thresh = (uint64_t)ceil(ldexp(pe, 48));
            T.site_thresh[ls] = thresh;
        }
        for (int s = 0; s < N; ++s) {
            const int dp = T.sdp[e0 + s];
            T.sst_hap[e0 + s] = st1;
            T.sst_base[e0 + s] = st0;
            if (dp == 0) continue;
            const uint32_t g = T.gt[e0 + s];
            const int a0 = g & 0xF, a1 = (g >> 4) & 0xF;
            for (int r = 0; r < dp; ++r) {                          // vcfgl.cpp:469-613
                bool fwd;
                last_base = sample_read_base(st1, st0, a0, a1, thresh, P.sample_strand != 0, fwd);
                if (P.error_qs == 2) T.errp[(size_t)r * plane + e0 + s] = serial_beta(P, S);
            }
        }
        if (P.add_i16 && T.site_tail) {                             // vcfgl.cpp:647-663
            VglSiteTail t; t.sum = 0.0f; t.sumsq = 0.0f; t.base = last_base; t.pad = 0;
            for (long long i = 0; i < info_dp; ++i) {
                int td = 1 + glibc_rand(S) / (2147483647 / (50 - 1 + 1) + 1);       // sample_from_range_rng_rand(1,50)
                if (td > 25) td = 25;                                            // CAP_TAIL_DIST
                t.sum += td; t.sumsq += (td * td);
            }
            T.site_tail[ls] = t;
        }
    }
    S->st0 = st0; S->st1 = st1;
}

// Wave-parallel scout for runs without per-read beta deviates (--error-qs 0/1).  The chains stay
// sequential, but the expensive work on them is done 64 stream positions at a time:
//   depth stream   every lane evaluates the rejection attempt that would START at its position; the
//                  chain (1 draw if em<0, else 2; accept ends a sample) is then walked with bit tests
//   base stream    the error tests of 64 consecutive reads are 64 independent compares on jumped
//                  states; only a read whose test fires (rate = error rate) is stepped sequentially
//   haplotype      one draw per read: the state of a sample's first read is a table jump
// Per-sample start states come from J^k jump tables relative to the current block.
// ordering between the lanes of the scout's single wavefront: LDS needs only program order,
// global scratch needs the writes to be visible to the other lanes' loads
__device__ __forceinline__ void scout_sync(const bool in_lds) {
    if (in_lds) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    else __threadfence();
}

__global__ __launch_bounds__(64) void k_scout_wave(const VglDevParams P, const VglTilePtrs T, VglSerialState* S) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    // jump table J^k in LDS: it sits on the critical path of every chain step
    VglAffine* stp = (VglAffine*)lds_all;
    { const VglAffine* gtab = P.step_tab; for (int i = threadIdx.x; i < 192; i += 64) stp[i] = gtab[i]; }
    uint8_t* lds_raw = lds_all + 192 * sizeof(VglAffine);
    scout_sync(true);
    const int lane = threadIdx.x;
    const int N = P.n_samples;
    const int stride = P.sample_strand ? 2 : 1;
    // per-site working set (depths, first-read indices, genotypes) lives in LDS when it fits: every
    // step of the sequential chains then waits on LDS latency, not on global memory
    const bool in_lds = (size_t)N * 9 <= (size_t)P.scout_lds_bytes;
    int32_t* dpv = in_lds ? (int32_t*)lds_raw : nullptr;
    int32_t* offv = in_lds ? (int32_t*)lds_raw + N : T.scout_off;
    uint8_t* gtl = in_lds ? (lds_raw + (size_t)8 * N) : nullptr;
    uint64_t st0 = S->st0, st1 = S->st1;                           // identical in every lane
    for (int ls = 0; ls < T.n_sites; ++ls) {
        const size_t e0 = (size_t)ls * N;
        int32_t* dps = in_lds ? dpv : (T.sdp + e0);
        const uint8_t* gts = T.gt + e0;
        if (in_lds) {
            for (int i = lane; i < N; i += 64) gtl[i] = gts[i];
            gts = gtl;
        }
        scout_sync(in_lds);
        // ---------------- depths of all samples (vcfgl.cpp:364-389)
        long long info_dp = 0;
        if (!P.per_sample_depth && !P.pois0.st12) {
            int s = 0, sw = -64;
            uint64_t missm = 0;
            while (s < N) {
                const uint64_t x1 = aff(stp[lane + 1], st1);
                const uint64_t x2 = lcg_next(x1);
                bool neg, rej; double em;
                poisson_attempt(P.pois0, x1, x2, true, P.gamma_ln_tab, P.gamma_ln_n, neg, rej, em);
                const uint64_t negm = __ballot(neg), rejm = __ballot(rej & !neg);
                const int emi = (int)em;
                int pos = 0;
                while (s < N && pos <= 63) {                       // scalar walk of the attempt chain
                    if ((negm >> pos) & 1) { pos += 1; continue; }
                    if ((rejm >> pos) & 1) { pos += 2; continue; }
                    if (s >= sw + 64) {                            // missing-genotype flags of the next 64 samples
                        sw = s;
                        const int sl = s + lane;
                        const uint32_t g = (sl < N) ? gts[sl] : 0u;
                        missm = __ballot((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF);
                    }
                    int n = __builtin_amdgcn_readlane(emi, __builtin_amdgcn_readfirstlane(pos));
                    if ((missm >> (s - sw)) & 1) n = 0;
                    if (n > P.read_cap) { if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
                    if (lane == 0) { dps[s] = n; if (in_lds) T.sdp[e0 + s] = n; }
                    info_dp += n; ++s; pos += 2;
                }
                st1 = aff(stp[pos], st1);                   // pos <= 65 draws consumed
            }
        } else if (!P.per_sample_depth) {
            // product method (rng.h:289-297): 64 uniforms per block from jumped states, then the
            // multiplication chain t *= u walked with scalar lane reads
            int s = 0, sw = -64;
            uint64_t missm = 0;
            double t = 1.0; int em = -1;
            while (s < N) {
                const double u = u01(aff(stp[lane + 1], st1));
                const int ulo = __double2loint(u), uhi = __double2hiint(u);
                int pos = 0;
                while (s < N && pos < 64) {
                    const int p1 = __builtin_amdgcn_readfirstlane(pos);
                    const double uk = __hiloint2double(__builtin_amdgcn_readlane(uhi, p1), __builtin_amdgcn_readlane(ulo, p1));
                    ++em; t *= uk; ++pos;
                    if (!(t > P.pois0.g)) {
                        if (s >= sw + 64) {
                            sw = s;
                            const int sl = s + lane;
                            const uint32_t g = (sl < N) ? gts[sl] : 0u;
                            missm = __ballot((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF);
                        }
                        int n = em;
                        if ((missm >> (s - sw)) & 1) n = 0;
                        if (n > P.read_cap) { if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
                        if (lane == 0) { dps[s] = n; if (in_lds) T.sdp[e0 + s] = n; }
                        info_dp += n; ++s; em = -1; t = 1.0;
                    }
                }
                st1 = aff(stp[pos], st1);
            }
        } else {
            for (int s = 0; s < N; ++s) {                          // per-sample means: lane-uniform
                int n;
                if (P.per_sample_depth) { const VglPois pc = P.pois[s]; n = poisson_draw_fast(pc, st1, P.gamma_ln_tab, P.gamma_ln_n); }
                else n = poisson_draw_fast(P.pois0, st1, P.gamma_ln_tab, P.gamma_ln_n);
                const uint32_t g = gts[s];
                if ((g & 0xF) == 0xF || ((g >> 4) & 0xF) == 0xF) n = 0;
                if (n > P.read_cap) { if (lane == 0) atomicOr(T.errflag, VGL_DEVERR_CAPACITY); n = P.read_cap; }
                if (lane == 0) { dps[s] = n; if (in_lds) T.sdp[e0 + s] = n; }
                info_dp += n;
            }
        }
        uint64_t thresh = P.err_thresh;
        if (T.site_tail && lane == 0) { VglSiteTail z; z.sum = 0.0f; z.sumsq = 0.0f; z.base = -1; z.pad = 0; T.site_tail[ls] = z; }
        if (info_dp == 0) { if (P.error_qs == 1 && lane == 0) T.site_thresh[ls] = thresh; continue; }
        if (P.error_qs == 1) {                                     // one beta deviate per site (vcfgl.cpp:425-437)
            uint32_t lo = 0, hi = 0;
            if (lane == 0) {
                const double pe = serial_beta(P, S);
                const uint64_t th = (uint64_t)ceil(ldexp(pe, 48));
                T.site_thresh[ls] = th; lo = (uint32_t)th; hi = (uint32_t)(th >> 32);
            }
            lo = __shfl(lo, 0, 64); hi = __shfl(hi, 0, 64);
            thresh = ((uint64_t)hi << 32) | lo;
        }
        scout_sync(in_lds);                                        // depths written by lane 0, read by all lanes below
        // ---------------- exclusive prefix sums of the depths = first read index of each sample
        {
            int run = 0;
            for (int c0 = 0; c0 < N; c0 += 64) {
                const int s = c0 + lane;
                const int d = (s < N) ? dps[s] : 0;
                int incl = d;
#pragma unroll
                for (int k = 1; k < 64; k <<= 1) { const int t = __shfl_up(incl, k, 64); if (lane >= k) incl += t; }
                if (s < N) offv[s] = run + incl - d;
                run += __shfl(incl, 63, 64);
            }
        }
        scout_sync(in_lds);
        // ---------------- reads of the site as one sequence of R = INFO/DP reads (vcfgl.cpp:469-613)
        const int R = (int)info_dp;
        int q0 = 0, sn = 0;                                        // reads done / samples whose start state is recorded
        uint64_t hb = st1, bb = st0;                               // haplotype / base stream states before read q0
        int last_err_read = -1, last_err_base = -1;
        while (sn < N || q0 < R) {                                 // every sample recorded and every read consumed
            const int nb = (R - q0 < 64) ? (R - q0) : 64;
            bool cand = false;
            if (lane < nb) cand = aff(stp[lane * stride + 1], bb) < thresh;      // error test of read q0+lane (u < e)
            const uint64_t m = __ballot(cand);
            const int nclean = m ? (__ffsll((unsigned long long)m) - 1) : nb;
            int cnt;
            do {                                                   // samples that start inside the clean range
                const int s = sn + lane;
                const int o = (s < N) ? offv[s] : 0x7fffffff;
                const bool ok = (s < N) & (o <= q0 + nclean);
                if (ok) {
                    T.sst_hap[e0 + s] = aff(stp[o - q0], hb);
                    T.sst_base[e0 + s] = aff(stp[(o - q0) * stride], bb);
                }
                cnt = __popcll(__ballot(ok));
                sn += cnt;
            } while (cnt == 64);
            if (m == 0) {
                hb = aff(stp[nb], hb); bb = aff(stp[nb * stride], bb); q0 += nb;
            } else {
                // read q0+nclean is miscalled: its sample is the last recorded one with reads
                const int qe = q0 + nclean;
                int se = -1;
                for (int back = 0; se < 0; back += 64) {
                    const int s = sn - 1 - back - lane;
                    const bool has = (s >= 0) && (dps[s] > 0);
                    const uint64_t hm = __ballot(has);
                    if (hm) se = sn - 1 - back - (__ffsll((unsigned long long)hm) - 1);
                    else if (sn - 1 - back - 64 < 0) break;
                }
                const uint32_t g = gts[se < 0 ? 0 : se];
                const uint64_t hx = aff(stp[nclean + 1], hb);
                const int true_base = (hx < (1ULL << 47)) ? (int)(g & 0xF) : (int)((g >> 4) & 0xF);
                uint64_t x = aff(stp[nclean * stride + 1], bb);
                int rb;
                do { x = lcg_next(x); rb = (int)(x >> 46); } while (rb == true_base);      // vcfgl.cpp:487
                if (P.sample_strand) x = lcg_next(x);
                bb = x; hb = hx; q0 = qe + 1;
                last_err_read = qe; last_err_base = rb;
            }
        }
        st1 = hb; st0 = bb;
        if (P.add_i16 && T.site_tail && lane == 0) {               // vcfgl.cpp:647-663 (stale r_base = base of the last read)
            int last_base = last_err_base;
            if (last_err_read != R - 1) {
                int se = N - 1;
                while (se > 0 && dps[se] == 0) --se;
                const uint32_t g = gts[se];
                last_base = (hb < (1ULL << 47)) ? (int)(g & 0xF) : (int)((g >> 4) & 0xF);
            }
            VglSiteTail t; t.sum = 0.0f; t.sumsq = 0.0f; t.base = last_base; t.pad = 0;
            for (int i = 0; i < R; ++i) {
                int td = 1 + glibc_rand(S) / (2147483647 / (50 - 1 + 1) + 1);
                if (td > 25) td = 25;
                t.sum += td; t.sumsq += (td * td);
            }
            T.site_tail[ls] = t;
        }
    }
    if (lane == 0) { S->st0 = st0; S->st1 = st1; }
}

// parallel evaluation from the recorded states: same per-read code as k_sample
__global__ __launch_bounds__(256) void k_sample_serial(const VglDevParams P, const VglTilePtrs T) {
    const WavePos wp = wave_pos(P, T);
    if (!wp.valid) return;
    const int lane = threadIdx.x & 63;
    const int N = P.n_samples;
    const int ls = wp.ls;
    const int s = wp.chunk * 64 + lane;
    const bool active = s < N;
    const size_t ev = (size_t)ls * N + (size_t)wp.chunk * 64 + (active ? lane : 0);
    const size_t plane = (size_t)T.n_sites * N;
    int dp = 0;
    uint64_t ad4 = 0, adf4 = 0;
    uint32_t qs[4] = {0, 0, 0, 0}, qq[4] = {0, 0, 0, 0};
    if (active) {
        uint64_t st_hap = T.sst_hap[ev], st_base = T.sst_base[ev];
        const uint32_t g = T.gt[ev];
        const int a0 = g & 0xF, a1 = (g >> 4) & 0xF;
        dp = T.sdp[ev];                                            // the scout's depth draw (0 for a missing genotype)
        if (dp >= P.read_cap) atomicOr(T.errflag, (dp > P.read_cap) ? VGL_DEVERR_CAPACITY : 0u);
        const uint64_t err_thresh = (P.error_qs == 1) ? T.site_thresh[ls] : P.err_thresh;
        for (int r = 0; r < dp; ++r) {
            bool fwd;
            const int r_base = sample_read_base(st_hap, st_base, a0, a1, err_thresh, P.sample_strand != 0, fwd);
            int q_i = P.pre_q, aq_i = P.pre_adjq;
            if (P.error_qs == 2) errprob_to_qs(P, T.errp[(size_t)r * plane + ev], q_i, aq_i, T.errflag);
            const int q_gl = (P.adjust_qs & 1) ? aq_i : q_i;
            T.reads[(size_t)r * plane + ev] = (uint8_t)((q_gl << 2) | r_base);
            if (T.reads_out && r < T.reads_out_cap) T.reads_out[(size_t)r * plane + ev] = (uint8_t)((q_i << 2) | r_base);
            if (P.need_qsum) {
                const uint32_t qv = (uint32_t)((P.adjust_qs & 2) ? aq_i : q_i);
                const uint32_t q2 = (uint32_t)qs_to_qssq((int)qv);
#pragma unroll
                for (int b = 0; b < 4; ++b) { qs[b] += (r_base == b) ? qv : 0u; qq[b] += (r_base == b) ? q2 : 0u; }
            }
            const uint64_t one = 1ULL << (16 * r_base);
            ad4 += one;
            if (fwd) adf4 += one;
        }
        if (!P.sample_strand) adf4 = ad4;
        if (T.fmt_dp) T.fmt_dp[ev] = dp;
        T.ad4[ev] = ad4;
        if (P.need_adf) T.adf4[ev] = adf4;
        if (P.need_qsum) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                T.qsum[((size_t)ls * 4 + b) * N + s] = qs[b];
                if (P.need_qsumsq) T.qsumsq[((size_t)ls * 4 + b) * N + s] = qq[b];
            }
        }
        if (T.reads_out) for (int r = dp; r < T.reads_out_cap; ++r) T.reads_out[(size_t)r * plane + ev] = 0xFF;
    }
    int v[9];
    v[0] = dp;
#pragma unroll
    for (int b = 0; b < 4; ++b) { v[1 + b] = (int)((ad4 >> (16 * b)) & 0xFFFF); v[5 + b] = (int)((adf4 >> (16 * b)) & 0xFFFF); }
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) {
        int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
#pragma unroll
        for (int k = 0; k < 9; ++k) if (v[k]) atomicAdd(&acc[k], v[k]);
    }
}

// ------------------------------------------------------------------------------------
// one lane per site: status + allele order (vcfgl.cpp:396-404, 665-766; no-reads :228-315)
__global__ __launch_bounds__(256) void k_site(const VglDevParams P, const VglTilePtrs T) {
    const int ls = blockIdx.x * blockDim.x + threadIdx.x;
    if (ls >= T.n_sites) return;
    const int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
    const int A = P.A;
    const int info_dp = acc[0];
    int ad[4] = {acc[1], acc[2], acc[3], acc[4]};
    int a2b[5] = {-1, -1, -1, -1, -1}, b2a[5] = {-1, -1, -1, -1, -1};
    int status = SITE_OK, nAll = 0, nObs = 0;

    if (0 == info_dp) {
        if (P.rm_empty_sites) status = SITE_SKIP_EMPTY;
        else {
            status = SITE_NO_READS;
            switch (P.do_unobserved) {
                case 0: nAll = 1; nObs = 0; break;
                case 1: case 2: nAll = 1; nObs = 0; a2b[0] = 4; break;
                case 3: nAll = 4; nObs = 4; for (int a = 0; a < 4; a++) a2b[a] = a; break;
                default: nAll = 5; nObs = 4; for (int a = 0; a < 4; a++) a2b[a] = a; a2b[4] = 4; break;
            }
        }
    } else {
        int nObservedBases = 0;
        for (int b = 0; b < 4; b++) if (ad[b] > 0) nObservedBases++;
        if ((P.rm_invar_sites & 4) && 1 == nObservedBases) status = SITE_SKIP_INVAR;
        else {
            int sorted[4] = {0, 1, 2, 3};
            for (int i = 1; i < 4; i++)
                for (int j = i; j > 0 && ad[sorted[j]] > ad[sorted[j - 1]]; j--) { int t = sorted[j]; sorted[j] = sorted[j - 1]; sorted[j - 1] = t; }
            for (int a = 0; a < 4; a++) { a2b[a] = sorted[a]; b2a[sorted[a]] = a; }
            const bool explode = P.do_unobserved >= 3, add_unobs = (A == 5);
            for (int b = 0; b < 4; b++)
                if (!(ad[b] > 0) && !explode) { a2b[b2a[b]] = -1; b2a[b] = -1; }
            int unobs = -1, n_all = 0;
            for (int a = 0; a < 5; a++) { if (-1 == a2b[a]) { if (add_unobs) unobs = a; break; } ++n_all; }
            if (add_unobs) { a2b[unobs] = 4; b2a[4] = unobs; }
            nObs = n_all; nAll = n_all + (add_unobs ? 1 : 0);
        }
    }
    VglSiteInfo si;
    si.status = status; si.n_alleles = nAll;
    uint32_t pa = 0, pb = 0;
    for (int k = 0; k < 5; k++) { pa |= (uint32_t)(b2a[k] & 0xF) << (4 * k); pb |= (uint32_t)(a2b[k] & 0xF) << (4 * k); }
    si.acgt2alleles = pa; si.alleles2acgt = pb;
    T.sinfo[ls] = si;

    T.site_status[ls] = status;
    T.n_alleles[ls] = nAll;
    if (T.n_alleles_obs) T.n_alleles_obs[ls] = nObs;
    for (int k = 0; k < 5; k++) T.alleles2acgt[(size_t)ls * 5 + k] = (int8_t)a2b[k];
    if (T.info_dp) T.info_dp[ls] = info_dp;
    const bool have = (status == SITE_OK);
    for (int a = 0; a < A; a++) {
        const int b = (have && a < nAll) ? a2b[a] : -1;
        const bool real = (b >= 0 && b < 4);
        const int tot = real ? acc[1 + b] : 0;
        const int totf = real ? acc[5 + b] : 0;
        if (T.info_ad) T.info_ad[(size_t)ls * A + a] = tot;
        if (T.info_adf) T.info_adf[(size_t)ls * A + a] = totf;
        if (T.info_adr) T.info_adr[(size_t)ls * A + a] = tot - totf;
    }
}

// ------------------------------------------------------------------------------------
__device__ __forceinline__ float f32_missing() { return __uint_as_float(F32_MISSING_BITS); }
__device__ __forceinline__ int nib(uint32_t v, int k) { return (int)((v >> (4 * k)) & 0xF); }
__device__ __forceinline__ int cnt_of(uint64_t ad4, int b) { return b < 4 ? (int)((ad4 >> (16 * b)) & 0xFFFF) : 0; }

// Evaluations differ in depth, and the likelihood loop runs once per read: a wavefront is busy for
// its deepest evaluation.  The 256 evaluations of a workgroup are therefore re-dealt to the lanes in
// depth order (LDS counting sort of the thread ids), so each wavefront works on evaluations of
// similar depth; loads and stores stay inside the workgroup's 256-evaluation window of each plane.
template <int A>
__global__ __launch_bounds__(256) void k_gl(const VglDevParams P, const VglTilePtrs T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    __shared__ uint32_t s_hist[1026];
    __shared__ uint16_t s_perm[256];
    constexpr int NG = A * (A + 1) / 2;
    const int N = P.n_samples;
    const int tid = threadIdx.x;
    const int64_t nwaves = (int64_t)T.n_sites * P.chunks;
    // ---- depth of the evaluation this thread would own in natural order
    int dp0 = -1;                                                      // -1: no evaluation (padding lane)
    {
        const int64_t w = (int64_t)blockIdx.x * 4 + (tid >> 6);
        if (w < nwaves) {
            const int ls0 = (int)(w / P.chunks);
            const int s0 = (int)(w - (int64_t)ls0 * P.chunks) * 64 + (tid & 63);
            if (s0 < N) {
                const uint64_t a = T.ad4[(size_t)ls0 * N + s0];
                dp0 = (int)((a & 0xFFFF) + ((a >> 16) & 0xFFFF) + ((a >> 32) & 0xFFFF) + ((a >> 48) & 0xFFFF));
                if (dp0 > 1023) dp0 = 1023;
            }
        }
    }
    for (int i = tid; i < 1026; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int key = 1024 - dp0;                                        // deepest first; padding (dp0 = -1) last
    atomicAdd(&s_hist[key], 1u);
    __syncthreads();
    if (tid < 64) {                                                    // exclusive scan of 1026 bins by one wavefront
        uint32_t run = 0;
        for (int base = 0; base < 1026; base += 64) {
            const int i = base + tid;
            const uint32_t v = (i < 1026) ? s_hist[i] : 0u;
            uint32_t incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (tid >= d) incl += t; }
            if (i < 1026) s_hist[i] = run + incl - v;
            run += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();
    s_perm[atomicAdd(&s_hist[key], 1u)] = (uint16_t)tid;
    __syncthreads();
    const int otid = s_perm[tid];                                      // thread id whose evaluation this lane processes
    const int lane = tid & 63;
    const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t w = (int64_t)blockIdx.x * 4 + (otid >> 6);
    if (w >= nwaves) return;
    const int ls = (int)(w / P.chunks);
    const int s = (int)(w - (int64_t)ls * P.chunks) * 64 + (otid & 63);
    if (s >= N) return;
    const size_t ev = (size_t)ls * N + s;
    const size_t plane = (size_t)T.n_sites * N;
    const VglSiteInfo si = T.sinfo[ls];
    const int nA = si.n_alleles;
    const int nG = nA * (nA + 1) / 2;
    const bool have = (si.status == SITE_OK);
    const float MISS = f32_missing();

    const uint64_t ad4 = T.ad4[ev];
    const int dp = have ? (int)((ad4 & 0xFFFF) + ((ad4 >> 16) & 0xFFFF) + ((ad4 >> 32) & 0xFFFF) + ((ad4 >> 48) & 0xFFFF)) : 0;

    float acc[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) acc[i] = -0.0f;                                     // bcf_utils.h:310

    if (dp > 0) {
        if (P.gl_model == 2) {
            // gl_methods.cpp:22-59 / :94-139 / :171-220
            const bool per_read = (P.error_qs == 2);
            double homT = P.pre_homT, het = P.pre_het, homF = P.pre_homF;
            for (int r = 0; r < dp; ++r) {
                const uint32_t rb = T.reads[(size_t)r * plane + ev];
                const int ao = nib(si.acgt2alleles, (int)(rb & 3));
                if (per_read) {
                    if (!P.precise_gl) {
                        const int q = (int)(rb >> 2);
                        homT = P.q2gl[q]; het = P.q2gl[257 + q]; homF = P.q2gl[514 + q];
                    } else {
                        const double e = T.errp[(size_t)r * plane + ev];
                        if (0.0 == e) { homT = 0.0; het = -0.30103; homF = -INFINITY; }
                        else { homT = log10(1.0 - e); het = log10((1.0 - e) / 2.0 + e / 6.0); homF = log10(e / 3.0); }
                    }
                }
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < A; ++i) {
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        const int idx = i * (i + 1) / 2 + j;                         // bcf_alleles2gt
                        const double t = (i == j) ? ((ao == i) ? homT : homF) : ((ao == i || ao == j) ? het : homF);
                        const float v = (float)((double)acc[idx] + t);
                        acc[idx] = v;
                        if (i < nA) mx = (v > mx) ? v : mx;
                    }
                }
#pragma unroll
                for (int i = 0; i < NG; ++i) acc[i] -= mx;
            }
        } else {
            // GL model 1 with one fixed qScore: errmod_cal() reduces to table lookups on the
            // per-base depths (gl_methods.cpp:304-369; htslib errmod.c restated in vgl_host.cpp)
            int n = dp;
            if (n > 255) { atomicOr(T.errflag, VGL_DEVERR_GL1DEPTH); n = 255; }
            int c[5]; double bs[5];
#pragma unroll
            for (int b = 0; b < 4; ++b) { c[b] = (int)((ad4 >> (16 * b)) & 0xFFFF); if (c[b] > 255) c[b] = 255; }
            c[4] = 0; bs[4] = 0.0;
            if (P.error_qs != 2) {
#pragma unroll
                for (int b = 0; b < 4; ++b) bs[b] = P.gl1_bsum[n * 256 + c[b]];
            } else {
                // per-read qScores (gl_methods.cpp:233-302): errmod_cal() walks the reads in descending
                // (qual, base) order, so each base accumulates fk[i]*beta[q][n][i] over its own reads in
                // descending quality: a per-lane (base, qual) histogram in LDS replaces the sort
                uint8_t* h = lds_raw + (size_t)wib * 16384 + lane;
                for (int bin = 0; bin < 256; ++bin) h[bin * 64] = 0;
                for (int r = 0; r < n; ++r) {
                    const uint32_t rb = T.reads[(size_t)r * plane + ev];
                    const int bin = (int)(((rb & 3) << 6) | (rb >> 2));
                    h[bin * 64] = (uint8_t)(h[bin * 64] + 1);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    double acc_b = 0.0; int i = 0;
                    for (int q = 63; q >= 0; --q) {
                        const int cnt = h[(b * 64 + q) * 64];
                        const int qq = q < 4 ? 4 : q;                          // errmod_cal clamps qual to [4,63]
                        for (int j = 0; j < cnt; ++j) { acc_b += P.gl1_fk[i] * P.gl1_beta[((size_t)qq << 16) | ((size_t)n << 8) | (size_t)i]; ++i; }
                    }
                    bs[b] = acc_b;
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < A; ++i) {
#pragma unroll
                for (int j = 0; j <= i; ++j) {
                    const int idx = i * (i + 1) / 2 + j;
                    const int b1 = nib(si.alleles2acgt, j), b2 = nib(si.alleles2acgt, i);
                    float tmp1 = 0.0f; int tmp2 = 0;
#pragma unroll
                    for (int k = 0; k < 5; ++k)
                        if (k != b1 && k != b2) { tmp1 = (float)((double)tmp1 + bs[k]); tmp2 += c[k]; }
                    float q;
                    if (b1 == b2) q = tmp2 ? tmp1 : 0.0f;
                    else {
                        const int lo = b1 < b2 ? b1 : b2, hi = b1 < b2 ? b2 : b1;
                        const int chi = cnt_of(ad4, hi) > 255 ? 255 : cnt_of(ad4, hi);
                        int cjk = cnt_of(ad4, lo) + cnt_of(ad4, hi); if (cjk > 255) cjk = 255;
                        const double lh = P.gl1_lhet[cjk << 8 | chi];
                        q = tmp2 ? (float)(-4.343 * lh + (double)tmp1) : (float)(-4.343 * lh);
                    }
                    if (q < 0.0f) q = 0.0f;
                    const float v = (float)((-1.0 * (double)q) / 10.0);
                    acc[idx] = v;
                    if (i < nA) mx = (v > mx) ? v : mx;
                }
            }
#pragma unroll
            for (int i = 0; i < NG; ++i) acc[i] -= mx;
        }
    }

    // ---- GL / PL / GP planes (vcfgl.cpp:907-970; no-reads site :297-305)
    float gp[NG]; float sum_gps = 0.0f;
    const bool sample_ok = have && dp > 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const bool valid = sample_ok && i < nG;
        const float v = valid ? acc[i] : MISS;
        const size_t o = ((size_t)ls * NG + i) * N + s;
        if (T.gl) T.gl[o] = v;
        if (T.pl) {
            int32_t x;
            if (!valid) x = I32_MISSING;
            else if (v == -INFINITY) x = MAXPL;
            else { x = (int32_t)lroundf((float)(-10.0 * (double)v)); if (x > MAXPL) x = MAXPL; }
            T.pl[o] = x;
        }
        if (T.gp) { gp[i] = valid ? (float)pow(10.0, (double)v) : 0.0f; if (valid) sum_gps += gp[i]; }
    }
    if (T.gp) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const bool valid = sample_ok && i < nG;
            T.gp[((size_t)ls * NG + i) * N + s] = valid ? gp[i] / sum_gps : MISS;
        }
    }
    // ---- FORMAT/AD, ADF, ADR in allele order (vcfgl.cpp:806-843)
    if (T.fmt_ad || T.fmt_adf || T.fmt_adr) {
        const uint64_t adf4 = P.need_adf ? T.adf4[ev] : ad4;
#pragma unroll
        for (int a = 0; a < A; ++a) {
            const int b = (have && a < nA) ? nib(si.alleles2acgt, a) : 0xF;
            const int v = cnt_of(ad4, b), vf = cnt_of(adf4, b);
            const size_t o = ((size_t)ls * A + a) * N + s;
            if (T.fmt_ad) T.fmt_ad[o] = v;
            if (T.fmt_adf) T.fmt_adf[o] = vf;
            if (T.fmt_adr) T.fmt_adr[o] = v - vf;
        }
    }
}

// ------------------------------------------------------------------------------------
// one lane per site, samples in order: the reference accumulates these in float32 in sample
// order (vcfgl.cpp:875-897, 997-1066), which a tree reduction would not reproduce bit for bit.
__global__ __launch_bounds__(64) void k_siteagg(const VglDevParams P, const VglTilePtrs T) {
    const int ls = blockIdx.x * blockDim.x + threadIdx.x;
    if (ls >= T.n_sites) return;
    const int N = P.n_samples, A = P.A;
    const VglSiteInfo si = T.sinfo[ls];
    const bool have = (si.status == SITE_OK);
    const int nA = si.n_alleles;
    const uint32_t* qsum = T.qsum + (size_t)ls * 4 * N;
    if (T.qs) {
        float qsv[5] = {0, 0, 0, 0, 0};
        if (have && P.need_qsum)
            for (int s = 0; s < N; ++s) {
                float sum = 0.0f;
                uint32_t q[4];
                for (int b = 0; b < 4; ++b) { q[b] = qsum[(size_t)b * N + s]; sum += (float)(int)q[b]; }
                if (0.0f != sum)
                    for (int b = 0; b < 4; ++b) {
                        const int a = nib(si.acgt2alleles, b);
                        if (a == 0xF) continue;
                        const float add = (float)((float)(int)q[b] / sum);
                        qsv[0] += (a == 0) ? add : 0.0f; qsv[1] += (a == 1) ? add : 0.0f; qsv[2] += (a == 2) ? add : 0.0f;
                        qsv[3] += (a == 3) ? add : 0.0f; qsv[4] += (a == 4) ? add : 0.0f;
                    }
            }
        for (int a = 0; a < A; ++a) T.qs[(size_t)ls * A + a] = qsv[a];
    }
    if (T.i16) {
        float v[16];
        for (int k = 0; k < 16; ++k) v[k] = 0.0f;
        if (have && P.add_i16 && nA > 1) {
            const int32_t* acc = T.acc + (size_t)ls * VGL_ACC_STRIDE;
            const uint32_t* qsq = T.qsumsq + (size_t)ls * 4 * N;
            const int refb = nib(si.alleles2acgt, 0);
            const int nObs = (A == 5) ? nA - 1 : nA;
            v[0] = (float)acc[5 + refb]; v[1] = (float)(acc[1 + refb] - acc[5 + refb]);
            const float mq = (float)P.i16_mapq, mq2 = (float)(P.i16_mapq * P.i16_mapq);
            for (int s = 0; s < N; ++s) {
                v[4] += (float)(int)qsum[(size_t)refb * N + s];
                v[5] += (float)(int)qsq[(size_t)refb * N + s];
                const uint64_t ad4 = T.ad4[(size_t)ls * N + s];
                for (int a = 0; a < nA; ++a) {
                    if (a == nObs) continue;
                    const int cnt = cnt_of(ad4, nib(si.alleles2acgt, a));
                    for (int i = 0; i < cnt; ++i) {
                        if (0 == a) { v[8] += mq; v[9] += mq2; } else { v[10] += mq; v[11] += mq2; }
                    }
                }
            }
            for (int a = 1; a < nA; ++a) {
                if (a == nObs) continue;
                const int b = nib(si.alleles2acgt, a);
                v[2] += (float)acc[5 + b]; v[3] += (float)(acc[1 + b] - acc[5 + b]);
                for (int s = 0; s < N; ++s) { v[6] += (float)(int)qsum[(size_t)b * N + s]; v[7] += (float)(int)qsq[(size_t)b * N + s]; }
            }
            // tail distance (vcfgl.cpp:1029-1071): drawn from libc rand() by the serial-mode scout; zero in tile mode
            if (T.site_tail) {
                const VglSiteTail tl = T.site_tail[ls];
                if (tl.base == refb) { v[12] = tl.sum; v[13] = tl.sumsq; }
                for (int a = 1; a < nA; ++a) {
                    if (a == nObs) continue;
                    if (nib(si.alleles2acgt, a) == tl.base) { v[14] += tl.sum; v[15] += tl.sumsq; }
                }
            }
        }
        for (int k = 0; k < 16; ++k) T.i16[(size_t)ls * 16 + k] = v[k];
    }
}

// ------------------------------------------------------------------------------------
extern "C" int vgl_launch_scout(const VglDevParams* p, const VglTilePtrs* t, VglSerialState* st, void* stream) {
    if (t->n_sites == 0) return 0;
    if (p->error_qs != 2) hipLaunchKernelGGL(k_scout_wave, dim3(1), dim3(64), (size_t)p->scout_lds_bytes + 192 * sizeof(VglAffine), (hipStream_t)stream, *p, *t, st);
    else hipLaunchKernelGGL(k_scout, dim3(1), dim3(64), 0, (hipStream_t)stream, *p, *t, st);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_sample(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    const int64_t waves = (int64_t)t->n_sites * p->chunks;
    if (waves == 0) return 0;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    if (p->serial) {
        hipLaunchKernelGGL(k_sample_serial, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, *t);
        return (int)hipGetLastError();
    }
    const bool dbg = t->dbg != nullptr;                         // VGL_DEBUG_STAMPS / VGL_DEBUG_PHASE
    if (p->error_qs == 2) {
        if (dbg) hipLaunchKernelGGL((k_sample<2, true>), dim3(blocks), dim3(256), (size_t)4 * p->pool_lds_bytes, (hipStream_t)stream, *p, *t);
        else hipLaunchKernelGGL((k_sample<2, false>), dim3(blocks), dim3(256), (size_t)4 * p->pool_lds_bytes, (hipStream_t)stream, *p, *t);
    } else if (p->error_qs == 1) hipLaunchKernelGGL((k_sample<1, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, *t);
    else hipLaunchKernelGGL((k_sample<0, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_site(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0) return 0;
    hipLaunchKernelGGL(k_site, dim3((t->n_sites + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_gl(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    const int64_t waves = (int64_t)t->n_sites * p->chunks;
    if (waves == 0) return 0;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    const size_t lds = (p->gl_model == 1 && p->error_qs == 2) ? (size_t)4 * 16384 : 0;   // (base,qual) histograms
    if (p->A == 5) hipLaunchKernelGGL(k_gl<5>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, *p, *t);
    else hipLaunchKernelGGL(k_gl<4>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

extern "C" int vgl_launch_siteagg(const VglDevParams* p, const VglTilePtrs* t, void* stream) {
    if (t->n_sites == 0) return 0;
    hipLaunchKernelGGL(k_siteagg, dim3((t->n_sites + 63) / 64), dim3(64), 0, (hipStream_t)stream, *p, *t);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------
// debug hook (not part of the C ABI): raw v_log_f32 over a buffer, used by
// tests/test_gpu_parity.py to check the error bound the fast decision paths assume
__global__ void k_dbg_vlog(const float* in, float* out, int n, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mode == 0) out[i] = __builtin_amdgcn_logf(in[i]);          // v_log_f32
    else if (mode == 1) out[i] = tanf(in[i]);                      // ocml tanf
    else out[i] = __builtin_amdgcn_exp2f(in[i]);                   // v_exp_f32
}
extern "C" int vgl_dbg_vlog(const float* d_in, float* d_out, int n, int mode) {
    hipLaunchKernelGGL(k_dbg_vlog, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_out, n, mode);
    return (int)hipDeviceSynchronize();
}
