// vgl_bounds.hip -- the error bounds that the float32 decision paths of vgl_common.hip.h assume, measured on the
// device over EVERY float32 argument of the range each one is used on (a sweep of 10^9 bit patterns takes a fraction of
// a second here).  Diagnostic entry point, not part of the public C ABI: tests/test_gpu_bounds.py asserts
// `violations == 0` for every mode, so an ocml / compiler / hardware change that moves a bound fails the -m gpu suite
// instead of silently changing an integer somewhere in 10^9 evaluations.
//
// Every mode evaluates the SAME inline helper the kernels use (fast_ln, gamma_test_lu, gamma_rhs_series, qs_tf,
// the bound expressions of poisson_fast, div_inrange ...) against a float64 evaluation of the exact quantity, and compares the difference with
// the SAME bound expression the kernels use, minus the part of that bound which is reserved for the rounding of the
// argument from double to float (stated per mode).
#include "vgl_common.hip.h"

enum {
    VGL_BOUND_FAST_LN = 0,      // fast_ln / fast_ln_err on (0, 1]                      (normal_slow_test)
    VGL_BOUND_GAMMA_LU = 1,     // gamma_test_lu / log part of gamma_test_margin on (0, 1]  (gamma_slow_test)
    VGL_BOUND_GAMMA_SERIES = 2, // gamma_rhs_series against the analytic series, |s| <= 0.3333   (param = a1)
    VGL_BOUND_GAMMA_REFEXPR = 3,// gamma_rhs_series against the reference's own double expression (param = a1)
    VGL_BOUND_QS_TF = 4,        // qs_tf / qs_tf_margin on (1e-37, 1)                    (qs_decide_pf)
    VGL_BOUND_RCP = 5,          // v_rcp_f32 within 1 ulp on every normal float           (qs_stage_pf)
    VGL_BOUND_TANF = 6,         // vgl_tanf_0pi against tan on (0, VGL_PI]: the function's share of poisson_fast's dy
    VGL_BOUND_EXP2 = 7,         // v_exp_f32 on [-126, 8]                                 (poisson_attempt)
    VGL_BOUND_DIV = 8,          // div_inrange == IEEE quotient, operands shaped like the pool loop's (count = pairs)
    VGL_BOUND_QUOT = 9,         // quot_int24 == IEEE float32 quotient of integers q <= sum <= 2^24 (k_siteagg; count = pairs)
    VGL_BOUND_DIV10 = 10,       // div10_f32(x) == (float)((double)x / 10.0) for every float32 bit pattern of the sweep (k_gl, GL model 1)
    VGL_BOUND_POISSON = 11,     // poisson_fast == poisson_exact wherever it does not call the attempt ambiguous (param = mean depth; count = attempts)
    VGL_BOUND_POOL32 = 12,      // the float32 pool loop of k_sample<2>: every decision it takes equals the float64 one, values within their bounds (param = shape alpha >= 8; count = attempts)
    VGL_BOUND_QS_FIX = 13,      // qs_decide_fix: every float32 p it decides has ONE score over the whole interval its true value can lie in (param = adjust_by, 0 = none)
    VGL_BOUND_N
};

struct BoundAcc { unsigned long long n, viol, max_ratio_bits, arg_bits; };

__device__ __forceinline__ void acc_update(double& mr, uint32_t& arg, unsigned long long& viol, const double err, const double bound, const uint32_t bits) {
    // ratio err / bound; a non-positive bound with a non-zero error is a violation outright
    const double r = (bound > 0.0) ? err / bound : (err > 0.0 ? 1e300 : 0.0);
    if (!(r <= 1.0)) viol++;                                  // NaN counts as a violation
    if (r > mr || r != r) { mr = (r != r) ? 1e300 : r; arg = bits; }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_bound_sweep(const uint32_t lo, const unsigned long long count, const double param, BoundAcc* out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    double mr = 0.0; uint32_t arg = 0; unsigned long long viol = 0, n = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint32_t bits = lo + (uint32_t)i;
        const float x = __uint_as_float(bits);
        const double xd = (double)x;
        ++n;
        if (MODE == VGL_BOUND_FAST_LN) {
            // kernel: u double in (0,1), uf = (float)u, l = fast_ln(uf), claim |l - ln(u)| <= fast_ln_err(l).
            // |ln(u) - ln(uf)| <= 2^-24 / (1 - 2^-24) < 5.97e-8 is the argument's share.
            const double l = fast_ln(x);
            acc_update(mr, arg, viol, fabs(l - log(xd)), fast_ln_err(l) - 5.97e-8, bits);
        } else if (MODE == VGL_BOUND_GAMMA_LU) {
            const float lu = gamma_test_lu(x);
            const float b = fabsf(lu) * 0x1p-20f + 0x1p-21f;   // the log part of gamma_test_margin
            // d = lu + g is rounded once more (2^-24 of the larger term, g <= |lu| near the decision): 2^-24 |lu| reserved
            acc_update(mr, arg, viol, fabs((double)lu - log(xd)), (double)b - 5.97e-8 - fabs((double)lu) * 0x1p-24, bits);
        } else if (MODE == VGL_BOUND_DIV10) {
            // exact equality wanted: error 0 against a bound of 0 passes, anything else is a violation (acc_update)
            const float want = (float)(xd / 10.0);
            acc_update(mr, arg, viol, (__float_as_uint(div10_f32(x)) == __float_as_uint(want)) ? 0.0 : 1.0, 0.0, bits);
        } else if (MODE == VGL_BOUND_GAMMA_SERIES || MODE == VGL_BOUND_GAMMA_REFEXPR) {
            if (!(fabsf(x) <= 0.3333f)) { --n; continue; }
            const float a1f = (float)param;
            const float g = gamma_rhs_series(x, a1f);
            const double a1 = param, s = xd;
            double exact;
            if (MODE == VGL_BOUND_GAMMA_SERIES) {
                double p = 0.0;                                   // sum_{k<60} (-s)^k / (k+4), Horner
                for (int k = 59; k >= 0; --k) p = p * (-s) + 1.0 / (double)(k + 4);
                exact = 3.0 * a1 * (s * s) * (s * s) * p;
            } else {
                // the reference's right-hand side as it computes it (rng.h:139-145), x = s / a2, a2 = 1/sqrt(9 a1)
                const double a2 = 1.0 / sqrt(9.0 * a1);
                const double xn = s / a2;
                const double w = 1.0 + a2 * xn;
                const double v = w * w * w;
                exact = -(0.5 * (xn * xn) + a1 * (1.0 - v + log(v)));
            }
            // argument rounding: sf = (float)s and a1f = (float)a1 move g by <= (4.4 + 1) 2^-24 < 3.3e-7 relative
            const double bound = (double)g * (4e-6 - 3.3e-7) + (MODE == VGL_BOUND_GAMMA_REFEXPR ? 1e-10 : 1e-30);
            acc_update(mr, arg, viol, fabs((double)g - exact), bound, bits);
        } else if (MODE == VGL_BOUND_QS_TF) {
            // kernel: pf within 1.25 x 2^-22 relative of p (qs_stage_pf) => 10 log10(e) x 2.98e-7 < 1.3e-6 of the margin is the argument's
            const float tf = qs_tf(x);
            acc_update(mr, arg, viol, fabs((double)tf - (-10.0 * log10(xd))), (double)qs_tf_margin(tf) - 1.3e-6, bits);
        } else if (MODE == VGL_BOUND_QS_FIX) {
            // the float32 pool loop leaves p within 84 x 2^-24 (relative) of the reference's X / (X + Y) (vgl_common.hip.h): a decided score
            // must be the exact (int)(-10 log10 p') for every p' in that interval, and so must the adjusted score
            VglDevParams Pz; Pz.adjust_by = param;
            int q, aq;
            const bool ok = qs_decide_fix(Pz, x, q, aq, param != 0.0 ? 1 : 0);
            if (!ok) { --n; continue; }
            const double d = 84.0 * 0x1p-24;
            const double t_lo = -10.0 * log10(xd * (1.0 + d)), t_hi = -10.0 * log10(xd * (1.0 - d));
            const bool good = (int)t_lo == q && (int)t_hi == q && (param == 0.0 || ((int)(t_lo + param) == aq && (int)(t_hi + param) == aq)) && t_lo > 0.0;
            acc_update(mr, arg, viol, good ? 0.0 : 1.0, 0.0, bits);
        } else if (MODE == VGL_BOUND_RCP) {
            const float r = __builtin_amdgcn_rcpf(x);
            const double ex = 1.0 / xd;
            const float exf = (float)ex;
            // result normal: within 1 ulp (2^-23 relative); where 1/x is below the normal range nothing is claimed
            if (!(fabsf(exf) >= 0x1p-126f)) { --n; continue; }
            acc_update(mr, arg, viol, fabs((double)r - ex), fabs(ex) * 0x1p-23, bits);
        } else if (MODE == VGL_BOUND_TANF) {
            const float yf = vgl_tanf_0pi(x);
            const float y2 = yf * yf;
            // poisson_fast's dy = |yf| 2^-21 + (1 + yf^2) (af 7 x 2^-24 + 2e-9): one of the seven parts of the second term is the function's (this bound), the rest the argument's
            const double bound = (double)(fabsf(yf) * 0x1p-21f) + (double)((1.0f + y2) * x * 0x1p-24f);
            acc_update(mr, arg, viol, fabs((double)yf - tan(xd)), bound, bits);
        } else if (MODE == VGL_BOUND_EXP2) {
            if (!(x >= -126.0f && x <= 8.0f)) { --n; continue; }
            const float e = __builtin_amdgcn_exp2f(x);
            const double ex = exp2(xd);
            // poisson_fast's rel_t grants 2^-19 beyond the argument's |z| 2^-22; 2^-22 of it covers the three float
            // multiplications of tt = 0.9f (1 + y2) ex
            acc_update(mr, arg, viol, fabs((double)e - ex), ex * (0x1p-19 - 0x1p-22), bits);
        }
    }
    if (n) {
        atomicAdd(&out->n, n);
        if (viol) atomicAdd(&out->viol, viol);
        const unsigned long long mb = (unsigned long long)__double_as_longlong(mr);      // non-negative doubles order like their bits
        const unsigned long long old = atomicMax(&out->max_ratio_bits, mb);
        if (mb > old) out->arg_bits = arg;                                                // diagnostic only (racy between equal maxima)
    }
}

// splitmix-style hash -> 48-bit generator states shaped like the pool loop's operands
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDULL; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ULL; x ^= x >> 33;
    return x;
}

// div_inrange(v, u) against the compiler's IEEE division for v = 1.7156 (u' - 0.5), u, u' = X 2^-48 (rng.h:72-79):
// every 8th pair draws u from the low end of the range (X < 2^k, k = 1..40), and pair 0 of each thread uses X = 1.
__global__ __launch_bounds__(256) void k_bound_div(const unsigned long long count, BoundAcc* out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long viol = 0, n = 0; uint32_t arg = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint64_t st1 = mix64(i * 2 + 1) & VGL_MASK48;
        if ((i & 7) == 7) st1 >>= 8 + (uint32_t)((i >> 3) % 40);
        if (i < stride) st1 = 1;
        if (st1 == 0) st1 = 1;
        const uint64_t st2 = (i & 1) ? lcg_next(st1) : (mix64(i * 2 + 2) & VGL_MASK48);
        const double u = u01(st1);
        const double v = 1.7156 * (u01(st2) - 0.5);
        const double q0 = div_inrange(v, u);
        const double q1 = v / u;
        ++n;
        if (__double_as_longlong(q0) != __double_as_longlong(q1)) { ++viol; arg = (uint32_t)i; }
    }
    atomicAdd(&out->n, n);
    if (viol) { atomicAdd(&out->viol, viol); out->arg_bits = arg; }
}

// quot_int24(q, recip_int24(sum)) against the compiler's IEEE float32 division: pairs 0 .. 2^24 - 1 are EVERY (sum, q) with sum, q in
// [1, 4096] x [0, 4095] (q <= sum kept), the rest pseudo-random sums up to 2^24 (every 4th one a power of two or one off it) with q <= sum
__global__ __launch_bounds__(256) void k_bound_quot(const unsigned long long count, BoundAcc* out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long viol = 0, n = 0; uint32_t arg = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        uint32_t sum, q;
        if (i < (1ULL << 24)) { sum = 1u + (uint32_t)(i >> 12); q = (uint32_t)(i & 4095u); if (q > sum) continue; }
        else {
            const uint64_t h = mix64(i);
            sum = 1u + (uint32_t)(h & 0xFFFFFFu);
            if ((i & 3) == 3) { sum = 1u << (1 + (uint32_t)((h >> 50) % 24)); sum += (uint32_t)((h >> 60) % 3) - 1u; }
            q = (uint32_t)((h >> 24) % ((uint64_t)sum + 1u));
        }
        const float fs = (float)sum, fq = (float)q;
        const float a = quot_int24(fq, recip_int24(fs)), b = fq / fs;
        ++n;
        if (__float_as_uint(a) != __float_as_uint(b)) { ++viol; arg = (uint32_t)i; }
    }
    atomicAdd(&out->n, n);
    if (viol) { atomicAdd(&out->viol, viol); out->arg_bits = arg; }
}

// poisson_fast against poisson_exact (param = the mean depth).  Attempts by class (i & 7):
//   0-3  pseudo-random st1, st2 = the generator's next state (as the kernels pair them) -- the ambiguous share of THESE is reported;
//   4    a = PI u next to the pole PI / 2, 2^-k away on either side (k = 8 .. 47);
//   5    sq tan(a) + lm next to an integer m in [0, 4 lm]: st1 = 2^48 atan((m - lm) / sq) / PI +- 2^k (k = 0 .. 31);
//   6    the same next to m = 0 (the sign test);
//   7    acceptance draws from the low end (st2 < 2^k, zero included) with a far out on the tail (large y).
// A violation: poisson_fast() does not call the attempt ambiguous and `neg`, or (when not neg) `rej`, or (when accepted) em differ.
__global__ __launch_bounds__(256) void k_bound_poisson(const unsigned long long count, const VglPois p, const double* glt, const int glt_n, const float* zt, BoundAcc* out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long viol = 0, n = 0, n_rand = 0, amb_rand = 0; uint32_t arg = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t h = mix64(i * 2 + 1), h2 = mix64(i * 2 + 2);
        const int cls = (int)(i & 7);
        uint64_t st1 = h & VGL_MASK48, st2;
        if (cls == 4) {
            const double pole = (1.5707963267948966 / VGL_PI) * 281474976710656.0;
            const uint64_t off = (h2 & VGL_MASK48) >> (8 + (uint32_t)((i >> 3) % 40));
            st1 = (uint64_t)((h >> 60) & 1 ? pole + (double)off : pole - (double)off) & VGL_MASK48;
        } else if (cls == 5 || cls == 6) {
            const double m = (cls == 6) ? 0.0 : floor((double)((h >> 48) & 0xFFFF) * (1.0 / 65536.0) * 4.0 * p.lm);
            double u = atan((m - p.lm) / p.sq) / VGL_PI;
            if (u < 0.0) u += 1.0;
            const uint64_t off = (h2 & 0xFFFFFFFFULL) >> ((uint32_t)((i >> 3) & 31));
            st1 = (uint64_t)((h >> 47) & 1 ? u * 281474976710656.0 + (double)off : u * 281474976710656.0 - (double)off) & VGL_MASK48;
        } else if (cls == 7) {
            st1 = ((1ULL << 47) - ((h & VGL_MASK48) >> (9 + (uint32_t)((i >> 3) % 38)))) & VGL_MASK48;     // a just below PI / 2 .. a quarter turn
        }
        st2 = lcg_next(st1);
        if (cls == 7) st2 = (h2 & VGL_MASK48) >> (8 + (uint32_t)((i >> 9) % 41));                       // down to 0
        bool neg, rej, amb, nege, reje; int em, eme;
        if (zt) poisson_fast<true>(p, st1, st2, glt, glt_n, zt, neg, rej, em, amb);
        else poisson_fast<false>(p, st1, st2, glt, glt_n, nullptr, neg, rej, em, amb);
        poisson_exact(p, st1, st2, glt, glt_n, nege, reje, eme);
        ++n;
        if (cls < 4) { ++n_rand; amb_rand += amb ? 1u : 0u; }
        if (!amb) {
            const bool bad = (neg != nege) || (!nege && rej != reje) || (!nege && !reje && em != eme);
            if (bad) { ++viol; arg = (uint32_t)i; }
        }
    }
    atomicAdd(&out->n, n);
    atomicAdd(&out->max_ratio_bits, amb_rand);                      // (this mode: the count of ambiguous attempts among the pseudo-random ones)
    if (viol) { atomicAdd(&out->viol, viol); out->arg_bits = arg; }
    (void)n_rand;
}


// The float32 pool loop (vgl_common.hip.h: lcg52_step, pool32_*) against the reference's float64 expressions (rng.h:72-78, 139-145) on `count`
// attempts = triples of generator states (the states of an attempt's three uniforms; the helpers do not depend on their being consecutive
// outputs, so the classes below place them freely):
//   0-2  pseudo-random states;                       3  consecutive outputs of the generator (as the loop draws them)
//   4    v placed next to the curve q = 0.27597 or q = 0.27846 for the drawn u, 2^-k 2^48 away (k = 0 .. 47);
//   5    u2 next to the squeeze 1 - 0.0331 x^4;      6  u2 next to the sure-accept bound 1 - 0.15 a2^2 x^4 (half of them with a small deviate, |xn| < 0.1: where the squeeze is the weaker of the two);
//   7    small u (2^-k, k up to 40) with v inside or next to the acceptance region: the range checks and the logarithm test's margin.
// Violations: lcg52_step differs from the 64-bit step in bits 0-51; |q_f - q| > VGL_P32_QBAND; the loop decides an attempt (no redo)
// differently from the float64 expressions; a1 w^3 off by more than 40 x 2^-24 where the loop says its value bound holds.
// out->max_ratio_bits: the attempts of classes 0-3 the loop hands to k_redo (the rate, on record).
__global__ __launch_bounds__(256) void k_bound_pool32(const unsigned long long count, const double a1, const double a2, const double sure_margin, BoundAcc* out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long viol = 0, n = 0, redo_rand = 0; uint32_t arg = 0;
    const float ga1 = (float)a1, ga2 = (float)a2;
    const float c015s = (float)((((a2 * a2) * 0.15) * (1.0 + 1e-5) + 1e-8) * 4294967296.0), sure_ms = (float)((sure_margin + 3e-7) * 4294967296.0);
    const double TWO48 = 281474976710656.0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t h1 = mix64(i * 4 + 1), h2 = mix64(i * 4 + 2), h3 = mix64(i * 4 + 3), h4 = mix64(i * 4 + 4);
        const int cls = (int)(i & 7);
        uint64_t X1 = h1 & VGL_MASK48, X2 = h2 & VGL_MASK48, X3 = h3 & VGL_MASK48;
        const uint32_t k = (uint32_t)((i >> 3) % 48);
        if (cls == 3) { X2 = lcg_next(X1); X3 = lcg_next(X2); }
        if (cls == 7) {
            X1 = (h1 & VGL_MASK48) >> (k % 41);
            const double u = (double)X1 / TWO48;
            if (u > 0.0) {
                const double vmax = 2.0 * u * sqrt(-log(u));                 // the acceptance region's edge
                const double v = vmax * ((double)(h4 & 0xFFFFF) / 524288.0 - 1.0) * 1.01;
                const double t = v / 1.7156 + 0.5;
                if (t >= 0.0 && t < 1.0) X2 = (uint64_t)(t * TWO48) & VGL_MASK48;
            }
        }
        if (cls == 4) {
            const double u = (double)X1 / TWO48, x = u - 0.449871, T = (h4 & 1) ? 0.27597 : 0.27846;
            const double disc = (0.25472 * x) * (0.25472 * x) - 4.0 * 0.196 * (x * x - T);
            if (disc >= 0.0) {
                const double y = (0.25472 * x + sqrt(disc)) / (2.0 * 0.196), v = ((h4 >> 1) & 1) ? (y - 0.386595) : -(y - 0.386595);
                const double t = v / 1.7156 + 0.5;
                if (y >= 0.386595 && t >= 0.0 && t < 1.0) {
                    const double off = (double)((h4 >> 2) & VGL_MASK48) / (double)(1ULL << k);
                    const double X = t * TWO48 + (((h4 >> 60) & 1) ? off : -off);
                    if (X >= 0.0 && X < TWO48) X2 = (uint64_t)X & VGL_MASK48;
                }
            }
        }
        // ---- float64, as the reference writes it
        const double ud = (double)X1 / TWO48, vd = 1.7156 * ((double)X2 / TWO48 - 0.5);
        const double xd = ud - 0.449871, yd = fabs(vd) + 0.386595, qd = (xd * xd) + yd * (0.19600 * yd - 0.25472 * xd);
        const double xnd = vd / ud, wd = 1.0 + a2 * xnd, xsqd = xnd * xnd;
        if (cls == 5 || cls == 6) {
            const double edge = (cls == 5) ? 1.0 - 0.0331 * (xsqd * xsqd) : 1.0 - ((0.15 * a2 * a2) * (xsqd * xsqd) + sure_margin);
            const double off = (double)(h4 & VGL_MASK48) / (double)(1ULL << k);
            const double X = edge * TWO48 + (((h4 >> 60) & 1) ? off : -off);
            if (X >= 0.0 && X < TWO48) X3 = (uint64_t)X & VGL_MASK48;
        }
        const double u2d = (double)X3 / TWO48;
        const bool rejN = (ud > 0.0) ? ((qd > 0.27597) && (qd > 0.27846 || (vd * vd) > -4.0 * log(ud) * (ud * ud))) : true;
        // ---- the loop's float32
        ++n;
        bool bad = false;
        {   // the three-instruction step against the 64-bit one
            const uint64_t s52 = X1 << 4, want = lcg_next52r(s52);
            uint32_t nl, nh;
            lcg52_step((uint32_t)s52, (uint32_t)(s52 >> 32) | (uint32_t)(h4 & 0xFFF00000u), nl, nh);       // (garbage above bit 51, as the raw form carries)
            if (nl != (uint32_t)want || ((nh ^ (uint32_t)(want >> 32)) & 0xFFFFFu)) bad = true;
        }
        const uint32_t T1 = (uint32_t)(X1 >> 16), T2 = (uint32_t)(X2 >> 16), T3 = (uint32_t)(X3 >> 16);
        if (lcg52_top32((uint32_t)(X1 << 4), (uint32_t)((X1 << 4) >> 32)) != T1) bad = true;
        const float uf = pool32_u(T1), sv = pool32_sv(T2), q = pool32_q(uf, sv);
        if (!(fabs((double)q - qd) <= (double)VGL_P32_QBAND)) bad = true;
        const bool q_lo = q > VGL_P32_QLO - VGL_P32_QBAND, q_hi = q > VGL_P32_QHI + VGL_P32_QBAND;
        const bool n_amb = q_lo && !q_hi;
        bool redo = false, slow_n = false;
        const float xn = sv * __builtin_amdgcn_rcpf(uf);
        if (n_amb) { bool und; slow_n = pool32_normal_slow(xn, uf, q, true, und); redo = und; }
        const bool acc_n = !(q_lo && (q_hi || slow_n));
        if (!redo && acc_n == rejN) bad = true;                              // decided, and not as the reference decides
        if (!redo && acc_n) {
            const float w = __builtin_fmaf(ga2, xn, 1.0f);
            const float vv = (w * w) * w;
            const float u2f = pool32_u(T3);
            const float xsq = xn * xn, x4 = xsq * xsq;
            const bool in_range = (w >= 0.5f) && (uf >= VGL_P32_UMIN);
            if (!in_range) redo = true;
            else {
                const double vvd = (wd * wd) * wd, vald = a1 * vvd;
                if (!(wd > 0.0) || !(fabs((double)(ga1 * vv) - vald) <= vald * 40.0 * 0x1p-24)) bad = true;
                const bool sqd = u2d > 1.0 - 0.0331 * (xsqd * xsqd);
                const bool logd = (u2d > 0.0) ? (log(u2d) > 0.5 * xsqd + a1 * (1.0 - vvd + log(vvd))) : false;
                const bool rejG = sqd && logd;
                const bool sure = (0x1p32f - u2f) >= __builtin_fmaf(x4, c015s, sure_ms);
                if (!sure) {
                    bool und;
                    const bool slow_g = pool32_gamma_slow(u2f, ga2 * xn, ga1, x4, true, und);
                    if (und) redo = true;
                    else if (slow_g != rejG) bad = true;
                } else if (rejG) bad = true;                                 // accepted on the sure bound, rejected by the reference
            }
        }
        if (cls < 4 && redo) ++redo_rand;
        if (bad) { ++viol; arg = (uint32_t)i; }
    }
    atomicAdd(&out->n, n);
    atomicAdd(&out->max_ratio_bits, redo_rand);
    if (viol) { atomicAdd(&out->viol, viol); out->arg_bits = arg; }
}

extern "C" __attribute__((visibility("default"))) int vgl_dbg_bound_sweep(int mode, uint32_t lo_bits, unsigned long long count, double param, double out[4]) {
    BoundAcc* d = nullptr; double* d_glt = nullptr; float* d_zt = nullptr;
    if (hipMalloc((void**)&d, sizeof(BoundAcc)) != hipSuccess) return -1;
    if (hipMemset(d, 0, sizeof(BoundAcc)) != hipSuccess) { (void)hipFree(d); return -1; }
    const dim3 g(256 * 16), b(256);
    switch (mode) {
        case VGL_BOUND_FAST_LN: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_FAST_LN>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_GAMMA_LU: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_GAMMA_LU>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_GAMMA_SERIES: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_GAMMA_SERIES>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_GAMMA_REFEXPR: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_GAMMA_REFEXPR>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_QS_TF: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_QS_TF>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_QS_FIX: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_QS_FIX>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_RCP: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_RCP>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_TANF: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_TANF>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_EXP2: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_EXP2>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_DIV10: hipLaunchKernelGGL((k_bound_sweep<VGL_BOUND_DIV10>), g, b, 0, 0, lo_bits, count, param, d); break;
        case VGL_BOUND_DIV: hipLaunchKernelGGL(k_bound_div, g, b, 0, 0, count, d); break;
        case VGL_BOUND_QUOT: hipLaunchKernelGGL(k_bound_quot, g, b, 0, 0, count, d); break;
        case VGL_BOUND_POISSON: {
            // param > 0: the exponent from the float64 expression (per-sample depths); param < 0: mean depth -param with the float32 exponent table
            const bool tab = param < 0.0;
            if (tab) param = -param;
            if (!(param >= 12.0)) { (void)hipFree(d); return -2; }
            VglPois pp;
            vgl_pois_init(&pp, param);
            const int gn = 2048;
            double* hg = (double*)malloc(sizeof(double) * gn);
            if (!hg) { (void)hipFree(d); return -1; }
            hg[0] = 0.0;
            for (int k = 1; k < gn; k++) hg[k] = vgl_gamma_ln_host((double)k);
            float* hz = (float*)malloc(sizeof(float) * gn);
            if (!hz) { free(hg); (void)hipFree(d); return -1; }
            vgl_pois_zt_host(&pp, hg, gn, hz);
            if (hipMalloc((void**)&d_glt, sizeof(double) * gn) != hipSuccess || hipMemcpy(d_glt, hg, sizeof(double) * gn, hipMemcpyHostToDevice) != hipSuccess ||
                hipMalloc((void**)&d_zt, sizeof(float) * gn) != hipSuccess || hipMemcpy(d_zt, hz, sizeof(float) * gn, hipMemcpyHostToDevice) != hipSuccess) {
                free(hg); free(hz); (void)hipFree(d_glt); (void)hipFree(d_zt); (void)hipFree(d); return -1;
            }
            free(hg); free(hz);
            hipLaunchKernelGGL(k_bound_poisson, g, b, 0, 0, count, pp, d_glt, gn, tab ? d_zt : (const float*)nullptr, d);
            break;
        }
        case VGL_BOUND_POOL32: {
            if (!(param >= 8.0)) { (void)hipFree(d); return -2; }
            const double a1 = param - 1.0 / 3.0, a2 = 1.0 / sqrt(9. * a1);                       // Gamma1Sampler_init, rng.h:155-173
            hipLaunchKernelGGL(k_bound_pool32, g, b, 0, 0, count, a1, a2, 1e-9 + 1e-14 * a1, d);
            break;
        }
        default: (void)hipFree(d); return -2;
    }
    BoundAcc h;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(d_glt); (void)hipFree(d_zt); (void)hipFree(d); return -1; }
    (void)hipFree(d_glt); (void)hipFree(d_zt); (void)hipFree(d);
    out[0] = (double)h.n; out[1] = (double)h.viol;
    out[2] = (mode == VGL_BOUND_POISSON || mode == VGL_BOUND_POOL32) ? (double)h.max_ratio_bits / (0.5 * (double)(h.n ? h.n : 1))     // ambiguous share of the pseudo-random half
                                         : __builtin_bit_cast(double, h.max_ratio_bits);
    out[3] = (double)h.arg_bits;
    return 0;
}
