"""The error bounds behind every float32 decision of the HIP kernels, asserted on the device.

The kernels decide the reference's double-precision comparisons (rejection tests, floor of a scaled tan,
(int)(-10 log10 p)) from float32 hardware transcendentals and fall back to the exact double expression inside an
explicit error band (vgl_common.hip.h).  Every result is exact only if those bands really contain the error of
v_log_f32 / v_exp_f32 / v_rcp_f32 / ocml tanf on this toolchain and part.  vgl_bounds.hip sweeps EVERY float32 argument
of each function's range of use through the same inline helpers the kernels call and counts the arguments whose
error exceeds the bound the kernels use (less the share reserved for rounding the double argument to float).
Zero violations are required; the worst ratio error / bound is printed so that the margin is on record."""
import ctypes as C
import struct

import pytest

from vcfgl_amd import _abi

pytestmark = pytest.mark.gpu

FAST_LN, GAMMA_LU, GAMMA_SERIES, GAMMA_REFEXPR, QS_TF, RCP, TANF, EXP2, DIV, QUOT, DIV10, POISSON, POOL32, QS_FIX = range(14)


def f2b(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def sweep(mode, lo_bits, hi_bits, param=0.0, count=None):
    lib = _abi.load_library(hooks=True)                   # vgl_bounds.hip is part of the -DVGL_TEST_HOOKS build only
    lib.vgl_dbg_bound_sweep.argtypes = [C.c_int, C.c_uint32, C.c_ulonglong, C.c_double, C.POINTER(C.c_double)]
    out = (C.c_double * 4)()
    n = count if count is not None else hi_bits - lo_bits + 1
    assert lib.vgl_dbg_bound_sweep(mode, lo_bits, n, param, out) == 0
    return {"n": int(out[0]), "violations": int(out[1]), "max_ratio": out[2], "arg_bits": int(out[3])}


def check(name, r, n_min):
    print(f"{name}: {r['n']} arguments, worst error/bound {r['max_ratio']:.4f} at bits 0x{r['arg_bits']:08x}, violations {r['violations']}")
    assert r["n"] >= n_min, (name, r)
    assert r["violations"] == 0, (name, r)
    assert r["max_ratio"] <= 1.0, (name, r)


def test_fast_ln_bound_every_float_in_unit_interval():
    """normal_slow_test: |fast_ln(uf) - ln(u)| <= |l| 2^-21 + 2^-22 for every float in [2^-49, 1] (u = X 2^-48 >= 2^-48)"""
    check("fast_ln", sweep(FAST_LN, f2b(2.0 ** -49), f2b(1.0)), 4e8)


def test_gamma_test_log_bound_every_float_in_unit_interval():
    """gamma_slow_test: |lu - ln(u)| <= |lu| 2^-20 + 2^-21"""
    check("gamma_test_lu", sweep(GAMMA_LU, f2b(2.0 ** -49), f2b(1.0)), 4e8)


@pytest.mark.parametrize("a1", [0.7, 1.2266666, 9.5676, 979.3, 31000.0])
def test_gamma_series_bound_every_float_s(a1):
    """gamma_slow_test: the float32 series 3 a1 s^4 P(s) against the analytic series and against the reference's own double
    expression -(0.5 x^2 + a1 (1 - v + log v)) for every float |s| <= 0.3333, within 4e-6 g (+ 1e-10 for the reference's
    cancellation); a1 spans --error-qs shapes from alpha' in (1, 2) to beta ~ 3e4"""
    hi = f2b(0.3333)
    for mode, name in ((GAMMA_SERIES, "series"), (GAMMA_REFEXPR, "refexpr")):
        check(f"gamma {name} a1={a1} s>0", sweep(mode, 1, hi, a1), 1e9)
        check(f"gamma {name} a1={a1} s<0", sweep(mode, 0x80000001, 0x80000000 | hi, a1), 1e9)


def test_qs_tf_bound_every_float_probability():
    """qs_decide_pf: |tf - (-10 log10 p)| <= tf 2^-19 + 4e-6 for every float pf in (1e-37, 1)"""
    check("qs_tf", sweep(QS_TF, f2b(1.0e-37) + 1, f2b(1.0) - 1), 9e8)


def test_v_rcp_f32_one_ulp_every_normal_float():
    """qs_stage_pf: v_rcp_f32 within 1 ulp wherever 1/x is a normal float"""
    check("v_rcp_f32", sweep(RCP, f2b(2.0 ** -126), f2b(2.0 ** 126)), 2e9)


def test_tanf_bound_every_float_up_to_pi():
    """poisson_attempt: |tanf(af) - tan(af)| <= |y| 2^-21 + (1 + y^2) af 2^-24 on (2^-50, (float)3.141592654]"""
    check("tanf", sweep(TANF, f2b(2.0 ** -50), f2b(3.141592654)), 4e8)


def test_v_exp_f32_bound():
    """poisson_attempt: v_exp_f32 relative error <= 2^-19 - 2^-22 on [-126, 8]"""
    check("v_exp_f32 x>0", sweep(EXP2, 0, f2b(8.0)), 1e9)
    check("v_exp_f32 x<0", sweep(EXP2, 0x80000000, f2b(-126.0)), 1e9)


def test_div_inrange_equals_ieee_division():
    """k_sample<2>: the division sequence without v_div_scale / v_div_fixup gives the IEEE quotient of v / u for the
    operands of the ratio-of-uniforms normal sampler (u = X 2^-48 incl. the smallest, v = 1.7156 (u' - 0.5))"""
    r = sweep(DIV, 0, 0, count=4_000_000_000)
    check("div_inrange", r, 4e9)


def test_quot_int24_equals_ieee_float_division():
    """k_siteagg (INFO/QS): (float)q / sum for integers q <= sum <= 2^24 from one double reciprocal per sum is the IEEE float32 quotient --
    every pair up to 4096, then 4e9 pseudo-random pairs up to 2^24 incl. sums at and next to powers of two"""
    r = sweep(QUOT, 0, 0, count=(1 << 24) + 4_000_000_000)
    check("quot_int24", r, 4e9)


def test_div10_equals_the_reference_division_for_every_nonnegative_float():
    """k_gl, GL model 1: GL = -cost / 10 (gl_methods.cpp:285) as x RN(1/10) + one residual correction: the same float32 as
    (float)((double)x / 10.0) for EVERY non-negative finite float32 (zero, subnormals and the largest value included)"""
    r = sweep(DIV10, 0, f2b(3.4028234663852886e38))
    print(f"div10: {r['n']} arguments, violations {r['violations']}")
    assert r["n"] >= 2.1e9 and r["violations"] == 0, r


@pytest.mark.parametrize("table", [False, True])
@pytest.mark.parametrize("depth", [12.0, 20.0, 30.5, 100.0, 1000.25, 40000.0])
def test_poisson_attempt_float32_decisions_equal_the_exact_ones(depth, table):
    """k_depth / poisson_attempt: the rejection method's attempt (rng.h:302-309) decided in float32 -- sign of sq tan(PI u) + lm, its floor,
    the acceptance test, and the two shortcuts (certainly negative; beyond e_hi certainly rejected) -- equals the float64 evaluation
    wherever poisson_fast() does not ask for it: 2^32 attempts per mean depth (half pseudo-random, half next to the pole of tan, next to
    integer values of the scaled tangent, next to zero, and with the smallest acceptance draws).  The ambiguous share of the
    pseudo-random attempts is what sends a wavefront of k_depth into the float64 path: on record, and bounded for the usual depths.
    table: the acceptance bound's exponent from the float32 table of the one-depth build (k_depth<true>) instead of the float64 expression."""
    r = sweep(POISSON, 0, 0, param=-depth if table else depth, count=1 << 32)
    print(f"poisson depth {depth}{' (exponent table)' if table else ''}: {r['n']} attempts, violations {r['violations']} (last at index {r['arg_bits']}), ambiguous share of random attempts {r['max_ratio']:.3e}")
    assert r["n"] == 1 << 32 and r["violations"] == 0, r
    if depth <= 100.0:
        assert r["max_ratio"] < 2e-4, r


@pytest.mark.parametrize("alpha", [8.0, 9.889, 979.011, 55.5, 2.0e5])
def test_float32_pool_loop_decisions_equal_the_float64_ones(alpha):
    """k_sample<2>'s float32 pool loop (round 5; vgl_common.hip.h: lcg52_step, pool32_*): on 2^32 attempts per beta shape parameter -- three
    eighths pseudo-random states, one eighth consecutive generator outputs, the rest placed next to the curves q = 0.27597 / 0.27846,
    next to the squeeze and the sure-accept bound, and at small u around the acceptance region's edge -- the three-instruction generator
    step equals the 64-bit one, q_f lies within its band of q, every attempt the loop decides itself (normal attempt accepted /
    rejected, gamma step accepted / rejected) is decided as rng.h:72-78 / :139-145 decide it in float64, and a1 w^3 is within
    40 x 2^-24 wherever the loop uses it.  9.889 / 979.011: config C3's shapes (--error-rate 0.01 --beta-variance 1e-5).
    The share of pseudo-random attempts handed to k_redo is on record and bounded."""
    r = sweep(POOL32, 0, 0, param=alpha, count=1 << 32)
    print(f"pool32 alpha {alpha}: {r['n']} attempts, violations {r['violations']} (last at index {r['arg_bits']}), share of random attempts sent to k_redo {r['max_ratio']:.3e}")
    assert r["n"] == 1 << 32 and r["violations"] == 0, r
    # (alpha 8: 8.5e-4 -- nearly all of it gamma steps with |a2 xn| > 1/3 whose sure-accept bound does not hold: the range of the
    # bounded test's series, as in the float64 loop; C3's shapes: 3e-4 and 1e-5)
    assert r["max_ratio"] < 1.5e-3, r


@pytest.mark.parametrize("adjust_by", [0.0, 0.499, 3.25])
def test_fixed_point_score_decision_every_float_probability(adjust_by):
    """qs_decide_fix (round 5: the finish block of k_sample<2>'s two-byte-item loop): for EVERY float32 p in (1e-37, 1) that it decides, the
    score -- and the adjusted score (int)(tf + adjust_by) -- is the exact one for every true value within the float32 pool loop's
    84 x 2^-24 of p; the undecided share is on record."""
    r = sweep(QS_FIX, f2b(1.0e-37), f2b(1.0), param=adjust_by)
    n_all = f2b(1.0) - f2b(1.0e-37) + 1
    print(f"qs_fix adjust_by {adjust_by}: {r['n']} of {n_all} arguments decided, violations {r['violations']}")
    assert r["violations"] == 0, r
    assert r["n"] > 0.02 * n_all          # (p >= 4e-7, i.e. tf < 64, is 2.3 % of the float32 values below 1; all of them but a band of 3.7e-4 per score)
