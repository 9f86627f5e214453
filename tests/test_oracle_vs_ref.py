"""Checks the oracle's samplers and tables against the reference's OWN code (rng.h +
shared.cpp compiled from /root/reference into oracle/_ref by oracle/Makefile).  Skipped when
oracle/_ref has not been built (the reference checkout is absent and no prebuilt copy travelled)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = os.path.join(ROOT, "oracle", "_ref", "libvgl_ref_stdbeta.so")
R48 = os.path.join(ROOT, "oracle", "_ref", "libvgl_ref_rand48beta.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(STD) and os.path.exists(R48)), reason="oracle/_ref not built")


def _ref(path):
    r = C.CDLL(path)
    for f in ("ref_rng0", "ref_rng1", "ref_rng2"):
        getattr(r, f).restype = C.c_double
    r.ref_gamma_ln.restype = C.c_double
    r.ref_gamma_ln.argtypes = [C.c_double]
    r.ref_q2log10gl.restype = C.c_double
    r.ref_poisson.argtypes = [C.c_double, C.c_int, C.c_void_p]
    r.ref_poisson_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    r.ref_beta.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p]
    return r


@pytest.mark.parametrize("seed", [42, 0, -1, 123456789, -7, 2 ** 31 - 1])
def test_rand48_streams(oracle, seed):
    """glibc drand48/erand48 == closed-form LCG; all three reference streams start equal."""
    r, o = _ref(STD), oracle.lib()
    r.ref_seed(seed)
    x = o.vgl_oracle_rand48_seed(seed)
    for _ in range(2000):
        x = o.vgl_oracle_rand48_jump(x, 1)
        u = x / 2.0 ** 48
        assert r.ref_rng0() == u and r.ref_rng1() == u and r.ref_rng2() == u


def test_rand48_jump(oracle):
    o = oracle.lib()
    x0 = o.vgl_oracle_rand48_seed(42)
    x = x0
    for i in range(1, 5000):
        x = o.vgl_oracle_rand48_jump(x, 1)
        if i % 97 == 0:
            assert x == o.vgl_oracle_rand48_jump(x0, i)
    big = 3 * 2 ** 40 + 12345
    assert o.vgl_oracle_rand48_jump(o.vgl_oracle_rand48_jump(x0, big), 2 ** 48 - big) == x0  # full period


def test_gamma_ln(oracle):
    r, o = _ref(STD), oracle.lib()
    for v in np.linspace(0.5, 700, 20000):
        assert o.vgl_oracle_gamma_ln(v) == r.ref_gamma_ln(v)


def test_qscore_lut_all_entries(oracle):
    """shared.cpp:110-114 (3 x 257 seven-digit literals) == the oracle's regenerated table."""
    r, o = _ref(STD), oracle.lib()
    for row in range(3):
        for q in range(257):
            assert o.vgl_oracle_q2gl(row, q) == r.ref_q2log10gl(row, q), (row, q)


@pytest.mark.parametrize("lam", [0, 0.1, 2, 5, 10, 11.99, 12, 20, 30, 100, 499.5])
def test_poisson(oracle, lam):
    r, o = _ref(STD), oracle.lib()
    n = 20000
    a, b = np.zeros(n, np.int32), np.zeros(n, np.int32)
    r.ref_seed(42)
    r.ref_poisson(lam, n, a.ctypes.data)
    st = C.c_uint64(o.vgl_oracle_rand48_seed(42))
    o.vgl_oracle_poisson_draws(lam, C.byref(st), n, b.ctypes.data)
    assert (a == b).all()
    if lam == 100:
        assert a[0] == 85          # DP of the first record of test/reference/test18


@pytest.mark.parametrize("mv", [(0.01, 1e-5), (0.024, 1e-5), (0.4, 0.1), (0.2, 0.1), (0.001, 1e-7)])
def test_beta_samplers(oracle, mv):
    m, v = mv
    o = oracle.lib()
    n = 20000
    a, b = np.zeros(n), np.zeros(n)
    rs = _ref(STD)
    rs.ref_beta(m, v, 42, n, a.ctypes.data)
    o.vgl_oracle_beta_std_draws(m, v, 42, n, b.ctypes.data)
    assert (a == b).all(), "std::mt19937 / std::gamma_distribution restatement"
    rr = _ref(R48)
    rr.ref_seed(42)
    rr.ref_beta(m, v, 42, n, a.ctypes.data)
    st = C.c_uint64(o.vgl_oracle_rand48_seed(42))
    o.vgl_oracle_beta_rand48_draws(m, v, C.byref(st), n, b.ctypes.data)
    assert (a == b).all(), "rand48 beta (reference built with -D__USE_STD_BETA__=0)"


def test_doc_known_answers(oracle):
    """doc/error_qs.MD:98-101 of the reference: beta(0.4, 0.1) seed 42."""
    o = oracle.lib()
    b = np.zeros(4)
    o.vgl_oracle_beta_std_draws(0.4, 0.1, 42, 4, b.ctypes.data)
    assert [f"{x:.6f}" for x in b] == ["0.621323", "0.960007", "0.525474", "0.002457"]


def test_host_program_qs_to_errprob_table():
    """QS_TO_ERRPROB (shared.h:493 over the literals of shared.cpp:31), as the host program restates it for the
    gl_error_prob lines of -printGlError: equal to the reference's table for every quality score."""
    import subprocess
    binp = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")
    if not os.path.exists(binp):
        pytest.skip("vcfgl_hip not built")
    r = _ref(STD)
    r.ref_qs_to_errprob.restype = C.c_double
    qs = list(range(0, 80))
    out = subprocess.run([binp, "--qs-to-errprob"] + [str(q) for q in qs], capture_output=True, text=True, check=True).stdout.split()
    assert [float(x) for x in out] == [r.ref_qs_to_errprob(q) for q in qs]
