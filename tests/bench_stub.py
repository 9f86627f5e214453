"""Stand-in of vcfgl_amd.Simulator for the CPU rehearsal of bench.py's N-rank path (BENCH_TEST_STUB=1, tests/test_bench_world8_cpu.py).
It simulates NOTHING: it fills the tile's status / allele-count arrays with a fixed pattern of the absolute site index (so that the
record index gather, the counters and the packed-record gather have defined, checkable contents) and leaves the tag planes as
they are.  Test infrastructure only -- bench.py never reaches it without the environment variable."""
import ctypes as C

import numpy as np

KERNELS = 6


def _arr(addr, n, ctype, dtype):
    return np.frombuffer((ctype * n).from_address(addr), dtype=dtype)


class StubSimulator:
    A, G = 5, 15

    def __init__(self, args, n_samples, device=0, max_sites_per_tile=4096):
        self.args, self.n_samples, self.ctx, self.lib = args, n_samples, None, self
        self._ms, self._n, self._timing = [0.0] * KERNELS, [0] * KERNELS, False

    # the one entry point bench.py's step() calls
    def vgl_simulate_tile_device(self, ctx, site0, n_sites, gt_ptr, tile_ref, stream):
        t = tile_ref._obj
        site = np.arange(site0, site0 + n_sites, dtype=np.int64)
        _arr(t.site_status, n_sites, C.c_int32, np.int32)[:] = np.where(site % 7 == 3, -3, 0)          # every 7th site "simulated invariant"
        _arr(t.n_alleles, n_sites, C.c_int32, np.int32)[:] = 2 + (site % 3)
        if t.alleles2acgt:
            _arr(t.alleles2acgt, n_sites * 5, C.c_int8, np.int8)[:] = np.tile(np.array([0, 1, 2, 3, -1], dtype=np.int8), n_sites)
        if t.fmt_dp:
            _arr(t.fmt_dp, n_sites * self.n_samples, C.c_int32, np.int32)[:] = np.repeat(site % 50, self.n_samples).astype(np.int32)
        if t.gl:
            _arr(t.gl, n_sites * self.G * self.n_samples, C.c_float, np.float32)[:] = -1.5
        if self._timing:
            for k, ms in enumerate((0.4, 12.0, 0.07, 0.01, 2.2, 0.0)):
                self._ms[k] += ms
                self._n[k] += 1 if ms else 0
        return 0

    def _check(self, rc):
        assert rc == 0

    def info(self):
        return dict(fused=0, fused_split=0, sample_lean=2, depth_mode=1, gl_sort=2, gl_wpb=8, read_cap=72, pool_cap=1472, workspace_bytes=0)

    def check(self, stream=None):
        pass

    def timing(self, enable=True):
        self._timing = bool(enable)

    def kernel_ms(self, reset=True):
        out = (list(self._ms), list(self._n))
        if reset:
            self._ms, self._n = [0.0] * KERNELS, [0] * KERNELS
        return out

    def close(self):
        pass
