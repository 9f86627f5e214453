"""Simulator: the call that replaces `simulate_record_values(sim)` (vcfgl.cpp:327) for a
whole tile of records, through the C ABI of libvcfgl_hip.so."""
import ctypes as C

import numpy as np

from . import _abi
from .params import VcfglArgs
from .tile import Tile


class VglError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvcfgl_hip error {code}: {msg}")
        self.code = code


def pack_gt(a0, a1):
    """ACGT allele indices (0..3, -1 = missing) of the two haplotypes -> one byte per sample,
    the `gt` operand of vgl_simulate_tile (true_gts_acgt_int, vcfgl.cpp:132-147)."""
    a0 = np.asarray(a0).astype(np.int16)
    a1 = np.asarray(a1).astype(np.int16)
    lo = np.where(a0 < 0, _abi.VGL_GT_MISSING, a0).astype(np.uint8)
    hi = np.where(a1 < 0, _abi.VGL_GT_MISSING, a1).astype(np.uint8)
    return (lo | (hi << 4)).astype(np.uint8)


class Simulator:
    """One vgl_ctx.  `simulate(site0, gt)` takes host numpy GT bytes [n_sites][n_samples];
    `simulate_device(site0, gt_tensor, tile, stream)` takes torch device tensors."""

    def __init__(self, args: VcfglArgs, n_samples: int, device: int = 0, max_sites_per_tile: int = 4096, hooks: bool = False):
        self.args = args
        self.n_samples = n_samples
        self.lib = _abi.load_library(hooks=hooks)         # hooks: the -DVGL_TEST_HOOKS build (tests / tools only)
        self.params, self._keep = args.to_struct(n_samples)
        self.ctx = C.c_void_p()
        rc = self.lib.vgl_ctx_create(C.byref(self.params), device, max_sites_per_tile, C.byref(self.ctx))
        if rc != _abi.VGL_OK:
            self.ctx = None
            raise VglError(rc, self.lib.vgl_last_error().decode())
        self.A = self.lib.vgl_max_alleles(C.byref(self.params))
        self.G = self.lib.vgl_max_genotypes(C.byref(self.params))
        self.max_sites_per_tile = max_sites_per_tile

    def _check(self, rc):
        if rc != _abi.VGL_OK:
            raise VglError(rc, self.lib.vgl_last_error().decode())

    def default_fields(self):
        """Every output the context can produce: QS / I16 need -addQS / -addI16 (their inputs,
        the per-base quality sums, are only accumulated when asked for)."""
        skip = {"pl_u8"}                                   # the narrow PL plane is produced on request
        if not (self.args.add_qs or self.args.add_i16):
            skip.add("qs")
        if not self.args.add_i16:
            skip.add("i16")
        return [f for f, _, _ in _abi.TILE_FIELDS if f not in skip]

    def new_tile(self, n_sites, fields=None, device=None, read_capacity=0, deviates=False):
        if fields is None:
            fields = self.default_fields()
        return Tile(n_sites, self.n_samples, self.A, self.G, fields=fields, device=device, read_capacity=read_capacity, deviates=deviates)

    def simulate(self, site0, gt, fields=None, read_capacity=0, deviates=False):
        gt = np.ascontiguousarray(gt, dtype=np.uint8)
        n_sites = gt.shape[0]
        assert gt.shape == (n_sites, self.n_samples)
        tile = self.new_tile(n_sites, fields=fields, read_capacity=read_capacity, deviates=deviates)
        self._check(self.lib.vgl_simulate_tile(self.ctx, site0, n_sites, gt.ctypes.data, tile.byref()))
        return tile

    def simulate_device(self, site0, gt, tile, stream=None):
        n_sites = gt.shape[0]
        assert tuple(gt.shape) == (n_sites, self.n_samples) and gt.is_contiguous()
        self._check(self.lib.vgl_simulate_tile_device(self.ctx, site0, n_sites, gt.data_ptr(), tile.byref(),
                                                      C.c_void_p(stream) if stream else None))

    def check(self, stream=None):
        self._check(self.lib.vgl_ctx_check(self.ctx, C.c_void_p(stream) if stream else None))

    def timing(self, enable=True):
        self._check(self.lib.vgl_ctx_timing(self.ctx, 1 if enable else 0))

    def kernel_ms(self, reset=True):
        """accumulated milliseconds and launch counts per bucket (_abi.TIMING_BUCKETS: k_depth, k_sample, k_redo, k_site, k_gl, k_siteagg)"""
        nb = len(_abi.TIMING_BUCKETS)
        ms = (C.c_double * nb)()
        n = (C.c_int64 * nb)()
        self._check(self.lib.vgl_ctx_kernel_ms(self.ctx, ms, n, nb, 1 if reset else 0))
        return list(ms), list(n)

    def info(self):
        """vgl_ctx_info(): the builds this context launches, its capacities and workspace (a dict)"""
        ci = _abi.CtxInfo()
        ci.size = C.sizeof(_abi.CtxInfo)
        self._check(self.lib.vgl_ctx_info(self.ctx, C.byref(ci)))
        return {f: getattr(ci, f) for f, _ in _abi.CtxInfo._fields_}

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vgl_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
