"""ctypes wrapper of oracle/libvgl_oracle.so -- TEST INFRASTRUCTURE (the checker).
Uses the same vgl_params / vgl_tile_out structs as the product's C ABI."""
import ctypes as C
import os

import numpy as np

from vcfgl_amd import _abi
from vcfgl_amd.tile import Tile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(ROOT, "oracle", "libvgl_oracle.so"))
        _lib.vgl_oracle_create.argtypes = [C.POINTER(_abi.Params), C.POINTER(C.c_void_p)]
        _lib.vgl_oracle_destroy.argtypes = [C.c_void_p]
        _lib.vgl_oracle_simulate.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(_abi.TileOut)]
        _lib.vgl_oracle_last_error.restype = C.c_char_p
        _lib.vgl_oracle_default_layout.argtypes = [C.POINTER(_abi.Params), C.POINTER(_abi.RngLayout)]
        _lib.vgl_oracle_rand48_seed.restype = C.c_uint64
        _lib.vgl_oracle_rand48_seed.argtypes = [C.c_int32]
        _lib.vgl_oracle_rand48_jump.restype = C.c_uint64
        _lib.vgl_oracle_rand48_jump.argtypes = [C.c_uint64, C.c_uint64]
        _lib.vgl_oracle_gamma_ln.restype = C.c_double
        _lib.vgl_oracle_gamma_ln.argtypes = [C.c_double]
        _lib.vgl_oracle_q2gl.restype = C.c_double
        _lib.vgl_oracle_q2gl.argtypes = [C.c_int, C.c_int]
        _lib.vgl_oracle_poisson_draws.argtypes = [C.c_double, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]
        _lib.vgl_oracle_beta_rand48_draws.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_uint64), C.c_int, C.c_void_p]
        _lib.vgl_oracle_beta_std_draws.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_int, C.c_void_p]
        _lib.vgl_oracle_errmod_cal.argtypes = [C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


class Oracle:
    def __init__(self, args, n_samples):
        self.args, self.n_samples = args, n_samples
        self.params, self._keep = args.to_struct(n_samples)
        self.h = C.c_void_p()
        rc = lib().vgl_oracle_create(C.byref(self.params), C.byref(self.h))
        if rc != 0:
            raise OracleError(rc, lib().vgl_oracle_last_error().decode())
        self.A, self.G = args.max_alleles, args.max_genotypes

    def simulate(self, site0, gt, fields=None, read_capacity=0, deviates=False):
        gt = np.ascontiguousarray(gt, dtype=np.uint8)
        n_sites = gt.shape[0]
        tile = Tile(n_sites, self.n_samples, self.A, self.G, fields=fields, read_capacity=read_capacity, deviates=deviates)
        rc = lib().vgl_oracle_simulate(self.h, site0, n_sites, gt.ctypes.data, tile.byref())
        if rc != 0:
            raise OracleError(rc, lib().vgl_oracle_last_error().decode())
        return tile

    def close(self):
        if self.h:
            lib().vgl_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
