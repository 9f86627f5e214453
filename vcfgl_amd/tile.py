"""Tile buffers: the structure-of-arrays outputs of one vgl_simulate_tile call
(include/vcfgl_hip.h, `vgl_tile_out`), as numpy (host) or torch (device) arrays."""
import ctypes as C

import numpy as np

from . import _abi


def _shape(kind, n_sites, n_samples, A, G):
    return {
        "site": (n_sites,), "site5": (n_sites, 5), "siteA": (n_sites, A), "site16": (n_sites, 16),
        "eval": (n_sites, n_samples), "planeG": (n_sites, G, n_samples), "planeA": (n_sites, A, n_samples),
    }[kind]


class Tile:
    """Owns one array per requested field; `.struct` is the vgl_tile_out to pass through the C ABI."""

    ALWAYS = ("site_status", "n_alleles", "n_alleles_obs", "alleles2acgt")

    def __init__(self, n_sites, n_samples, max_alleles, max_genotypes, fields=None, device=None, read_capacity=0, deviates=False):
        self.n_sites, self.n_samples, self.A, self.G = n_sites, n_samples, max_alleles, max_genotypes
        self.device = device
        want = set(self.ALWAYS) | set(fields if fields is not None else [f for f, _, _ in _abi.TILE_FIELDS])
        self.arrays = {}
        self.struct = _abi.TileOut()
        for name, dtype, kind in _abi.TILE_FIELDS:
            if name not in want:
                continue
            shape = _shape(kind, n_sites, n_samples, self.A, self.G)
            arr = self._alloc(shape, dtype)
            self.arrays[name] = arr
            setattr(self.struct, name, self._ptr(arr))
        if read_capacity:
            arr = self._alloc((read_capacity, n_sites, n_samples), "uint8")
            self.arrays["reads"] = arr
            self.struct.reads = self._ptr(arr)
            self.struct.read_capacity = read_capacity
            if deviates:                                  # ABI 2: error_prob_forQs_i of every read (error_qs 2)
                arr = self._alloc((read_capacity, n_sites, n_samples), "float64")
                self.arrays["read_errp"] = arr
                self.struct.read_errp = self._ptr(arr)
        if deviates:                                      # ABI 2: base_pick_error_prob of every site (error_qs 1)
            arr = self._alloc((n_sites,), "float64")
            self.arrays["site_pick_err"] = arr
            self.struct.site_pick_err = self._ptr(arr)

    def _alloc(self, shape, dtype):
        if self.device is None:
            return np.zeros(shape, dtype=dtype)
        import torch
        return torch.zeros(shape, dtype=getattr(torch, dtype), device=self.device)

    def _ptr(self, arr):
        return arr.ctypes.data if self.device is None else arr.data_ptr()

    def __getitem__(self, name):
        return self.arrays[name]

    def numpy(self, name):
        a = self.arrays[name]
        return a if self.device is None else a.cpu().numpy()

    def site_records(self, name, site, n_values):
        """VGL_LAYOUT_SAMPLE_MAJOR: the [n_samples][n_values] array of one site of a multi-valued FORMAT field (what the
        reference keeps in simRecord::gl_arr etc.), n_values = nGenotypes(site) or n_alleles(site)"""
        a = self.numpy(name)
        return a[site].reshape(-1)[: self.n_samples * n_values].reshape(self.n_samples, n_values)

    def byref(self):
        return C.byref(self.struct)
