"""The per-site checksum used for oracle parity at scale (tests/oracle_pool.py): numpy and torch give the same number,
a flipped bit anywhere changes it, and the sharded worker pool equals the oracle called directly."""
import dataclasses

import numpy as np
import torch

import oracle_pool
import synth
from vcfgl_amd import VcfglArgs, _abi


def test_checksum_numpy_equals_torch_and_detects_a_flipped_bit():
    rng = np.random.default_rng(5)
    for arr in (rng.integers(-2 ** 31, 2 ** 31 - 1, size=(7, 15, 33), dtype=np.int64).astype(np.int32),
                rng.standard_normal((5, 10, 4)).astype(np.float32),
                rng.integers(-1, 5, size=(9, 5)).astype(np.int8)):
        a = oracle_pool.site_checksums_numpy(arr)
        b = oracle_pool.site_checksums_torch(torch.from_numpy(arr))
        assert a.dtype == np.uint64 and np.array_equal(a, b)
        flipped = arr.copy()
        flat = flipped.reshape(flipped.shape[0], -1)
        if flat.dtype == np.float32:
            flat.view(np.int32)[2, -1] ^= 1
        else:
            flat[2, -1] ^= 1
        c = oracle_pool.site_checksums_numpy(flipped)
        assert c[2] != a[2] and np.array_equal(np.delete(c, 2), np.delete(a, 2))


def test_pool_equals_direct_oracle(oracle):
    a = VcfglArgs(seed=42, depth=6.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_pl=1)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    N, S, site0 = 20, 70, 12345
    fields = ["site_status", "alleles2acgt", "fmt_dp", "gl", "pl"]
    got = oracle_pool.oracle_site_checksums(dataclasses.asdict(a), N, site0, S, fields, workers=3)
    t = oracle.Oracle(a, N).simulate(site0, synth.binary_sites(site0, S, N), fields=fields)
    want = np.stack([oracle_pool.site_checksums_numpy(t.numpy(f)) for f in fields], axis=1)
    assert np.array_equal(got, want)
