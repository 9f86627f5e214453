"""Synthetic true-genotype tiles: msprime-style binary sites whose
derived-allele count follows a (piecewise) 1/x site-frequency spectrum, haplotypes iid
Bernoulli(k/2N) from a counter hash.  Integer arithmetic only, so the numpy generator (CPU
legs, tests) and the torch generator (device-resident bench inputs) give identical bytes."""
import numpy as np

GEN_SEED = 20251003
_C1, _C2 = 0xFF51AFD7ED558CCD, 0xC4CEB9FE1A85EC53
_K1, _K2 = 0x9E3779B97F4A7C15, 0xD6E8FEB86659FD93


def _mix(x):
    x = (x ^ (x >> np.uint64(33))) * np.uint64(_C1)
    x = (x ^ (x >> np.uint64(33))) * np.uint64(_C2)
    return x ^ (x >> np.uint64(33))


def binary_sites(site0, n_sites, n_samples, seed=GEN_SEED):
    """uint8 [n_sites][n_samples]: (a1 << 4) | a0 with REF->A(0), ALT->C(1) (vcfgl.cpp:103-128)."""
    twoN = 2 * n_samples
    nbits = max(int(twoN - 1).bit_length(), 1)
    with np.errstate(over="ignore"):
        sites = (np.arange(site0, site0 + n_sites, dtype=np.uint64) + np.uint64(1)) * np.uint64(_K1) + np.uint64(seed)
        h = _mix(sites)
        j = (h >> np.uint64(40)) % np.uint64(nbits)                       # octave: uniform => P(k) ~ 1/k
        k = (np.uint64(1) << j) + ((h & np.uint64(0xFFFFFFFF)) & ((np.uint64(1) << j) - np.uint64(1)))
        k = np.minimum(k, np.uint64(twoN - 1))
        idx = sites[:, None] * np.uint64(_K2) + np.arange(twoN, dtype=np.uint64)[None, :]
        hap = ((_mix(idx) >> np.uint64(20)) % np.uint64(twoN)) < k[:, None]
    a0 = hap[:, 0::2].astype(np.uint8)
    a1 = hap[:, 1::2].astype(np.uint8)
    return (a0 | (a1 << 4)).astype(np.uint8)


def binary_sites_torch(site0, n_sites, n_samples, device, seed=GEN_SEED):
    """Same bytes as binary_sites(), generated on `device` (int64 arithmetic wraps like uint64)."""
    import torch

    def s64(v):
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def mix(x):
        x = (x ^ lsr(x, 33)) * s64(_C1)
        x = (x ^ lsr(x, 33)) * s64(_C2)
        return x ^ lsr(x, 33)

    twoN = 2 * n_samples
    nbits = max(int(twoN - 1).bit_length(), 1)
    sites = (torch.arange(site0, site0 + n_sites, dtype=torch.int64, device=device) + 1) * s64(_K1) + seed
    h = mix(sites)
    j = lsr(h, 40) % nbits
    one = torch.ones_like(j)
    k = (one << j) + ((h & 0xFFFFFFFF) & ((one << j) - 1))
    k = torch.clamp(k, max=twoN - 1)
    idx = sites[:, None] * s64(_K2) + torch.arange(twoN, dtype=torch.int64, device=device)[None, :]
    hap = (lsr(mix(idx), 20) % twoN) < k[:, None]
    a0 = hap[:, 0::2].to(torch.uint8)
    a1 = hap[:, 1::2].to(torch.uint8)
    return (a0 | (a1 << 4)).contiguous()


def acgt_sites(n_sites, n_samples, seed=7, missing=0.0, n_alleles=4):
    """Multi-allelic ACGT-space genotypes with optional missing calls (source 1 inputs)."""
    rng = np.random.default_rng(seed)
    major = rng.integers(0, 4, size=(n_sites, 1))
    alt = rng.integers(0, n_alleles, size=(n_sites, n_samples, 2))
    pick = rng.random((n_sites, n_samples, 2)) < rng.random((n_sites, 1, 1)) * 0.6
    al = np.where(pick, alt, major[:, :, None]).astype(np.uint8)
    if missing > 0:
        al = np.where(rng.random((n_sites, n_samples, 2)) < missing, 0xF, al).astype(np.uint8)
    return (al[:, :, 0] | (al[:, :, 1] << 4)).astype(np.uint8)


def acgt_range(site0, n_sites, n_samples, missing=0.02, block=128):
    """acgt_sites() addressable by absolute site index: sites [128 b, 128 (b + 1)) come from the generator seeded with b, so any
    range, however it is chunked, sees the same bytes."""
    out = np.empty((n_sites, n_samples), dtype=np.uint8)
    s = site0
    while s < site0 + n_sites:
        b = s // block
        blk = acgt_sites(block, n_samples, seed=1000 + b, missing=missing)
        lo, hi = s - b * block, min(block, site0 + n_sites - b * block)
        out[s - site0:s - site0 + hi - lo] = blk[lo:hi]
        s += hi - lo
    return out
