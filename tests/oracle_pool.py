"""TEST INFRASTRUCTURE: the CPU oracle over a long site range, sharded over worker processes, reduced to one 64-bit
checksum per site and field -- so that 10^8 evaluations of the HIP path can be compared with the oracle field by field
without shipping 6.5 GB of oracle output between processes.

checksum(site, field) = sum_j bits_j * w_j  (mod 2^64),  w_j = (2 j + 1) * 0x9E3779B97F4A7C15,
bits_j = the field's j-th element of that site, sign-extended from its storage type (float32 by its bit pattern).
`site_checksums_torch` computes the same number from device tensors.

Workers are separate interpreter processes started with `python tests/oracle_pool.py <spec.json>`: they never touch the
GPU, so the pool can be started from a process that has initialised HIP."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_K = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1


def weights(m):
    with np.errstate(over="ignore"):
        return (np.arange(m, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) * np.uint64(_K)


def site_checksums_numpy(arr):
    """arr: [n_sites, ...] int8 / int32 / float32 -> uint64 [n_sites]"""
    n = arr.shape[0]
    a = arr.reshape(n, -1)
    if a.dtype == np.float32:
        a = a.view(np.int32)
    bits = a.astype(np.int64).view(np.uint64)
    with np.errstate(over="ignore"):
        return (bits * weights(bits.shape[1])[None, :]).sum(axis=1, dtype=np.uint64)


def site_checksums_torch(t):
    """the same checksum from a torch tensor [n_sites, ...] on any device -> numpy uint64 [n_sites]"""
    import torch
    n = t.shape[0]
    a = t.reshape(n, -1)
    if a.dtype == torch.float32:
        a = a.view(torch.int32)
    w = torch.from_numpy(weights(a.shape[1]).view(np.int64)).to(a.device)
    out = torch.empty((n,), dtype=torch.int64, device=a.device)
    step = max(1, (1 << 27) // max(a.shape[1], 1))                 # <= 1 GiB of int64 temporaries per chunk
    for s0 in range(0, n, step):
        out[s0:s0 + step] = (a[s0:s0 + step].to(torch.int64) * w[None, :]).sum(dim=1)
    return out.cpu().numpy().view(np.uint64)


def field_view(name, arr):
    """what of a field is compared: all of it (round 6: I16's tail-distance fields 12-15 are counter-addressed in tile mode, k_tail)"""
    return arr


def _gt(kind, site0, n, n_samples):
    import synth
    if kind == "homref":
        return np.zeros((n, n_samples), dtype=np.uint8)
    if kind == "acgt":                                              # multi-allelic, 2 % missing calls
        return synth.acgt_range(site0, n, n_samples)
    return synth.binary_sites(site0, n, n_samples)


def _worker(spec):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from vcfgl_amd import VcfglArgs
    args = VcfglArgs(**spec["args"])
    N, fields = spec["n_samples"], spec["fields"]
    o = oracle_lib.Oracle(args, N)
    site0, n = spec["site0"], spec["n_sites"]
    out = np.zeros((n, len(fields)), dtype=np.uint64)
    chunk = spec.get("chunk", 128)
    for s0 in range(0, n, chunk):
        m = min(chunk, n - s0)
        t = o.simulate(site0 + s0, _gt(spec["gt"], site0 + s0, m, N), fields=fields)
        for k, f in enumerate(fields):
            out[s0:s0 + m, k] = site_checksums_numpy(field_view(f, t.numpy(f)))
    np.save(spec["out"], out)


def oracle_site_checksums(args_dict, n_samples, site0, n_sites, fields, gt="binary", workers=None):
    """uint64 [n_sites, len(fields)]: the oracle's per-site checksums of `fields` over sites [site0, site0 + n_sites)."""
    if workers is None:
        workers = max(1, min(os.cpu_count() or 1, 32))
    workers = min(workers, max(1, n_sites // 16))
    per = -(-n_sites // workers)
    procs, outs = [], []
    with tempfile.TemporaryDirectory() as td:
        for w in range(workers):
            b = w * per
            n = min(per, n_sites - b)
            if n <= 0:
                break
            spec = {"args": args_dict, "n_samples": n_samples, "site0": site0 + b, "n_sites": n, "fields": list(fields), "gt": gt,
                    "out": os.path.join(td, f"w{w}.npy")}
            sp = os.path.join(td, f"w{w}.json")
            with open(sp, "w") as fh:
                json.dump(spec, fh)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), sp]))
            outs.append(spec["out"])
        for p in procs:
            rc = p.wait(timeout=900)
            assert rc == 0, f"oracle worker exited with {rc}"
        return np.concatenate([np.load(o) for o in outs], axis=0)


if __name__ == "__main__":
    _worker(json.load(open(sys.argv[1])))
