"""vgl_pack_plan_device / vgl_pack_records_device (csrc/vgl_pack.hip, ABI 6) against the torch formulation of shard.pack_records on the same tile:
the kept sites of a tile as variable-length records -- skipped sites dropped, nG(site) / nA(site) planes per FORMAT tag -- bit for bit."""
import numpy as np
import pytest
import torch

from vcfgl_amd import _abi, shard

pytestmark = pytest.mark.gpu


def random_tile(S, N, A, G, seed, p_skip=0.2, fields=None):
    rng = np.random.default_rng(seed)
    na = rng.integers(2, A + 1, size=S).astype(np.int32)
    status = np.where(rng.random(S) < p_skip, rng.choice([-1, -2, -3], size=S), 0).astype(np.int32)
    t = {"site_status": status, "n_alleles": na}
    shapes = {"site": (S,), "site5": (S, 5), "siteA": (S, A), "site16": (S, 16), "eval": (S, N), "planeG": (S, G, N), "planeA": (S, A, N)}
    for name, dt, kind in _abi.TILE_FIELDS:
        if name in t or (fields is not None and name not in fields):
            continue
        shp = shapes[kind]
        if dt.startswith("float"):
            t[name] = rng.standard_normal(shp).astype(dt)
        else:
            info = np.iinfo(dt)
            t[name] = rng.integers(max(info.min, -1000), min(info.max, 1000), size=shp).astype(dt)
    return {k: torch.from_numpy(v) for k, v in t.items()}


def same(a, b):
    assert a.site0 == b.site0 and a.n_samples == b.n_samples
    na, nb = a.tensors(), b.tensors()
    assert [k for k, _ in na] == [k for k, _ in nb]
    for (k, x), (_, y) in zip(na, nb):
        assert x.shape == y.shape and x.dtype == y.dtype, k
        assert torch.equal(x.cpu().view(torch.uint8), y.cpu().view(torch.uint8)), k


@pytest.mark.parametrize("S,N,A,G", [(1, 1, 4, 10), (7, 3, 5, 15), (64, 100, 5, 15), (1000, 64, 4, 10), (1025, 1000, 5, 15), (5000, 257, 5, 15), (3, 1001, 5, 15), (2049, 5, 5, 15)])
def test_device_packer_equals_the_torch_formulation(S, N, A, G):
    """every field kind (per-site vectors of 4 / 5 / 16 / 20 bytes, one value per sample, genotype and allele planes incl. one-byte PL),
    row lengths that are / are not multiples of 16 and 4 bytes, more sites than one scan chunk, a fifth of the sites skipped"""
    tile = random_tile(S, N, A, G, seed=S * 31 + N)
    want = shard.pack_records(tile, site0=123)                         # host tensors: the torch formulation
    got = shard.pack_records({k: v.cuda() for k, v in tile.items()}, site0=123)
    torch.cuda.synchronize()
    same(want, got)
    dense = shard.unpack_records(got, A, G)
    kept = tile["site_status"] >= 0
    assert torch.equal(dense["fmt_dp"].cpu(), tile["fmt_dp"][kept])


@pytest.mark.parametrize("p_skip", [0.0, 1.0])
def test_all_kept_and_all_skipped(p_skip):
    tile = random_tile(300, 130, 5, 15, seed=5, p_skip=p_skip, fields=["alleles2acgt", "fmt_dp", "gl", "fmt_ad", "pl_u8"])
    want = shard.pack_records(tile, site0=0)
    got = shard.pack_records({k: v.cuda() for k, v in tile.items()}, site0=0)
    torch.cuda.synchronize()
    same(want, got)
    assert got.n_kept == (300 if p_skip == 0.0 else 0)


def test_empty_tile_and_bad_arguments():
    tile = random_tile(0, 10, 5, 15, seed=1, fields=["fmt_dp", "gl"])
    got = shard.pack_records({k: v.cuda() for k, v in tile.items()}, site0=9)
    assert got.n_kept == 0 and got.nbytes() == 0
    import ctypes as C
    lib = _abi.load_library()
    plan = _abi.PackPlan()
    assert lib.vgl_pack_plan_device(0, 5, None, None, None, C.byref(plan), None) == -1          # VGL_E_ARG, with a message
    assert b"vgl_pack_plan_device" in lib.vgl_last_error()
    f = (_abi.PackField * 1)(_abi.PackField(None, None, 0, 1, 4))
    st = torch.zeros(4, dtype=torch.int32, device="cuda")
    off = torch.zeros((3, 5), dtype=torch.int32, device="cuda")
    assert lib.vgl_pack_records_device(0, 4, st.data_ptr(), st.data_ptr(), off.data_ptr(), None, f, 1, None) == -1


def test_packing_rate_is_a_copy():
    """a 16384 x 1000 tile of GL + DP: the packer moves its bytes at a device-copy rate (>= 1.5 TB/s read + write: loose, the bench line reports the figure)"""
    S, N, G = 16384, 1000, 15
    tile = {"site_status": torch.zeros(S, dtype=torch.int32, device="cuda"), "n_alleles": torch.full((S,), 5, dtype=torch.int32, device="cuda"),
            "fmt_dp": torch.zeros((S, N), dtype=torch.int32, device="cuda"), "gl": torch.zeros((S, G, N), dtype=torch.float32, device="cuda")}
    tile["site_status"][::7] = -3
    for _ in range(2):
        p = shard.pack_records(tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); p = shard.pack_records(tile); e1.record(); torch.cuda.synchronize()
    rate = 2 * p.nbytes() / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(f"packed {p.nbytes() / 1e9:.2f} GB in {e0.elapsed_time(e1):.3f} ms: {rate:.2f} TB/s read + write")
    assert rate > 1.5
