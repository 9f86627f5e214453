"""ctypes view of include/vcfgl_hip.h (the C ABI of libvcfgl_hip.so).

Pure declarations: no compute, no fallback.  `load_library()` raises if the HIP library has
not been built -- the product path never falls back to a CPU implementation.
"""
import ctypes as C
import os

ABI_VERSION = 6

VGL_OK = 0
VGL_E_ARG, VGL_E_NODEVICE, VGL_E_NOMEM, VGL_E_CAPACITY, VGL_E_UNSUPPORTED, VGL_E_QSBIN, VGL_E_ADJQ = -1, -2, -3, -4, -5, -6, -7
VGL_SITE_OK, VGL_SITE_SKIP_INVAR, VGL_SITE_SKIP_EMPTY, VGL_SITE_NO_READS = 0, -3, -4, 1
VGL_RNG_TILE, VGL_RNG_SERIAL = 0, 1
VGL_BETA_RAND48, VGL_BETA_STD = 0, 1
VGL_LAYOUT_PLANES, VGL_LAYOUT_SAMPLE_MAJOR = 0, 1
VGL_GT_MISSING = 0xF
FLOAT_MISSING_BITS = 0x7F800001
INT32_MISSING = -(2 ** 31)


class RngLayout(C.Structure):
    _fields_ = [("block", C.c_uint64), ("off", C.c_uint64 * 4), ("qs_read_stride", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("seed", C.c_int32), ("n_samples", C.c_int32),
        ("rng_mode", C.c_int32), ("beta_sampler", C.c_int32),
        ("depth", C.c_double), ("depths", C.POINTER(C.c_double)),
        ("error_rate", C.c_double), ("error_qs", C.c_int32), ("beta_variance", C.c_double),
        ("gl_model", C.c_int32), ("gl1_theta", C.c_double), ("precise_gl", C.c_int32),
        ("adjust_qs", C.c_int32), ("adjust_by", C.c_double),
        ("n_qs_bins", C.c_int32), ("qs_bins", C.POINTER(C.c_int32)), ("i16_mapq", C.c_int32),
        ("do_unobserved", C.c_int32), ("rm_invar_sites", C.c_int32), ("rm_empty_sites", C.c_int32),
        ("do_gvcf", C.c_int32),
        ("add_gl", C.c_int32), ("add_gp", C.c_int32), ("add_pl", C.c_int32), ("add_i16", C.c_int32),
        ("add_qs", C.c_int32), ("add_fmt_dp", C.c_int32), ("add_info_dp", C.c_int32),
        ("add_fmt_ad", C.c_int32), ("add_info_ad", C.c_int32),
        ("add_fmt_adf", C.c_int32), ("add_info_adf", C.c_int32),
        ("add_fmt_adr", C.c_int32), ("add_info_adr", C.c_int32),
        ("layout", RngLayout),
        ("out_layout", C.c_int32),
    ]


class TileOut(C.Structure):
    _fields_ = [
        ("site_status", C.c_void_p), ("n_alleles", C.c_void_p), ("n_alleles_obs", C.c_void_p),
        ("alleles2acgt", C.c_void_p), ("info_dp", C.c_void_p), ("info_ad", C.c_void_p),
        ("info_adf", C.c_void_p), ("info_adr", C.c_void_p), ("qs", C.c_void_p), ("i16", C.c_void_p),
        ("fmt_dp", C.c_void_p), ("gl", C.c_void_p), ("pl", C.c_void_p), ("gp", C.c_void_p),
        ("fmt_ad", C.c_void_p), ("fmt_adf", C.c_void_p), ("fmt_adr", C.c_void_p),
        ("reads", C.c_void_p), ("read_capacity", C.c_int32),
        ("read_errp", C.c_void_p), ("site_pick_err", C.c_void_p),
        ("pl_u8", C.c_void_p),
    ]


class CtxInfo(C.Structure):
    """vgl_ctx_info_t (ABI 5): what a context will launch"""
    _fields_ = [
        ("size", C.c_int32), ("abi_version", C.c_int32), ("device", C.c_int32),
        ("n_samples", C.c_int32), ("max_sites_per_tile", C.c_int32), ("max_alleles", C.c_int32), ("max_genotypes", C.c_int32),
        ("rng_mode", C.c_int32), ("depth_mode", C.c_int32), ("fused", C.c_int32), ("fused_split", C.c_int32), ("sample_lean", C.c_int32),
        ("gl_sort", C.c_int32), ("gl_wpb", C.c_int32), ("read_cap", C.c_int32), ("pool_cap", C.c_int32), ("pool_lds_bytes", C.c_int32),
        ("test_hooks", C.c_int32), ("workspace_bytes", C.c_int64), ("rng_tile_max_sites", C.c_int64),
    ]


VGL_DEPTH_INPLACE_MIXED, VGL_DEPTH_KDEPTH, VGL_DEPTH_INPLACE_PRODUCT, VGL_DEPTH_SERIAL_SCOUT = 0, 1, 2, 3
# vgl_ctx_kernel_ms buckets (VGL_T_*)
TIMING_BUCKETS = ["k_depth", "k_sample", "k_redo", "k_site", "k_gl", "k_siteagg"]
T_DEPTH, T_SAMPLE, T_REDO, T_SITE, T_GL, T_SITEAGG = range(6)

# (field, dtype, shape-kind): shape kinds resolved by tile.py
TILE_FIELDS = [
    ("site_status", "int32", "site"), ("n_alleles", "int32", "site"), ("n_alleles_obs", "int32", "site"),
    ("alleles2acgt", "int8", "site5"), ("info_dp", "int32", "site"), ("info_ad", "int32", "siteA"),
    ("info_adf", "int32", "siteA"), ("info_adr", "int32", "siteA"), ("qs", "float32", "siteA"),
    ("i16", "float32", "site16"), ("fmt_dp", "int32", "eval"), ("gl", "float32", "planeG"),
    ("pl", "int32", "planeG"), ("gp", "float32", "planeG"), ("fmt_ad", "int32", "planeA"),
    ("fmt_adf", "int32", "planeA"), ("fmt_adr", "int32", "planeA"), ("pl_u8", "uint8", "planeG"),
]

# every symbol include/vcfgl_hip.h declares
EXPORTS = [
    "vgl_max_alleles", "vgl_max_genotypes", "vgl_default_rng_layout", "vgl_abi_version",
    "vgl_last_error", "vgl_ctx_create", "vgl_ctx_destroy", "vgl_simulate_tile",
    "vgl_simulate_tile_device", "vgl_ctx_check", "vgl_ctx_timing", "vgl_ctx_kernel_ms", "vgl_rng_tile_max_sites", "vgl_rng_tile_site_hash",
    "vgl_simulate_tile_async", "vgl_tile_wait", "vgl_host_alloc", "vgl_host_alloc_on", "vgl_host_free", "vgl_ctx_info",
    "vgl_pack_plan_device", "vgl_pack_records_device",
]
VGL_PACK_ROW, VGL_PACK_ROWS_G, VGL_PACK_ROWS_A = 0, 1, 2


class PackField(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("kind", C.c_int32), ("planes", C.c_int32), ("row_bytes", C.c_int64)]


class PackPlan(C.Structure):
    _fields_ = [("n_kept", C.c_int64), ("rows_g", C.c_int64), ("rows_a", C.c_int64)]

# entry points of the -DVGL_TEST_HOOKS build only (lib/libvcfgl_hip_hooks.so): never in the shipped library
HOOK_EXPORTS = ["vgl_dbg_bound_sweep", "vgl_dbg_chain", "vgl_dbg_redo_count", "vgl_dbg_site_base", "vgl_dbg_stamps", "vgl_dbg_vlog"]

_LIB = {}


def library_path(hooks=False):
    # VGL_LIB: another build of the same library (A/B timing of kernel variants: tools/ab_build.sh)
    if os.environ.get("VGL_LIB"):
        return os.environ["VGL_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libvcfgl_hip_hooks.so" if hooks else "libvcfgl_hip.so")


def load_library(hooks=False):
    """dlopen libvcfgl_hip.so and set prototypes.  Raises RuntimeError when it is not built.
    hooks=True: the -DVGL_TEST_HOOKS build of the same sources (environment overrides such as VGL_NO_FUSE / VGL_DEBUG_READ_CAP and
    the vgl_dbg_* entry points) -- what the hook cases of the test-suite and tools/ load; the product path never does."""
    hooks = bool(hooks)
    if hooks in _LIB:
        return _LIB[hooks]
    path = library_path(hooks)
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vcfgl_amd/csrc`.  There is no CPU fallback.")
    try:
        # torch wheels bundle their own libamdhip64.so.7; the first copy loaded serves the whole
        # process, so load torch's first: device pointers of torch tensors and this library's
        # launches must go through one HIP runtime.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.vgl_max_alleles.argtypes = [C.POINTER(Params)]
    lib.vgl_max_genotypes.argtypes = [C.POINTER(Params)]
    lib.vgl_default_rng_layout.argtypes = [C.POINTER(Params), C.POINTER(RngLayout)]
    lib.vgl_last_error.restype = C.c_char_p
    lib.vgl_rng_tile_max_sites.argtypes = [C.POINTER(Params), C.POINTER(C.c_int64)]
    lib.vgl_rng_tile_site_hash.argtypes = [C.POINTER(Params), C.c_int64, C.POINTER(C.c_int64)]
    lib.vgl_ctx_create.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.vgl_ctx_destroy.argtypes = [C.c_void_p]
    lib.vgl_simulate_tile.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(TileOut)]
    lib.vgl_simulate_tile_async.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(TileOut), C.POINTER(C.c_int32)]
    lib.vgl_tile_wait.argtypes = [C.c_void_p, C.c_int32]
    lib.vgl_host_alloc.restype = C.c_void_p
    lib.vgl_host_alloc.argtypes = [C.c_size_t]
    lib.vgl_host_alloc_on.restype = C.c_void_p
    lib.vgl_host_alloc_on.argtypes = [C.c_int32, C.c_size_t]
    lib.vgl_host_free.argtypes = [C.c_void_p]
    lib.vgl_simulate_tile_device.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.POINTER(TileOut), C.c_void_p]
    lib.vgl_ctx_check.argtypes = [C.c_void_p, C.c_void_p]
    lib.vgl_ctx_timing.argtypes = [C.c_void_p, C.c_int32]
    lib.vgl_ctx_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32, C.c_int32]
    lib.vgl_ctx_info.argtypes = [C.c_void_p, C.POINTER(CtxInfo)]
    lib.vgl_pack_plan_device.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PackPlan), C.c_void_p]
    lib.vgl_pack_records_device.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(PackField), C.c_int32, C.c_void_p]
    if lib.vgl_abi_version() != ABI_VERSION:
        raise RuntimeError("libvcfgl_hip.so ABI version mismatch")
    _LIB[hooks] = lib
    return lib
