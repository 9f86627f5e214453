"""vcfgl_amd -- MI355X (gfx950) implementation of vcfgl's per-site genotype-likelihood
simulation hot path.  The compute lives in libvcfgl_hip.so (vcfgl_amd/csrc, hand-written
HIP) behind the C ABI of include/vcfgl_hip.h; this package is the host-side mirror of the
reference's flag surface and record loop.  There is no CPU fallback."""
from . import _abi
from .params import VcfglArgs, VcfglArgError
from .tile import Tile
from .simulator import Simulator, VglError, pack_gt

__all__ = ["VcfglArgs", "VcfglArgError", "Tile", "Simulator", "VglError", "pack_gt", "_abi"]
