"""Host-side mirror of vcfgl's flag surface for the simulation hot path.

Flag names, defaults and range checks follow the reference's hand-rolled parser
(io.cpp:428-526 defaults, :538-752 names, :757-1000 validation); only flags the hot path
reads are kept.  `VcfglArgs.to_struct()` produces the `vgl_params` of include/vcfgl_hip.h.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _abi


class VcfglArgError(ValueError):
    """The reference prints an [ERROR] and exit(1)s (shared.h:292-299); here it raises."""


# flag -> (attribute, type); every flag takes exactly one value (io.cpp:538-752)
_FLAGS = {
    "--seed": ("seed", int), "-s": ("seed", int),
    "--source": ("source", int),
    "--depth": ("depth", str), "-d": ("depth", str),
    "--depths-file": ("depths_file", str), "-df": ("depths_file", str),
    "--error-rate": ("error_rate", float), "-e": ("error_rate", float),
    "--error-qs": ("error_qs", int), "-eq": ("error_qs", int),
    "--beta-variance": ("beta_variance", float), "-bv": ("beta_variance", float),
    "--gl-model": ("gl_model", int), "-GL": ("gl_model", int),
    "--gl1-theta": ("gl1_theta", float),
    "--qs-bins": ("qs_bins_file", str),
    "--precise-gl": ("precise_gl", int),
    "--i16-mapq": ("i16_mapq", int),
    "--gvcf-dps": ("gvcf_dps", str),
    "--adjust-qs": ("adjust_qs", int),
    "--adjust-by": ("adjust_by", float),
    "-explode": ("explode", int),
    "--rm-invar-sites": ("rm_invar_sites", int),
    "--rm-empty-sites": ("rm_empty_sites", int),
    "-doUnobserved": ("do_unobserved", int),
    "-doGVCF": ("do_gvcf", int),
    "-printPileup": ("print_pileup", int), "-printTruth": ("print_truth", int),
    "-printBasePickError": ("print_base_pick_error", int), "-printQsError": ("print_qs_error", int),
    "-printGlError": ("print_gl_error", int), "-printQScores": ("print_qscores", int),
    "-addGL": ("add_gl", int), "-addFormatGL": ("add_gl", int),
    "-addGP": ("add_gp", int), "-addFormatGP": ("add_gp", int),
    "-addPL": ("add_pl", int), "-addFormatPL": ("add_pl", int),
    "-addI16": ("add_i16", int), "-addQS": ("add_qs", int),
    "-addFormatDP": ("add_fmt_dp", int), "-addInfoDP": ("add_info_dp", int),
    "-addFormatAD": ("add_fmt_ad", int), "-addInfoAD": ("add_info_ad", int),
    "-addFormatADF": ("add_fmt_adf", int), "-addInfoADF": ("add_info_adf", int),
    "-addFormatADR": ("add_fmt_adr", int), "-addInfoADR": ("add_info_adr", int),
    "--output-mode": ("output_mode", str), "-O": ("output_mode", str),
    "--threads": ("threads", int), "-@": ("threads", int),
    "--input": ("input", str), "-i": ("input", str),
    "--output": ("output", str), "-o": ("output", str),
    "--verbose": ("verbose", int), "-V": ("verbose", int),
    # extensions of this implementation (not reference flags)
    "--rng-mode": ("rng_mode", int), "--beta-sampler": ("beta_sampler", int),
}


@dataclass
class VcfglArgs:
    seed: int = -1
    source: int = 0                      # ARG_GTSOURCE_BINARY
    depth: Optional[float] = None        # --depth ("inf" is the truth path, out of scope)
    depths: Optional[Sequence[float]] = None   # --depths-file contents
    error_rate: Optional[float] = None
    error_qs: int = 0
    beta_variance: float = -1.0
    gl_model: int = 2
    gl1_theta: float = 0.83
    qs_bins: Optional[List[Sequence[int]]] = None
    precise_gl: int = 0
    i16_mapq: int = 20
    adjust_qs: int = 0
    adjust_by: float = 0.499
    explode: int = 0
    rm_invar_sites: int = 0
    rm_empty_sites: int = 0
    do_unobserved: int = 1               # ARG_DOUNOBSERVED_STAR
    do_gvcf: int = 0
    print_pileup: int = 0
    print_truth: int = 0
    print_base_pick_error: int = 0       # TSV lines on stdout (vcfgl.cpp:430-435, 533-554)
    print_qs_error: int = 0
    print_gl_error: int = 0
    print_qscores: int = 0
    add_gl: int = 1
    add_gp: int = 0
    add_pl: int = 0
    add_i16: int = 0
    add_qs: int = 0
    add_fmt_dp: int = 1
    add_info_dp: int = 0
    add_fmt_ad: int = 0
    add_info_ad: int = 0
    add_fmt_adf: int = 0
    add_info_adf: int = 0
    add_fmt_adr: int = 0
    add_info_adr: int = 0
    rng_mode: int = _abi.VGL_RNG_TILE
    beta_sampler: int = _abi.VGL_BETA_RAND48
    out_layout: int = _abi.VGL_LAYOUT_PLANES       # layout of the multi-valued FORMAT arrays (include/vcfgl_hip.h)
    extra: dict = field(default_factory=dict)   # parsed but unused flags (output mode, threads, ...)

    # ------------------------------------------------------------------ parsing
    @classmethod
    def from_argv(cls, argv: Sequence[str], base_dir: str = ".") -> "VcfglArgs":
        import os
        a = cls()
        it = iter(argv)
        for flag in it:
            if flag not in _FLAGS:
                raise VcfglArgError(f"Unknown argument: {flag}")
            try:
                val = next(it)
            except StopIteration:
                raise VcfglArgError(f"Argument {flag} requires a value")
            name, typ = _FLAGS[flag]
            if name == "depth":
                if val == "inf":
                    raise VcfglArgError("--depth inf (true-value output, vcfgl.cpp:1089-1262) is outside the simulation hot path")
                a.depth = float(val)
            elif name == "depths_file":
                with open(os.path.join(base_dir, val)) as fh:
                    a.depths = [float(x) for x in fh.read().split()]
            elif name == "qs_bins_file":
                with open(os.path.join(base_dir, val)) as fh:
                    a.qs_bins = [tuple(int(x) for x in ln.split(",")) for ln in fh.read().split()]
            elif hasattr(a, name):
                setattr(a, name, typ(val))
            else:
                a.extra[name] = typ(val)
        return a

    # ------------------------------------------------------------------ validation (io.cpp:757-1000)
    def validate(self):
        def rng(v, lo, hi, s):
            if v < lo or v > hi:
                raise VcfglArgError(f"[Bad argument value: '{s} {v}'] Allowed range is [{lo},{hi}]")
        if self.depth is None and self.depths is None:
            raise VcfglArgError("Average per-site read depth value is required. Please set it using --depth or --depths-file and re-run.")
        if self.depths is None:
            rng(self.depth, 0.0, 500.0, "--depth")
        if self.error_rate is None:
            raise VcfglArgError("Error rate is not specified. Please use --error-rate option to specify the error rate. Allowed range: [0.0, 1.0]")
        if self.error_rate < 0.0 or self.error_rate >= 1.0:
            raise VcfglArgError(f"[Bad argument value: '--error-rate {self.error_rate}'] Allowed range is [0.0,1.0]")
        rng(self.error_qs, 0, 2, "--error-qs")
        rng(self.gl_model, 1, 2, "--gl-model")
        rng(self.gl1_theta, 0.0, 1.0, "--gl1-theta")
        rng(self.precise_gl, 0, 1, "--precise-gl")
        rng(self.i16_mapq, 0, 60, "--i16-mapq")
        rng(self.adjust_qs, 0, 31, "--adjust-qs")
        rng(self.do_unobserved, 0, 5, "-doUnobserved")
        rng(self.rm_invar_sites, 0, 7, "--rm-invar-sites")
        if self.adjust_qs and self.adjust_by == 0.0:
            raise VcfglArgError(f"--adjust-qs {self.adjust_qs} requires a non-zero value for --adjust-by.")
        if (self.adjust_qs & 1) and self.precise_gl:
            raise VcfglArgError("--adjust-qs 1 requires --precise-gl 0.")
        if (self.adjust_qs & 2) and not self.add_qs:
            raise VcfglArgError("--adjust-qs 2 requires -addQS 1.")
        if (self.adjust_qs & 4) and not self.print_pileup:                     # io.cpp:891-898
            raise VcfglArgError("--adjust-qs 4 requires --printPileup 1.")
        if (self.adjust_qs & 8) and not self.print_qscores:
            raise VcfglArgError("--adjust-qs 8 requires --printQScores 1.")
        if (self.adjust_qs & 16) and not self.print_gl_error:
            raise VcfglArgError("--adjust-qs 16 requires --printGlError 1.")
        if self.print_gl_error and self.gl_model == 1:                         # io.cpp:993
            raise VcfglArgError("-printGlError 1 is not supported with genotype likelihood model 1 (--gl-model 1).")
        if self.gl_model == 1 and self.precise_gl:
            raise VcfglArgError("Precise genotype likelihood error (--precise-gl 1) is not supported with genotype likelihood model 1 (--gl-model 1).")
        if self.error_qs == 0 and self.beta_variance >= 0:
            raise VcfglArgError(f"--beta-variance {self.beta_variance:e} requires --error-qs 1 or 2.")
        if self.error_qs != 0:
            if not self.error_rate > 0:
                raise VcfglArgError("--error-qs 1 or 2 requires --error-rate > 0")
            if not self.beta_variance > 0:
                raise VcfglArgError("--error-qs 1 or 2 requires --beta-variance > 0")
        return self

    # ------------------------------------------------------------------ C struct
    def to_struct(self, n_samples: int):
        """Returns (Params struct, keepalive list)."""
        p = _abi.Params()
        keep = []
        p.abi_version = _abi.ABI_VERSION
        p.seed = C.c_int32(int(self.seed) & 0xFFFFFFFF).value      # truncated like the reference's (int) / srand48 (io.cpp:1047-1061)
        p.n_samples = n_samples
        p.rng_mode = self.rng_mode
        p.beta_sampler = self.beta_sampler
        p.depth = float(self.depth) if self.depth is not None else -1.0
        if self.depths is not None:
            if len(self.depths) != n_samples:
                raise VcfglArgError("--depths-file must hold one depth per sample")
            arr = (C.c_double * n_samples)(*[float(x) for x in self.depths])
            keep.append(arr)
            p.depths = C.cast(arr, C.POINTER(C.c_double))
        p.error_rate = float(self.error_rate)
        p.error_qs = self.error_qs
        p.beta_variance = float(self.beta_variance)
        p.gl_model = self.gl_model
        p.gl1_theta = float(self.gl1_theta)
        p.precise_gl = self.precise_gl
        p.adjust_qs = self.adjust_qs
        p.adjust_by = float(self.adjust_by)
        if self.qs_bins:
            flat = [int(v) for t in self.qs_bins for v in t]
            arr = (C.c_int32 * len(flat))(*flat)
            keep.append(arr)
            p.n_qs_bins = len(self.qs_bins)
            p.qs_bins = C.cast(arr, C.POINTER(C.c_int32))
        p.i16_mapq = self.i16_mapq
        p.do_unobserved = self.do_unobserved
        p.rm_invar_sites = self.rm_invar_sites
        p.rm_empty_sites = self.rm_empty_sites
        p.do_gvcf = self.do_gvcf
        for f in ("add_gl", "add_gp", "add_pl", "add_i16", "add_qs", "add_fmt_dp", "add_info_dp", "add_fmt_ad",
                  "add_info_ad", "add_fmt_adf", "add_info_adf", "add_fmt_adr", "add_info_adr"):
            setattr(p, f, int(getattr(self, f)))
        p.out_layout = int(self.out_layout)
        return p, keep

    @property
    def max_alleles(self) -> int:          # PROGRAM_WILL_ADD_UNOBSERVED, shared.h:151-152
        return 5 if self.do_unobserved in (1, 2, 4, 5) else 4

    @property
    def max_genotypes(self) -> int:        # lut_nAlleles_to_nGenotypes, shared.cpp:29
        return 15 if self.max_alleles == 5 else 10
