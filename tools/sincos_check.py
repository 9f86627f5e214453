#!/usr/bin/env python3
"""Diagnostic: absolute error of v_sin_f32 / v_cos_f32 (inputs in revolutions) on x in (0, 1/2), i.e. sin / cos of pi*u for
u in (0, 1) -- the bound the float32 Poisson attempt of k_depth assumes for tan(pi u) = sin / cos."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from vcfgl_amd import _abi
lib = _abi.load_library(hooks=True)
lib.vgl_dbg_vlog.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
rng = np.random.default_rng(3)
u = np.concatenate([rng.random(8_000_000), 0.5 + (rng.random(2_000_000) - 0.5) * 1e-2, rng.random(1_000_000) * 1e-3, 1 - rng.random(1_000_000) * 1e-3,
                    np.arange(1, 4096) / 4096.0])
x = (u * 0.5).astype(np.float32)
x = x[(x > 0) & (x < 0.5)]
xi = torch.from_numpy(x).cuda(); xo = torch.empty_like(xi)
for mode, name, fn in ((3, "v_sin_f32", np.sin), (4, "v_cos_f32", np.cos)):
    assert lib.vgl_dbg_vlog(xi.data_ptr(), xo.data_ptr(), x.size, mode) == 0
    y = xo.cpu().numpy().astype(np.float64)
    t = fn(2 * np.pi * x.astype(np.float64))
    err = np.abs(y - t)
    print(f"{name}: n {x.size} max abs err {err.max():.3e} = 2^{np.log2(err.max()):.2f}; max rel err where |t| > 1e-3: {(err / np.maximum(np.abs(t), 1e-300))[np.abs(t) > 1e-3].max():.3e}")
