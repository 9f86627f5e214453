import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, ctypes as C
from vcfgl_amd import _abi
lib=_abi.load_library(hooks=True)
lib.vgl_dbg_vlog.argtypes=[C.c_void_p,C.c_void_p,C.c_int,C.c_int]
rng=np.random.default_rng(1)
parts=[rng.random(4_000_000).astype(np.float32), (1-rng.random(2_000_000)*1e-3).astype(np.float32), (1+ (rng.random(2_000_000)-0.5)*0.2).astype(np.float32),
       np.exp(rng.uniform(-33,3,4_000_000)).astype(np.float32), np.float32(2.0)**rng.integers(-48,2,1000), (1-2.0**-np.arange(1,25)).astype(np.float32), (1+2.0**-np.arange(1,24)).astype(np.float32)]
x=np.concatenate(parts).astype(np.float32); x=x[x>0]
xi=torch.from_numpy(x).cuda(); xo=torch.empty_like(xi)
assert lib.vgl_dbg_vlog(xi.data_ptr(), xo.data_ptr(), x.size, 0)==0
y=xo.cpu().numpy().astype(np.float64)
t=np.log2(x.astype(np.float64))
err=np.abs(y-t)
rel=err/np.maximum(np.abs(t),1e-300)
print("n",x.size,"max abs err",err.max(),"max rel err (|t|>1e-3)",rel[np.abs(t)>1e-3].max(), "ulps:", (rel[np.abs(t)>1e-3].max()/2**-24))
near=np.abs(x-1)<0.1
print("near 1: max abs err", err[near].max(), "max err/|x-1|", (err[near]/np.maximum(np.abs(x[near].astype(np.float64)-1),1e-30)).max())
b=np.abs(t)*2.0**-21+2.0**-23
print("bound violated:", np.sum(err>b))

# ---- tanf on (0, pi): error relative to (1+y^2)*2^-23*a + |y|*2^-21 ; v_exp_f32 on [-60, 2]
a=np.concatenate([rng.random(6_000_000)*3.141592654, 1.5707963+ (rng.random(1_000_000)-0.5)*1e-2, rng.random(500_000)*1e-3, 3.141592654-rng.random(500_000)*1e-3]).astype(np.float32)
a=a[(a>0)&(a<3.1415927)]
ai=torch.from_numpy(a).cuda(); ao=torch.empty_like(ai)
assert lib.vgl_dbg_vlog(ai.data_ptr(), ao.data_ptr(), a.size, 1)==0
y=ao.cpu().numpy().astype(np.float64); t=np.tan(a.astype(np.float64))
err=np.abs(y-t); rel=err/np.maximum(np.abs(t),1e-300)
print("tanf: max rel err", rel.max(), "ulps(2^-23):", rel.max()/2**-23, "bound |y|*2^-21 violated:", int(np.sum(err>np.abs(t)*2.0**-21)))
e=np.concatenate([rng.uniform(-90,3,6_000_000), rng.uniform(-1,1,1_000_000)]).astype(np.float32)
ei=torch.from_numpy(e).cuda(); eo=torch.empty_like(ei)
assert lib.vgl_dbg_vlog(ei.data_ptr(), eo.data_ptr(), e.size, 2)==0
y=eo.cpu().numpy().astype(np.float64); t=np.exp2(e.astype(np.float64))
rel=np.abs(y-t)/t
print("v_exp_f32: max rel err", rel.max(), "ulps:", rel.max()/2**-23, "bound 2^-21 violated:", int(np.sum(rel>2.0**-21)))
