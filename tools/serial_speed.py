import sys,time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
for kw,S,N in ((dict(depth=10,error_rate=0.01,gl_model=1),2000,100),(dict(depth=20,error_rate=0.01,error_qs=2,beta_variance=1e-5),200,100)):
    a=VcfglArgs(seed=42,**kw); a.rng_mode=_abi.VGL_RNG_SERIAL; a.beta_sampler=_abi.VGL_BETA_STD
    sim=Simulator(a,N,max_sites_per_tile=S); gt=synth.binary_sites(0,S,N)
    sim.simulate(0,gt[:10],fields=["fmt_dp","gl"])
    t=time.time(); sim.simulate(10,gt[10:],fields=["fmt_dp","gl"]); dt=time.time()-t
    print(kw, "serial-mode evals/s:", (S-10)*N/dt)
