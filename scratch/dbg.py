import sys
sys.path.insert(0,'/root/repo/climaocean.jl_amd')
import numpy as np, torch
from coflux import interface_computations as ic
from coflux.runtime import FluxContext
ctx=FluxContext(16,16,2,2,ic.flux_params())
rng=np.random.default_rng(1)
x=np.concatenate([rng.uniform(1,2,200000), 10.0**rng.uniform(-100,100,100000)])
for fn,ref,name in ((7,1/x,'raw rcp'),(9,1/x,'rcp+1NR'),(4,1/x,'frcp(2NR)'),(8,1/np.sqrt(x),'raw rsq'),(3,np.sqrt(x),'fsqrt')):
    got=ctx.debug_eval(fn,ctx.to_device(x)).cpu().numpy()
    print(name,'max rel err %.3e'%np.max(np.abs(got/ref-1)))
