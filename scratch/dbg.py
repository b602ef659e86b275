import sys
sys.path.insert(0,'/root/repo/climaocean.jl_amd')
import numpy as np, torch
from coflux import interface_computations as ic
from coflux.runtime import FluxContext
ctx=FluxContext(16,16,2,2,ic.flux_params())
x=np.array([-611.64370537,-160.6,436.867,0.5,-0.5,1.0,10.0,-10.0,0.0,100.0,-100.0,50.0])
for fn in range(5):
    got=ctx.debug_eval(fn,ctx.to_device(x)).cpu().numpy()
    print(fn,got)
print(np.exp(x))
