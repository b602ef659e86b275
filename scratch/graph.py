"""Does replaying update_state! from a HIP graph shrink the inter-kernel gaps?"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, ic.flux_params())
ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
def step(): ctx.update_state(src, w, ocean, atmos, fluxes, net, time_fraction=0.37)
for _ in range(20): step()
torch.cuda.synchronize()
def timeit(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("stream launches: %.1f us/step" % timeit(step))
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    ctx._check(ctx.lib.cf_set_stream(ctx._h, C.c_void_p(side.cuda_stream)), "set_stream")
    step(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        step()
torch.cuda.synchronize()
print("graph replay:    %.1f us/step" % timeit(g.replay))
