"""Can the LDS-free interpolation run inside the solver?  Two contexts on two streams."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
A = FluxContext(nx, ny, h, h, ic.flux_params())
B = FluxContext(nx, ny, h, h, ic.flux_params())
for c in (A, B):
    c._check(c.lib.cf_set_stream(c._h, None), "own stream")   # each context's own non-blocking stream
ocean = {k: A.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: A.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=A.to_device(fi), fj=A.to_device(fj), latitude=A.to_device(phi))
atmos = A.field_set(EXCHANGE_NAMES); atmos2 = A.field_set(EXCHANGE_NAMES); fluxes = A.field_set(FLUX_NAMES)
A.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37); A.sync()
def run(n, ao=True, interp=True, cap=None):
    if cap is not None: B.set_option(abi.OPT_INTERP_TILE_CAP, cap)
    A.sync(); B.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if ao: A.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
        if interp: B.interpolate_atmosphere_state(src, w, atmos2, 0, 1, 0.5)
    A.sync(); B.sync()
    return (time.perf_counter() - t0) / n * 1e6
for _ in range(3): run(20)
print("AO alone            %.1f us" % run(200, True, False))
print("interp tiled alone  %.1f us" % run(200, False, True, 128))
print("interp gather alone %.1f us" % run(200, False, True, 0))
print("AO || interp tiled  %.1f us" % run(200, True, True, 128))
print("AO || interp gather %.1f us" % run(200, True, True, 0))
