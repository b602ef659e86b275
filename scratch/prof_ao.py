"""Run the solver stage a few times (profiling target)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
fl = ic.SimilarityTheoryFluxes() if cfg == "default" else ic.corrected_atmosphere_ocean_fluxes()
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
for _ in range(5):
    ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
torch.cuda.synchronize()
