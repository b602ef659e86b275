"""The sea-ice interface solve (bench.py's config-3 workload) with exact, alternating and disabled trip hints."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd")]
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, FluxContext
nx, ny, h = 1440, 560, 7
oc = syn.ocean_state(nx, ny, h, h); si = syn.sea_ice_state(nx, ny, h, h); src_np = syn.jra55_snapshots(4, temporal_correlation=0.95)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
for hints in (1, 0):
    ctx = FluxContext(nx, ny, h, h, ic.flux_params())
    ctx.set_sea_ice_formulation(ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes()))
    ctx.set_option(abi.OPT_TRIP_HINTS, hints)
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    ocean = {k: ctx.to_device(oc[k]) for k in ("T", "S", "u", "v", "mask")}
    st = dict(concentration=ctx.to_device(oc["ice_concentration"]), **{k: ctx.to_device(si[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
    atmos = [ctx.field_set(EXCHANGE_NAMES) for _ in range(4)]
    for n, a in enumerate(atmos): ctx.interpolate_atmosphere_state(src, w, a, 0, 1, n / 9.0)
    out = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL); out["iterations"] = ctx.zeros(torch.int32)
    def timed(sets, reps=40):
        for n in range(12): ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, sets[n % len(sets)], out)
        ctx.sync(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for n in range(reps): ctx.compute_atmosphere_sea_ice_fluxes(st, ocean, sets[n % len(sets)], out)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    print(f"hints {hints}: same inputs {timed(atmos[:1]):.0f} us, two alternating states {timed(atmos[:2]):.0f} us, four in rotation {timed(atmos):.0f} us", flush=True)
    ctx.close()
