"""Quick A/B timing of the solver stage on the 1/4-degree surface (scratch tool, not a test)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ocean_np = syn.ocean_state(nx, ny, h, h)
src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
for name, fl in (("default", ic.SimilarityTheoryFluxes()), ("corrected", ic.corrected_atmosphere_ocean_fluxes()), ("ncar", ic.ncar_atmosphere_ocean_fluxes())):
    ctx = FluxContext(nx, ny, h, h, ic.flux_params(fl))
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    res = {}
    if os.environ.get("HINTS"): ctx.set_option(abi.OPT_TRIP_HINTS, int(os.environ["HINTS"]))
    if os.environ.get("CHUNK"): ctx.set_option(abi.OPT_AO_CHUNK, int(os.environ["CHUNK"]))
    if os.environ.get("CAP"): ctx.set_option(abi.OPT_INTERP_TILE_CAP, int(os.environ["CAP"]))
    for mb in [int(a) for a in sys.argv[1:]] or [1024]:
        ctx.set_option(abi.OPT_MAX_BLOCKS, mb)
        ao = min(ctx.time_stage(abi.STAGE_AO_FLUXES, 20, ocean=ocean, atmos=atmos, fluxes=fluxes) for _ in range(3))
        fu = min(ctx.time_stage(abi.STAGE_UPDATE_STATE, 20, src=src, weights=w, ocean=ocean, atmos=atmos, fluxes=fluxes, net=net, time_fraction=0.37) for _ in range(3))
        it = min(ctx.time_stage(abi.STAGE_INTERPOLATE, 20, src=src, weights=w, atmos=atmos, time_fraction=0.37) for _ in range(3))
        res[mb] = dict(ao_us=round(ao * 1e3, 1), update_state_us=round(fu * 1e3, 1), interp_us=round(it * 1e3, 1))
    print(name, json.dumps(res))
    ctx.close()
