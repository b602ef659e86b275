"""Solver time vs FixedIterations(n): separates per-iteration cost from fixed overhead."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, int(os.environ.get("NY", 560)), 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
for name, mk in (("default", lambda: ic.SimilarityTheoryFluxes()), ("corrected", ic.corrected_atmosphere_ocean_fluxes)):
    res = {}
    ctx = None
    for n in (0, 1, 2, 3, 4, 8, 16, 32):
        fl = mk(); fl.solver_stop_criteria = ic.FixedIterations(n)
        P = ic.flux_params(fl)
        if ctx is None:
            ctx = FluxContext(nx, ny, h, h, P)
            ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
            src = {k: ctx.to_device(v) for k, v in src_np.items()}
            w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
            atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES)
            ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
        else:
            ctx.set_flux_params(P)
        if os.environ.get("SOLVER"): ctx.set_option(abi.OPT_SOLVER, int(os.environ["SOLVER"]))
        if os.environ.get("LAYOUT"): ctx.set_option(abi.OPT_LATENCY_LAYOUT, int(os.environ["LAYOUT"]))
        if os.environ.get("HINTS"): ctx.set_option(abi.OPT_TRIP_HINTS, int(os.environ["HINTS"]))
        res[n] = round(min(ctx.time_stage(abi.STAGE_AO_FLUXES, 20, ocean=ocean, atmos=atmos, fluxes=fluxes) for _ in range(3)) * 1e3, 1)
    print(name, json.dumps(res))
    ctx.close()
