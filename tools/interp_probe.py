"""Stand-alone interpolation launch timed back to back (tools: A/B libraries through LIBCOFLUX)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = 1440, 560, 7
ctx = FluxContext(nx, ny, h, h, ic.flux_params(), ring=1)
o = syn.ocean_state(nx, ny, h, h)
ocean = {k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(2).items()}
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos = ctx.field_set(EXCHANGE_NAMES); fl = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
kw = dict(src=src, weights=w, ocean=ocean, atmos=atmos, fluxes=fl, net=net, time_fraction=0.37)
for _ in range(3): ctx.time_stage(abi.STAGE_INTERPOLATE, 200, **kw)
print(json.dumps(dict(lib=os.environ.get("LIBCOFLUX", "prod").split("_")[-1], interp_us=[round(1e3 * ctx.time_stage(abi.STAGE_INTERPOLATE, 200, **kw), 3) for _ in range(5)])))
