"""interpolate_kernel stand-alone time on the 1/4-degree surface under COFLUX_INTERP_BLOCKS / tile-cap settings (scratch)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
ROOT = os.environ["AB_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
nx, ny, h = int(os.environ.get("NX", 1440)), int(os.environ.get("NY", 560)), 7
ocean_np = syn.ocean_state(nx, ny, h, h); src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
ctx = FluxContext(nx, ny, h, h, ic.flux_params(ic.SimilarityTheoryFluxes()))
if os.environ.get("CAP"): ctx.set_option(abi.OPT_INTERP_TILE_CAP, int(os.environ["CAP"]))
ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
src = {k: ctx.to_device(v) for k, v in src_np.items()}
w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
kw = dict(src=src, weights=w, ocean=ocean, atmos=atmos, fluxes=fluxes, net=net, time_fraction=0.37)
for _ in range(3): ctx.time_stage(abi.STAGE_INTERPOLATE, 300, **kw)
t = min(ctx.time_stage(abi.STAGE_INTERPOLATE, 100, **kw) for _ in range(5))
print(json.dumps(dict(interp_us=round(t * 1e3, 2))))
ctx.close()
'''
for spec in sys.argv[1:] or ["default"]:
    env = dict(os.environ, AB_ROOT=ROOT, COFLUX_EXPERIMENTS="1")
    for kv in spec.split(","):
        if "=" in kv:
            k, v = kv.split("="); env[k] = v
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(spec, line[-1] if line else "FAILED " + out.stderr[-400:], flush=True)
