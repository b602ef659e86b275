"""Per-kernel time (back-to-back launches, HIP events around the batch) vs slab height: what one rank of an
N-way strong-scaled 1440x560 surface costs.  usage: slab_scan.py [config] [ny ...]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
cfg = sys.argv[1] if len(sys.argv) > 1 else "default"
nys = [int(a) for a in sys.argv[2:]] or [560, 280, 140, 70]
mk = {"default": ic.SimilarityTheoryFluxes, "corrected": ic.corrected_atmosphere_ocean_fluxes, "ncar": ic.ncar_atmosphere_ocean_fluxes}[cfg]
nx, h = 1440, 7
for ny in nys:
    ocean_np = syn.ocean_state(nx, ny, h, h, ny_global=560, j_offset=(560 - ny) // 2); src_np = syn.jra55_snapshots(2)
    fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h, ny_global=560, j_offset=(560 - ny) // 2)
    ctx = FluxContext(nx, ny, h, h, ic.flux_params(mk()))
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
    kw = dict(src=src, weights=w, ocean=ocean, atmos=atmos, fluxes=fluxes, net=net, time_fraction=0.37)
    ctx.update_state(src, w, ocean, atmos, fluxes, net, time_fraction=0.37)
    res = {}
    for name, st in (("interp", abi.STAGE_INTERPOLATE), ("ao", abi.STAGE_AO_FLUXES), ("net", abi.STAGE_NET_FLUXES), ("step", abi.STAGE_UPDATE_STATE)):
        res[name] = round(min(ctx.time_stage(st, 200, **kw) for _ in range(3)) * 1e3, 2)
    print(cfg, ny, json.dumps(res), flush=True)
    ctx.close()
