"""A/B of the solver bodies on the 1/4-degree surface (scratch tool): CF_SOLVER_TABLES (round-3 lean iteration) vs
CF_SOLVER_TABLES_R2 (round-2 body on the same tables): kernel time on identical inputs and parity against the C oracle
(worst scaled error per field, number of cells whose trip count differs).
usage: ab_lean.py [config ...]   env: NX, NY, REPS, NOCHECK=1"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

from coflux import abi, interface_computations as ic, synthetic as syn  # noqa: E402
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FluxContext  # noqa: E402

nx, ny, h = int(os.environ.get("NX", 1440)), int(os.environ.get("NY", 560)), 7
reps = int(os.environ.get("REPS", 100))
check = not os.environ.get("NOCHECK")
SOLVER_R2 = 2
configs = sys.argv[1:] or ["default", "corrected"]
mk = {"default": ic.SimilarityTheoryFluxes, "corrected": ic.corrected_atmosphere_ocean_fluxes, "ncar": ic.ncar_atmosphere_ocean_fluxes}
ocean_np = syn.ocean_state(nx, ny, h, h)
src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
SCALE = dict(sensible_heat=1.0, latent_heat=1.0, water_vapor=1e-6, x_momentum=1e-3, y_momentum=1e-3, temperature=1.0)

for cfg in configs:
    params = ic.flux_params(mk[cfg]())
    ctx = FluxContext(nx, ny, h, h, params)
    if os.environ.get("CHUNK"): ctx.set_option(abi.OPT_AO_CHUNK, int(os.environ["CHUNK"]))
    if os.environ.get("HINTS"): ctx.set_option(abi.OPT_TRIP_HINTS, int(os.environ["HINTS"]))
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES)
    import torch
    fluxes = ctx.field_set(FLUX_NAMES)
    fluxes["iterations"] = ctx.zeros(torch.int32)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    ref = None
    if check:
        import oracle as orc
        g = orc.make_grid(nx, ny, h, h, 1)
        at_np = {k: v.cpu().numpy() for k, v in atmos.items()}
        ref = orc.compute_atmosphere_ocean_fluxes(g, params, ocean_np, at_np, nthreads=0)
    out = {}
    for _ in range(4):  # clocks settle (a cold device is ~9 % slower)
        ctx.time_stage(abi.STAGE_AO_FLUXES, 500, ocean=ocean, atmos=atmos, fluxes=fluxes)
    for name, solver in (("lean", abi.SOLVER_TABLES), ("r2", SOLVER_R2), ("lean_r2outer", 3), ("lean", abi.SOLVER_TABLES), ("r2", SOLVER_R2), ("lean_r2outer", 3)):
        ctx.set_option(abi.OPT_SOLVER, solver)
        for _ in range(3):
            ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes)
        ctx.sync()
        t = min(ctx.time_stage(abi.STAGE_AO_FLUXES, reps, ocean=ocean, atmos=atmos, fluxes=fluxes) for _ in range(5))
        rec = dict(ao_us=round(min(t * 1e3, out.get(name, {}).get("ao_us", 1e9)), 2))
        if ref is not None:
            W = (slice(h - 1, h + ny + 1), slice(h - 1, h + nx + 1))
            got = {k: v.cpu().numpy() for k, v in fluxes.items()}
            worst = {}
            for k, s in SCALE.items():
                worst[k] = float(np.max(np.abs(got[k][W] - ref[k][W]) / np.maximum(np.abs(ref[k][W]), s)))
            dit = got["iterations"][W].astype(np.int64) != ref["iterations"][W].astype(np.int64)
            rec.update(worst_scaled=max(worst.values()), worst_field=max(worst, key=worst.get), trip_diff_cells=int(dit.sum()),
                       wet=int((ref["iterations"][W] > 0).sum()))
            same = ~dit
            rec["worst_scaled_same_trip"] = float(max(np.max((np.abs(got[k][W] - ref[k][W]) / np.maximum(np.abs(ref[k][W]), s))[same]) for k, s in SCALE.items()))
        out[name] = rec
    print(cfg, json.dumps(out), flush=True)
    ctx.close()
