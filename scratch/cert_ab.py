"""Certified vs exact solver path on the 1/4-degree surface: time of the solver stage and of cf_update_state, parity of the
certified path against the C oracle's exact path (all six flux fields), share of cells sent down the exact path, evaluations
per cell.  Scratch tool (GPU)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "climaocean.jl_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
import oracle as orc
nx, ny, h = (int(os.environ.get("NX", 1440)), int(os.environ.get("NY", 560)), 7)
ocean_np = syn.ocean_state(nx, ny, h, h)
src_np = syn.jra55_snapshots(2)
fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
SCALE = dict(sensible_heat=1.0, latent_heat=1.0, water_vapor=1e-6, x_momentum=1e-3, y_momentum=1e-3)
budgets = [int(b) for b in os.environ.get("BUDGETS", "800").split(",")]
for name, mk in (("default", ic.SimilarityTheoryFluxes), ("corrected", ic.corrected_atmosphere_ocean_fluxes)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != name: continue
    P = ic.flux_params(mk())
    ctx = FluxContext(nx, ny, h, h, P)
    ocean = {k: ctx.to_device(ocean_np[k]) for k in ("T", "S", "u", "v", "mask")}
    src = {k: ctx.to_device(v) for k, v in src_np.items()}
    w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
    atmos = ctx.field_set(EXCHANGE_NAMES); fluxes = ctx.field_set(FLUX_NAMES); net = ctx.field_set(NET_NAMES)
    fluxes["iterations"] = ctx.zeros(torch.int32)
    ctx.interpolate_atmosphere_state(src, w, atmos, 0, 1, 0.37)
    out = {}
    ref = None
    for mode in ["exact"] + ["cert%d" % b for b in budgets]:
        if mode == "exact":
            ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_EXACT)
        else:
            ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED); ctx.set_option(abi.OPT_CERTIFIED_BUDGET, int(mode[4:]))
        path = ctx.solver_iteration_path()
        ctx.compute_atmosphere_ocean_fluxes(ocean, atmos, fluxes); ctx.sync()
        got = {k: fluxes[k].cpu().numpy().copy() for k in list(SCALE) + ["iterations"]}
        ao = min(ctx.time_stage(abi.STAGE_AO_FLUXES, 20, ocean=ocean, atmos=atmos, fluxes=fluxes) for _ in range(3))
        fu = min(ctx.time_stage(abi.STAGE_UPDATE_STATE, 20, src=src, weights=w, ocean=ocean, atmos=atmos, fluxes=fluxes, net=net, time_fraction=0.37) for _ in range(3))
        r = dict(path=path, ao_us=round(ao * 1e3, 1), update_state_us=round(fu * 1e3, 1))
        if mode == "exact":
            ref = got
            if os.environ.get("ORACLE", "1") == "1":
                g = orc.make_grid(nx, ny, h, h, 1)
                at = {k: atmos[k].cpu().numpy() for k in EXCHANGE_NAMES}
                o = orc.compute_atmosphere_ocean_fluxes(g, P, ocean_np, at, nthreads=16)
                r["vs_oracle"] = max(float(np.max(np.abs(got[k] - o[k]) / np.maximum(np.abs(o[k]), SCALE[k]))) for k in SCALE)
                ref = {k: o[k] for k in list(SCALE) + ["iterations"]}
        else:
            wet = ocean_np["mask"] != 0
            it = got["iterations"]
            win = np.zeros_like(wet); win[h - 1:h + ny + 1, h - 1:h + nx + 1] = True
            m = wet & win
            ex = (it & abi.CERTIFIED_EXACT_FLAG) != 0
            r["exact_share"] = round(float(ex[m].mean()), 5)
            ev = it[m & ~ex]
            r["evals_mean"] = round(float(ev.mean()), 3); r["evals_hist"] = np.bincount(ev, minlength=11)[3:11].tolist()
            r["exact_trips_mean"] = round(float((it[m & ex] & 0xff).mean()), 2) if ex[m].any() else None
            errs = {k: np.abs(got[k] - ref[k]) / np.maximum(np.abs(ref[k]), SCALE[k]) for k in SCALE}
            r["max_err"] = {k: float("%.3g" % e[m].max()) for k, e in errs.items()}
            worst = np.maximum.reduce([errs[k] for k in SCALE])
            r["max_err_certified_cells"] = float("%.3g" % worst[m & ~ex].max())
            r["max_err_exact_cells"] = float("%.3g" % worst[m & ex].max()) if ex[m].any() else None
            r["cells_beyond_5e-7"] = int((worst[m] > 5e-7).sum())
        out[mode] = r
    print(name, json.dumps(out))
    ctx.close()
