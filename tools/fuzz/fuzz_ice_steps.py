"""GPU fuzz, part 6: cf_update_state_sea_ice over a host loop with CF_OPT_MERGED_PREFETCH = 2 (the ocean solve and the next
interpolation as workgroups of the interface solve's launch, net sea-ice fluxes in its epilogue) against the plain sequence —
bitwise in every output — on random sizes, both ice formulations, every ocean formulation, with and without requests."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "climaocean.jl_amd")]
import numpy as np, torch
from coflux import abi, synthetic as syn, interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, NET_NAMES, FluxContext
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for case in range(ncases):
    nx = int(rng.choice([33, 192, 640, 1440, int(rng.integers(4, 1000))])); ny = int(rng.choice([5, 48, 141, 280, int(rng.integers(3, 200))])); h = int(rng.integers(2, 6))
    n_levels = int(rng.integers(2, 5)); inc = float(rng.choice([1 / 9, 1 / 3, 0.37])); nsteps = int(rng.integers(1, 8)); request = bool(rng.integers(0, 4))
    ocean_cfg = [ic.SimilarityTheoryFluxes, ic.corrected_atmosphere_ocean_fluxes, ic.ncar_atmosphere_ocean_fluxes][int(rng.integers(0, 3))]
    ice_cfg = [ic.corrected_atmosphere_sea_ice_fluxes, ic.ncar_atmosphere_sea_ice_fluxes][int(rng.integers(0, 2))]
    desc = dict(nx=nx, ny=ny, h=h, n_levels=n_levels, inc=inc, nsteps=nsteps, request=request, ocean=ocean_cfg.__name__, ice=ice_cfg.__name__)
    try:
        results = []
        for tail in (False, True):
            ctx = FluxContext(nx, ny, h, h, ic.flux_params(ocean_cfg()), ring=1)
            ctx.set_sea_ice_formulation(ic.flux_params(ice_cfg()))
            o0 = syn.ocean_state(nx, ny, h, h); o1 = syn.evolved_ocean_state(o0, nx, ny, h, h, 1)
            states = [{k: ctx.to_device(o[k]) for k in ("T", "S", "u", "v", "mask")} for o in (o0, o1)]
            states[1]["mask"] = states[0]["mask"]
            src = {k: ctx.to_device(v) for k, v in syn.jra55_snapshots(n_levels).items()}
            fi, fj, phi = syn.latlon_fractional_indices(nx, ny, h, h)
            w = dict(separable=True, fi=ctx.to_device(fi), fj=ctx.to_device(fj), latitude=ctx.to_device(phi))
            si = syn.sea_ice_state(nx, ny, h, h)
            ice = {k: ctx.to_device(o0["ice_" + k]) for k in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")}
            ice_state = dict(concentration=ice["concentration"], **{k: ctx.to_device(si[k]) for k in ("thickness", "top_temperature", "u", "v", "albedo")})
            sets = [ctx.field_set(EXCHANGE_NAMES) for _ in range(2)]
            fl, net, ai = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES), ctx.field_set(FLUX_NAMES)
            net_ice = ctx.field_set(("top_heat", "bottom_heat"))
            ai["temperature"].copy_(ice_state["top_temperature"]); ice_state["top_temperature"] = ai["temperature"]
            if tail:
                ctx.set_option(abi.OPT_MERGED_PREFETCH, 2)
            for s in range(nsteps):
                tot = s * inc; l1 = int(tot) % n_levels
                if tail and request:
                    nxt = (s + 1) * inc; l1n = int(nxt) % n_levels
                    ctx.prefetch_atmosphere_state(src, w, sets[(s + 1) % 2], level1=l1n, level2=(l1n + 1) % n_levels, time_fraction=nxt - int(nxt))
                ctx.update_state_sea_ice(src, w, states[s % 2], sets[s % 2], fl, net, ice, ice_state, ai, net_ice,
                                         level1=l1, level2=(l1 + 1) % n_levels, time_fraction=tot - int(tot))
            ctx.sync()
            out = {}
            for name, d in (("atmos", sets[(nsteps - 1) % 2]), ("fl", fl), ("net", net), ("ai", ai), ("net_ice", net_ice)):
                for k, v in d.items():
                    out[f"{name}.{k}"] = v.clone()
            results.append(out); ctx.close()
        diff = [k for k in results[0] if not torch.equal(results[0][k], results[1][k])]
        if diff:
            bad += 1; print("MISMATCH", desc, diff[:6])
    except Exception as e:  # noqa: BLE001
        bad += 1; print("ERROR", desc, repr(e)[:300])
print(f"{ncases - bad} of {ncases} cases passed")
