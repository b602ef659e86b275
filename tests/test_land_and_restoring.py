"""JRA55PrescribedLand (friver + licalvf → JS) and the MultipleFluxes{flux_field, additional_fluxes} receiver with a
SurfaceFluxRestoring + NormalizeSalinity (atmosphere.jl:46, jra55_data_staging.jl:8, omip_simulation.jl:175-220, 507-523).
CPU: C oracle vs NumPy restatement + known answers.  GPU: kernels vs the oracle, and the model-level sequence."""
import numpy as np
import pytest

import numpy_oracle as npo
import oracle as orc
import util
from coflux import interface_computations as ic
from coflux import synthetic as syn

NX, NY, H = 90, 40, 3


def _case():
    case = util.build_case(NX, NY, H, H)
    g = orc.make_grid(NX, NY, H, H, 1)
    P = ic.flux_params(ocean_minimum_salinity=31.0)
    at = orc.interpolate_atmosphere_state(g, case["src"], case["weights"], 0, 1, 0.37)
    fl = orc.compute_atmosphere_ocean_fluxes(g, P, case["ocean"], at)
    land = syn.jra55_land_snapshots(2)
    return case, g, P, at, fl, land


def test_land_freshwater_reaches_the_salinity_flux():
    case, g, P, at, fl, land = _case()
    Mr = orc.interpolate_land_freshwater(g, land["friver"], land["licalvf"], case["weights"], 0, 1, 0.37)
    w = case["weights"]
    FI, FJ = np.broadcast_arrays(w["fi"][None, :], w["fj"][:, None])
    ref = npo.interpolate_atmosphere_state({**case["src"], "prra": land["friver"], "prsn": land["licalvf"]}, FI, FJ, 0, 1, 0.37)["Mp"]
    win = (slice(H - 1, H + NY + 1), slice(H - 1, H + NX + 1))
    np.testing.assert_allclose(Mr[win], ref[win], rtol=0, atol=1e-18)          # same interpolation as rain + snow
    assert Mr.max() > 1e-5 and (Mr[win] == 0).mean() > 0.5                      # concentrated at a few river mouths
    base = orc.compute_net_ocean_fluxes(g, P, case["ocean"], at, fl, weights=w)
    with_land = orc.compute_net_ocean_fluxes(g, P, case["ocean"], at, fl, weights=w, land=Mr)
    c = (slice(H, H + NY), slice(H, H + NX))
    wet = case["ocean"]["mask"][c] != 0
    So = case["ocean"]["S"][c]
    want = np.where(wet & (So >= 31.0), So * Mr[c] / 1000.0, 0.0)               # JS += −S·(−M/ρ_f); suppressed below the floor
    np.testing.assert_allclose((with_land["S"] - base["S"])[c], want, rtol=0, atol=1e-18)
    for k in ("u", "v", "T"):
        np.testing.assert_array_equal(with_land[k], base[k])
    ref2 = npo.net_ocean_fluxes(case["ocean"], at, fl, hx=H, hy=H, ocean_properties=ic.OceanProperties(), albedo=0.06,
                                min_salinity=31.0, land=Mr)
    assert util.rel_err(with_land["S"][c], ref2["S"][c], util.FIELD_SCALE["S"]) < 1e-12


@pytest.mark.gpu
def test_gpu_land_freshwater_and_restoring_match_the_oracle():
    import torch
    from coflux.runtime import FLUX_NAMES, NET_NAMES, FluxContext
    case, g, P, at, fl, land = _case()
    ctx = FluxContext(NX, NY, H, H, P)
    w = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in case["weights"].items()}
    Mr = ctx.zeros()
    ctx.interpolate_land_freshwater(ctx.to_device(land["friver"]), ctx.to_device(land["licalvf"]), w, Mr, 0, 1, 0.37)
    ctx.sync()
    ref_Mr = orc.interpolate_land_freshwater(g, land["friver"], land["licalvf"], case["weights"], 0, 1, 0.37)
    win = (slice(H - 1, H + NY + 1), slice(H - 1, H + NX + 1))
    np.testing.assert_allclose(Mr.cpu().numpy()[win], ref_Mr[win], rtol=0, atol=1e-17)
    ocean = {k: ctx.to_device(case["ocean"][k]) for k in ("T", "S", "u", "v", "mask")}
    atmos = {k: ctx.to_device(v) for k, v in at.items()}
    dfl = {k: ctx.to_device(fl[k]) for k in FLUX_NAMES}
    ref = orc.compute_net_ocean_fluxes(g, P, case["ocean"], at, fl, weights=case["weights"], land=ref_Mr)
    c = (slice(H, H + NY), slice(H, H + NX))
    for fused in (0, 1):                                   # the separate kernel and the solver's fused epilogue
        net = ctx.field_set(NET_NAMES)
        ctx.set_land_freshwater(Mr)
        if fused:
            from coflux import abi
            ctx.set_option(abi.OPT_FUSED_NET, 1)
            src = {k: ctx.to_device(v) for k, v in case["src"].items()}
            a2, f2 = ctx.field_set(("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")), ctx.field_set(FLUX_NAMES)
            ctx.update_state(src, w, ocean, a2, f2, net, time_fraction=0.37)
        else:
            ctx.compute_net_ocean_fluxes(ocean, atmos, dfl, net, weights=w)
        ctx.sync()
        for k in ("u", "v", "T", "S"):
            assert util.rel_err(net[k].cpu().numpy()[c], ref[k][c], util.FIELD_SCALE[k]) < 1e-9, (fused, k)
    ctx.set_land_freshwater(None)
    # SurfaceFluxRestoring materialised: v_p (S − S★) on wet cells
    target = ctx.to_device(np.full(case["ocean"]["S"].shape, 34.5))
    buf = ctx.zeros()
    ctx.materialize_salinity_restoring(1.0 / 6.0 / 86400.0, target, ocean, buf)
    ctx.sync()
    wet = case["ocean"]["mask"][c] != 0
    want = np.where(wet, (case["ocean"]["S"][c] - 34.5) / 6.0 / 86400.0, 0.0)
    np.testing.assert_allclose(buf.cpu().numpy()[c], want, rtol=1e-15, atol=0)
    ctx.close()


@pytest.mark.gpu
def test_model_with_land_multiple_fluxes_and_normalize_salinity():
    """ocean_simulation(grid; additional_surface_fluxes = (; S = SurfaceFluxRestoring(…))) gives the salinity top BC the
    MultipleFluxes layout; OceanSeaIceModel(…; land) adds river + calving freshwater; NormalizeSalinity(ocean) then removes
    the global mean of (bulk + restoring) from the bulk field (omip_simulation.jl:187-220) — against the oracle's replay."""
    import torch
    from coflux import models as cm
    nx, ny, nz, h = NX, NY, 10, H
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
    state = syn.ocean_state(nx, ny, h, h)
    target = torch.full(grid.surface_shape, 34.5, dtype=torch.float64, device="cuda")
    ocean = cm.ocean_simulation(grid, additional_surface_fluxes=dict(S=cm.SurfaceFluxRestoring(target, piston_velocity=1.0 / 6.0)))
    assert isinstance(ocean.model.top_boundary_conditions.S, cm.MultipleFluxes)
    cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
    snaps, land_snaps = syn.jra55_snapshots(2), syn.jra55_land_snapshots(2)
    atmosphere = cm.JRA55PrescribedAtmosphere(snaps)
    land = cm.JRA55PrescribedLand(land_snaps)
    model = cm.OceanSeaIceModel(ocean, atmosphere=atmosphere, land=land)           # update_state! once
    normalize = cm.NormalizeSalinity(ocean)
    normalize(model)
    model.interfaces.context.sync()

    g = orc.make_grid(nx, ny, h, h, 1)
    fi, fj, phi = grid.fractional_indices()
    w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
    P = ic.flux_params(ocean=ic.OceanProperties(surface_z=grid.surface_z))
    at = orc.interpolate_atmosphere_state(g, snaps, w, 0, 1, 0.0)
    fl = orc.compute_atmosphere_ocean_fluxes(g, P, state, at)
    Mr = orc.interpolate_land_freshwater(g, land_snaps["friver"], land_snaps["licalvf"], w, 0, 1, 0.0)
    net = orc.compute_net_ocean_fluxes(g, P, state, at, fl, weights=w, land=Mr)
    c = (slice(h, h + ny), slice(h, h + nx))
    add = np.zeros_like(state["S"])
    add[c] = np.where(state["mask"][c] != 0, (state["S"][c] - 34.5) / 6.0 / 86400.0, 0.0)
    want, mean = orc.normalize_salinity_flux(g, P, net["S"], state["mask"], additional=add)
    got = ocean.model.top_boundary_conditions.S.flux_field.cpu().numpy()
    assert util.rel_err(got[c], want[c], util.FIELD_SCALE["S"]) < 1e-9
    assert abs(float(normalize.mean_total.cpu()) - mean) < 1e-9 * max(abs(mean), 1e-9)
    wet = state["mask"][c] != 0
    assert abs(np.mean((got + add)[c][wet])) < 1e-12 * np.abs(got[c]).max() + 1e-20      # the combined flux integrates to zero
