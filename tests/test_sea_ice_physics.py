"""SeaIceAlbedo(hi, hs, Ts) (CCSM3) and compute_sea_ice_ocean_fluxes! (ThreeEquationHeatFlux with a momentum-based
friction velocity + frazil) — SURVEY §8f rank 1, omip_simulation.jl:71-77, atmosphere.jl:30-44.
CPU: the C oracle against an independently typed NumPy restatement and against known answers of the cited schemes
(Briegleb et al. 2004; Holland & Jenkins 1999 / McPhee et al. 2008).  GPU: the HIP kernels against the oracle."""
import ctypes as C

import numpy as np
import pytest

import numpy_oracle as npo
import oracle as orc
import util
from coflux import abi
from coflux import interface_computations as ic
from coflux import synthetic as syn

NX, NY, H = 131, 67, 3


def albedo_params(**kw):
    lib = abi.load_library()
    p = abi.SeaIceAlbedoParams()
    assert lib.cf_default_sea_ice_albedo_params(C.byref(p)) == 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def ice_ocean_params(**kw):
    lib = abi.load_library()
    p = abi.IceOceanParams()
    assert lib.cf_default_ice_ocean_params(C.byref(p)) == 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def snow_field(shape, seed=11):
    rng = np.random.default_rng(seed)
    return np.where(rng.uniform(size=shape) < 0.4, 0.0, 0.3 * rng.uniform(size=shape) ** 2)


def test_ccsm3_albedo_known_answers_and_restatements_agree():
    A = albedo_params()
    one = lambda hi, hs, Ts: float(orc.sea_ice_albedo(A, np.array([hi]), np.array([hs]), np.array([Ts]))[0])  # noqa: E731
    assert abs(one(2.0, 0.0, -20.0) - 0.5 * (0.78 + 0.36)) < 1e-15          # thick, cold, bare ice: the two band values
    assert abs(one(2.0, 0.0, 0.0) - (0.57 - 0.075)) < 1e-15                 # at the melting point ice darkens by 0.075
    assert abs(one(2.0, 0.0, -0.5) - (0.57 - 0.0375)) < 1e-15               # half way through the 1 K melt range
    assert abs(one(0.0, 0.0, -20.0) - 0.06) < 1e-15                         # vanishing ice: open-ocean albedo
    deep = one(2.0, 10.0, -20.0)
    assert abs(deep - (0.84 * 10 / 10.02 + 0.57 * 0.02 / 10.02)) < 1e-14    # deep cold snow: snow albedo × cover fraction
    assert one(0.1, 0.0, -20.0) < one(0.3, 0.0, -20.0) == one(1.0, 0.0, -20.0)  # grows with thickness up to h_max
    st = syn.sea_ice_state(NX, NY, H, H)
    hs = snow_field(st["thickness"].shape)
    Ts = np.minimum(st["top_temperature"] + 7.5, 0.0)                       # many cells inside the melt range
    got = orc.sea_ice_albedo(A, st["thickness"], hs, Ts)
    np.testing.assert_allclose(got, npo.sea_ice_albedo(st["thickness"], hs, Ts), rtol=0, atol=2e-16)
    assert 0.06 <= got.min() and got.max() <= 0.85


def three_equation_residuals(To, So, Sb, Q, c_o=3991.86795711963):
    Tb = -Q.liquidus_slope * Sb
    lhs = Q.salt_transfer_coefficient * (So - Sb)
    rhs = c_o * Q.heat_transfer_coefficient / Q.latent_heat_of_fusion * (To - Tb) * (Sb - Q.ice_salinity)
    return lhs - rhs


def test_three_equation_fluxes_known_answers_and_restatements_agree():
    case = util.build_case(NX, NY, H, H)
    g = orc.make_grid(NX, NY, H, H, 1)
    P = ic.flux_params()
    Q = ice_ocean_params(time_step=1200.0, top_cell_thickness=5.0)
    oc = dict(case["ocean"])
    oc["T"] = np.where(case["ocean"]["ice_concentration"] > 0, -1.9 + 0.3 * (case["ocean"]["T"] % 1.0), case["ocean"]["T"])
    conc, tx, ty = case["ice"]["concentration"], case["ice"]["x_stress"], case["ice"]["y_stress"]
    out = orc.sea_ice_ocean_fluxes(g, P, Q, oc, conc, tx, ty)
    c = (slice(H, H + NY), slice(H, H + NX))
    E = (slice(H, H + NY), slice(H + 1, H + NX + 1))
    N = (slice(H + 1, H + NY + 1), slice(H, H + NX))
    Qio, Js, Qfr, us, Sb = npo.sea_ice_ocean_fluxes(oc["T"][c], oc["S"][c], conc[c], 0.5 * (tx[c] + tx[E]), 0.5 * (ty[c] + ty[N]),
                                                    dz=5.0, dt=1200.0)
    wet = oc["mask"][c] != 0
    for got, ref, scale in ((out["interface_heat"][c], Qio, 1.0), (out["salt_flux"][c], Js, 1e-7), (out["frazil_heat"][c], Qfr, 1.0),
                            (out["friction_velocity"][c], us, 1e-3)):
        assert util.rel_err(got[wet], ref[wet], scale) < 1e-11
    assert np.all(out["interface_heat"][c][~wet] == 0) and np.all(out["frazil_heat"][c][~wet] == 0)
    # the interface salinity solves the three equations, and lies between the ice and the ocean salinity while melting
    Tf = -Q.liquidus_slope * oc["S"][c]
    T_eff = np.maximum(oc["T"][c], Tf)
    assert np.max(np.abs(three_equation_residuals(T_eff, oc["S"][c], Sb, Q))) < 1e-15
    melting = (T_eff > Tf + 1e-9) & (conc[c] > 0) & wet
    assert melting.any() and np.all(Sb[melting] < oc["S"][c][melting]) and np.all(Sb[melting] > Q.ice_salinity)
    assert np.all(out["interface_heat"][c][melting] > 0) and np.all(out["salt_flux"][c][melting] > 0)
    # water at its freezing point exchanges nothing; water below it makes frazil and then exchanges nothing either
    one = dict(T=np.full((7, 7), -0.054 * 34.0), S=np.full((7, 7), 34.0), u=np.zeros((7, 7)), v=np.zeros((7, 7)),
               mask=np.ones((7, 7), np.uint8))
    g1 = orc.make_grid(3, 3, 2, 2, 1)
    full = np.ones((7, 7))
    r = orc.sea_ice_ocean_fluxes(g1, P, Q, one, full, 1e-4 * full, 0 * full)
    assert np.max(np.abs(r["interface_heat"])) < 1e-9 and np.max(np.abs(r["salt_flux"])) < 1e-16
    one["T"] = one["T"] - 0.1
    r = orc.sea_ice_ocean_fluxes(g1, P, Q, one, full, 1e-4 * full, 0 * full)
    want = 1026.0 * 3991.86795711963 * 5.0 * (-0.1) / 1200.0
    assert abs(r["frazil_heat"][3, 3] - want) < 1e-9 * abs(want) and np.max(np.abs(r["interface_heat"])) < 1e-9
    # heat flux scales with u★ = |τ|^½ (McPhee: Q = ρ c α_h u★ ΔT) and with the concentration
    one["T"] = one["T"] + 0.6
    r1 = orc.sea_ice_ocean_fluxes(g1, P, Q, one, full, 1e-4 * full, 0 * full)
    r4 = orc.sea_ice_ocean_fluxes(g1, P, Q, one, 0.5 * full, 4e-4 * full, 0 * full)
    assert abs(r4["interface_heat"][3, 3] / r1["interface_heat"][3, 3] - 1.0) < 1e-12   # ½ × the concentration, 2 × u★
    assert abs(r1["friction_velocity"][3, 3] - 1e-2) < 1e-15


@pytest.mark.gpu
def test_gpu_albedo_and_ice_ocean_fluxes_match_the_oracle():
    import torch
    from coflux.runtime import FluxContext
    case = util.build_case(NX, NY, H, H)
    g = orc.make_grid(NX, NY, H, H, 1)
    P = ic.flux_params()
    ctx = FluxContext(NX, NY, H, H, P)
    st = case["ice_state"]
    hs = snow_field(st["thickness"].shape)
    Ts = np.minimum(st["top_temperature"] + 7.5, 0.0)
    A = ctx.default_sea_ice_albedo_params()
    out = ctx.zeros()
    ctx.compute_sea_ice_albedo(A, ctx.to_device(st["thickness"]), ctx.to_device(hs), ctx.to_device(Ts), out)
    ctx.sync()
    np.testing.assert_allclose(out.cpu().numpy(), orc.sea_ice_albedo(A, st["thickness"], hs, Ts), rtol=0, atol=1e-15)
    Q = ctx.default_ice_ocean_params(time_step=1200.0, top_cell_thickness=5.0)
    oc = dict(case["ocean"])
    oc["T"] = np.where(case["ocean"]["ice_concentration"] > 0, -1.9 + 0.3 * (case["ocean"]["T"] % 1.0), case["ocean"]["T"])
    ref = orc.sea_ice_ocean_fluxes(g, P, Q, oc, case["ice"]["concentration"], case["ice"]["x_stress"], case["ice"]["y_stress"])
    d_oc = {k: ctx.to_device(oc[k]) for k in ("T", "S", "u", "v", "mask")}
    got = {k: ctx.zeros() for k in ("interface_heat", "salt_flux", "frazil_heat", "friction_velocity")}
    ctx.compute_sea_ice_ocean_fluxes(Q, d_oc, ctx.to_device(case["ice"]["concentration"]), ctx.to_device(case["ice"]["x_stress"]),
                                     ctx.to_device(case["ice"]["y_stress"]), got)
    ctx.sync()
    c = (slice(H, H + NY), slice(H, H + NX))
    for k, scale in (("interface_heat", 1.0), ("salt_flux", 1e-7), ("frazil_heat", 1.0), ("friction_velocity", 1e-3)):
        assert util.rel_err(got[k].cpu().numpy()[c], ref[k][c], scale) < 1e-12, k
    ctx.close()
