"""BASELINE.json's full sizes on the GPU.

Config 2 (1/4° 1440×560) and config 5's surface (1/6° 2160×1080, general weights + rotation) are compared directly
with the C oracle (≈ 1 s and a few seconds on the GPU box's host cores with OpenMP).  Config 5's shape is
also checked through size-independent properties: slab invariance (the latitude-slab decomposition of SURVEY.md §8e
reproduces the single-domain result bit for bit), run-to-run determinism with cold and warm trip-count
hints, exact zeros on land, stress antiparallel to the relative wind, latent heat = ℒᵥ·vapour flux, and the
net-flux assembly recomputed with NumPy from the device's own turbulent fluxes."""
import numpy as np
import pytest
import torch

import numpy_oracle as npo
import oracle as orc
import util
from coflux import interface_computations as ic
from coflux.runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, NET_NAMES, FluxContext
from test_gpu_parity import TOL_LINEAR, TOL_SOLVER, compare, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", ["default", "corrected", "ncar"])
def test_config2_quarter_degree_full_surface_against_the_oracle(config):
    fluxes, vd = util.CONFIGS[config]()
    params = ic.flux_params(fluxes, velocity_difference=vd, ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    case = util.build_case(1440, 560, 7, 7)
    got = run_gpu(case, params, fused=True, ice=True)
    ref = run_oracle(case, params, ice=True)
    compare(case, got, ref, 1)
    np.testing.assert_array_equal(util.window(got["fluxes"]["iterations"], 7, 7, 1440, 560, 1),
                                  util.window(ref["fluxes"]["iterations"], 7, 7, 1440, 560, 1))


def _slab(case, j0, j1):
    """Rows [j0, j1) of a global case as a slab case whose halos are cut out of the global arrays."""
    h = case["hy"]
    rows = slice(j0, j1 + 2 * h)
    cut = lambda a: np.ascontiguousarray(a[rows]) if isinstance(a, np.ndarray) and a.ndim == 2 else a
    w = {k: cut(v) for k, v in case["weights"].items()}
    return dict(nx=case["nx"], ny=j1 - j0, hx=case["hx"], hy=h, src=case["src"], weights=w,
                ocean={k: cut(v) for k, v in case["ocean"].items()}, ice={k: cut(v) for k, v in case["ice"].items()})


def test_config5_sixth_degree_surface_against_the_oracle():
    """BASELINE config 5's surface (1/6° 2160×1080, sixth_degree_tripolar.jl:33-36) compared DIRECTLY: general 2-D
    interpolation weights + wind rotation, `:corrected` fluxes (omip_simulation.jl:40-49), sea-ice partition — the C oracle
    does the 2.3 M cells in a few seconds of host time.  1e-12 on the linear stages, 1e-9 on the
    solver, identical trip counts."""
    nx, ny, h = 2160, 1080, 7
    params = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes(), velocity_difference=ic.RelativeVelocity())
    case = util.build_case(nx, ny, h, h, weights="tripolar")
    got = run_gpu(case, params, fused=True, ice=True)
    ref = run_oracle(case, params, ice=True)
    compare(case, got, ref, 1)
    np.testing.assert_array_equal(util.window(got["fluxes"]["iterations"], h, h, nx, ny, 1),
                                  util.window(ref["fluxes"]["iterations"], h, h, nx, ny, 1))


def test_config5_sixth_degree_shape_properties():
    nx, ny, h = 2160, 1080, 5
    params = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes(), velocity_difference=ic.RelativeVelocity())
    case = util.build_case(nx, ny, h, h, weights="tripolar")
    full = run_gpu(case, params, fused=True, ice=True)
    again = run_gpu(case, params, fused=False, ice=True)          # separate launches, cold hints: same bits
    for grp in ("atmos", "fluxes", "net"):
        for k in full[grp]:
            np.testing.assert_array_equal(full[grp][k], again[grp][k], err_msg=f"{grp}.{k}")

    # slab invariance: 8 latitude slabs (Partition(1, 8), pbs_launch.sh:51) with halos from the global state
    for r in (0, 3, 7):
        j0, j1 = r * ny // 8, (r + 1) * ny // 8
        part = run_gpu(_slab(case, j0, j1), params, fused=True, ice=True)
        for grp, ring in (("atmos", 1), ("fluxes", 1), ("net", 0)):
            for k in part[grp]:
                a = util.window(part[grp][k], h, h, nx, j1 - j0, ring)
                lo = 0 if ring == 0 else 1
                b = full[grp][k][h + j0 - lo:h + j1 + lo, h - lo:h + nx + lo]
                np.testing.assert_array_equal(a, b, err_msg=f"slab {r} {grp}.{k}")

    W = lambda a, ring=1: util.window(a, h, h, nx, ny, ring)
    wet = W(case["ocean"]["mask"]) != 0
    fl, at = full["fluxes"], full["atmos"]
    for k in FLUX_NAMES:
        if k != "temperature":
            assert np.all(W(fl[k])[~wet] == 0.0), k
    assert np.all(np.isfinite(W(fl["sensible_heat"])))
    # stress antiparallel to Δu = uₐ − uₒ (cell-centred ocean velocity from the two faces)
    uo = 0.5 * (case["ocean"]["u"] + np.roll(case["ocean"]["u"], -1, axis=1))
    vo = 0.5 * (case["ocean"]["v"] + np.roll(case["ocean"]["v"], -1, axis=0))
    du, dv = W(at["u"]) - W(uo), W(at["v"]) - W(vo)
    tx, ty = W(fl["x_momentum"]), W(fl["y_momentum"])
    cross = tx * dv - ty * du
    assert np.max(np.abs(cross[wet])) <= 1e-12 * np.max(np.hypot(tx, ty) * np.hypot(du, dv))
    assert np.all((tx * du + ty * dv)[wet] <= 0.0)
    # latent heat = ℒᵥ(Tₐ)·vapour flux with ℒᵥ = LH_v0 + (cp_v − cp_l)(T − T₀)
    th = ic.AtmosphereThermodynamicsParameters()
    Lv = th.reference_vaporization_enthalpy + (th.water_vapor_heat_capacity - th.liquid_water_heat_capacity) * (
        W(at["T"]) - th.reference_temperature)
    sel = wet & (np.abs(W(fl["water_vapor"])) > 1e-9)
    assert np.max(np.abs(W(fl["latent_heat"])[sel] / W(fl["water_vapor"])[sel] / Lv[sel] - 1.0)) < 1e-12

    # net-flux assembly recomputed with NumPy from the device's own turbulent fluxes
    net = npo.net_ocean_fluxes(case["ocean"], at, fl, hx=h, hy=h, ocean_properties=ic.OceanProperties(),
                               albedo=0.06, emissivity=1.0, min_salinity=params.ocean_minimum_salinity,
                               penetrating=bool(params.penetrating_shortwave), ice=case["ice"],
                               sigma=params.stefan_boltzmann)
    for k in ("u", "v", "T", "S"):
        e = util.rel_err(util.window(full["net"][k], h, h, nx, ny, 0), util.window(net[k], h, h, nx, ny, 0),
                         util.FIELD_SCALE[k])
        assert e <= TOL_LINEAR * 10, (k, e)


@pytest.mark.parametrize("scheme", [0, 1])
def test_config3_sea_ice_interface_full_surface_against_the_oracle(scheme):
    """BASELINE config 3 at its full size: the atmosphere–sea-ice interface on the 1440×560 surface under a polar
    atmosphere, both skin-temperature schemes.  One bar for every cell that converges — 1e-9, or the north star's 1e-6 for
    cells that need more than 40 iterations (a weakly contracting orbit amplifies rounding by ≈ 1.3× per iteration) —,
    identical trip counts, and the share of cells each scheme leaves at maxiter is printed, not hidden."""
    from test_gpu_parity import run_ice
    got, ref = run_ice(util.build_case(1440, 560, 7, 7), "sea_ice_corrected", scheme=scheme)
    # cells both sides abandon at maxiter are on a limiter-driven orbit that amplifies rounding without bound: they must be
    # the SAME cells with finite values, their values are not compared (a 1e-3 bar there would be decoration, not parity)
    worst = util.compare_ice_fluxes(got, ref, 1e-9, tol_unconverged=None)
    wet = ref["iterations"] > 0
    print(f"scheme {scheme}: {100 * (ref['iterations'][wet] >= 100).mean():.1f} % of the wet cells at maxiter; worst errors {worst}")


@pytest.mark.parametrize("config,scheme", [("sea_ice_corrected", 0), ("sea_ice_corrected", 1), ("sea_ice_ncar", 0)])
def test_sea_ice_orbit_shortcut_returns_the_bits_of_the_full_iteration(config, scheme):
    """Where the skin-temperature balance does not contract the iteration falls into a period-2 orbit that becomes exact
    in floating point; the kernel then stops and returns the state the remaining iterations up to maxiter would end on
    (CF_OPT_ICE_ORBIT_SHORTCUT).  Same bits as iterating to maxiter — every field, every cell of the 1/4° surface —
    and it must actually fire (most abandoned cells are on such an orbit)."""
    from coflux import abi
    from test_gpu_parity import run_ice
    case = util.build_case(1440, 560, 7, 7)
    fast, ref = run_ice(case, config, scheme=scheme)
    slow, _ = run_ice(case, config, scheme=scheme, options=((abi.OPT_ICE_ORBIT_SHORTCUT, 0),))
    for k in fast:
        np.testing.assert_array_equal(fast[k], slow[k], err_msg=k)
