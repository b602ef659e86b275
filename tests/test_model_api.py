"""The reference's user-facing recipe (README.md:56-77) on the mirrored API, BASELINE config 1 shape
(4° 90×40×10): OceanSeaIceModel(ocean; atmosphere) → time_step! → net fluxes in the ocean's top
boundary conditions, checked against the oracle at the model's clock time."""
import numpy as np
import pytest
import torch

import oracle as orc
import util
from coflux import interface_computations as ic
from coflux import models as cm
from coflux import synthetic as syn


def test_api_surface_mirrors_the_reference_names():
    # test/test_module.jl:11-45 style presence checks
    for name in ("OceanSeaIceModel", "OceanOnlyModel", "ocean_simulation", "JRA55PrescribedAtmosphere",
                 "JRA55PrescribedRadiation", "ComponentInterfaces", "Simulation", "run", "time_step", "update_state",
                 "LatitudeLongitudeGrid"):
        assert hasattr(cm, name), name
    for name in ("SimilarityTheoryFluxes", "MomentumRoughnessLength", "ScalarRoughnessLength",
                 "WindDependentWaveFormulation", "TemperatureDependentAirViscosity", "COARELogarithmicSimilarityProfile",
                 "atmosphere_sea_ice_stability_functions", "large_yeager_stability_functions", "FixedIterations",
                 "RelativeVelocity", "WindVelocity", "SurfaceRadiationProperties"):
        assert hasattr(ic, name), name


@pytest.mark.gpu
def test_readme_recipe_config1_matches_oracle():
    nx, ny, nz, h = 90, 40, 10, 3
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
    ocean = cm.ocean_simulation(grid)
    state = syn.ocean_state(nx, ny, h, h)
    cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
    snaps = syn.jra55_snapshots(4)
    atmosphere = cm.JRA55PrescribedAtmosphere(snaps)
    coupled = cm.OceanSeaIceModel(ocean, atmosphere=atmosphere)
    sim = cm.Simulation(coupled, dt=20 * cm.minutes, stop_iteration=11)
    cm.run(sim)
    assert coupled.clock.iteration == 11 and ocean.model.clock.iteration == 11
    t = coupled.clock.time
    n1, n2, frac = atmosphere.time_indices(t)
    assert (n1, n2) == (1, 2) and abs(frac - (t / (3 * 3600) - 1)) < 1e-12

    g = orc.make_grid(nx, ny, h, h, 1)
    fi, fj, phi = grid.fractional_indices()
    w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
    params = ic.flux_params(ocean=ic.OceanProperties(surface_z=grid.surface_z))
    at = orc.interpolate_atmosphere_state(g, snaps, w, n1, n2, frac)
    fl = orc.compute_atmosphere_ocean_fluxes(g, params, state, at)
    net = orc.compute_net_ocean_fluxes(g, params, state, at, fl, weights=w)
    bc = ocean.model.top_boundary_conditions
    for name, tensor in (("u", bc.u), ("v", bc.v), ("T", bc.T), ("S", bc.S),
                         ("shortwave_surface_flux", ocean.model.shortwave_surface_flux)):
        got = tensor.cpu().numpy()
        assert util.rel_err(util.window(got, h, h, nx, ny, 0), util.window(net[name], h, h, nx, ny, 0),
                            util.FIELD_SCALE[name]) < 1e-9, name
    Qc = coupled.interfaces.atmosphere_ocean_interface.fluxes.sensible_heat.cpu().numpy()  # omip_diagnostics.jl:81
    assert util.rel_err(util.window(Qc, h, h, nx, ny, 1), util.window(fl["sensible_heat"], h, h, nx, ny, 1), 1.0) < 1e-9
    # hfds [W/m²] = JT·ρ·cp (visualize/cache.jl:359-361) is O(100)
    hfds = util.window(bc.T.cpu().numpy(), h, h, nx, ny, 0) * 1026.0 * 3991.86795711963
    assert 10 < np.abs(hfds).max() < 2000
