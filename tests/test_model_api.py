"""The reference's user-facing recipe (README.md:56-77) on the mirrored API, BASELINE config 1 shape
(4° 90×40×10): OceanSeaIceModel(ocean; atmosphere) → time_step! → net fluxes in the ocean's top
boundary conditions, checked against the oracle at the model's clock time."""
import numpy as np
import pytest
import torch

import oracle as orc
import util
from coflux import abi
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


@pytest.mark.gpu
def test_run_requests_every_next_state_and_gives_the_bits_of_the_plain_loop():
    """run!(simulation) knows the clock one step ahead: update_state! requests the next atmosphere state with every step
    (it rides in the solver launch's tail workgroups, CF_OPT_MERGED_PREFETCH = 2, through a second set of exchange fields).
    Same bits as time_step! called in a plain loop — net fluxes, interface fluxes and the exchange state itself —
    across a snapshot boundary."""
    import torch
    nx, ny, nz, h = 90, 40, 10, 3
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
    state = syn.ocean_state(nx, ny, h, h)
    snaps = syn.jra55_snapshots(4)
    got = []
    for piped in (True, False):
        ocean = cm.ocean_simulation(grid)
        cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
        coupled = cm.OceanSeaIceModel(ocean, atmosphere=cm.JRA55PrescribedAtmosphere(snaps))
        if piped:
            held = coupled.interfaces.exchange_atmosphere_state        # what an output writer would hold on to
            cm.run(cm.Simulation(coupled, dt=50 * cm.minutes, stop_iteration=8))
            assert coupled.interfaces._exchange_other is not None
            # outside run! the public field set is the one handed out before it, with the state at the model's clock in it
            assert coupled.interfaces.exchange_atmosphere_state is held
            # ... and a step taken by hand afterwards does not mistake the state run! had requested ahead for its own
            cm.time_step(coupled, 50 * cm.minutes)
            coupled.interfaces.context.sync()
            assert coupled.interfaces.exchange_atmosphere_state is held
        else:
            for _ in range(9):
                cm.time_step(coupled, 50 * cm.minutes)
            coupled.interfaces.context.sync()
            assert coupled.interfaces._exchange_other is None
        bc = ocean.model.top_boundary_conditions
        fields = dict(u=bc.u, v=bc.v, T=bc.T, S=bc.S, Qc=coupled.interfaces.atmosphere_ocean_interface.fluxes.sensible_heat,
                      **{"atmos_" + k: v for k, v in coupled.interfaces.exchange_atmosphere_state.items()})
        got.append({k: v.clone() for k, v in fields.items()})
        coupled.interfaces.context.close()
    for k in got[0]:
        assert torch.equal(got[0][k], got[1][k]), k


@pytest.mark.gpu
def test_config3_sea_ice_coupling_through_the_model_api():
    """BASELINE config 3: OceanSeaIceModel(ocean, sea_ice; atmosphere) with the :corrected interfaces
    (omip_simulation.jl:139-147).  Ocean partition with the ice-concentration mask, atmosphere–sea-ice interface
    with skin temperature, net sea-ice fluxes; the skin temperature is carried from step to step."""
    import torch
    nx, ny, nz, h = 90, 40, 10, 3
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
    ocean = cm.ocean_simulation(grid)
    state = syn.ocean_state(nx, ny, h, h)
    cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
    ice_np = syn.sea_ice_state(nx, ny, h, h)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda")
    sea_ice = cm.PrescribedSeaIce(concentration=dev(state["ice_concentration"]), interface_heat=dev(state["ice_interface_heat"]),
                                  salt_flux=dev(state["ice_salt_flux"]), x_stress=dev(state["ice_x_stress"]),
                                  y_stress=dev(state["ice_y_stress"]), thickness=dev(ice_np["thickness"]),
                                  top_surface_temperature=dev(ice_np["top_temperature"]), u=dev(ice_np["u"]),
                                  v=dev(ice_np["v"]), albedo=dev(ice_np["albedo"]))
    snaps = syn.jra55_snapshots(2)
    atmosphere = cm.JRA55PrescribedAtmosphere(snaps)
    interfaces = cm.ComponentInterfaces(atmosphere, ocean, sea_ice,
                                        atmosphere_ocean_fluxes=ic.corrected_atmosphere_ocean_fluxes(),
                                        atmosphere_sea_ice_fluxes=ic.corrected_atmosphere_sea_ice_fluxes(),
                                        atmosphere_ocean_velocity_difference=ic.RelativeVelocity(),
                                        atmosphere_sea_ice_velocity_difference=ic.RelativeVelocity(),
                                        ocean_minimum_salinity=1.0, store_similarity_scales=True)
    model = cm.OceanSeaIceModel(ocean, sea_ice, atmosphere=atmosphere, interfaces=interfaces)   # update_state! once
    Ts1 = sea_ice.top_surface_temperature.cpu().numpy().copy()

    # oracle replay of that first update_state!
    g = orc.make_grid(nx, ny, h, h, 1)
    fi, fj, phi = grid.fractional_indices()
    w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
    props = ic.OceanProperties(surface_z=grid.surface_z)
    at = orc.interpolate_atmosphere_state(g, snaps, w, 0, 1, 0.0)
    p_ao = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes(), velocity_difference=ic.RelativeVelocity(), ocean=props,
                          ocean_minimum_salinity=1.0)
    fl = orc.compute_atmosphere_ocean_fluxes(g, p_ao, state, at)
    ice_fields = dict(concentration=state["ice_concentration"], interface_heat=state["ice_interface_heat"],
                      salt_flux=state["ice_salt_flux"], x_stress=state["ice_x_stress"], y_stress=state["ice_y_stress"])
    net = orc.compute_net_ocean_fluxes(g, p_ao, state, at, fl, ice=ice_fields, weights=w)
    p_ai = ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes(), velocity_difference=ic.RelativeVelocity(), ocean=props)
    iprops = ic.SeaIceInterfaceProperties().to_params()
    ice_state = dict(ice_np, concentration=state["ice_concentration"])
    ai = orc.compute_atmosphere_sea_ice_fluxes(g, p_ai, iprops, ice_state, state, at)
    nsi = orc.compute_net_sea_ice_fluxes(g, p_ai, iprops, ice_state, state, at, ai, None, state["ice_interface_heat"])

    W = lambda a, r=0: util.window(a, h, h, nx, ny, r)
    bc = ocean.model.top_boundary_conditions
    for name, tensor in (("u", bc.u), ("v", bc.v), ("T", bc.T), ("S", bc.S)):
        assert util.rel_err(W(tensor.cpu().numpy()), W(net[name]), util.FIELD_SCALE[name]) < 1e-9, name
    got_ai = {k: W(getattr(model.interfaces.atmosphere_sea_ice_interface.fluxes, k).cpu().numpy(), 1)
              for k in util.ICE_FLUX_FIELDS}
    got_ai["iterations"] = W(ai["iterations"], 1)      # (not stored by the model: compare the fields only)
    util.compare_ice_fluxes(got_ai, {k: W(ai[k], 1) for k in list(util.ICE_FLUX_FIELDS) + ["iterations"]}, 1e-9)
    conv = W(ai["iterations"]) < 100
    assert util.rel_err(W(Ts1)[conv], W(ai["temperature"])[conv], 1.0) < 1e-9
    top = model.interfaces.net_fluxes.sea_ice.top_heat.cpu().numpy()
    bot = model.interfaces.net_fluxes.sea_ice.bottom_heat.cpu().numpy()
    assert util.rel_err(W(top)[conv], W(nsi["top_heat"])[conv], 1.0) < 1e-8
    assert util.rel_err(W(bot), W(nsi["bottom_heat"]), 1.0) < 1e-12

    # the skin temperature is the next step's first guess: a second step changes it only where it had not converged
    cm.time_step(model, 20 * cm.minutes)
    Ts2 = sea_ice.top_surface_temperature.cpu().numpy()
    assert np.isfinite(Ts2).all() and np.all(W(Ts2)[W(state["mask"]) != 0] <= 0.0)
    assert model.clock.iteration == 1


class _FakeWindow:
    """Host-side stand-in for runtime.SnapshotWindow: records what the sliding-window logic asks for."""

    def __init__(self, ctx, nsx, nsy, n_slots):
        self.n_slots, self.nsx, self.nsy = n_slots, nsx, nsy
        self.slots = [None] * n_slots          # committed snapshot per slot
        self.staging = [dict() for _ in range(n_slots)]
        self.log = []

    def host_view(self, slot, var):
        return self.staging[slot].setdefault(var, np.zeros((self.nsy, self.nsx), np.float32))

    def wait_slot(self, slot):
        self.log.append(("wait", slot))

    def commit(self, slot, n):
        assert slot == n % self.n_slots
        self.slots[slot] = (n, float(self.staging[slot]["tas"][0, 0]))
        self.log.append(("commit", slot, n))


    def find(self, n):
        s = self.slots[n % self.n_slots]
        return n % self.n_slots if s is not None and s[0] == n else -1

    def source(self, k1, k2, frac):
        assert self.find(k1) >= 0 and self.find(k2) >= 0, (k1, k2, self.slots)
        # the staged data really is the record asked for (the provider writes its record index into tas)
        return ("src", self.slots[k1 % self.n_slots][1], self.slots[k2 % self.n_slots][1], frac)

    def close(self):
        pass


@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("n_slots", [2, 3, 5])
def test_sliding_window_bookkeeping_without_a_gpu(monkeypatch, prefetch, n_slots):
    """JRA55PrescribedAtmosphere(provider=…): the two bracketing snapshots are always resident when asked for, a slot
    is never refilled while it holds one of them, the repeat-year record wraps, and prefetching reads ahead."""
    from coflux import runtime
    monkeypatch.setattr(runtime, "SnapshotWindow", _FakeWindow)
    total, reads = 7, []

    def provider(n):
        reads.append(n)
        return {v: np.full((4, 8), float(n), np.float32) for v in abi.JRA55_VARIABLES}

    atm = cm.JRA55PrescribedAtmosphere(provider=provider, total_snapshots=total, time_indices_in_memory=n_slots,
                                       prefetch=prefetch, source_size=(8, 4))
    dt, t = 20 * cm.minutes, 0.0
    for step in range(120):                       # 40 h: wraps the 7-snapshot record almost twice
        src, n1, n2, frac = atm.source(None, t)
        n = int(np.floor(t / (3 * 3600)))
        assert (n1, n2) == (n % total, (n + 1) % total) and abs(frac - (t / (3 * 3600) - n)) < 1e-12
        assert src == ("src", float(n1), float(n2), frac)
        t += dt
    assert set(reads) == set(range(total))
    assert len(reads) <= 16 + (n_slots if prefetch else 0)        # ≈ one read per 3-hourly snapshot crossed, no thrashing
    assert (atm._reader is not None) == (prefetch and n_slots > 2)
    atm.close()


@pytest.mark.gpu
def test_config3_with_three_equation_exchange_and_ccsm3_albedo():
    """BASELINE config 3 with nothing supplied from outside: sea_ice_ocean_heat_flux = ThreeEquationHeatFlux(…) computes
    Q_io, Jˢ_io and frazil from the ocean surface and the ice–ocean stress (omip_simulation.jl:71-77,145), the sea-ice
    albedo is SeaIceAlbedo(hi, hs, Ts) from the live ice / snow fields (atmosphere.jl:30-44).  One update_state! against
    the oracle's replay of the same sequence."""
    import torch
    nx, ny, nz, h = 90, 40, 10, 3
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
    ocean = cm.ocean_simulation(grid)
    state = syn.ocean_state(nx, ny, h, h)
    state["T"] = np.where(state["ice_concentration"] > 0, np.minimum(state["T"], -1.5 - 0.5 * (state["S"] % 1.0)), state["T"])
    cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
    ice_np = syn.sea_ice_state(nx, ny, h, h)
    snow = 0.2 * (ice_np["albedo"] - 0.3)           # some deterministic snow cover, 0 … 0.12 m
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda")  # noqa: E731
    sea_ice = cm.PrescribedSeaIce(concentration=dev(state["ice_concentration"]), x_stress=dev(state["ice_x_stress"]),
                                  y_stress=dev(state["ice_y_stress"]), thickness=dev(ice_np["thickness"]),
                                  top_surface_temperature=dev(ice_np["top_temperature"]), u=dev(ice_np["u"]),
                                  v=dev(ice_np["v"]), snow_thickness=dev(snow))
    snaps = syn.jra55_snapshots(2)
    atmosphere = cm.JRA55PrescribedAtmosphere(snaps)
    interfaces = cm.ComponentInterfaces(atmosphere, ocean, sea_ice,
                                        atmosphere_ocean_fluxes=ic.corrected_atmosphere_ocean_fluxes(),
                                        atmosphere_sea_ice_fluxes=ic.corrected_atmosphere_sea_ice_fluxes(),
                                        sea_ice_ocean_heat_flux=ic.corrected_ice_ocean_heat_flux(),
                                        sea_ice_albedo=ic.SeaIceAlbedo(), time_step=20 * cm.minutes, store_similarity_scales=True)
    model = cm.OceanSeaIceModel(ocean, sea_ice, atmosphere=atmosphere, interfaces=interfaces)   # update_state! once

    g = orc.make_grid(nx, ny, h, h, 1)
    fi, fj, phi = grid.fractional_indices()
    w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
    props = ic.OceanProperties(surface_z=grid.surface_z)
    at = orc.interpolate_atmosphere_state(g, snaps, w, 0, 1, 0.0)
    p_ao = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes(), ocean=props)
    Q = ic.corrected_ice_ocean_heat_flux().to_params(300.0, 1200.0)
    io = orc.sea_ice_ocean_fluxes(g, p_ao, Q, state, state["ice_concentration"], state["ice_x_stress"], state["ice_y_stress"])
    fl = orc.compute_atmosphere_ocean_fluxes(g, p_ao, state, at)
    ice_fields = dict(concentration=state["ice_concentration"], interface_heat=io["interface_heat"], salt_flux=io["salt_flux"],
                      x_stress=state["ice_x_stress"], y_stress=state["ice_y_stress"])
    net = orc.compute_net_ocean_fluxes(g, p_ao, state, at, fl, ice=ice_fields, weights=w)
    alb = orc.sea_ice_albedo(ic.SeaIceAlbedo().to_params(), ice_np["thickness"], snow, ice_np["top_temperature"])
    p_ai = ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes(), ocean=props)
    iprops = ic.SeaIceInterfaceProperties().to_params()
    ice_state = dict(ice_np, concentration=state["ice_concentration"], albedo=alb)
    ai = orc.compute_atmosphere_sea_ice_fluxes(g, p_ai, iprops, ice_state, state, at)
    nsi = orc.compute_net_sea_ice_fluxes(g, p_ai, iprops, ice_state, state, at, ai, io["frazil_heat"], io["interface_heat"])

    W = lambda a, r=0: util.window(a, h, h, nx, ny, r)  # noqa: E731
    f = model.interfaces.sea_ice_ocean_fluxes
    for k, scale in (("interface_heat", 1.0), ("salt_flux", 1e-7), ("frazil_heat", 1.0)):
        assert util.rel_err(W(f[k].cpu().numpy()), W(io[k]), scale) < 1e-12, k
    assert np.abs(W(io["interface_heat"])).max() > 1.0 and (W(io["frazil_heat"]) < 0).any()      # both mechanisms active
    bc = ocean.model.top_boundary_conditions
    for name, tensor in (("u", bc.u), ("v", bc.v), ("T", bc.T), ("S", bc.S)):
        assert util.rel_err(W(tensor.cpu().numpy()), W(net[name]), util.FIELD_SCALE[name]) < 1e-9, name
    got_ai = {k: W(getattr(model.interfaces.atmosphere_sea_ice_interface.fluxes, k).cpu().numpy(), 1) for k in util.ICE_FLUX_FIELDS}
    got_ai["iterations"] = W(ai["iterations"], 1)
    util.compare_ice_fluxes(got_ai, {k: W(ai[k], 1) for k in list(util.ICE_FLUX_FIELDS) + ["iterations"]}, 1e-9)
    conv = W(ai["iterations"]) < 100
    top = model.interfaces.net_fluxes.sea_ice.top_heat.cpu().numpy()
    bot = model.interfaces.net_fluxes.sea_ice.bottom_heat.cpu().numpy()
    assert util.rel_err(W(top)[conv], W(nsi["top_heat"])[conv], 1.0) < 1e-8
    assert util.rel_err(W(bot), W(nsi["bottom_heat"]), 1.0) < 1e-12


@pytest.mark.gpu
def test_model_runs_on_the_certified_solver_path_within_its_budget():
    """ComponentInterfaces(…; solver_path = "certified"): run!(simulation) with the tail-workgroup pipeline on the certified
    path against the same run on the exact path — net fluxes and interface fluxes within 1e-6 of the components' scales,
    and the context says which path ran."""
    import torch
    from coflux import abi
    nx, ny, nz, h = 360, 140, 10, 4
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h), latitude=(-70, 70), z=(-3000, 0))
    state = syn.ocean_state(nx, ny, h, h)
    snaps = syn.jra55_snapshots(4)
    out = {}
    for path in ("exact", "certified"):
        ocean = cm.ocean_simulation(grid)
        cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
        atmosphere = cm.JRA55PrescribedAtmosphere(snaps)
        itf = cm.ComponentInterfaces(atmosphere, ocean, solver_path=path)
        coupled = cm.OceanSeaIceModel(ocean, atmosphere=atmosphere, interfaces=itf)
        assert itf.context.solver_iteration_path() == (abi.SOLVER_PATH_CERTIFIED if path == "certified" else abi.SOLVER_PATH_EXACT)
        cm.run(cm.Simulation(coupled, dt=30 * cm.minutes, stop_iteration=5))
        f = itf.atmosphere_ocean_interface.fluxes
        out[path] = {k: getattr(f, k).cpu().numpy().copy() for k in ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum")}
        itf.context.close()
    for k, scale in (("sensible_heat", 1.0), ("latent_heat", 1.0), ("water_vapor", 1e-6), ("x_momentum", 1e-3), ("y_momentum", 1e-3)):
        a, b = out["certified"][k], out["exact"][k]
        err = np.abs(a - b) / np.maximum(np.abs(b), scale)
        assert err.max() <= 1e-6, (k, float(err.max()))
        assert err.max() > 0, k     # (it IS another path)
