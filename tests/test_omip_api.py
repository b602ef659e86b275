"""The OMIP-side host API of the mirror: `TripolarGrid`, `build_coupled_model(…,
flux_configuration; velocity_formulation)` with the reference's error strings
(/root/reference/src/OMIPConfigurations/omip_simulation.jl:115-164), `omip_forcing` (atmosphere.jl:13-49) and the
JRA55 file side — `RepeatYearJRA55` / `MultiYearJRA55`, start_date / end_date → snapshot records, raw Float32
plane files feeding the sliding HBM window (jra55_data_staging.jl:8,134)."""
import datetime as dt

import numpy as np
import pytest

import oracle as orc
import util
from coflux import abi
from coflux import interface_computations as ic
from coflux import jra55
from coflux import models as cm
from coflux import synthetic as syn


def test_api_surface_of_the_omip_configurations():
    for name in ("TripolarGrid", "build_coupled_model", "omip_forcing", "JRA55PrescribedLand", "NormalizeSalinity"):
        assert hasattr(cm, name), name
    for name in ("RepeatYearJRA55", "MultiYearJRA55", "SnapshotCalendar", "RawPlaneFiles", "ClassicNetCDFFiles", "plane_files",
                 "atmosphere_provider", "JRA55_SHORTNAMES"):
        assert hasattr(jra55, name), name
    assert jra55.JRA55_SHORTNAMES == ("tas", "huss", "psl", "uas", "vas", "rlds", "rsds", "prra", "prsn", "friver", "licalvf")  # jra55_data_staging.jl:8


def test_repeat_year_calendar_wraps_and_multi_year_clamps():
    ry = jra55.SnapshotCalendar(jra55.RepeatYearJRA55(year=1990), dt.datetime(1990, 1, 1), dt.datetime(1995, 1, 1))
    assert ry.total == 2920 and ry.record_of(0) == (1990, 0) and ry.record_of(2919) == (1990, 2919)
    assert ry.record_of(2920) == (1990, 0) and ry.record_of(2920 * 3 + 17) == (1990, 17)      # cyclic: year after year
    # a repeat-year run that starts in May begins at that snapshot of the year
    may = jra55.SnapshotCalendar(jra55.RepeatYearJRA55(year=1990), dt.datetime(1990, 5, 1), dt.datetime(1991, 5, 1))
    assert may.record_of(0) == (1990, (31 + 28 + 31 + 30) * 8) and may.record_of(2920) == may.record_of(0)
    my = jra55.SnapshotCalendar(jra55.MultiYearJRA55(), dt.datetime(1959, 12, 31), dt.datetime(1960, 3, 1, 12))
    # 8 snapshots of 1959-12-31, then 1960 (a leap year) up to March 1st 09:00
    assert my.record_of(0) == (1959, 364 * 8) and my.record_of(8) == (1960, 0)
    assert my.total == 8 + (31 + 29) * 8 + 4 and my.record_of(my.total - 1) == (1960, (31 + 29) * 8 + 3)
    assert my.record_of(-5) == my.record_of(0) and my.record_of(10 ** 6) == my.record_of(my.total - 1)   # clamped
    with pytest.raises(ValueError, match="MultiYearJRA55 covers"):
        jra55.SnapshotCalendar(jra55.MultiYearJRA55(), dt.datetime(1900, 1, 1), dt.datetime(1901, 1, 1))
    with pytest.raises(ValueError, match="not after"):
        jra55.SnapshotCalendar(jra55.RepeatYearJRA55(), dt.datetime(1990, 2, 1), dt.datetime(1990, 1, 1))


def test_raw_plane_files_round_trip(tmp_path):
    snaps = syn.jra55_snapshots(5)
    land = syn.jra55_land_snapshots(5)
    jra55.write_raw_year(str(tmp_path), 1990, snaps, land)
    cal = jra55.SnapshotCalendar(jra55.RepeatYearJRA55(year=1990))
    cal.records, cal.total = [(1990, k) for k in range(5)], 5       # a five-snapshot "year" (the files hold five planes)
    provider = jra55.atmosphere_provider(str(tmp_path), cal)
    for n in (0, 3, 4, 5, 12):                                       # 5 wraps to 0, 12 to 2
        got = provider(n)
        assert set(got) == set(abi.JRA55_VARIABLES)
        for v in abi.JRA55_VARIABLES:
            assert got[v].dtype == np.float32 and got[v].shape == (320, 640)
            np.testing.assert_array_equal(got[v], snaps[v][n % 5])
    lw = jra55.land_snapshots(str(tmp_path), cal, first=3, count=3)
    np.testing.assert_array_equal(lw["friver"], land["friver"][[3, 4, 0]])
    with pytest.raises(FileNotFoundError, match="plane file missing"):
        jra55.RawPlaneFiles(str(tmp_path)).plane("tas", 1991, 0)
    with pytest.raises(IndexError):
        jra55.RawPlaneFiles(str(tmp_path)).plane("tas", 1990, 5)


@pytest.mark.filterwarnings("ignore:Cannot close a netcdf_file")
def test_classic_netcdf_files_feed_the_same_planes(tmp_path):
    """NetCDF classic files (what `nccopy -k classic` makes of the distributed NetCDF-4 ones) through scipy: the same planes
    as the raw-plane reader, scale_factor / add_offset applied, the variable found by shortname or as the only 3-D one."""
    from scipy.io import netcdf_file
    snaps = syn.jra55_snapshots(3)
    for var in abi.JRA55_VARIABLES:
        f = netcdf_file(str(tmp_path / f"{var}_1990.nc"), "w")
        f.createDimension("time", None); f.createDimension("lat", 320); f.createDimension("lon", 640)
        name = var if var != "psl" else "sea_level_pressure"       # one file names its variable differently
        v = f.createVariable(name, "f4", ("time", "lat", "lon"))
        if var == "tas":                                            # packed: value = stored * scale + offset
            v[:] = ((snaps[var] - 250.0) / 0.5).astype("f4"); v.scale_factor = 0.5; v.add_offset = 250.0
        else:
            v[:] = snaps[var]
        f.close()
    cal = jra55.SnapshotCalendar(jra55.RepeatYearJRA55(year=1990))
    cal.records, cal.total = [(1990, k) for k in range(3)], 3
    assert isinstance(jra55.plane_files(str(tmp_path)), jra55.ClassicNetCDFFiles)
    provider = jra55.atmosphere_provider(str(tmp_path), cal)
    for n in (0, 2, 4):
        got = provider(n)
        for var in abi.JRA55_VARIABLES:
            assert got[var].dtype == np.float32 and got[var].shape == (320, 640)
            if var == "tas":
                np.testing.assert_allclose(got[var], snaps[var][n % 3], rtol=1e-6)
            else:
                np.testing.assert_array_equal(got[var], snaps[var][n % 3])
    with pytest.raises(IndexError):
        jra55.ClassicNetCDFFiles(str(tmp_path)).plane("tas", 1990, 3)
    with pytest.raises(FileNotFoundError, match="classic"):
        jra55.ClassicNetCDFFiles(str(tmp_path)).plane("tas", 1991, 0)


def test_build_coupled_model_rejects_unknown_options_with_the_reference_strings():
    # omip_simulation.jl:135-137,160 — the checks come before anything touches the device
    with pytest.raises(ValueError, match=r"Unknown velocity_formulation: sideways\. Options: :relative, :wind"):
        cm.build_coupled_model(None, None, None, None, None, "corrected", velocity_formulation="sideways")
    with pytest.raises(ValueError, match=r"Unknown flux_configuration: shear_aware\. Options: :default, :corrected, :ncar"):
        cm.build_coupled_model(None, None, None, None, None, ":shear_aware")   # launch.sh:350 emits it; build_coupled_model rejects it
    # the formulation itself exists (include/coflux.h: shear_gustiness_coefficient) and is reachable behind an explicit flag
    f = ic.shear_aware_atmosphere_ocean_fluxes()
    assert f.shear_gustiness_coefficient == 0.04 and ic.flux_params(f).shear_gustiness_coefficient == 0.04
    assert ic.flux_params(ic.corrected_atmosphere_ocean_fluxes()).shear_gustiness_coefficient == 0.0


def test_tripolar_grid_weights_fold_and_rotation_without_a_gpu():
    grid = cm.TripolarGrid(size=(72, 36, 4), halo=(3, 3, 2))
    w = grid.interpolation_weights(lambda a: a)
    nx, ny, h = 72, 36, 3
    assert w["separable"] is False and w["fi"].shape == (ny + 2 * h, nx + 2 * h) == grid.surface_shape
    c, s_ = w["cos_rot"], w["sin_rot"]
    np.testing.assert_allclose(c[h:h + ny, h:h + nx] ** 2 + s_[h:h + ny, h:h + nx] ** 2, 1.0, atol=1e-12)
    # the first north halo row is the fold image of the row below the last one, i-axis reversed
    np.testing.assert_array_equal(w["latitude"][h + ny, h:h + nx], w["latitude"][h + ny - 2, h:h + nx][::-1])
    np.testing.assert_array_equal(c[h + ny, h:h + nx], -c[h + ny - 2, h:h + nx][::-1])
    # periodic x-halos
    np.testing.assert_array_equal(w["fi"][:, :h], w["fi"][:, nx:nx + h])
    with pytest.raises(ValueError, match="come together"):
        cm.TripolarGrid(size=(72, 36, 4), halo=(3, 3, 2), longitude=np.zeros((36, 72))).mesh()


@pytest.mark.gpu
def test_one_degree_tripolar_model_matches_the_oracle():
    """BASELINE config 4's grid through the model API: OceanSeaIceModel on TripolarGrid(size = (360, 180, Nz), halo = (5, 5, 4))
    (one_degree_tripolar.jl:32,48-51) — general weights, wind rotation, the fold applied to the ocean state by
    update_state! — against the oracle run on the folded state."""
    nx, ny, nz, h = 360, 180, 10, 5
    grid = cm.TripolarGrid(size=(nx, ny, nz), halo=(h, h, 4))
    case = syn.tripolar_case(nx, ny, h, h)
    ocean = cm.ocean_simulation(grid)
    raw = syn.ocean_state(nx, ny, h, h, latitude=(-80.0, 90.0))      # north halos NOT folded: update_state! must do it
    cm.set_surface(ocean, T=raw["T"], S=raw["S"], u=raw["u"], v=raw["v"], mask=case["ocean"]["mask"])
    snaps = syn.jra55_snapshots(2)
    atmosphere = cm.JRA55PrescribedAtmosphere(snaps)
    coupled = cm.build_coupled_model(ocean, None, atmosphere, None, None, "corrected", velocity_formulation="relative",
                                     ocean_minimum_salinity=0)
    cm.time_step(coupled, 20 * cm.minutes)
    n1, n2, frac = atmosphere.time_indices(coupled.clock.time)
    g = orc.make_grid(nx, ny, h, h, 1)
    params = ic.flux_params(ic.corrected_atmosphere_ocean_fluxes(), velocity_difference=ic.RelativeVelocity(),
                            ocean=ic.OceanProperties(surface_z=grid.surface_z))
    w = grid.interpolation_weights(lambda a: a)
    at = orc.interpolate_atmosphere_state(g, snaps, w, n1, n2, frac)
    fl = orc.compute_atmosphere_ocean_fluxes(g, params, case["ocean"], at)
    net = orc.compute_net_ocean_fluxes(g, params, case["ocean"], at, fl, weights=w)
    bc = ocean.model.top_boundary_conditions
    for name, tensor in (("u", bc.u), ("v", bc.v), ("T", bc.T), ("S", bc.S)):
        e = util.rel_err(util.window(tensor.cpu().numpy(), h, h, nx, ny, 0), util.window(net[name], h, h, nx, ny, 0), util.FIELD_SCALE[name])
        assert e < 1e-9, (name, e)
    Qv = coupled.interfaces.atmosphere_ocean_interface.fluxes.latent_heat.cpu().numpy()
    assert util.rel_err(util.window(Qv, h, h, nx, ny, 1), util.window(fl["latent_heat"], h, h, nx, ny, 1), 1.0) < 1e-9
    coupled.interfaces.context.close()


@pytest.mark.gpu
def test_omip_forcing_from_raw_planes_wraps_the_repeat_year(tmp_path):
    """omip_forcing(arch, sea_ice; forcing_dir, start_date, end_date, repeat_year_forcing = true, backend_size) →
    (atmosphere, radiation, land) on the sliding window: stepping past the end of the (five-snapshot) repeat year
    reads record 0 again, and every step equals the in-memory atmosphere at the wrapped time."""
    snaps, land = syn.jra55_snapshots(5), syn.jra55_land_snapshots(5)
    jra55.write_raw_year(str(tmp_path), 1990, snaps, land)
    atmosphere, radiation, land_c = cm.omip_forcing(None, None, forcing_dir=str(tmp_path), start_date=dt.datetime(1990, 1, 1),
                                                    end_date=dt.datetime(1991, 1, 1), repeat_year_forcing=True, backend_size=3)
    assert radiation.ocean_surface.albedo == 0.06 and radiation.ocean_surface.emissivity == 1.0      # atmosphere.jl:43
    assert isinstance(atmosphere.dataset, jra55.RepeatYearJRA55) and atmosphere.calendar.total == 2920
    atmosphere.calendar.records, atmosphere.calendar.total = [(1990, k) for k in range(5)], 5   # the files hold five planes
    atmosphere.n_levels = land_c.n_levels = 5
    assert land_c.provider is not None and land_c.n_slots == 3 and land_c.calendar is atmosphere.calendar
    nx, ny, nz, h = 90, 40, 10, 3
    grid = cm.LatitudeLongitudeGrid(size=(nx, ny, nz), halo=(h, h, h))
    state = syn.ocean_state(nx, ny, h, h)
    results = []
    for atm, lnd in ((atmosphere, land_c), (cm.JRA55PrescribedAtmosphere(snaps), cm.JRA55PrescribedLand(land))):
        ocean = cm.ocean_simulation(grid)
        cm.set_surface(ocean, T=state["T"], S=state["S"], u=state["u"], v=state["v"], mask=state["mask"])
        coupled = cm.OceanSeaIceModel(ocean, atmosphere=atm, radiation=radiation, land=lnd)
        out = []
        for _ in range(8):
            # 2.5 h per step: leaves the land's three-slot window after step 3 (it used to replay the first
            # `backend_size` snapshots for ever) and crosses the repeat-year wrap (5 × 3 h) at step 6
            cm.time_step(coupled, 150 * cm.minutes)
            out.append(ocean.model.top_boundary_conditions.T.clone())
            out.append(ocean.model.top_boundary_conditions.S.clone())
        results.append(out)
        coupled.interfaces.context.close()
    for a, b in zip(*results):
        assert torch_equal(a, b)
    atmosphere.close()


def test_land_window_streams_the_whole_record_past_its_slots():
    """JRA55PrescribedLand on the provider backend: a record longer than the window is streamed through the
    slots by snapshot counter — cyclic for a repeat year, clamped at the ends of a multi-year record — and the two
    bracketing snapshots never share a slot, also across the wrap of a record whose length is no multiple of the slots."""
    total, slots, ny, nx = 7, 3, 4, 6
    planes = {v: np.arange(total, dtype=np.float32)[:, None, None] * (1.0 if v == "friver" else -2.0) + np.zeros((total, ny, nx), np.float32)
              for v in ("friver", "licalvf")}
    reads = []

    def provider(n):
        reads.append(n)
        return {v: planes[v][n] for v in planes}
    for cyclic in (True, False):
        land = cm.JRA55PrescribedLand(provider=provider, total_snapshots=total, time_indices_in_memory=slots, cyclic=cyclic,
                                      device="cpu", source_size=(nx, ny))
        in_memory = cm.JRA55PrescribedLand(planes, cyclic=cyclic, device="cpu")
        for step in range(-2, 40):
            t = (step * 0.7 + 0.1) * land.time_interval
            l1, l2, frac = land.levels(t)
            m1, m2, mfrac = in_memory.levels(t)
            assert l1 != l2 or m1 == m2
            assert frac == mfrac
            for v in planes:
                assert np.array_equal(land.data[v][l1].numpy(), in_memory.data[v][m1].numpy()), (cyclic, step, v)
                assert np.array_equal(land.data[v][l2].numpy(), in_memory.data[v][m2].numpy()), (cyclic, step, v)
        if not cyclic:   # beyond the record's last snapshot: held there, no blending
            l1, l2, frac = land.levels(100 * land.time_interval)
            assert frac == 0.0 and float(land.data["friver"][l1][0, 0]) == total - 1 == float(land.data["friver"][l2][0, 0])
    assert max(reads) == total - 1 and min(reads) == 0


def torch_equal(a, b):
    import torch
    return bool(torch.equal(a, b))
