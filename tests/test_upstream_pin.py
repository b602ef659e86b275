"""The upstream pin.  The arithmetic of the path lives in NumericalEarth.jl, which this image
cannot run.  `climaocean.jl_amd/julia/oracle_dump.jl` feeds the committed inputs of tests/golden/upstream_inputs/ through
the reference's public API wherever Julia + ClimaOcean exist and writes tests/golden/upstream/*.npy; the tests below hold
the CPU oracle (and, on the GPU box, the HIP path) against those files AS SOON AS THEY EXIST.  Until then they skip with
the reason spelled out, and every report keeps saying "parity unpinned"."""
import glob
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle as orc
import util
from coflux import interface_computations as ic

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INPUTS = os.path.join(HERE, "golden", "upstream_inputs")
UPSTREAM = os.path.join(HERE, "golden", "upstream")
DUMP = os.path.join(ROOT, "climaocean.jl_amd", "julia", "oracle_dump.jl")
TOL = 1e-6   # north star: all six flux fields within 1e-6 relative of the CPU reference
FORMULATIONS = {"default": ic.SimilarityTheoryFluxes, "corrected": ic.corrected_atmosphere_ocean_fluxes,
                "ncar": ic.ncar_atmosphere_ocean_fluxes}
FIELDS = ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum")


def julia_with_reference():
    """True where `julia -e 'using ClimaOcean'` works (never in the build image)."""
    exe = shutil.which("julia")
    if not exe:
        return False
    try:
        return subprocess.run([exe, "-e", "using ClimaOcean"], capture_output=True, timeout=600).returncode == 0
    except Exception:
        return False


def load_inputs():
    nx, ny, h, ring = (int(v) for v in np.load(os.path.join(INPUTS, "shape.npy")))
    ocean = {k: np.load(os.path.join(INPUTS, f"ocean_{k}.npy")) for k in ("T", "S", "u", "v")}
    ocean["mask"] = np.load(os.path.join(INPUTS, "ocean_mask.npy")).astype(np.uint8)
    atmos = {k: np.load(os.path.join(INPUTS, f"atmos_{k}.npy")) for k in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    return nx, ny, h, ring, ocean, atmos


def test_dump_script_and_inputs_are_shipped():
    """The script, its inputs, and the inputs' identity with the golden vectors the oracle is already pinned to."""
    assert os.path.exists(DUMP)
    text = open(DUMP).read()
    for name in ("OceanSeaIceModel", "PrescribedAtmosphere", "SimilarityTheoryFluxes", "ComponentInterfaces", "upstream_inputs"):
        assert name in text, name
    nx, ny, h, ring, ocean, atmos = load_inputs()
    gold = np.load(os.path.join(HERE, "golden", "flux_path_24x12.npz"))
    for k in ("T", "S", "u", "v"):
        np.testing.assert_array_equal(ocean[k], gold["ocean." + k])
    for k in atmos:
        np.testing.assert_array_equal(atmos[k], gold["atmos." + k])


@pytest.mark.skipif(not glob.glob(os.path.join(UPSTREAM, "*.npy")),
                    reason="parity unpinned: tests/golden/upstream/*.npy absent (run climaocean.jl_amd/julia/oracle_dump.jl "
                           "on a box with Julia + ClimaOcean; none exists in this image)")
@pytest.mark.parametrize("name", sorted(FORMULATIONS))
def test_cpu_oracle_matches_upstream(name):
    nx, ny, h, ring, ocean, atmos = load_inputs()
    g = orc.make_grid(nx, ny, h, h, ring)
    P = ic.flux_params(FORMULATIONS[name](), ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    fl = orc.compute_atmosphere_ocean_fluxes(g, P, ocean, atmos, nthreads=1, scales=False)
    net = orc.compute_net_ocean_fluxes(g, P, ocean, atmos, fl)
    inner = (slice(h, h + ny), slice(h, h + nx))
    for k in FIELDS:
        ref = np.load(os.path.join(UPSTREAM, f"{name}_{k}.npy"))
        assert util.rel_err(fl[k][inner], ref, util.FIELD_SCALE[k]) <= TOL, (name, k)
    for k in ("u", "v", "T", "S"):
        ref = np.load(os.path.join(UPSTREAM, f"{name}_net_{k}.npy"))
        assert util.rel_err(net[k][inner], ref, util.FIELD_SCALE[k]) <= TOL, (name, "net." + k)


@pytest.mark.gpu
@pytest.mark.skipif(not glob.glob(os.path.join(UPSTREAM, "*.npy")),
                    reason="parity unpinned: tests/golden/upstream/*.npy absent (see test_cpu_oracle_matches_upstream)")
@pytest.mark.parametrize("name", sorted(FORMULATIONS))
def test_hip_path_matches_upstream(name):
    import torch
    from coflux.runtime import FLUX_NAMES, NET_NAMES, FluxContext
    nx, ny, h, ring, ocean, atmos = load_inputs()
    P = ic.flux_params(FORMULATIONS[name](), ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    ctx = FluxContext(nx, ny, h, h, P, ring=ring)
    oc = {k: ctx.to_device(v) for k, v in ocean.items()}
    at = {k: ctx.to_device(v) for k, v in atmos.items()}
    fl, net = ctx.field_set(FLUX_NAMES), ctx.field_set(NET_NAMES)
    ctx.compute_atmosphere_ocean_fluxes(oc, at, fl)
    ctx.compute_net_ocean_fluxes(oc, at, fl, net)
    ctx.sync()
    inner = (slice(h, h + ny), slice(h, h + nx))
    for k in FIELDS:
        ref = np.load(os.path.join(UPSTREAM, f"{name}_{k}.npy"))
        assert util.rel_err(fl[k].cpu().numpy()[inner], ref, util.FIELD_SCALE[k]) <= TOL, (name, k)
    for k in ("u", "v", "T", "S"):
        ref = np.load(os.path.join(UPSTREAM, f"{name}_net_{k}.npy"))
        assert util.rel_err(net[k].cpu().numpy()[inner], ref, util.FIELD_SCALE[k]) <= TOL, (name, "net." + k)
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# The widened pin: parameters by reflection, the reference's own interpolation, the sea-ice side, the
# land freshwater.  Inputs are committed (tests/golden/make_upstream_inputs.py); the outputs exist only after
# oracle_dump.jl has run somewhere with Julia.
# ---------------------------------------------------------------------------------------------------------------------
def _have(name):
    return os.path.exists(os.path.join(UPSTREAM, name))


def load_small_source():
    nsx, nsy, lon0, dlon, lat0, dlat, tf, dt_snap = np.load(os.path.join(INPUTS, "jra64_grid.npy"))
    from coflux import abi
    src = {v: np.stack([np.load(os.path.join(INPUTS, f"jra64_{v}_{n}.npy")) for n in (1, 2)]).astype(np.float32) for v in abi.JRA55_VARIABLES}
    w = dict(separable=True, fi=np.load(os.path.join(INPUTS, "interp_fi.npy")), fj=np.load(os.path.join(INPUTS, "interp_fj.npy")),
             latitude=np.load(os.path.join(INPUTS, "latitude.npy")))
    return src, w, float(tf)


def test_widened_pin_inputs_are_shipped_and_exercise_the_conventions():
    text = open(DUMP).read()
    for name in ("parameters.json", "interp_", "sea_ice_", "land_net_S", "STATUS.txt", "jvalue", "corrected_ice_ocean_heat_flux",
                 "SeaIceAlbedo", "ThreeEquationHeatFlux", "iterations"):
        assert name in text, name
    src, w, tf = load_small_source()
    assert all(v.shape == (2, 32, 64) for v in src.values()) and tf == 0.37
    assert w["fi"].min() < 0 < w["fi"].max()          # the western cells of the tile wrap: negative fractional indices
    nx, ny, h, ring, ocean, atmos = load_inputs()
    assert w["fi"].shape == (nx + 2 * h,) and w["fj"].shape == (ny + 2 * h,)
    for k in ("concentration", "thickness", "top_temperature", "u", "v"):
        assert np.load(os.path.join(INPUTS, f"ice_{k}.npy")).shape == ocean["T"].shape, k
    # the restatement interpolates this source on these indices without complaint (values are pinned only by the dump)
    g = orc.make_grid(nx, ny, h, h, ring)
    at = orc.interpolate_atmosphere_state(g, src, w, 0, 1, tf)
    assert np.all(np.isfinite(at["T"])) and 200 < at["T"][h:h + ny, h:h + nx].mean() < 320


def _find(d, key, out=None):
    """every value stored under `key` anywhere in a nested dict (field names survive version drift better than paths)"""
    out = [] if out is None else out
    if isinstance(d, dict):
        for k, v in d.items():
            if k == key:
                out.append(v)
            _find(v, key, out)
    elif isinstance(d, list):
        for v in d:
            _find(v, key, out)
    return out


# what this repository assumes, by the field name NumericalEarth is recalled to use: (object in parameters.json, field, value)
PINNED_DEFAULTS = (
    ("SimilarityTheoryFluxes", "von_karman_constant", 0.4), ("SimilarityTheoryFluxes", "gustiness_parameter", 1.0),
    ("SimilarityTheoryFluxes", "minimum_gustiness", ic.SimilarityTheoryFluxes().minimum_gustiness),           # UNVERIFIED default
    ("SimilarityTheoryFluxes", "tolerance", 1e-8), ("SimilarityTheoryFluxes", "maxiter", 100),
    ("corrected_atmosphere_ocean_fluxes", "minimum_gustiness", 0.5),                                          # omip_simulation.jl:44
    ("corrected_atmosphere_sea_ice_fluxes", "minimum_gustiness", 0.2),                                        # :66
    ("ncar_atmosphere_sea_ice_fluxes", "gustiness_parameter", 0.0),                                           # :109
    ("corrected_ice_ocean_heat_flux", "heat_transfer_coefficient", ic.ThreeEquationHeatFlux().heat_transfer_coefficient),
    ("atmosphere_reference_height", None, 10.0), ("atmosphere_boundary_layer_height", None, 600.0),
)


@pytest.mark.skipif(not _have("parameters.json"), reason="parity unpinned: tests/golden/upstream/parameters.json absent (oracle_dump.jl, section parameters)")
def test_default_parameters_match_upstream():
    import json
    params = json.load(open(os.path.join(UPSTREAM, "parameters.json")))
    wrong = []
    for obj, field, value in PINNED_DEFAULTS:
        got = params.get(obj)
        if isinstance(got, str) and got.startswith("unavailable"):
            continue
        found = [got] if field is None else _find(got, field)
        if not found:
            wrong.append((obj, field, "field not found (renamed upstream?)"))
        elif not any(isinstance(v, (int, float)) and abs(v - value) <= 1e-12 * max(1.0, abs(value)) for v in found):
            wrong.append((obj, field, found, "expected", value))
    assert not wrong, wrong


@pytest.mark.skipif(not _have("interp_T.npy"), reason="parity unpinned: tests/golden/upstream/interp_*.npy absent (oracle_dump.jl, section interpolation)")
def test_cpu_oracle_interpolation_matches_upstream():
    nx, ny, h, ring, ocean, atmos = load_inputs()
    src, w, tf = load_small_source()
    at = orc.interpolate_atmosphere_state(orc.make_grid(nx, ny, h, h, ring), src, w, 0, 1, tf)
    inner = (slice(h, h + ny), slice(h, h + nx))
    for k in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp"):
        ref = np.load(os.path.join(UPSTREAM, f"interp_{k}.npy"))
        assert util.rel_err(at[k][inner], ref, util.ATMOS_SCALE[k]) <= TOL, k


@pytest.mark.gpu
@pytest.mark.skipif(not _have("interp_T.npy"), reason="parity unpinned: tests/golden/upstream/interp_*.npy absent")
def test_hip_interpolation_matches_upstream():
    from coflux.runtime import EXCHANGE_NAMES, FluxContext
    nx, ny, h, ring, ocean, atmos = load_inputs()
    src, w, tf = load_small_source()
    ctx = FluxContext(nx, ny, h, h, ic.flux_params(), ring=ring)
    dsrc = {k: ctx.to_device(v) for k, v in src.items()}
    dw = {k: (ctx.to_device(v) if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    at = ctx.field_set(EXCHANGE_NAMES)
    ctx.interpolate_atmosphere_state(dsrc, dw, at, 0, 1, tf)
    ctx.sync()
    inner = (slice(h, h + ny), slice(h, h + nx))
    for k in EXCHANGE_NAMES:
        ref = np.load(os.path.join(UPSTREAM, f"interp_{k}.npy"))
        assert util.rel_err(at[k].cpu().numpy()[inner], ref, util.ATMOS_SCALE[k]) <= TOL, k
    ctx.close()


def _sea_ice_case():
    nx, ny, h, ring, ocean, atmos = load_inputs()
    ice = {k: np.load(os.path.join(INPUTS, f"ice_{k}.npy")) for k in ("concentration", "thickness", "top_temperature", "u", "v")}
    return nx, ny, h, ring, ocean, atmos, ice


@pytest.mark.skipif(not _have("sea_ice_corrected_sensible_heat.npy"),
                    reason="parity unpinned: tests/golden/upstream/sea_ice_*.npy absent (oracle_dump.jl, section sea_ice)")
@pytest.mark.parametrize("name", ["corrected", "ncar"])
def test_cpu_oracle_sea_ice_interface_matches_upstream(name):
    """Also answers DESIGN §5.4's open question with data: if upstream's iteration counts are there, the share of cells it
    leaves at maxiter is compared with the restatement's."""
    nx, ny, h, ring, ocean, atmos, ice = _sea_ice_case()
    make = {"corrected": ic.corrected_atmosphere_sea_ice_fluxes, "ncar": ic.ncar_atmosphere_sea_ice_fluxes}[name]
    g = orc.make_grid(nx, ny, h, h, ring)
    P = ic.flux_params(make())
    got = orc.compute_atmosphere_sea_ice_fluxes(g, P, ic.SeaIceInterfaceProperties().to_params(), ice, ocean, atmos)
    inner = (slice(h, h + ny), slice(h, h + nx))
    wet = ocean["mask"][inner] != 0
    for k in FIELDS:
        ref = np.load(os.path.join(UPSTREAM, f"sea_ice_{name}_{k}.npy"))
        assert util.rel_err(got[k][inner][wet], ref[wet], util.FIELD_SCALE[k]) <= TOL, (name, k)
    ref = np.load(os.path.join(UPSTREAM, f"sea_ice_{name}_skin_temperature.npy"))
    assert np.max(np.abs(got["temperature"][inner][wet] - ref[wet])) <= 1e-5, name
    if _have(f"sea_ice_{name}_iterations.npy"):
        its = np.load(os.path.join(UPSTREAM, f"sea_ice_{name}_iterations.npy"))
        share_up, share_here = float((its[wet] >= 100).mean()), float((got["iterations"][inner][wet] >= 100).mean())
        assert abs(share_up - share_here) <= 0.02, ("share of cells left at maxiter", share_up, share_here)


def _polar_case():
    nx, ny, h, ring = (int(v) for v in np.load(os.path.join(INPUTS, "polar_shape.npy")))
    ld = lambda grp, k: np.load(os.path.join(INPUTS, f"polar_{grp}_{k}.npy"))
    ocean = {k: ld("ocean", k) for k in ("T", "S", "u", "v")}
    ocean["mask"] = ld("ocean", "mask").astype(np.uint8)
    atmos = {k: ld("atmos", k) for k in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")}
    ice = {k: ld("ice", k) for k in ("concentration", "thickness", "top_temperature", "u", "v", "albedo")}
    return nx, ny, h, ring, ocean, atmos, ice


def test_polar_tile_inputs_are_shipped_and_the_restatement_does_not_converge_on_them():
    """The question the polar tile puts to upstream, stated as data: ≥ 1 000 ice-covered wet
    cells under a cold atmosphere, on which THIS repository's restatement of the skin-temperature iteration leaves a large
    share at maxiter.  The dump's sea_ice_polar section writes upstream's answer; the test below compares."""
    text = open(DUMP).read()
    for name in ("sea_ice_polar", "polar_shape.npy", "skin_temperature", "iterations"):
        assert name in text, name
    nx, ny, h, ring, ocean, atmos, ice = _polar_case()
    inner = (slice(h, h + ny), slice(h, h + nx))
    assert nx * ny >= 1000 and np.all(ocean["mask"][inner] != 0) and np.all(ice["concentration"][inner] > 0)
    g = orc.make_grid(nx, ny, h, h, ring)
    got = orc.compute_atmosphere_sea_ice_fluxes(g, ic.flux_params(ic.corrected_atmosphere_sea_ice_fluxes()),
                                                ic.SeaIceInterfaceProperties().to_params(), ice, ocean, atmos)
    its = got["iterations"][inner]
    share = float((its >= 100).mean())
    print(f"polar tile: {100 * share:.1f} % of {its.size} cells at maxiter; iterations min/median/max {its.min()}/{int(np.median(its))}/{its.max()}")
    assert np.all(np.isfinite(got["temperature"][inner])) and share > 0.2      # the finding DESIGN §5.4 reports, reproduced on the tile


@pytest.mark.skipif(not _have("sea_ice_polar_corrected_skin_temperature.npy"),
                    reason="parity unpinned: tests/golden/upstream/sea_ice_polar_*.npy absent (oracle_dump.jl, section sea_ice_polar)")
@pytest.mark.parametrize("name", ["corrected", "ncar"])
def test_cpu_oracle_polar_tile_matches_upstream(name):
    """Which way DESIGN §5.4's fork goes: if upstream converges where the restatement orbits, the histogram comparison fails
    first and says so; if upstream orbits too, converged cells must agree at 1e-6 and the abandoned sets must coincide."""
    nx, ny, h, ring, ocean, atmos, ice = _polar_case()
    make = {"corrected": ic.corrected_atmosphere_sea_ice_fluxes, "ncar": ic.ncar_atmosphere_sea_ice_fluxes}[name]
    got = orc.compute_atmosphere_sea_ice_fluxes(orc.make_grid(nx, ny, h, h, ring), ic.flux_params(make()),
                                                ic.SeaIceInterfaceProperties().to_params(), ice, ocean, atmos)
    inner = (slice(h, h + ny), slice(h, h + nx))
    here = got["iterations"][inner]
    if _have(f"sea_ice_polar_{name}_iterations.npy"):
        up_its = np.load(os.path.join(UPSTREAM, f"sea_ice_polar_{name}_iterations.npy"))
        share_up, share_here = float((up_its >= 100).mean()), float((here >= 100).mean())
        assert abs(share_up - share_here) <= 0.02, (
            f"upstream leaves {100 * share_up:.1f} % of the polar tile at maxiter, the restatement {100 * share_here:.1f} %: "
            "the recalled skin-temperature balance is not upstream's (DESIGN.md §5.4, first row of the decision table)")
        np.testing.assert_array_equal(up_its >= 100, here >= 100)
    ok = here < 100
    for k in FIELDS:
        ref = np.load(os.path.join(UPSTREAM, f"sea_ice_polar_{name}_{k}.npy"))
        assert util.rel_err(got[k][inner][ok], ref[ok], util.FIELD_SCALE[k]) <= TOL, (name, k)
    ref = np.load(os.path.join(UPSTREAM, f"sea_ice_polar_{name}_skin_temperature.npy"))
    assert np.max(np.abs(got["temperature"][inner][ok] - ref[ok])) <= 1e-5, name


@pytest.mark.skipif(not _have("land_net_S.npy"), reason="parity unpinned: tests/golden/upstream/land_net_S.npy absent (oracle_dump.jl, section land)")
def test_land_freshwater_enters_the_salinity_flux_as_upstream_does():
    """M_land inside Mp (ice-masked, one S_min guard with the rain) or outside (this repository)?  The dump decides."""
    nx, ny, h, ring, ocean, atmos = load_inputs()
    src, w, tf = load_small_source()
    g = orc.make_grid(nx, ny, h, h, ring)
    land = {v: np.stack([np.load(os.path.join(INPUTS, f"land_{v}_{n}.npy")) for n in (1, 2)]).astype(np.float32) for v in ("friver", "licalvf")}
    P = ic.flux_params(ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.0))
    fl = orc.compute_atmosphere_ocean_fluxes(g, P, ocean, atmos, nthreads=1, scales=False)
    M = orc.interpolate_land_freshwater(g, land, w, 0, 1, 0.0) if hasattr(orc, "interpolate_land_freshwater") else None
    if M is None:
        pytest.skip("the oracle wrapper has no land interpolation entry point")
    net = orc.compute_net_ocean_fluxes(g, P, ocean, atmos, fl, land=M)
    inner = (slice(h, h + ny), slice(h, h + nx))
    assert util.rel_err(net["S"][inner], np.load(os.path.join(UPSTREAM, "land_net_S.npy")), util.FIELD_SCALE["S"]) <= TOL


def test_probe_reports_the_image_honestly():
    """In this image the probe must say no (no Julia): the reports' "parity unpinned" is not a default but a finding."""
    if shutil.which("julia") is None:
        assert julia_with_reference() is False
