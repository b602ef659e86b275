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


def test_probe_reports_the_image_honestly():
    """In this image the probe must say no (no Julia): the reports' "parity unpinned" is not a default but a finding."""
    if shutil.which("julia") is None:
        assert julia_with_reference() is False
