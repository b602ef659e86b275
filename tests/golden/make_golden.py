"""Generates tests/golden/flux_path_24x12.npz: seeded inputs + outputs of the flux path on a
24×12 tile for every formulation the reference tree configures.

PARITY UNPINNED: the reference (NumericalEarth.jl via ClimaOcean) cannot be imported or run in this
image (no Julia; the package is un-vendored, Project.toml:21,31-32), so these vectors are produced
by the NumPy restatement oracle/numpy_oracle.py — they pin the C oracle and the HIP path against
regressions and against each other, not against upstream.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy_oracle as npo  # noqa: E402
import util  # noqa: E402
from coflux import interface_computations as ic  # noqa: E402

NX, NY, H, RING = 24, 12, 3, 1
TF = 0.37


def main():
    case = util.build_case(NX, NY, H, H, ny_global=560, j_offset=500)  # a 55°N–58°N band of the 1/4° grid (ocean, land and ice)
    w = case["weights"]
    FI, FJ = np.broadcast_arrays(w["fi"][None, :], w["fj"][:, None])
    win = (slice(H - RING, H + NY + RING), slice(H - RING, H + NX + RING))
    at_w = npo.interpolate_atmosphere_state(case["src"], FI[win], FJ[win], 0, 1, TF)
    atmos = {}
    for k, v in at_w.items():
        full = np.zeros(case["ocean"]["T"].shape)
        full[win] = v
        atmos[k] = full
    out = {}
    for k in ("T", "S", "u", "v", "mask", "ice_concentration", "ice_interface_heat", "ice_salt_flux",
              "ice_x_stress", "ice_y_stress"):
        out["ocean." + k] = case["ocean"][k]
    # the JRA55 window is stored cropped to the rows/columns this tile reads
    fj_lo, fj_hi = int(np.floor(FJ[win].min())), int(np.ceil(FJ[win].max())) + 1
    fi_lo, fi_hi = int(np.floor(FI[win].min())), int(np.ceil(FI[win].max())) + 1
    out["src.crop"] = np.array([fi_lo, fi_hi, fj_lo, fj_hi])
    for k, v in case["src"].items():
        out["src." + k] = v[:, fj_lo:fj_hi + 1, fi_lo:fi_hi + 1]
    out["weights.fi"], out["weights.fj"], out["weights.latitude"] = w["fi"], w["fj"], w["latitude"]
    for k, v in atmos.items():
        out["atmos." + k] = v
    ice = case["ice"]
    for name, make in util.CONFIGS.items():
        fluxes, vd = make()
        fl = npo.atmosphere_ocean_fluxes(fluxes, case["ocean"], atmos, hx=H, hy=H, ring=RING,
                                         thermodynamics=ic.AtmosphereThermodynamicsParameters(),
                                         seawater=ic.SeawaterComposition(), ocean_properties=ic.OceanProperties(),
                                         velocity_difference="wind" if isinstance(vd, ic.WindVelocity) else "relative")
        main = ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature")
        for k, v in fl.items():
            if name == "default" or k in main:
                out[f"fluxes.{name}.{k}"] = v
        if name == "default":
            net = npo.net_ocean_fluxes(case["ocean"], atmos, fl, hx=H, hy=H, ocean_properties=ic.OceanProperties(),
                                       albedo=0.06, ice=None)
            for k, v in net.items():
                out[f"net.{name}.{k}"] = v
            net = npo.net_ocean_fluxes(case["ocean"], atmos, fl, hx=H, hy=H, ocean_properties=ic.OceanProperties(),
                                       albedo=ic.LatitudeDependentAlbedo(), emissivity=0.97, min_salinity=34.0,
                                       penetrating=False, ice=ice, latitude2d=case["ocean"]["latitude"])
            for k, v in net.items():
                out[f"net_ice.{name}.{k}"] = v
    # the same inputs as plain .npy files for climaocean.jl_amd/julia/oracle_dump.jl (Julia reads .npy with twenty lines
    # of code, a zipped .npz would need a package): run THERE, it writes tests/golden/upstream/*.npy
    inp = os.path.join(HERE, "upstream_inputs")
    os.makedirs(inp, exist_ok=True)
    for k in ("T", "S", "u", "v", "mask"):
        np.save(os.path.join(inp, f"ocean_{k}.npy"), np.ascontiguousarray(case["ocean"][k], dtype=np.float64))
    for k, v in atmos.items():
        np.save(os.path.join(inp, f"atmos_{k}.npy"), np.ascontiguousarray(v, dtype=np.float64))
    np.save(os.path.join(inp, "latitude.npy"), np.ascontiguousarray(w["latitude"], dtype=np.float64))
    np.save(os.path.join(inp, "shape.npy"), np.array([NX, NY, H, RING], dtype=np.float64))
    path = os.path.join(HERE, "flux_path_24x12.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
