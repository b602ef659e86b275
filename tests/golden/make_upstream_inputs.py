"""Inputs of the WIDENED upstream pin, written beside the ones make_golden.py ships in
tests/golden/upstream_inputs/ (which stay byte-identical: tests/test_upstream_pin.py holds them to the golden npz):

  jra64_<var>_<n>.npy, jra64_grid.npy   a 64 × 32 two-snapshot Float32 atmosphere on a regular source grid whose first
                                        column sits at λ = 0 (as JRA55's TL319 longitudes do), so that the tile's western
                                        cells interpolate with NEGATIVE fractional indices (periodic wrap), and the time
                                        fraction ñ = 0.37 between the snapshots — pins interpolate_atmosphere_state! (a4)
  interp_fi.npy, interp_fj.npy          the fractional source indices this repository derives for the tile's cells
  ice_<field>.npy                       the sea-ice state of the tile (concentration, thickness, snow thickness, top
                                        temperature, ice velocity) — pins the atmosphere–sea-ice interface, the
                                        three-equation exchange and the CCSM3 albedo (f1)
  land_<var>_<n>.npy                    river + calving freshwater on the 64 × 32 source grid — pins where M_land enters JS

Re-run:  python tests/golden/make_upstream_inputs.py     (deterministic: counter-based generator, seed 20260612)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "climaocean.jl_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from coflux import abi, synthetic as syn  # noqa: E402

NX, NY, H, RING = 24, 12, 3, 1
NSX, NSY = 64, 32
LAT0, LAT1 = -87.1875, 87.1875     # source rows at lat0 + j·Δφ, Δφ = 5.625° (regular; the reference's own interpolation decides)
TF = 0.37
# the tile of make_golden.py: rows 500…511 of the 1/4° grid's 560 rows over φ ∈ (−70, 70): a 55–58°N band, λ ∈ (0, 6)
J0, NYG = 500, 560


def main():
    inp = os.path.join(HERE, "upstream_inputs")
    os.makedirs(inp, exist_ok=True)
    src = syn.jra55_snapshots(2, NSX, NSY)
    land = syn.jra55_land_snapshots(2, NSX, NSY)
    for v in abi.JRA55_VARIABLES:
        for n in range(2):
            np.save(os.path.join(inp, f"jra64_{v}_{n + 1}.npy"), np.ascontiguousarray(src[v][n], dtype=np.float64))
    for v in ("friver", "licalvf"):
        for n in range(2):
            np.save(os.path.join(inp, f"land_{v}_{n + 1}.npy"), np.ascontiguousarray(land[v][n], dtype=np.float64))
    dlam, dphi = 360.0 / NSX, (LAT1 - LAT0) / (NSY - 1)
    np.save(os.path.join(inp, "jra64_grid.npy"), np.array([NSX, NSY, 0.0, dlam, LAT0, dphi, TF, 10800.0]))
    # the tile's cell centres and the fractional indices this repository feeds its kernels for them
    i = np.arange(-H, NX + H)
    j = np.arange(-H, NY + H) + J0
    lam = ((i + 0.5) * 0.25) % 360.0
    lam = np.where(i < 0, (i + 0.5) * 0.25, lam)         # western halo columns: negative longitudes, not wrapped (the wrap is the kernel's)
    phi = -70.0 + (j + 0.5) * 0.25
    np.save(os.path.join(inp, "interp_fi.npy"), np.ascontiguousarray(lam / dlam))
    np.save(os.path.join(inp, "interp_fj.npy"), np.ascontiguousarray((phi - LAT0) / dphi))
    ice = syn.sea_ice_state(NX, NY, H, H, ny_global=NYG, j_offset=J0)
    oc = syn.ocean_state(NX, NY, H, H, ny_global=NYG, j_offset=J0)
    ice["concentration"] = oc["ice_concentration"]
    for k, v in ice.items():
        np.save(os.path.join(inp, f"ice_{k}.npy"), np.ascontiguousarray(v, dtype=np.float64))
    for k in ("x_stress", "y_stress"):
        np.save(os.path.join(inp, f"ice_ocean_{k}.npy"), np.ascontiguousarray(oc["ice_" + k], dtype=np.float64))
    print("wrote", len(os.listdir(inp)), "files to", inp, "(%d KB)" % (sum(os.path.getsize(os.path.join(inp, f)) for f in os.listdir(inp)) // 1024))


if __name__ == "__main__":
    main()
