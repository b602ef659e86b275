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
  polar_<group>_<field>.npy             a 48 × 24 POLAR tile (64–70°N of the 1/4° grid, every cell wet and ice-covered, a cold
                                        atmosphere already on the ocean grid): 1 152 cells of the atmosphere–sea-ice interface
                                        solve — enough to show whether upstream's skin-temperature iteration, too, leaves most of
                                        an ice pack at maxiter: the dump writes the
                                        iteration histogram, the skin temperature and all five interface fluxes

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
    polar_tile(inp)
    print("wrote", len(os.listdir(inp)), "files to", inp, "(%d KB)" % (sum(os.path.getsize(os.path.join(inp, f)) for f in os.listdir(inp)) // 1024))


PNX, PNY, PJ0 = 48, 24, 536      # rows 536…559 of the 1/4° grid's 560: 64–70°N


def polar_tile(inp):
    import util
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    oc = syn.ocean_state(PNX, PNY, H, H, ny_global=NYG, j_offset=PJ0, land_fraction=False)
    oc["T"] = np.minimum(oc["T"], -1.0 + 0.0 * oc["T"])                      # water under ice: near freezing
    ice = syn.sea_ice_state(PNX, PNY, H, H, ny_global=NYG, j_offset=PJ0)
    i, j = syn.ocean_indices(PNX, PNY, H, H, PJ0)
    ice["concentration"] = np.clip(0.75 + 0.25 * syn.normal("ice", i, j, 5) + 0.0 * oc["T"], 0.3, 1.0)
    ice["thickness"] = np.clip(ice["thickness"], 0.05, 3.0)
    fi, fj, phi = syn.latlon_fractional_indices(PNX, PNY, H, H, ny_global=NYG, j_offset=PJ0)
    g = orc.make_grid(PNX, PNY, H, H, 1)
    at = util.polar_atmosphere(orc.interpolate_atmosphere_state(g, syn.jra55_snapshots(2), dict(separable=True, fi=fi, fj=fj, latitude=phi), 0, 1, TF))
    np.save(os.path.join(inp, "polar_shape.npy"), np.array([PNX, PNY, H, RING], dtype=np.float64))
    for k in ("T", "S", "u", "v", "mask"):
        np.save(os.path.join(inp, f"polar_ocean_{k}.npy"), np.ascontiguousarray(oc[k], dtype=np.float64))
    for k in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp"):
        np.save(os.path.join(inp, f"polar_atmos_{k}.npy"), np.ascontiguousarray(at[k], dtype=np.float64))
    for k in ("concentration", "thickness", "top_temperature", "u", "v", "albedo"):
        np.save(os.path.join(inp, f"polar_ice_{k}.npy"), np.ascontiguousarray(ice[k], dtype=np.float64))


if __name__ == "__main__":
    main()
