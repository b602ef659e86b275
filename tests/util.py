"""Shared builders for the parity tests: seeded synthetic inputs (coflux.synthetic), the oracle
run on them, and the comparison metric  |Δ| ≤ tol · max(|ref|, field scale)  (SURVEY.md §7 H4:
a pure relative error blows up where a flux crosses zero)."""
import numpy as np

from coflux import interface_computations as ic
from coflux import synthetic as syn

# field scales used as the floor of the relative-error denominator
FIELD_SCALE = dict(sensible_heat=1.0, latent_heat=1.0, water_vapor=1e-6, x_momentum=1e-3, y_momentum=1e-3,
                   temperature=1.0, friction_velocity=1e-3, temperature_scale=1e-3, humidity_scale=1e-6,
                   u=1e-6, v=1e-6, T=1e-6, S=1e-7, shortwave_surface_flux=1e-6, upwelling_longwave=1.0,
                   downwelling_longwave=1.0, downwelling_shortwave=1.0,
                   p=1.0, q=1e-4, Qs=1.0, Ql=1.0, Mp=1e-6)
ATMOS_SCALE = dict(u=1.0, v=1.0, T=1.0, p=1.0, q=1e-4, Qs=1.0, Ql=1.0, Mp=1e-6)

CONFIGS = {
    "default": lambda: (ic.SimilarityTheoryFluxes(), None),
    "corrected": lambda: (ic.corrected_atmosphere_ocean_fluxes(), ic.RelativeVelocity()),
    "corrected_wind": lambda: (ic.corrected_atmosphere_ocean_fluxes(), ic.WindVelocity()),
    "sea_ice_corrected": lambda: (ic.corrected_atmosphere_sea_ice_fluxes(), None),
    "sea_ice_ncar": lambda: (ic.ncar_atmosphere_sea_ice_fluxes(), None),
    "ncar": lambda: (ic.ncar_atmosphere_ocean_fluxes(), None),
    "fixed5": lambda: (ic.SimilarityTheoryFluxes(solver_stop_criteria=ic.FixedIterations(5)), None),
}


def build_case(nx, ny, hx=3, hy=3, *, weights="latlon", land=True, n_levels=2, ny_global=None, j_offset=0):
    ocean = syn.ocean_state(nx, ny, hx, hy, ny_global=ny_global, j_offset=j_offset, land_fraction=land)
    src = syn.jra55_snapshots(n_levels)
    if weights == "latlon":
        fi, fj, phi = syn.latlon_fractional_indices(nx, ny, hx, hy, ny_global=ny_global, j_offset=j_offset)
        w = dict(separable=True, fi=fi, fj=fj, latitude=phi)
    else:
        fi, fj, c, s, phi = syn.tripolar_like_weights(nx, ny, hx, hy, ny_global=ny_global, j_offset=j_offset)
        w = dict(separable=False, fi=fi, fj=fj, cos_rot=c, sin_rot=s, latitude=phi)
    ice = dict(concentration=ocean["ice_concentration"], interface_heat=ocean["ice_interface_heat"],
               salt_flux=ocean["ice_salt_flux"], x_stress=ocean["ice_x_stress"], y_stress=ocean["ice_y_stress"])
    return dict(nx=nx, ny=ny, hx=hx, hy=hy, ocean=ocean, src=src, weights=w, ice=ice)


def window(a, hx, hy, nx, ny, ring):
    return a[hy - ring:hy + ny + ring, hx - ring:hx + nx + ring]


def rel_err(got, ref, scale):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), scale)))
