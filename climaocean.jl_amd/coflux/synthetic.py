"""Deterministic synthetic inputs for the flux path.

Counter-based: value(field, i, j) = f(splitmix64(seed ⊕ field·2⁴⁰ ⊕ j·2²⁰ ⊕ i)), so any rank /
language / tile reproduces the same global field bit for bit from global indices alone.  This is
data generation shared by tests and bench.py; it is neither oracle nor product compute.
"""
import numpy as np

SEED = 20260612

_FIELD_IDS = {name: k + 1 for k, name in enumerate(
    ["To", "So", "uo", "vo", "land", "ice", "Ta", "pa", "qa", "ua", "va", "Qs", "Ql", "rain",
     "snow", "rot", "Qio", "Jsio", "txio", "tyio", "hi", "Tsi", "ui", "vi", "ai"])}


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x.copy()
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _counter(field, ii, jj, stream, seed):
    with np.errstate(over="ignore"):
        key = (np.uint64(seed) ^ (np.uint64(_FIELD_IDS[field]) << np.uint64(40))
               ^ (np.uint64(stream) << np.uint64(56))
               ^ ((jj.astype(np.int64) & 0xFFFFF).astype(np.uint64) << np.uint64(20))
               ^ (ii.astype(np.int64) & 0xFFFFF).astype(np.uint64))
        return splitmix64(key)


def uniform(field, ii, jj, stream=0, seed=SEED):
    """U(0,1) on the broadcast of global index arrays ii, jj."""
    with np.errstate(over="ignore"):
        h = _counter(field, *np.broadcast_arrays(ii, jj), stream, seed)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(field, ii, jj, stream=0, seed=SEED):
    """N(0,1) by Box–Muller from two counter streams."""
    u1 = uniform(field, ii, jj, 2 * stream, seed)
    u2 = uniform(field, ii, jj, 2 * stream + 1, seed)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ---------------------------------------------------------------------------------------------
# grids
# ---------------------------------------------------------------------------------------------
JRA55_NX, JRA55_NY = 640, 320          # launch.sh:86-87
JRA55_DLON = 360.0 / JRA55_NX
JRA55_LAT0 = -89.57                     # TL319 Gaussian grid is regular to ~1e-3°; we use regular
JRA55_DLAT = 2 * 89.57 / (JRA55_NY - 1)


def zonal_sst(phi_deg):
    return -1.8 + 30.0 * np.cos(np.deg2rad(phi_deg)) ** 2


def _qsat_tetens(T, p):
    es = 611.2 * np.exp(17.67 * (T - 273.15) / (T - 29.65))
    return 0.622 * es / (p - 0.378 * es)


def ocean_indices(nx, ny, hx, hy, j_offset=0, nx_global=None):
    """Global (i, j) index arrays for a halo-inclusive local slab; x is periodic."""
    nxg = nx_global or nx
    i = (np.arange(-hx, nx + hx) % nxg)[None, :]
    j = (np.arange(-hy, ny + hy) + j_offset)[:, None]
    return i, j


def ocean_latlon(nx_global, ny_global, i, j, latitude=(-70.0, 70.0), longitude=(0.0, 360.0)):
    dlam = (longitude[1] - longitude[0]) / nx_global
    dphi = (latitude[1] - latitude[0]) / ny_global
    lam = longitude[0] + (i + 0.5) * dlam
    phi = latitude[0] + (j + 0.5) * dphi
    return lam, phi


def ocean_state(nx, ny, hx, hy, *, ny_global=None, j_offset=0, latitude=(-70.0, 70.0),
                land_fraction=True, seed=SEED):
    """Synthetic ocean surface (k = Nz) state with halos: dict of (ny+2hy, nx+2hx) f64 arrays
    T [°C], S, u (x-faces), v (y-faces), mask (uint8, 1 = wet), ice concentration + ice–ocean fluxes."""
    nyg = ny_global or ny
    i, j = ocean_indices(nx, ny, hx, hy, j_offset)
    lam, phi = ocean_latlon(nx, nyg, i, j, latitude)
    lam, phi = np.broadcast_arrays(lam, phi)
    T = np.maximum(-1.8, zonal_sst(phi) + 0.5 * normal("To", i, j, seed=seed))
    S = np.clip(35.0 + normal("So", i, j, seed=seed), 30.0, 40.0)
    u = 0.1 * normal("uo", i, j, seed=seed)
    v = 0.1 * normal("vo", i, j, seed=seed)
    if land_fraction:
        lr, pr = np.deg2rad(lam), np.deg2rad(phi)
        blob = (np.sin(3 * lr + 1.3) * np.cos(4 * pr + 0.4) + 0.6 * np.sin(5 * lr - 2 * pr + 2.1)
                + 0.4 * np.cos(7 * lr + 3 * pr))
        speck = uniform("land", i, j, seed=seed) < 0.02
        mask = ((blob < 0.45) & ~speck).astype(np.uint8)
    else:
        mask = np.ones(T.shape, np.uint8)
    ice = np.clip((np.abs(phi) - 55.0) / 10.0 + 0.1 * normal("ice", i, j, seed=seed), 0.0, 1.0)
    out = dict(T=T, S=S, u=u, v=v, mask=mask, ice_concentration=ice,
               ice_interface_heat=ice * (5.0 + 2.0 * normal("Qio", i, j, seed=seed)),
               ice_salt_flux=ice * 1e-6 * normal("Jsio", i, j, seed=seed),
               ice_x_stress=ice * 1e-5 * normal("txio", i, j, seed=seed),
               ice_y_stress=ice * 1e-5 * normal("tyio", i, j, seed=seed),
               longitude=lam, latitude=phi)
    return {k: np.ascontiguousarray(a) for k, a in out.items()}


def evolved_ocean_state(state, nx, ny, hx, hy, step, *, ny_global=None, j_offset=0, seed=SEED):
    """The surface state `step` coupled steps (Δt = 20 min) after `state`: T, S, u, v drift by what an ocean
    surface changes in twenty minutes (≈ 0.02 K, 0.002 g/kg, 5 mm/s per step, white in space — the synthetic
    fields have no dynamics to evolve them with).  Same mask; a function of global indices like everything here."""
    i, j = ocean_indices(nx, ny, hx, hy, j_offset)
    out = dict(state)
    out["T"] = np.maximum(-1.8, state["T"] + 0.02 * step * normal("To", i, j, 7, seed))
    out["S"] = np.clip(state["S"] + 0.002 * step * normal("So", i, j, 7, seed), 30.0, 40.0)
    out["u"] = state["u"] + 0.005 * step * normal("uo", i, j, 7, seed)
    out["v"] = state["v"] + 0.005 * step * normal("vo", i, j, 7, seed)
    return {k: np.ascontiguousarray(a) for k, a in out.items()}


def sea_ice_state(nx, ny, hx, hy, *, ny_global=None, j_offset=0, latitude=(-70.0, 70.0), seed=SEED):
    """Synthetic ClimaSeaIce surface state on the ocean grid: thickness [m] (including ice thinner than the
    consolidation thickness), previous top temperature [°C], ice drift, per-cell albedo."""
    nyg = ny_global or ny
    i, j = ocean_indices(nx, ny, hx, hy, j_offset)
    lam, phi = ocean_latlon(nx, nyg, i, j, latitude)
    lam, phi = np.broadcast_arrays(lam, phi)
    h = np.clip(0.02 + 1.5 * uniform("hi", i, j, seed=seed) ** 2 + 0.0 * phi, 0.0, 4.0)
    Ts = np.minimum(0.0, -8.0 + 6.0 * normal("Tsi", i, j, seed=seed) + 0.0 * phi)
    out = dict(thickness=h, top_temperature=Ts, u=0.05 * normal("ui", i, j, seed=seed) + 0.0 * phi,
               v=0.05 * normal("vi", i, j, seed=seed) + 0.0 * phi,
               albedo=np.clip(0.65 + 0.1 * normal("ai", i, j, seed=seed) + 0.0 * phi, 0.3, 0.9))
    return {k: np.ascontiguousarray(a) for k, a in out.items()}


def jra55_snapshots(n_levels=2, nsx=JRA55_NX, nsy=JRA55_NY, seed=SEED, temporal_correlation=None):
    """Synthetic JRA55 window: dict var -> float32 [n_levels, nsy, nsx] (jra55_data_staging.jl:8).
    Default: the snapshots are independent draws (every test and golden vector uses this).  With
    `temporal_correlation` = ρ the noise of consecutive 3-hourly snapshots is correlated ρ (a rotation by
    arccos ρ per snapshot in the plane of two independent fields, so every snapshot keeps the same marginal
    distribution): synoptic fields decorrelate over days, not hours, and bench.py steps the clock through them."""
    i = np.arange(nsx)[None, :]
    j = np.arange(nsy)[:, None]
    phi = JRA55_LAT0 + j * (2 * 89.57 / (nsy - 1)) + 0 * i
    out = {k: np.empty((n_levels, nsy, nsx), np.float32)
           for k in ("tas", "huss", "psl", "uas", "vas", "rlds", "rsds", "prra", "prsn")}
    if temporal_correlation is None:
        noise = normal
    else:
        theta = np.arccos(temporal_correlation)

        def noise(field, ii, jj, n, sd):
            return np.cos(n * theta) * normal(field, ii, jj, 100, sd) + np.sin(n * theta) * normal(field, ii, jj, 101, sd)
    for n in range(n_levels):
        Ta = 273.15 + np.maximum(-1.8, zonal_sst(phi)) - 1.0 + 2.0 * noise("Ta", i, j, n, seed)
        pa = 101325.0 + 800.0 * noise("pa", i, j, n, seed)
        qa = (0.8 + 0.05 * noise("qa", i, j, n, seed)) * _qsat_tetens(Ta, pa)
        out["tas"][n] = Ta
        out["psl"][n] = pa
        out["huss"][n] = np.maximum(qa, 1e-5)
        out["uas"][n] = 7.0 * noise("ua", i, j, n, seed)
        out["vas"][n] = 7.0 * noise("va", i, j, n, seed)
        out["rsds"][n] = np.maximum(0.0, 300.0 * np.cos(np.deg2rad(phi)) + 50.0 * noise("Qs", i, j, n, seed))
        out["rlds"][n] = 350.0 + 30.0 * noise("Ql", i, j, n, seed)
        out["prra"][n] = np.maximum(0.0, 3e-5 * (1.0 + noise("rain", i, j, n, seed)))
        out["prsn"][n] = np.where(np.abs(phi) > 60.0,
                                  np.maximum(0.0, 1e-5 * (1.0 + noise("snow", i, j, n, seed))), 0.0)
    return out


def latlon_fractional_indices(nx, ny, hx, hy, *, ny_global=None, j_offset=0, latitude=(-70.0, 70.0),
                              nsx=JRA55_NX, nsy=JRA55_NY):
    """Separable fractional source indices (0-based) of a lat-lon ocean grid into the JRA55 grid,
    halo-inclusive: fi[nx+2hx], fj[ny+2hy], plus latitude per row."""
    nyg = ny_global or ny
    i = np.arange(-hx, nx + hx)
    j = np.arange(-hy, ny + hy) + j_offset
    lam = ((i + 0.5) * (360.0 / nx)) % 360.0
    phi = latitude[0] + (j + 0.5) * (latitude[1] - latitude[0]) / nyg
    fi = lam / (360.0 / nsx)
    fj = (phi - JRA55_LAT0) / (2 * 89.57 / (nsy - 1))
    return np.ascontiguousarray(fi), np.ascontiguousarray(fj), np.ascontiguousarray(phi)


def tripolar_like_weights(nx, ny, hx, hy, *, ny_global=None, j_offset=0, nsx=JRA55_NX, nsy=JRA55_NY,
                          seed=SEED):
    """General (2-D) fractional indices + rotation for an index-space 'tripolar-like' grid: the
    southern part is lat-lon, north of 60° the mesh is sheared and rotated smoothly towards two
    poles (synthetic; the real TripolarGrid mesh is built by Oceananigans, out of scope)."""
    nyg = ny_global or ny
    i, j = ocean_indices(nx, ny, hx, hy, j_offset)
    ii = np.arange(-hx, nx + hx)[None, :]
    lam = ((ii + 0.5) * (360.0 / nx)) + 0.0 * j
    phi = -80.0 + (j + 0.5) * (170.0 / nyg) + 0.0 * ii
    t = np.clip((phi - 60.0) / 30.0, 0.0, 1.0)
    theta = 0.9 * t * np.sin(2 * np.deg2rad(lam))              # rotation angle of the grid i-axis
    lam2 = (lam + 20.0 * t * np.sin(np.deg2rad(lam))) % 360.0
    phi2 = phi - 8.0 * t * t * np.cos(2 * np.deg2rad(lam)) ** 2
    fi = lam2 / (360.0 / nsx)
    fj = (phi2 - JRA55_LAT0) / (2 * 89.57 / (nsy - 1))
    return (np.ascontiguousarray(fi), np.ascontiguousarray(fj), np.ascontiguousarray(np.cos(theta)),
            np.ascontiguousarray(np.sin(theta)), np.ascontiguousarray(phi2))
