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


def jra55_land_snapshots(n_levels=2, nsx=JRA55_NX, nsy=JRA55_NY, seed=SEED):
    """Synthetic JRA55PrescribedLand window (jra55_data_staging.jl:8: friver, licalvf), float32 [n, nsy, nsx], kg m⁻² s⁻¹:
    river discharge concentrated in a few per cent of the source cells, calving only poleward of 60°."""
    i = np.arange(nsx)[None, :]
    j = np.arange(nsy)[:, None]
    phi = JRA55_LAT0 + j * (2 * 89.57 / (nsy - 1)) + 0 * i
    out = {k: np.zeros((n_levels, nsy, nsx), np.float32) for k in ("friver", "licalvf")}
    mouth = uniform("land", i, j, 3, seed) < 0.03
    for n in range(n_levels):
        out["friver"][n] = np.where(mouth, 2e-4 * (1.0 + 0.3 * normal("rain", i, j, 10 + n, seed)) ** 2, 0.0)
        out["licalvf"][n] = np.where(np.abs(phi) > 60.0, 2e-5 * uniform("snow", i, j, 10 + n, seed), 0.0)
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


# ---------------------------------------------------------------------------------------------
# a tripolar grid with a REAL fold (BASELINE configs 4–5: TripolarGrid(size = (360, 180, Nz)),
# OceanConfigurations/one_degree_tripolar.jl:48-51; size = (2160, 1080, Nz), sixth_degree_tripolar.jl:33-36)
# ---------------------------------------------------------------------------------------------
def tripolar_mesh(nx, ny, *, southernmost_latitude=-80.0, cap_latitude=55.0, cap_rows=None, pole_longitude=75.0):
    """Cell-centre longitude / latitude and the rotation of the grid's i-axis against geographic east for a synthetic
    tripolar grid, arrays (ny, nx) over the GLOBAL interior.

    South of `cap_latitude` the mesh is latitude–longitude.  The cap is a bipolar mesh in the stereographic plane of the
    north pole: confocal ellipses (rows) and hyperbolas (columns) about two grid poles on land at ±a on the axis
    λ = pole_longitude, x = a cosh μ cos ν, y = a sinh μ sin ν.  The LAST row of cell centres lies on the segment
    between the poles (μ = 0), where ν and −ν are the same point: column i and column nx − 1 − i coincide there — the
    tracer-point pivot of Oceananigans' tripolar fold, which is what cf_fold_north_halo implements.  (The outermost
    ellipse of the cap is circular to 1.3 %; the mesh is synthetic, its topology is the real one.)

    The rotation follows experiments/OMIPSimulations/scripts/visualize/cache.jl:406-427: from the two x-face nodes
    bracketing the cell, dE = cos φ · Δλ, dN = Δφ, (cos θ, sin θ) = (dE, dN)/‖·‖."""
    nyc = cap_rows if cap_rows is not None else max(2, int(round(ny * (90.0 - cap_latitude) / (90.0 - southernmost_latitude))))
    nys = ny - nyc
    mu0 = 2.5
    r0 = 2.0 * np.tan(np.deg2rad(90.0 - cap_latitude) / 2.0)       # stereographic radius of the cap's rim
    a = r0 / np.cosh(mu0)

    def position(fi, fj):
        """(λ, φ) in degrees at fractional global indices: centres at integer + 0.5 in i, rows at integers in j."""
        fi, fj = np.broadcast_arrays(np.asarray(fi, float), np.asarray(fj, float))
        lam = (fi * (360.0 / nx) + pole_longitude) % 360.0     # column 0 starts on the meridian of the grid poles
        phi = southernmost_latitude + (fj + 0.5) * (cap_latitude - southernmost_latitude) / nys
        cap = fj > nys - 0.5
        mu = np.clip((ny - 1 - fj) / (nyc - 0.5) * mu0, 0.0, None)      # 0 on the last row of centres, μ0 at the rim
        nu = np.deg2rad(fi * (360.0 / nx))                      # ν ↦ −ν is i ↦ nx − 1 − i for the centres at i + ½
        # (the fold segment is displaced off the geographic pole, tapering to zero at the rim: no cell straddles λ's singularity)
        x, y = a * np.cosh(mu) * np.cos(nu), a * np.sinh(mu) * np.sin(nu) + 0.04 * r0 * (1.0 - mu / mu0)
        lam_c = (np.rad2deg(np.arctan2(y, x)) + pole_longitude) % 360.0
        phi_c = 90.0 - 2.0 * np.rad2deg(np.arctan(np.hypot(x, y) / 2.0))
        return np.where(cap, lam_c, lam), np.where(cap, phi_c, phi)

    i = np.arange(nx)[None, :] + 0.5
    j = np.arange(ny)[:, None] + 0.0
    lam, phi = position(i, j)
    lw, pw = position(i - 0.5, j)
    le, pe = position(i + 0.5, j)
    dlam = (le - lw + 180.0) % 360.0 - 180.0
    dE, dN = np.cos(np.deg2rad(phi)) * dlam, pe - pw
    r = np.hypot(dE, dN)
    cos_t = np.where(r > 0, dE / np.where(r > 0, r, 1.0), 1.0)
    sin_t = np.where(r > 0, dN / np.where(r > 0, r, 1.0), 0.0)
    return lam, phi, cos_t, sin_t


def fold_north(a, nx, ny, hx, hy, rows, location="center", sign=1.0):
    """NumPy statement of the tripolar fold (include/coflux.h: cf_fold_north_halo) on a halo-inclusive GLOBAL array,
    in place: north halo rows from the mirrored interior rows, periodic x-halos of those rows included."""
    i = np.arange(-hx, nx + hx) % nx
    for r in range(1, rows + 1):
        if location == "x_face":
            src_i, src_j, sg = (nx - i) % nx, ny - 1 - r, np.where(i == 0, abs(sign), sign)
        elif location == "y_face":
            src_i, src_j, sg = nx - 1 - i, ny - r, sign
        else:
            src_i, src_j, sg = nx - 1 - i, ny - 1 - r, sign
        a[hy + ny - 1 + r, :] = sg * a[hy + src_j, hx + src_i]
    return a


FOLD_LOCATION = dict(T="center", S="center", u="x_face", v="y_face")
FOLD_SIGN = dict(T=1.0, S=1.0, u=-1.0, v=-1.0)


def tripolar_case(nx, ny, hx, hy, *, j0=0, j1=None, rows=2, nsx=JRA55_NX, nsy=JRA55_NY, seed=SEED):
    """Ocean state + general (2-D) interpolation weights + rotation on the synthetic tripolar grid for the latitude slab
    [j0, j1) of the global nx × ny grid (default: the whole grid).  Halos: periodic in x; the LAST slab's north halo is
    the fold of its own rows (only `rows` rows are meaningful, like after an exchange); inner slab seams carry the
    neighbours' rows because every field is a function of the global index."""
    j1 = ny if j1 is None else j1
    lam, phi, cos_t, sin_t = tripolar_mesh(nx, ny)
    full = ocean_state(nx, ny, hx, hy, latitude=(-80.0, 90.0), seed=seed)          # global, halo-inclusive, index-space fields
    out = {}
    for k in ("T", "S", "u", "v"):
        g = full[k].copy()
        fold_north(g, nx, ny, hx, hy, rows, FOLD_LOCATION[k], FOLD_SIGN[k])
        out[k] = np.ascontiguousarray(g[j0:j1 + 2 * hy])
    # a wet mask that is symmetric under the fold on the last row (a cell and its mirror image are the same water)
    m = full["mask"].copy()
    last = m[hy + ny - 1, hx:hx + nx]
    m[hy + ny - 1, hx:hx + nx] = np.minimum(last, last[::-1])
    for pole in (0, nx // 2):                  # the two grid poles sit on land, as on the real grid (mirror-symmetric patches)
        cols = np.arange(pole - 3, pole + 3) % nx
        m[hy + ny - 4:hy + ny, hx + cols] = 0
    m[hy + ny - 1, :hx], m[hy + ny - 1, hx + nx:] = m[hy + ny - 1, nx:nx + hx], m[hy + ny - 1, hx:2 * hx]
    fold_north(m, nx, ny, hx, hy, rows, "center", 1)
    out["mask"] = np.ascontiguousarray(m[j0:j1 + 2 * hy])

    def halo2d(a):          # (ny, nx) interior → halo-inclusive slab, periodic in x, edge rows repeated / folded in y
        g = np.empty((ny + 2 * hy, nx + 2 * hx))
        g[hy:hy + ny, hx:hx + nx] = a
        g[:hy] = g[hy:hy + 1]
        g[hy + ny:] = g[hy + ny - 1:hy + ny]
        g[:, :hx], g[:, hx + nx:] = g[:, nx:nx + hx], g[:, hx:2 * hx]
        return g
    LAM, PHI, COS, SIN = (halo2d(a) for a in (lam, phi, cos_t, sin_t))
    for a, loc, sg in ((LAM, "center", 1), (PHI, "center", 1), (COS, "center", 1), (SIN, "center", 1)):
        fold_north(a, nx, ny, hx, hy, min(rows, hy), loc, sg)
    # across the fold the grid's i-axis points the other way: the halo image's own rotation is the source cell's, reversed
    COS[hy + ny:hy + ny + rows] *= -1.0
    SIN[hy + ny:hy + ny + rows] *= -1.0
    sl = slice(j0, j1 + 2 * hy)
    fi = LAM[sl] / (360.0 / nsx)
    fj = (PHI[sl] - JRA55_LAT0) / (2 * 89.57 / (nsy - 1))
    weights = dict(separable=False, fi=np.ascontiguousarray(fi), fj=np.ascontiguousarray(fj), cos_rot=np.ascontiguousarray(COS[sl]),
                   sin_rot=np.ascontiguousarray(SIN[sl]), latitude=np.ascontiguousarray(PHI[sl]))
    return dict(nx=nx, ny=j1 - j0, hx=hx, hy=hy, ocean=out, weights=weights, src=jra55_snapshots(2, nsx, nsy, seed))
