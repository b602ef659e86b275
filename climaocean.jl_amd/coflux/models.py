"""Host-side mirror of the user-facing API around the flux path — same names, argument meaning
and call order as the reference:

    grid       = LatitudeLongitudeGrid(size=(1440, 560, 10), halo=(7, 7, 7), longitude=(0, 360),
                                       latitude=(-70, 70), z=(-3000, 0))          # README.md:56-61
    ocean      = ocean_simulation(grid)                                            # README.md:67
    atmosphere = JRA55PrescribedAtmosphere()                                       # README.md:74
    coupled    = OceanSeaIceModel(ocean; atmosphere)                               # README.md:75
    simulation = Simulation(coupled, Δt=20minutes, stop_time=30days); run!(sim)    # README.md:76-77

Only the flux path is implemented: `time_step!(coupled)` advances the clock, lets
the ocean component step (the hydrostatic dynamical core is OUT OF SCOPE — `ocean_simulation`
returns a prescribed-state ocean whose step is a no-op unless a callback is given) and then runs
`update_state!`, which is the accelerated part: interpolate_atmosphere_state! →
compute_atmosphere_ocean_fluxes! → compute_net_ocean_fluxes!, all inside libcoflux on the GPU.
The net fluxes land in the ocean's top-boundary-condition fields exactly where the reference puts
them (model.interfaces.net_fluxes.ocean.{u,v,T,S}, omip_diagnostics.jl:77-80).

The atmosphere holds a window of snapshots in HBM filled from a provider: synthetic in tests/bench, or the raw
Float32 plane files of coflux/jra55.py (`omip_forcing`; NetCDF decoding itself is not possible in this image).
"""
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import abi, synthetic
from . import interface_computations as ic
from .runtime import EXCHANGE_NAMES, FLUX_NAMES, FLUX_OPTIONAL, NET_NAMES, FluxContext

minutes, hours, days = 60.0, 3600.0, 86400.0


# ---------------------------------------------------------------------------------------------
# grid
# ---------------------------------------------------------------------------------------------
@dataclass
class LatitudeLongitudeGrid:
    """README.md:56-61 (only the metadata the surface path needs)."""
    size: tuple = (1440, 560, 10)
    halo: tuple = (7, 7, 7)
    longitude: tuple = (0.0, 360.0)
    latitude: tuple = (-70.0, 70.0)
    z: tuple = (-3000.0, 0.0)
    device: int = 0

    @property
    def surface_shape(self):
        (nx, ny, _), (hx, hy, _) = self.size, self.halo
        return (ny + 2 * hy, nx + 2 * hx)

    @property
    def surface_z(self):
        nz = self.size[2]
        dz = (self.z[1] - self.z[0]) / nz
        return self.z[1] - 0.5 * dz  # centre of the top cell (uniform spacing)

    def fractional_indices(self, nsx=synthetic.JRA55_NX, nsy=synthetic.JRA55_NY):
        nx, ny, _ = self.size
        hx, hy, _ = self.halo
        return synthetic.latlon_fractional_indices(nx, ny, hx, hy, latitude=self.latitude, nsx=nsx, nsy=nsy)

    fold_north = False

    def interpolation_weights(self, to_device, nsx=synthetic.JRA55_NX, nsy=synthetic.JRA55_NY):
        """cf_interp_weights for this grid: separable fractional indices into the JRA55 grid (no rotation)."""
        fi, fj, phi = self.fractional_indices(nsx, nsy)
        return dict(separable=True, fi=to_device(fi), fj=to_device(fj), latitude=to_device(phi))


@dataclass
class TripolarGrid:
    """TripolarGrid(arch; size = (360, 180, Nz), halo = (5, 5, 4), z, …) — OceanConfigurations/one_degree_tripolar.jl:32,48-51
    (1°), half_degree_tripolar.jl:48-51 (720×360), sixth_degree_tripolar.jl:33-36 (2160×1080, halo 7), tenth_degree… (3600×1800).
    Only what the surface path needs: cell-centre positions → general (2-D) fractional indices into the JRA55 grid, the
    rotation of the grid's i-axis against geographic east (visualize/cache.jl:406-427: the winds are rotated into the
    grid frame) and the north fold (the last row of tracer points is its own mirror image; Oceananigans' zipper
    boundary).  The mesh itself comes from `synthetic.tripolar_mesh` (a bipolar cap with the real fold topology) unless
    `longitude` / `latitude` / rotation arrays of the real grid are handed in."""
    size: tuple = (360, 180, 10)
    halo: tuple = (5, 5, 4)
    z: tuple = (-3000.0, 0.0)
    southernmost_latitude: float = -80.0
    first_pole_longitude: float = 75.0
    north_poles_latitude: float = 55.0
    device: int = 0
    longitude: Optional[np.ndarray] = None     # (ny, nx) cell-centre λ [deg] of a real grid (else synthetic)
    latitude: Optional[np.ndarray] = None      # (ny, nx) cell-centre φ [deg]
    cos_rotation: Optional[np.ndarray] = None
    sin_rotation: Optional[np.ndarray] = None

    fold_north = True

    @property
    def surface_shape(self):
        (nx, ny, _), (hx, hy, _) = self.size, self.halo
        return (ny + 2 * hy, nx + 2 * hx)

    @property
    def surface_z(self):
        nz = self.size[2]
        dz = (self.z[1] - self.z[0]) / nz
        return self.z[1] - 0.5 * dz

    def mesh(self):
        nx, ny, _ = self.size
        if self.longitude is not None:
            if self.latitude is None or self.cos_rotation is None or self.sin_rotation is None:
                raise ValueError("TripolarGrid: longitude, latitude, cos_rotation and sin_rotation come together")
            return self.longitude, self.latitude, self.cos_rotation, self.sin_rotation
        return synthetic.tripolar_mesh(nx, ny, southernmost_latitude=self.southernmost_latitude,
                                       cap_latitude=self.north_poles_latitude, pole_longitude=self.first_pole_longitude)

    def interpolation_weights(self, to_device, nsx=synthetic.JRA55_NX, nsy=synthetic.JRA55_NY, rows=2):
        """General weights + rotation, halo-inclusive: periodic in x, the north halo rows are the fold images of the
        interior rows (their i-axis points the other way: rotation reversed), as synthetic.tripolar_case builds them."""
        (nx, ny, _), (hx, hy, _) = self.size, self.halo
        lam, phi, cos_t, sin_t = self.mesh()

        def halo2d(a):
            g = np.empty((ny + 2 * hy, nx + 2 * hx))
            g[hy:hy + ny, hx:hx + nx] = a
            g[:hy] = g[hy:hy + 1]
            g[hy + ny:] = g[hy + ny - 1:hy + ny]
            g[:, :hx], g[:, hx + nx:] = g[:, nx:nx + hx], g[:, hx:2 * hx]
            return g
        LAM, PHI, COS, SIN = (halo2d(a) for a in (lam, phi, cos_t, sin_t))
        r = min(rows, hy)
        for a in (LAM, PHI, COS, SIN):
            synthetic.fold_north(a, nx, ny, hx, hy, r, "center", 1)
        COS[hy + ny:hy + ny + r] *= -1.0
        SIN[hy + ny:hy + ny + r] *= -1.0
        fi = LAM / (360.0 / nsx)
        fj = (PHI - synthetic.JRA55_LAT0) / (2 * 89.57 / (nsy - 1))
        c = np.ascontiguousarray
        return dict(separable=False, fi=to_device(c(fi)), fj=to_device(c(fj)), cos_rot=to_device(c(COS)), sin_rot=to_device(c(SIN)),
                    latitude=to_device(c(PHI)))


# ---------------------------------------------------------------------------------------------
# ocean (boundary only: surface state in, top-BC flux fields out)
# ---------------------------------------------------------------------------------------------
@dataclass
class SurfaceFluxRestoring:
    """SurfaceFluxRestoring(DatasetRestoring(…; rate = piston_velocity / (Δz days))) — salinity_surface_restoring,
    omip_simulation.jl:507-523: a surface-only restoring toward `target` (2-D, halo-inclusive device array, g/kg) that rides
    on the ocean's top-flux boundary condition through `additional_surface_fluxes`.  piston_velocity in m/day as there."""
    target: torch.Tensor
    piston_velocity: float = 1.0 / 6.0

    @property
    def velocity(self):
        return self.piston_velocity / days


@dataclass
class MultipleFluxes:
    """MultipleFluxes{flux_field, additional_fluxes} (omip_simulation.jl:175-206): the top boundary condition of a tracer
    when `ocean_simulation(grid; additional_surface_fluxes = (; S = …))` is used.  The coupled model writes the bulk flux
    into `flux_field`; the ocean applies flux_field + additional_fluxes."""
    flux_field: torch.Tensor
    additional_fluxes: object


class OceanSimulation:
    """What `ocean_simulation(grid)` returns: `.model.tracers.{T,S}`, `.model.velocities.{u,v}` as
    (Nz+2Hz, Ny+2Hy, Nx+2Hx) device arrays (k slowest), `.model.clock`, and the top boundary
    condition fields the coupled model writes (omip_simulation.jl:175-206: a bare 2-D field, or MultipleFluxes when
    `additional_surface_fluxes` is given)."""

    def __init__(self, grid, step_callback=None, additional_surface_fluxes=None):
        dev = torch.device("cuda", grid.device)
        nz, hz = grid.size[2], grid.halo[2]
        shape3 = (nz + 2 * hz,) + grid.surface_shape
        z3 = lambda: torch.zeros(shape3, dtype=torch.float64, device=dev)  # noqa: E731
        z2 = lambda: torch.zeros(grid.surface_shape, dtype=torch.float64, device=dev)  # noqa: E731
        self.grid = grid
        self.k_top = hz + nz - 1
        self.model = SimpleNamespace(
            grid=grid,
            tracers=SimpleNamespace(T=z3(), S=z3()),
            velocities=SimpleNamespace(u=z3(), v=z3()),
            clock=SimpleNamespace(time=0.0, iteration=0),
            wet_mask=torch.ones(grid.surface_shape, dtype=torch.uint8, device=dev),
            top_boundary_conditions=SimpleNamespace(u=z2(), v=z2(), T=z2(), S=z2()),
            shortwave_surface_flux=z2())  # radiation.surface_flux, KPP/kpp_surface_forcing.jl:47-51
        self.step_callback = step_callback
        for name, extra in (additional_surface_fluxes or {}).items():
            bc = self.model.top_boundary_conditions
            setattr(bc, name, MultipleFluxes(getattr(bc, name), extra))

    def surface_state(self):
        m, k = self.model, self.k_top
        return dict(T=m.tracers.T[k], S=m.tracers.S[k], u=m.velocities.u[k], v=m.velocities.v[k], mask=m.wet_mask)

    def time_step(self, dt):
        if self.step_callback is not None:
            self.step_callback(self, dt)
        self.model.clock.time += dt
        self.model.clock.iteration += 1


def ocean_simulation(grid, **kw):
    """README.md:67; OceanConfigurations/latitude_longitude.jl:50-55."""
    return OceanSimulation(grid, **kw)


def set_surface(ocean, *, T=None, S=None, u=None, v=None, mask=None):
    """set!(ocean.model, T=…, S=…) for the top level (README.md:69-71), from host arrays with halos."""
    m, k = ocean.model, ocean.k_top
    for tgt, src in ((m.tracers.T, T), (m.tracers.S, S), (m.velocities.u, u), (m.velocities.v, v)):
        if src is not None:
            tgt[k].copy_(torch.as_tensor(np.ascontiguousarray(src)))
    if mask is not None:
        m.wet_mask.copy_(torch.as_tensor(np.ascontiguousarray(mask)))


# ---------------------------------------------------------------------------------------------
# atmosphere / radiation
# ---------------------------------------------------------------------------------------------
class JRA55PrescribedAtmosphere:
    """JRA55PrescribedAtmosphere(arch; dir, dataset, start_date, end_date, time_indices_in_memory,
    prefetch) — atmosphere.jl:20-29, README.md:74.  Holds `time_indices_in_memory` 3-hourly
    snapshots of the 9 variables (jra55_data_staging.jl:8) as Float32 640×320 fields in HBM.

    Two backends, as in the reference: everything in memory (`snapshots` = dict var -> [n, 320, 640]), or a
    sliding window (`provider(n) -> dict var -> [320, 640]` reads snapshot n from wherever the files are;
    `total_snapshots` per repeat-year/multi-year record).  With `prefetch` the snapshots the clock will reach next
    are read by a background thread straight into the window's pinned staging buffers and committed to HBM by the
    stepping thread, so file reads and PCIe copies overlap the flux kernels (launch.sh:86-93)."""

    def __init__(self, snapshots=None, *, provider=None, total_snapshots=None, time_indices_in_memory=2,
                 prefetch=True, time_interval=3 * hours, device=0, reference_height=10.0, boundary_layer_height=600.0,
                 cyclic=True, source_size=(synthetic.JRA55_NX, synthetic.JRA55_NY)):
        self.time_interval = time_interval
        self.reference_height = reference_height
        self.boundary_layer_height = boundary_layer_height
        self.cyclic = cyclic  # RepeatYearJRA55-style cyclic time indexing vs clamped (MultiYearJRA55 record ends)
        self.provider, self.prefetch = provider, prefetch
        self.window = None
        self._pending = {}   # time index -> Future of a background read into the pinned buffers
        self._reader = None
        if provider is not None:
            if total_snapshots is None:
                raise ValueError("a snapshot provider needs total_snapshots (2920 for a repeat year of 3-hourly JRA55)")
            if time_indices_in_memory < 2:
                raise ValueError("time_indices_in_memory must be >= 2")
            self.n_levels = total_snapshots
            self.n_slots = time_indices_in_memory
            self.source_size = source_size
            self.data = None
            return
        dev = torch.device("cuda", device)
        if snapshots is None:
            snapshots = synthetic.jra55_snapshots(time_indices_in_memory)
        self.data = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in snapshots.items()}
        self.n_levels = next(iter(self.data.values())).shape[0]

    def time_indices(self, t):
        """(n₁, n₂, ñ): the bracketing snapshots and the fractional position between them."""
        x = t / self.time_interval
        n = int(np.floor(x))
        frac = x - n
        if self.cyclic:
            return n % self.n_levels, (n + 1) % self.n_levels, frac
        n1 = min(max(n, 0), self.n_levels - 1)
        n2 = min(max(n + 1, 0), self.n_levels - 1)
        return n1, n2, (frac if 0 <= n < self.n_levels - 1 else 0.0)

    # ---- sliding-window backend ---------------------------------------------------------------
    def _wrap(self, n):
        return n % self.n_levels if self.cyclic else min(max(n, 0), self.n_levels - 1)

    def _read_into_staging(self, k):
        """Snapshot COUNTER k (monotone in time, not wrapped) → the provider's record index wrap(k), staged in slot
        k mod n_slots.  Slots follow the counter, not the record index: at the end of a repeat year the indices
        jump from total−1 to 0, and two consecutive snapshots must never share a slot."""
        slot = k % self.n_slots
        self.window.wait_slot(slot)          # the previous copy out of these pinned buffers has finished
        snap = self.provider(self._wrap(k))
        for v in abi.JRA55_VARIABLES:
            np.copyto(self.window.host_view(slot, v), snap[v], casting="same_kind")
        return slot

    def source(self, context, t):
        """The cf_atmos_source for time t.  In-memory backend: (data, n₁, n₂, ñ).  Window backend: makes n₁, n₂
        resident (blocking only if the prefetch has not delivered them), commits finished prefetches and queues
        the reads for the snapshots ahead."""
        n1, n2, frac = self.time_indices(t)
        if self.provider is None:
            return self.data, n1, n2, frac
        if self.window is None:
            from . import runtime
            self.window = runtime.SnapshotWindow(context, self.source_size[0], self.source_size[1], self.n_slots)
            if self.prefetch and self.n_slots > 2:
                from concurrent.futures import ThreadPoolExecutor
                self._reader = ThreadPoolExecutor(max_workers=1, thread_name_prefix="jra55-prefetch")
        w = self.window
        base = int(np.floor(t / self.time_interval))
        if self.cyclic:
            k1, k2 = base, base + 1
        else:   # clamped record: the counters stop at its ends
            k1 = min(max(base, 0), self.n_levels - 1)
            k2 = min(max(base + 1, 0), self.n_levels - 1)
        for k in (k1, k2):
            if w.find(k) >= 0:
                continue
            fut = self._pending.pop(k, None)
            slot = fut.result() if fut is not None else self._read_into_staging(k)
            w.commit(slot, k)
        for k, fut in list(self._pending.items()):      # prefetches that have landed in pinned memory
            if fut.done():
                w.commit(fut.result(), k)
                del self._pending[k]
        if self._reader is not None:
            for ahead in range(2, self.n_slots):        # the slots that hold neither k₁ nor k₂
                k = base + ahead
                if not self.cyclic and not (0 <= k < self.n_levels):
                    continue
                if k in (k1, k2) or w.find(k) >= 0 or k in self._pending:
                    continue
                if any((m % self.n_slots) == (k % self.n_slots) for m in list(self._pending) + [k1, k2]):
                    continue
                self._pending[k] = self._reader.submit(self._read_into_staging, k)
        return w.source(k1, k2, frac), n1, n2, frac

    def close(self):
        if self._reader is not None:
            self._reader.shutdown(wait=True)
            self._reader = None
        self._pending.clear()
        if self.window is not None:
            self.window.close()
            self.window = None


@dataclass
class Radiation:
    """JRA55PrescribedRadiation(arch; ocean_surface = SurfaceRadiationProperties(0.06, 1.00), …),
    atmosphere.jl:41-44 (the downwelling fields themselves ride in the atmosphere window)."""
    ocean_surface: ic.SurfaceRadiationProperties = field(default_factory=ic.SurfaceRadiationProperties)
    stefan_boltzmann_constant: float = 5.67e-8
    sea_ice_surface: Optional[ic.SurfaceRadiationProperties] = None   # SurfaceRadiationProperties(SeaIceAlbedo(hi, hs, Ts), 1.0)


JRA55PrescribedRadiation = Radiation


@dataclass
class PrescribedSeaIce:
    """The sea-ice inputs the partition reads (atmosphere.jl:34-39, src/ClimaOcean.jl:62-63); a
    prognostic ClimaSeaIce model is out of scope, its fields are taken as given."""
    concentration: torch.Tensor
    interface_heat: Optional[torch.Tensor] = None
    salt_flux: Optional[torch.Tensor] = None
    x_stress: Optional[torch.Tensor] = None
    y_stress: Optional[torch.Tensor] = None
    # sea_ice.model.{ice_thickness, ice_thermodynamics.top_surface_temperature, velocities} (atmosphere.jl:34-39):
    # with these the atmosphere–sea-ice interface is solved too and top_surface_temperature is updated in place
    thickness: Optional[torch.Tensor] = None
    top_surface_temperature: Optional[torch.Tensor] = None   # °C
    u: Optional[torch.Tensor] = None
    v: Optional[torch.Tensor] = None
    albedo: Optional[torch.Tensor] = None
    frazil_heat: Optional[torch.Tensor] = None
    snow_thickness: Optional[torch.Tensor] = None            # sea_ice.model.snow_thickness (atmosphere.jl:34)

    def fields(self):
        return {k: getattr(self, k) for k in ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")
                if getattr(self, k) is not None}

    def has_surface_state(self):
        return self.thickness is not None and self.top_surface_temperature is not None

    def surface_state(self):
        st = dict(concentration=self.concentration, thickness=self.thickness, top_temperature=self.top_surface_temperature)
        for k in ("u", "v", "albedo", "snow_thickness"):
            if getattr(self, k) is not None:
                st[k] = getattr(self, k)
        return st


# ---------------------------------------------------------------------------------------------
# ComponentInterfaces / OceanSeaIceModel
# ---------------------------------------------------------------------------------------------
class ComponentInterfaces:  # noqa: D101 — documented below
    @property
    def exchange_atmosphere_state(self):
        return self._exchange_sets[self._exchange_current]

    @property
    def _exchange_other(self):
        return self._exchange_sets[1 - self._exchange_current]

    """ComponentInterfaces(atmosphere, ocean, sea_ice; radiation, atmosphere_ocean_fluxes,
    atmosphere_ocean_velocity_difference, ocean_minimum_salinity) — omip_simulation.jl:128-158."""

    def __init__(self, atmosphere, ocean, sea_ice=None, *, radiation=None, atmosphere_ocean_fluxes=None,
                 atmosphere_sea_ice_fluxes=None, atmosphere_ocean_velocity_difference=None,
                 atmosphere_sea_ice_velocity_difference=None, sea_ice_properties=None, ocean_minimum_salinity=0.0,
                 ocean_properties=None, store_similarity_scales=False, sea_ice_ocean_heat_flux=None,
                 sea_ice_albedo=None, time_step=20 * minutes, solver_path="exact", certified_budget=8e-7,
                 ice_free_cells="iterate"):
        grid = ocean.grid
        (nx, ny, _), (hx, hy, _) = grid.size, grid.halo
        self.radiation = radiation or Radiation()
        self.atmosphere_ocean_fluxes = atmosphere_ocean_fluxes or ic.SimilarityTheoryFluxes()
        props = ocean_properties or ic.OceanProperties(surface_z=grid.surface_z)
        params = ic.flux_params(self.atmosphere_ocean_fluxes,
                                velocity_difference=atmosphere_ocean_velocity_difference,
                                ocean=props, ocean_surface=self.radiation.ocean_surface,
                                reference_height=atmosphere.reference_height,
                                boundary_layer_height=atmosphere.boundary_layer_height,
                                ocean_minimum_salinity=ocean_minimum_salinity,
                                stefan_boltzmann_constant=self.radiation.stefan_boltzmann_constant)
        self.context = FluxContext(nx, ny, hx, hy, params, ring=1, device=grid.device)
        ctx = self.context
        # implementation choices of this backend, not reference keywords (include/coflux.h): how the similarity fixed point is
        # reached — "exact" = the reference's iteration, "certified" = the reduced-iteration solve, every cell within
        # `certified_budget` of the exact path's fluxes — and what the sea-ice interface does on open water
        if solver_path not in ("exact", "certified") or ice_free_cells not in ("iterate", "zero"):
            raise ValueError(f"solver_path = {solver_path!r} (exact | certified), ice_free_cells = {ice_free_cells!r} (iterate | zero)")
        if solver_path == "certified":
            ctx.set_option(abi.OPT_CERTIFIED_BUDGET, int(round(certified_budget * 1e9)))
            ctx.set_option(abi.OPT_SOLVER_PATH, abi.SOLVER_PATH_CERTIFIED)
        if ice_free_cells == "zero":
            ctx.set_option(abi.OPT_ICE_FREE_CELLS, abi.ICE_FREE_ZERO)
        self.weights = grid.interpolation_weights(ctx.to_device)
        self.fold_north = bool(getattr(grid, "fold_north", False))
        # interfaces.exchange_atmosphere_state — the reference updates ONE field set in place.  Here it is a property over two
        # private sets: outside run!(simulation) it is always set 0 (stable tensors: what a writer or a diagnostic may hold
        # on to); inside run! every step's solver launch also interpolates the NEXT step's state, into the other set, so the
        # current state alternates between the two — read it through the property there, do not keep the dict or its tensors
        # across steps.  run! leaves the final state in set 0 and restores the context's options.
        self._exchange_sets = [ctx.field_set(EXCHANGE_NAMES), None]
        self._exchange_current = 0
        fluxes = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL if store_similarity_scales else ())
        self.atmosphere_ocean_interface = SimpleNamespace(fluxes=SimpleNamespace(**fluxes), _fields=fluxes)
        bc = ocean.model.top_boundary_conditions
        field_of = lambda b: b.flux_field if isinstance(b, MultipleFluxes) else b  # noqa: E731
        net = dict(u=field_of(bc.u), v=field_of(bc.v), T=field_of(bc.T), S=field_of(bc.S),
                   shortwave_surface_flux=ocean.model.shortwave_surface_flux,
                   upwelling_longwave=ctx.zeros(), downwelling_longwave=ctx.zeros(), downwelling_shortwave=ctx.zeros())
        self.net_fluxes = SimpleNamespace(ocean=SimpleNamespace(**net), _ocean_fields=net)
        # atmosphere–sea-ice interface (omip_simulation.jl:145,154; atmosphere.jl:34-44)
        self.atmosphere_sea_ice_interface = None
        if sea_ice is not None and sea_ice.has_surface_state():
            self.atmosphere_sea_ice_fluxes = atmosphere_sea_ice_fluxes or ic.SimilarityTheoryFluxes(
                stability_functions=ic.atmosphere_sea_ice_stability_functions())
            ice_params = ic.flux_params(self.atmosphere_sea_ice_fluxes,
                                        velocity_difference=atmosphere_sea_ice_velocity_difference, ocean=props,
                                        reference_height=atmosphere.reference_height,
                                        boundary_layer_height=atmosphere.boundary_layer_height,
                                        stefan_boltzmann_constant=self.radiation.stefan_boltzmann_constant)
            self.sea_ice_properties = sea_ice_properties or ic.SeaIceInterfaceProperties()
            ctx.set_sea_ice_formulation(ice_params, self.sea_ice_properties.to_params())
            ai = ctx.field_set(FLUX_NAMES, FLUX_OPTIONAL if store_similarity_scales else ())
            self.atmosphere_sea_ice_interface = SimpleNamespace(fluxes=SimpleNamespace(**ai), _fields=ai)
            net_ice = ctx.field_set(("top_heat", "bottom_heat"))
            self.net_fluxes.sea_ice = SimpleNamespace(**net_ice)
            self.net_fluxes._sea_ice_fields = net_ice
            # SurfaceRadiationProperties(SeaIceAlbedo(hi, hs, Ts), 1.0), atmosphere.jl:39-44
            if sea_ice_albedo is not None:
                ctx.set_sea_ice_albedo(sea_ice_albedo.to_params())
        # sea_ice_ocean_heat_flux = ThreeEquationHeatFlux(...) (omip_simulation.jl:145,154): the ice–ocean exchange and
        # frazil are then computed here each step instead of being taken from the sea-ice component
        self.sea_ice_ocean_heat_flux = None
        if sea_ice is not None and sea_ice_ocean_heat_flux is not None:
            dz = (grid.z[1] - grid.z[0]) / grid.size[2]
            self.sea_ice_ocean_heat_flux = sea_ice_ocean_heat_flux
            self.ice_ocean_params = sea_ice_ocean_heat_flux.to_params(dz, time_step)
            self.sea_ice_ocean_fluxes = ctx.field_set(("interface_heat", "salt_flux", "frazil_heat", "friction_velocity"))


class OceanSeaIceModel:
    """OceanSeaIceModel(ocean[, sea_ice]; atmosphere, radiation, interfaces) — README.md:75,
    examples/one_degree_tripolar_ocean_sea_ice.jl:42, omip_simulation.jl:132,163."""

    def __init__(self, ocean, sea_ice=None, *, atmosphere, radiation=None, interfaces=None, land=None):
        self.ocean, self.sea_ice, self.atmosphere, self.land = ocean, sea_ice, atmosphere, land
        self.interfaces = interfaces or ComponentInterfaces(atmosphere, ocean, sea_ice, radiation=radiation)
        self.clock = SimpleNamespace(time=0.0, iteration=0)
        update_state(self)


def omip_forcing(arch, sea_ice, *, forcing_dir, start_date, end_date, repeat_year_forcing=False, backend_size=30, device=0):
    """omip_forcing(arch, sea_ice; forcing_dir, start_date, end_date, repeat_year_forcing = false, backend_size = 30)
    — /root/reference/src/OMIPConfigurations/atmosphere.jl:13-49: the prescribed forcing components of an OMIP-2
    simulation.  Returns `(atmosphere, radiation, land)`: the JRA55-do atmosphere on a sliding window of `backend_size`
    snapshots with prefetch, the downwelling radiation with the OMIP-2 ocean surface (albedo 0.06, emissivity 1) and
    the CCSM3 `SeaIceAlbedo(hi, hs, Ts)` over ice, and the land freshwater (river runoff + iceberg calving).
    `arch` is accepted for signature parity (the device is `device`).  Files: raw Float32 planes, see coflux/jra55.py."""
    from . import jra55
    dataset = jra55.RepeatYearJRA55(year=start_date.year) if repeat_year_forcing else jra55.MultiYearJRA55()
    calendar = jra55.SnapshotCalendar(dataset, start_date, end_date)
    files = jra55.RawPlaneFiles(forcing_dir)
    atmosphere = JRA55PrescribedAtmosphere(provider=jra55.atmosphere_provider(forcing_dir, calendar, files), total_snapshots=calendar.total,
                                           time_indices_in_memory=backend_size, prefetch=True, cyclic=dataset.cyclic, device=device)
    atmosphere.dataset, atmosphere.calendar = dataset, calendar
    sea_ice_albedo = ic.SeaIceAlbedo() if sea_ice is not None else None   # SeaIceAlbedo(hi, hs, Ts): reads the live ice fields
    radiation = Radiation(ocean_surface=ic.SurfaceRadiationProperties(0.06, 1.00),
                          sea_ice_surface=ic.SurfaceRadiationProperties(sea_ice_albedo, 1.0) if sea_ice is not None else None)
    # JRA55PrescribedLand(arch; …) streams the whole record like the atmosphere: same calendar, same window length
    land = JRA55PrescribedLand(provider=jra55.land_provider(forcing_dir, calendar, files), total_snapshots=calendar.total,
                               time_indices_in_memory=max(2, backend_size), cyclic=dataset.cyclic, device=device)
    land.dataset, land.calendar = dataset, calendar
    return atmosphere, radiation, land


def build_coupled_model(ocean, sea_ice, atmosphere, radiation, land, flux_configuration, *, velocity_formulation="relative",
                        ocean_minimum_salinity=1, allow_shear_aware=False, shear_gustiness_coefficient=0.04):
    """build_coupled_model(ocean, sea_ice, atmosphere, radiation, land, flux_configuration; velocity_formulation = :relative,
    ocean_minimum_salinity = 1) — /root/reference/src/OMIPConfigurations/omip_simulation.jl:115-164, the same three-way
    switch and the same error strings.  Options for `flux_configuration`: "default", "corrected", "ncar" (Julia symbols
    as strings); for `velocity_formulation`: "relative", "wind".  With allow_shear_aware=True (not a reference keyword)
    "shear_aware" — the configuration launch.sh:67-72,350 describes and the reference's own switch rejects — builds the
    `:corrected` interfaces with the shear-aware gustiness (c = shear_gustiness_coefficient); without the flag it raises
    the reference's error."""
    flux_configuration = str(flux_configuration).lstrip(":")
    velocity_formulation = str(velocity_formulation).lstrip(":")
    albedo = getattr(getattr(radiation, "sea_ice_surface", None), "albedo", None)
    common = dict(radiation=radiation, ocean_minimum_salinity=float(ocean_minimum_salinity),
                  sea_ice_albedo=albedo if isinstance(albedo, ic.SeaIceAlbedo) else None)
    if flux_configuration == "default":
        interfaces = ComponentInterfaces(atmosphere, ocean, sea_ice, **common)
        return OceanSeaIceModel(ocean, sea_ice, atmosphere=atmosphere, land=land, interfaces=interfaces)
    if velocity_formulation == "relative":
        velocity_difference = ic.RelativeVelocity()
    elif velocity_formulation == "wind":
        velocity_difference = ic.WindVelocity()
    else:
        raise ValueError(f"Unknown velocity_formulation: {velocity_formulation}. Options: :relative, :wind")
    if flux_configuration == "corrected":
        ao, ai = ic.corrected_atmosphere_ocean_fluxes(), ic.corrected_atmosphere_sea_ice_fluxes()
    elif flux_configuration == "ncar":
        ao, ai = ic.ncar_atmosphere_ocean_fluxes(), ic.ncar_atmosphere_sea_ice_fluxes()
    elif flux_configuration == "shear_aware" and allow_shear_aware:
        ao = ic.shear_aware_atmosphere_ocean_fluxes(shear_gustiness_coefficient=shear_gustiness_coefficient)
        ai = ic.corrected_atmosphere_sea_ice_fluxes()
    else:
        raise ValueError(f"Unknown flux_configuration: {flux_configuration}. Options: :default, :corrected, :ncar")
    interfaces = ComponentInterfaces(atmosphere, ocean, sea_ice, atmosphere_ocean_fluxes=ao, atmosphere_sea_ice_fluxes=ai,
                                     sea_ice_ocean_heat_flux=ic.corrected_ice_ocean_heat_flux() if sea_ice is not None else None,
                                     atmosphere_ocean_velocity_difference=velocity_difference,
                                     atmosphere_sea_ice_velocity_difference=velocity_difference, **common)
    return OceanSeaIceModel(ocean, sea_ice, atmosphere=atmosphere, land=land, interfaces=interfaces)


def OceanOnlyModel(ocean, *, atmosphere, **kw):
    """docs/src/index.md:64 alias."""
    return OceanSeaIceModel(ocean, None, atmosphere=atmosphere, **kw)


def update_state(model):
    """update_state!(coupled_model) — the accelerated path."""
    itf, atm = model.interfaces, model.atmosphere
    src, n1, n2, frac = atm.source(itf.context, model.clock.time)
    if itf.fold_north:
        # TripolarGrid: the north halo of the ocean surface state is the fold of its own rows (Oceananigans' zipper
        # boundary fills it in the reference; here cf_fold_north_halo, ring + 1 rows: the solver's ring row reads v[j+1])
        st = model.ocean.surface_state()
        itf.context.fold_north_halo([st["T"], st["S"], st["u"], st["v"]],
                                    [abi.FOLD_CENTER, abi.FOLD_CENTER, abi.FOLD_X_FACE, abi.FOLD_Y_FACE], [1.0, 1.0, -1.0, -1.0], rows=2)
    if getattr(model, "land", None) is not None:
        # JRA55PrescribedLand: river discharge + calving at the model time, handed to compute_net_ocean_fluxes!
        land = model.land
        if not hasattr(itf, "land_freshwater"):
            itf.land_freshwater = itf.context.zeros()
        l1, l2, lfrac = land.levels(model.clock.time)   # (the record's calendar: cyclic repeat year / clamped multi-year)
        itf.context.interpolate_land_freshwater(land.data["friver"], land.data.get("licalvf"), itf.weights, itf.land_freshwater,
                                                level1=l1, level2=l2, time_fraction=lfrac)
        itf.context.set_land_freshwater(itf.land_freshwater)
    if model.sea_ice is not None and itf.sea_ice_ocean_heat_flux is not None:
        # compute_sea_ice_ocean_fluxes!: the three-equation exchange and frazil from the current ocean surface and the
        # ice–ocean stress; its outputs ARE the partition's interface_heat / salt_flux and the ice's frazil heat
        si, f = model.sea_ice, itf.sea_ice_ocean_fluxes
        itf.context.compute_sea_ice_ocean_fluxes(itf.ice_ocean_params, model.ocean.surface_state(), si.concentration,
                                                 si.x_stress, si.y_stress, f)
        si.interface_heat, si.salt_flux, si.frazil_heat = f["interface_heat"], f["salt_flux"], f["frazil_heat"]
    ice = model.sea_ice.fields() if model.sea_ice is not None else None
    # Inside run!(simulation) the clock is known one step ahead: the NEXT step's atmosphere state is requested into the other
    # set of exchange fields and rides in this step's solver launch (CF_OPT_MERGED_PREFETCH = 2: tail workgroups; the same
    # bits as interpolating it when its step comes).  A sliding window needs a third slot for that: the request makes the
    # next bracketing snapshots resident while this step may still read the current ones.
    if getattr(itf, "_next_state_in_other", False):   # requested by the previous step: this step's state is in the other set
        itf._exchange_current = 1 - itf._exchange_current
        itf._next_state_in_other = False
    dt_next = getattr(model, "_pipeline_dt", None)
    pipelined = dt_next is not None and (atm.provider is None or atm.n_slots >= 3)
    if pipelined:
        if itf._exchange_other is None:
            itf._exchange_sets[1 - itf._exchange_current] = itf.context.field_set(EXCHANGE_NAMES)
        if not getattr(itf, "_tail_mode", False):   # (restored by run!: a lone update_state! is ≈ 5 µs slower on the tail plan)
            itf.context.set_option(abi.OPT_MERGED_PREFETCH, 2)
            itf._tail_mode = True
        src_n, n1n, n2n, frac_n = atm.source(itf.context, model.clock.time + dt_next)
        itf.context.prefetch_atmosphere_state(src_n, itf.weights, itf._exchange_other, level1=n1n, level2=n2n, time_fraction=frac_n)
    if itf.atmosphere_sea_ice_interface is not None:
        # the ocean path, then compute_atmosphere_sea_ice_fluxes! + compute_net_sea_ice_fluxes! in one ABI call; the
        # skin temperature found by the iteration becomes the sea ice's top surface temperature (and the next step's
        # first guess)
        si, ai = model.sea_ice, itf.atmosphere_sea_ice_interface._fields
        itf.context.update_state_sea_ice(src, itf.weights, model.ocean.surface_state(), itf.exchange_atmosphere_state,
                                         itf.atmosphere_ocean_interface._fields, itf.net_fluxes._ocean_fields, ice,
                                         si.surface_state(), ai, itf.net_fluxes._sea_ice_fields,
                                         frazil_heat=si.frazil_heat, interface_heat=si.interface_heat,
                                         level1=n1, level2=n2, time_fraction=frac)
        si.top_surface_temperature.copy_(ai["temperature"])
    else:
        itf.context.update_state(src, itf.weights, model.ocean.surface_state(), itf.exchange_atmosphere_state,
                                 itf.atmosphere_ocean_interface._fields, itf.net_fluxes._ocean_fields, ice=ice,
                                 level1=n1, level2=n2, time_fraction=frac)
    itf._next_state_in_other = pipelined   # (the next update_state! swaps the sets first: until then
                                           #  `exchange_atmosphere_state` is the state at the model's clock)


def time_step(model, dt):
    """time_step!(coupled_model, Δt): component steps, tick, update_state!."""
    model.ocean.time_step(dt)
    model.clock.time += dt
    model.clock.iteration += 1
    update_state(model)


@dataclass
class Simulation:
    model: OceanSeaIceModel
    dt: float = 20 * minutes          # README.md:76
    stop_time: float = float("inf")
    stop_iteration: int = 2 ** 62


def run(simulation):
    """run!(simulation) — README.md:77."""
    m = simulation.model
    m._pipeline_dt = simulation.dt   # (update_state requests every next atmosphere state: see there)
    try:
        while m.clock.time < simulation.stop_time and m.clock.iteration < simulation.stop_iteration:
            time_step(m, simulation.dt)
    finally:
        m._pipeline_dt = None
        itf = m.interfaces
        # the state at the model's clock goes back to set 0 (what callers may hold), the pending request is dropped, and the
        # context returns to the options it had: the chunk plan of a lone update_state!, the auxiliary-stream prefetch
        if getattr(itf, "_next_state_in_other", False):
            itf._next_state_in_other = False     # (the requested state of a step that will not run; the next run! asks again)
        itf.context.discard_prefetched_atmosphere_state()
        if itf._exchange_current != 0:
            for k, t in itf._exchange_sets[0].items():
                t.copy_(itf._exchange_sets[1][k])
            itf._exchange_current = 0
        if getattr(itf, "_tail_mode", False):
            itf.context.set_option(abi.OPT_MERGED_PREFETCH, 0)
            itf._tail_mode = False
    m.interfaces.context.sync()


class JRA55PrescribedLand:
    """JRA55PrescribedLand(arch; …) — atmosphere.jl:46: river discharge and calving (friver, licalvf of
    jra55_data_staging.jl:8) as 3-hourly Float32 windows on the JRA55 grid; OceanSeaIceModel(…; land) interpolates them
    each step and the freshwater reaches the salinity flux.

    Two backends, like the atmosphere: the whole record in memory (`snapshots` = {friver, licalvf}: [n, 320, 640]), or a
    sliding window of `time_indices_in_memory` device slots fed by `provider(n) -> {friver, licalvf}: [320, 640]`
    (`total_snapshots` records; `cyclic` = RepeatYearJRA55, clamped = MultiYearJRA55).  The window follows the monotone
    snapshot COUNTER (slot = counter mod slots), so a repeat year whose length is no multiple of the slot count never
    puts two consecutive snapshots in one slot.  Two 0.8 MB planes per 3 h: read synchronously, no prefetch thread."""

    def __init__(self, snapshots=None, *, provider=None, total_snapshots=None, time_indices_in_memory=2, time_interval=3 * hours,
                 device=0, cyclic=True, source_size=(synthetic.JRA55_NX, synthetic.JRA55_NY)):
        dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.time_interval = time_interval
        self.cyclic = cyclic
        self.provider = provider
        if provider is not None:
            if total_snapshots is None:
                raise ValueError("a snapshot provider needs total_snapshots")
            if time_indices_in_memory < 2:
                raise ValueError("time_indices_in_memory must be >= 2")
            self.n_levels = total_snapshots
            self.n_slots = time_indices_in_memory
            shape = (self.n_slots, source_size[1], source_size[0])
            self.data = {k: torch.zeros(shape, dtype=torch.float32, device=dev) for k in ("friver", "licalvf")}
            self._slot_counter = [None] * self.n_slots   # which snapshot counter each slot holds
            return
        if snapshots is None:
            snapshots = synthetic.jra55_land_snapshots(time_indices_in_memory)
        self.data = {k: torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(dev) for k, v in snapshots.items()}
        self.n_levels = self.data["friver"].shape[0]

    def _wrap(self, n):
        return n % self.n_levels if self.cyclic else min(max(n, 0), self.n_levels - 1)

    def levels(self, t):
        """(level1, level2, ñ) into `self.data` for model time t — the bracketing snapshots made resident first on the
        sliding-window backend.  Cyclic (repeat year) or clamped (multi-year record ends) like the atmosphere."""
        x = t / self.time_interval
        base = int(np.floor(x))
        frac = x - base
        if self.cyclic:
            k1, k2 = base, base + 1
        else:
            k1 = min(max(base, 0), self.n_levels - 1)
            k2 = min(max(base + 1, 0), self.n_levels - 1)
            if not (0 <= base < self.n_levels - 1):
                frac = 0.0
        if self.provider is None:
            return self._wrap(k1), self._wrap(k2), frac
        for k in (k1, k2):
            slot = k % self.n_slots
            if self._slot_counter[slot] != k:
                snap = self.provider(self._wrap(k))
                for v, dst in self.data.items():
                    plane = snap.get(v)
                    if plane is None:
                        dst[slot].zero_()
                    else:
                        dst[slot].copy_(torch.from_numpy(np.array(plane, dtype=np.float32)))  # (a copy: memory-mapped planes are read-only)
                self._slot_counter[slot] = k
        return k1 % self.n_slots, k2 % self.n_slots, frac


class NormalizeSalinity:
    """NormalizeSalinity (omip_simulation.jl:187-220): callable on the coupled model; subtracts the area-weighted global
    mean of (bulk salinity flux + materialised additional flux) from the bulk flux field.  Dispatches on the salinity top
    boundary condition as `salinity_normalizer` does: MultipleFluxes ⇒ the additional flux is materialised into a buffer
    first (`_materialize_top_flux!`), bare field ⇒ no additional flux."""

    def __init__(self, ocean, area=None):
        bc = ocean.model.top_boundary_conditions.S
        self.flux_field = bc.flux_field if isinstance(bc, MultipleFluxes) else bc
        self.additional_fluxes = bc.additional_fluxes if isinstance(bc, MultipleFluxes) else None
        self.additional_buffer = torch.zeros_like(self.flux_field) if self.additional_fluxes is not None else None
        self.area = area
        self.mean_total = torch.zeros(1, dtype=torch.float64, device=self.flux_field.device)

    def __call__(self, model):
        ctx, ocean = model.interfaces.context, model.ocean
        if self.additional_fluxes is not None:
            r = self.additional_fluxes
            ctx.materialize_salinity_restoring(r.velocity, r.target, ocean.surface_state(), self.additional_buffer)
        ctx.normalize_salinity_flux(self.flux_field, ocean.model.wet_mask, additional=self.additional_buffer, area=self.area,
                                    mean_out=self.mean_total)
