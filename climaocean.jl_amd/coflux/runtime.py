"""Thin runtime over the C ABI: a FluxContext owns one cf_ctx (one GPU), fields are torch CUDA
tensors used purely as device-memory plumbing (their data_ptr() goes through the ABI).

There is no CPU compute path here.  Creating a FluxContext without a GPU / without
libcoflux.so raises.
"""
import ctypes as C

import numpy as np
import torch

from . import abi

EXCHANGE_NAMES = ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")
FLUX_NAMES = ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum", "temperature")
FLUX_OPTIONAL = ("friction_velocity", "temperature_scale", "humidity_scale")
NET_NAMES = ("u", "v", "T", "S", "shortwave_surface_flux", "upwelling_longwave",
             "downwelling_longwave", "downwelling_shortwave")
ICE_NAMES = ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")


class CofluxError(RuntimeError):
    pass


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class FluxContext:
    """One libcoflux context bound to `device` (cuda:N)."""

    def __init__(self, nx, ny, hx, hy, params, ring=1, device=0):
        self.lib = abi.load_library()
        if not torch.cuda.is_available():
            raise CofluxError("coflux needs a HIP device (torch.cuda.is_available() is False); "
                              "there is no CPU compute path")
        self.device = torch.device("cuda", device)
        self.grid = abi.Grid(nx, ny, hx, hy, ring, 0)
        self.params = params
        self.shape = (ny + 2 * hy, nx + 2 * hx)
        self._h = C.c_void_p()
        rc = self.lib.cf_create(C.byref(self._h), device, C.byref(self.grid), C.byref(params))
        if rc != 0:
            raise CofluxError(f"cf_create failed ({rc}): {self.lib.cf_last_error(None).decode()}")
        self.use_torch_stream()

    # -- lifecycle ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.cf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise CofluxError(f"{what} failed ({rc}): {self.lib.cf_last_error(self._h).decode()}")

    def use_torch_stream(self):
        """Order libcoflux launches on torch's current stream for this device."""
        s = torch.cuda.current_stream(self.device).cuda_stream
        # torch's default stream has handle 0 = the legacy null stream; 0 means "own stream" in the ABI
        self._check(self.lib.cf_set_stream(self._h, C.c_void_p(s if s else 1)), "cf_set_stream")

    def set_flux_params(self, params):
        self._check(self.lib.cf_set_flux_params(self._h, C.byref(params)), "cf_set_flux_params")
        self.params = params

    def set_option(self, option, value):
        self._check(self.lib.cf_set_option(self._h, option, value), "cf_set_option")

    def debug_eval(self, function, x):
        """y = f(x) with the device primitives of the solver (self-test hook)."""
        y = torch.empty_like(x)
        self._check(self.lib.cf_debug_eval(self._h, function, x.numel(), _ptr(x), _ptr(y)), "cf_debug_eval")
        return y

    def sync(self):
        self._check(self.lib.cf_sync(self._h), "cf_sync")

    # -- field helpers ------------------------------------------------------------------------
    def zeros(self, dtype=torch.float64):
        return torch.zeros(self.shape, dtype=dtype, device=self.device)

    def to_device(self, a):
        return torch.as_tensor(np.ascontiguousarray(a)).to(self.device)

    def field_set(self, names, optional=()):
        d = {n: self.zeros() for n in names}
        for n in optional:
            d[n] = self.zeros()
        return d

    # -- struct builders ----------------------------------------------------------------------
    @staticmethod
    def _struct(cls, fields, names):
        s = cls()
        for n in names:
            t = fields.get(n) if fields is not None else None
            if t is not None:
                assert t.is_cuda and t.is_contiguous(), n
                setattr(s, n, t.data_ptr())
        return s

    def source_struct(self, src, level1, level2, time_fraction):
        if isinstance(src, abi.AtmosSource):  # built by SnapshotWindow.source(): levels and fraction are in it
            return src
        s = abi.AtmosSource()
        shape = None
        for k, name in enumerate(abi.JRA55_VARIABLES):
            t = src[name]
            assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous(), name
            s.data[k] = t.data_ptr()
            shape = t.shape
        s.n_levels, s.ns_y, s.ns_x = shape
        s.level1, s.level2, s.time_fraction = level1, level2, float(time_fraction)
        return s

    def weights_struct(self, w):
        s = abi.InterpWeights()
        if w is None:
            return s
        s.separable = 1 if w.get("separable", True) else 0
        for n in ("fi", "fj", "cos_rot", "sin_rot", "latitude"):
            t = w.get(n)
            if t is not None:
                assert t.dtype == torch.float64 and t.is_cuda and t.is_contiguous(), n
                setattr(s, n, t.data_ptr())
        return s

    def ocean_struct(self, ocean):
        return self._struct(abi.OceanSurface, ocean, ("T", "S", "u", "v", "mask"))

    def exchange_struct(self, atmos):
        return self._struct(abi.ExchangeFields, atmos, EXCHANGE_NAMES)

    def fluxes_struct(self, fluxes):
        return self._struct(abi.InterfaceFluxes, fluxes, FLUX_NAMES + FLUX_OPTIONAL + ("iterations",))

    def ice_struct(self, ice):
        return None if ice is None else self._struct(abi.SeaIceFields, ice, ICE_NAMES)

    def net_struct(self, net):
        return self._struct(abi.NetOceanFluxes, net, NET_NAMES)

    # -- the hot path ---------------------------------------------------------------------------
    def interpolate_atmosphere_state(self, src, weights, atmos, level1=0, level2=1, time_fraction=0.0):
        s = self.source_struct(src, level1, level2, time_fraction)
        w = self.weights_struct(weights)
        e = self.exchange_struct(atmos)
        self._check(self.lib.cf_interpolate_atmosphere_state(self._h, C.byref(s), C.byref(w), C.byref(e)),
                    "cf_interpolate_atmosphere_state")

    def compute_atmosphere_ocean_fluxes(self, ocean, atmos, fluxes):
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(fluxes)
        self._check(self.lib.cf_compute_atmosphere_ocean_fluxes(self._h, C.byref(o), C.byref(e), C.byref(f)),
                    "cf_compute_atmosphere_ocean_fluxes")

    def set_sea_ice_formulation(self, ice_flux_params, sea_ice_params=None):
        """atmosphere_sea_ice_fluxes + SkinTemperature(ConductiveFlux) properties of the
        atmosphere–sea-ice interface (ComponentInterfaces, omip_simulation.jl:139-158)."""
        if sea_ice_params is None:
            sea_ice_params = abi.SeaIceParams()
            self._check(self.lib.cf_default_sea_ice_params(C.byref(sea_ice_params)), "cf_default_sea_ice_params")
        self._check(self.lib.cf_set_sea_ice_formulation(self._h, C.byref(ice_flux_params), C.byref(sea_ice_params)),
                    "cf_set_sea_ice_formulation")

    def compute_atmosphere_sea_ice_fluxes(self, ice_state, ocean, atmos, fluxes):
        st = self._struct(abi.SeaIceState, ice_state,
                          ("concentration", "thickness", "top_temperature", "u", "v", "albedo", "snow_thickness"))
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(fluxes)
        self._check(self.lib.cf_compute_atmosphere_sea_ice_fluxes(self._h, C.byref(st), C.byref(o), C.byref(e),
                                                                  C.byref(f)),
                    "cf_compute_atmosphere_sea_ice_fluxes")

    def compute_net_sea_ice_fluxes(self, ice_state, ocean, atmos, ai_fluxes, out, frazil_heat=None, interface_heat=None):
        """compute_net_sea_ice_fluxes!: out = dict(top_heat=…, bottom_heat=…)."""
        st = self._struct(abi.SeaIceState, ice_state, ("concentration", "thickness", "top_temperature", "albedo", "snow_thickness"))
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(ai_fluxes)
        n = self._struct(abi.NetSeaIceFluxes, out, ("top_heat", "bottom_heat"))
        self._check(self.lib.cf_compute_net_sea_ice_fluxes(
            self._h, C.byref(st), C.byref(o), C.byref(e), C.byref(f),
            frazil_heat.data_ptr() if frazil_heat is not None else None,
            interface_heat.data_ptr() if interface_heat is not None else None, C.byref(n)),
            "cf_compute_net_sea_ice_fluxes")

    def default_sea_ice_albedo_params(self):
        p = abi.SeaIceAlbedoParams()
        self._check(self.lib.cf_default_sea_ice_albedo_params(C.byref(p)), "cf_default_sea_ice_albedo_params")
        return p

    def set_sea_ice_albedo(self, params=None):
        """SeaIceAlbedo(hi, hs, Ts) (CCSM3, atmosphere.jl:30-44) wherever the sea-ice state carries no albedo field."""
        self._check(self.lib.cf_set_sea_ice_albedo(self._h, C.byref(params) if params is not None else None),
                    "cf_set_sea_ice_albedo")

    def compute_sea_ice_albedo(self, params, thickness, snow_thickness, top_temperature, out):
        self._check(self.lib.cf_compute_sea_ice_albedo(self._h, C.byref(params), _ptr(thickness), _ptr(snow_thickness),
                                                       _ptr(top_temperature), _ptr(out)), "cf_compute_sea_ice_albedo")

    def default_ice_ocean_params(self, **kw):
        p = abi.IceOceanParams()
        self._check(self.lib.cf_default_ice_ocean_params(C.byref(p)), "cf_default_ice_ocean_params")
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def compute_sea_ice_ocean_fluxes(self, params, ocean, concentration, x_stress, y_stress, out):
        """compute_sea_ice_ocean_fluxes!: ThreeEquationHeatFlux(MomentumBasedFrictionVelocity) + frazil
        (omip_simulation.jl:71-77).  out: dict interface_heat, salt_flux[, frazil_heat, friction_velocity]."""
        o = self.ocean_struct(ocean)
        f = self._struct(abi.IceOceanFluxes, out, ("interface_heat", "salt_flux", "frazil_heat", "friction_velocity"))
        self._check(self.lib.cf_compute_sea_ice_ocean_fluxes(self._h, C.byref(params), C.byref(o), _ptr(concentration),
                                                             _ptr(x_stress), _ptr(y_stress), C.byref(f)),
                    "cf_compute_sea_ice_ocean_fluxes")

    def interpolate_land_freshwater(self, friver, licalvf, weights, out, level1=0, level2=1, time_fraction=0.0):
        """JRA55PrescribedLand (atmosphere.jl:46): friver + licalvf windows [n, ns_y, ns_x] float32 → one ocean-grid field."""
        s = abi.LandSource()
        assert friver.dtype == torch.float32 and friver.is_cuda and friver.is_contiguous()
        s.friver = friver.data_ptr()
        s.licalvf = licalvf.data_ptr() if licalvf is not None else None
        s.n_levels, s.ns_y, s.ns_x = friver.shape
        s.level1, s.level2, s.time_fraction = level1, level2, float(time_fraction)
        w = self.weights_struct(weights)
        self._check(self.lib.cf_interpolate_land_freshwater(self._h, C.byref(s), C.byref(w), _ptr(out)),
                    "cf_interpolate_land_freshwater")

    def set_land_freshwater(self, field):
        self._land = field      # keep the tensor alive: the library borrows the pointer
        self._check(self.lib.cf_set_land_freshwater(self._h, _ptr(field)), "cf_set_land_freshwater")

    def materialize_salinity_restoring(self, piston_velocity, target, ocean, out):
        """SurfaceFluxRestoring as the additional flux of MultipleFluxes (omip_simulation.jl:175-206, 507-523)."""
        o = self.ocean_struct(ocean)
        self._check(self.lib.cf_materialize_salinity_restoring(self._h, float(piston_velocity), _ptr(target), C.byref(o), _ptr(out)),
                    "cf_materialize_salinity_restoring")

    def compute_net_ocean_fluxes(self, ocean, atmos, fluxes, net, ice=None, weights=None):
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(fluxes)
        i = self.ice_struct(ice)
        w = self.weights_struct(weights)
        n = self.net_struct(net)
        self._check(self.lib.cf_compute_net_ocean_fluxes(self._h, C.byref(o), C.byref(e), C.byref(f),
                                                         C.byref(i) if i is not None else None,
                                                         C.byref(w), C.byref(n)),
                    "cf_compute_net_ocean_fluxes")

    def update_state(self, src, weights, ocean, atmos, fluxes, net, ice=None, level1=0, level2=1,
                     time_fraction=0.0):
        s = self.source_struct(src, level1, level2, time_fraction)
        w = self.weights_struct(weights)
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(fluxes)
        i = self.ice_struct(ice)
        n = self.net_struct(net)
        self._check(self.lib.cf_update_state(self._h, C.byref(s), C.byref(w), C.byref(o), C.byref(e),
                                             C.byref(f), C.byref(i) if i is not None else None, C.byref(n)),
                    "cf_update_state")

    def update_state_sea_ice(self, src, weights, ocean, atmos, fluxes, net, ice, ice_state, ai_fluxes, net_ice,
                             frazil_heat=None, interface_heat=None, level1=0, level2=1, time_fraction=0.0):
        """update_state! of a model with sea ice: the ocean path, then the atmosphere–sea-ice interface and the
        net sea-ice fluxes (cf_update_state_sea_ice)."""
        s = self.source_struct(src, level1, level2, time_fraction)
        w = self.weights_struct(weights)
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(fluxes)
        i = self.ice_struct(ice)
        n = self.net_struct(net)
        st = self._struct(abi.SeaIceState, ice_state, ("concentration", "thickness", "top_temperature", "u", "v", "albedo", "snow_thickness"))
        af = self.fluxes_struct(ai_fluxes)
        ni = self._struct(abi.NetSeaIceFluxes, net_ice, ("top_heat", "bottom_heat"))
        self._check(self.lib.cf_update_state_sea_ice(
            self._h, C.byref(s), C.byref(w), C.byref(o), C.byref(e), C.byref(f), C.byref(i) if i is not None else None,
            C.byref(n), C.byref(st), C.byref(af), frazil_heat.data_ptr() if frazil_heat is not None else None,
            interface_heat.data_ptr() if interface_heat is not None else None, C.byref(ni)), "cf_update_state_sea_ice")

    def time_stage(self, stage, launches, *, src=None, weights=None, ocean=None, atmos=None, fluxes=None,
                   net=None, ice=None, level1=0, level2=1, time_fraction=0.0):
        """Average ms per launch of one stage, measured with HIP events on the launch stream."""
        s = self.source_struct(src, level1, level2, time_fraction) if src is not None else abi.AtmosSource()
        w = self.weights_struct(weights)
        o, e, f = self.ocean_struct(ocean), self.exchange_struct(atmos), self.fluxes_struct(fluxes)
        i = self.ice_struct(ice)
        n = self.net_struct(net)
        ms = C.c_double()
        self._check(self.lib.cf_time_stage(self._h, stage, launches, C.byref(s), C.byref(w), C.byref(o),
                                           C.byref(e), C.byref(f), C.byref(i) if i is not None else None,
                                           C.byref(n), C.byref(ms)), "cf_time_stage")
        return ms.value

    def ensure_chunk_table(self, mask):
        """Build the solver's schedule for `mask` now instead of inside the first step."""
        self._check(self.lib.cf_ensure_chunk_table(self._h, _ptr(mask)), "cf_ensure_chunk_table")

    def solver_path(self):
        """(lean_kernel, fused): which kernels cf_update_state launches for the current formulation and options; fused = 0
        three launches, 1 net fluxes in the solver's epilogue, 2 the interpolation in its prologue as well."""
        lean, fused = C.c_int(), C.c_int()
        self._check(self.lib.cf_solver_path(self._h, C.byref(lean), C.byref(fused)), "cf_solver_path")
        return bool(lean.value), fused.value

    def solver_iteration_path(self):
        """abi.SOLVER_PATH_EXACT or abi.SOLVER_PATH_CERTIFIED: what the ocean solve would run with the current options."""
        path = C.c_int()
        self._check(self.lib.cf_solver_iteration_path(self._h, C.byref(path)), "cf_solver_iteration_path")
        return path.value

    def solver_latency_layout(self):
        """True when the exact path's launch would take the kernels laid out for one or two waves per SIMD
        (CF_OPT_LATENCY_LAYOUT; valid once the chunk table is built)."""
        layout = C.c_int()
        self._check(self.lib.cf_solver_latency_layout(self._h, C.byref(layout)), "cf_solver_latency_layout")
        return bool(layout.value)

    def time_copy(self, nbytes, launches=20):
        a = torch.empty(nbytes // 8, dtype=torch.float64, device=self.device)
        b = torch.empty_like(a)
        ms = C.c_double()
        self._check(self.lib.cf_time_copy(self._h, _ptr(b), _ptr(a), nbytes, launches, C.byref(ms)),
                    "cf_time_copy")
        return ms.value

    def normalize_salinity_flux(self, flux, mask, additional=None, area=None, mean_out=None):
        """NormalizeSalinity (omip_simulation.jl:182-220): flux -= area-weighted mean over wet cells."""
        self._check(self.lib.cf_normalize_salinity_flux(self._h, _ptr(flux), _ptr(additional), _ptr(area), _ptr(mask),
                                                        _ptr(mean_out)), "cf_normalize_salinity_flux")

    def profile_enable(self, max_records):
        self._check(self.lib.cf_profile_enable(self._h, max_records), "cf_profile_enable")

    def profile_read(self, kernel):
        """(average ms, records) of kernel 0 (interpolate), 1 (atmosphere–ocean fluxes) or 2 (net fluxes)."""
        ms, n = C.c_double(), C.c_int()
        self._check(self.lib.cf_profile_read(self._h, kernel, C.byref(ms), C.byref(n)), "cf_profile_read")
        return ms.value, n.value

    def prefetch_atmosphere_state(self, src, weights, atmos_next, level1=0, level2=1, time_fraction=0.0):
        """Start the NEXT step's interpolate_atmosphere_state! on the auxiliary stream (cf_prefetch_atmosphere_state)."""
        s = self.source_struct(src, level1, level2, time_fraction)
        w = self.weights_struct(weights)
        e = self.exchange_struct(atmos_next)
        self._check(self.lib.cf_prefetch_atmosphere_state(self._h, C.byref(s), C.byref(w), C.byref(e)),
                    "cf_prefetch_atmosphere_state")

    def discard_prefetched_atmosphere_state(self):
        """Forget requested-ahead atmosphere states (cf_discard_prefetched_atmosphere_state): on leaving a stepping loop."""
        self._check(self.lib.cf_discard_prefetched_atmosphere_state(self._h), "cf_discard_prefetched_atmosphere_state")

    def make_schedule(self, ocean_states, atmos_sets, *, first_level=0, time_fraction=0.0, time_fraction_increment=0.0,
                      pipeline=False, halo_backend=abi.HALO_NONE, halo_rows=0, fold_north=False):
        """cf_run_schedule for cf_time_steps; keeps the ctypes arrays alive on the returned object."""
        sch = abi.RunSchedule()
        oc = (abi.OceanSurface * len(ocean_states))(*[self.ocean_struct(o) for o in ocean_states])
        at = (abi.ExchangeFields * len(atmos_sets))(*[self.exchange_struct(a) for a in atmos_sets])
        sch.struct_size = C.sizeof(abi.RunSchedule)
        sch.n_ocean_states, sch.ocean_states = len(ocean_states), oc
        sch.n_atmos_sets, sch.atmos, sch.pipeline = len(atmos_sets), at, int(pipeline)   # False / True (within the call) / abi.PIPELINE_CONTINUING
        sch.first_level, sch.halo_backend, sch.halo_rows = first_level, halo_backend, halo_rows
        sch.fold_north = 1 if fold_north else 0
        sch.time_fraction, sch.time_fraction_increment = float(time_fraction), float(time_fraction_increment)
        sch._keep = (oc, at, ocean_states, atmos_sets)
        return sch

    def time_steps(self, first_step, nsteps, schedule, src, weights, fluxes, net, ice=None):
        """run!(simulation) of a prescribed-ocean model: nsteps × (halo rows → update_state!) inside libcoflux."""
        s = self.source_struct(src, 0, 0, 0.0)
        w = self.weights_struct(weights)
        f, n, i = self.fluxes_struct(fluxes), self.net_struct(net), self.ice_struct(ice)
        self._check(self.lib.cf_time_steps(self._h, first_step, nsteps, C.byref(schedule), C.byref(s), C.byref(w),
                                           C.byref(f), C.byref(i) if i is not None else None, C.byref(n)),
                    "cf_time_steps")

    # -- peer-direct halo rows / tripolar fold -----------------------------------------------------
    def peer_halo_export(self, max_fields=4, max_rows=2):
        buf = C.create_string_buffer(abi.PEER_HANDLE_BYTES)
        self._check(self.lib.cf_peer_halo_export(self._h, max_fields, max_rows, buf), "cf_peer_halo_export")
        return buf.raw

    def peer_halo_connect(self, south, north, rank, nranks):
        sb = C.create_string_buffer(south, abi.PEER_HANDLE_BYTES) if south is not None else None
        nb = C.create_string_buffer(north, abi.PEER_HANDLE_BYTES) if north is not None else None
        self._check(self.lib.cf_peer_halo_connect(self._h, sb, nb, rank, nranks), "cf_peer_halo_connect")

    def peer_halo_stats(self):
        """(exchanges issued, of which as riders of a solver launch: CF_OPT_HALO_IN_SOLVER_LAUNCH)."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        self._check(self.lib.cf_peer_halo_stats(self._h, C.byref(a), C.byref(b)), "cf_peer_halo_stats")
        return a.value, b.value

    def halo_exchange_rows_peer(self, tensors, rows=2):
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        self._check(self.lib.cf_halo_exchange_rows_peer(self._h, arr, len(tensors), rows), "cf_halo_exchange_rows_peer")

    def fold_north_halo(self, tensors, locations, signs, rows=2):
        n = len(tensors)
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        loc = (C.c_int * n)(*locations)
        sg = (C.c_double * n)(*signs)
        self._check(self.lib.cf_fold_north_halo(self._h, arr, loc, sg, n, rows), "cf_fold_north_halo")

    # -- RCCL halo rows -------------------------------------------------------------------------
    def comm_init(self, unique_id: bytes, rank, nranks):
        buf = C.create_string_buffer(unique_id, abi.COMM_ID_BYTES)
        self._check(self.lib.cf_comm_init(self._h, buf, rank, nranks), "cf_comm_init")

    def comm_count(self):
        """(nranks, rank, device) as RCCL itself reports them for the communicator of comm_init (cf_comm_count)."""
        n, r, d = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.cf_comm_count(self._h, C.byref(n), C.byref(r), C.byref(d)), "cf_comm_count")
        return n.value, r.value, d.value

    def halo_exchange_rows(self, tensors, rows=1):
        arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        self._check(self.lib.cf_halo_exchange_rows(self._h, arr, len(tensors), rows), "cf_halo_exchange_rows")


class SnapshotWindow:
    """JRA55PrescribedAtmosphere(arch; time_indices_in_memory = n_slots, prefetch = true): cf_window_* —
    `n_slots` snapshots of the nine JRA55 variables in HBM, refilled asynchronously from pinned staging
    buffers on a copy stream (atmosphere.jl:20-29).  Snapshot t lives in slot t mod n_slots."""

    def __init__(self, ctx, ns_x, ns_y, n_slots):
        self.ctx, self.lib = ctx, ctx.lib
        self.ns_x, self.ns_y, self.n_slots = ns_x, ns_y, n_slots
        h = C.c_void_p()
        ctx._check(self.lib.cf_window_create(ctx._h, ns_x, ns_y, n_slots, C.byref(h)), "cf_window_create")
        self._h = h

    def close(self):
        if self._h:
            self.lib.cf_window_destroy(self._h)
            self._h = None

    def host_view(self, slot, variable):
        """NumPy view (ns_y, ns_x) float32 of the pinned staging buffer of (slot, variable)."""
        k = abi.JRA55_VARIABLES.index(variable) if isinstance(variable, str) else variable
        p = self.lib.cf_window_host_buffer(self._h, slot, k)
        if not p:
            raise CofluxError(f"no staging buffer for slot {slot}, variable {variable}")
        return np.ctypeslib.as_array(p, shape=(self.ns_y, self.ns_x))

    def wait_slot(self, slot):
        self.ctx._check(self.lib.cf_window_wait_slot(self._h, slot), "cf_window_wait_slot")

    def commit(self, slot, time_index):
        self.ctx._check(self.lib.cf_window_commit(self._h, slot, time_index), "cf_window_commit")

    def upload(self, time_index, snapshot):
        """snapshot: dict variable -> float32 (ns_y, ns_x) host array."""
        keep = [np.ascontiguousarray(snapshot[v], dtype=np.float32) for v in abi.JRA55_VARIABLES]
        for a in keep:
            assert a.shape == (self.ns_y, self.ns_x), a.shape
        ptrs = (C.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
        self.ctx._check(self.lib.cf_window_upload(self._h, time_index, ptrs), "cf_window_upload")

    def find(self, time_index):
        return self.lib.cf_window_find(self._h, time_index)

    def source(self, n1, n2, time_fraction):
        s = abi.AtmosSource()
        self.ctx._check(self.lib.cf_window_source(self._h, n1, n2, float(time_fraction), C.byref(s)), "cf_window_source")
        return s


def comm_unique_id():
    lib = abi.load_library()
    buf = C.create_string_buffer(abi.COMM_ID_BYTES)
    rc = lib.cf_comm_unique_id(buf)
    if rc != 0:
        raise CofluxError(f"cf_comm_unique_id failed: {lib.cf_last_error(None).decode()}")
    return buf.raw
