"""ctypes mirror of include/coflux.h (the C ABI of libcoflux).

Field order and types must match the header exactly; tests/test_abi.py checks sizeof against
the value the library reports and that every declared symbol is exported.
"""
import ctypes as C
import os

ABI_VERSION = 3
COMM_ID_BYTES = 128
PEER_HANDLE_BYTES = 64
HALO_NONE, HALO_RCCL, HALO_PEER = 0, 1, 2
FOLD_CENTER, FOLD_X_FACE, FOLD_Y_FACE = 0, 1, 2
SKIN_EXPLICIT, SKIN_SEMI_IMPLICIT = 0, 1

# enums (values from include/coflux.h)
SIMILARITY_LOGARITHMIC, SIMILARITY_COARE_LOGARITHMIC = 0, 1
STABILITY_EDSON2013, STABILITY_SHEBA, STABILITY_LARGE_YEAGER = 0, 1, 2
ROUGHNESS_CONSTANT, ROUGHNESS_CHARNOCK, ROUGHNESS_WIND_CHARNOCK = 0, 1, 2
SCALAR_ROUGHNESS_CONSTANT, SCALAR_ROUGHNESS_REYNOLDS = 0, 1
VISCOSITY_CONSTANT, VISCOSITY_TEMPERATURE_DEPENDENT = 0, 1
STOP_CONVERGENCE, STOP_FIXED = 0, 1
FORMULATION_SIMILARITY, FORMULATION_LARGE_YEAGER = 0, 1
VELOCITY_RELATIVE, VELOCITY_WIND = 0, 1
MASK_NONE, MASK_U8, MASK_BOTTOM_HEIGHT = 0, 1, 2
ALBEDO_CONSTANT, ALBEDO_LATITUDE_DEPENDENT = 0, 1
OPT_SOLVER, OPT_TRIP_HINTS, OPT_FUSED_NET, OPT_ICE_ORBIT_SHORTCUT, OPT_MERGED_PREFETCH = 0, 3, 6, 7, 9
OPT_INTERP_TILE_CAP, OPT_AO_CHUNK = 1, 4   # experiment options: the library accepts them only with COFLUX_EXPERIMENTS=1 in the environment
OPT_SOLVER_PATH, OPT_CERTIFIED_BUDGET, OPT_ICE_FREE_CELLS, OPT_LATENCY_LAYOUT, OPT_HALO_IN_SOLVER_LAUNCH = 10, 11, 12, 13, 14
ICE_FREE_ITERATE, ICE_FREE_ZERO = 0, 1
PIPELINE_WITHIN_CALL, PIPELINE_CONTINUING = 1, 2   # cf_run_schedule.pipeline
SOLVER_PATH_EXACT, SOLVER_PATH_CERTIFIED = 0, 1      # how the Monin–Obukhov fixed point is reached (include/coflux.h)
CERTIFIED_EXACT_FLAG = 0x100                         # `iterations` of a cell the certified path solved on the exact path
SOLVER_TABLES, SOLVER_LIBM = 0, 1
STAGE_INTERPOLATE, STAGE_AO_FLUXES, STAGE_NET_FLUXES, STAGE_UPDATE_STATE = 0, 1, 2, 3

JRA55_VARIABLES = ("tas", "huss", "psl", "uas", "vas", "rlds", "rsds", "prra", "prsn")
JRA55_NVARS = len(JRA55_VARIABLES)

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class Grid(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("hx", C.c_int32), ("hy", C.c_int32),
                ("ring", C.c_int32), ("reserved", C.c_int32)]


class Roughness(C.Structure):
    _fields_ = [("kind", C.c_int32), ("viscosity_kind", C.c_int32),
                ("constant_length", C.c_double), ("maximum_length", C.c_double),
                ("charnock", C.c_double), ("laminar", C.c_double),
                ("wind_a1", C.c_double), ("wind_a2", C.c_double), ("wind_umax", C.c_double),
                ("reynolds_A", C.c_double), ("reynolds_b", C.c_double),
                ("viscosity", C.c_double * 4)]


class Thermodynamics(C.Structure):
    _fields_ = [("gas_constant", C.c_double), ("dry_air_molar_mass", C.c_double),
                ("water_molar_mass", C.c_double), ("kappa_d", C.c_double),
                ("cp_v", C.c_double), ("cp_l", C.c_double), ("cp_i", C.c_double),
                ("LH_v0", C.c_double), ("LH_s0", C.c_double),
                ("T_0", C.c_double), ("T_triple", C.c_double), ("p_triple", C.c_double),
                ("T_freeze", C.c_double), ("T_icenuc", C.c_double), ("pow_icenuc", C.c_double)]


class Seawater(C.Structure):
    _fields_ = [("water_molar_mass", C.c_double),
                ("constituent_molar_mass", C.c_double * 4),
                ("constituent_mass_fraction", C.c_double * 4)]


class FluxParams(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("abi_version", C.c_int32),
                ("similarity_form", C.c_int32), ("stability_functions", C.c_int32),
                ("stop_kind", C.c_int32), ("maxiter", C.c_int32),
                ("velocity_difference", C.c_int32), ("mask_kind", C.c_int32),
                ("tolerance", C.c_double), ("von_karman", C.c_double),
                ("gustiness_parameter", C.c_double), ("minimum_gustiness", C.c_double),
                ("shear_gustiness_coefficient", C.c_double),
                ("similarity_profile_floor", C.c_double),
                ("momentum_roughness", Roughness), ("temperature_roughness", Roughness),
                ("water_vapor_roughness", Roughness),
                ("reference_height", C.c_double), ("boundary_layer_height", C.c_double),
                ("gravitational_acceleration", C.c_double),
                ("thermo", Thermodynamics), ("seawater", Seawater),
                ("ocean_reference_density", C.c_double), ("ocean_heat_capacity", C.c_double),
                ("ocean_freshwater_density", C.c_double),
                ("ocean_temperature_offset", C.c_double), ("ocean_minimum_salinity", C.c_double),
                ("ocean_surface_z", C.c_double),
                ("ocean_albedo_kind", C.c_int32), ("penetrating_shortwave", C.c_int32),
                ("ocean_albedo", C.c_double), ("ocean_albedo_diffuse", C.c_double),
                ("ocean_albedo_direct", C.c_double), ("ocean_emissivity", C.c_double),
                ("stefan_boltzmann", C.c_double),
                ("flux_formulation", C.c_int32), ("reserved1", C.c_int32),
                ("ly_minimum_wind", C.c_double), ("ly_zeta_bound", C.c_double), ("ly_cd", C.c_double * 4),
                ("ly_high_wind", C.c_double), ("ly_cd_high", C.c_double), ("ly_ce", C.c_double),
                ("ly_ch_stable", C.c_double), ("ly_ch_unstable", C.c_double)]


class OceanSurface(C.Structure):
    _fields_ = [("T", C.c_void_p), ("S", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p),
                ("mask", C.c_void_p)]


class ExchangeFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("u", "v", "T", "p", "q", "Qs", "Ql", "Mp")]


class InterfaceFluxes(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("sensible_heat", "latent_heat", "water_vapor", "x_momentum", "y_momentum",
                 "temperature", "friction_velocity", "temperature_scale", "humidity_scale",
                 "iterations")]


class SeaIceFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("concentration", "interface_heat", "salt_flux", "x_stress", "y_stress")]


class SeaIceParams(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("skin_temperature_scheme", C.c_int32),
                ("conductivity", C.c_double), ("consolidation_thickness", C.c_double),
                ("maximum_temperature_change", C.c_double), ("ice_salinity", C.c_double),
                ("liquidus_slope", C.c_double), ("freshwater_melting_temperature", C.c_double),
                ("albedo", C.c_double), ("emissivity", C.c_double), ("temperature_offset", C.c_double)]


class NetSeaIceFluxes(C.Structure):
    _fields_ = [("top_heat", C.c_void_p), ("bottom_heat", C.c_void_p)]


class SeaIceState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("concentration", "thickness", "top_temperature", "u", "v", "albedo",
                                          "snow_thickness")]


class SeaIceAlbedoParams(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32)] + [(n, C.c_double) for n in (
        "ice_visible", "ice_near_infrared", "snow_visible", "snow_near_infrared", "ocean_albedo", "reference_thickness",
        "melt_temperature_range", "ice_melt_change", "snow_melt_change_visible", "snow_melt_change_near_infrared",
        "snow_patch_thickness", "visible_fraction", "melting_temperature")]


class IceOceanParams(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("reserved", C.c_int32)] + [(n, C.c_double) for n in (
        "heat_transfer_coefficient", "salt_transfer_coefficient", "minimum_friction_velocity", "ice_density",
        "latent_heat_of_fusion", "ice_salinity", "liquidus_slope", "top_cell_thickness", "time_step")]


class IceOceanFluxes(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("interface_heat", "salt_flux", "frazil_heat", "friction_velocity")]


class NetOceanFluxes(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("u", "v", "T", "S", "shortwave_surface_flux", "upwelling_longwave",
                 "downwelling_longwave", "downwelling_shortwave")]


class AtmosSource(C.Structure):
    _fields_ = [("data", C.c_void_p * JRA55_NVARS),
                ("ns_x", C.c_int32), ("ns_y", C.c_int32), ("n_levels", C.c_int32),
                ("level1", C.c_int32), ("level2", C.c_int32),
                ("time_fraction", C.c_double)]


class LandSource(C.Structure):
    _fields_ = [("friver", C.c_void_p), ("licalvf", C.c_void_p),
                ("ns_x", C.c_int32), ("ns_y", C.c_int32), ("n_levels", C.c_int32),
                ("level1", C.c_int32), ("level2", C.c_int32), ("reserved", C.c_int32),
                ("time_fraction", C.c_double)]


class RunSchedule(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("n_ocean_states", C.c_int32),
                ("ocean_states", C.POINTER(OceanSurface)),
                ("n_atmos_sets", C.c_int32), ("pipeline", C.c_int32),
                ("atmos", C.POINTER(ExchangeFields)),
                ("first_level", C.c_int32), ("halo_backend", C.c_int32),
                ("halo_rows", C.c_int32), ("fold_north", C.c_int32),
                ("time_fraction", C.c_double), ("time_fraction_increment", C.c_double)]


class InterpWeights(C.Structure):
    _fields_ = [("separable", C.c_int32), ("reserved", C.c_int32),
                ("fi", C.c_void_p), ("fj", C.c_void_p),
                ("cos_rot", C.c_void_p), ("sin_rot", C.c_void_p), ("latitude", C.c_void_p)]


# Every symbol include/coflux.h declares (tests check they are all exported).
EXPORTED_SYMBOLS = (
    "cf_version", "cf_default_flux_params", "cf_create", "cf_destroy", "cf_last_error",
    "cf_set_flux_params", "cf_set_stream", "cf_set_option", "cf_debug_eval", "cf_debug_chunk_plan", "cf_sync",
    "cf_device_alloc", "cf_device_free", "cf_h2d", "cf_d2h",
    "cf_interpolate_atmosphere_state", "cf_compute_atmosphere_ocean_fluxes",
    "cf_compute_net_ocean_fluxes", "cf_update_state", "cf_normalize_salinity_flux",
    "cf_default_sea_ice_params", "cf_set_sea_ice_formulation", "cf_compute_atmosphere_sea_ice_fluxes",
    "cf_compute_net_sea_ice_fluxes", "cf_update_state_sea_ice",
    "cf_time_stage", "cf_time_copy", "cf_profile_enable", "cf_profile_read",
    "cf_comm_unique_id", "cf_comm_init", "cf_comm_destroy", "cf_halo_exchange_rows",
    "cf_peer_halo_export", "cf_peer_halo_connect", "cf_halo_exchange_rows_peer", "cf_peer_halo_stats", "cf_fold_north_halo",
    "cf_time_steps", "cf_prefetch_atmosphere_state",
    "cf_default_sea_ice_albedo_params", "cf_set_sea_ice_albedo", "cf_compute_sea_ice_albedo",
    "cf_default_ice_ocean_params", "cf_compute_sea_ice_ocean_fluxes",
    "cf_interpolate_land_freshwater", "cf_set_land_freshwater", "cf_materialize_salinity_restoring",
    "cf_window_create", "cf_window_destroy", "cf_window_host_buffer", "cf_window_wait_slot", "cf_window_commit",
    "cf_window_upload", "cf_window_find", "cf_window_source",
    "cf_ensure_chunk_table", "cf_solver_path", "cf_solver_iteration_path", "cf_solver_latency_layout", "cf_comm_count", "cf_build_stamp", "cf_discard_prefetched_atmosphere_state",
)

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIBCOFLUX") or os.path.join(os.path.dirname(PACKAGE_DIR), "csrc", "libcoflux.so")


class CofluxLibraryMissing(RuntimeError):
    pass


_lib = None


def source_stamp():
    """sha256[:16] over csrc/*.{hip,cpp,hpp,h} + include/coflux.h + csrc/tools/gcn_sched.py (the slab kernels are built through
    it) in the Makefile's order, or None when the sources are not there (an installed library without its tree)."""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
    names = sorted(os.path.basename(f) for ext in ("hip", "cpp", "hpp", "h") for f in glob.glob(os.path.join(csrc, "*." + ext)))
    header = os.path.join(csrc, "..", "..", "include", "coflux.h")
    if not names or not os.path.exists(header):
        return None
    # GNU make's $(sort) orders byte-wise, "../../include/coflux.h" ahead of every plain file name
    h = hashlib.sha256()
    tool = os.path.join(csrc, "tools", "gcn_sched.py")
    for f in [header] + [os.path.join(csrc, n) for n in names] + ([tool] if os.path.exists(tool) else []):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_library(path=None):
    """dlopen libcoflux.so.  There is NO CPU fallback: a missing library is a hard error."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise CofluxLibraryMissing(
            f"{p} not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). The HIP library is the only compute path.")
    lib = C.CDLL(p)
    vp = C.c_void_p
    lib.cf_version.restype = C.c_int
    lib.cf_build_stamp.restype = C.c_char_p
    want, have = source_stamp(), lib.cf_build_stamp().decode()
    if want is not None and have != want and os.environ.get("COFLUX_ALLOW_STALE_LIBRARY") != "1":
        raise CofluxLibraryMissing(
            f"{p} was built from other sources than this tree's (stamp {have}, tree {want}): rebuild it with "
            "`python __graft_entry__.py build` (COFLUX_ALLOW_STALE_LIBRARY=1 loads it anyway: A/B builds only)")
    lib.cf_default_flux_params.argtypes = [C.POINTER(FluxParams)]
    lib.cf_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Grid), C.POINTER(FluxParams)]
    lib.cf_destroy.argtypes = [vp]
    lib.cf_last_error.argtypes = [vp]
    lib.cf_last_error.restype = C.c_char_p
    lib.cf_set_flux_params.argtypes = [vp, C.POINTER(FluxParams)]
    lib.cf_set_stream.argtypes = [vp, vp]
    lib.cf_set_option.argtypes = [vp, C.c_int, C.c_int]
    lib.cf_solver_iteration_path.argtypes = [vp, C.POINTER(C.c_int)]
    lib.cf_solver_latency_layout.argtypes = [vp, C.POINTER(C.c_int)]
    lib.cf_peer_halo_stats.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    lib.cf_discard_prefetched_atmosphere_state.argtypes = [vp]
    lib.cf_comm_count.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cf_debug_eval.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    lib.cf_debug_chunk_plan.argtypes = [C.c_longlong, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.cf_sync.argtypes = [vp]
    lib.cf_device_alloc.argtypes = [vp, C.c_size_t]
    lib.cf_device_alloc.restype = vp
    lib.cf_device_free.argtypes = [vp, vp]
    lib.cf_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.cf_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.cf_interpolate_atmosphere_state.argtypes = [
        vp, C.POINTER(AtmosSource), C.POINTER(InterpWeights), C.POINTER(ExchangeFields)]
    lib.cf_compute_atmosphere_ocean_fluxes.argtypes = [
        vp, C.POINTER(OceanSurface), C.POINTER(ExchangeFields), C.POINTER(InterfaceFluxes)]
    lib.cf_compute_net_ocean_fluxes.argtypes = [
        vp, C.POINTER(OceanSurface), C.POINTER(ExchangeFields), C.POINTER(InterfaceFluxes),
        C.POINTER(SeaIceFields), C.POINTER(InterpWeights), C.POINTER(NetOceanFluxes)]
    lib.cf_update_state.argtypes = [
        vp, C.POINTER(AtmosSource), C.POINTER(InterpWeights), C.POINTER(OceanSurface),
        C.POINTER(ExchangeFields), C.POINTER(InterfaceFluxes), C.POINTER(SeaIceFields),
        C.POINTER(NetOceanFluxes)]
    lib.cf_normalize_salinity_flux.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.cf_default_sea_ice_params.argtypes = [C.POINTER(SeaIceParams)]
    lib.cf_set_sea_ice_formulation.argtypes = [vp, C.POINTER(FluxParams), C.POINTER(SeaIceParams)]
    lib.cf_compute_atmosphere_sea_ice_fluxes.argtypes = [
        vp, C.POINTER(SeaIceState), C.POINTER(OceanSurface), C.POINTER(ExchangeFields), C.POINTER(InterfaceFluxes)]
    lib.cf_compute_net_sea_ice_fluxes.argtypes = [
        vp, C.POINTER(SeaIceState), C.POINTER(OceanSurface), C.POINTER(ExchangeFields), C.POINTER(InterfaceFluxes),
        vp, vp, C.POINTER(NetSeaIceFluxes)]
    lib.cf_update_state_sea_ice.argtypes = [
        vp, C.POINTER(AtmosSource), C.POINTER(InterpWeights), C.POINTER(OceanSurface), C.POINTER(ExchangeFields),
        C.POINTER(InterfaceFluxes), C.POINTER(SeaIceFields), C.POINTER(NetOceanFluxes), C.POINTER(SeaIceState),
        C.POINTER(InterfaceFluxes), vp, vp, C.POINTER(NetSeaIceFluxes)]
    lib.cf_time_stage.argtypes = [
        vp, C.c_int, C.c_int, C.POINTER(AtmosSource), C.POINTER(InterpWeights),
        C.POINTER(OceanSurface), C.POINTER(ExchangeFields), C.POINTER(InterfaceFluxes),
        C.POINTER(SeaIceFields), C.POINTER(NetOceanFluxes), C.POINTER(C.c_double)]
    lib.cf_time_copy.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    lib.cf_profile_enable.argtypes = [vp, C.c_int]
    lib.cf_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.cf_comm_unique_id.argtypes = [vp]
    lib.cf_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.cf_comm_destroy.argtypes = [vp]
    lib.cf_halo_exchange_rows.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int]
    lib.cf_peer_halo_export.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.cf_peer_halo_connect.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    lib.cf_halo_exchange_rows_peer.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int]
    lib.cf_fold_north_halo.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.c_int]
    lib.cf_time_steps.argtypes = [
        vp, C.c_int64, C.c_int, C.POINTER(RunSchedule), C.POINTER(AtmosSource), C.POINTER(InterpWeights),
        C.POINTER(InterfaceFluxes), C.POINTER(SeaIceFields), C.POINTER(NetOceanFluxes)]
    lib.cf_prefetch_atmosphere_state.argtypes = [vp, C.POINTER(AtmosSource), C.POINTER(InterpWeights), C.POINTER(ExchangeFields)]
    lib.cf_ensure_chunk_table.argtypes = [vp, vp]
    lib.cf_solver_path.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.cf_default_sea_ice_albedo_params.argtypes = [C.POINTER(SeaIceAlbedoParams)]
    lib.cf_set_sea_ice_albedo.argtypes = [vp, C.POINTER(SeaIceAlbedoParams)]
    lib.cf_compute_sea_ice_albedo.argtypes = [vp, C.POINTER(SeaIceAlbedoParams), vp, vp, vp, vp]
    lib.cf_default_ice_ocean_params.argtypes = [C.POINTER(IceOceanParams)]
    lib.cf_compute_sea_ice_ocean_fluxes.argtypes = [vp, C.POINTER(IceOceanParams), C.POINTER(OceanSurface), vp, vp, vp,
                                                    C.POINTER(IceOceanFluxes)]
    lib.cf_interpolate_land_freshwater.argtypes = [vp, C.POINTER(LandSource), C.POINTER(InterpWeights), vp]
    lib.cf_set_land_freshwater.argtypes = [vp, vp]
    lib.cf_materialize_salinity_restoring.argtypes = [vp, C.c_double, vp, C.POINTER(OceanSurface), vp]
    lib.cf_window_create.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.cf_window_destroy.argtypes = [vp]
    lib.cf_window_host_buffer.argtypes = [vp, C.c_int32, C.c_int32]
    lib.cf_window_host_buffer.restype = C.POINTER(C.c_float)
    lib.cf_window_wait_slot.argtypes = [vp, C.c_int32]
    lib.cf_window_commit.argtypes = [vp, C.c_int32, C.c_int64]
    lib.cf_window_upload.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    lib.cf_window_find.argtypes = [vp, C.c_int64]
    lib.cf_window_source.argtypes = [vp, C.c_int64, C.c_int64, C.c_double, C.POINTER(AtmosSource)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("cf_last_error", "cf_device_alloc", "cf_window_host_buffer", "cf_build_stamp"):
            fn.restype = C.c_int
    if path is None:
        _lib = lib
    return lib
