# CoFluxMI355X.jl — the reference-side binding of libcoflux (include/coflux.h).
#
# NOT EXECUTED IN THIS REPOSITORY: the build image has no Julia toolchain (SURVEY.md F3), so this
# file is the stub a ClimaOcean / NumericalEarth maintainer would add; every ccall below is kept
# one-to-one with a C entry point that IS exercised (through ctypes) by tests/.
#
# Seam: the reference reaches the flux path by multiple dispatch, not through an FFI
# (SURVEY.md §8b).  `update_state!(::OceanSeaIceModel)` calls, in order,
#     interpolate_atmosphere_state!, compute_atmosphere_ocean_fluxes!, compute_net_ocean_fluxes!
# (NEMOTKE/nemo_tke_compute_closure_fields.jl:7-8 names the call; the bodies live in
# NumericalEarth.EarthSystemModels).  A coupled model whose `interfaces` carry a `CoFluxBackend`
# dispatches those three functions here; nothing else in the user's script changes:
#
#     ocean      = ocean_simulation(grid)                        # README.md:67
#     atmosphere = JRA55PrescribedAtmosphere(arch)               # README.md:74
#     coupled    = OceanSeaIceModel(ocean; atmosphere,
#                      interfaces = CoFluxMI355X.interfaces(atmosphere, ocean))
#     run!(Simulation(coupled, Δt = 20minutes, stop_time = 30days))
#
# Host code stays Julia; no AMDGPU.jl / KernelAbstractions: device buffers are raw pointers owned
# by libcoflux (cf_device_alloc) and wrapped in `unsafe_wrap`-free handle structs.
module CoFluxMI355X

const libcoflux = get(ENV, "LIBCOFLUX", "libcoflux.so")

# ---- mirrors of the POD structs in include/coflux.h (field order is the ABI) -----------------
struct CfGrid
    nx::Int32; ny::Int32; hx::Int32; hy::Int32; ring::Int32; reserved::Int32
end

struct CfRoughness
    kind::Int32; viscosity_kind::Int32
    constant_length::Float64; maximum_length::Float64; charnock::Float64; laminar::Float64
    wind_a1::Float64; wind_a2::Float64; wind_umax::Float64
    reynolds_A::Float64; reynolds_b::Float64
    viscosity::NTuple{4, Float64}
end

struct CfThermodynamics
    gas_constant::Float64; dry_air_molar_mass::Float64; water_molar_mass::Float64; kappa_d::Float64
    cp_v::Float64; cp_l::Float64; cp_i::Float64; LH_v0::Float64; LH_s0::Float64
    T_0::Float64; T_triple::Float64; p_triple::Float64; T_freeze::Float64; T_icenuc::Float64; pow_icenuc::Float64
end

struct CfSeawater
    water_molar_mass::Float64
    constituent_molar_mass::NTuple{4, Float64}
    constituent_mass_fraction::NTuple{4, Float64}
end

mutable struct CfFluxParams
    struct_size::Int32; abi_version::Int32
    similarity_form::Int32; stability_functions::Int32; stop_kind::Int32; maxiter::Int32
    velocity_difference::Int32; mask_kind::Int32
    tolerance::Float64; von_karman::Float64; gustiness_parameter::Float64; minimum_gustiness::Float64
    shear_gustiness_coefficient::Float64
    similarity_profile_floor::Float64
    momentum_roughness::CfRoughness; temperature_roughness::CfRoughness; water_vapor_roughness::CfRoughness
    reference_height::Float64; boundary_layer_height::Float64; gravitational_acceleration::Float64
    thermo::CfThermodynamics; seawater::CfSeawater
    ocean_reference_density::Float64; ocean_heat_capacity::Float64; ocean_freshwater_density::Float64
    ocean_temperature_offset::Float64; ocean_minimum_salinity::Float64; ocean_surface_z::Float64
    ocean_albedo_kind::Int32; penetrating_shortwave::Int32
    ocean_albedo::Float64; ocean_albedo_diffuse::Float64; ocean_albedo_direct::Float64
    ocean_emissivity::Float64; stefan_boltzmann::Float64
    flux_formulation::Int32; reserved1::Int32            # 0 SimilarityTheoryFluxes, 1 CoefficientBasedFluxes (Large–Yeager)
    ly_minimum_wind::Float64; ly_zeta_bound::Float64; ly_cd::NTuple{4, Float64}
    ly_high_wind::Float64; ly_cd_high::Float64; ly_ce::Float64; ly_ch_stable::Float64; ly_ch_unstable::Float64
    CfFluxParams() = new()
end

struct CfOceanSurface;   T::Ptr{Float64}; S::Ptr{Float64}; u::Ptr{Float64}; v::Ptr{Float64}; mask::Ptr{Cvoid}; end
struct CfExchangeFields; u::Ptr{Float64}; v::Ptr{Float64}; T::Ptr{Float64}; p::Ptr{Float64}; q::Ptr{Float64}
                         Qs::Ptr{Float64}; Ql::Ptr{Float64}; Mp::Ptr{Float64}; end
struct CfInterfaceFluxes
    sensible_heat::Ptr{Float64}; latent_heat::Ptr{Float64}; water_vapor::Ptr{Float64}
    x_momentum::Ptr{Float64}; y_momentum::Ptr{Float64}; temperature::Ptr{Float64}
    friction_velocity::Ptr{Float64}; temperature_scale::Ptr{Float64}; humidity_scale::Ptr{Float64}
    iterations::Ptr{Int32}
end
struct CfSeaIceFields
    concentration::Ptr{Float64}; interface_heat::Ptr{Float64}; salt_flux::Ptr{Float64}
    x_stress::Ptr{Float64}; y_stress::Ptr{Float64}
end
struct CfNetOceanFluxes
    u::Ptr{Float64}; v::Ptr{Float64}; T::Ptr{Float64}; S::Ptr{Float64}; shortwave_surface_flux::Ptr{Float64}
    upwelling_longwave::Ptr{Float64}; downwelling_longwave::Ptr{Float64}; downwelling_shortwave::Ptr{Float64}
end
struct CfAtmosSource
    data::NTuple{9, Ptr{Float32}}            # tas huss psl uas vas rlds rsds prra prsn (jra55_data_staging.jl:8)
    ns_x::Int32; ns_y::Int32; n_levels::Int32; level1::Int32; level2::Int32
    time_fraction::Float64
end
struct CfInterpWeights
    separable::Int32; reserved::Int32
    fi::Ptr{Float64}; fj::Ptr{Float64}; cos_rot::Ptr{Float64}; sin_rot::Ptr{Float64}; latitude::Ptr{Float64}
end

# ---- error handling: Julia exceptions on the Julia side, status codes across the ABI ----------
struct CoFluxError <: Exception; code::Cint; msg::String; end
last_error(ctx) = unsafe_string(ccall((:cf_last_error, libcoflux), Cstring, (Ptr{Cvoid},), ctx))
check(ctx, rc) = rc == 0 ? nothing : throw(CoFluxError(rc, last_error(ctx)))

# ---- lifecycle ---------------------------------------------------------------------------------
mutable struct CoFluxBackend
    ctx::Ptr{Cvoid}
    grid::CfGrid
    params::CfFluxParams
end

function CoFluxBackend(device::Integer, grid::CfGrid, params::CfFluxParams)
    ctx = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:cf_create, libcoflux), Cint, (Ref{Ptr{Cvoid}}, Cint, Ref{CfGrid}, Ref{CfFluxParams}),
               ctx, device, grid, params)
    rc == 0 || throw(CoFluxError(rc, last_error(C_NULL)))
    backend = CoFluxBackend(ctx[], grid, params)
    finalizer(b -> ccall((:cf_destroy, libcoflux), Cint, (Ptr{Cvoid},), b.ctx), backend)
    return backend
end

# sha256[:16] of the sources the loaded library was built from (include/coflux.h): compare with the tree the stub ships with
build_stamp() = unsafe_string(ccall((:cf_build_stamp, libcoflux), Cstring, ()))

default_flux_params() = (p = CfFluxParams(); ccall((:cf_default_flux_params, libcoflux), Cint, (Ref{CfFluxParams},), p); p)

device_alloc(b, bytes) = ccall((:cf_device_alloc, libcoflux), Ptr{Cvoid}, (Ptr{Cvoid}, Csize_t), b.ctx, bytes)
h2d!(b, dst, src::Array) = check(b.ctx, ccall((:cf_h2d, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                                              b.ctx, dst, src, sizeof(src)))
d2h!(b, dst::Array, src) = check(b.ctx, ccall((:cf_d2h, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Csize_t),
                                              b.ctx, dst, src, sizeof(dst)))
sync(b) = check(b.ctx, ccall((:cf_sync, libcoflux), Cint, (Ptr{Cvoid},), b.ctx))

# How the SimilarityTheoryFluxes fixed point is reached (include/coflux.h, CF_OPT_SOLVER_PATH): the reference's own
# iteration (:exact, default) or the certified reduced-iteration solve with per-cell exact-path fallback (:certified,
# every cell within `budget` — default 8e-7, in the flux metric of coflux.h — of the exact path's result).
const CF_OPT_SOLVER_PATH = Cint(10)
const CF_OPT_CERTIFIED_BUDGET = Cint(11)
const CF_OPT_LATENCY_LAYOUT = Cint(13)   # 0 never, 1 automatic (default), 2 always: the exact path's kernels for one or two waves per SIMD
set_option!(b, option, value) = check(b.ctx, ccall((:cf_set_option, libcoflux), Cint, (Ptr{Cvoid}, Cint, Cint), b.ctx, option, value))
function set_solver_path!(b, path::Symbol; budget = 8e-7)
    path in (:exact, :certified) || throw(ArgumentError("solver path must be :exact or :certified"))
    set_option!(b, CF_OPT_SOLVER_PATH, path == :certified ? 1 : 0)
    path == :certified && set_option!(b, CF_OPT_CERTIFIED_BUDGET, round(Cint, budget * 1e9))
    return b
end
solver_iteration_path(b) = (p = Ref{Cint}(0); check(b.ctx, ccall((:cf_solver_iteration_path, libcoflux), Cint, (Ptr{Cvoid}, Ref{Cint}), b.ctx, p));
                            p[] == 1 ? :certified : :exact)

solver_latency_layout(b) = (p = Ref{Cint}(0); check(b.ctx, ccall((:cf_solver_latency_layout, libcoflux), Cint, (Ptr{Cvoid}, Ref{Cint}), b.ctx, p)); p[] == 1)
# CF_OPT_HALO_IN_SOLVER_LAUNCH: a step's peer-direct halo rows as rider workgroups of its solver launch (include/coflux.h)
const CF_OPT_HALO_IN_SOLVER_LAUNCH = Cint(14)
halo_in_solver_launch!(b, on::Bool) = set_option!(b, CF_OPT_HALO_IN_SOLVER_LAUNCH, on ? 1 : 0)
function peer_halo_stats(b)
    n = Ref{Culonglong}(0); m = Ref{Culonglong}(0)
    check(b.ctx, ccall((:cf_peer_halo_stats, libcoflux), Cint, (Ptr{Cvoid}, Ref{Culonglong}, Ref{Culonglong}), b.ctx, n, m))
    return (exchanges = n[], in_solver_launch = m[])
end

# ---- the three functions of update_state! ------------------------------------------------------
# Each body is ONE ccall; these are the methods a maintainer adds for
#   interpolate_atmosphere_state!(interfaces::…{<:CoFluxBackend}, atmosphere, coupled_model) etc.
interpolate_atmosphere_state!(b::CoFluxBackend, src::CfAtmosSource, w::CfInterpWeights, out::CfExchangeFields) =
    check(b.ctx, ccall((:cf_interpolate_atmosphere_state, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfAtmosSource}, Ref{CfInterpWeights}, Ref{CfExchangeFields}), b.ctx, src, w, out))

compute_atmosphere_ocean_fluxes!(b::CoFluxBackend, ocean::CfOceanSurface, atmos::CfExchangeFields, out::CfInterfaceFluxes) =
    check(b.ctx, ccall((:cf_compute_atmosphere_ocean_fluxes, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfOceanSurface}, Ref{CfExchangeFields}, Ref{CfInterfaceFluxes}), b.ctx, ocean, atmos, out))

compute_net_ocean_fluxes!(b::CoFluxBackend, ocean, atmos, fluxes, ice::Union{CfSeaIceFields, Nothing}, w, out::CfNetOceanFluxes) =
    check(b.ctx, ccall((:cf_compute_net_ocean_fluxes, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfOceanSurface}, Ref{CfExchangeFields}, Ref{CfInterfaceFluxes}, Ptr{CfSeaIceFields},
                        Ref{CfInterpWeights}, Ref{CfNetOceanFluxes}),
                       b.ctx, ocean, atmos, fluxes, ice === nothing ? C_NULL : Ref(ice), w, out))

# update_state!(coupled_model): the fused path (one launch for interpolation + solver, one for the net fluxes)
update_state!(b::CoFluxBackend, src, w, ocean, atmos, fluxes, ice, net) =
    check(b.ctx, ccall((:cf_update_state, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfAtmosSource}, Ref{CfInterpWeights}, Ref{CfOceanSurface}, Ref{CfExchangeFields},
                        Ref{CfInterfaceFluxes}, Ptr{CfSeaIceFields}, Ref{CfNetOceanFluxes}),
                       b.ctx, src, w, ocean, atmos, fluxes, ice === nothing ? C_NULL : Ref(ice), net))

# NormalizeSalinity callback (src/OMIPConfigurations/omip_simulation.jl:182-220): flux .-= ⟨flux + additional⟩_A,wet
normalize_salinity_flux!(b::CoFluxBackend, flux::Ptr{Float64}, additional, area, mask, mean_out = C_NULL) =
    check(b.ctx, ccall((:cf_normalize_salinity_flux, libcoflux), Cint,
                       (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}, Ptr{Float64}),
                       b.ctx, flux, additional, area, mask, mean_out))

# ---- atmosphere–sea-ice interface: compute_atmosphere_sea_ice_fluxes!(coupled_model) ---------------
# SkinTemperature(ConductiveFlux) + SurfaceRadiationProperties(sea_ice_albedo, 1.0) (atmosphere.jl:34-44)
mutable struct CfSeaIceParams
    struct_size::Int32; skin_temperature_scheme::Int32       # CF_SKIN_EXPLICIT = 0 / CF_SKIN_SEMI_IMPLICIT = 1
    conductivity::Float64; consolidation_thickness::Float64; maximum_temperature_change::Float64
    ice_salinity::Float64; liquidus_slope::Float64; freshwater_melting_temperature::Float64
    albedo::Float64; emissivity::Float64; temperature_offset::Float64
    CfSeaIceParams() = new()
end
struct CfSeaIceState   # sea_ice.model.{ice_concentration, ice_thickness, top_surface_temperature, velocities}
    concentration::Ptr{Float64}; thickness::Ptr{Float64}; top_temperature::Ptr{Float64}
    u::Ptr{Float64}; v::Ptr{Float64}; albedo::Ptr{Float64}; snow_thickness::Ptr{Float64}   # snow: atmosphere.jl:34
end
function default_sea_ice_params()
    p = CfSeaIceParams()
    ccall((:cf_default_sea_ice_params, libcoflux), Cint, (Ref{CfSeaIceParams},), p)
    return p
end
# `ice_fluxes` = corrected_atmosphere_sea_ice_fluxes(FT) / ncar_atmosphere_sea_ice_fluxes(FT) lowered to CfFluxParams
set_sea_ice_formulation!(b::CoFluxBackend, ice_fluxes::CfFluxParams, ice::CfSeaIceParams = default_sea_ice_params()) =
    check(b.ctx, ccall((:cf_set_sea_ice_formulation, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfFluxParams}, Ref{CfSeaIceParams}), b.ctx, ice_fluxes, ice))
compute_atmosphere_sea_ice_fluxes!(b::CoFluxBackend, ice::CfSeaIceState, ocean::CfOceanSurface,
                                   atmos::CfExchangeFields, out::CfInterfaceFluxes) =
    check(b.ctx, ccall((:cf_compute_atmosphere_sea_ice_fluxes, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfSeaIceState}, Ref{CfOceanSurface}, Ref{CfExchangeFields}, Ref{CfInterfaceFluxes}),
                       b.ctx, ice, ocean, atmos, out))

struct CfNetSeaIceFluxes; top_heat::Ptr{Float64}; bottom_heat::Ptr{Float64}; end
# compute_net_sea_ice_fluxes!(coupled_model): ΣQt, ΣQb handed to the sea-ice thermodynamics (frazil / interface may be C_NULL)
compute_net_sea_ice_fluxes!(b::CoFluxBackend, ice::CfSeaIceState, ocean::CfOceanSurface, atmos::CfExchangeFields,
                            ai::CfInterfaceFluxes, frazil::Ptr{Float64}, interface::Ptr{Float64}, out::CfNetSeaIceFluxes) =
    check(b.ctx, ccall((:cf_compute_net_sea_ice_fluxes, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfSeaIceState}, Ref{CfOceanSurface}, Ref{CfExchangeFields}, Ref{CfInterfaceFluxes},
                        Ptr{Float64}, Ptr{Float64}, Ref{CfNetSeaIceFluxes}), b.ctx, ice, ocean, atmos, ai, frazil, interface, out))

# update_state!(coupled_model) with sea ice: ocean path + atmosphere–sea-ice interface + net sea-ice fluxes, five launches
update_state_sea_ice!(b::CoFluxBackend, src::CfAtmosSource, w::CfInterpWeights, ocean::CfOceanSurface, atmos::CfExchangeFields,
                      ao::CfInterfaceFluxes, partition::CfSeaIceFields, net::CfNetOceanFluxes, ice::CfSeaIceState,
                      ai::CfInterfaceFluxes, frazil::Ptr{Float64}, interface::Ptr{Float64}, net_ice::CfNetSeaIceFluxes) =
    check(b.ctx, ccall((:cf_update_state_sea_ice, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfAtmosSource}, Ref{CfInterpWeights}, Ref{CfOceanSurface}, Ref{CfExchangeFields},
                        Ref{CfInterfaceFluxes}, Ref{CfSeaIceFields}, Ref{CfNetOceanFluxes}, Ref{CfSeaIceState},
                        Ref{CfInterfaceFluxes}, Ptr{Float64}, Ptr{Float64}, Ref{CfNetSeaIceFluxes}),
                       b.ctx, src, w, ocean, atmos, ao, partition, net, ice, ai, frazil, interface, net_ice))

# ---- JRA55 snapshot window in HBM: JRA55PrescribedAtmosphere(arch; time_indices_in_memory, prefetch) ---------
mutable struct CoFluxWindow
    ptr::Ptr{Cvoid}
    backend::CoFluxBackend
    nsx::Int; nsy::Int; nslots::Int
end
function CoFluxWindow(b::CoFluxBackend, nsx, nsy, nslots)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(b.ctx, ccall((:cf_window_create, libcoflux), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
                       b.ctx, nsx, nsy, nslots, out))
    w = CoFluxWindow(out[], b, nsx, nsy, nslots)
    finalizer(x -> ccall((:cf_window_destroy, libcoflux), Cint, (Ptr{Cvoid},), x.ptr), w)
    return w
end
# pinned staging buffer of (slot, variable) as a Julia array the NetCDF reader fills in place (0-based slot/variable)
staging(w::CoFluxWindow, slot, var) =
    unsafe_wrap(Array, ccall((:cf_window_host_buffer, libcoflux), Ptr{Float32}, (Ptr{Cvoid}, Int32, Int32), w.ptr, slot, var),
                (w.nsx, w.nsy))
wait_slot!(w::CoFluxWindow, slot) =
    check(w.backend.ctx, ccall((:cf_window_wait_slot, libcoflux), Cint, (Ptr{Cvoid}, Int32), w.ptr, slot))
# time_index = the monotone snapshot counter ⌊t/Δt⌋ (slot = counter mod nslots), not the index inside a repeat-year record
commit!(w::CoFluxWindow, slot, time_index) =
    check(w.backend.ctx, ccall((:cf_window_commit, libcoflux), Cint, (Ptr{Cvoid}, Int32, Int64), w.ptr, slot, time_index))
resident(w::CoFluxWindow, time_index) = ccall((:cf_window_find, libcoflux), Cint, (Ptr{Cvoid}, Int64), w.ptr, time_index) >= 0
function source(w::CoFluxWindow, n₁, n₂, ñ)          # → CfAtmosSource for update_state!
    src = Ref{CfAtmosSource}()
    check(w.backend.ctx, ccall((:cf_window_source, libcoflux), Cint, (Ptr{Cvoid}, Int64, Int64, Float64, Ref{CfAtmosSource}),
                               w.ptr, n₁, n₂, ñ, src))
    return src[]
end

# ---- latitude-slab halo rows over RCCL (Distributed(GPU(), partition = Partition(1, R))) -------
comm_unique_id() = (id = zeros(UInt8, 128); ccall((:cf_comm_unique_id, libcoflux), Cint, (Ptr{UInt8},), id); id)
# (nranks, rank, device) as RCCL itself reports them for this backend's communicator — beside MPI.Comm_size in a launcher's log
function comm_count(b)
    n, r, d = Ref{Cint}(0), Ref{Cint}(0), Ref{Cint}(0)
    check(b.ctx, ccall((:cf_comm_count, libcoflux), Cint, (Ptr{Cvoid}, Ref{Cint}, Ref{Cint}, Ref{Cint}), b.ctx, n, r, d))
    return (n[], r[], d[])
end
comm_init!(b, id::Vector{UInt8}, rank, nranks) =   # `id` is MPI.bcast from rank 0
    check(b.ctx, ccall((:cf_comm_init, libcoflux), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), b.ctx, id, rank, nranks))
halo_exchange_rows!(b, fields::Vector{Ptr{Float64}}, rows = 2) =   # rows = ring + 1: the ring row reads v[j+1]
    check(b.ctx, ccall((:cf_halo_exchange_rows, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Cint, Cint),
                       b.ctx, fields, length(fields), rows))

# ---- peer-direct halo rows (HIP IPC mailboxes; handles travel by MPI.Allgather) and the tripolar fold ----------------
peer_halo_export!(b, max_fields = 4, max_rows = 2) = (h = zeros(UInt8, 64);
    check(b.ctx, ccall((:cf_peer_halo_export, libcoflux), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}), b.ctx, max_fields, max_rows, h)); h)
peer_halo_connect!(b, south::Union{Nothing, Vector{UInt8}}, north::Union{Nothing, Vector{UInt8}}, rank, nranks) =
    check(b.ctx, ccall((:cf_peer_halo_connect, libcoflux), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Ptr{UInt8}, Cint, Cint), b.ctx,
                       isnothing(south) ? C_NULL : south, isnothing(north) ? C_NULL : north, rank, nranks))
halo_exchange_rows_peer!(b, fields::Vector{Ptr{Float64}}, rows = 2) =
    check(b.ctx, ccall((:cf_halo_exchange_rows_peer, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Cint, Cint),
                       b.ctx, fields, length(fields), rows))
# TripolarGrid: the last rank's north halo (T, S centres; u x-faces, v y-faces; vectors change sign) — one_degree_tripolar.jl:48-51
fold_north_halo!(b, fields::Vector{Ptr{Float64}}, locations::Vector{Cint}, signs::Vector{Float64}, rows = 2) =
    check(b.ctx, ccall((:cf_fold_north_halo, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Cint}, Ptr{Float64}, Cint, Cint),
                       b.ctx, fields, locations, signs, length(fields), rows))

# ---- pipelined update_state!: the prescribed atmosphere of the NEXT step on the auxiliary stream ---------------------
prefetch_atmosphere_state!(b, src_next::CfAtmosSource, w::CfInterpWeights, out::CfExchangeFields) =
    check(b.ctx, ccall((:cf_prefetch_atmosphere_state, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfAtmosSource}, Ref{CfInterpWeights}, Ref{CfExchangeFields}), b.ctx, src_next, w, out))

# on leaving the stepping loop: no later update_state! may mistake a requested-ahead state for the one it asks for
discard_prefetched_atmosphere_state!(b) =
    check(b.ctx, ccall((:cf_discard_prefetched_atmosphere_state, libcoflux), Cint, (Ptr{Cvoid},), b.ctx))

# the solver's schedule for a wet mask, built ahead of the first step (optional: the first call builds it otherwise)
ensure_chunk_table!(b, mask::Ptr{Cvoid}) =
    check(b.ctx, ccall((:cf_ensure_chunk_table, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), b.ctx, mask))

# ---- run!(simulation) of a prescribed-ocean model inside the library (bench / offline forcing runs) -------------------
struct CfRunSchedule
    struct_size::Int32; n_ocean_states::Int32
    ocean_states::Ptr{CfOceanSurface}
    n_atmos_sets::Int32; pipeline::Int32
    atmos::Ptr{CfExchangeFields}
    first_level::Int32; halo_backend::Int32; halo_rows::Int32; fold_north::Int32
    time_fraction::Float64; time_fraction_increment::Float64
end
time_steps!(b, first_step, nsteps, sched::CfRunSchedule, src, w, fluxes, ice, net) =
    check(b.ctx, ccall((:cf_time_steps, libcoflux), Cint,
                       (Ptr{Cvoid}, Int64, Cint, Ref{CfRunSchedule}, Ref{CfAtmosSource}, Ref{CfInterpWeights}, Ref{CfInterfaceFluxes},
                        Ptr{CfSeaIceFields}, Ref{CfNetOceanFluxes}), b.ctx, first_step, nsteps, sched, src, w, fluxes, ice, net))

# ---- SeaIceAlbedo(hi, hs, Ts) (atmosphere.jl:30-44) and compute_sea_ice_ocean_fluxes! (omip_simulation.jl:71-77) ------
mutable struct CfSeaIceAlbedoParams
    struct_size::Int32; reserved::Int32
    ice_visible::Float64; ice_near_infrared::Float64; snow_visible::Float64; snow_near_infrared::Float64; ocean_albedo::Float64
    reference_thickness::Float64; melt_temperature_range::Float64; ice_melt_change::Float64
    snow_melt_change_visible::Float64; snow_melt_change_near_infrared::Float64; snow_patch_thickness::Float64
    visible_fraction::Float64; melting_temperature::Float64
    CfSeaIceAlbedoParams() = new()
end
default_sea_ice_albedo_params() = (p = CfSeaIceAlbedoParams();
    ccall((:cf_default_sea_ice_albedo_params, libcoflux), Cint, (Ref{CfSeaIceAlbedoParams},), p); p)
set_sea_ice_albedo!(b, p::CfSeaIceAlbedoParams) =
    check(b.ctx, ccall((:cf_set_sea_ice_albedo, libcoflux), Cint, (Ptr{Cvoid}, Ref{CfSeaIceAlbedoParams}), b.ctx, p))
compute_sea_ice_albedo!(b, p, hi, hs, Ts, out) =
    check(b.ctx, ccall((:cf_compute_sea_ice_albedo, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfSeaIceAlbedoParams}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), b.ctx, p, hi, hs, Ts, out))
mutable struct CfIceOceanParams   # ThreeEquationHeatFlux(; friction_velocity = MomentumBasedFrictionVelocity())
    struct_size::Int32; reserved::Int32
    heat_transfer_coefficient::Float64; salt_transfer_coefficient::Float64; minimum_friction_velocity::Float64
    ice_density::Float64; latent_heat_of_fusion::Float64; ice_salinity::Float64; liquidus_slope::Float64
    top_cell_thickness::Float64; time_step::Float64
    CfIceOceanParams() = new()
end
struct CfIceOceanFluxes; interface_heat::Ptr{Float64}; salt_flux::Ptr{Float64}; frazil_heat::Ptr{Float64}; friction_velocity::Ptr{Float64}; end
default_ice_ocean_params() = (p = CfIceOceanParams(); ccall((:cf_default_ice_ocean_params, libcoflux), Cint, (Ref{CfIceOceanParams},), p); p)
compute_sea_ice_ocean_fluxes!(b, p::CfIceOceanParams, ocean::CfOceanSurface, concentration, x_stress, y_stress, out::CfIceOceanFluxes) =
    check(b.ctx, ccall((:cf_compute_sea_ice_ocean_fluxes, libcoflux), Cint,
                       (Ptr{Cvoid}, Ref{CfIceOceanParams}, Ref{CfOceanSurface}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{CfIceOceanFluxes}),
                       b.ctx, p, ocean, concentration, x_stress, y_stress, out))

# ---- JRA55PrescribedLand (atmosphere.jl:46) and the MultipleFluxes additional flux (omip_simulation.jl:175-206, 507-523) --
struct CfLandSource
    friver::Ptr{Float32}; licalvf::Ptr{Float32}
    ns_x::Int32; ns_y::Int32; n_levels::Int32; level1::Int32; level2::Int32; reserved::Int32
    time_fraction::Float64
end
interpolate_land_freshwater!(b, src::CfLandSource, w::CfInterpWeights, out) =
    check(b.ctx, ccall((:cf_interpolate_land_freshwater, libcoflux), Cint, (Ptr{Cvoid}, Ref{CfLandSource}, Ref{CfInterpWeights}, Ptr{Float64}),
                       b.ctx, src, w, out))
set_land_freshwater!(b, field) = check(b.ctx, ccall((:cf_set_land_freshwater, libcoflux), Cint, (Ptr{Cvoid}, Ptr{Float64}), b.ctx, field))
materialize_salinity_restoring!(b, piston_velocity, target, ocean::CfOceanSurface, buffer) =
    check(b.ctx, ccall((:cf_materialize_salinity_restoring, libcoflux), Cint,
                       (Ptr{Cvoid}, Float64, Ptr{Float64}, Ref{CfOceanSurface}, Ptr{Float64}), b.ctx, piston_velocity, target, ocean, buffer))

end # module
