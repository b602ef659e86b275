/*
 * coflux.h — C ABI of libcoflux: the MI355X-native surface-flux hot path of
 * ClimaOcean's OceanSeaIceModel (update_state! of the coupled model).
 *
 * Every entry point replaces one Julia function that the reference reaches through
 * multiple dispatch (there is no FFI in the reference; the seam is described in
 * SURVEY.md §8b).  All `path:line` citations are relative to the reference tree
 * (CliMA/ClimaOcean.jl v0.10.0).  The arithmetic itself lives in the un-vendored
 * NumericalEarth.jl package (Project.toml:21,31-32), so each entry cites the in-tree
 * call/config site that pins its signature and parameters.
 *
 * Conventions
 *  - plain C types only; no torch / HIP types in signatures (a stream is a `void*`
 *    holding a hipStream_t; NULL = the context's own stream).
 *  - every pointer named `d_*` or living inside a `cf_*_fields` struct is a DEVICE pointer.
 *  - ocean-grid 2-D arrays are column-major with halos, `i` fastest (Oceananigans
 *    `parent(field)` layout, omip_simulation.jl:184, KPP/kpp_surface_forcing.jl:49):
 *        element (i, j), 0-based interior index, lives at
 *        ptr[(j + hy) * (nx + 2*hx) + (i + hx)].
 *    For 3-D ocean fields pass the pointer to the k = Nz level slab.
 *  - all entry points return 0 on success, <0 on error; cf_last_error() explains.
 *    Nothing throws across the ABI.
 *  - a context is single-threaded; different contexts may be used from different threads.
 *    One context per GPU (one MPI rank / one process per GPU, launch.sh:229-232).
 */
#ifndef COFLUX_H
#define COFLUX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CF_ABI_VERSION 3 /* 2: cf_flux_params.shear_gustiness_coefficient; 3: CF_OPT_MAX_BLOCKS, CF_OPT_PROFILE_STRIDE, CF_OPT_FUSED_INTERP,
                          * CF_SOLVER_TABLES_R2(_OUTER) and the 768-thread geometry (CF_OPT_AO_CHUNK = 3072) retired; CF_OPT_AO_CHUNK and
                          * CF_OPT_INTERP_TILE_CAP are experiment options (COFLUX_EXPERIMENTS=1) */

/* status codes */
#define CF_OK 0
#define CF_ERR_INVALID (-1)   /* bad argument / inconsistent config */
#define CF_ERR_HIP (-2)       /* HIP runtime error (message has file:line) */
#define CF_ERR_NODEVICE (-3)  /* no usable gfx950 device */
#define CF_ERR_COMM (-4)      /* RCCL error */

typedef struct cf_ctx cf_ctx;

/* ------------------------------------------------------------------------------------------
 * Grid / launch description.
 * Reference: LatitudeLongitudeGrid(arch; size=(1440,560,10), halo=(7,7,7)) README.md:56-61;
 * flux kernels are launched on 0:Nx+1 × 0:Ny+1 (one halo ring) in the reference
 * [UPSTREAM-RECALL interface_kernel_parameters] ⇒ ring = 1; ring = 0 computes the interior only.
 * ---------------------------------------------------------------------------------------- */
typedef struct cf_grid {
    int32_t nx, ny;      /* interior surface cells (local slab when sharded)            */
    int32_t hx, hy;      /* halo widths of every ocean-grid array handed to the library */
    int32_t ring;        /* 0 or 1: extra ring of cells the kernels also compute        */
    int32_t reserved;
} cf_grid;

/* ------------------------------------------------------------------------------------------
 * Flux formulation parameter block.
 * Mirrors SimilarityTheoryFluxes(FT; similarity_form, minimum_gustiness, gustiness_parameter,
 * stability_functions, momentum_roughness_length, temperature_roughness_length,
 * water_vapor_roughness_length) — omip_simulation.jl:42-49 (":corrected"), :63-69 (sea ice),
 * :106-113 (":ncar" sea ice), and the defaults used at README.md:75 / omip_simulation.jl:128-132
 * (":default" = Edson/COARE with constant Charnock 0.02, omip_simulation.jl:263).
 * ---------------------------------------------------------------------------------------- */

/* similarity_form */
#define CF_SIMILARITY_LOGARITHMIC 0        /* log(h/l) - psi(h/L) + psi(l/L)                */
#define CF_SIMILARITY_COARE_LOGARITHMIC 1  /* COARELogarithmicSimilarityProfile(): no psi(l/L), omip_simulation.jl:36,43 */

/* stability_functions */
#define CF_STABILITY_EDSON2013 0     /* default atmosphere–ocean (docs/climaocean.bib:1-10)            */
#define CF_STABILITY_SHEBA 1         /* atmosphere_sea_ice_stability_functions, omip_simulation.jl:56,64 */
#define CF_STABILITY_LARGE_YEAGER 2  /* large_yeager_stability_functions, omip_simulation.jl:96,107      */

/* momentum roughness */
#define CF_ROUGHNESS_CONSTANT 0        /* FT(5e-4), omip_simulation.jl:67                          */
#define CF_ROUGHNESS_CHARNOCK 1        /* MomentumRoughnessLength, constant Charnock (0.02)        */
#define CF_ROUGHNESS_WIND_CHARNOCK 2   /* WindDependentWaveFormulation, Edson 2013 eq. 13, :35,46 */

/* scalar roughness */
#define CF_SCALAR_ROUGHNESS_CONSTANT 0  /* FT(5e-5), omip_simulation.jl:68-69 */
#define CF_SCALAR_ROUGHNESS_REYNOLDS 1  /* ScalarRoughnessLength(FT; air_kinematic_viscosity), :48-49 */

/* air viscosity */
#define CF_VISCOSITY_CONSTANT 0
#define CF_VISCOSITY_TEMPERATURE_DEPENDENT 1 /* TemperatureDependentAirViscosity(FT), :41 */

/* flux formulation */
#define CF_FORMULATION_SIMILARITY 0      /* SimilarityTheoryFluxes (iteration on u★, θ★, q★ with roughness lengths) */
#define CF_FORMULATION_LARGE_YEAGER 1    /* CoefficientBasedFluxes + LargeYeagerTransferCoefficients, omip_simulation.jl:86-89 */

/* solver stop criteria */
#define CF_STOP_CONVERGENCE 0  /* default: |Δu★|+|Δθ★|+|Δq★| < tolerance or iteration ≥ maxiter */
#define CF_STOP_FIXED 1        /* FixedIterations(5), omip_simulation.jl:22,89                  */

/* velocity difference */
#define CF_VELOCITY_RELATIVE 0 /* RelativeVelocity(): Δu = u_atm − u_ocean, omip_simulation.jl:135 */
#define CF_VELOCITY_WIND 1     /* WindVelocity():     Δu = u_atm,           omip_simulation.jl:136 */

/* wet mask encoding */
#define CF_MASK_NONE 0
#define CF_MASK_U8 1            /* uint8 per cell, 1 = wet                                          */
#define CF_MASK_BOTTOM_HEIGHT 2 /* f64 bottom height; cell is land when z_surface_center <= zb      */

/* ocean albedo */
#define CF_ALBEDO_CONSTANT 0            /* SurfaceRadiationProperties(0.06, 1.00), atmosphere.jl:43 */
#define CF_ALBEDO_LATITUDE_DEPENDENT 1  /* α = a_diffuse − a_direct·cos(2φ) (Large & Yeager 2009)    */

typedef struct cf_roughness {
    int32_t kind;                 /* CF_ROUGHNESS_* (momentum) or CF_SCALAR_ROUGHNESS_* (scalars) */
    int32_t viscosity_kind;       /* CF_VISCOSITY_*                                              */
    double constant_length;       /* used when kind == *_CONSTANT                                 */
    double maximum_length;        /* cap (momentum 1.0; scalars 1.6e-4, COARE 3.6)                */
    double charnock;              /* Charnock parameter (constant form) / floor of α (wind-dependent form) */
    double laminar;               /* laminar parameter 0.11                                       */
    double wind_a1, wind_a2, wind_umax; /* wind-dependent α = max(charnock, a1·min(U,umax) + a2)  */
    double reynolds_A, reynolds_b;      /* scalar roughness ℓ = A / R★^b                            */
    double viscosity[4];          /* ν = c0 + c1 T' + c2 T'^2 + c3 T'^3 (T' in °C); c0 only if constant */
} cf_roughness;

typedef struct cf_thermodynamics {
    double gas_constant;          /* 8.3144598 */
    double dry_air_molar_mass;    /* 0.02897   */
    double water_molar_mass;      /* 0.018015  */
    double kappa_d;               /* 2/7       */
    double cp_v, cp_l, cp_i;      /* 1859, 4181, 2100 */
    double LH_v0, LH_s0;          /* 2500800, 2834400 */
    double T_0, T_triple, p_triple; /* 273.16, 273.16, 611.657 */
    double T_freeze, T_icenuc;    /* 273.15, 233 */
    double pow_icenuc;            /* 1 */
} cf_thermodynamics;

typedef struct cf_seawater {
    double water_molar_mass;      /* 18.02 (g/mol; used only as a ratio) */
    double constituent_molar_mass[4];    /* Cl 35.45, Na 22.99, SO4 96.06, Mg 24.31 */
    double constituent_mass_fraction[4]; /*    0.56,     0.31,      0.08,     0.05  */
} cf_seawater;

typedef struct cf_flux_params {
    int32_t struct_size;          /* sizeof(cf_flux_params), checked by the library */
    int32_t abi_version;          /* CF_ABI_VERSION */

    /* SimilarityTheoryFluxes */
    int32_t similarity_form;      /* CF_SIMILARITY_* */
    int32_t stability_functions;  /* CF_STABILITY_*  */
    int32_t stop_kind;            /* CF_STOP_*       */
    int32_t maxiter;              /* convergence: cap (100); fixed: the iteration count */
    int32_t velocity_difference;  /* CF_VELOCITY_*   */
    int32_t mask_kind;            /* CF_MASK_*       */
    double tolerance;             /* 1e-8 */
    double von_karman;            /* 0.4  */
    double gustiness_parameter;   /* β, 1 (0 in :ncar sea ice, omip_simulation.jl:109) */
    double minimum_gustiness;     /* 0.5 ocean :40,44; 0.2 ice :66 */
    double shear_gustiness_coefficient; /* c of the shear-aware gustiness (Mahrt & Sun 1995 / Edson 2013;
                                     experiments/OMIPSimulations/scripts/launch.sh:67-72,350, `:shear_aware`):
                                     0 (default) = off, the wind-speed scale is U² = |Δu|² + max((β w★)², U_G,min²);
                                     c > 0 (launch.sh: 0.04) = U² = |Δu|² + U_G², U_G² = (β w★)² + (c |Δu|)² + U_G,min²
                                     — the convective, shear and background terms ADD, and the shear term raises the
                                     gust at every wind speed.  The reference's own build_coupled_model rejects the
                                     symbol (omip_simulation.jl:160); the formula is the one its launcher states. */
    double similarity_profile_floor; /* guard: log(h/ℓ) − ψ(h/L) [+ψ(ℓ/L)] is floored at this value (1.0),
                                        i.e. transfer coefficients are capped at κ/floor.  Only the
                                        pathological first iterates from the 1e-4 initial guess (ζ ≈ −10⁵)
                                        ever reach it; converged states have 4 ≲ profile ≲ 20. */
    cf_roughness momentum_roughness;
    cf_roughness temperature_roughness;
    cf_roughness water_vapor_roughness;

    /* atmosphere properties (PrescribedAtmosphere) */
    double reference_height;      /* 10 m  */
    double boundary_layer_height; /* 600 m */
    double gravitational_acceleration; /* 9.81 */
    cf_thermodynamics thermo;
    cf_seawater seawater;

    /* ocean properties */
    double ocean_reference_density;   /* 1026,            visualize/common.jl:17 */
    double ocean_heat_capacity;       /* 3991.86795711963, visualize/common.jl:18 */
    double ocean_freshwater_density;  /* 1000: converts P, E mass fluxes to volume fluxes */
    double ocean_temperature_offset;  /* 273.15: ocean T is in °C                 */
    double ocean_minimum_salinity;    /* omip_simulation.jl:125,131,314; launch.sh:74-78 */
    double ocean_surface_z;           /* z of the top cell centre, for CF_MASK_BOTTOM_HEIGHT */

    /* radiation: SurfaceRadiationProperties(albedo, emissivity), atmosphere.jl:41-44 */
    int32_t ocean_albedo_kind;    /* CF_ALBEDO_* */
    int32_t penetrating_shortwave; /* 1: transmitted SW goes to the separate surface_flux field
                                      (KPP/kpp_surface_forcing.jl:47-51) and not into JT */
    double ocean_albedo;          /* 0.06 */
    double ocean_albedo_diffuse;  /* 0.069 */
    double ocean_albedo_direct;   /* 0.011 */
    double ocean_emissivity;      /* 1.0  */
    double stefan_boltzmann;      /* 5.67e-8 */

    /* CoefficientBasedFluxes(FT; transfer_coefficients = LargeYeagerTransferCoefficients(FT),
     * solver_stop_criteria = FixedIterations(5)) — omip_simulation.jl:79-89 (":ncar", OMIP-2): the
     * iteration runs on the transfer coefficients (Cd, Ch, Ce), not on roughness lengths.            */
    int32_t flux_formulation;     /* CF_FORMULATION_* */
    int32_t reserved1;
    double ly_minimum_wind;       /* 0.5 m/s floor on |Δu| (NCAR/CORE convention)                       */
    double ly_zeta_bound;         /* |ζ| ≤ 10                                                           */
    double ly_cd[4];              /* 10³·Cd_N10 = c0/U + c1 + c2·U + c3·U⁶: 2.7, 0.142, 0.0764, −3.14807e-10 */
    double ly_high_wind;          /* U ≥ 33 m/s ⇒ 10³·Cd_N10 = ly_cd_high (Large & Yeager 2009)         */
    double ly_cd_high;            /* 2.34 */
    double ly_ce;                 /* 10³·Ce_N10 = 34.6·√Cd_N10 */
    double ly_ch_stable;          /* 10³·Ch_N10 = 18.0·√Cd_N10 (ζ > 0) */
    double ly_ch_unstable;        /* 32.7·√Cd_N10 (ζ ≤ 0) */
} cf_flux_params;

/* Fill `p` with the ":default" configuration (omip_simulation.jl:128-132, :263). */
int cf_default_flux_params(cf_flux_params* p);

/* ------------------------------------------------------------------------------------------
 * Field bundles (device pointers, ocean-grid layout unless noted)
 * ---------------------------------------------------------------------------------------- */

/* Ocean surface state read by compute_atmosphere_ocean_fluxes! (k = Nz level): T (°C), S (g/kg)
 * at centres, u at x-faces, v at y-faces (src/ClimaOcean.jl:29 imports the ℑ operators).      */
typedef struct cf_ocean_surface {
    const double* T;
    const double* S;
    const double* u;
    const double* v;
    const void* mask;      /* per cf_flux_params.mask_kind; may be NULL for CF_MASK_NONE */
} cf_ocean_surface;

/* Atmosphere state on the exchange (ocean) grid: output of interpolate_atmosphere_state!,
 * input of the flux solver and of the net-flux assembly.                                       */
typedef struct cf_exchange_fields {
    double* u;   /* m/s, rotated to the grid-intrinsic frame */
    double* v;
    double* T;   /* K      (JRA55 tas)  */
    double* p;   /* Pa     (psl)        */
    double* q;   /* kg/kg  (huss)       */
    double* Qs;  /* W/m²   (rsds)       */
    double* Ql;  /* W/m²   (rlds)       */
    double* Mp;  /* kg/m²/s (prra+prsn) */
} cf_exchange_fields;

/* interface (turbulent) fluxes: model.interfaces.atmosphere_ocean_interface.fluxes.*,
 * omip_diagnostics.jl:81-82; positive = upward (ocean loses).                                  */
typedef struct cf_interface_fluxes {
    double* sensible_heat;  /* Qc  W/m²     */
    double* latent_heat;    /* Qv  W/m²     */
    double* water_vapor;    /* Fv  kg/m²/s  */
    double* x_momentum;     /* ρτx N/m²     */
    double* y_momentum;     /* ρτy N/m²     */
    double* temperature;    /* interface temperature Ts, ocean units (°C) */
    double* friction_velocity;    /* optional (may be NULL): u★ */
    double* temperature_scale;    /* optional: θ★ */
    double* humidity_scale;       /* optional: q★ */
    int32_t* iterations;          /* optional: iteration count per cell (diagnostic) */
} cf_interface_fluxes;

/* sea-ice inputs to the partition (atmosphere.jl:34-39, src/ClimaOcean.jl:62-63); all may be
 * NULL ⇒ ice-free.                                                                             */
typedef struct cf_sea_ice_fields {
    const double* concentration;   /* ℵ                        */
    const double* interface_heat;  /* Qio  W/m² (ice→ocean)    */
    const double* salt_flux;       /* Jˢio g/kg m/s            */
    const double* x_stress;        /* ice–ocean stress at u-faces, kinematic m²/s² */
    const double* y_stress;
} cf_sea_ice_fields;

/* net ocean fluxes: model.interfaces.net_fluxes.ocean.{u,v,T,S}, omip_diagnostics.jl:77-80.    */
typedef struct cf_net_ocean_fluxes {
    double* u;   /* τx kinematic m²/s² at u-faces (KPP/kpp_surface_forcing.jl:18-22) */
    double* v;   /* τy kinematic at v-faces                                           */
    double* T;   /* JT  K m/s   (hfds = JT·ρ·cp, visualize/cache.jl:359-361)          */
    double* S;   /* JS  g/kg m/s                                                      */
    double* shortwave_surface_flux; /* radiation.surface_flux (KPP/kpp_surface_forcing.jl:47-51); may be NULL */
    double* upwelling_longwave;     /* optional diagnostics, may be NULL */
    double* downwelling_longwave;
    double* downwelling_shortwave;
} cf_net_ocean_fluxes;

/* JRA55 source window on its native 640×320 grid (launch.sh:86-87), Float32, NO halos:
 *   value(var, level, js, is) = d_data[var][(level * ns_y + js) * ns_x + is].
 * Variable order follows jra55_data_staging.jl:8.                                               */
#define CF_JRA55_TAS 0
#define CF_JRA55_HUSS 1
#define CF_JRA55_PSL 2
#define CF_JRA55_UAS 3
#define CF_JRA55_VAS 4
#define CF_JRA55_RLDS 5
#define CF_JRA55_RSDS 6
#define CF_JRA55_PRRA 7
#define CF_JRA55_PRSN 8
#define CF_JRA55_NVARS 9

typedef struct cf_atmos_source {
    const float* data[CF_JRA55_NVARS]; /* device pointers, each [n_levels][ns_y][ns_x] */
    int32_t ns_x, ns_y;     /* 640, 320 */
    int32_t n_levels;       /* time indices in memory (atmosphere.jl:26, backend_size)  */
    int32_t level1, level2; /* the two bracketing snapshots n₁, n₂ (memory indices)     */
    double time_fraction;   /* ñ ∈ [0,1): value = ψ₂·ñ + ψ₁·(1−ñ)                       */
} cf_atmos_source;

/* Fractional source indices of every target cell (the reference precomputes the same pair,
 * `space_fractional_indices`).  Either separable (lat-lon → lat-lon: fi[i], fj[j]) or general
 * (tripolar: 2-D arrays in ocean-grid layout).  Optional rotation to the grid-intrinsic frame.  */
typedef struct cf_interp_weights {
    int32_t separable;      /* 1: fi has nx+2*hx entries, fj has ny+2*hy entries (halo-inclusive) */
    int32_t reserved;
    const double* fi;       /* 0-based fractional index along source x (periodic)  */
    const double* fj;       /* 0-based fractional index along source y (clamped)   */
    const double* cos_rot;  /* ocean-grid 2-D arrays or NULL (no rotation)         */
    const double* sin_rot;
    const double* latitude; /* φ (deg) per row (ny+2*hy) if separable else 2-D; needed only for
                               CF_ALBEDO_LATITUDE_DEPENDENT; may be NULL            */
} cf_interp_weights;

/* ------------------------------------------------------------------------------------------
 * Lifecycle
 * ---------------------------------------------------------------------------------------- */
int cf_version(void);
/* The first 16 hex digits of the sha256 over the library's sources (csrc/ *.hip, *.cpp, *.hpp, *.h and this header, in sorted
 * order) as they were when it was built: a host binding compares it with the tree it ships with and refuses a stale build. */
const char* cf_build_stamp(void);
/* Creates a context bound to HIP device `device`.  Fails (CF_ERR_NODEVICE) when there is no GPU. */
int cf_create(cf_ctx** out, int device, const cf_grid* grid, const cf_flux_params* params);
int cf_destroy(cf_ctx* ctx);
const char* cf_last_error(const cf_ctx* ctx); /* ctx may be NULL: last error of the calling thread */
int cf_set_flux_params(cf_ctx* ctx, const cf_flux_params* params);
/* hip_stream: a hipStream_t; NULL ⇒ the library-owned non-blocking stream; CF_STREAM_LEGACY ⇒ the
 * legacy default (null) stream, which is what torch's default stream is (its handle is 0).        */
#define CF_STREAM_LEGACY ((void*)1) /* == hipStreamLegacy */
int cf_set_stream(cf_ctx* ctx, void* hip_stream);

/* Options (cf_set_option).  None of them changes what is computed beyond the stated tolerance.  TEN are part of the drop-in
 * surface: CF_OPT_SOLVER, _TRIP_HINTS, _FUSED_NET, _ICE_ORBIT_SHORTCUT, _MERGED_PREFETCH, _SOLVER_PATH, _CERTIFIED_BUDGET,
 * _ICE_FREE_CELLS, _LATENCY_LAYOUT, _HALO_IN_SOLVER_LAUNCH.  Two more are EXPERIMENT options, accepted only in a process started with
 * COFLUX_EXPERIMENTS=1 (measurements, and the test-suite's schedule-invariance checks): CF_OPT_INTERP_TILE_CAP, CF_OPT_AO_CHUNK.
 * Numbers 2, 5 and 8 were CF_OPT_MAX_BLOCKS, _PROFILE_STRIDE and _FUSED_INTERP (retired in ABI version 3: cf_set_option
 * answers CF_ERR_INVALID).                                                                                              */
#define CF_OPT_SOLVER 0           /* CF_SOLVER_*                                                   */
#define CF_OPT_INTERP_TILE_CAP 1  /* EXPERIMENT option.  Source nodes per variable in a wave's LDS JRA55 tile (128; 16…224: 4 waves × 9 variables × cap × 8 B of LDS), or 0:
                                     the LDS-free one-cell-per-lane gather kernel (≤ 56 VGPRs: small enough to run
                                     beside the resident solver workgroups from a second stream)            */
#define CF_OPT_TRIP_HINTS 3       /* order each chunk's cells by the iteration count of the previous call (batches of equal trip
                                   * counts): 0 off, 1 on, 2 (default) automatic = on for the atmosphere–sea-ice solve, whose counts span
                                   * 10…100, off for the ocean solve, where with forcing that evolves from call to call the scattered
                                   * memory access of a sorted batch costs more than the one-step-old order saves (0.085 vs 0.073 ms);
                                   * 3 = as 1, but the round-3 ocean kernel sorts each QUARTER of a chunk's list separately (a batch
                                   * stays within a quarter of the chunk's cell range: 16 instead of 64 lines per access) — measured
                                   * −2 % on the solver alone, +0.5 % on the fused cf_update_state: not the default either          */
#define CF_OPT_AO_CHUNK 4         /* EXPERIMENT option.  Wet cells per workgroup of the flux solver: 0 = automatic (arrival layers of
                                     1024 / 768 / 512 on a surface that fills the device, 256 on a slab that does not), 256 … 1280 = that
                                     size for every workgroup.  Results do not depend on it, bit for bit (tested).               */
#define CF_OPT_FUSED_NET 6        /* cf_update_state computes the cell-local net ocean fluxes in the solver's epilogue and follows
                                   * with a face-stress kernel instead of the three-launch sequence (bitwise the same results):
                                   * 0 never, 1 whenever the configuration allows it (constant ocean albedo), 2 (default) when the
                                   * round-3 ocean kernel runs — its batches are in index order, so the epilogue's nine extra
                                   * accesses per cell are coalesced: update_state 0.107 → 0.096 ms — or CoefficientBasedFluxes
                                   * (fixed trip count: index order too; 0.072 → 0.063 ms); in round 2's trip-sorted kernels
                                   * the same accesses were scattered and cost more than the net-flux kernel they save.        */
#define CF_OPT_MERGED_PREFETCH 9   /* 1: an interpolation requested ahead (cf_prefetch_atmosphere_state, cf_time_steps with pipelining) is
                                   * launched TOGETHER with the current step's face stresses — one kernel on the context's stream whose
                                   * workgroups do one or the other (same arithmetic, same bits) — instead of on the auxiliary stream: a
                                   * step is then two launches (solver; stresses + next interpolation).  Pays where launch boundaries
                                   * dominate (a latitude slab of a strongly scaled run); needs the fused net fluxes and the tiled
                                   * interpolation.  0 (default): auxiliary stream.
                                   * 2: the requested interpolation becomes extra workgroups BEHIND the solver's in the solver launch
                                   * (round-3 ocean kernel and the CoefficientBasedFluxes kernel): they take the slots the solver's
                                   * workgroups free as they retire; with a sea-ice formulation (cf_update_state_sea_ice) the riders
                                   * sit in the interface solve's launch — the longest of the step — and the OCEAN SOLVE is one of
                                   * them: one dispatch order holds its first arrival layer, the interface solve's workgroups, the
                                   * rest of its own and the interpolation's; the face stresses follow as a launch of their own
                                   * (two solver launches on two queues do not overlap on this device; workgroups of one launch do).
                                   * A context in this mode cuts its solver chunks for it (equal chunks per CU where the surface fills
                                   * one dispatch generation, so that workgroups retire staggered): meant for stepping loops that
                                   * request every next state; a lone cf_update_state without a request runs ≈ 5 µs slower on that
                                   * plan than on the default one.  Results are the same bits in every mode — with one aliasing caveat:
                                   * in this mode cf_update_state_sea_ice computes compute_net_sea_ice_fluxes! in the interface solve's
                                   * epilogue with the albedo that solve used; a caller that (a) uses the SeaIceAlbedo(hi, hs, Ts) scheme
                                   * and (b) passes the SAME buffer as cf_sea_ice_state.top_temperature and as the interface
                                   * temperature output gets, from the separate launches of the other modes, an albedo re-evaluated at
                                   * the NEW skin temperature for the net fluxes.                                                    */
#define CF_OPT_ICE_ORBIT_SHORTCUT 7 /* 1 (default): the atmosphere–sea-ice iteration stops as soon as its state repeats the state of two
                                     iterations ago bit for bit — an exact period-2 orbit, where the skin-temperature balance does not
                                     contract — and returns the iterate the remaining steps up to maxiter would end on (the same
                                     bits, tested); 0: iterate to maxiter */
#define CF_OPT_SOLVER_PATH 10      /* how the SimilarityTheoryFluxes fixed point of compute_atmosphere_ocean_fluxes! is reached
                                   * (omip_simulation.jl:42-49):
                                   * CF_SOLVER_PATH_EXACT (0, default): the reference's own iteration — same first guess (1e-4), same
                                   * update order, same stop rule, identical trip counts, results within 7e-12 of the oracle.
                                   * CF_SOLVER_PATH_CERTIFIED (1): the same map in FP64 from a neutral-profile first guess with
                                   * Anderson(2) steps (4–6 evaluations per cell instead of 8–20), and a per-cell CERTIFICATE that
                                   * the fixed point lies within CF_OPT_CERTIFIED_BUDGET of where the reference's stop rule would
                                   * have left its iterate, in the metric |Δ flux| ≤ budget · max(|flux|, floor) with the floors
                                   * 1 W m⁻² (sensible, latent heat), 1e-6 kg m⁻² s⁻¹ (water vapour), 1e-3 N m⁻² (ρτx, ρτy) — and,
                                   * because the net salinity flux J_S = −S (F_v − M_p)/ρ_f can cancel to nothing, the vapour flux also
                                   * within budget · max(|F_v − M_p|, 1e-7 ρ_f / S) of the exact path's (M_p: the interpolated rain + snow),
                                   * which holds J_S of open water to budget · max(|J_S|, 1e-7 m s⁻¹ psu).  The
                                   * bound is the worst case over every last drift the stop rule admits, from the map's Jacobian at
                                   * the fixed point (csrc/coflux_certified.hpp).  Cells that cannot be certified (≈ 2 % at the default
                                   * budget: near-neutral and dead-calm cells, where an absolute drift tolerance leaves the stopped
                                   * iterate loosely determined — half of them — and cells whose evaporation all but cancels their
                                   * precipitation) are solved by the exact path inside the same launch.
                                   * What kind of statement this is: an ESTIMATE with a measured safety factor, not a proof.
                                   * The Jacobian is a two-pair secant estimate in FP32; the bound presumes the linear regime of the map
                                   * (spectral radius of the estimate < 0.6 is required); the safety factor 1.25 is the measured ± 4 %
                                   * accuracy of that estimate (scratch/certified_study.py); and the accepted Anderson-extrapolated state is
                                   * not re-evaluated — "within 2e-8 of the fixed point" rests on the residual at the previous iterate and
                                   * the scheme's measured superlinear convergence.  Measured on the 1/4° and 1/6° surfaces every certified
                                   * cell is within 6e-7 of the exact path at the default budget (tests/test_certified.py holds 1e-6 on
                                   * the full surface); no cell outside the budget has been observed, none is excluded by construction.
                                   * That is why the path is opt-in and why bench.py's `value` is the exact path's.  Decisions are per cell: a result never
                                   * depends on which cells share its wave, chunk or rank.  Applies where the round-3 ocean kernel
                                   * runs in its narrow geometry under the convergence stop rule with tolerance ≥ 1e-9 and maxiter ≥ 40 (cf_solver_iteration_path tells);
                                   * FixedIterations(n), CoefficientBasedFluxes and the sea-ice interface always take the exact path.
                                   * In this mode the optional `iterations` output is a diagnostic: the number of map evaluations of a
                                   * certified cell, or CF_CERTIFIED_EXACT_FLAG | (the reference's trip count) for an exact-path cell;
                                   * the optional friction_velocity / temperature_scale / humidity_scale outputs are the fixed point's. */
#define CF_SOLVER_PATH_EXACT 0
#define CF_SOLVER_PATH_CERTIFIED 1
#define CF_CERTIFIED_EXACT_FLAG 0x100
#define CF_OPT_CERTIFIED_BUDGET 11 /* the certificate's budget in units of 1e-9 (default 800 = 8e-7; 50 … 1000000 — anything above ≈ 900 aims beyond the 1e-6 tolerance and is for measurements).  The solve's own
                                   * convergence error (≈ 2e-8 in the same metric) comes on top: at the default every cell measured so far
                                   * is within 1e-6 of the exact path, the north star's tolerance (worst ≈ 6e-7); see CF_OPT_SOLVER_PATH
                                   * for what the certificate does and does not establish. */
#define CF_OPT_ICE_FREE_CELLS 12   /* what compute_atmosphere_sea_ice_fluxes! does on wet cells that carry no ice (ℵ = 0 AND hᵢ = 0):
                                   * CF_ICE_FREE_ITERATE (0, default): the interface iteration runs on every wet cell, as the recalled
                                   * upstream kernel does — on a 1/4° surface with polar ice 77 % of the interface solve's time;
                                   * CF_ICE_FREE_ZERO (1): such cells get zero_interface_state — zero interface fluxes, zero iterations,
                                   * the skin temperature left at its input — and whole batches of open water skip the solve.
                                   * The five net ocean fields do not read the atmosphere–sea-ice interface, and the net sea-ice
                                   * fluxes of every cell with ℵ > 0 are the same bits in both modes (tested); the interface
                                   * fluxes and the net sea-ice fluxes of ice-free cells differ (they weight nothing: ℵ = 0).  Which of
                                   * the two upstream does is one of the forks julia/oracle_dump.jl records. */
#define CF_ICE_FREE_ITERATE 0
#define CF_OPT_LATENCY_LAYOUT 13   /* which kernels carry the EXACT path of the SimilarityTheoryFluxes ocean solve when a launch leaves every
                                   * SIMD with one or two waves (a latitude slab of a strongly scaled run, a small surface):
                                   * 1 (default, automatic): with the COARE similarity profile (`:corrected`), chunk plans of at most two
                                   * workgroups per CU take the kernels of coflux_solver_slab.hip — the same iteration laid out in big
                                   * basic blocks and re-scheduled for latency after register allocation (csrc/tools/gcn_sched.py):
                                   * 1440×70 steps in 22.0 instead of 24.3 µs; 0: never; 2: always (measurements: with three waves
                                   * per SIMD the layout buys nothing and costs occupancy, and with the plain logarithmic profile the
                                   * slowest waves sit in the per-lane branch to the general ψ at the roughness lengths, which a lone
                                   * wave issues no faster re-ordered).  Results are the same bits in every mode (tests/test_slab_line.py).  Needs gustiness_parameter != 0 (every
                                   * SimilarityTheoryFluxes preset of the reference); other parameter sets keep the production kernels. */
#define CF_ICE_FREE_ZERO 1
#define CF_OPT_HALO_IN_SOLVER_LAUNCH 14 /* 1: cf_time_steps with CF_HALO_PEER hands each step's halo rows to that step's solver launch instead of
                                   * launching the exchange kernel in front of it: 2 × 4 rider workgroups at the head of the launch run the
                                   * same mailbox protocol (one per direction and field), the chunks whose cells read halo rows — the south
                                   * ring row, the last interior row, the north ring row — are dispatched behind every other chunk and wait
                                   * for the riders' counter after their own start phase; interior chunks never wait, and the neighbours'
                                   * latency passes under interior work.  Applies where the step's solver launch is the exact path of the
                                   * round-3 ocean kernel with tail workgroups (CF_OPT_MERGED_PREFETCH = 2 inside a pipelined cf_time_steps);
                                   * any other step gets the exchange kernel as before.  The rows that arrive are the same bits
                                   * (tests/test_steps.py: 2 and 4 ranks, lat-lon and tripolar).  0 (default): the exchange kernel of its
                                   * own.  What it buys on N devices is UNMEASURED: no multi-GPU node has run either form;
                                   * one launch boundary and a 3–5 µs kernel per step are what it removes from a 26 µs slab step.        */
#define CF_SOLVER_TABLES 0  /* default: reference iteration path on LDS-tabulated ψ / log / exp.  Accuracy of the tabulated primitives
                               against libm (tests/test_gpu_parity.py::test_device_primitives_accuracy): ψ_m, ψ_h ≤ 5e-12 of
                               max(|ψ|, 1) for |ζ| < 1024 (every state a converging iteration can stop on) and ≤ 2e-10 for
                               1024 ≤ |ζ| ≤ 1e9 (the first one or two iterates from the 1e-4 first guess; extremely stable sea
                               ice); log ≤ 2e-13 absolute; exp for ℓ_q ≤ 6e-10 relative.  Every solver path shares the tables.
                               What that buys in the returned fluxes (all ≥ 100 × inside the 1e-6 target): convergence mode —
                               early-iterate error is contracted away, measured worst 7e-12 scaled, identical trip counts on
                               the 1/4° surface; FixedIterations(n) — a stopped iterate can still sit in the coarse tier, the
                               tests hold 1e-9 on the ocean presets and 1e-8 on the sea-ice interface
                               (test_gpu_parity.py: sea_ice_fixed5), where the skin-temperature balance amplifies. */
#define CF_SOLVER_LIBM 1    /* same iteration on ocml's libm (slow; cross-check)                     */
int cf_set_option(cf_ctx* ctx, int option, int value);
/* *path = CF_SOLVER_PATH_* that cf_compute_atmosphere_ocean_fluxes / cf_update_state would run with the current
 * options and flux parameters (the certified path falls back to the exact one where it does not apply) — the predicate
 * the launch itself decides on.  One launch runs the exact iteration whatever this answers: cf_update_state_sea_ice with
 * CF_OPT_MERGED_PREFETCH = 2 carries the ocean solve in the interface solve's launch, whose rider is the exact kernel.   */
int cf_solver_iteration_path(cf_ctx* ctx, int* path);
/* *layout = 1 when the ocean solve's EXACT path would be carried by the kernels laid out for one or two waves per SIMD
 * (CF_OPT_LATENCY_LAYOUT; coflux_solver_slab.hip) with the current options, flux parameters and the chunk table as built
 * (cf_ensure_chunk_table first: the automatic mode looks at its workgroup count), else 0.  A measurement aid.         */
int cf_solver_latency_layout(cf_ctx* ctx, int* layout);
/* Self-test hook: y[k] = f(x[k]) with the device primitives the solver uses
 * (f: 0 log, 1 exp, 2 cbrt, 3 sqrt, 4 1/x, 5 ψ_m(ζ), 6 ψ_h(ζ)); d_x, d_y device pointers.        */
int cf_debug_eval(cf_ctx* ctx, int function, int n, const double* d_x, double* d_y);
/* Self-test hook (host arithmetic only, works without a GPU): the solver's chunk plan for a surface of total
 * cost `total_cost` (wet cell = the returned cost unit, land cell = 1) on `cu_count` compute units.
 * out[0] = number of rounds, then per round (wet cells per chunk, number of chunks).  Returns the wet-cell cost
 * unit, or −1 if `capacity` ints are too few. */
int cf_debug_chunk_plan(long long total_cost, int cu_count, int forced_wet_per_chunk, int* out, int capacity);
int cf_sync(cf_ctx* ctx);

/* Device memory for callers that cannot own HIP memory themselves (Julia without AMDGPU.jl). */
void* cf_device_alloc(cf_ctx* ctx, size_t bytes);
int cf_device_free(cf_ctx* ctx, void* d_ptr);
int cf_h2d(cf_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int cf_d2h(cf_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* ------------------------------------------------------------------------------------------
 * The hot path — one entry per reference function of SURVEY.md §3.1
 * ---------------------------------------------------------------------------------------- */

/* interpolate_atmosphere_state!(interfaces, atmosphere::JRA55PrescribedAtmosphere, model)
 * (construction site atmosphere.jl:20-29, README.md:74): bilinear in (λ,φ) × linear in time,
 * rain+snow summed, winds rotated to the grid frame.                                           */
int cf_interpolate_atmosphere_state(cf_ctx* ctx, const cf_atmos_source* src,
                                    const cf_interp_weights* w, const cf_exchange_fields* out);

/* compute_atmosphere_ocean_fluxes!(coupled_model) with SimilarityTheoryFluxes
 * (omip_simulation.jl:42-49; README.md:75): the per-cell Monin–Obukhov fixed point.            */
int cf_compute_atmosphere_ocean_fluxes(cf_ctx* ctx, const cf_ocean_surface* ocean,
                                       const cf_exchange_fields* atmos,
                                       const cf_interface_fluxes* out);

/* compute_net_ocean_fluxes!(coupled_model): radiation (atmosphere.jl:41-44) + (1−ℵ) partition +
 * unit conversion → τx, τy, JT, JS (omip_diagnostics.jl:77-80).                                 */
int cf_compute_net_ocean_fluxes(cf_ctx* ctx, const cf_ocean_surface* ocean,
                                const cf_exchange_fields* atmos,
                                const cf_interface_fluxes* fluxes,
                                const cf_sea_ice_fields* ice /* may be NULL */,
                                const cf_interp_weights* w /* latitude only; may be NULL */,
                                const cf_net_ocean_fluxes* out);

/* update_state!(coupled_model) (NEMOTKE/nemo_tke_compute_closure_fields.jl:7-8): the three
 * stages above back to back on the context's stream (three launches, no host synchronisation).  */
int cf_update_state(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                    const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                    const cf_interface_fluxes* fluxes, const cf_sea_ice_fields* ice,
                    const cf_net_ocean_fluxes* net);

/* ------------------------------------------------------------------------------------------
 * JRA55 snapshot window in HBM: JRA55PrescribedAtmosphere(arch; time_indices_in_memory = n,
 * prefetch = true) of atmosphere.jl:20-29 / launch.sh:86-93 — `n_slots` 3-hourly snapshots of the nine
 * variables resident on the device, refilled from host memory while the model steps.  `time_index` is the
 * monotone snapshot COUNTER ⌊t/Δt⌋ (not the index inside a repeat-year record, which jumps back at the wrap and
 * would put two consecutive snapshots into one slot); snapshot counter t lives in slot t mod n_slots.  Uploads run on the window's own copy stream out of pinned staging buffers and are
 * ordered against the context's compute stream with events in both directions: a slot is not overwritten
 * before the interpolations already queued have read it, and an interpolation does not start before the two
 * snapshots it brackets have landed.  Reading the files (NetCDF) stays on the host side of this boundary.
 * ---------------------------------------------------------------------------------------- */
typedef struct cf_window cf_window;
int cf_window_create(cf_ctx* ctx, int32_t ns_x, int32_t ns_y, int32_t n_slots, cf_window** out);
int cf_window_destroy(cf_window* w);
/* Pinned staging buffer [ns_y][ns_x] of (slot, variable) for readers that fill it in place.  Call
 * cf_window_wait_slot first: the previous upload out of this buffer may still be in flight. */
float* cf_window_host_buffer(cf_window* w, int32_t slot, int32_t variable);
int cf_window_wait_slot(cf_window* w, int32_t slot);
/* The slot's nine staging buffers hold snapshot `time_index`: start its asynchronous copy into HBM. */
int cf_window_commit(cf_window* w, int32_t slot, int64_t time_index);
/* Convenience: copy nine host arrays (CF_JRA55_* order) into slot time_index mod n_slots and commit. */
int cf_window_upload(cf_window* w, int64_t time_index, const float* const* host_vars);
/* Slot that holds (or is receiving) snapshot `time_index`, or -1. */
int cf_window_find(cf_window* w, int64_t time_index);
/* Source descriptor for interpolating between snapshots n1 and n2 (both must be in the window; the compute
 * stream is made to wait for their uploads).  Pass it to cf_interpolate_atmosphere_state / cf_update_state. */
int cf_window_source(cf_window* w, int64_t n1, int64_t n2, double time_fraction, cf_atmos_source* out);

/* ------------------------------------------------------------------------------------------
 * Atmosphere–sea-ice interface (config sites omip_simulation.jl:62-69 ":corrected",
 * :105-113 ":ncar", atmosphere.jl:34-44).  Same Monin–Obukhov iteration, but the interface temperature is
 * a skin temperature found inside the loop from the surface energy balance against the conductive flux
 * through the ice (SkinTemperature(ConductiveFlux)), humidity saturates over ice, latent heat is that of
 * sublimation.  The ice state itself (ClimaSeaIce) is prescribed input.
 * ---------------------------------------------------------------------------------------- */
/* flux_balance_temperature(SkinTemperature(ConductiveFlux)) inside the iteration ([UPSTREAM-RECALL]; the pin of
 * julia/oracle_dump.jl will tell which one upstream ships):
 *   EXPLICIT       T★ = Tᵢ − (h/k)(Q_v + Q_c + Q_d + εσTₛ⁴) with every flux at the previous iterate; its gain
 *                  (h/k)·∂Q/∂T exceeds 1 for ice thicker than ≈ 0.1–0.2 m in wind, where it orbits under the ±ΔT_max
 *                  limiter until maxiter;
 *   SEMI_IMPLICIT  the upwelling longwave linearised about the previous skin temperature,
 *                  T★ = (Tᵢ − (h/k)(Q_v + Q_c + Q_d)) / (1 + (h/k) εσ Tₛ³)  — the damped form.                    */
#define CF_SKIN_EXPLICIT 0
#define CF_SKIN_SEMI_IMPLICIT 1
typedef struct cf_sea_ice_params {
    int32_t struct_size;                   /* sizeof(cf_sea_ice_params) */
    int32_t skin_temperature_scheme;       /* CF_SKIN_* */
    double conductivity;                   /* 2.0 W m⁻¹ K⁻¹  (ClimaSeaIce ConductiveFlux)          */
    double consolidation_thickness;        /* 0.05 m: thinner ice conducts as if this thick          */
    double maximum_temperature_change;     /* 5 K per iteration (SkinTemperature limiter)            */
    double ice_salinity;                   /* reserved (the skin is capped at the FRESHWATER melting point) */
    double liquidus_slope;                 /* 0.054 K per g/kg: ice bottom at T_fw − m·S_ocean           */
    double freshwater_melting_temperature; /* 273.15 K                                               */
    double albedo;                         /* 0.7 unless cf_sea_ice_state.albedo is given            */
    double emissivity;                     /* 1.0, SurfaceRadiationProperties(sea_ice_albedo, 1.0), atmosphere.jl:44 */
    double temperature_offset;             /* 273.15: top_surface_temperature is in °C, atmosphere.jl:38 */
} cf_sea_ice_params;

int cf_default_sea_ice_params(cf_sea_ice_params* p);

/* sea_ice.model.{ice_concentration, ice_thickness, ice_thermodynamics.top_surface_temperature, velocities}
 * (atmosphere.jl:34-39, src/ClimaOcean.jl:62-63); all at cell centres, ocean-grid layout. */
typedef struct cf_sea_ice_state {
    const double* concentration;    /* ℵ                                                             */
    const double* thickness;        /* hᵢ [m]                                                        */
    const double* top_temperature;  /* previous skin temperature [°C]: initial guess of the iteration.  May be the very buffer
                                       cf_compute_atmosphere_sea_ice_fluxes writes the new one to (out->temperature): every
                                       cell reads its guess before it writes its result — the in-place update of
                                       sea_ice.model.ice_thermodynamics.top_surface_temperature in a coupled run          */
    const double* u;                /* ice velocity [m/s] or NULL (⇒ 0)                              */
    const double* v;
    const double* albedo;           /* per-cell albedo or NULL: then cf_set_sea_ice_albedo's scheme, else the constant */
    const double* snow_thickness;   /* hₛ [m] (sea_ice.model.snow_thickness, atmosphere.jl:34) or NULL (⇒ 0): read by the
                                       CCSM3 albedo only                                                            */
} cf_sea_ice_state;

/* SeaIceAlbedo(hi, hs, Ts) — the CCSM3 sea-ice albedo the reference hands to SurfaceRadiationProperties
 * (atmosphere.jl:30-44; "reads live model fields").  The scheme itself lives in NumericalEarth; restated from its
 * source, Briegleb et al. 2004 (NCAR/TN-463) as implemented in CICE's `ccsm3` option: thickness-dependent bare-ice
 * albedo  α_ice = α_cold·f_h + α_ocean·(1 − f_h),  f_h = min(atan(4 hᵢ)/atan(4 h_max), 1);  both ice and snow darken
 * linearly over the last ΔT_melt below the melting point;  snow covers the fraction hₛ/(hₛ + h_patch);  the two
 * spectral bands are averaged with `visible_fraction` (JRA55 carries one broadband rsds).                          */
typedef struct cf_sea_ice_albedo_params {
    int32_t struct_size;
    int32_t reserved;
    double ice_visible, ice_near_infrared;     /* 0.78, 0.36  cold, thick, bare ice                       */
    double snow_visible, snow_near_infrared;   /* 0.98, 0.70  cold snow                                   */
    double ocean_albedo;                       /* 0.06        limit of vanishing thickness                */
    double reference_thickness;                /* 0.3 m       h_max                                       */
    double melt_temperature_range;             /* 1 K         ΔT_melt                                     */
    double ice_melt_change;                    /* 0.075       darkening of ice at the melting point       */
    double snow_melt_change_visible;           /* 0.10                                                    */
    double snow_melt_change_near_infrared;     /* 0.15                                                    */
    double snow_patch_thickness;               /* 0.02 m                                                  */
    double visible_fraction;                   /* 0.5         weight of the visible band in the broadband albedo [UNVERIFIED] */
    double melting_temperature;                /* 0 °C        in the units of top_temperature             */
} cf_sea_ice_albedo_params;
int cf_default_sea_ice_albedo_params(cf_sea_ice_albedo_params* p);
/* Switch the atmosphere–sea-ice interface and the net sea-ice fluxes to this scheme wherever cf_sea_ice_state.albedo
 * is NULL (params = NULL: back to the constant albedo of cf_sea_ice_params).                                       */
int cf_set_sea_ice_albedo(cf_ctx* ctx, const cf_sea_ice_albedo_params* params);
/* The albedo field on its own (diagnostics, tests): d_albedo[k] = SeaIceAlbedo(hᵢ, hₛ, Tₛ) for every ocean-grid cell. */
int cf_compute_sea_ice_albedo(cf_ctx* ctx, const cf_sea_ice_albedo_params* params, const double* d_ice_thickness,
                              const double* d_snow_thickness /* may be NULL */, const double* d_top_temperature,
                              double* d_albedo);

/* compute_sea_ice_ocean_fluxes!(coupled_model) with ThreeEquationHeatFlux(; friction_velocity =
 * MomentumBasedFrictionVelocity()) (omip_simulation.jl:71-77: "three-equation ice-ocean heat flux with momentum-based
 * friction velocity computed from actual ice-ocean stress, McPhee 1992, 2008") and frazil formation.  Per wet cell:
 *   u★   = max(√|τ_io|, u★_min),  τ_io the kinematic ice–ocean stress averaged from its faces to the cell centre:
 *           cell (i, j) reads x_stress[i], x_stress[i+1], y_stress[j], y_stress[j+1], so the two stress fields need
 *           their EAST x-halo column and NORTH halo row filled (periodic / slab exchange / tripolar fold) by whoever
 *           produces them — cf_time_steps and the halo entry points only move T, S, u, v;
 *   frazil: T_o < T_f(S_o) = −m S_o  ⇒  Q_frazil = ρ_o c_o Δz (T_o − T_f)/Δt (< 0: heat the ice model must supply by
 *           freezing), and the exchange below sees T_o = T_f;
 *   three equations (Holland & Jenkins 1999; McPhee et al. 2008) for the interface (T_b, S_b) and melt rate w:
 *           ρ_o c_o α_h u★ (T_o − T_b) = ρ_i ℒ w,   ρ_o α_s u★ (S_o − S_b) = ρ_i w (S_b − S_i),   T_b = −m S_b
 *           ⇒ a quadratic in S_b that does not depend on u★;
 *   Q_io = ℵ ρ_o c_o α_h u★ (T_o − T_b)  [W m⁻², positive = the ocean loses heat],
 *   Jˢ_io = ℵ α_s u★ (S_o − S_b)         [g/kg m s⁻¹, positive = the ocean loses salt (melt water dilutes it)].
 * The outputs are exactly the fields cf_sea_ice_fields.interface_heat / salt_flux and cf_compute_net_sea_ice_fluxes'
 * frazil_heat / interface_heat take.  Coefficients marked UNVERIFIED are the recalled ClimaSeaIce defaults.          */
typedef struct cf_ice_ocean_params {
    int32_t struct_size;
    int32_t reserved;
    double heat_transfer_coefficient;   /* α_h = 0.0095 [UNVERIFIED]                                   */
    double salt_transfer_coefficient;   /* α_s = α_h / 35 [UNVERIFIED]                                  */
    double minimum_friction_velocity;   /* 0 m/s                                                        */
    double ice_density;                 /* 917 kg m⁻³                                                   */
    double latent_heat_of_fusion;       /* 334 000 J kg⁻¹                                               */
    double ice_salinity;                /* 4 g/kg                                                       */
    double liquidus_slope;              /* m = 0.054 K per g/kg: T_f = −m S in °C, i.e. freshwater melts at 0 °C = the 273.15 K of
                                           cf_sea_ice_params.freshwater_melting_temperature (the two interfaces agree at the default) */
    double top_cell_thickness;          /* Δz of the ocean's top cell [m] (frazil)                      */
    double time_step;                   /* Δt [s] (frazil); ≤ 0 disables frazil                         */
} cf_ice_ocean_params;
int cf_default_ice_ocean_params(cf_ice_ocean_params* p);
typedef struct cf_ice_ocean_fluxes {
    double* interface_heat;   /* Q_io                                             */
    double* salt_flux;        /* Jˢ_io                                            */
    double* frazil_heat;      /* Q_frazil (may be NULL)                           */
    double* friction_velocity;/* u★ (optional diagnostic, may be NULL)           */
} cf_ice_ocean_fluxes;
int cf_compute_sea_ice_ocean_fluxes(cf_ctx* ctx, const cf_ice_ocean_params* params, const cf_ocean_surface* ocean,
                                    const double* d_concentration, const double* d_x_stress /* u-faces, kinematic */,
                                    const double* d_y_stress /* v-faces */, const cf_ice_ocean_fluxes* out);

/* `ice_fluxes` = the atmosphere_sea_ice_fluxes formulation (e.g. corrected_atmosphere_sea_ice_fluxes). */
int cf_set_sea_ice_formulation(cf_ctx* ctx, const cf_flux_params* ice_fluxes, const cf_sea_ice_params* ice);
/* compute_atmosphere_sea_ice_fluxes!(coupled_model): out.temperature receives the new skin temperature
 * [°C]; latent_heat uses the sublimation enthalpy.  Needs exchange fields Qs, Ql as well. */
int cf_compute_atmosphere_sea_ice_fluxes(cf_ctx* ctx, const cf_sea_ice_state* ice, const cf_ocean_surface* ocean,
                                         const cf_exchange_fields* atmos, const cf_interface_fluxes* out);

/* compute_net_sea_ice_fluxes!(coupled_model): the heat the sea-ice thermodynamics receives at its two faces,
 *   ΣQt = (Q_d + Q_u + Q_c + Q_v)·[ℵ > 0],  Q_u = εσT_s⁴ at the skin temperature just computed,
 *                                            Q_d = −(1 − α)Q_s − εQ_ℓ   (positive upward, W m⁻²),
 *   ΣQb = Q_frazil + Q_interface             (the ice–ocean exchange, taken as given: SURVEY §8f),
 * zero on land.  `ai_fluxes` is the output of cf_compute_atmosphere_sea_ice_fluxes; radiative properties are
 * those of cf_set_sea_ice_formulation.  frazil_heat / interface_heat may be NULL (⇒ 0).                  */
typedef struct cf_net_sea_ice_fluxes {
    double* top_heat;     /* ΣQt */
    double* bottom_heat;  /* ΣQb */
} cf_net_sea_ice_fluxes;
int cf_compute_net_sea_ice_fluxes(cf_ctx* ctx, const cf_sea_ice_state* ice, const cf_ocean_surface* ocean,
                                  const cf_exchange_fields* atmos, const cf_interface_fluxes* ai_fluxes,
                                  const double* frazil_heat, const double* interface_heat,
                                  const cf_net_sea_ice_fluxes* out);

/* update_state!(coupled_model) of a model WITH sea ice (BASELINE config 3; omip_simulation.jl:139-163): cf_update_state
 * followed by cf_compute_atmosphere_sea_ice_fluxes and cf_compute_net_sea_ice_fluxes on the same stream — five
 * launches, no host synchronisation.  `ice_partition` (ℵ, ice–ocean fluxes) feeds the ocean partition, `ice_state`
 * the interface solve; `ai_fluxes->temperature` receives the new skin temperature. */
int cf_update_state_sea_ice(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                            const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                            const cf_interface_fluxes* ao_fluxes, const cf_sea_ice_fields* ice_partition,
                            const cf_net_ocean_fluxes* net, const cf_sea_ice_state* ice_state,
                            const cf_interface_fluxes* ai_fluxes, const double* frazil_heat,
                            const double* interface_heat, const cf_net_sea_ice_fluxes* net_ice);

/* ------------------------------------------------------------------------------------------
 * JRA55PrescribedLand(arch; …) (atmosphere.jl:46): river discharge `friver` and iceberg calving `licalvf`
 * (jra55_data_staging.jl:8), kg m⁻² s⁻¹ on the JRA55 grid, the last two of the eleven staged variables.  Same window
 * layout, same bilinear × linear-in-time interpolation as the atmosphere; the two are summed into ONE ocean-grid field.
 * cf_set_land_freshwater hands that field to compute_net_ocean_fluxes!: the freshwater reaches the ocean as
 *     JS += −Sₒ · (−M_land / ρ_f)      (not scaled by 1 − ℵ: rivers run under ice; the minimum-salinity guard applies)
 * [UPSTREAM-RECALL: NumericalEarth adds the land freshwater flux to the ocean's freshwater budget; the line above is
 * this library's statement of it, both oracles restate the same].  NULL switches it off.                          */
typedef struct cf_land_source {
    const float* friver;    /* device, [n_levels][ns_y][ns_x] */
    const float* licalvf;   /* device, may be NULL */
    int32_t ns_x, ns_y, n_levels, level1, level2, reserved;
    double time_fraction;
} cf_land_source;
int cf_interpolate_land_freshwater(cf_ctx* ctx, const cf_land_source* src, const cf_interp_weights* w, double* d_out);
int cf_set_land_freshwater(cf_ctx* ctx, const double* d_land_freshwater /* ocean-grid field, kg m⁻² s⁻¹, or NULL */);

/* SurfaceFluxRestoring(DatasetRestoring(…; rate = piston_velocity / (Δz days))) riding on the salinity top boundary
 * condition as the `additional_fluxes` of MultipleFluxes{flux_field, additional_fluxes} (omip_simulation.jl:175-206,
 * 507-523): `_materialize_top_flux!` evaluates it into a 2-D buffer, J_add = v_p (Sₒ − S★) on wet cells (positive = salt
 * leaves where the surface is saltier than the target).  The buffer is what cf_normalize_salinity_flux takes as
 * d_additional; the ocean applies flux_field + additional.                                                      */
int cf_materialize_salinity_restoring(cf_ctx* ctx, double piston_velocity /* m/s */, const double* d_target_salinity,
                                      const cf_ocean_surface* ocean, double* d_buffer);

/* NormalizeSalinity callback (src/OMIPConfigurations/omip_simulation.jl:182-220, added at :385-388):
 * subtract the global, area-weighted mean over wet cells of (salinity flux [+ additional flux]) from
 * the salinity-flux field — `compute!(mean_total); parent(flux_field) .-= mean_total`, so the constant
 * is subtracted from the whole parent array, halos and land included.  `d_area` are the horizontal cell
 * areas Az in ocean-grid layout (NULL ⇒ uniform); `d_additional` is the materialised additional top
 * flux (SurfaceFluxRestoring …) or NULL.  With an initialised communicator (cf_comm_init) the two sums
 * are all-reduced over the ranks (one RCCL all-reduce of two doubles) so that every slab subtracts the
 * same global mean.  The mean is also written to d_mean_out (device double, may be NULL).            */
int cf_normalize_salinity_flux(cf_ctx* ctx, double* d_flux, const double* d_additional, const double* d_area,
                               const void* d_mask, double* d_mean_out);

/* ------------------------------------------------------------------------------------------
 * Measurement helper: run `launches` back-to-back launches of one stage on the context's stream
 * bracketed by HIP events ON THAT STREAM and return the average milliseconds per launch.
 * stage: 0 = interpolate, 1 = atmosphere–ocean fluxes, 2 = net ocean fluxes, 3 = update_state.  */
#define CF_STAGE_INTERPOLATE 0
#define CF_STAGE_AO_FLUXES 1
#define CF_STAGE_NET_FLUXES 2
#define CF_STAGE_UPDATE_STATE 3
#define CF_STAGE_COPY 4 /* device copy of `bytes` (calibrates the HBM denominator) */
int cf_time_stage(cf_ctx* ctx, int stage, int launches, const cf_atmos_source* src,
                  const cf_interp_weights* w, const cf_ocean_surface* ocean,
                  const cf_exchange_fields* atmos, const cf_interface_fluxes* fluxes,
                  const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net,
                  double* ms_per_launch);
int cf_time_copy(cf_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int launches,
                 double* ms_per_launch);
/* Per-kernel timing INSIDE a caller's timed region: while enabled, cf_update_state brackets each of
 * its kernels with HIP events on the launch stream (no host sync).  cf_profile_read synchronises and
 * returns the average duration of `kernel` over the recorded steps.  cf_profile_enable(ctx, n)
 * (re)arms the recorder for n records; 0 disables.                                                 */
#define CF_KERNEL_INTERPOLATE 0
#define CF_KERNEL_AO_FLUXES 1
#define CF_KERNEL_NET_FLUXES 2
int cf_profile_enable(cf_ctx* ctx, int max_records);
int cf_profile_read(cf_ctx* ctx, int kernel, double* avg_ms, int* records);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: latitude-slab halo rows over RCCL (Partition(1,4) launch.sh:165,
 * Partition(1,8) pbs_launch.sh:51).  The unique id is produced on rank 0 and distributed by the
 * host (MPI.jl bcast in Julia; torch.distributed in the Python mirror).
 * ---------------------------------------------------------------------------------------- */
#define CF_COMM_ID_BYTES 128
int cf_comm_unique_id(void* id128);
int cf_comm_init(cf_ctx* ctx, const void* id128, int rank, int nranks);
int cf_comm_destroy(cf_ctx* ctx);
/* RCCL's own view of the communicator cf_comm_init made: the number of ranks it connected (ncclCommCount), this rank's index
 * (ncclCommUserRank) and the HIP device it sits on (ncclCommCuDevice); any of the three may be NULL.  What a launcher
 * prints beside its own WORLD_SIZE so that a run whose ranks never met shows (bench.py: config.rccl_comm_ranks).       */
int cf_comm_count(cf_ctx* ctx, int* nranks, int* rank, int* device);
/* Exchange `rows` boundary rows of `nfields` ocean-grid arrays with the south (rank−1) and north
 * (rank+1) neighbours: my first/last interior rows → their north/south halos.  With ring = 1 the kernels
 * also compute the ring row j = ny, whose cell-centre v needs the y-face at j = ny + 1: exchange
 * rows = ring + 1 of the ocean state (cf_time_steps insists on it).                                */
int cf_halo_exchange_rows(cf_ctx* ctx, double* const* d_fields, int nfields, int rows);

/* Peer-direct halo rows (the reference's analogue is the CUDA-IPC transport of its GPU-aware MPI,
 * experiments/OMIPSimulations/scripts/launch.sh:270-290).  At 46 KB per neighbour the exchange is latency bound, so
 * beside the RCCL path there is one that needs no collective library at all: every rank owns a MAILBOX in
 * fine-grained device memory, exported as a HIP IPC handle; a neighbour maps it and one kernel per step (i) stores
 * this rank's boundary rows straight into both neighbours' mailboxes over xGMI, (ii) publishes a sequence number with
 * a system-scope release, (iii) waits (bounded spin) for the neighbours' sequence numbers and (iv) copies their rows
 * into the halo rows — one launch, no host round trip.  Mailboxes are double-buffered by sequence parity; a rank can
 * only write parity p again after its own step-(s+1) wait, which implies the neighbour has drained step s.
 *   cf_peer_halo_export   allocate this rank's mailbox for at most `max_fields` fields × `max_rows` rows, write its
 *                         IPC handle (CF_PEER_HANDLE_BYTES) — the host distributes handles (MPI / torch.distributed).
 *                         Fails with CF_ERR_COMM when the device has no fine-grained memory to give (the protocol is
 *                         not safe on coarse-grained memory): use the RCCL exchange then.
 *   cf_peer_halo_connect  map the south (rank−1) and north (rank+1) mailboxes; NULL at the ends of the slab ring.
 *                         On the LAST rank of a tripolar grid the north handle is NULL too: its north boundary is the
 *                         fold (one_degree_tripolar.jl:48-51), served locally by cf_fold_north_halo, not by a peer.
 *   cf_halo_exchange_rows_peer   the per-step exchange, same meaning as cf_halo_exchange_rows.
 * A spin that exceeds its bound sets a sticky error that the next cf_sync reports (CF_ERR_COMM).            */
#define CF_PEER_HANDLE_BYTES 64
int cf_peer_halo_export(cf_ctx* ctx, int max_fields, int max_rows, void* handle_out);
int cf_peer_halo_connect(cf_ctx* ctx, const void* south_handle, const void* north_handle, int rank, int nranks);
int cf_halo_exchange_rows_peer(cf_ctx* ctx, double* const* d_fields, int nfields, int rows);
/* How many peer-direct exchanges this context has issued, and how many of them rode in a solver launch
 * (CF_OPT_HALO_IN_SOLVER_LAUNCH) instead of the exchange kernel of their own.  A measurement / test aid.      */
int cf_peer_halo_stats(cf_ctx* ctx, unsigned long long* exchanges, unsigned long long* in_solver_launch);

/* Tripolar fold (TripolarGrid(arch; size=(360,180,Nz)), OceanConfigurations/one_degree_tripolar.jl:48-51; the
 * fold itself lives in Oceananigans' zipper boundary condition, [UPSTREAM-RECALL] fold_north_center_center! /
 * _face_center! / _center_face!).  The northern boundary of the global grid is folded onto itself about the LAST
 * row of tracer points (that row is its own mirror image), so the north halo rows of the last latitude slab come
 * from that slab's own northern interior rows mirrored in i.  0-based interior indices, r = 1…rows:
 *   centres (T, S, fluxes):  f[i, ny−1+r] = sign · f[nx−1−i, ny−1−r]
 *   x-faces (u):             f[i, ny−1+r] = s    · f[(nx−i) mod nx, ny−1−r],  s = |sign| where nx−i wraps (i = 0)
 *   y-faces (v):             f[i, ny−1+r] = sign · f[nx−1−i, ny−r]
 * `sign` = −1 for the components of a vector (u, v change sign across the fold), +1 for scalars.  The periodic
 * x-halos of the written rows are filled from the written interior columns.  Folding is an involution on the
 * mirrored rows (tests/test_tripolar.py).                                                                    */
#define CF_FOLD_CENTER 0
#define CF_FOLD_X_FACE 1
#define CF_FOLD_Y_FACE 2
/* `nfields` fields in one launch: locations[f] ∈ CF_FOLD_*, signs[f] = ±1. */
int cf_fold_north_halo(cf_ctx* ctx, double* const* d_fields, const int* locations, const double* signs, int nfields,
                       int rows);

/* ------------------------------------------------------------------------------------------
 * run!(simulation) for a coupled model whose ocean component is prescribed (README.md:76-77): `nsteps` × time_step!
 * without returning to the host language — per step (optional halo rows) → cf_update_state.  What varies from step
 * to step is what varies in a coupled run: the clock (the JRA55 time fraction ñ advances by Δt/Δt_snapshot and the
 * bracketing snapshots move through the window) and the ocean surface state (step s reads ocean_states[s mod n]).
 * With `pipeline` ≠ 0 the atmosphere state of step s+1 is interpolated while the solver of step s runs (on the
 * auxiliary stream, or inside the step's launches: CF_OPT_MERGED_PREFETCH; the prescribed atmosphere does not depend on
 * the ocean); this needs two sets of exchange fields, step s uses atmos[s mod 2].
 *   CF_PIPELINE_WITHIN_CALL (1): only steps of this call are prefetched — the call reads the source at the levels of
 *     steps first_step … first_step + nsteps − 1 and nothing else, and leaves the other exchange set as the last-but-one
 *     step left it.
 *   CF_PIPELINE_CONTINUING (2): the last step also requests step first_step + nsteps, for a loop that goes on in the
 *     next call (which recognises the pending state by its output set, levels and time fraction).  The caller
 *     guarantees that the source slots of that step are resident and FINAL when this call is made (a sliding window
 *     must not recommit them in between — cf_window_commit / an upload into a level a pending request reads voids it
 *     silently), and that exchange set (first_step + nsteps) mod 2 is not read after this call's last step: it is
 *     overwritten.  A caller that stops after such a call has one unused interpolation in that set.
 * All launches are stream ordered; nothing synchronises with the host.       */
#define CF_PIPELINE_WITHIN_CALL 1
#define CF_PIPELINE_CONTINUING 2
#define CF_HALO_NONE 0
#define CF_HALO_RCCL 1
#define CF_HALO_PEER 2
typedef struct cf_run_schedule {
    int32_t struct_size;             /* sizeof(cf_run_schedule) */
    int32_t n_ocean_states;          /* ≥ 1 */
    const cf_ocean_surface* ocean_states;
    int32_t n_atmos_sets;            /* 1, or 2 with pipeline */
    int32_t pipeline;                /* 0, CF_PIPELINE_WITHIN_CALL or CF_PIPELINE_CONTINUING */
    const cf_exchange_fields* atmos; /* n_atmos_sets entries */
    int32_t first_level;             /* memory level of snapshot n₁ at step 0; levels advance cyclically */
    int32_t halo_backend;            /* CF_HALO_* */
    int32_t halo_rows;               /* rows exchanged per step (ring + 1) */
    int32_t fold_north;              /* 1: tripolar grid, this rank owns the fold (last rank) */
    double time_fraction;            /* ñ at step 0 */
    double time_fraction_increment;  /* Δt / Δt_snapshot: 20 min / 3 h = 1/9 (README.md:76) */
} cf_run_schedule;
int cf_time_steps(cf_ctx* ctx, int64_t first_step, int nsteps, const cf_run_schedule* schedule,
                  const cf_atmos_source* src /* levels / fraction are overridden per step */,
                  const cf_interp_weights* w, const cf_interface_fluxes* fluxes, const cf_sea_ice_fields* ice,
                  const cf_net_ocean_fluxes* net);

/* Builds the flux solver's schedule for `mask` (the cost-balanced chunk table and the wet lists, three tiny kernels and
 * two 4-byte read-backs) ahead of the first step instead of inside it.  Optional: cf_compute_atmosphere_ocean_fluxes,
 * cf_update_state and cf_time_steps build it on first use and whenever the mask pointer, its kind or the surface z
 * changes.  A mask rewritten in place keeps the old schedule: slower at worst, never wrong.                          */
int cf_ensure_chunk_table(cf_ctx* ctx, const void* mask);

/* Which kernels cf_update_state launches for the context's current formulation and options (a measurement aid:
 * bench.py names the dominant kernel and its algorithmic bytes from it).  *lean_kernel = 1: the round-3 ocean kernel
 * (coflux_solver_lean.hip), 0: the general solver; *fused_net = 1: the solver's epilogue also writes the cell-local net
 * ocean fluxes and a face-stress kernel follows (2: and interpolates the atmosphere state in its prologue), 0:
 * compute_net_ocean_fluxes! is its own launch.                                                                       */
int cf_solver_path(cf_ctx* ctx, int* lean_kernel, int* fused_net);

/* The pipelined form of update_state! for callers that drive the steps themselves: start the interpolation of the
 * NEXT step's atmosphere state into `out` on the auxiliary stream now; the next cf_update_state whose (levels, time
 * fraction, exchange fields) match finds it done (it waits on an event instead of launching the interpolation).
 * `out` must not be the set the current step's kernels still read.  cf_update_state's signature — the reference
 * seam — is unchanged.                                                                                          */
int cf_prefetch_atmosphere_state(cf_ctx* ctx, const cf_atmos_source* src_next, const cf_interp_weights* w,
                                 const cf_exchange_fields* out);
/* Forget every atmosphere state that was requested ahead and not consumed (cf_prefetch_atmosphere_state, cf_time_steps with
 * CF_PIPELINE_CONTINUING): a caller that leaves its stepping loop, overwrites an exchange set that holds such a state, or
 * changes the clock calls this so that no later cf_update_state mistakes the set's contents for the state it asks for.
 * An interpolation already running on the auxiliary stream is ordered ahead of whatever follows on the context's stream. */
int cf_discard_prefetched_atmosphere_state(cf_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* COFLUX_H */
