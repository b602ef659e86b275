// coflux_abi.cpp — the C ABI of libcoflux (include/coflux.h): context, parameter lowering,
// stream-ordered launches, HIP-event timing, and the RCCL latitude-slab halo exchange.
// There is no CPU backend: every compute entry point launches gfx950 kernels or fails.
#include "coflux_ctx.hpp"

static thread_local std::string g_error;
static RcclApi g_rccl;

int cf_fail(cf_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    if (ctx) {
        std::lock_guard<std::mutex> lock(ctx->error_mutex);
        ctx->error = buf;
    }
    return code;
}

static int wait_for_halos(cf_ctx* ctx);
int cf_flush_deferred_prefetch(cf_ctx* ctx);  // coflux_steps.cpp

static bool roughness_ok(const cf_roughness& r, bool scalar) {
    if (scalar) {
        if (r.kind == CF_SCALAR_ROUGHNESS_CONSTANT) return r.constant_length > 0;
        if (r.kind == CF_SCALAR_ROUGHNESS_REYNOLDS) return r.maximum_length > 0 && r.reynolds_A > 0;
        return false;
    }
    if (r.kind == CF_ROUGHNESS_CONSTANT) return r.constant_length > 0;
    if (r.kind == CF_ROUGHNESS_CHARNOCK || r.kind == CF_ROUGHNESS_WIND_CHARNOCK) return r.maximum_length > 0;
    return false;
}

static int lower_params(cf_ctx* ctx, const cf_flux_params* p, DevParams* d) {
    if (!p) return fail(ctx, CF_ERR_INVALID, "flux params are NULL");
    if (p->struct_size != (int32_t)sizeof(cf_flux_params))
        return fail(ctx, CF_ERR_INVALID, "cf_flux_params.struct_size = %d, library expects %zu", p->struct_size,
                    sizeof(cf_flux_params));
    if (p->abi_version != CF_ABI_VERSION)
        return fail(ctx, CF_ERR_INVALID, "cf_flux_params.abi_version = %d, library is %d", p->abi_version,
                    CF_ABI_VERSION);
    if (p->similarity_form < 0 || p->similarity_form > 1)
        return fail(ctx, CF_ERR_INVALID, "Unknown similarity_form: %d", p->similarity_form);
    if (p->stability_functions < 0 || p->stability_functions > 2)
        return fail(ctx, CF_ERR_INVALID, "Unknown stability_functions: %d", p->stability_functions);
    if (p->stop_kind < 0 || p->stop_kind > 1) return fail(ctx, CF_ERR_INVALID, "Unknown stop criteria: %d", p->stop_kind);
    if (p->velocity_difference < 0 || p->velocity_difference > 1)
        return fail(ctx, CF_ERR_INVALID, "Unknown velocity_formulation: %d. Options: relative(0), wind(1)",
                    p->velocity_difference);
    if (p->mask_kind < 0 || p->mask_kind > 2) return fail(ctx, CF_ERR_INVALID, "Unknown mask_kind: %d", p->mask_kind);
    if (p->maxiter < 0) return fail(ctx, CF_ERR_INVALID, "maxiter must be >= 0");
    if (!roughness_ok(p->momentum_roughness, false) || !roughness_ok(p->temperature_roughness, true) ||
        !roughness_ok(p->water_vapor_roughness, true))
        return fail(ctx, CF_ERR_INVALID, "invalid roughness-length block");
    if (!(p->reference_height > 0) || !(p->von_karman > 0) || !(p->gravitational_acceleration > 0))
        return fail(ctx, CF_ERR_INVALID, "reference_height, von_karman and gravitational_acceleration must be > 0");

    if (p->flux_formulation != CF_FORMULATION_SIMILARITY && p->flux_formulation != CF_FORMULATION_LARGE_YEAGER)
        return fail(ctx, CF_ERR_INVALID, "Unknown flux_formulation: %d", p->flux_formulation);
    if (p->flux_formulation == CF_FORMULATION_LARGE_YEAGER) {
        if (p->stop_kind != CF_STOP_FIXED)
            return fail(ctx, CF_ERR_INVALID, "CoefficientBasedFluxes needs solver_stop_criteria = FixedIterations(n)");
        if (!(p->ly_minimum_wind > 0) || !(p->ly_zeta_bound > 0))
            return fail(ctx, CF_ERR_INVALID, "Large-Yeager minimum wind and zeta bound must be > 0");
    }
    const cf_thermodynamics& t = p->thermo;
    DevParams D{};
    D.R_d = t.gas_constant / t.dry_air_molar_mass;
    D.R_v = t.gas_constant / t.water_molar_mass;
    D.eps = t.dry_air_molar_mass / t.water_molar_mass;
    D.delta = D.eps - 1.0;
    D.cp_d = D.R_d / t.kappa_d;
    D.cp_v = t.cp_v;
    D.cp_l = t.cp_l;
    D.cp_i = t.cp_i;
    D.LH_v0 = t.LH_v0;
    D.LH_s0 = t.LH_s0;
    D.T_0 = t.T_0;
    D.T_triple = t.T_triple;
    D.inv_T_triple = 1.0 / t.T_triple;
    D.p_triple = t.p_triple;
    D.T_freeze = t.T_freeze;
    D.T_icenuc = t.T_icenuc;
    D.inv_icenuc_span = 1.0 / (t.T_freeze - t.T_icenuc);
    D.pow_icenuc = t.pow_icenuc;
    D.inv_R_v = 1.0 / D.R_v;
    D.inv_R_d = 1.0 / D.R_d;
    D.Rd_over_Rv = D.R_d / D.R_v;
    const double dcp = t.cp_v - t.cp_l;
    D.svp_a_liq = dcp / D.R_v;
    D.svp_b_liq = (t.LH_v0 - dcp * t.T_0) / D.R_v;
    D.sw_inv_w = 1.0 / p->seawater.water_molar_mass;
    D.sw_inv_mu = 0.0;
    for (int k = 0; k < 4; ++k)
        D.sw_inv_mu += p->seawater.constituent_mass_fraction[k] / p->seawater.constituent_molar_mass[k];
    D.kappa = p->von_karman;
    D.beta_gust = p->gustiness_parameter;
    D.min_gust = p->minimum_gustiness;
    D.wind2_scale = 1.0;
    D.wind2_add = 0.0;
    if (p->shear_gustiness_coefficient > 0.0) {  // U_G² = (β w★)² + (c|Δu|)² + U_G,min²: the three terms add, nothing is floored
        D.wind2_scale = 1.0 + p->shear_gustiness_coefficient * p->shear_gustiness_coefficient;
        D.wind2_add = p->minimum_gustiness * p->minimum_gustiness;
        D.min_gust = 0.0;
    }
    D.profile_floor = p->similarity_profile_floor;
    D.tol = p->tolerance;
    D.h_ref = p->reference_height;
    D.h_bl = p->boundary_layer_height;
    D.g = p->gravitational_acceleration;
    D.inv_g = 1.0 / D.g;
    D.log_h = std::log(D.h_ref);
    D.similarity_form = p->similarity_form;
    // the Large–Yeager iteration uses the Paulson / −5ζ functions whatever the similarity block says
    D.stability = p->flux_formulation == CF_FORMULATION_LARGE_YEAGER ? CF_STABILITY_LARGE_YEAGER : p->stability_functions;
    D.stop_kind = p->stop_kind;
    D.maxiter = p->maxiter;
    D.velocity_difference = p->velocity_difference;
    D.mask_kind = p->mask_kind;
    D.rm = p->momentum_roughness;
    D.rt = p->temperature_roughness;
    D.rq = p->water_vapor_roughness;
    D.rho_o_inv = 1.0 / p->ocean_reference_density;
    D.c_o_inv = 1.0 / p->ocean_heat_capacity;
    D.rho_f_inv = 1.0 / p->ocean_freshwater_density;
    D.T_offset = p->ocean_temperature_offset;
    D.S_min = p->ocean_minimum_salinity;
    D.z_surface = p->ocean_surface_z;
    D.albedo = p->ocean_albedo;
    D.albedo_diffuse = p->ocean_albedo_diffuse;
    D.albedo_direct = p->ocean_albedo_direct;
    D.emissivity = p->ocean_emissivity;
    D.sigma = p->stefan_boltzmann;
    D.albedo_kind = p->ocean_albedo_kind;
    D.penetrating_sw = p->penetrating_shortwave;
    *d = D;
    return CF_OK;
}

// safety factor of the certificate for the accuracy of the secant Jacobian (measured ±4 %, scratch/certified_study.py)
static constexpr double CERT_SAFETY = 1.25;

static LoopParams loop_params(const cf_flux_params& p, const DevParams& d) {
    LoopParams C{};
    auto lg = [](double x) { return x > 0 ? std::log(x) : 0.0; };
    C.kappa = d.kappa;
    C.beta_gust = d.beta_gust;
    C.h_bl = d.h_bl;
    C.min_gust = d.min_gust;
    C.h_ref = d.h_ref;
    C.log_h = d.log_h;
    C.profile_floor = d.profile_floor;
    C.tol = d.tol;
    const cf_roughness &m = p.momentum_roughness, &t = p.temperature_roughness, &q = p.water_vapor_roughness;
    C.lm_m = m.maximum_length;
    C.const_m = m.constant_length;
    C.log_const_m = lg(m.constant_length);
    C.b_q = q.reynolds_b;
    C.log_A_q = lg(q.reynolds_A);
    C.log_lm_q = lg(q.maximum_length);
    C.log_const_q = lg(q.constant_length);
    C.b_t = t.reynolds_b;
    C.log_A_t = lg(t.reynolds_A);
    C.log_lm_t = lg(t.maximum_length);
    C.log_const_t = lg(t.constant_length);
    C.maxiter = p.maxiter;
    C.fixed = p.stop_kind == CF_STOP_FIXED;
    C.m_kind = m.kind;
    C.q_kind = q.kind;
    C.t_kind = t.kind;
    C.same_scalar = std::memcmp(&t, &q, sizeof(cf_roughness)) == 0;
    // branch-free instruction streams for the two production configurations (omip_simulation.jl:42-49, :63-69)
    const bool gusty = p.minimum_gustiness > 0;  // ⇒ U > 0 ⇒ u★ > 0: no division guards needed
    if (gusty && m.kind != CF_ROUGHNESS_CONSTANT && q.kind == CF_SCALAR_ROUGHNESS_REYNOLDS && C.same_scalar)
        C.specialization = SOLVER_OCEAN_LEAN;
    else if (gusty && m.kind == CF_ROUGHNESS_CONSTANT && q.kind == CF_SCALAR_ROUGHNESS_CONSTANT &&
             t.kind == CF_SCALAR_ROUGHNESS_CONSTANT)
        C.specialization = SOLVER_ICE;
    else
        C.specialization = SOLVER_GENERIC;
    if (p.flux_formulation == CF_FORMULATION_LARGE_YEAGER) {
        C.specialization = SOLVER_LY;
        C.ly_min_wind = p.ly_minimum_wind;
        C.ly_zeta_bound = p.ly_zeta_bound;
        C.ly_cd0 = p.ly_cd[0];
        C.ly_cd1 = p.ly_cd[1];
        C.ly_cd2 = p.ly_cd[2];
        C.ly_cd3 = p.ly_cd[3];
        C.ly_high_wind = p.ly_high_wind;
        C.ly_cd_high = p.ly_cd_high;
        C.ly_ce = p.ly_ce;
        C.ly_ch_s = p.ly_ch_stable;
        C.ly_ch_u = p.ly_ch_unstable;
        C.ly_lz = std::log(d.h_ref / 10.0);
    }
    C.inv_kappa = 1.0 / d.kappa;
    C.gust_c = d.beta_gust * d.beta_gust * d.beta_gust * d.h_bl / d.kappa;
    C.min_gust2 = d.min_gust * d.min_gust;
    C.x_scale = PSI_A * d.h_ref;
    C.two_inv_kappa = 2.0 / d.kappa;
    // the certified path's first guess and thresholds (coflux_certified.hpp): neutral profile over a 1e-4 m surface
    C.cert_u0 = 0.035;
    C.cert_two_inv_u0 = 2.0 / C.cert_u0;
    C.cert_chi0 = d.kappa / std::log(d.h_ref / 1e-4);
    C.cert_accept = 0x1p-23;   // (informational: the kernel's threshold is the compile-time CERT_ACCEPT)
    C.cert_budget = 8e-7 / CERT_SAFETY;
    C.cert_max_evals = 10;
    return C;
}

// (Re)build the LDS tables for the configured stability functions and upload them.
// The solver's chunk table depends on the wet mask only (coflux_solver.hip).  Built on the first call and
// whenever the mask pointer, its kind or the surface z changes; costs three tiny kernels and two 4-byte
// read-backs.  A mask rewritten in place keeps the old table — slower at worst, never wrong.
static int ensure_chunk_table(cf_ctx* ctx, const void* mask) {
    const DevParams& d = ctx->dev;
    const void* key = d.mask_kind == CF_MASK_NONE ? nullptr : mask;
    if (ctx->chunk_valid && ctx->chunk_mask == key && ctx->chunk_mask_kind == d.mask_kind &&
        ctx->chunk_z_surface == d.z_surface)
        return CF_OK;
    const int ncells = (ctx->grid.nx + 2 * ctx->grid.ring) * (ctx->grid.ny + 2 * ctx->grid.ring);
    if (!ctx->d_chunk_sums) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_chunk_sums, sizeof(int) * chunk_sums_capacity(ncells)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_chunk_begins, sizeof(int) * chunk_table_capacity(ncells)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_chunk_meta, sizeof(int) * 4));
    }
    int wet = 0, n = 0;
    // (a context whose steps carry tail workgroups gets the plan made for them)
    // (not with a sea-ice formulation: there the riders sit in the interface solve's tail, and both solves do best on the
    // arrival layers — measured 293 vs 283 µs per step)
    const int plan = (ctx->launch.ao_chunk == 0 && ctx->merged_prefetch == 2 && !ctx->ice_ready) ? AO_PLAN_TAIL : ctx->launch.ao_chunk;
    HIP_TRY(ctx, build_chunk_table(ctx->stream, ctx->d_params, ctx->grid, mask, ctx->launch.cu_count, plan,
                                   ctx->d_chunk_sums, ctx->d_chunk_begins, ctx->d_chunk_meta, &wet, &n));
    {   // the lists' storage: n chunks at the geometry's fixed stride (grown when a rebuild needs more)
        const size_t want = std::max(wet_list_capacity(ncells), (size_t)(n + 1) * wet_list_stride());
        if (want > ctx->wet_list_entries) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_wet_pos);
            (void)hipFree(ctx->d_trip);
            (void)hipFree(ctx->d_trip_ice);
            (void)hipFree(ctx->d_lean_sorted);
            (void)hipFree(ctx->d_lean_info);
            ctx->d_wet_pos = ctx->d_lean_sorted = nullptr;
            ctx->d_lean_info = nullptr;
            ctx->d_trip = ctx->d_trip_ice = nullptr;
            ctx->wet_list_entries = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_wet_pos, sizeof(uint32_t) * want));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_lean_sorted, sizeof(uint32_t) * want));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_lean_info, sizeof(int) * 4 * (size_t)chunk_table_capacity(ncells)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trip, want));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_trip_ice, want));
            ctx->wet_list_entries = want;
        }
    }
    if (n <= 0 || n + 1 > chunk_table_capacity(ncells)) return fail(ctx, CF_ERR_HIP, "chunk table of %d entries is invalid", n);
    ctx->chunk_mask = key;
    ctx->chunk_mask_kind = d.mask_kind;
    ctx->chunk_z_surface = d.z_surface;
    ctx->chunk_wet = wet;
    ctx->chunk_valid = true;
    ctx->launch.d_chunk_begins = ctx->d_chunk_begins;
    ctx->launch.n_chunks = n;
    {   // which chunks hold cells that read halo rows of the ocean state (HaloRider): the ring rows south of the interior read
        // rows j < 0; the last interior row and the ring rows north of it read row j + 1 ≥ ny (ℑy v)
        std::vector<int> begins((size_t)n + 1);
        HIP_TRY(ctx, hipMemcpyAsync(begins.data(), ctx->d_chunk_begins, sizeof(int) * begins.size(), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const long wx = ctx->grid.nx + 2 * ctx->grid.ring, south_cells = (long)ctx->grid.ring * wx,
                   north_from = (long)(ctx->grid.ny - 1 + ctx->grid.ring) * wx;
        int cs = 0, cn = n;
        while (cs < n && begins[cs] < south_cells) ++cs;
        while (cn > 0 && begins[cn] > north_from) --cn;
        ctx->launch.chunk_south = cs;
        ctx->launch.chunk_north = cn;
    }
    int overflow = 0;
    HIP_TRY(ctx, build_wet_lists(ctx->stream, ctx->d_params, ctx->grid, mask, n, ctx->d_chunk_begins, ctx->d_wet_pos,
                                 ctx->d_trip, ctx->d_chunk_meta, &overflow));
    ctx->launch.d_wet_pos = overflow ? nullptr : ctx->d_wet_pos;
    // the lean ocean kernel's lists: the static lists in index order until the first call has run
    ctx->launch.d_lean_sorted = nullptr;
    ctx->launch.d_lean_info = nullptr;
    if (!overflow) HIP_TRY(ctx, build_lean_lists(ctx->stream, n, ctx->d_wet_pos, ctx->d_chunk_begins, ctx->d_lean_sorted, ctx->d_lean_info));
    ctx->launch.d_lean_sorted = ctx->d_lean_sorted;
    ctx->launch.d_lean_info = ctx->d_lean_info;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_trip_ice, 0, (size_t)n * wet_list_stride(), ctx->stream));
    ctx->launch.d_trip = ctx->trip_hints ? ctx->d_trip : nullptr;
    if (std::getenv("COFLUX_DEBUG"))
        std::fprintf(stderr, "[coflux] chunk table: %d chunks of %d wet cells (%d CUs)\n", n, wet, ctx->launch.cu_count);
    return CF_OK;
}

static int install_params(cf_ctx* ctx, const cf_flux_params* params) {
    DevParams d;
    int rc = lower_params(ctx, params, &d);
    if (rc != CF_OK) return rc;
    if (ctx->tables_kind != d.stability) {
        std::vector<double> t = build_solver_tables(d.stability);
        if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, CF_ERR_HIP, "hipSetDevice(%d) failed", ctx->device);
        if (!ctx->d_tables && hipMalloc((void**)&ctx->d_tables, sizeof(double) * TABLE_DOUBLES) != hipSuccess)
            return fail(ctx, CF_ERR_HIP, "hipMalloc of the solver tables failed");
        if (hipMemcpy(ctx->d_tables, t.data(), sizeof(double) * TABLE_DOUBLES, hipMemcpyHostToDevice) != hipSuccess)
            return fail(ctx, CF_ERR_HIP, "upload of the solver tables failed");
        ctx->tables_kind = d.stability;
        ctx->launch.d_tables = ctx->d_tables;
    }
    ctx->params = *params;
    ctx->dev = d;
    ctx->fast = loop_params(*params, d);
    ctx->fast.cert_budget = ctx->certified_budget / CERT_SAFETY;
    if (!ctx->d_params && hipMalloc((void**)&ctx->d_params, sizeof(DevParams)) != hipSuccess)
        return fail(ctx, CF_ERR_HIP, "hipMalloc of the device parameter block failed");
    if (hipMemcpy(ctx->d_params, &d, sizeof(DevParams), hipMemcpyHostToDevice) != hipSuccess)
        return fail(ctx, CF_ERR_HIP, "upload of the device parameter block failed");
    ctx->launch.d_params = ctx->d_params;
    ctx->chunk_valid = false;
    return CF_OK;
}

void cf_peer_forget_local(const char* mailbox);  // coflux_steps.cpp

extern "C" {

int cf_version(void) { return CF_ABI_VERSION; }

#ifndef CF_BUILD_STAMP
#define CF_BUILD_STAMP "unstamped"
#endif
const char* cf_build_stamp(void) { return CF_BUILD_STAMP; }

const char* cf_last_error(const cf_ctx* ctx) {
    if (!ctx) return g_error.c_str();
    static thread_local std::string copy;  // the text may be rewritten by another thread (window reader) after we return
    {
        std::lock_guard<std::mutex> lock(const_cast<cf_ctx*>(ctx)->error_mutex);
        copy = ctx->error;
    }
    return copy.c_str();
}

int cf_default_flux_params(cf_flux_params* p) {
    if (!p) return fail(nullptr, CF_ERR_INVALID, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->abi_version = CF_ABI_VERSION;
    p->similarity_form = CF_SIMILARITY_LOGARITHMIC;
    p->stability_functions = CF_STABILITY_EDSON2013;
    p->stop_kind = CF_STOP_CONVERGENCE;
    p->maxiter = 100;
    p->velocity_difference = CF_VELOCITY_RELATIVE;
    p->mask_kind = CF_MASK_U8;
    p->tolerance = 1e-8;
    p->von_karman = 0.4;
    p->gustiness_parameter = 1.0;
    p->minimum_gustiness = 0.2;
    p->shear_gustiness_coefficient = 0.0;
    p->similarity_profile_floor = 1.0;
    const double nu0 = 1.326e-5;
    cf_roughness m{};
    m.kind = CF_ROUGHNESS_CHARNOCK;
    m.viscosity_kind = CF_VISCOSITY_TEMPERATURE_DEPENDENT;
    m.maximum_length = 1.0;
    m.charnock = 0.02; /* omip_simulation.jl:263 */
    m.laminar = 0.11;
    m.viscosity[0] = nu0;
    m.viscosity[1] = nu0 * 6.542e-3;
    m.viscosity[2] = nu0 * 8.301e-6;
    m.viscosity[3] = -nu0 * 4.84e-9;
    cf_roughness s = m;
    s.kind = CF_SCALAR_ROUGHNESS_REYNOLDS;
    s.maximum_length = 1.6e-4;
    s.charnock = 0;
    s.laminar = 0;
    s.reynolds_A = 5.85e-5;
    s.reynolds_b = 0.72;
    p->momentum_roughness = m;
    p->temperature_roughness = s;
    p->water_vapor_roughness = s;
    p->reference_height = 10.0;
    p->boundary_layer_height = 600.0;
    p->gravitational_acceleration = 9.81;
    p->thermo = cf_thermodynamics{8.3144598, 0.02897, 0.018015, 2.0 / 7.0, 1859.0,  4181.0, 2100.0, 2500800.0,
                                  2834400.0, 273.16,  273.16,   611.657,   273.15, 233.0,  1.0};
    p->seawater.water_molar_mass = 18.02;
    const double mm[4] = {35.45, 22.99, 96.06, 24.31}, mf[4] = {0.56, 0.31, 0.08, 0.05};
    for (int k = 0; k < 4; ++k) {
        p->seawater.constituent_molar_mass[k] = mm[k];
        p->seawater.constituent_mass_fraction[k] = mf[k];
    }
    p->ocean_reference_density = 1026.0;
    p->ocean_heat_capacity = 3991.86795711963;
    p->ocean_freshwater_density = 1000.0;
    p->ocean_temperature_offset = 273.15;
    p->ocean_minimum_salinity = 0.0;
    p->ocean_surface_z = -150.0;
    p->ocean_albedo_kind = CF_ALBEDO_CONSTANT;
    p->penetrating_shortwave = 1;
    p->ocean_albedo = 0.06;
    p->ocean_albedo_diffuse = 0.069;
    p->ocean_albedo_direct = 0.011;
    p->ocean_emissivity = 1.0;
    p->stefan_boltzmann = 5.67e-8;
    p->flux_formulation = CF_FORMULATION_SIMILARITY;
    p->ly_minimum_wind = 0.5;
    p->ly_zeta_bound = 10.0;
    p->ly_cd[0] = 2.70;
    p->ly_cd[1] = 0.142;
    p->ly_cd[2] = 0.0764;
    p->ly_cd[3] = -3.14807e-10;
    p->ly_high_wind = 33.0;
    p->ly_cd_high = 2.34;
    p->ly_ce = 34.6;
    p->ly_ch_stable = 18.0;
    p->ly_ch_unstable = 32.7;
    return CF_OK;
}

int cf_create(cf_ctx** out, int device, const cf_grid* grid, const cf_flux_params* params) {
    if (!out || !grid) return fail(nullptr, CF_ERR_INVALID, "cf_create: NULL argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, CF_ERR_NODEVICE, "no HIP device available (%s); libcoflux has no CPU backend",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, CF_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
    if (grid->nx <= 0 || grid->ny <= 0 || grid->hx < 0 || grid->hy < 0 || grid->ring < 0 || grid->ring > 1)
        return fail(nullptr, CF_ERR_INVALID, "invalid grid: nx=%d ny=%d hx=%d hy=%d ring=%d", grid->nx, grid->ny,
                    grid->hx, grid->hy, grid->ring);
    if (grid->hx < grid->ring + 1 || grid->hy < grid->ring + 1)
        return fail(nullptr, CF_ERR_INVALID, "halo (%d,%d) too small: need >= ring+1 = %d for the face stencils",
                    grid->hx, grid->hy, grid->ring + 1);
    if ((long long)(grid->nx + 2 * grid->ring) * (grid->ny + 2 * grid->ring) >= (1LL << 24))
        return fail(nullptr, CF_ERR_INVALID, "surface of %d x %d cells: the kernels index a surface with fewer than 2^24 cells per context "
                    "(shard it by latitude slabs)", grid->nx, grid->ny);
    if ((long long)(grid->nx + 2 * grid->hx) * (grid->ny + 2 * grid->hy) >= (1LL << 29))
        return fail(nullptr, CF_ERR_INVALID, "fields of %d x %d doubles with halos (%d, %d): the kernels address a field with 32-bit byte "
                    "offsets (fewer than 2^29 doubles)", grid->nx, grid->ny, grid->hx, grid->hy);
    cf_ctx* ctx = new cf_ctx();   // (every validation of the grid is above: nothing below returns without deleting it)
    ctx->device = device;
    ctx->grid = GridDesc{grid->nx, grid->ny, grid->hx, grid->hy, grid->ring, grid->nx + 2 * grid->hx};
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return fail(nullptr, CF_ERR_HIP, "cannot create a stream on device %d", device);
    }
    cf_flux_params defaults;
    if (!params) {
        cf_default_flux_params(&defaults);
        params = &defaults;
    }
    int rc = install_params(ctx, params);
    if (rc != CF_OK) {
        g_error = ctx->error;
        cf_destroy(ctx);
        return rc;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        ctx->launch.cu_count = prop.multiProcessorCount;
        ctx->launch.latency_layout = 1;  // CF_OPT_LATENCY_LAYOUT: automatic
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return CF_OK;
}

int cf_destroy(cf_ctx* ctx) {
    if (!ctx) return CF_OK;
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    if (ctx->comm_stream) {
        (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamDestroy(ctx->comm_stream);
        (void)hipEventDestroy(ctx->ev_main_idle);
        (void)hipEventDestroy(ctx->ev_comm_done);
    }
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
        (void)hipEventDestroy(ctx->ev_aux_gate);
        for (auto& p : ctx->prefetch)
            if (p.done) (void)hipEventDestroy(p.done);
    }
    if (ctx->peer_south_mapped) (void)hipIpcCloseMemHandle(ctx->peer.south);
    if (ctx->peer_north_mapped) (void)hipIpcCloseMemHandle(ctx->peer.north);
    if (ctx->peer.mine) {
        cf_peer_forget_local(ctx->peer.mine);
        (void)hipFree(ctx->peer.mine);
    }
    if (ctx->d_peer_status) (void)hipFree(ctx->d_peer_status);
    if (ctx->d_halo_counters) (void)hipFree(ctx->d_halo_counters);
    if (ctx->d_trip) (void)hipFree(ctx->d_trip);
    if (ctx->d_trip_ice) (void)hipFree(ctx->d_trip_ice);
    if (ctx->d_wet_pos) (void)hipFree(ctx->d_wet_pos);
    if (ctx->d_lean_sorted) (void)hipFree(ctx->d_lean_sorted);
    if (ctx->d_lean_info) (void)hipFree(ctx->d_lean_info);
    if (ctx->d_chunk_sums) (void)hipFree(ctx->d_chunk_sums);
    if (ctx->d_chunk_begins) (void)hipFree(ctx->d_chunk_begins);
    if (ctx->d_chunk_meta) (void)hipFree(ctx->d_chunk_meta);
    if (ctx->d_ice_albedo) (void)hipFree(ctx->d_ice_albedo);
    if (ctx->d_ice_tables) (void)hipFree(ctx->d_ice_tables);
    if (ctx->d_ice_params) (void)hipFree(ctx->d_ice_params);
    if (ctx->d_reduce) (void)hipFree(ctx->d_reduce);
    if (ctx->d_params) (void)hipFree(ctx->d_params);
    if (ctx->own_stream) {
        hipSetDevice(ctx->device);
        hipStreamSynchronize(ctx->own_stream);
        hipStreamDestroy(ctx->own_stream);
    }
    delete ctx;
    return CF_OK;
}

int cf_set_flux_params(cf_ctx* ctx, const cf_flux_params* params) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // tables may be rewritten
    return install_params(ctx, params);
}

int cf_set_option(cf_ctx* ctx, int option, int value) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    switch (option) {
        case CF_OPT_SOLVER:
            if (value != CF_SOLVER_TABLES && value != CF_SOLVER_LIBM) return fail(ctx, CF_ERR_INVALID, "unknown solver %d", value);
            ctx->launch.solver = value;
            return CF_OK;
        case CF_OPT_INTERP_TILE_CAP:
            if (!experiment_knob("COFLUX_EXPERIMENTS")) return fail(ctx, CF_ERR_INVALID, "CF_OPT_INTERP_TILE_CAP is an experiment option: start the process with COFLUX_EXPERIMENTS=1");
            // 4 waves × 9 variables × cap × 8 B of dynamic LDS must fit a workgroup's 64 KB
            if (value != 0 && (value < 16 || value > 224)) return fail(ctx, CF_ERR_INVALID, "interp tile cap %d: 0 (LDS-free gather kernel) or 16…224", value);
            ctx->launch.interp_cap = value;
            return CF_OK;
        case CF_OPT_TRIP_HINTS:
            if (value < 0 || value > 3) return fail(ctx, CF_ERR_INVALID, "trip hints %d: 0 (off), 1 (on), 2 (automatic), 3 (on, the lean kernel in quarter-chunk windows)", value);
            if (ctx->lean_hints && value != 1 && value != 3) ctx->chunk_valid = false;  // the lean kernel's lists go back to index order
            if (ctx->launch.lean_hints != (value == 1 ? 1 : (value == 3 ? 4 : 0))) ctx->chunk_valid = false;  // another window layout: start from index order
            ctx->trip_hints = value != 0;
            ctx->lean_hints = value == 1 || value == 3;
            ctx->launch.d_trip = ctx->trip_hints ? ctx->d_trip : nullptr;
            ctx->launch.lean_hints = value == 1 ? 1 : (value == 3 ? 4 : 0);
            return CF_OK;
        case CF_OPT_FUSED_NET:
            if (value < 0 || value > 2) return fail(ctx, CF_ERR_INVALID, "fused net fluxes %d: 0 (never), 1 (when possible), 2 (automatic)", value);
            ctx->fused_net = value;
            return CF_OK;
        case CF_OPT_MERGED_PREFETCH:
            if (value < 0 || value > 2)
                return fail(ctx, CF_ERR_INVALID, "merged prefetch %d: 0 (auxiliary stream), 1 (in the face-stress launch), 2 (tail workgroups of the solver launch)", value);
            if ((ctx->merged_prefetch == 2) != (value == 2)) ctx->chunk_valid = false;   // the solver's chunk plan follows (next call rebuilds)
            ctx->merged_prefetch = value;
            return CF_OK;
        case CF_OPT_ICE_ORBIT_SHORTCUT:
            ctx->ice_orbit_shortcut = value != 0;
            ctx->ice_kernel.orbit_shortcut = value != 0 ? 1.0 : 0.0;
            return CF_OK;
        case CF_OPT_ICE_FREE_CELLS:
            if (value != CF_ICE_FREE_ITERATE && value != CF_ICE_FREE_ZERO) return fail(ctx, CF_ERR_INVALID, "ice-free cells %d: 0 (iterate) or 1 (zero)", value);
            ctx->ice_free_zero = value == CF_ICE_FREE_ZERO;
            ctx->ice_kernel.ice_free_zero = ctx->ice_free_zero ? 1.0 : 0.0;
            return CF_OK;
        case CF_OPT_SOLVER_PATH:
            if (value != CF_SOLVER_PATH_EXACT && value != CF_SOLVER_PATH_CERTIFIED) return fail(ctx, CF_ERR_INVALID, "solver path %d: 0 (exact) or 1 (certified)", value);
            ctx->launch.certified = value;
            return CF_OK;
        case CF_OPT_HALO_IN_SOLVER_LAUNCH:
            if (value < 0 || value > 1) return fail(ctx, CF_ERR_INVALID, "halo rows in the solver launch %d: 0 (the exchange kernel of its own) or 1", value);
            if (value && !ctx->d_halo_counters) {
                HIP_TRY(ctx, hipSetDevice(ctx->device));
                HIP_TRY(ctx, hipMalloc((void**)&ctx->d_halo_counters, 4 * sizeof(unsigned long long)));
                HIP_TRY(ctx, hipMemset(ctx->d_halo_counters, 0, 4 * sizeof(unsigned long long)));
            }
            ctx->halo_in_launch = value;
            return CF_OK;
        case CF_OPT_LATENCY_LAYOUT:
            if (value < 0 || value > 2) return fail(ctx, CF_ERR_INVALID, "latency layout %d: 0 (never), 1 (automatic), 2 (always)", value);
            ctx->launch.latency_layout = value;
            return CF_OK;
        case CF_OPT_CERTIFIED_BUDGET:
            if (value < 50 || value > 1000000) return fail(ctx, CF_ERR_INVALID, "certified budget %d: 50 … 1000000 (units of 1e-9)", value);
            ctx->certified_budget = value * 1e-9;
            ctx->fast.cert_budget = ctx->certified_budget / CERT_SAFETY;
            return CF_OK;
        case CF_OPT_AO_CHUNK:
            if (!experiment_knob("COFLUX_EXPERIMENTS")) return fail(ctx, CF_ERR_INVALID, "CF_OPT_AO_CHUNK is an experiment option: start the process with COFLUX_EXPERIMENTS=1");
            if (value != 0 && value != 256 && value != 512 && value != 768 && value != 1024 && value != 1280)
                return fail(ctx, CF_ERR_INVALID, "solver chunk %d: wet cells per workgroup must be 0 (automatic), 256, 512, 768, 1024 or 1280 "
                            "(uniform size)", value);
            ctx->launch.ao_chunk = value;
            ctx->chunk_valid = false;
            return CF_OK;
        default: return fail(ctx, CF_ERR_INVALID, "unknown option %d", option);
    }
}

int cf_debug_eval(cf_ctx* ctx, int function, int n, const double* d_x, double* d_y) {
    if (!ctx || !d_x || !d_y || n < 0) return fail(ctx, CF_ERR_INVALID, "cf_debug_eval: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    HIP_TRY(ctx, launch_debug_eval(ctx->stream, ctx->launch, function, n, d_x, d_y));
    return CF_OK;
}

int cf_set_stream(cf_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    hipStream_t next = hip_stream == CF_STREAM_LEGACY ? nullptr  // the null stream handle: legacy default-stream semantics in every HIP call
                       : (hip_stream ? (hipStream_t)hip_stream : ctx->own_stream);
    if (next != ctx->stream) {
        // a requested-ahead atmosphere state that was launched ON the old stream is consumed by stream order alone (no event):
        // the new stream has to wait for it
        bool pending = false;
        for (auto& p : ctx->prefetch) pending |= p.valid && p.on_main;
        if (pending) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            hipEvent_t ev = nullptr;
            HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(ev, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(next, ev, 0));
            HIP_TRY(ctx, hipEventDestroy(ev));
        }
    }
    ctx->stream = next;
    return CF_OK;
}

int cf_sync(cf_ctx* ctx) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    if (int rc = wait_for_halos(ctx)) return rc;
    CHECK(cf_flush_deferred_prefetch(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->aux_stream));
    if (ctx->d_peer_status && ctx->peer_seq) {  // a peer-direct exchange whose neighbour never arrived
        int st = 0;
        HIP_TRY(ctx, hipMemcpy(&st, ctx->d_peer_status, sizeof st, hipMemcpyDeviceToHost));
        if (st) {
            // reported ONCE: the caller may fall back to another exchange (bench.py does) and go on with this context — the
            // mailbox protocol's sequence numbers only grow, a late arrival of the missed step disturbs nothing
            HIP_TRY(ctx, hipMemset(ctx->d_peer_status, 0, sizeof st));
            return fail(ctx, CF_ERR_COMM, "peer-direct halo exchange timed out waiting for the %s neighbour's rows",
                        st == 1 ? "south" : "north");
        }
    }
    return CF_OK;
}

void* cf_device_alloc(cf_ctx* ctx, size_t bytes) {
    if (!ctx) return nullptr;
    void* p = nullptr;
    hipSetDevice(ctx->device);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        fail(ctx, CF_ERR_HIP, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

int cf_device_free(cf_ctx* ctx, void* d_ptr) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipFree(d_ptr));
    return CF_OK;
}

int cf_h2d(cf_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

int cf_d2h(cf_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CF_OK;
}

// ---- argument checks ------------------------------------------------------------------------
static int check_source(cf_ctx* ctx, const cf_atmos_source* s) {
    if (!s) return fail(ctx, CF_ERR_INVALID, "atmosphere source is NULL");
    for (int v = 0; v < CF_JRA55_NVARS; ++v)
        if (!s->data[v]) return fail(ctx, CF_ERR_INVALID, "atmosphere source variable %d is NULL", v);
    if (s->ns_x <= 0 || s->ns_y <= 0 || s->n_levels <= 0) return fail(ctx, CF_ERR_INVALID, "invalid source shape");
    if (s->level1 < 0 || s->level1 >= s->n_levels || s->level2 < 0 || s->level2 >= s->n_levels)
        return fail(ctx, CF_ERR_INVALID, "time levels (%d,%d) outside the %d levels in memory", s->level1, s->level2,
                    s->n_levels);
    return CF_OK;
}
static int check_weights(cf_ctx* ctx, const cf_interp_weights* w) {
    if (!w || !w->fi || !w->fj) return fail(ctx, CF_ERR_INVALID, "interpolation weights (fi, fj) are NULL");
    if ((w->cos_rot == nullptr) != (w->sin_rot == nullptr))
        return fail(ctx, CF_ERR_INVALID, "cos_rot and sin_rot must both be given or both be NULL");
    return CF_OK;
}
static int check_exchange(cf_ctx* ctx, const cf_exchange_fields* e, bool all) {
    if (!e || !e->u || !e->v || !e->T || !e->p || !e->q) return fail(ctx, CF_ERR_INVALID, "exchange fields u,v,T,p,q are NULL");
    if (all && (!e->Qs || !e->Ql || !e->Mp)) return fail(ctx, CF_ERR_INVALID, "exchange fields Qs,Ql,Mp are NULL");
    return CF_OK;
}
static int check_ocean(cf_ctx* ctx, const cf_ocean_surface* o) {
    if (!o || !o->T || !o->S || !o->u || !o->v) return fail(ctx, CF_ERR_INVALID, "ocean surface fields are NULL");
    if (ctx->dev.mask_kind != CF_MASK_NONE && !o->mask)
        return fail(ctx, CF_ERR_INVALID, "mask_kind = %d but ocean mask is NULL", ctx->dev.mask_kind);
    return CF_OK;
}
static int check_fluxes(cf_ctx* ctx, const cf_interface_fluxes* f) {
    if (!f || !f->sensible_heat || !f->latent_heat || !f->water_vapor || !f->x_momentum || !f->y_momentum || !f->temperature)
        return fail(ctx, CF_ERR_INVALID, "interface flux fields are NULL");
    return CF_OK;
}
static int check_net(cf_ctx* ctx, const cf_net_ocean_fluxes* n, const cf_interp_weights* w) {
    if (!n || !n->u || !n->v || !n->T || !n->S) return fail(ctx, CF_ERR_INVALID, "net ocean flux fields are NULL");
    if (ctx->dev.albedo_kind == CF_ALBEDO_LATITUDE_DEPENDENT && (!w || !w->latitude))
        return fail(ctx, CF_ERR_INVALID, "latitude-dependent albedo needs cf_interp_weights.latitude");
    return CF_OK;
}

// Ocean-reading kernels must see the halo rows of a preceding cf_halo_exchange_rows.
static int wait_for_halos(cf_ctx* ctx) {
    if (ctx->comm_pending) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_comm_done, 0));
        ctx->comm_pending = false;
    }
    return CF_OK;
}

// (coflux_steps.cpp: the table is built before the first halo exchange of a step loop is queued — building it
// synchronises the stream, which must not hold a peer-direct exchange that waits for a neighbour still to be launched)
int cf_ensure_chunk_table(cf_ctx* ctx, const void* mask) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return ensure_chunk_table(ctx, mask);
}

// A requested interpolation that rides in another launch stages its JRA55 tile (4 waves × 9 variables × cap × 8 B) in THAT
// launch's dynamic LDS: a solver launch has ≥ 40 960 B (CF_OPT_INTERP_TILE_CAP ≤ 142), the face-stress launch 64 KB.  A cap
// beyond that falls through to the stand-alone interpolation on the main stream.
static bool interp_tile_fits(const cf_ctx* ctx, size_t lds_bytes) {
    return ctx->launch.interp_cap > 0 && (size_t)4 * CF_JRA55_NVARS * (size_t)ctx->launch.interp_cap * sizeof(double) <= lds_bytes;
}

static bool net_fluxes_fused(const cf_ctx* ctx) {
    const bool lean = ctx->fast.specialization == SOLVER_OCEAN_LEAN && ctx->launch.solver == CF_SOLVER_TABLES;
    // (CoefficientBasedFluxes runs a fixed trip count: its lists stay in index order, the epilogue's accesses coalesced —
    // measured cf_update_state 71.8 → 63.1 µs)
    const bool fixed_trips = ctx->fast.specialization == SOLVER_LY && ctx->launch.solver == CF_SOLVER_TABLES;
    return (ctx->fused_net == 1 || (ctx->fused_net == 2 && (lean || fixed_trips))) && ctx->launch.solver == CF_SOLVER_TABLES &&
           ctx->dev.albedo_kind == CF_ALBEDO_CONSTANT;
}

int cf_solver_iteration_path(cf_ctx* ctx, int* path) {
    if (!ctx || !path) return fail(ctx, CF_ERR_INVALID, "cf_solver_iteration_path: bad arguments");
    const bool lean = ctx->fast.specialization == SOLVER_OCEAN_LEAN && ctx->launch.solver == CF_SOLVER_TABLES;
    // the predicate launch_ao_fluxes_lean itself decides on.  One launch runs the exact body whatever this says: with sea ice
    // and CF_OPT_MERGED_PREFETCH = 2 the ocean solve rides in the interface solve's launch (ice_ocean_kernel), whose rider is the
    // exact kernel — cf_update_state_sea_ice
    *path = lean && lean_certified_applies(ctx->launch, ctx->fast) ? CF_SOLVER_PATH_CERTIFIED : CF_SOLVER_PATH_EXACT;
    return CF_OK;
}

int cf_solver_latency_layout(cf_ctx* ctx, int* layout) {
    if (!ctx || !layout) return fail(ctx, CF_ERR_INVALID, "cf_solver_latency_layout: bad arguments");
    const bool lean = ctx->fast.specialization == SOLVER_OCEAN_LEAN && ctx->launch.solver == CF_SOLVER_TABLES;
    *layout = lean && ctx->chunk_valid && lean_line_applies(ctx->launch, ctx->fast, ctx->dev.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC) ? 1 : 0;
    return CF_OK;
}

int cf_solver_path(cf_ctx* ctx, int* lean_kernel, int* fused_net) {
    if (!ctx || !lean_kernel || !fused_net) return fail(ctx, CF_ERR_INVALID, "cf_solver_path: bad arguments");
    *lean_kernel = ctx->fast.specialization == SOLVER_OCEAN_LEAN && ctx->launch.solver == CF_SOLVER_TABLES;
    *fused_net = net_fluxes_fused(ctx) ? 1 : 0;
    return CF_OK;
}

int cf_interpolate_atmosphere_state(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                                    const cf_exchange_fields* out) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    CHECK(check_source(ctx, src));
    CHECK(check_weights(ctx, w));
    CHECK(check_exchange(ctx, out, true));
    HIP_TRY(ctx, launch_interpolate(ctx->stream, ctx->launch, ctx->grid, src, w, out));
    return CF_OK;
}

int cf_compute_atmosphere_ocean_fluxes(cf_ctx* ctx, const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                                       const cf_interface_fluxes* out) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    CHECK(check_ocean(ctx, ocean));
    CHECK(check_exchange(ctx, atmos, false));
    CHECK(check_fluxes(ctx, out));
    CHECK(wait_for_halos(ctx));
    if (ctx->launch.solver == CF_SOLVER_LIBM && ctx->params.flux_formulation == CF_FORMULATION_LARGE_YEAGER)
        return fail(ctx, CF_ERR_INVALID, "CF_SOLVER_LIBM implements SimilarityTheoryFluxes only");
    CHECK(ensure_chunk_table(ctx, ocean->mask));
    HIP_TRY(ctx, launch_ao_fluxes(ctx->stream, ctx->launch, ctx->dev, ctx->fast, ctx->grid, ocean, atmos, out));
    return CF_OK;
}

int cf_compute_net_ocean_fluxes(cf_ctx* ctx, const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                                const cf_interface_fluxes* fluxes, const cf_sea_ice_fields* ice,
                                const cf_interp_weights* w, const cf_net_ocean_fluxes* out) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    CHECK(check_ocean(ctx, ocean));
    CHECK(check_exchange(ctx, atmos, true));
    CHECK(check_fluxes(ctx, fluxes));
    CHECK(check_net(ctx, out, w));
    CHECK(wait_for_halos(ctx));
    HIP_TRY(ctx, launch_net_fluxes(ctx->stream, ctx->dev, ctx->grid, ocean, atmos, fluxes, ice, w, out, ctx->d_land_freshwater));
    return CF_OK;
}

// A deferred next-step interpolation (cf_prefetch_atmosphere_state) that has just been launched on the MAIN stream — as the
// solver launch's tail workgroups, inside the face-stress launch, or as a plain launch: stream order replaces the event.
int deferred_went_out_on_main(cf_ctx* ctx) {
    cf_ctx::Prefetch* slot = nullptr;
    for (auto& p : ctx->prefetch)
        if (p.valid && p.key == ctx->deferred.out.u) slot = &p;
    if (!slot)
        for (auto& p : ctx->prefetch)
            if (!p.valid) slot = &p;
    if (!slot) return fail(ctx, CF_ERR_INVALID, "two prefetched atmosphere states are already pending");
    slot->key = ctx->deferred.out.u;
    slot->level1 = ctx->deferred.src.level1;
    slot->level2 = ctx->deferred.src.level2;
    slot->tf = ctx->deferred.src.time_fraction;
    slot->valid = true;
    slot->on_main = true;
    ctx->deferred.valid = false;
    return CF_OK;
}

// hold_tail_work (cf_update_state_sea_ice with CF_OPT_MERGED_PREFETCH = 2): the face stresses and a requested next-step
// interpolation are NOT launched here — they ride in the tail of the sea-ice interface launch that follows, whose workgroups
// retire over a much longer span than the ocean solver's; *stress_held tells the caller whether the stresses are still due.
static int update_state_impl(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                             const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                             const cf_interface_fluxes* fluxes, const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net,
                             bool hold_tail_work, bool* stress_held, OceanRider* ocean_rider = nullptr) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    CHECK(check_source(ctx, src));
    CHECK(check_weights(ctx, w));
    CHECK(check_ocean(ctx, ocean));
    CHECK(check_exchange(ctx, atmos, true));
    CHECK(check_fluxes(ctx, fluxes));
    CHECK(check_net(ctx, net, w));
    // interpolate → solver → net fluxes, stream-ordered.  (Fusing the interpolation into the solver
    // saves the 40 B/cell re-read of the atmosphere state — ≈ 6 µs — but ties the FP64-issue-bound
    // solver to the interpolation's tile geometry and LDS footprint; measured slower, see DESIGN.md.)
    CHECK(ensure_chunk_table(ctx, ocean->mask));
    const bool rec = ctx->prof_count < ctx->prof_capacity;
    hipEvent_t* ev = rec ? &ctx->prof_events[4 * (size_t)ctx->prof_count] : nullptr;
    if (rec) HIP_TRY(ctx, hipEventRecord(ev[0], ctx->stream));
    // a prefetched atmosphere state (cf_prefetch_atmosphere_state) for exactly this step and this set of exchange
    // fields is already on its way on the auxiliary stream: wait for it instead of interpolating again
    bool prefetched = false;
    if (ctx->deferred.valid && ctx->deferred.out.u == atmos->u) CHECK(cf_flush_deferred_prefetch(ctx));  // asked for THIS step
    for (auto& p : ctx->prefetch) {
        if (!p.valid || p.key != atmos->u) continue;
        if (!p.on_main) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p.done, 0));  // also orders a mismatched prefetch before our writes
        prefetched = p.level1 == src->level1 && p.level2 == src->level2 && p.tf == src->time_fraction;
        p.valid = false;
    }
    // Fused forms.  Net fluxes: the cell-local part of compute_net_ocean_fluxes! (everything but the two face stresses,
    // which need the west / south neighbour's ρτ) is computed in the solver's epilogue from registers, and a thin stress
    // kernel follows — bitwise the same numbers as the three-launch sequence (shared arithmetic, contraction off).
    const bool fuse = net_fluxes_fused(ctx);
    if (!prefetched) HIP_TRY(ctx, launch_interpolate(ctx->stream, ctx->launch, ctx->grid, src, w, atmos));
    if (rec) HIP_TRY(ctx, hipEventRecord(ev[1], ctx->stream));
    CHECK(wait_for_halos(ctx));  // the interpolation above overlapped the halo rows
    // CF_OPT_MERGED_PREFETCH = 2: a requested next-step interpolation becomes the TAIL workgroups of this solver launch
    const bool tail_lean = ctx->launch.d_lean_info && ctx->fast.specialization == SOLVER_OCEAN_LEAN && ctx->launch.solver == CF_SOLVER_TABLES;
    const bool tail_ly = ctx->fast.specialization == SOLVER_LY && ctx->launch.solver == CF_SOLVER_TABLES;
    const bool tail = fuse && !hold_tail_work && ctx->merged_prefetch == 2 && ctx->deferred.valid &&
                      interp_tile_fits(ctx, 40960) && ctx->deferred.out.u != atmos->u && (tail_lean || tail_ly);
    // With sea ice (cf_update_state_sea_ice): the ocean solve itself is handed to the caller, whose interface-solve launch carries
    // its workgroups behind its own (ice_ocean_kernel) — the stresses then follow that launch
    const bool ride = hold_tail_work && ocean_rider && fuse && tail_lean && !rec;
    // the step's peer-direct halo rows (cf_time_steps left them as a request): riders of this solver launch where it is the
    // stepping loop's exact-path launch with tail workgroups, else the exchange kernel of its own, now, in front of the solver
    HaloRider halo_rider{};
    const HaloRider* halo = nullptr;
    if (ctx->halo_request.valid) {
        ctx->halo_request.valid = false;
        if (tail && tail_lean && !ride && ctx->d_halo_counters && lean_halo_rides(ctx->launch, ctx->fast)) {
            HaloRider& H = halo_rider;
            H.M = ctx->peer;
            H.F = ctx->halo_request.F;
            H.counters = ctx->d_halo_counters;
            H.seq = ++ctx->peer_seq;
            ++ctx->halo_in_launch_count;
            for (int dir = 0; dir < 2; ++dir)
                if ((dir == 0 ? ctx->peer.south : ctx->peer.north) != nullptr) {
                    ctx->halo_expect_sent[dir] += (unsigned long long)H.F.n;
                    ctx->halo_expect_done[dir] += (unsigned long long)H.F.n;
                }
            for (int dir = 0; dir < 2; ++dir) {
                H.expect_sent[dir] = ctx->halo_expect_sent[dir];
                H.expect_done[dir] = ctx->halo_expect_done[dir];
            }
            H.status = ctx->d_peer_status;
            H.rows = ctx->halo_request.rows;
            H.blocks = 2 * H.F.n;
            H.chunk_south = ctx->launch.chunk_south;
            H.chunk_north = ctx->launch.chunk_north;
            H.wait_south = ctx->peer.south != nullptr;
            H.wait_north = ctx->peer.north != nullptr;
            halo = &halo_rider;
        } else {
            CHECK(cf_peer_halo_launch_now(ctx, &ctx->halo_request.F, ctx->halo_request.rows));
        }
    }
    if (ride) {
        HIP_TRY(ctx, make_ocean_rider(ctx->launch, ctx->dev, ctx->fast, ctx->grid, ocean, atmos, fluxes, ice, net, ctx->d_land_freshwater,
                                      ocean_rider));
    } else if (tail) {
        int rows = 4, blocks = 1;
        interpolate_grid(ctx->launch, ctx->grid, &rows, &blocks);
        static const int tail_cap = [] {  // (experiments: COFLUX_EXPERIMENTS=1 COFLUX_TAIL_BLOCKS=n, read once)
            const char* e = experiment_knob("COFLUX_TAIL_BLOCKS");
            return e ? std::max(1, std::atoi(e)) : 0;
        }();
        if (tail_cap > 0) blocks = std::min(blocks, tail_cap);
        static const int tail_pos = [] {  // (experiments: COFLUX_TAIL_POS = dispatch index of the first interpolation workgroup; default: behind the solver)
            const char* e = experiment_knob("COFLUX_TAIL_POS");
            return e ? std::atoi(e) : -1;
        }();
        if (tail_lean)
            HIP_TRY(ctx, launch_ao_fluxes_lean(ctx->stream, ctx->launch, ctx->dev, ctx->fast, ctx->grid, ocean, atmos, fluxes, ice, net,
                                               ctx->d_land_freshwater, &ctx->deferred.src, &ctx->deferred.w, &ctx->deferred.out, rows, blocks, tail_pos, halo));
        else
            HIP_TRY(ctx, launch_ly_fluxes_with_tail(ctx->stream, ctx->launch, ctx->dev, ctx->fast, ctx->grid, ocean, atmos, fluxes, ice, net,
                                                    ctx->d_land_freshwater, &ctx->deferred.src, &ctx->deferred.w, &ctx->deferred.out, rows, blocks));
        CHECK(deferred_went_out_on_main(ctx));
    } else
    HIP_TRY(ctx, launch_ao_fluxes(ctx->stream, ctx->launch, ctx->dev, ctx->fast, ctx->grid, ocean, atmos, fluxes,
                                  fuse ? ice : nullptr, fuse ? net : nullptr, ctx->d_land_freshwater));
    // A requested next-step interpolation (cf_prefetch_atmosphere_state).  CF_OPT_MERGED_PREFETCH: it rides in THIS step's
    // face-stress launch on the main stream — two independent memory-bound kernels, one launch boundary fewer (on a
    // latitude slab a boundary is a tenth of the step).  Otherwise it goes out on the auxiliary stream right behind the
    // solver: the solver's workgroups are dispatched first, the gather kernel takes what they leave free.
    const bool merge = fuse && !hold_tail_work && ctx->merged_prefetch == 1 && ctx->deferred.valid && interp_tile_fits(ctx, 65536) &&
                       ctx->deferred.out.u != atmos->u;
    if (ctx->merged_prefetch == 0) CHECK(cf_flush_deferred_prefetch(ctx));
    if (rec) HIP_TRY(ctx, hipEventRecord(ev[2], ctx->stream));
    if (merge) {
        HIP_TRY(ctx, launch_interpolate_and_stress(ctx->stream, ctx->launch, ctx->dev, ctx->grid, &ctx->deferred.src, &ctx->deferred.w,
                                                   &ctx->deferred.out, ocean, fluxes, ice, net));
        CHECK(deferred_went_out_on_main(ctx));
    } else if (fuse && hold_tail_work) {
        if (stress_held) *stress_held = true;   // (the caller's next launch carries them)
    } else if (fuse)
        HIP_TRY(ctx, launch_net_stress(ctx->stream, ctx->dev, ctx->grid, ocean, fluxes, ice, net));
    else
        HIP_TRY(ctx, launch_net_fluxes(ctx->stream, ctx->dev, ctx->grid, ocean, atmos, fluxes, ice, w, net, ctx->d_land_freshwater));
    if (rec) {
        HIP_TRY(ctx, hipEventRecord(ev[3], ctx->stream));
        ++ctx->prof_count;
    }
    if (!hold_tail_work && ctx->merged_prefetch != 0 && ctx->deferred.valid && ctx->deferred.out.u != atmos->u) {
        // neither merged form applies to this step (another solver kernel, un-fused net fluxes …): the requested interpolation
        // goes out as a launch of its own on the same stream — the sequence of an un-pipelined step, one step early
        HIP_TRY(ctx, launch_interpolate(ctx->stream, ctx->launch, ctx->grid, &ctx->deferred.src, &ctx->deferred.w, &ctx->deferred.out));
        CHECK(deferred_went_out_on_main(ctx));
    }
    return CF_OK;
}

int cf_update_state(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                    const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                    const cf_interface_fluxes* fluxes, const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net) {
    return update_state_impl(ctx, src, w, ocean, atmos, fluxes, ice, net, false, nullptr);
}

int cf_profile_enable(cf_ctx* ctx, int max_records) {
    if (!ctx || max_records < 0) return fail(ctx, CF_ERR_INVALID, "cf_profile_enable: bad arguments");
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    ctx->prof_capacity = ctx->prof_count = 0;
    ctx->prof_events.resize(4 * (size_t)max_records);
    for (auto& e : ctx->prof_events) HIP_TRY(ctx, hipEventCreate(&e));
    ctx->prof_capacity = max_records;
    return CF_OK;
}

int cf_profile_read(cf_ctx* ctx, int kernel, double* avg_ms, int* records) {
    if (!ctx || !avg_ms || kernel < 0 || kernel > 2) return fail(ctx, CF_ERR_INVALID, "cf_profile_read: bad arguments");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (int n = 0; n < ctx->prof_count; ++n) {
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_events[4 * (size_t)n + kernel],
                                        ctx->prof_events[4 * (size_t)n + kernel + 1]));
        sum += ms;
    }
    *avg_ms = ctx->prof_count ? sum / ctx->prof_count : 0.0;
    if (records) *records = ctx->prof_count;
    return CF_OK;
}

static int run_stage(cf_ctx* ctx, int stage, const cf_atmos_source* src, const cf_interp_weights* w,
                     const cf_ocean_surface* ocean, const cf_exchange_fields* atmos, const cf_interface_fluxes* fluxes,
                     const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net) {
    switch (stage) {
        case CF_STAGE_INTERPOLATE: return cf_interpolate_atmosphere_state(ctx, src, w, atmos);
        case CF_STAGE_AO_FLUXES: return cf_compute_atmosphere_ocean_fluxes(ctx, ocean, atmos, fluxes);
        case CF_STAGE_NET_FLUXES: return cf_compute_net_ocean_fluxes(ctx, ocean, atmos, fluxes, ice, w, net);
        case CF_STAGE_UPDATE_STATE: return cf_update_state(ctx, src, w, ocean, atmos, fluxes, ice, net);
        default: return fail(ctx, CF_ERR_INVALID, "unknown stage %d", stage);
    }
}

int cf_time_stage(cf_ctx* ctx, int stage, int launches, const cf_atmos_source* src, const cf_interp_weights* w,
                  const cf_ocean_surface* ocean, const cf_exchange_fields* atmos, const cf_interface_fluxes* fluxes,
                  const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net, double* ms_per_launch) {
    if (!ctx || !ms_per_launch || launches <= 0) return fail(ctx, CF_ERR_INVALID, "cf_time_stage: bad arguments");
    hipEvent_t t0, t1;
    HIP_TRY(ctx, hipEventCreate(&t0));
    HIP_TRY(ctx, hipEventCreate(&t1));
    CHECK(run_stage(ctx, stage, src, w, ocean, atmos, fluxes, ice, net));  // one untimed launch (code load)
    HIP_TRY(ctx, hipEventRecord(t0, ctx->stream));
    for (int n = 0; n < launches; ++n) CHECK(run_stage(ctx, stage, src, w, ocean, atmos, fluxes, ice, net));
    HIP_TRY(ctx, hipEventRecord(t1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(t1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, t0, t1));
    hipEventDestroy(t0);
    hipEventDestroy(t1);
    *ms_per_launch = (double)ms / launches;
    return CF_OK;
}

int cf_time_copy(cf_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int launches, double* ms_per_launch) {
    if (!ctx || !ms_per_launch || launches <= 0 || !d_dst || !d_src)
        return fail(ctx, CF_ERR_INVALID, "cf_time_copy: bad arguments");
    hipEvent_t t0, t1;
    HIP_TRY(ctx, hipEventCreate(&t0));
    HIP_TRY(ctx, hipEventCreate(&t1));
    HIP_TRY(ctx, launch_copy(ctx->stream, d_dst, d_src, bytes));
    HIP_TRY(ctx, hipEventRecord(t0, ctx->stream));
    for (int n = 0; n < launches; ++n) HIP_TRY(ctx, launch_copy(ctx->stream, d_dst, d_src, bytes));
    HIP_TRY(ctx, hipEventRecord(t1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(t1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, t0, t1));
    hipEventDestroy(t0);
    hipEventDestroy(t1);
    *ms_per_launch = (double)ms / launches;
    return CF_OK;
}

// ---- RCCL halo rows ---------------------------------------------------------------------------
static int load_rccl(cf_ctx* ctx) {
    if (g_rccl.handle) return CF_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    // a copy the host process already holds (torch ships its own librccl.so) comes first: two RCCL instances in one
    // process would each keep their own bootstrap state and shared-memory segments
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) break;
    if (!h && dlsym(RTLD_DEFAULT, "ncclCommInitRank")) h = dlopen(nullptr, RTLD_NOW);
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(ctx, CF_ERR_COMM, "cannot dlopen librccl.so: %s", dlerror());
#define SYM(field, name)                                                     \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                   \
    if (!g_rccl.field) return fail(ctx, CF_ERR_COMM, "librccl lacks %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(AllReduce, "ncclAllReduce");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
    SYM(CommCuDevice, "ncclCommCuDevice");
#undef SYM
    g_rccl.handle = h;
    return CF_OK;
}

#define NCCL_TRY(ctx, expr)                                                                            \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail(ctx, CF_ERR_COMM, "%s:%d: %s: %s", __FILE__, __LINE__, #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

int cf_comm_unique_id(void* id128) {
    if (!id128) return fail(nullptr, CF_ERR_INVALID, "id buffer is NULL");
    static_assert(sizeof(ncclUniqueId) == CF_COMM_ID_BYTES, "ncclUniqueId size");
    CHECK(load_rccl(nullptr));
    NCCL_TRY(nullptr, g_rccl.GetUniqueId((ncclUniqueId*)id128));
    return CF_OK;
}

int cf_comm_init(cf_ctx* ctx, const void* id128, int rank, int nranks) {
    if (!ctx || !id128 || nranks <= 0 || rank < 0 || rank >= nranks)
        return fail(ctx, CF_ERR_INVALID, "cf_comm_init: bad arguments");
    CHECK(load_rccl(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    NCCL_TRY(ctx, g_rccl.CommInitRank(&ctx->comm, nranks, id, rank));
    ctx->rank = rank;
    ctx->nranks = nranks;
    return CF_OK;
}

int cf_comm_count(cf_ctx* ctx, int* nranks, int* rank, int* device) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    if (!ctx->comm) return fail(ctx, CF_ERR_COMM, "cf_comm_init has not been called");
    int n = 0, r = -1, d = -1;
    NCCL_TRY(ctx, g_rccl.CommCount(ctx->comm, &n));
    NCCL_TRY(ctx, g_rccl.CommUserRank(ctx->comm, &r));
    NCCL_TRY(ctx, g_rccl.CommCuDevice(ctx->comm, &d));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return CF_OK;
}

int cf_comm_destroy(cf_ctx* ctx) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    if (ctx->comm) {
        NCCL_TRY(ctx, g_rccl.CommDestroy(ctx->comm));
        ctx->comm = nullptr;
    }
    return CF_OK;
}

int cf_default_sea_ice_params(cf_sea_ice_params* p) {
    if (!p) return fail(nullptr, CF_ERR_INVALID, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->conductivity = 2.0;
    p->consolidation_thickness = 0.05;
    p->maximum_temperature_change = 5.0;
    p->ice_salinity = 4.0;
    p->liquidus_slope = 0.054;
    p->freshwater_melting_temperature = 273.15;
    p->albedo = 0.7;
    p->emissivity = 1.0;
    p->temperature_offset = 273.15;
    return CF_OK;
}

int cf_set_sea_ice_formulation(cf_ctx* ctx, const cf_flux_params* ice_fluxes, const cf_sea_ice_params* ice) {
    if (!ctx || !ice) return fail(ctx, CF_ERR_INVALID, "cf_set_sea_ice_formulation: NULL argument");
    if (ice->struct_size != (int32_t)sizeof(cf_sea_ice_params))
        return fail(ctx, CF_ERR_INVALID, "cf_sea_ice_params.struct_size = %d, library expects %zu", ice->struct_size,
                    sizeof(cf_sea_ice_params));
    if (ice->skin_temperature_scheme != CF_SKIN_EXPLICIT && ice->skin_temperature_scheme != CF_SKIN_SEMI_IMPLICIT)
        return fail(ctx, CF_ERR_INVALID, "Unknown skin_temperature_scheme: %d", ice->skin_temperature_scheme);
    if (!(ice->conductivity > 0) || !(ice->maximum_temperature_change > 0))
        return fail(ctx, CF_ERR_INVALID, "sea-ice conductivity and maximum temperature change must be > 0");
    DevParams d;
    int rc = lower_params(ctx, ice_fluxes, &d);
    if (rc != CF_OK) return rc;
    if (ice_fluxes->flux_formulation != CF_FORMULATION_SIMILARITY)
        return fail(ctx, CF_ERR_INVALID, "the atmosphere-sea-ice interface takes SimilarityTheoryFluxes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->d_ice_tables) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ice_tables, sizeof(double) * TABLE_DOUBLES));
    if (!ctx->d_ice_params) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ice_params, sizeof(DevParams)));
    std::vector<double> t = build_solver_tables(d.stability);
    HIP_TRY(ctx, hipMemcpy(ctx->d_ice_tables, t.data(), sizeof(double) * TABLE_DOUBLES, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ctx->d_ice_params, &d, sizeof(DevParams), hipMemcpyHostToDevice));
    ctx->ice_params = *ice_fluxes;
    ctx->ice_props = *ice;
    ctx->ice_dev = d;
    ctx->ice_loop = loop_params(*ice_fluxes, d);
    IceParams K{};
    K.inv_k = 1.0 / ice->conductivity;
    K.hk_min = ice->consolidation_thickness / ice->conductivity;
    K.dT_max = ice->maximum_temperature_change;
    K.T_melt = ice->freshwater_melting_temperature;
    K.T_fw = ice->freshwater_melting_temperature;
    K.liquidus_slope = ice->liquidus_slope;
    K.emissivity = ice->emissivity;
    K.eps_sigma = ice->emissivity * ice_fluxes->stefan_boltzmann;
    K.albedo = ice->albedo;
    K.T_offset = ice->temperature_offset;
    K.semi_implicit = ice->skin_temperature_scheme == CF_SKIN_SEMI_IMPLICIT ? 1.0 : 0.0;
    K.orbit_shortcut = ctx->ice_orbit_shortcut ? 1.0 : 0.0;
    K.ice_free_zero = ctx->ice_free_zero ? 1.0 : 0.0;
    ctx->ice_kernel = K;
    if (!ctx->ice_ready && ctx->merged_prefetch == 2) ctx->chunk_valid = false;   // (the chunk plan depends on it, see ensure_chunk_table)
    ctx->ice_ready = true;
    return CF_OK;
}

int cf_default_sea_ice_albedo_params(cf_sea_ice_albedo_params* p) {
    if (!p) return fail(nullptr, CF_ERR_INVALID, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->ice_visible = 0.78;
    p->ice_near_infrared = 0.36;
    p->snow_visible = 0.98;
    p->snow_near_infrared = 0.70;
    p->ocean_albedo = 0.06;
    p->reference_thickness = 0.3;
    p->melt_temperature_range = 1.0;
    p->ice_melt_change = 0.075;
    p->snow_melt_change_visible = 0.10;
    p->snow_melt_change_near_infrared = 0.15;
    p->snow_patch_thickness = 0.02;
    p->visible_fraction = 0.5;
    p->melting_temperature = 0.0;
    return CF_OK;
}

static int check_albedo_params(cf_ctx* ctx, const cf_sea_ice_albedo_params* p) {
    if (p->struct_size != (int32_t)sizeof(cf_sea_ice_albedo_params))
        return fail(ctx, CF_ERR_INVALID, "cf_sea_ice_albedo_params.struct_size = %d, library expects %zu", p->struct_size,
                    sizeof(cf_sea_ice_albedo_params));
    if (!(p->reference_thickness > 0) || !(p->melt_temperature_range > 0) || !(p->snow_patch_thickness > 0) ||
        !(p->visible_fraction >= 0 && p->visible_fraction <= 1))
        return fail(ctx, CF_ERR_INVALID, "sea-ice albedo: reference thickness, melt range, snow patch must be > 0, visible fraction in [0,1]");
    return CF_OK;
}

int cf_set_sea_ice_albedo(cf_ctx* ctx, const cf_sea_ice_albedo_params* params) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    if (!params) {
        ctx->ice_albedo_ccsm3 = false;
        return CF_OK;
    }
    CHECK(check_albedo_params(ctx, params));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_ice_albedo)
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_ice_albedo, sizeof(double) * (size_t)ctx->grid.sj * (ctx->grid.ny + 2 * ctx->grid.hy)));
    ctx->ice_albedo = *params;
    ctx->ice_albedo_ccsm3 = true;
    return CF_OK;
}

int cf_compute_sea_ice_albedo(cf_ctx* ctx, const cf_sea_ice_albedo_params* params, const double* d_hi, const double* d_hs,
                              const double* d_Ts, double* d_albedo) {
    if (!ctx || !params || !d_hi || !d_Ts || !d_albedo) return fail(ctx, CF_ERR_INVALID, "cf_compute_sea_ice_albedo: NULL argument");
    CHECK(check_albedo_params(ctx, params));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_sea_ice_albedo(ctx->stream, *params, ctx->grid, d_hi, d_hs, d_Ts, d_albedo));
    return CF_OK;
}

// the sea-ice state the kernels see: with the CCSM3 scheme switched on and no albedo field given, the library's own
static int resolve_ice_albedo(cf_ctx* ctx, const cf_sea_ice_state* in, cf_sea_ice_state* out) {
    *out = *in;
    if (in->albedo || !ctx->ice_albedo_ccsm3) return CF_OK;
    if (!in->thickness || !in->top_temperature) return fail(ctx, CF_ERR_INVALID, "SeaIceAlbedo(hi, hs, Ts) needs ice thickness and top temperature");
    HIP_TRY(ctx, launch_sea_ice_albedo(ctx->stream, ctx->ice_albedo, ctx->grid, in->thickness, in->snow_thickness, in->top_temperature,
                                       ctx->d_ice_albedo));
    out->albedo = ctx->d_ice_albedo;
    return CF_OK;
}

int cf_default_ice_ocean_params(cf_ice_ocean_params* p) {
    if (!p) return fail(nullptr, CF_ERR_INVALID, "params is NULL");
    std::memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->heat_transfer_coefficient = 0.0095;
    p->salt_transfer_coefficient = 0.0095 / 35.0;
    p->minimum_friction_velocity = 0.0;
    p->ice_density = 917.0;
    p->latent_heat_of_fusion = 334000.0;
    p->ice_salinity = 4.0;
    p->liquidus_slope = 0.054;
    p->top_cell_thickness = 10.0;
    p->time_step = 0.0;
    return CF_OK;
}

int cf_compute_sea_ice_ocean_fluxes(cf_ctx* ctx, const cf_ice_ocean_params* params, const cf_ocean_surface* ocean,
                                    const double* d_concentration, const double* d_x_stress, const double* d_y_stress,
                                    const cf_ice_ocean_fluxes* out) {
    if (!ctx || !params || !ocean || !ocean->T || !ocean->S || !out || !out->interface_heat || !out->salt_flux)
        return fail(ctx, CF_ERR_INVALID, "cf_compute_sea_ice_ocean_fluxes: NULL argument");
    if (params->struct_size != (int32_t)sizeof(cf_ice_ocean_params))
        return fail(ctx, CF_ERR_INVALID, "cf_ice_ocean_params.struct_size = %d, library expects %zu", params->struct_size,
                    sizeof(cf_ice_ocean_params));
    if (!(params->heat_transfer_coefficient > 0) || !(params->salt_transfer_coefficient > 0) || !(params->liquidus_slope > 0) ||
        !(params->latent_heat_of_fusion > 0))
        return fail(ctx, CF_ERR_INVALID, "ice-ocean transfer coefficients, liquidus slope and latent heat must be > 0");
    if (ctx->dev.mask_kind != CF_MASK_NONE && !ocean->mask) return fail(ctx, CF_ERR_INVALID, "ocean mask is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    CHECK(wait_for_halos(ctx));
    HIP_TRY(ctx, launch_sea_ice_ocean_fluxes(ctx->stream, ctx->dev, *params, ctx->grid, ocean, d_concentration, d_x_stress, d_y_stress, out));
    return CF_OK;
}

static int atmosphere_sea_ice_fluxes_impl(cf_ctx* ctx, const cf_sea_ice_state* ice_in, const cf_ocean_surface* ocean,
                                          const cf_exchange_fields* atmos, const cf_interface_fluxes* out, const AiTail* tail,
                                          const NetIceOut* net_ice = nullptr) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    if (!ctx->ice_ready) return fail(ctx, CF_ERR_INVALID, "cf_set_sea_ice_formulation has not been called");
    if (!ice_in || !ice_in->thickness || !ice_in->top_temperature)
        return fail(ctx, CF_ERR_INVALID, "sea-ice thickness and top temperature are NULL");
    cf_sea_ice_state resolved;
    CHECK(resolve_ice_albedo(ctx, ice_in, &resolved));
    const cf_sea_ice_state* ice = &resolved;
    if (!ocean || !ocean->S) return fail(ctx, CF_ERR_INVALID, "ocean salinity is NULL");
    if (ctx->ice_dev.mask_kind != CF_MASK_NONE && !ocean->mask) return fail(ctx, CF_ERR_INVALID, "ocean mask is NULL");
    CHECK(check_exchange(ctx, atmos, true));
    CHECK(check_fluxes(ctx, out));
    CHECK(wait_for_halos(ctx));
    CHECK(ensure_chunk_table(ctx, ocean->mask));
    HIP_TRY(ctx, launch_ai_fluxes(ctx->stream, ctx->launch, ctx->ice_dev, ctx->ice_loop, ctx->ice_kernel, ctx->grid, ice, ocean,
                                  atmos, out, ctx->d_ice_tables, ctx->d_ice_params, ctx->trip_hints ? ctx->d_trip_ice : nullptr, tail, net_ice));
    return CF_OK;
}

int cf_compute_atmosphere_sea_ice_fluxes(cf_ctx* ctx, const cf_sea_ice_state* ice_in, const cf_ocean_surface* ocean,
                                         const cf_exchange_fields* atmos, const cf_interface_fluxes* out) {
    return atmosphere_sea_ice_fluxes_impl(ctx, ice_in, ocean, atmos, out, nullptr);
}

int cf_compute_net_sea_ice_fluxes(cf_ctx* ctx, const cf_sea_ice_state* ice_in, const cf_ocean_surface* ocean,
                                  const cf_exchange_fields* atmos, const cf_interface_fluxes* ai_fluxes,
                                  const double* frazil_heat, const double* interface_heat,
                                  const cf_net_sea_ice_fluxes* out) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    if (!ctx->ice_ready) return fail(ctx, CF_ERR_INVALID, "cf_set_sea_ice_formulation has not been called");
    if (!ice_in || !ice_in->concentration) return fail(ctx, CF_ERR_INVALID, "sea-ice concentration is NULL");
    cf_sea_ice_state resolved;
    CHECK(resolve_ice_albedo(ctx, ice_in, &resolved));
    const cf_sea_ice_state* ice = &resolved;
    if (!atmos || !atmos->Qs || !atmos->Ql) return fail(ctx, CF_ERR_INVALID, "downwelling radiation fields are NULL");
    if (!ai_fluxes || !ai_fluxes->sensible_heat || !ai_fluxes->latent_heat || !ai_fluxes->temperature)
        return fail(ctx, CF_ERR_INVALID, "atmosphere-sea-ice interface fluxes are NULL");
    if (!out || !out->top_heat || !out->bottom_heat) return fail(ctx, CF_ERR_INVALID, "net sea-ice flux outputs are NULL");
    if (ctx->ice_dev.mask_kind != CF_MASK_NONE && (!ocean || !ocean->mask)) return fail(ctx, CF_ERR_INVALID, "ocean mask is NULL");
    const IceParams& K = ctx->ice_kernel;
    HIP_TRY(ctx, launch_net_sea_ice_fluxes(ctx->stream, ctx->ice_dev, ctx->grid, ocean ? ocean->mask : nullptr, ice, K.albedo,
                                           K.emissivity, K.eps_sigma, K.T_offset, atmos, ai_fluxes, frazil_heat,
                                           interface_heat, out));
    return CF_OK;
}

int cf_update_state_sea_ice(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                            const cf_ocean_surface* ocean, const cf_exchange_fields* atmos,
                            const cf_interface_fluxes* ao_fluxes, const cf_sea_ice_fields* ice_partition,
                            const cf_net_ocean_fluxes* net, const cf_sea_ice_state* ice_state,
                            const cf_interface_fluxes* ai_fluxes, const double* frazil_heat,
                            const double* interface_heat, const cf_net_sea_ice_fluxes* net_ice) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    if (!ctx->ice_ready) return fail(ctx, CF_ERR_INVALID, "cf_set_sea_ice_formulation has not been called");
    // CF_OPT_MERGED_PREFETCH = 2: the face stresses of this step and a requested next-step interpolation ride in the tail
    // workgroups of the interface solve — the longest launch of the step, whose workgroups retire over tens of microseconds
    if (ocean) CHECK(ensure_chunk_table(ctx, ocean->mask));
    const bool ice_tail = ctx->merged_prefetch == 2 && ctx->ice_loop.specialization == SOLVER_ICE &&
                          ctx->launch.solver == CF_SOLVER_TABLES;
    bool stress_held = false;
    // … and the ocean solve itself: its workgroups ride behind the interface solve's (both FP64-bound and independent of each
    // other; two queues do not overlap them, one launch does — profiles/r04_experiments.md §17); the stresses, which need the
    // ocean solve's ρτ everywhere, then get a launch of their own behind it
    static const bool ocean_rides_allowed = [] {  // (experiments: COFLUX_EXPERIMENTS=1 COFLUX_OCEAN_RIDER=0 keeps the two solver launches)
        const char* e = experiment_knob("COFLUX_OCEAN_RIDER");
        return !(e && e[0] == '0');
    }();
    OceanRider rider;
    CHECK(update_state_impl(ctx, src, w, ocean, atmos, ao_fluxes, ice_partition, net, ice_tail, &stress_held,
                            ice_tail && ocean_rides_allowed ? &rider : nullptr));
    AiTail T{};
    bool interp_rides = false;
    const bool stress_after = rider.valid && stress_held;
    if (rider.valid) {
        T.ocean = &rider;
        stress_held = false;
    }
    if (ice_tail) {
        if (stress_held) {
            T.d_ocean_params = ctx->launch.d_params;
            T.stress_ocean = ocean;
            T.stress_fluxes = ao_fluxes;
            T.stress_ice = ice_partition;
            T.stress_net = net;
        }
        if (ctx->deferred.valid && ctx->deferred.out.u != atmos->u && interp_tile_fits(ctx, 40960)) {
            interpolate_grid(ctx->launch, ctx->grid, &T.interp_rows, &T.interp_blocks);
            T.next_src = &ctx->deferred.src;
            T.w = &ctx->deferred.w;
            T.next_out = &ctx->deferred.out;
            interp_rides = true;
        }
    }
    const bool any_tail = stress_held || interp_rides || rider.valid;
    // … and compute_net_sea_ice_fluxes! is pointwise on the interface solve's own outputs: in its epilogue (same function, same
    // bits as net_sea_ice_flux_kernel) instead of an 8 µs launch behind it
    const bool net_in_epilogue = ice_tail && ice_state && ice_state->concentration && net_ice && net_ice->top_heat && net_ice->bottom_heat &&
                                 ai_fluxes && ai_fluxes->sensible_heat && ai_fluxes->latent_heat && ai_fluxes->temperature;
    const NetIceOut NI{net_in_epilogue ? ice_state->concentration : nullptr, frazil_heat, interface_heat,
                       net_in_epilogue ? net_ice->top_heat : nullptr, net_in_epilogue ? net_ice->bottom_heat : nullptr};
    CHECK(atmosphere_sea_ice_fluxes_impl(ctx, ice_state, ocean, atmos, ai_fluxes, any_tail ? &T : nullptr, net_in_epilogue ? &NI : nullptr));
    if (stress_after) HIP_TRY(ctx, launch_net_stress(ctx->stream, ctx->dev, ctx->grid, ocean, ao_fluxes, ice_partition, net));
    if (interp_rides) CHECK(deferred_went_out_on_main(ctx));
    if (ice_tail && ctx->deferred.valid && ctx->deferred.out.u != atmos->u) {   // (no tiled interpolation configured: a launch of its own)
        HIP_TRY(ctx, launch_interpolate(ctx->stream, ctx->launch, ctx->grid, &ctx->deferred.src, &ctx->deferred.w, &ctx->deferred.out));
        CHECK(deferred_went_out_on_main(ctx));
    }
    if (net_in_epilogue) return CF_OK;
    return cf_compute_net_sea_ice_fluxes(ctx, ice_state, ocean, atmos, ai_fluxes, frazil_heat, interface_heat, net_ice);
}

int cf_interpolate_land_freshwater(cf_ctx* ctx, const cf_land_source* src, const cf_interp_weights* w, double* d_out) {
    if (!ctx || !src || !src->friver || !d_out) return fail(ctx, CF_ERR_INVALID, "cf_interpolate_land_freshwater: NULL argument");
    CHECK(check_weights(ctx, w));
    if (src->ns_x <= 0 || src->ns_y <= 0 || src->n_levels <= 0 || src->level1 < 0 || src->level1 >= src->n_levels || src->level2 < 0 ||
        src->level2 >= src->n_levels)
        return fail(ctx, CF_ERR_INVALID, "land source: shape %dx%d, levels (%d,%d) of %d", src->ns_x, src->ns_y, src->level1, src->level2, src->n_levels);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_interpolate_land(ctx->stream, ctx->grid, src, w, d_out));
    return CF_OK;
}

int cf_set_land_freshwater(cf_ctx* ctx, const double* d_land_freshwater) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    ctx->d_land_freshwater = d_land_freshwater;
    return CF_OK;
}

int cf_materialize_salinity_restoring(cf_ctx* ctx, double piston_velocity, const double* d_target, const cf_ocean_surface* ocean,
                                      double* d_buffer) {
    if (!ctx || !d_target || !ocean || !ocean->S || !d_buffer) return fail(ctx, CF_ERR_INVALID, "cf_materialize_salinity_restoring: NULL argument");
    if (ctx->dev.mask_kind != CF_MASK_NONE && !ocean->mask) return fail(ctx, CF_ERR_INVALID, "ocean mask is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_salinity_restoring(ctx->stream, ctx->dev, ctx->grid, ocean->mask, piston_velocity, d_target, ocean->S, d_buffer));
    return CF_OK;
}

int cf_normalize_salinity_flux(cf_ctx* ctx, double* d_flux, const double* d_additional, const double* d_area,
                               const void* d_mask, double* d_mean_out) {
    if (!ctx || !d_flux) return fail(ctx, CF_ERR_INVALID, "cf_normalize_salinity_flux: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    if (ctx->dev.mask_kind != CF_MASK_NONE && !d_mask)
        return fail(ctx, CF_ERR_INVALID, "mask_kind = %d but the mask is NULL", ctx->dev.mask_kind);
    if (!ctx->d_reduce) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_reduce, sizeof(double) * (2 * SALINITY_PARTIAL_BLOCKS + 2)));
    double* sums = ctx->d_reduce + 2 * SALINITY_PARTIAL_BLOCKS;
    const int ncells = ctx->grid.nx * ctx->grid.ny;
    const int nblocks = std::max(1, std::min(SALINITY_PARTIAL_BLOCKS, (ncells + 255) / 256));
    HIP_TRY(ctx, launch_salinity_partial_sums(ctx->stream, ctx->dev, ctx->grid, d_flux, d_additional, d_area, d_mask,
                                              ctx->d_reduce, nblocks, sums));
    if (ctx->comm && ctx->nranks > 1)  // the only collective near the path: two doubles
        NCCL_TRY(ctx, g_rccl.AllReduce(sums, sums, 2, ncclFloat64, ncclSum, ctx->comm, ctx->stream));
    HIP_TRY(ctx, launch_salinity_subtract(ctx->stream, ctx->grid, d_flux, sums, d_mean_out));
    return CF_OK;
}

int cf_halo_exchange_rows(cf_ctx* ctx, double* const* d_fields, int nfields, int rows) {
    if (!ctx || !d_fields || nfields <= 0) return fail(ctx, CF_ERR_INVALID, "cf_halo_exchange_rows: bad arguments");
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    if (!ctx->comm) return fail(ctx, CF_ERR_COMM, "cf_comm_init has not been called");
    const GridDesc& G = ctx->grid;
    if (rows <= 0 || rows > G.hy || rows > G.ny) return fail(ctx, CF_ERR_INVALID, "rows = %d outside [1, min(hy, ny)]", rows);
    const size_t count = (size_t)rows * G.sj;  // whole rows, x-halos included
    const int south = ctx->rank - 1, north = ctx->rank + 1;
    // Three-launch step: the rows travel on a communication stream beside the interpolation kernel, which does not read
    // the ocean state.  With tail workgroups (CF_OPT_MERGED_PREFETCH = 2) the step has no such kernel — the solver launch
    // that needs the rows comes next — and the two events each way would only be four queue packets (≈ 3 µs each,
    // measured on the event this round removed from the step): the exchange goes onto the context's own stream.
    const bool own_stream = ctx->merged_prefetch != 2;
    if (own_stream && !ctx->comm_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_main_idle, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_comm_done, hipEventDisableTiming));
    }
    hipStream_t cs = ctx->stream;
    if (own_stream) {
        // the rows may only be overwritten once everything already queued on the main stream has read them
        HIP_TRY(ctx, hipEventRecord(ctx->ev_main_idle, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_main_idle, 0));
        cs = ctx->comm_stream;
    }
    NCCL_TRY(ctx, g_rccl.GroupStart());
    for (int f = 0; f < nfields; ++f) {
        double* base = d_fields[f];
        double* first_interior = base + (size_t)G.hy * G.sj;
        double* last_interior = base + (size_t)(G.hy + G.ny - rows) * G.sj;
        double* south_halo = base + (size_t)(G.hy - rows) * G.sj;
        double* north_halo = base + (size_t)(G.hy + G.ny) * G.sj;
        if (south >= 0) {
            NCCL_TRY(ctx, g_rccl.Send(first_interior, count, ncclFloat64, south, ctx->comm, cs));
            NCCL_TRY(ctx, g_rccl.Recv(south_halo, count, ncclFloat64, south, ctx->comm, cs));
        }
        if (north < ctx->nranks) {
            NCCL_TRY(ctx, g_rccl.Send(last_interior, count, ncclFloat64, north, ctx->comm, cs));
            NCCL_TRY(ctx, g_rccl.Recv(north_halo, count, ncclFloat64, north, ctx->comm, cs));
        }
    }
    NCCL_TRY(ctx, g_rccl.GroupEnd());
    if (own_stream) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_comm_done, cs));
        ctx->comm_pending = true;  // consumed by the next ocean-reading launch (or cf_sync)
    }
    return CF_OK;
}

}  // extern "C"
