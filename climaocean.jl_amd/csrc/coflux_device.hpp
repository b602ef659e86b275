// coflux_device.hpp — device-side physics of the surface-flux path for gfx950 (CDNA4).
//
// Written for the MI355X: every per-cell quantity that does not change across the
// Monin–Obukhov iteration (air state, saturation humidity, virtual temperature, viscosity,
// Δθ, Δq, |Δu|²) is hoisted into registers before the loop, the loop itself is predicated per
// lane and left together on a wave64 ballot, and nothing here touches memory.
//
// What it computes is stated by the reference's configuration sites
// (src/OMIPConfigurations/omip_simulation.jl:40-113) and by the algorithm of the un-vendored
// NumericalEarth.jl InterfaceComputations module; see DESIGN.md §2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/coflux.h"

namespace coflux {

// Host-derived constants, passed by value in the kernarg segment (scalar loads, SGPR-resident).
struct DevParams {
    // thermodynamics
    double R_d, R_v, eps, delta, cp_d, cp_v, cp_l, cp_i;
    double LH_v0, LH_s0, T_0, T_triple, inv_T_triple, p_triple, T_freeze, T_icenuc, inv_icenuc_span, pow_icenuc;
    double svp_a_liq, svp_b_liq;  // svp_liquid(T) = p_tr·exp(a·log(T/T_tr) + b·(1/T_tr − 1/T))
    double Rd_over_Rv, inv_R_v, inv_R_d;
    // sea water (Raoult)
    double sw_inv_w, sw_inv_mu;
    // similarity theory
    double kappa, beta_gust, min_gust, profile_floor, tol;  // min_gust: the FLOOR of the gust term — U_G,min, or 0 in the shear-aware form
    // U² = wind2_scale·|Δu|² + wind2_add + max((β w★)², min_gust²): (1, 0) by default; shear-aware gustiness
    // (cf_flux_params.shear_gustiness_coefficient c > 0): (1 + c², U_G,min²) with min_gust = 0
    double wind2_scale, wind2_add;
    double h_ref, h_bl, g, inv_g, log_h;
    int32_t similarity_form, stability, stop_kind, maxiter, velocity_difference, mask_kind;
    cf_roughness rm, rt, rq;
    // ocean / radiation
    double rho_o_inv, c_o_inv, rho_f_inv, T_offset, S_min, z_surface;
    double albedo, albedo_diffuse, albedo_direct, emissivity, sigma;
    int32_t albedo_kind, penetrating_sw;
};

struct GridDesc {
    int32_t nx, ny, hx, hy, ring, sj;  // sj = nx + 2 hx (row stride)
};

__device__ __forceinline__ size_t cell_index(const GridDesc& g, int i, int j) {
    return (size_t)(j + g.hy) * (size_t)g.sj + (size_t)(i + g.hx);
}

// ---------------------------------------------------------------------------------------------
// moist air
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double liquid_fraction(const DevParams& P, double T) {
    double r = (T - P.T_icenuc) * P.inv_icenuc_span;
    double ramp = (P.pow_icenuc == 1.0) ? r : pow(fmax(r, 0.0), P.pow_icenuc);
    return T > P.T_freeze ? 1.0 : (T > P.T_icenuc ? ramp : 0.0);
}

__device__ __forceinline__ double svp_liquid(const DevParams& P, double T) {
    return P.p_triple * exp(P.svp_a_liq * log(T * P.inv_T_triple) + P.svp_b_liq * (P.inv_T_triple - 1.0 / T));
}

// liquid-fraction weighted saturation vapour pressure (PhaseEquil)
__device__ __forceinline__ double svp_equil(const DevParams& P, double T, double lam) {
    if (lam == 1.0) return svp_liquid(P, T);
    double LH_0 = lam * P.LH_v0 + (1.0 - lam) * P.LH_s0;
    double dcp = lam * (P.cp_v - P.cp_l) + (1.0 - lam) * (P.cp_v - P.cp_i);
    double a = dcp * P.inv_R_v, b = (LH_0 - dcp * P.T_0) * P.inv_R_v;
    return P.p_triple * exp(a * log(T * P.inv_T_triple) + b * (P.inv_T_triple - 1.0 / T));
}

struct AirState {
    double rho, cp_m, q_vap, T_virtual;
};

// PhaseEquil_pTq(p, T, q) and the derived quantities the path needs, with one svp evaluation.
__device__ __forceinline__ AirState air_state(const DevParams& P, double p, double T, double q_tot,
                                              double lam, double p_vs) {
    AirState s;
    double q = fmin(fmax(q_tot, 0.0), 1.0);
    const double tiny = 2.220446049250313e-16;
    double q_vs_p = (p - p_vs >= tiny) ? P.Rd_over_Rv * (1.0 - q) * p_vs / (p - p_vs) : 1.0 / tiny;
    double q_c0 = fmax(q - q_vs_p, 0.0);
    s.rho = p / (P.R_d * (1.0 + P.delta * q - P.eps * q_c0) * T);
    // PhasePartition(ts): condensate recomputed from (T, ρ, q)
    double q_vs_rho = p_vs / (s.rho * P.R_v * T);
    double q_c = fmax(q - q_vs_rho, 0.0);
    double q_l = lam * q_c, q_i = (1.0 - lam) * q_c;
    s.cp_m = P.cp_d + (P.cp_v - P.cp_d) * q + (P.cp_l - P.cp_v) * q_l + (P.cp_i - P.cp_v) * q_i;
    s.q_vap = fmax(0.0, q - q_l - q_i);
    s.T_virtual = (1.0 + P.delta * q - P.eps * q_c) * T;
    return s;
}

__device__ __forceinline__ double water_mole_fraction(const DevParams& P, double S) {
    double s = S / 1000.0;
    double alpha = s / (1.0 - s);
    return P.sw_inv_w / (P.sw_inv_w + alpha * P.sw_inv_mu);
}

// ---------------------------------------------------------------------------------------------
// stability functions
// ---------------------------------------------------------------------------------------------
#define CF_PI 3.14159265358979323846
#define CF_SQRT3 1.7320508075688772

__device__ __forceinline__ double paulson_momentum(double x) {  // x = (1 − cζ)^¼
    return 2.0 * log((1.0 + x) * 0.5) + log((1.0 + x * x) * 0.5) - 2.0 * atan(x) + CF_PI / 2.0;
}
__device__ __forceinline__ double convective_branch(double y) {  // y = (1 − cζ)^⅓
    return 1.5 * log((1.0 + y + y * y) * (1.0 / 3.0)) - CF_SQRT3 * atan((1.0 + 2.0 * y) * (1.0 / CF_SQRT3)) +
           CF_PI / CF_SQRT3;
}

template <int STAB>
__device__ __forceinline__ double psi_momentum(double zeta) {
    double zm = fmin(0.0, zeta), zp = fmax(0.0, zeta);
    if constexpr (STAB == CF_STABILITY_EDSON2013) {
        if (zeta < 0.0) {
            double pu1 = paulson_momentum(sqrt(sqrt(1.0 - 15.0 * zm)));
            double pu2 = convective_branch(cbrt(1.0 - 10.15 * zm));
            double z2 = zm * zm;
            double f = z2 / (1.0 + z2);
            return (1.0 - f) * pu1 + f * pu2;
        }
        double dz = fmin(50.0, 0.35 * zp);
        return -0.7 * zp - 0.75 * (zp - 5.0 / 0.35) * exp(-dz) - 0.75 * 5.0 / 0.35;
    } else if constexpr (STAB == CF_STABILITY_SHEBA) {
        if (zeta < 0.0) return paulson_momentum(sqrt(sqrt(1.0 - 16.0 * zm)));
        const double a = 5.0, b = 5.0 / 6.5;
        const double B = 0.6694329500821695;  // cbrt((1-b)/b)
        double x = cbrt(1.0 + zp);
        return -3.0 * a / b * (x - 1.0) +
               a * B / (2.0 * b) *
                   (2.0 * log((x + B) / (1.0 + B)) - log((x * x - x * B + B * B) / (1.0 - B + B * B)) +
                    2.0 * CF_SQRT3 * (atan((2.0 * x - B) / (CF_SQRT3 * B)) - atan((2.0 - B) / (CF_SQRT3 * B))));
    } else {
        if (zeta < 0.0) return paulson_momentum(sqrt(sqrt(1.0 - 16.0 * zm)));
        return -5.0 * zp;
    }
}

template <int STAB>
__device__ __forceinline__ double psi_scalar(double zeta) {
    double zm = fmin(0.0, zeta), zp = fmax(0.0, zeta);
    if constexpr (STAB == CF_STABILITY_EDSON2013) {
        if (zeta < 0.0) {
            double pu1 = 2.0 * log((1.0 + sqrt(1.0 - 15.0 * zm)) * 0.5);
            double pu2 = convective_branch(cbrt(1.0 - 34.15 * zm));
            double z2 = zm * zm;
            double f = z2 / (1.0 + z2);
            return (1.0 - f) * pu1 + f * pu2;
        }
        double dz = fmin(50.0, 0.35 * zp);
        double base = 1.0 + 2.0 / 3.0 * zp;
        return -(base * sqrt(base)) - 2.0 / 3.0 * (zp - 14.28) * exp(-dz) - 8.525;
    } else if constexpr (STAB == CF_STABILITY_SHEBA) {
        if (zeta < 0.0) return 2.0 * log((1.0 + sqrt(1.0 - 16.0 * zm)) * 0.5);
        const double a = 5.0, b = 5.0, c = 3.0;
        const double B = 2.23606797749979;  // sqrt(c² − 4)
        return -b / 2.0 * log(1.0 + c * zp + zp * zp) +
               (-a / B + b * c / (2.0 * B)) *
                   (log((2.0 * zp + c - B) / (2.0 * zp + c + B)) - log((c - B) / (c + B)));
    } else {
        if (zeta < 0.0) return 2.0 * log((1.0 + sqrt(1.0 - 16.0 * zm)) * 0.5);
        return -5.0 * zp;
    }
}

// ---------------------------------------------------------------------------------------------
// roughness lengths
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double air_viscosity(const cf_roughness& r, double T_kelvin) {
    if (r.viscosity_kind == CF_VISCOSITY_CONSTANT) return r.viscosity[0];
    double Tc = T_kelvin - 273.15;
    return r.viscosity[0] + Tc * (r.viscosity[1] + Tc * (r.viscosity[2] + Tc * r.viscosity[3]));
}

__device__ __forceinline__ double momentum_roughness(const cf_roughness& r, double inv_g, double ustar,
                                                     double alpha, double nu) {
    if (r.kind == CF_ROUGHNESS_CONSTANT) return r.constant_length;
    double lm = r.maximum_length;
    double lR = (ustar == 0.0) ? lm : r.laminar * nu / ustar;
    return fmin(alpha * ustar * ustar * inv_g + lR, lm);
}

__device__ __forceinline__ double scalar_roughness(const cf_roughness& r, double lu, double ustar, double nu) {
    if (r.kind == CF_SCALAR_ROUGHNESS_CONSTANT) return r.constant_length;
    double lm = r.maximum_length;
    double Rstar = lu * ustar / nu;
    double lq = (Rstar == 0.0) ? 0.0 : r.reynolds_A * exp(-r.reynolds_b * log(Rstar));
    lq = (ustar == 0.0) ? lm : lq;
    return fmin(lq, lm);
}

// ---------------------------------------------------------------------------------------------
// per-cell Monin–Obukhov solve
// ---------------------------------------------------------------------------------------------
struct CellFluxes {
    double Qc, Qv, Fv, rho_tau_x, rho_tau_y, Ts_ocean, ustar, tstar, qstar;
    int iterations;
};

template <int STAB, bool COARE>
__device__ __forceinline__ double similarity_profile(double log_h, double h, double l, double inv_L,
                                                     double floor_, bool scalar) {
    double zeta = h * inv_L;
    double r = log_h - log(l) - (scalar ? psi_scalar<STAB>(zeta) : psi_momentum<STAB>(zeta));
    if constexpr (!COARE) r += scalar ? psi_scalar<STAB>(l * inv_L) : psi_momentum<STAB>(l * inv_L);
    return r < floor_ ? floor_ : r;
}

// Solves one cell.  `wet` lanes iterate; with FixedIterations every lane iterates and land is
// zeroed afterwards, as the reference does.  All lanes of the wave must call this together
// (wave64 ballot inside).
template <int STAB, bool COARE, bool FIXED>
__device__ __forceinline__ CellFluxes solve_cell(const DevParams& P, double ua, double va, double Ta,
                                                 double pa, double qa, double uo, double vo, double To,
                                                 double So, bool wet, bool in_range) {
    CellFluxes R;
    const double Ts = To + P.T_offset;

    // --- iteration-invariant state -------------------------------------------------------------
    const double lam_a = liquid_fraction(P, Ta);
    const double pvs_a = svp_equil(P, Ta, lam_a);
    const AirState A = air_state(P, pa, Ta, qa, lam_a, pvs_a);

    const double pstar_s = svp_liquid(P, Ts);
    const double qs = water_mole_fraction(P, So) * pstar_s / (A.rho * P.R_v * Ts);
    const double dq = A.q_vap - qs;
    const double dtheta = Ta + P.g * P.h_ref / A.cp_m - Ts;
    double du = ua, dv = va;
    if (P.velocity_difference == CF_VELOCITY_RELATIVE) {
        du = ua - uo;
        dv = va - vo;
    }
    const double dU2 = du * du + dv * dv;
    const double dU = sqrt(dU2);

    const double lam_s = liquid_fraction(P, Ts);
    const double pvs_s = (lam_s == 1.0) ? pstar_s : svp_equil(P, Ts, lam_s);
    const AirState Sfc = air_state(P, pa, Ts, qs, lam_s, pvs_s);
    const double g_over_Tv = P.g / Sfc.T_virtual;
    const double b_theta = 1.0 + P.delta * Sfc.q_vap;     // b★ = g/Tv·(θ★·b_theta + q★·b_q)
    const double b_q = P.delta * Sfc.T_virtual;

    const double nu_m = air_viscosity(P.rm, Ts);
    const double nu_t = air_viscosity(P.rt, Ts);
    const double nu_q = air_viscosity(P.rq, Ts);
    double alpha = P.rm.charnock;
    if (P.rm.kind == CF_ROUGHNESS_WIND_CHARNOCK)
        alpha = fmax(P.rm.charnock, P.rm.wind_a1 * fmin(dU, P.rm.wind_umax) + P.rm.wind_a2);

    // --- the fixed point -----------------------------------------------------------------------
    double us = 1e-4, ts = 1e-4, qq = 1e-4;
    double drift = 0.0;
    int it = 0;
    const bool participates = in_range && (FIXED || wet);
    for (;;) {
        bool go;
        if constexpr (FIXED) {
            go = participates && it < P.maxiter;
        } else {
            bool converged = drift < P.tol;
            go = participates && ((it == 0) || !(converged || it >= P.maxiter));
        }
        if (__ballot(go) == 0ull) break;  // the whole wave leaves together
        if (go) {
            double bstar = g_over_Tv * (ts * b_theta + b_q * qq);
            double Jb = -us * bstar;
            double Ug = fmax(P.beta_gust * cbrt(fmax(Jb, 0.0) * P.h_bl), P.min_gust);
            double U = sqrt(fma(dU2, P.wind2_scale, P.wind2_add) + Ug * Ug);

            double lu = momentum_roughness(P.rm, P.inv_g, us, alpha, nu_m);
            double lq = scalar_roughness(P.rq, lu, us, nu_q);
            double lt = scalar_roughness(P.rt, lu, us, nu_t);

            // 1/L★ = κ b★ / u★²  (L★ = ∞ when b★ = 0); b★ < 0 ⇒ ζ < 0 ⇒ unstable
            double inv_L = (bstar == 0.0) ? 0.0 : (P.kappa * bstar) / (us * us);
            double chi_u = P.kappa / similarity_profile<STAB, COARE>(P.log_h, P.h_ref, lu, inv_L, P.profile_floor, false);
            double chi_t = P.kappa / similarity_profile<STAB, COARE>(P.log_h, P.h_ref, lt, inv_L, P.profile_floor, true);
            double chi_q = P.kappa / similarity_profile<STAB, COARE>(P.log_h, P.h_ref, lq, inv_L, P.profile_floor, true);

            double un = chi_u * U, tn = chi_t * dtheta, qn = chi_q * dq;
            drift = fabs(un - us) + fabs(tn - ts) + fabs(qn - qq);
            us = un;
            ts = tn;
            qq = qn;
            ++it;
        }
    }

    const bool zero = !wet;
    if (zero) us = ts = qq = 0.0;
    double inv_dU = (dU == 0.0) ? 0.0 : 1.0 / dU;
    double tau = -us * us * inv_dU;
    double Lv = P.LH_v0 + (P.cp_v - P.cp_l) * (Ta - P.T_0);
    double rho_u = A.rho * us;
    R.Fv = -rho_u * qq;
    R.Qv = R.Fv * Lv;
    R.Qc = -rho_u * A.cp_m * ts;
    R.rho_tau_x = A.rho * tau * du;
    R.rho_tau_y = A.rho * tau * dv;
    R.Ts_ocean = (zero ? 0.0 : Ts) - P.T_offset;
    R.ustar = us;
    R.tstar = ts;
    R.qstar = qq;
    R.iterations = it;
    return R;
}

__device__ __forceinline__ bool cell_is_wet(const DevParams& P, const void* mask, size_t k) {
    if (P.mask_kind == CF_MASK_NONE || mask == nullptr) return true;
    if (P.mask_kind == CF_MASK_U8) return ((const uint8_t*)mask)[k] != 0;
    return !(P.z_surface <= ((const double*)mask)[k]);
}

}  // namespace coflux
