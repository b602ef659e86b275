/*
 * coflux_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, double precision, scalar per-cell loops) of the surface-flux hot
 * path of ClimaOcean's OceanSeaIceModel: interpolate_atmosphere_state!, the
 * SimilarityTheoryFluxes Monin–Obukhov fixed point of compute_atmosphere_ocean_fluxes!, and
 * compute_net_ocean_fluxes!.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (libcoflux.so) never links or calls it.
 *
 * PARITY UNPINNED.  The arithmetic of this path is not in /root/reference: it lives in the
 * third-party package NumericalEarth.jl (uuid 904d977b-046a-4731-8b86-9235c0d1ef02), which
 * the reference takes un-vendored at `rev = "main"` with compat 0.4–0.8 and no Manifest
 * (Project.toml:21,31-32,48; .gitignore:1).  No Julia toolchain exists in this image, the
 * reference's tests hold no numerical vector for this path (test/test_module.jl:11-45 are
 * isdefined checks) and there are no golden files.  This file therefore restates the published
 * algorithm of that package's InterfaceComputations module (the code that was
 * ClimaOcean.OceanSeaIceModels.InterfaceComputations up to ClimaOcean v0.8.x, README.md:13-16)
 * together with CliMA Thermodynamics.jl's moist-air relations and the cited literature
 * (Edson et al. 2013; COARE 3.6; Large & Yeager 2009; Grachev et al. 2007;
 * docs/climaocean.bib:1-51), and is anchored on every parameter, keyword, unit and sign
 * convention the reference tree itself fixes:
 *   - formulation parameter sets .......... src/OMIPConfigurations/omip_simulation.jl:40-113
 *   - constant Charnock 0.02 default ...... omip_simulation.jl:263
 *   - velocity difference policy .......... omip_simulation.jl:135-137, 283-286
 *   - minimum-salinity semantics .......... experiments/OMIPSimulations/scripts/launch.sh:74-78
 *   - albedo 0.06 / emissivity 1.0 ........ src/OMIPConfigurations/atmosphere.jl:41-44
 *   - ρₒ = 1026, cₒ = 3991.86795711963 .... experiments/OMIPSimulations/scripts/visualize/common.jl:17-18
 *   - τ kinematic at faces, JT·ρ·cp=W/m² .. visualize/cache.jl:359-383, KPP/kpp_surface_forcing.jl:18-29
 *   - SW to radiation.surface_flux ........ KPP/kpp_surface_forcing.jl:47-51
 *   - JRA55 variable list, 640×320 f32 .... jra55_data_staging.jl:8, launch.sh:86-87
 * It is cross-checked by an independently written NumPy restatement (oracle/numpy_oracle.py)
 * and by physical known-answer tests (tests/test_oracle.py).
 *
 * Operation order follows the upstream kernels as recalled so that a future run against the
 * real package (julia/oracle_dump.jl, only possible where Julia+NumericalEarth exist) is as
 * close as floating point allows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/coflux.h"

#define IDX(g, i, j) ((size_t)((j) + (g)->hy) * (size_t)((g)->nx + 2 * (g)->hx) + (size_t)((i) + (g)->hx))

/* ------------------------------------------------------------------------------------------
 * Moist-air thermodynamics (CliMA Thermodynamics.jl relations behind
 * AtmosphericThermodynamics.PhaseEquil_pTq and friends)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double R_d, R_v, eps, cp_d;
} thermo_derived;

static thermo_derived derive(const cf_thermodynamics* t) {
    thermo_derived d;
    d.R_d = t->gas_constant / t->dry_air_molar_mass;
    d.R_v = t->gas_constant / t->water_molar_mass;
    d.eps = t->dry_air_molar_mass / t->water_molar_mass; /* molmass_ratio */
    d.cp_d = d.R_d / t->kappa_d;
    return d;
}

/* liquid_fraction(param_set, T, PhaseEquil) */
static double liquid_fraction(const cf_thermodynamics* t, double T) {
    if (T > t->T_freeze) return 1.0;
    if (T > t->T_icenuc) return pow((T - t->T_icenuc) / (t->T_freeze - t->T_icenuc), t->pow_icenuc);
    return 0.0;
}

/* saturation_vapor_pressure(param_set, T, LH_0, Δcp): Clausius–Clapeyron with linear L(T) */
static double svp_general(const cf_thermodynamics* t, const thermo_derived* d, double T, double LH_0,
                          double dcp) {
    return t->p_triple * pow(T / t->T_triple, dcp / d->R_v) *
           exp((LH_0 - dcp * t->T_0) / d->R_v * (1.0 / t->T_triple - 1.0 / T));
}

static double svp_liquid(const cf_thermodynamics* t, const thermo_derived* d, double T) {
    return svp_general(t, d, T, t->LH_v0, t->cp_v - t->cp_l);
}

/* saturation_vapor_pressure(param_set, PhaseEquil, T): liquid-fraction weighted */
static double svp_equil(const cf_thermodynamics* t, const thermo_derived* d, double T) {
    double lam = liquid_fraction(t, T);
    double LH_0 = lam * t->LH_v0 + (1.0 - lam) * t->LH_s0;
    double dcp = lam * (t->cp_v - t->cp_l) + (1.0 - lam) * (t->cp_v - t->cp_i);
    return svp_general(t, d, T, LH_0, dcp);
}

typedef struct {
    double rho, p, T, q_tot;
} thermo_state; /* PhaseEquil state (internal energy is never used on this path) */

static double gas_constant_air(const thermo_derived* d, double q_tot, double q_c) {
    return d->R_d * (1.0 + (d->eps - 1.0) * q_tot - d->eps * q_c);
}

/* PhaseEquil_pTq(param_set, p, T, q_tot) */
static thermo_state phase_equil_pTq(const cf_thermodynamics* t, const thermo_derived* d, double p,
                                    double T, double q_tot) {
    thermo_state s;
    double q = fmin(fmax(q_tot, 0.0), 1.0);
    double p_vs = svp_equil(t, d, T);
    /* q_vap_saturation_from_pressure */
    double q_vs = (p - p_vs >= 2.220446049250313e-16)
                      ? d->R_d / d->R_v * (1.0 - q) * p_vs / (p - p_vs)
                      : 1.0 / 2.220446049250313e-16;
    double q_c = fmax(q - q_vs, 0.0);
    s.rho = p / (gas_constant_air(d, q, q_c) * T);
    s.p = p;
    s.T = T;
    s.q_tot = q;
    return s;
}

/* PhasePartition(ts::PhaseEquil): condensate from (T, ρ, q_tot) */
static double condensate(const cf_thermodynamics* t, const thermo_derived* d, const thermo_state* s,
                         double* q_liq, double* q_ice) {
    double p_vs = svp_equil(t, d, s->T);
    double q_vs = p_vs / (s->rho * d->R_v * s->T);
    double q_c = fmax(s->q_tot - q_vs, 0.0);
    double lam = liquid_fraction(t, s->T);
    *q_liq = lam * q_c;
    *q_ice = (1.0 - lam) * q_c;
    return q_c;
}

static double cp_m(const cf_thermodynamics* t, const thermo_derived* d, const thermo_state* s) {
    double ql, qi;
    condensate(t, d, s, &ql, &qi);
    return d->cp_d + (t->cp_v - d->cp_d) * s->q_tot + (t->cp_l - t->cp_v) * ql + (t->cp_i - t->cp_v) * qi;
}

static double vapor_specific_humidity(const cf_thermodynamics* t, const thermo_derived* d,
                                      const thermo_state* s) {
    double ql, qi;
    condensate(t, d, s, &ql, &qi);
    return fmax(0.0, s->q_tot - ql - qi);
}

static double virtual_temperature(const cf_thermodynamics* t, const thermo_derived* d,
                                  const thermo_state* s) {
    double ql, qi;
    double q_c = condensate(t, d, s, &ql, &qi);
    return gas_constant_air(d, s->q_tot, q_c) / d->R_d * s->T;
}

static double latent_heat_vapor(const cf_thermodynamics* t, double T) {
    return t->LH_v0 + (t->cp_v - t->cp_l) * (T - t->T_0);
}

/* compute_water_mole_fraction(::WaterMoleFraction, S): Raoult's law factor of sea water */
static double water_mole_fraction(const cf_seawater* w, double S) {
    double s = S / 1000.0;
    double alpha = s / (1.0 - s);
    double inv_mu = 0.0;
    for (int k = 0; k < 4; ++k) inv_mu += w->constituent_mass_fraction[k] / w->constituent_molar_mass[k];
    double inv_w = 1.0 / w->water_molar_mass;
    return inv_w / (inv_w + alpha * inv_mu);
}

/* saturation_specific_humidity(SpecificHumidityFormulation(Liquid, WaterMoleFraction), ℂ, ρ, T, S) */
static double saturation_specific_humidity_ocean(const cf_flux_params* P, const thermo_derived* d,
                                                 double rho, double T, double S) {
    double x = water_mole_fraction(&P->seawater, S);
    double p_star = svp_liquid(&P->thermo, d, T);
    double q_star = p_star / (rho * d->R_v * T);
    return q_star * x;
}

/* ------------------------------------------------------------------------------------------
 * Stability functions ψ(ζ)
 * ---------------------------------------------------------------------------------------- */
static double psi_momentum(int kind, double zeta) {
    const double pi = 3.14159265358979323846;
    double zm = fmin(0.0, zeta), zp = fmax(0.0, zeta);
    if (kind == CF_STABILITY_EDSON2013) {
        /* Edson et al. 2013 / COARE 3.5 psiu_26 */
        double dz = fmin(50.0, 0.35 * zp);
        double ps = -0.7 * zp - 0.75 * (zp - 5.0 / 0.35) * exp(-dz) - 0.75 * 5.0 / 0.35;
        double f1 = sqrt(sqrt(1.0 - 15.0 * zm));
        double pu1 = 2.0 * log((1.0 + f1) / 2.0) + log((1.0 + f1 * f1) / 2.0) - 2.0 * atan(f1) + pi / 2.0;
        double f2 = cbrt(1.0 - 10.15 * zm);
        double pu2 = 1.5 * log((1.0 + f2 + f2 * f2) / 3.0) - sqrt(3.0) * atan((1.0 + 2.0 * f2) / sqrt(3.0)) +
                     pi / sqrt(3.0);
        double f = zm * zm / (1.0 + zm * zm);
        double pu = (1.0 - f) * pu1 + f * pu2;
        return zeta < 0.0 ? pu : ps;
    } else if (kind == CF_STABILITY_SHEBA) {
        /* stable: Grachev et al. 2007 eq. 12; unstable: Paulson 1970 */
        const double a = 5.0, b = 5.0 / 6.5;
        double B = cbrt((1.0 - b) / b);
        double x = cbrt(1.0 + zp);
        double r3 = sqrt(3.0);
        double ps = -3.0 * a / b * (x - 1.0) +
                    a * B / (2.0 * b) *
                        (2.0 * log((x + B) / (1.0 + B)) - log((x * x - x * B + B * B) / (1.0 - B + B * B)) +
                         2.0 * r3 * (atan((2.0 * x - B) / (r3 * B)) - atan((2.0 - B) / (r3 * B))));
        double y = sqrt(sqrt(1.0 - 16.0 * zm));
        double pu = 2.0 * log((1.0 + y) / 2.0) + log((1.0 + y * y) / 2.0) - 2.0 * atan(y) + pi / 2.0;
        return zeta < 0.0 ? pu : ps;
    } else {
        /* Large & Yeager 2009: Paulson unstable, −5ζ stable */
        double y = sqrt(sqrt(1.0 - 16.0 * zm));
        double pu = 2.0 * log((1.0 + y) / 2.0) + log((1.0 + y * y) / 2.0) - 2.0 * atan(y) + pi / 2.0;
        double ps = -5.0 * zp;
        return zeta < 0.0 ? pu : ps;
    }
}

static double psi_scalar(int kind, double zeta) {
    const double pi = 3.14159265358979323846;
    double zm = fmin(0.0, zeta), zp = fmax(0.0, zeta);
    if (kind == CF_STABILITY_EDSON2013) {
        /* COARE 3.5 psit_26 */
        double dz = fmin(50.0, 0.35 * zp);
        double base = 1.0 + 2.0 / 3.0 * zp;
        double ps = -(base * sqrt(base)) - 2.0 / 3.0 * (zp - 14.28) * exp(-dz) - 8.525;
        double f1 = sqrt(1.0 - 15.0 * zm);
        double pu1 = 2.0 * log((1.0 + f1) / 2.0);
        double f2 = cbrt(1.0 - 34.15 * zm);
        double pu2 = 1.5 * log((1.0 + f2 + f2 * f2) / 3.0) - sqrt(3.0) * atan((1.0 + 2.0 * f2) / sqrt(3.0)) +
                     pi / sqrt(3.0);
        double f = zm * zm / (1.0 + zm * zm);
        double pu = (1.0 - f) * pu1 + f * pu2;
        return zeta < 0.0 ? pu : ps;
    } else if (kind == CF_STABILITY_SHEBA) {
        /* Grachev et al. 2007 eq. 13 */
        const double a = 5.0, b = 5.0, c = 3.0;
        double B = sqrt(c * c - 4.0);
        double ps = -b / 2.0 * log(1.0 + c * zp + zp * zp) +
                    (-a / B + b * c / (2.0 * B)) *
                        (log((2.0 * zp + c - B) / (2.0 * zp + c + B)) - log((c - B) / (c + B)));
        double y2 = sqrt(1.0 - 16.0 * zm);
        double pu = 2.0 * log((1.0 + y2) / 2.0);
        return zeta < 0.0 ? pu : ps;
    } else {
        double y2 = sqrt(1.0 - 16.0 * zm);
        double pu = 2.0 * log((1.0 + y2) / 2.0);
        double ps = -5.0 * zp;
        return zeta < 0.0 ? pu : ps;
    }
}

/* similarity_profile(form, ψ, h, ℓ, L) */
static double similarity_profile(int form, int stab, int scalar, double h, double l, double L,
                                 double floor_) {
    double psi_h = scalar ? psi_scalar(stab, h / L) : psi_momentum(stab, h / L);
    double r = log(h / l) - psi_h;
    if (form == CF_SIMILARITY_LOGARITHMIC) r += scalar ? psi_scalar(stab, l / L) : psi_momentum(stab, l / L);
    /* restatement guard (include/coflux.h: similarity_profile_floor): the first iterate from the
       1e-4 initial guess has ζ ≈ −10⁵ and, in the COARE form, a negative profile ⇒ u★ < 0 ⇒
       log of a negative roughness length.  Upstream's own guard is unknown (parity unpinned). */
    return r < floor_ ? floor_ : r;
}

/* ------------------------------------------------------------------------------------------
 * Roughness lengths
 * ---------------------------------------------------------------------------------------- */
static double air_viscosity(const cf_roughness* r, double T_kelvin) {
    if (r->viscosity_kind == CF_VISCOSITY_CONSTANT) return r->viscosity[0];
    double Tc = T_kelvin - 273.15;
    return r->viscosity[0] + r->viscosity[1] * Tc + r->viscosity[2] * Tc * Tc + r->viscosity[3] * Tc * Tc * Tc;
}

/* roughness_length(ℓ::MomentumRoughnessLength, u★, …); `U` = |Δu| at the reference height feeds
 * the wind-dependent Charnock parameter (Edson 2013 eq. 13, omip_simulation.jl:35). */
/* Wind-speed scale of the similarity profiles, U² = |Δu|² + U_G².  Default: U_G = max(β w★, U_G,min) with the convective
 * velocity scale w★ = (J_b h_bl)^⅓ of an unstable layer (J_b = −u★ b★ > 0), 0 otherwise.  Shear-aware form
 * (launch.sh:67-72, Mahrt & Sun 1995 / Edson 2013): U_G² = (β w★)² + (c |Δu|)² + U_G,min², c = shear_gustiness_coefficient. */
static double wind_speed_scale(const cf_flux_params* P, double Jb, double dU2) {
    const double wstar = P->gustiness_parameter * cbrt(fmax(Jb, 0.0) * P->boundary_layer_height);
    const double c = P->shear_gustiness_coefficient;
    if (c > 0.0)
        return sqrt(dU2 + wstar * wstar + c * c * dU2 + P->minimum_gustiness * P->minimum_gustiness);
    const double Ug = fmax(wstar, P->minimum_gustiness);
    return sqrt(dU2 + Ug * Ug);
}

static double momentum_roughness(const cf_roughness* r, double g, double ustar, double U, double Ts) {
    if (r->kind == CF_ROUGHNESS_CONSTANT) return r->constant_length;
    double alpha = r->charnock;
    if (r->kind == CF_ROUGHNESS_WIND_CHARNOCK) /* `charnock` is the floor of the linear fit */
        alpha = fmax(r->charnock, r->wind_a1 * fmin(U, r->wind_umax) + r->wind_a2);
    double nu = air_viscosity(r, Ts);
    double lm = r->maximum_length;
    double lR = (ustar == 0.0) ? lm : r->laminar * nu / ustar;
    return fmin(alpha * ustar * ustar / g + lR, lm);
}

/* roughness_length(ℓ::ScalarRoughnessLength, ℓu, u★, …) */
static double scalar_roughness(const cf_roughness* r, double lu, double ustar, double Ts) {
    if (r->kind == CF_SCALAR_ROUGHNESS_CONSTANT) return r->constant_length;
    double nu = air_viscosity(r, Ts);
    double lm = r->maximum_length;
    double Rstar = lu * ustar / nu;
    double lq = (Rstar == 0.0) ? 0.0 : r->reynolds_A / pow(Rstar, r->reynolds_b);
    lq = (ustar == 0.0) ? lm : lq;
    return fmin(lq, lm);
}

/* ------------------------------------------------------------------------------------------
 * Per-cell solve: compute_interface_state + the flux formulas of
 * _compute_atmosphere_ocean_interface_state!
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double Qc, Qv, Fv, rho_tau_x, rho_tau_y, Ts_ocean_units;
    double ustar, theta_star, q_star;
    int iterations;
} cell_result;

static cell_result solve_cell(const cf_flux_params* P, double ua, double va, double Ta, double pa,
                              double qa, double uo, double vo, double To, double So, int wet) {
    cell_result R;
    memset(&R, 0, sizeof R);
    const cf_thermodynamics* t = &P->thermo;
    thermo_derived d = derive(t);
    const double g = P->gravitational_acceleration;
    const double kappa = P->von_karman;
    const double h = P->reference_height;

    double Ti = To + P->ocean_temperature_offset; /* convert_to_kelvin */
    thermo_state Qa = phase_equil_pTq(t, &d, pa, Ta, qa);

    /* Don't iterate (and write zeros) on land: needs_to_converge && not_water ⇒ zero state;
       FixedIterations ⇒ iterate, then zero the state with ifelse. */
    int skip = (!wet) && (P->stop_kind == CF_STOP_CONVERGENCE);

    double ustar = 1e-4, tstar = 1e-4, qstar = 1e-4;
    int iters = 0;
    double Ts = Ti;

    if (!skip) {
        /* interface specific humidity from the interior temperature (BulkTemperature ⇒ Ts = Ti) */
        double qs = saturation_specific_humidity_ocean(P, &d, Qa.rho, Ts, So);
        double qa_v = vapor_specific_humidity(t, &d, &Qa);
        double dq = qa_v - qs;
        double theta_a = Ta + g * h / cp_m(t, &d, &Qa); /* surface_atmosphere_temperature */
        double dtheta = theta_a - Ts;
        double du, dv;
        if (P->velocity_difference == CF_VELOCITY_RELATIVE) {
            du = ua - uo;
            dv = va - vo;
        } else {
            du = ua;
            dv = va;
        }
        thermo_state Qs = phase_equil_pTq(t, &d, pa, Ts, qs);
        double Tv = virtual_temperature(t, &d, &Qs);
        double qv_s = vapor_specific_humidity(t, &d, &Qs);
        double delta = d.eps - 1.0;

        double up = ustar, tp = tstar, qp = qstar;
        /* CoefficientBasedFluxes + LargeYeagerTransferCoefficients (omip_simulation.jl:86-89): the NCAR
           bulk algorithm of Large & Yeager (2004, 2009) — iterate on (Cd, Ch, Ce) at height h, restated
           on this package's Δθ, Δq, buoyancy scale and L★. */
        double ly_cd = 0, ly_ch = 0, ly_ce = 0, ly_cdn_rt = 0, ly_U = 0;
        if (P->flux_formulation == CF_FORMULATION_LARGE_YEAGER) {
            double dU0 = sqrt(du * du + dv * dv);
            ly_U = fmax(dU0, P->ly_minimum_wind);
            double u10 = ly_U;
            double cdn = (u10 >= P->ly_high_wind)
                             ? P->ly_cd_high * 1e-3
                             : (P->ly_cd[0] / u10 + P->ly_cd[1] + P->ly_cd[2] * u10 + P->ly_cd[3] * pow(u10, 6)) * 1e-3;
            ly_cdn_rt = sqrt(cdn);
            double stab = (dtheta > 0.0) ? 1.0 : 0.0; /* 0.5 + sign(0.5, t − ts) */
            ly_cd = cdn;
            ly_ce = P->ly_ce * ly_cdn_rt * 1e-3;
            ly_ch = (P->ly_ch_stable * stab + P->ly_ch_unstable * (1.0 - stab)) * ly_cdn_rt * 1e-3;
        }
        for (;;) {
            /* iterating(Ψⁿ, Ψ⁻, iteration, criteria) */
            int go;
            if (P->stop_kind == CF_STOP_FIXED) {
                go = iters < P->maxiter;
            } else {
                int hasnt_started = iters == 0;
                int reached = iters >= P->maxiter;
                double drift = fabs(ustar - up) + fabs(tstar - tp) + fabs(qstar - qp);
                int converged = drift < P->tolerance;
                go = (!(converged | reached)) | hasnt_started;
            }
            if (!go) break;
            up = ustar;
            tp = tstar;
            qp = qstar;

            if (P->flux_formulation == CF_FORMULATION_LARGE_YEAGER) {
                double cd_rt = sqrt(ly_cd);
                ustar = cd_rt * ly_U;                /* L-Y eq. 7a */
                tstar = ly_ch / cd_rt * dtheta;      /* 7b */
                qstar = ly_ce / cd_rt * dq;          /* 7c */
                double bs = g / Tv * (tstar * (1.0 + delta * qv_s) + delta * Tv * qstar);
                double zeta = kappa * bs * h / (ustar * ustar); /* 8a: ζ = h/L★, L★ = u★²/(κ b★); ζ > 0 stable */
                zeta = copysign(fmin(fabs(zeta), P->ly_zeta_bound), zeta);
                double psi_m = psi_momentum(CF_STABILITY_LARGE_YEAGER, zeta);
                double psi_h = psi_scalar(CF_STABILITY_LARGE_YEAGER, zeta);
                double lz = log(h / 10.0);
                double u10 = ly_U / (1.0 + ly_cdn_rt * (lz - psi_m) / kappa); /* 9 */
                double cdn = (u10 >= P->ly_high_wind)
                                 ? P->ly_cd_high * 1e-3
                                 : (P->ly_cd[0] / u10 + P->ly_cd[1] + P->ly_cd[2] * u10 + P->ly_cd[3] * pow(u10, 6)) * 1e-3;
                ly_cdn_rt = sqrt(cdn);
                double cen = P->ly_ce * ly_cdn_rt * 1e-3;
                double stab = (zeta > 0.0) ? 1.0 : 0.0;
                double chn = (P->ly_ch_stable * stab + P->ly_ch_unstable * (1.0 - stab)) * ly_cdn_rt * 1e-3;
                double xx = (lz - psi_m) / kappa;
                double den = 1.0 + ly_cdn_rt * xx;
                ly_cd = cdn / (den * den);           /* 10a */
                xx = (lz - psi_h) / kappa;
                double rt = sqrt(ly_cd / cdn);
                ly_ch = chn / (1.0 + chn * xx / ly_cdn_rt) * rt; /* 10b */
                ly_ce = cen / (1.0 + cen * xx / ly_cdn_rt) * rt; /* 10c */
                ++iters;
                continue;
            }
            /* iterate_interface_fluxes */
            double bstar = g / Tv * (tstar * (1.0 + delta * qv_s) + delta * Tv * qstar);
            double Jb = -ustar * bstar;
            double dU = sqrt(du * du + dv * dv);
            double U = wind_speed_scale(P, Jb, du * du + dv * dv);

            double lu = momentum_roughness(&P->momentum_roughness, g, ustar, dU, Ts);
            double lq = scalar_roughness(&P->water_vapor_roughness, lu, ustar, Ts);
            double lt = scalar_roughness(&P->temperature_roughness, lu, ustar, Ts);

            /* Obukhov length: buoyancy flux Jᵇ = −u★b★, L★ = −u★³/(κJᵇ) = u★²/(κb★); b★ < 0 (air colder /
               drier than the surface) ⇒ L★ < 0 ⇒ ζ < 0 ⇒ unstable branch of ψ. */
            double L = (bstar == 0.0) ? INFINITY : ustar * ustar / (kappa * bstar);
            double chi_u = kappa / similarity_profile(P->similarity_form, P->stability_functions, 0, h, lu, L, P->similarity_profile_floor);
            double chi_t = kappa / similarity_profile(P->similarity_form, P->stability_functions, 1, h, lt, L, P->similarity_profile_floor);
            double chi_q = kappa / similarity_profile(P->similarity_form, P->stability_functions, 1, h, lq, L, P->similarity_profile_floor);

            ustar = chi_u * U;
            tstar = chi_t * dtheta;
            qstar = chi_q * dq;
            ++iters;
        }

        if (P->flux_formulation == CF_FORMULATION_LARGE_YEAGER) {
            double cd_rt = sqrt(ly_cd);   /* fluxes from the coefficients of the last iteration */
            ustar = cd_rt * ly_U;
            tstar = ly_ch / cd_rt * dtheta;
            qstar = ly_ce / cd_rt * dq;
        }
        if (!wet) { /* FixedIterations on land: zeroed afterwards */
            ustar = tstar = qstar = 0.0;
            Ts = 0.0;
        }

        double dU = sqrt(du * du + dv * dv);
        double taux = (dU == 0.0) ? 0.0 : -ustar * ustar * du / dU;
        double tauy = (dU == 0.0) ? 0.0 : -ustar * ustar * dv / dU;
        double rho_a = Qa.rho;
        double cp = cp_m(t, &d, &Qa);
        double Lv = latent_heat_vapor(t, Qa.T);
        R.Qv = -rho_a * ustar * qstar * Lv;
        R.Qc = -rho_a * cp * ustar * tstar;
        R.Fv = -rho_a * ustar * qstar;
        R.rho_tau_x = rho_a * taux;
        R.rho_tau_y = rho_a * tauy;
    } else {
        ustar = tstar = qstar = 0.0;
        Ts = 0.0; /* zero_interface_state: T = 0 K */
    }
    /* convert_from_kelvin(ocean units, Ψₛ.T) */
    R.Ts_ocean_units = Ts - P->ocean_temperature_offset;
    R.ustar = ustar;
    R.theta_star = tstar;
    R.q_star = qstar;
    R.iterations = iters;
    return R;
}

static int is_wet(const cf_flux_params* P, const cf_grid* g, const void* mask, int i, int j) {
    if (P->mask_kind == CF_MASK_NONE || mask == NULL) return 1;
    size_t k = IDX(g, i, j);
    if (P->mask_kind == CF_MASK_U8) return ((const uint8_t*)mask)[k] != 0;
    return !(P->ocean_surface_z <= ((const double*)mask)[k]); /* inactive_node: z ≤ bottom height */
}

/* ------------------------------------------------------------------------------------------
 * Exported entry points (host pointers, same layouts as include/coflux.h)
 * ---------------------------------------------------------------------------------------- */
int oracle_compute_atmosphere_ocean_fluxes(const cf_grid* g, const cf_flux_params* P,
                                           const cf_ocean_surface* o, const cf_exchange_fields* a,
                                           const cf_interface_fluxes* out, int nthreads) {
    int r = g->ring;
    (void)nthreads;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int j = -r; j < g->ny + r; ++j) {
        for (int i = -r; i < g->nx + r; ++i) {
            size_t k = IDX(g, i, j);
            /* ℑxᶜᵃᵃ(i, j, k, grid, u) = (u[i] + u[i+1]) / 2 ; ℑyᵃᶜᵃ likewise */
            double uo = 0.5 * (o->u[k] + o->u[IDX(g, i + 1, j)]);
            double vo = 0.5 * (o->v[k] + o->v[IDX(g, i, j + 1)]);
            cell_result R = solve_cell(P, a->u[k], a->v[k], a->T[k], a->p[k], a->q[k], uo, vo, o->T[k],
                                       o->S[k], is_wet(P, g, o->mask, i, j));
            out->sensible_heat[k] = R.Qc;
            out->latent_heat[k] = R.Qv;
            out->water_vapor[k] = R.Fv;
            out->x_momentum[k] = R.rho_tau_x;
            out->y_momentum[k] = R.rho_tau_y;
            out->temperature[k] = R.Ts_ocean_units;
            if (out->friction_velocity) out->friction_velocity[k] = R.ustar;
            if (out->temperature_scale) out->temperature_scale[k] = R.theta_star;
            if (out->humidity_scale) out->humidity_scale[k] = R.q_star;
            if (out->iterations) out->iterations[k] = R.iterations;
        }
    }
    return 0;
}

/* Oceananigans `interpolator(fractional_idx)` + `_interpolate` restricted to 2-D, then the
 * linear blend in time of interp_atmos_time_series. */
static double interp_one(const float* data, int nsx, int nsy, int l1, int l2, double tf, double fi,
                         double fj) {
    double ti = trunc(fi), tj = trunc(fj);
    double xi = fi - floor(fi), eta = fj - floor(fj); /* ξ = mod(fractional_idx, 1) ∈ [0, 1), also for negative indices */
    long i0 = (long)ti, j0 = (long)tj;
    long i1 = i0 + (fi > 0 ? 1 : (fi < 0 ? -1 : 0));
    long j1 = j0 + (fj > 0 ? 1 : (fj < 0 ? -1 : 0));
    /* periodic in x (JRA55 longitude), clamped in y */
    i0 = ((i0 % nsx) + nsx) % nsx;
    i1 = ((i1 % nsx) + nsx) % nsx;
    if (j0 < 0) j0 = 0;
    if (j0 > nsy - 1) j0 = nsy - 1;
    if (j1 < 0) j1 = 0;
    if (j1 > nsy - 1) j1 = nsy - 1;
    double v[2];
    int lv[2] = {l1, l2};
    for (int n = 0; n < 2; ++n) {
        const float* d = data + (size_t)lv[n] * nsx * nsy;
        double a00 = d[j0 * nsx + i0], a10 = d[j0 * nsx + i1];
        double a01 = d[j1 * nsx + i0], a11 = d[j1 * nsx + i1];
        v[n] = (1.0 - xi) * (1.0 - eta) * a00 + (1.0 - xi) * eta * a01 + xi * (1.0 - eta) * a10 +
               xi * eta * a11;
    }
    return v[1] * tf + v[0] * (1.0 - tf);
}

int oracle_interpolate_atmosphere_state(const cf_grid* g, const cf_atmos_source* s,
                                        const cf_interp_weights* w, const cf_exchange_fields* out) {
    int r = g->ring;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = -r; j < g->ny + r; ++j) {
        for (int i = -r; i < g->nx + r; ++i) {
            size_t k = IDX(g, i, j);
            double fi = w->separable ? w->fi[i + g->hx] : w->fi[k];
            double fj = w->separable ? w->fj[j + g->hy] : w->fj[k];
#define ITP(var) interp_one(s->data[var], s->ns_x, s->ns_y, s->level1, s->level2, s->time_fraction, fi, fj)
            double ua = ITP(CF_JRA55_UAS), va = ITP(CF_JRA55_VAS);
            double Ta = ITP(CF_JRA55_TAS), qa = ITP(CF_JRA55_HUSS), pa = ITP(CF_JRA55_PSL);
            double Qs = ITP(CF_JRA55_RSDS), Ql = ITP(CF_JRA55_RLDS);
            double Mp = ITP(CF_JRA55_PRRA) + ITP(CF_JRA55_PRSN);
#undef ITP
            if (w->cos_rot && w->sin_rot) { /* intrinsic_vector: geographic (E,N) → grid frame */
                double c = w->cos_rot[k], sn = w->sin_rot[k];
                double ui = ua * c + va * sn;
                double vi = -ua * sn + va * c;
                ua = ui;
                va = vi;
            }
            out->u[k] = ua;
            out->v[k] = va;
            out->T[k] = Ta;
            out->p[k] = pa;
            out->q[k] = qa;
            out->Qs[k] = Qs;
            out->Ql[k] = Ql;
            out->Mp[k] = Mp;
        }
    }
    return 0;
}

/* JRA55PrescribedLand freshwater (river discharge + calving, kg m⁻² s⁻¹ on the ocean grid) for the next net-flux
 * call, or NULL (include/coflux.h: cf_set_land_freshwater; [UPSTREAM-RECALL] for how it enters JS).               */
static const double* g_land_freshwater = NULL;
void oracle_set_land_freshwater(const double* land) { g_land_freshwater = land; }

int oracle_interpolate_land_freshwater(const cf_grid* g, const cf_land_source* s, const cf_interp_weights* w, double* out) {
    int r = g->ring;
    for (int j = -r; j < g->ny + r; ++j)
        for (int i = -r; i < g->nx + r; ++i) {
            size_t k = IDX(g, i, j);
            double fi = w->separable ? w->fi[i + g->hx] : w->fi[k];
            double fj = w->separable ? w->fj[j + g->hy] : w->fj[k];
            double v = interp_one(s->friver, s->ns_x, s->ns_y, s->level1, s->level2, s->time_fraction, fi, fj);
            if (s->licalvf) v += interp_one(s->licalvf, s->ns_x, s->ns_y, s->level1, s->level2, s->time_fraction, fi, fj);
            out[k] = v;
        }
    return 0;
}

int oracle_compute_net_ocean_fluxes(const cf_grid* g, const cf_flux_params* P, const cf_ocean_surface* o,
                                    const cf_exchange_fields* a, const cf_interface_fluxes* f,
                                    const cf_sea_ice_fields* ice, const cf_interp_weights* w,
                                    const cf_net_ocean_fluxes* out) {
    const double rho_o_inv = 1.0 / P->ocean_reference_density;
    const double rho_f_inv = 1.0 / P->ocean_freshwater_density;
    const double c_o = P->ocean_heat_capacity;
    const double pi = 3.14159265358979323846;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int j = 0; j < g->ny; ++j) {
        for (int i = 0; i < g->nx; ++i) {
            size_t k = IDX(g, i, j), kw = IDX(g, i - 1, j), ks = IDX(g, i, j - 1);
            int wet = is_wet(P, g, o->mask, i, j);
            double aice = (ice && ice->concentration) ? ice->concentration[k] : 0.0;
            double aice_w = (ice && ice->concentration) ? ice->concentration[kw] : 0.0;
            double aice_s = (ice && ice->concentration) ? ice->concentration[ks] : 0.0;
            double So = o->S[k];
            double Ts = f->temperature[k] + P->ocean_temperature_offset;
            double Mp = a->Mp[k], Qs = a->Qs[k], Ql = a->Ql[k];
            double Qc = f->sensible_heat[k], Qv = f->latent_heat[k], Mv = f->water_vapor[k];

            double alb = P->ocean_albedo;
            if (P->ocean_albedo_kind == CF_ALBEDO_LATITUDE_DEPENDENT) {
                double phi = w->separable ? w->latitude[j + g->hy] : w->latitude[k];
                alb = P->ocean_albedo_diffuse - P->ocean_albedo_direct * cos(2.0 * phi * pi / 180.0);
            }
            double eps = P->ocean_emissivity;
            double Qu = eps * P->stefan_boltzmann * Ts * Ts * Ts * Ts; /* emitted longwave   */
            double Qal = -eps * Ql;                                    /* absorbed longwave  */
            double Qts = -(1.0 - alb) * Qs * (1.0 - aice);             /* transmitted SW     */
            double Qss = P->penetrating_shortwave ? 0.0 : Qts;
            double SQao = (Qu + Qc + Qv + Qal) * (1.0 - aice) + Qss;

            double SFao = -Mp * rho_f_inv + Mv * rho_f_inv;
            /* ocean_minimum_salinity: suppress freshening (ΣF < 0) below the floor, launch.sh:74-78 */
            double SFs = (So < P->ocean_minimum_salinity && SFao < 0.0) ? 0.0 : SFao;

            double Qio = (ice && ice->interface_heat) ? ice->interface_heat[k] : 0.0;
            double Jsio = (ice && ice->salt_flux) ? ice->salt_flux[k] : 0.0;
            double JTao = SQao * rho_o_inv / c_o;
            double JSao = -So * SFs;
            double JTio = Qio * rho_o_inv / c_o;

            double txao = 0.5 * (f->x_momentum[kw] + f->x_momentum[k]) * rho_o_inv;
            double tyao = 0.5 * (f->y_momentum[ks] + f->y_momentum[k]) * rho_o_inv;
            double ax = 0.5 * (aice_w + aice), ay = 0.5 * (aice_s + aice);
            double txio = (ice && ice->x_stress) ? ice->x_stress[k] : 0.0;
            double tyio = (ice && ice->y_stress) ? ice->y_stress[k] : 0.0;

            double wetf = wet ? 1.0 : 0.0; /* immersed cells carry zero flux */
            out->u[k] = wetf * ((1.0 - ax) * txao + ax * txio);
            out->v[k] = wetf * ((1.0 - ay) * tyao + ay * tyio);
            out->T[k] = wetf * (JTao + JTio);
            double SFl = g_land_freshwater ? -g_land_freshwater[k] * rho_f_inv : 0.0; /* rivers + calving, not ice-masked */
            double SFls = (So < P->ocean_minimum_salinity && SFl < 0.0) ? 0.0 : SFl;
            out->S[k] = wetf * ((1.0 - aice) * JSao + Jsio + (-So * SFls));
            if (out->shortwave_surface_flux) out->shortwave_surface_flux[k] = wetf * Qts * rho_o_inv / c_o;
            if (out->upwelling_longwave) out->upwelling_longwave[k] = wetf * Qu;
            if (out->downwelling_longwave) out->downwelling_longwave[k] = wetf * (-Qal);
            if (out->downwelling_shortwave) out->downwelling_shortwave[k] = wetf * (-Qts);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Atmosphere–sea-ice interface: compute_atmosphere_sea_ice_fluxes! with
 * SkinTemperature(ConductiveFlux) (restated from ClimaOcean ≤ v0.8 atmosphere_sea_ice_fluxes.jl /
 * interface_states.jl; parameters omip_simulation.jl:62-69, atmosphere.jl:34-44).  PARITY UNPINNED.
 * ---------------------------------------------------------------------------------------- */
static double svp_ice(const cf_thermodynamics* t, const thermo_derived* d, double T) {
    return svp_general(t, d, T, t->LH_s0, t->cp_v - t->cp_i);
}

static cell_result solve_ice_cell(const cf_flux_params* P, const cf_sea_ice_params* I, double ua, double va, double Ta,
                                  double pa, double qa, double Qs, double Ql, double ui, double vi, double So,
                                  double hi, double Ts_prev, double albedo, int wet) {
    cell_result R;
    memset(&R, 0, sizeof R);
    const cf_thermodynamics* t = &P->thermo;
    thermo_derived d = derive(t);
    const double g = P->gravitational_acceleration, kappa = P->von_karman, h = P->reference_height;
    int skip = (!wet) && (P->stop_kind == CF_STOP_CONVERGENCE);
    double ustar = 1e-4, tstar = 1e-4, qstar = 1e-4;
    double Ts = Ts_prev + I->temperature_offset;
    int iters = 0;
    if (skip) {
        R.Ts_ocean_units = 0.0 - I->temperature_offset;
        return R;
    }
    thermo_state Qa_ = phase_equil_pTq(t, &d, pa, Ta, qa);
    const double rho_a = Qa_.rho, cp = cp_m(t, &d, &Qa_), qa_v = vapor_specific_humidity(t, &d, &Qa_);
    const double Ls = t->LH_s0 + (t->cp_v - t->cp_i) * (Ta - t->T_0); /* latent_heat_sublim */
    /* bottom of the ice at the melting temperature of the liquidus at the ocean salinity; the skin is
       capped at the freshwater melting temperature under heating fluxes */
    const double Ti = I->freshwater_melting_temperature - I->liquidus_slope * So;
    const double Tm = I->freshwater_melting_temperature;
    const double heff = fmax(hi, I->consolidation_thickness);
    const double delta = d.eps - 1.0;
    double du = ua - ui, dv = va - vi;
    if (P->velocity_difference == CF_VELOCITY_WIND) {
        du = ua;
        dv = va;
    }
    const double dU = sqrt(du * du + dv * dv);
    double up = ustar, tp = tstar, qp = qstar;
    for (;;) {
        int go;
        if (P->stop_kind == CF_STOP_FIXED) {
            go = iters < P->maxiter;
        } else {
            double drift = fabs(ustar - up) + fabs(tstar - tp) + fabs(qstar - qp);
            go = (!((drift < P->tolerance) | (iters >= P->maxiter))) | (iters == 0);
        }
        if (!go) break;
        up = ustar;
        tp = tstar;
        qp = qstar;
        /* compute_interface_temperature(::SkinTemperature): surface energy balance with the scales of the
           previous iterate, upwelling longwave at the previous skin temperature */
        double Qu = I->emissivity * P->stefan_boltzmann * Ts * Ts * Ts * Ts;
        double Qd = -(1.0 - albedo) * Qs - I->emissivity * Ql;
        double Qc = -rho_a * cp * ustar * tstar;
        double Qv = -rho_a * Ls * ustar * qstar;
        double Qnet = Qv + Qu + Qc + Qd;
        double Tstar = Ti - Qnet * heff / I->conductivity; /* flux_balance_temperature(ConductiveFlux), explicit */
        if (I->skin_temperature_scheme == CF_SKIN_SEMI_IMPLICIT) /* upwelling longwave linearised about the previous Ts */
            Tstar = (Ti - (Qv + Qc + Qd) * heff / I->conductivity) /
                    (1.0 + heff / I->conductivity * I->emissivity * P->stefan_boltzmann * Ts * Ts * Ts);
        if (isnan(Tstar)) Tstar = Ts;
        double dT = fmin(fmax(Tstar - Ts, -I->maximum_temperature_change), I->maximum_temperature_change);
        Ts = fmin(Ts + dT, Tm);
        double qs = svp_ice(t, &d, Ts) / (rho_a * d.R_v * Ts);
        double dq = qa_v - qs;
        double dtheta = Ta + g * h / cp - Ts;
        thermo_state Sf = phase_equil_pTq(t, &d, pa, Ts, qs);
        double Tv = virtual_temperature(t, &d, &Sf), qv_s = vapor_specific_humidity(t, &d, &Sf);
        /* iterate_interface_fluxes */
        double bstar = g / Tv * (tstar * (1.0 + delta * qv_s) + delta * Tv * qstar);
        double Jb = -ustar * bstar;
        double U = wind_speed_scale(P, Jb, du * du + dv * dv);
        double lu = momentum_roughness(&P->momentum_roughness, g, ustar, dU, Ts);
        double lq = scalar_roughness(&P->water_vapor_roughness, lu, ustar, Ts);
        double lt = scalar_roughness(&P->temperature_roughness, lu, ustar, Ts);
        double L = (bstar == 0.0) ? INFINITY : ustar * ustar / (kappa * bstar);
        double chi_u = kappa / similarity_profile(P->similarity_form, P->stability_functions, 0, h, lu, L, P->similarity_profile_floor);
        double chi_t = kappa / similarity_profile(P->similarity_form, P->stability_functions, 1, h, lt, L, P->similarity_profile_floor);
        double chi_q = kappa / similarity_profile(P->similarity_form, P->stability_functions, 1, h, lq, L, P->similarity_profile_floor);
        ustar = chi_u * U;
        tstar = chi_t * dtheta;
        qstar = chi_q * dq;
        ++iters;
    }
    if (!wet) {
        ustar = tstar = qstar = 0.0;
        Ts = 0.0;
    }
    double taux = (dU == 0.0) ? 0.0 : -ustar * ustar * du / dU;
    double tauy = (dU == 0.0) ? 0.0 : -ustar * ustar * dv / dU;
    R.Qv = -rho_a * ustar * qstar * Ls;
    R.Qc = -rho_a * cp * ustar * tstar;
    R.Fv = -rho_a * ustar * qstar;
    R.rho_tau_x = rho_a * taux;
    R.rho_tau_y = rho_a * tauy;
    R.Ts_ocean_units = Ts - I->temperature_offset;
    R.ustar = ustar;
    R.theta_star = tstar;
    R.q_star = qstar;
    R.iterations = iters;
    return R;
}

int oracle_compute_atmosphere_sea_ice_fluxes(const cf_grid* g, const cf_flux_params* P, const cf_sea_ice_params* I,
                                             const cf_sea_ice_state* ice, const cf_ocean_surface* o,
                                             const cf_exchange_fields* a, const cf_interface_fluxes* out) {
    int r = g->ring;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int j = -r; j < g->ny + r; ++j)
        for (int i = -r; i < g->nx + r; ++i) {
            size_t k = IDX(g, i, j);
            cell_result R = solve_ice_cell(P, I, a->u[k], a->v[k], a->T[k], a->p[k], a->q[k], a->Qs[k], a->Ql[k],
                                           ice->u ? ice->u[k] : 0.0, ice->v ? ice->v[k] : 0.0, o->S[k],
                                           ice->thickness[k], ice->top_temperature[k],
                                           ice->albedo ? ice->albedo[k] : I->albedo, is_wet(P, g, o->mask, i, j));
            out->sensible_heat[k] = R.Qc;
            out->latent_heat[k] = R.Qv;
            out->water_vapor[k] = R.Fv;
            out->x_momentum[k] = R.rho_tau_x;
            out->y_momentum[k] = R.rho_tau_y;
            out->temperature[k] = R.Ts_ocean_units;
            if (out->friction_velocity) out->friction_velocity[k] = R.ustar;
            if (out->temperature_scale) out->temperature_scale[k] = R.theta_star;
            if (out->humidity_scale) out->humidity_scale[k] = R.q_star;
            if (out->iterations) out->iterations[k] = R.iterations;
        }
    return 0;
}

/* compute_net_sea_ice_fluxes! (NumericalEarth, [UPSTREAM-RECALL]; in-tree anchors atmosphere.jl:34-44 for the
 * radiative properties, omip_diagnostics.jl:86-89 for the ice–ocean terms): top and bottom heat fluxes handed to
 * the sea-ice thermodynamics.  ΣQt only where there is ice; zero on land. */
int oracle_compute_net_sea_ice_fluxes(const cf_grid* g, const cf_flux_params* P, const cf_sea_ice_params* I,
                                      const cf_sea_ice_state* ice, const cf_ocean_surface* o, const cf_exchange_fields* a,
                                      const cf_interface_fluxes* f, const double* frazil, const double* interface_heat,
                                      double* top, double* bottom) {
    for (int j = 0; j < g->ny; ++j)
        for (int i = 0; i < g->nx; ++i) {
            size_t k = IDX(g, i, j);
            double st = 0.0, sb = 0.0;
            if (is_wet(P, g, o->mask, i, j)) {
                double alb = ice->albedo ? ice->albedo[k] : I->albedo;
                double T = f->temperature[k] + I->temperature_offset;
                double Qu = I->emissivity * P->stefan_boltzmann * T * T * T * T;
                double Qd = -(1.0 - alb) * a->Qs[k] - I->emissivity * a->Ql[k];
                st = ice->concentration[k] > 0.0 ? (Qd + Qu + f->sensible_heat[k] + f->latent_heat[k]) : 0.0;
                sb = (frazil ? frazil[k] : 0.0) + (interface_heat ? interface_heat[k] : 0.0);
            }
            top[k] = st;
            bottom[k] = sb;
        }
    return 0;
}

/* NormalizeSalinity (src/OMIPConfigurations/omip_simulation.jl:182-220): `compute!(mean_total)` is
 * Oceananigans' Average over dims (1,2) — area weighted, immersed cells excluded — and
 * `parent(flux_field) .-= mean_total` subtracts it from the whole parent array. */
double oracle_normalize_salinity_flux(const cf_grid* g, const cf_flux_params* P, double* flux, const double* additional,
                                      const double* area, const void* mask) {
    double sj = 0.0, sa = 0.0;
    for (int j = 0; j < g->ny; ++j)
        for (int i = 0; i < g->nx; ++i) {
            size_t k = IDX(g, i, j);
            if (!is_wet(P, g, mask, i, j)) continue;
            double a = area ? area[k] : 1.0;
            sj += (flux[k] + (additional ? additional[k] : 0.0)) * a;
            sa += a;
        }
    double mean = sa > 0.0 ? sj / sa : 0.0;
    size_t n = (size_t)(g->nx + 2 * g->hx) * (size_t)(g->ny + 2 * g->hy);
    for (size_t k = 0; k < n; ++k) flux[k] -= mean;
    return mean;
}

/* scalar hooks for known-answer tests */
double oracle_psi_momentum(int kind, double zeta) { return psi_momentum(kind, zeta); }
double oracle_psi_scalar(int kind, double zeta) { return psi_scalar(kind, zeta); }
double oracle_saturation_vapor_pressure_liquid(const cf_flux_params* P, double T) {
    thermo_derived d = derive(&P->thermo);
    return svp_liquid(&P->thermo, &d, T);
}
double oracle_saturation_vapor_pressure_ice(const cf_flux_params* P, double T) {
    thermo_derived d = derive(&P->thermo);
    return svp_ice(&P->thermo, &d, T);
}
double oracle_water_mole_fraction(const cf_flux_params* P, double S) {
    return water_mole_fraction(&P->seawater, S);
}
double oracle_air_density(const cf_flux_params* P, double p, double T, double q) {
    thermo_derived d = derive(&P->thermo);
    thermo_state s = phase_equil_pTq(&P->thermo, &d, p, T, q);
    return s.rho;
}
/* out[10] = Qc, Qv, Fv, ρτx, ρτy, Ts, u★, θ★, q★, iterations */
int oracle_solve_cell(const cf_flux_params* P, double ua, double va, double Ta, double pa, double qa,
                      double uo, double vo, double To, double So, int wet, double* out) {
    cell_result R = solve_cell(P, ua, va, Ta, pa, qa, uo, vo, To, So, wet);
    out[0] = R.Qc;
    out[1] = R.Qv;
    out[2] = R.Fv;
    out[3] = R.rho_tau_x;
    out[4] = R.rho_tau_y;
    out[5] = R.Ts_ocean_units;
    out[6] = R.ustar;
    out[7] = R.theta_star;
    out[8] = R.q_star;
    out[9] = (double)R.iterations;
    return 0;
}
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * SeaIceAlbedo(hi, hs, Ts) — CCSM3 (reference: atmosphere.jl:30-44 hands it to SurfaceRadiationProperties; the scheme
 * is Briegleb et al. 2004, NCAR/TN-463, as CICE's `ccsm3` option states it).  [UPSTREAM-RECALL for the band average.]
 * ---------------------------------------------------------------------------------------- */
static double ccsm3_albedo(const cf_sea_ice_albedo_params* A, double hi, double hs, double Ts) {
    double fh = fmin(atan(4.0 * hi) / atan(4.0 * A->reference_thickness), 1.0);   /* thickness dependence of bare ice */
    double ocean_part = A->ocean_albedo * (1.0 - fh);
    double fT = fmin((A->melting_temperature - Ts) / A->melt_temperature_range - 1.0, 0.0); /* 0 cold … −1 melting */
    double ice_v = fmax(A->ice_visible * fh + ocean_part + A->ice_melt_change * fT, A->ocean_albedo);
    double ice_n = fmax(A->ice_near_infrared * fh + ocean_part + A->ice_melt_change * fT, A->ocean_albedo);
    double snow_v = A->snow_visible + A->snow_melt_change_visible * fT;
    double snow_n = A->snow_near_infrared + A->snow_melt_change_near_infrared * fT;
    double as = hs > 0.0 ? hs / (hs + A->snow_patch_thickness) : 0.0;              /* fractional snow cover */
    double v = ice_v * (1.0 - as) + snow_v * as, n = ice_n * (1.0 - as) + snow_n * as;
    return A->visible_fraction * v + (1.0 - A->visible_fraction) * n;
}

int oracle_sea_ice_albedo(const cf_sea_ice_albedo_params* A, long n, const double* hi, const double* hs,
                          const double* Ts, double* out) {
    for (long k = 0; k < n; ++k) out[k] = ccsm3_albedo(A, hi[k], hs ? hs[k] : 0.0, Ts[k]);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * compute_sea_ice_ocean_fluxes! with ThreeEquationHeatFlux(; friction_velocity = MomentumBasedFrictionVelocity())
 * (omip_simulation.jl:71-77) + frazil.  Three-equation interface model: Holland & Jenkins 1999 eqs. 1–3,
 * McPhee et al. 2008; see include/coflux.h for the statement.  out arrays in ocean-grid layout, interior cells.
 * ---------------------------------------------------------------------------------------- */
int oracle_sea_ice_ocean_fluxes(const cf_grid* g, const cf_flux_params* P, const cf_ice_ocean_params* Q,
                                const cf_ocean_surface* o, const double* conc, const double* tx, const double* ty,
                                double* Qio, double* Jsio, double* Qfr, double* ustar) {
    const double rho_o = P->ocean_reference_density, c_o = P->ocean_heat_capacity;
    for (int j = 0; j < g->ny; ++j)
        for (int i = 0; i < g->nx; ++i) {
            size_t k = IDX(g, i, j);
            double q_io = 0, j_io = 0, q_fr = 0, us = 0;
            int wet = 1;
            if (P->mask_kind == CF_MASK_U8 && o->mask) wet = ((const uint8_t*)o->mask)[k] != 0;
            if (P->mask_kind == CF_MASK_BOTTOM_HEIGHT && o->mask) wet = !(P->ocean_surface_z <= ((const double*)o->mask)[k]);
            if (wet) {
                double So = o->S[k], To = o->T[k], Tf = -Q->liquidus_slope * So;
                if (Q->time_step > 0.0 && To < Tf) {
                    q_fr = rho_o * c_o * Q->top_cell_thickness * (To - Tf) / Q->time_step;
                    To = Tf;
                }
                double a = conc ? conc[k] : 0.0;
                if (a > 0.0) {
                    double txc = tx ? 0.5 * (tx[k] + tx[IDX(g, i + 1, j)]) : 0.0;
                    double tyc = ty ? 0.5 * (ty[k] + ty[IDX(g, i, j + 1)]) : 0.0;
                    us = fmax(sqrt(sqrt(txc * txc + tyc * tyc)), Q->minimum_friction_velocity);
                    double ah = Q->heat_transfer_coefficient, as = Q->salt_transfer_coefficient, m = Q->liquidus_slope;
                    double gam = c_o * ah / Q->latent_heat_of_fusion;
                    double A = gam * m, B = gam * To - gam * m * Q->ice_salinity + as, C = gam * To * Q->ice_salinity + as * So;
                    double Sb = (-B + sqrt(B * B + 4.0 * A * C)) / (2.0 * A);
                    double Tb = -m * Sb;
                    q_io = a * rho_o * c_o * ah * us * (To - Tb);
                    j_io = a * as * us * (So - Sb);
                }
            }
            Qio[k] = q_io;
            Jsio[k] = j_io;
            if (Qfr) Qfr[k] = q_fr;
            if (ustar) ustar[k] = us;
        }
    return 0;
}
