// coflux_fast.hpp — the production Monin–Obukhov solver for gfx950.
//
// Same iteration, same initial guess and same stop rule as the reference path (see
// coflux_device.hpp::solve_cell, which is kept as the libm cross-check), but every
// transcendental is replaced by something CDNA4 can issue cheaply in FP64:
//   * ψ_m, ψ_h  : degree-9 piecewise polynomials in x = 1 + 16|ζ| staged in LDS, the segment taken from
//                 the exponent and top mantissa bits of x — no logarithm (coflux_tables.cpp); one
//                 instruction stream for both signs of ζ;
//   * log       : 128-entry mantissa table in LDS + degree-6 log1p polynomial;
//   * exp       : Cody–Waite reduction + degree-12 polynomial (v_ldexp_f64 to rebuild);
//   * cbrt      : v_log_f32 / v_exp_f32 seed + one FP64 Halley step;
//   * sqrt, 1/x : v_rsq_f64 / v_rcp_f64 + Newton steps, no IEEE division sequences.
// Everything is accurate to a few ulp (≤ 3e-14 for ψ), far inside the 1e-9 parity tolerance,
// and an iteration costs ≈ 200 VALU instructions instead of ≈ 2200 with ocml.
#pragma once
#include "coflux_device.hpp"
#include "coflux_tables.h"

namespace coflux {

// ---------------------------------------------------------------------------------------------
// FP64 primitives
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// one Newton step: (4.5e-8)² ≈ 2e-15 relative — used inside the iteration
__device__ __forceinline__ double frcp1(double x) {
    double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}

__device__ __forceinline__ double fsqrt1(double x) {  // x ≥ 0, ≈ 3e-15 relative
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r;
    g = __builtin_fma(__builtin_fma(-g, g, x), 0.5 * r, g);
    return x > 0.0 ? g : 0.0;
}

// a / b to ~1 ulp
__device__ __forceinline__ double fdiv(double a, double b) {
    double r = frcp(b);
    double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}

__device__ __forceinline__ double fsqrt(double x) {  // x ≥ 0
    double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x > 0.0 ? g : 0.0;
}

// natural log of a positive normal double; `logt` = LDS table of (1/c_k, log c_k)
__device__ __forceinline__ double flog(const double* logt, double x) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int e = (hi >> 20) - 1023;
    const int k = (hi >> 13) & (LOG_SEG - 1);
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);  // [1, 2)
    const double2 ck = *reinterpret_cast<const double2*>(logt + 2 * k);
    const double r = __builtin_fma(m, ck.x, -1.0);  // |r| ≤ 2^-8
    double q = __builtin_fma(r, -1.0 / 6.0, 1.0 / 5.0);
    q = __builtin_fma(r, q, -1.0 / 4.0);
    q = __builtin_fma(r, q, 1.0 / 3.0);
    q = __builtin_fma(r, q, -1.0 / 2.0);
    const double p = __builtin_fma(r * r, q, r);
    const double res = __builtin_fma((double)e, 0.6931471805599453094, ck.y + p);
    return x > 0.0 ? res : -__builtin_inf();
}

// p·r + c as ONE three-address v_fma_f64.  Written in C the compiler selects the two-address v_fmac_f64 for a
// Horner step and, because the coefficient lives on across loop iterations, copies it first (v_mov_b64 +
// v_fmac_f64: 15 such copies per solver iteration were measured).
__device__ __forceinline__ double horner_step(double p, double r, double c) {
    double out;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(out) : "v"(p), "v"(r), "v"(c));
    return out;
}

__device__ __forceinline__ double fexp(double x) {
    const double kf = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-kf, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = horner_step(p, r, 1.0 / 39916800.0);
    p = horner_step(p, r, 1.0 / 3628800.0);
    p = horner_step(p, r, 1.0 / 362880.0);
    p = horner_step(p, r, 1.0 / 40320.0);
    p = horner_step(p, r, 1.0 / 5040.0);
    p = horner_step(p, r, 1.0 / 720.0);
    p = horner_step(p, r, 1.0 / 120.0);
    p = horner_step(p, r, 1.0 / 24.0);
    p = horner_step(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const double kc = fmin(fmax(kf, -2000.0), 2000.0);
    return __builtin_amdgcn_ldexp(p, (int)kc);
}

__device__ __forceinline__ double fcbrt(double x) {  // x ≥ 0
    const float xf = (float)x;
    const float y0 = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * (1.0f / 3.0f));
    double y = (double)y0;
    const double y3 = y * y * y;
    y = y * (y3 + 2.0 * x) * frcp1(__builtin_fma(2.0, y3, x));  // Halley: cubic convergence
    return (x > 0.0 && y0 > 0.0f) ? y : 0.0;
}

// ---------------------------------------------------------------------------------------------
// tabulated stability functions
// ---------------------------------------------------------------------------------------------
struct PsiArg {
    int k;       // segment: 4·(binade of x) + top two mantissa bits, x = 1 + 16|ζ|
    double t;    // u = x − (segment start) ≥ 0: the polynomial variable (an exact subtraction)
    int side;    // 0: ζ < 0 (unstable table), 1: ζ ≥ 0
};

// No logarithm: the table is indexed by the floating-point representation of x itself (two tiers, coflux_tables.h).
// |ζ| > 1e9 (never a converged state) and NaN are evaluated at the table edge.
__device__ __forceinline__ PsiArg psi_arg_x(double x_unclamped, bool unstable) {
    PsiArg a;
    static_assert(PSI_BINADES == 34, "the clamp below is the largest double under 2^PSI_BINADES");
    const double x = fmin(x_unclamped, 0x1.fffffffffffffp33);
    const int hi = __double2hiint(x);
    const bool coarse = hi >= ((1023 + PSI_FINE_BINADES) << 20);
    const int kf = (hi >> (20 - PSI_SUB_BITS)) - (1023 << PSI_SUB_BITS);
    const int kc = (hi >> (20 - PSI_COARSE_SUB_BITS)) - (((1023 + PSI_FINE_BINADES) << PSI_COARSE_SUB_BITS) - PSI_FINE_SEG);
    a.k = coarse ? kc : kf;
    // (opaque to the compiler: it otherwise folds the exponent bias into every coefficient's address, which makes the
    // LDS offsets negative — not encodable — and costs one v_add per coefficient read)
    asm("" : "+v"(a.k));
    constexpr int FINE_MASK = (int)(0xfff00000u | (((1u << PSI_SUB_BITS) - 1u) << (20 - PSI_SUB_BITS)));
    constexpr int COARSE_MASK = (int)(0xfff00000u | (((1u << PSI_COARSE_SUB_BITS) - 1u) << (20 - PSI_COARSE_SUB_BITS)));
    a.t = x - __hiloint2double(hi & (coarse ? COARSE_MASK : FINE_MASK), 0);
    a.side = unstable ? 0 : 1;
    return a;
}
__device__ __forceinline__ PsiArg psi_arg(double zeta) {
    return psi_arg_x(__builtin_fma(PSI_A, fabs(zeta), 1.0), zeta < 0.0);
}

// fn: 0 = ψ_m, 1 = ψ_h.  Table layout: [side][coefficient][segment]{ψ_m, ψ_h} (coflux_tables.cpp).
__device__ __forceinline__ double psi_eval(const double* psi, int fn, const PsiArg& a) {
    const double* c = psi + ((size_t)(a.side * (PSI_DEG + 1)) * PSI_SEG + a.k) * 2 + fn;
    double p = c[PSI_DEG * 2 * PSI_SEG];
#pragma unroll
    for (int j = PSI_DEG - 1; j >= 0; --j) p = __builtin_fma(p, a.t, c[j * 2 * PSI_SEG]);
    return p;
}

// ψ_m and ψ_h at the same argument: 16-byte LDS reads feed both Horner chains
__device__ __forceinline__ double2 psi_eval_pair(const double* psi, const PsiArg& a) {
    const double2* c = reinterpret_cast<const double2*>(psi) + (size_t)(a.side * (PSI_DEG + 1)) * PSI_SEG + a.k;
    double2 v = c[PSI_DEG * PSI_SEG];
    double pm = v.x, ph = v.y;
#pragma unroll
    for (int j = PSI_DEG - 1; j >= 0; --j) {
        v = c[j * PSI_SEG];
        pm = __builtin_fma(pm, a.t, v.x);
        ph = __builtin_fma(ph, a.t, v.y);
    }
    return make_double2(pm, ph);
}

// ψ_m(a) and ψ_h(b) for two arguments of the same sign.  The roughness-length arguments ℓ/L★ are ≪ 1, so both
// almost always fall into the same table segment: then one chain of 16-byte reads feeds both (half the LDS time
// of two 8-byte chains).  The branch is wave-uniform; all active lanes must call this together.
__device__ __forceinline__ double2 psi_eval_mh(const double* psi, const PsiArg& a, const PsiArg& b) {
    if (__all(a.k == b.k)) {
        const double2* c = reinterpret_cast<const double2*>(psi) + (size_t)(a.side * (PSI_DEG + 1)) * PSI_SEG + a.k;
        double2 v = c[PSI_DEG * PSI_SEG];
        double pm = v.x, ph = v.y;
#pragma unroll
        for (int j = PSI_DEG - 1; j >= 0; --j) {
            v = c[j * PSI_SEG];
            pm = __builtin_fma(pm, a.t, v.x);
            ph = __builtin_fma(ph, b.t, v.y);
        }
        return make_double2(pm, ph);
    }
    return make_double2(psi_eval(psi, 0, a), psi_eval(psi, 1, b));
}

// ψ_m(a) and ψ_h(b) from the general table with the coefficient reads kept exactly two steps ahead of their FMAs.
// Written with fences because the scheduler, at the register cap, has been seen to choose either extreme for the
// plain form: every read sunk next to its use (read, wait, FMA, … twenty LDS latencies in a row) or all twenty
// reads first (forty registers, spills).  A fence with a "memory" clobber pins the LDS reads between two fences, and
// passing the accumulators through it pins the FMAs.
__device__ __forceinline__ double2 psi_eval_two(const double* psi, const PsiArg& a, const PsiArg& b) {
    constexpr int S = 2 * PSI_SEG;
    const double* ca = psi + ((size_t)(a.side * (PSI_DEG + 1)) * PSI_SEG + a.k) * 2;
    const double* cb = psi + ((size_t)(b.side * (PSI_DEG + 1)) * PSI_SEG + b.k) * 2 + 1;
    double qa[PSI_DEG + 1], qb[PSI_DEG + 1];
    constexpr int AHEAD = 2;
#pragma unroll
    for (int j = PSI_DEG; j > PSI_DEG - AHEAD; --j) {
        qa[j] = ca[j * S];
        qb[j] = cb[j * S];
    }
    double pa = 0.0, pb = 0.0;
#pragma unroll
    for (int j = PSI_DEG; j >= 0; --j) {
        asm volatile("" : "+v"(pa), "+v"(pb)::"memory");
        if (j - AHEAD >= 0) {
            qa[j - AHEAD] = ca[(j - AHEAD) * S];
            qb[j - AHEAD] = cb[(j - AHEAD) * S];
        }
        pa = j == PSI_DEG ? qa[j] : __builtin_fma(pa, a.t, qa[j]);
        pb = j == PSI_DEG ? qb[j] : __builtin_fma(pb, b.t, qb[j]);
    }
    return make_double2(pa, pb);
}

// ψ_m(zu), ψ_h(zq) for |zu|, |zq| < SMALL_Z0 and a common sign (both are a positive roughness length times 1/L★):
// degree-SMALL_DEG polynomials in |ζ|, one 16-byte LDS read per coefficient pair (two distinct addresses per wave).
__device__ __forceinline__ double2 psi_small_mh(const double* tab, bool unstable, double zu, double zq) {
    const double2* c = reinterpret_cast<const double2*>(tab + SMALL_OFFSET) + (unstable ? 0 : SMALL_DEG + 1);
    const double au = fabs(zu), aq = fabs(zq);
    double2 v = c[SMALL_DEG];
    double pm = v.x, ph = v.y;
#pragma unroll
    for (int j = SMALL_DEG - 1; j >= 0; --j) {
        v = c[j];
        pm = __builtin_fma(pm, au, v.x);
        ph = __builtin_fma(ph, aq, v.y);
    }
    return make_double2(pm, ph);
}

// exp(x) to ≈ 1e-14 relative for moderate |x| (the scalar roughness length from its logarithm; it only enters the
// iteration through ψ_h(ℓ_q/L★), a 1e-4-sized term): x = (32k' + k)·ln2/32 + r, |r| ≤ ln2/64, 2^(k/32) from LDS.
__device__ __forceinline__ double fexp_tab(const double* tab, double x) {
    const double kf = __builtin_rint(x * (EXP_SEG * 1.4426950408889634074));
    const double r = __builtin_fma(-kf, 0.6931471805599453094 / EXP_SEG, x);
    const int k = (int)kf;
    const double t = tab[EXP_OFFSET + (k & (EXP_SEG - 1))];
    double p = 1.0 / 120.0;
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(t * p, k >> 5);
}

// Cooperative copy of the tables into LDS (call once per workgroup, then __syncthreads()).
__device__ __forceinline__ void stage_tables(double* lds_tab, const double* __restrict__ g_tab, int tid, int nthreads) {
    const double2* src = reinterpret_cast<const double2*>(g_tab);
    double2* dst = reinterpret_cast<double2*>(lds_tab);
    for (int n = tid; n < TABLE_DOUBLES / 2; n += nthreads) dst[n] = src[n];
}

// ---------------------------------------------------------------------------------------------
// per-cell constants shared by the iteration
// ---------------------------------------------------------------------------------------------
// saturation_vapor_pressure(param_set, T, LH_0, Δcp) = p_tr · exp(a·log(T/T_tr) + b·(1/T_tr − 1/T)): the logarithm and the
// reciprocal difference depend on T only and are shared by every phase partition evaluated at that temperature
struct SvpArg {
    double L, D;
};
__device__ __forceinline__ SvpArg svp_arg(const DevParams& P, const double* logt, double T, double inv_T) {
    return SvpArg{flog(logt, T * P.inv_T_triple), P.inv_T_triple - inv_T};
}
__device__ __forceinline__ double svp_liquid_from(const DevParams& P, const SvpArg& s) {
    return P.p_triple * fexp(__builtin_fma(P.svp_a_liq, s.L, P.svp_b_liq * s.D));
}
// PhaseEquil: liquid-fraction weighted L₀ and Δcp.  No shortcut for λ = 1: a batch's cells come from anywhere on the
// globe, so a wave nearly always holds both kinds and would execute both sides of such a branch.
__device__ __forceinline__ double svp_equil_from(const DevParams& P, const SvpArg& s, double lam) {
    const double LH_0 = lam * P.LH_v0 + (1.0 - lam) * P.LH_s0;
    const double dcp = lam * (P.cp_v - P.cp_l) + (1.0 - lam) * (P.cp_v - P.cp_i);
    const double a = dcp * P.inv_R_v, b = (LH_0 - dcp * P.T_0) * P.inv_R_v;
    return P.p_triple * fexp(__builtin_fma(a, s.L, b * s.D));
}
__device__ __forceinline__ double svp_liquid_fast(const DevParams& P, const double* logt, double T, double inv_T) {
    return svp_liquid_from(P, svp_arg(P, logt, T, inv_T));
}

__device__ __forceinline__ double liquid_fraction_fast(const DevParams& P, const double* logt, double T) {
    double r = (T - P.T_icenuc) * P.inv_icenuc_span;
    double ramp = r;
    if (P.pow_icenuc != 1.0) ramp = r > 0.0 ? fexp(P.pow_icenuc * flog(logt, r)) : 0.0;
    return T > P.T_freeze ? 1.0 : (T > P.T_icenuc ? ramp : 0.0);
}

__device__ __forceinline__ double svp_equil_fast(const DevParams& P, const double* logt, double T, double inv_T, double lam) {
    return svp_equil_from(P, svp_arg(P, logt, T, inv_T), lam);
}

__device__ __forceinline__ AirState air_state_fast(const DevParams& P, double p, double T, double inv_T, double q_tot,
                                                   double lam, double p_vs) {
    AirState s;
    double q = fmin(fmax(q_tot, 0.0), 1.0);
    const double tiny = 2.220446049250313e-16;
    double q_vs_p = (p - p_vs >= tiny) ? P.Rd_over_Rv * (1.0 - q) * p_vs * frcp(p - p_vs) : 1.0 / tiny;
    double q_c0 = fmax(q - q_vs_p, 0.0);
    double inv_rho = P.R_d * (1.0 + P.delta * q - P.eps * q_c0) * T * frcp(p);
    s.rho = frcp(inv_rho);
    double q_vs_rho = p_vs * inv_rho * P.inv_R_v * inv_T;
    double q_c = fmax(q - q_vs_rho, 0.0);
    double q_l = lam * q_c, q_i = (1.0 - lam) * q_c;
    s.cp_m = P.cp_d + (P.cp_v - P.cp_d) * q + (P.cp_l - P.cp_v) * q_l + (P.cp_i - P.cp_v) * q_i;
    s.q_vap = fmax(0.0, q - q_l - q_i);
    s.T_virtual = (1.0 + P.delta * q - P.eps * q_c) * T;
    return s;
}

// unguarded log for arguments known to be positive and normal (everything inside the iteration)
__device__ __forceinline__ double flog_pos(const double* logt, double x) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    const int e = (hi >> 20) - 1023;
    const int k = (hi >> 13) & (LOG_SEG - 1);
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
    const double2 ck = *reinterpret_cast<const double2*>(logt + 2 * k);
    const double r = __builtin_fma(m, ck.x, -1.0);
    double q = __builtin_fma(r, -1.0 / 6.0, 1.0 / 5.0);
    q = __builtin_fma(r, q, -1.0 / 4.0);
    q = __builtin_fma(r, q, 1.0 / 3.0);
    q = __builtin_fma(r, q, -1.0 / 2.0);
    return __builtin_fma((double)e, 0.6931471805599453094, __builtin_fma(r * r, q, r) + ck.y);
}

// flog_pos in two halves, so that the table read can be started well ahead of the polynomial (the LDS latency then
// hides under whatever is computed in between); same operations, same order: bitwise flog_pos.
struct LogHalf {
    double2 ck;
    double m;
    int e;
};
__device__ __forceinline__ LogHalf flog_pos_begin(const double* logt, double x) {
    const int hi = __double2hiint(x), lo = __double2loint(x);
    LogHalf h;
    h.e = (hi >> 20) - 1023;
    const int k = (hi >> 13) & (LOG_SEG - 1);
    h.m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, lo);
    h.ck = *reinterpret_cast<const double2*>(logt + 2 * k);
    return h;
}
__device__ __forceinline__ double flog_pos_end(const LogHalf& h) {
    const double r = __builtin_fma(h.m, h.ck.x, -1.0);
    double q = __builtin_fma(r, -1.0 / 6.0, 1.0 / 5.0);
    q = __builtin_fma(r, q, -1.0 / 4.0);
    q = __builtin_fma(r, q, 1.0 / 3.0);
    q = __builtin_fma(r, q, -1.0 / 2.0);
    return __builtin_fma((double)h.e, 0.6931471805599453094, __builtin_fma(r * r, q, r) + h.ck.y);
}

// ---------------------------------------------------------------------------------------------
// Parameter split.  The iteration touches ~25 uniform scalars; the per-cell prologue touches ~60
// more.  Passing all of DevParams by value makes the compiler hoist every field into SGPRs and
// then spill them to VGPR lanes (77 v_readlane per iteration were measured).  So: LoopParams rides
// in the kernarg segment (SGPRs), and the prologue reads DevParams from an LDS copy.
// ---------------------------------------------------------------------------------------------
struct LoopParams {
    double kappa, beta_gust, h_bl, min_gust, h_ref, log_h, profile_floor, tol;
    double lm_m, const_m, log_const_m;                          // momentum roughness
    double b_q, log_A_q, log_lm_q, log_const_q;                 // water-vapour roughness
    double b_t, log_A_t, log_lm_t, log_const_t;                 // temperature roughness
    int32_t maxiter, fixed, m_kind, q_kind, t_kind, same_scalar;
    int32_t specialization;  // SOLVER_OCEAN_LEAN / SOLVER_ICE / SOLVER_GENERIC / SOLVER_LY (host-selected)
    int32_t cert_max_evals;  // certified path (coflux_certified.hpp): evaluations before a lane is sent down the exact path
    // CoefficientBasedFluxes + LargeYeagerTransferCoefficients
    double ly_min_wind, ly_zeta_bound, ly_cd0, ly_cd1, ly_cd2, ly_cd3, ly_high_wind, ly_cd_high, ly_ce, ly_ch_s, ly_ch_u;
    double ly_lz, inv_kappa;  // log(h / 10 m), 1/κ
    // the lean ocean iteration (mo_iterate_lean)
    double gust_c;            // β³ h_bl / κ:  U_G³ = −u★ (κ b★) · gust_c
    double min_gust2;         // U_G,min²
    double x_scale;           // PSI_A · h: the ψ table variable of h/L★ is 1 + x_scale·|1/L★|
    double two_inv_kappa;     // 2/κ
    // the certified reduced-iteration solve (mo_iterate_certified)
    double cert_u0, cert_two_inv_u0;  // start: u★ = cert_u0 · √(Δu² + U_G,min²); 2 / cert_u0
    double cert_chi0;                 // start: χ = κ / log(h / 1e-4 m)
    double cert_accept;               // relative residual below which the extrapolated state is accepted
    double cert_budget;               // flux-metric budget of the truncation certificate ÷ the safety factor
};

// (0 was SOLVER_OCEAN, round 2's body of the ocean presets' iteration: retired in round 6 with CF_SOLVER_TABLES_R2)
constexpr int SOLVER_ICE = 1;      // constant roughness lengths, U_G,min > 0
constexpr int SOLVER_GENERIC = 2;  // anything else (runtime kinds, u★ = 0 guards)
constexpr int SOLVER_LY = 3;       // CoefficientBasedFluxes: Large & Yeager iteration on (Cd, Ch, Ce)
constexpr int SOLVER_SEAICE = 4;   // atmosphere–sea-ice interface: skin temperature inside the iteration (ice_iterate)
constexpr int SOLVER_SEAICE_LEAN = 6;  // SOLVER_SEAICE with constant roughness lengths and U_G,min > 0 on ice_iterate_lean (coflux_lean.hpp)
constexpr int SOLVER_OCEAN_LEAN = 5;  // SOLVER_OCEAN's configurations on the round-3 iteration body (mo_iterate_lean): the default

using FastConsts = LoopParams;  // name kept for the launcher signatures

struct CellConsts {
    // iteration
    double gTv, b_theta, b_q, dU2, U_calm, dtheta, dq, alpha_g, lam_nu, inv_nu_q, inv_nu_t;
    // epilogue
    double du, dv, dU, rho, cp_m, Lv, Ts;
};

// Per-cell, iteration-invariant state.  `P` should point at the LDS copy of DevParams.
__device__ __forceinline__ CellConsts cell_prologue(const DevParams& P, double min_gust, const double* logt, double ua,
                                                    double va, double Ta, double pa, double qa, double uo, double vo,
                                                    double To, double So) {
    CellConsts c;
    const double Ts = To + P.T_offset;
    const double inv_Ta = frcp(Ta), inv_Ts = frcp(Ts);
    const double lam_a = liquid_fraction_fast(P, logt, Ta);
    const double pvs_a = svp_equil_fast(P, logt, Ta, inv_Ta, lam_a);
    const AirState A = air_state_fast(P, pa, Ta, inv_Ta, qa, lam_a, pvs_a);

    const SvpArg arg_s = svp_arg(P, logt, Ts, inv_Ts);
    const double pstar_s = svp_liquid_from(P, arg_s);
    const double sal = So * 1e-3;
    const double x_h2o = P.sw_inv_w * frcp(__builtin_fma(sal * frcp(1.0 - sal), P.sw_inv_mu, P.sw_inv_w));
    const double qs = x_h2o * pstar_s * frcp(A.rho * P.R_v * Ts);
    c.dq = A.q_vap - qs;
    c.dtheta = Ta + P.g * P.h_ref * frcp(A.cp_m) - Ts;
    double du = ua, dv = va;
    if (P.velocity_difference == CF_VELOCITY_RELATIVE) {
        du = ua - uo;
        dv = va - vo;
    }
    c.du = du;
    c.dv = dv;
    c.dU2 = du * du + dv * dv;
    c.dU = fsqrt(c.dU2);
    // from here on dU2 is what enters the wind-speed scale (DevParams::wind2_*: the shear-aware gustiness folds its
    // (c|Δu|)² + U_G,min² in; (1, 0) otherwise — the multiply-add is then exact)
    c.dU2 = __builtin_fma(c.dU2, P.wind2_scale, P.wind2_add);
    c.U_calm = fsqrt1(__builtin_fma(min_gust, min_gust, c.dU2));  // U when the gustiness sits at its floor

    const double lam_s = liquid_fraction_fast(P, logt, Ts);
    double pvs_s = pstar_s;  // (water below 0 °C — polar cells only: worth a wave-level branch, and the logarithm is shared)
    if (__any(lam_s != 1.0)) pvs_s = (lam_s == 1.0) ? pstar_s : svp_equil_from(P, arg_s, lam_s);
    const AirState Sfc = air_state_fast(P, pa, Ts, inv_Ts, qs, lam_s, pvs_s);
    c.gTv = P.g * frcp(Sfc.T_virtual);
    c.b_theta = 1.0 + P.delta * Sfc.q_vap;
    c.b_q = P.delta * Sfc.T_virtual;

    const double nu_m = air_viscosity(P.rm, Ts);
    c.inv_nu_t = frcp(air_viscosity(P.rt, Ts));
    c.inv_nu_q = frcp(air_viscosity(P.rq, Ts));
    double alpha = P.rm.charnock;
    if (P.rm.kind == CF_ROUGHNESS_WIND_CHARNOCK)
        alpha = fmax(P.rm.charnock, P.rm.wind_a1 * fmin(c.dU, P.rm.wind_umax) + P.rm.wind_a2);
    c.lam_nu = P.rm.laminar * nu_m;
    c.alpha_g = alpha * P.inv_g;
    c.rho = A.rho;
    c.cp_m = A.cp_m;
    c.Lv = P.LH_v0 + (P.cp_v - P.cp_l) * (Ta - P.T_0);
    c.Ts = Ts;
    return c;
}

// log ℓ of one scalar for the generic path
__device__ __forceinline__ double scalar_log_roughness(int kind, double b, double log_A, double log_lm, double log_const,
                                                       const double* logt, double lu, double us, double inv_nu) {
    if (kind == CF_SCALAR_ROUGHNESS_CONSTANT) return log_const;
    const double Rstar = lu * us * inv_nu;
    const double ll = fmin(__builtin_fma(-b, flog(logt, Rstar), log_A), log_lm);
    return us == 0.0 ? log_lm : ll;
}

struct Scales {
    double us, ts, qq;
    int it;    // iterations as the reference counts them (the reported trip count)
    int work;  // iterations actually executed (= it unless the sea-ice orbit shortcut ended the loop): the scheduling hint
};

// The fixed point for everything but the ocean presets (those: mo_iterate_lean, coflux_lean.hpp).  SOLVER_ICE is the
// branch-free stream for constant roughness lengths; SOLVER_GENERIC keeps every runtime switch and the u★ = 0 guards.
// All 64 lanes of a wave must call this together (wave64 ballot inside); `active` lanes iterate.
template <bool COARE, int SPEC>
__device__ __forceinline__ Scales mo_iterate(const LoopParams& L, const CellConsts& c, const double* tab, bool active) {
    const double* psi = tab;
    const double* logt = tab + LOG_OFFSET;
    double us = 1e-4, ts = 1e-4, qq = 1e-4;
    double drift = 0.0;
    int it = 0;
    for (;;) {
        bool go;
        if (L.fixed)
            go = active && it < L.maxiter;
        else
            go = active && ((it == 0) || !(drift < L.tol || it >= L.maxiter));
        if (__ballot(go) == 0ull) break;  // the wave leaves the loop together
        if (go) {
            const double bstar = c.gTv * __builtin_fma(ts, c.b_theta, c.b_q * qq);
            const double Jb = -us * bstar;
            const double inv_us = frcp1(us);
            double lu = 0.0;
            // gustiness: U_G = max(β·cbrt(max(Jᵇ,0)·h_bl), U_G,min); the cube root only where Jᵇ > 0
            double U = c.U_calm;
            if (L.beta_gust != 0.0 && __any(Jb > 0.0)) {
                const double Ug = fmax(L.beta_gust * fcbrt(fmax(Jb, 0.0) * L.h_bl), L.min_gust);
                const double Uc = fsqrt1(__builtin_fma(Ug, Ug, c.dU2));
                U = Jb > 0.0 ? Uc : c.U_calm;
            }

            double log_lu, log_lq, log_lt;
            if constexpr (SPEC == SOLVER_ICE) {
                lu = L.const_m;
                log_lu = L.log_const_m;
                log_lq = L.log_const_q;
                log_lt = L.log_const_t;
            } else {
                if (L.m_kind == CF_ROUGHNESS_CONSTANT) {
                    lu = L.const_m;
                    log_lu = L.log_const_m;
                } else {
                    const double lR = (us == 0.0) ? L.lm_m : c.lam_nu * inv_us;
                    lu = fmin(__builtin_fma(c.alpha_g * us, us, lR), L.lm_m);
                    log_lu = flog(logt, lu);
                }
                log_lq = scalar_log_roughness(L.q_kind, L.b_q, L.log_A_q, L.log_lm_q, L.log_const_q, logt, lu, us, c.inv_nu_q);
                log_lt = L.same_scalar ? log_lq
                                       : scalar_log_roughness(L.t_kind, L.b_t, L.log_A_t, L.log_lm_t, L.log_const_t, logt,
                                                              lu, us, c.inv_nu_t);
            }

            // 1/L★ = κ b★ / u★²  (0 when b★ = 0); b★ < 0 ⇒ ζ < 0 ⇒ unstable
            double inv_L = (L.kappa * bstar) * (inv_us * inv_us);
            if constexpr (SPEC == SOLVER_GENERIC) inv_L = (bstar == 0.0) ? 0.0 : inv_L;
            const PsiArg ah = psi_arg(L.h_ref * inv_L);
            const double2 psi_h2 = psi_eval_pair(psi, ah);
            double Du = L.log_h - log_lu - psi_h2.x;
            const double psi_hh = psi_h2.y;
            double Dq = L.log_h - log_lq - psi_hh;
            double Dt = L.log_h - log_lt - psi_hh;
            if constexpr (!COARE) {
                const double zu = lu * inv_L;
                Du += psi_eval(psi, 0, psi_arg(zu));
                const double psi_lq = psi_eval(psi, 1, psi_arg(fexp(log_lq) * inv_L));
                Dq += psi_lq;
                Dt += (L.same_scalar && SPEC != SOLVER_ICE) ? psi_lq : psi_eval(psi, 1, psi_arg(fexp(log_lt) * inv_L));
            }
            Du = fmax(Du, L.profile_floor);
            Dq = fmax(Dq, L.profile_floor);
            const double chi_q = L.kappa * frcp1(Dq);
            Dt = fmax(Dt, L.profile_floor);
            const double chi_t = L.kappa * frcp1(Dt);
            const double un = L.kappa * frcp1(Du) * U, tn = chi_t * c.dtheta, qn = chi_q * c.dq;
            drift = fabs(un - us) + fabs(tn - ts) + fabs(qn - qq);
            us = un;
            ts = tn;
            qq = qn;
            ++it;
        }
    }
    return Scales{us, ts, qq, it, it};
}

// CoefficientBasedFluxes(transfer_coefficients = LargeYeagerTransferCoefficients, FixedIterations(n))
// (omip_simulation.jl:86-89): the NCAR bulk algorithm of Large & Yeager (2004, 2009) on this
// package's Δθ, Δq and buoyancy scale.  Fixed trip count ⇒ no divergence, no ballot.
__device__ __forceinline__ Scales ly_iterate(const LoopParams& L, const CellConsts& c, const double* tab) {
    // The same recurrence with its square roots and divisions folded (round 3).  With C_d = C_dn / den², den = 1 + √C_dn x_m:
    //   √C_d = √C_dn / |den|,   √(C_d / C_dn) = 1 / |den|,   C_h / √C_d = c_h / (1 + c_h x_h) with c_h = 1e-3·CH(ζ) — √C_dn
    // cancels —, 1/u★² = (|den| / (√C_dn U))², 1/u₁₀ = (1 + √C_dn x_m) / U.  State: 1/√C_d, C_h/√C_d, C_e/√C_d, √C_dn.
    // Two reciprocals and one reciprocal square root per iteration instead of eight and three.
    auto cdn10 = [&](double u, double inv_u) {
        const double u2 = u * u;
        const double poly = (L.ly_cd0 * inv_u + L.ly_cd1 + L.ly_cd2 * u + L.ly_cd3 * (u2 * u2 * u2)) * 1e-3;
        return u >= L.ly_high_wind ? L.ly_cd_high * 1e-3 : poly;
    };
    auto sqrt_pair = [&](double x, double& root, double& inv_root) {  // one v_rsq_f64, one Newton step each
        const double r = __builtin_amdgcn_rsq(x);
        const double g0 = x * r, h0 = 0.5 * r;
        const double e = __builtin_fma(-g0, h0, 0.5);
        root = __builtin_fma(g0, e, g0);
        inv_root = 2.0 * __builtin_fma(h0, e, h0);
    };
    const double U = fmax(c.dU, L.ly_min_wind), inv_U = frcp1(U);
    double rt, inv_cr;
    sqrt_pair(cdn10(U, inv_U), rt, inv_cr);                       // C_d = C_dn: 1/√C_d = 1/√C_dn
    const double cek = L.ly_ce * 1e-3;
    double th = (c.dtheta > 0.0 ? L.ly_ch_s : L.ly_ch_u) * 1e-3;  // C_h/√C_d, C_e/√C_d: the neutral coefficients' factors
    double qh = cek;
    double den = 1.0;
    for (int it = 0; it < L.maxiter; ++it) {
        const double ts = th * c.dtheta, qq = qh * c.dq;                                   // L-Y eq. 7
        const double b = c.gTv * __builtin_fma(ts, c.b_theta, c.b_q * qq);
        const double w = inv_cr * inv_U;                                                   // 1/u★
        double z = (L.kappa * L.h_ref) * b * (w * w);                                      // 8a
        z = __builtin_copysign(fmin(fabs(z), L.ly_zeta_bound), z);
        const double2 ps = psi_eval_pair(tab, psi_arg(z));
        const double xm = (L.ly_lz - ps.x) * L.inv_kappa, xh = (L.ly_lz - ps.y) * L.inv_kappa;
        const double d0 = __builtin_fma(rt, xm, 1.0);                                      // with the PREVIOUS √C_dn
        const double u10 = U * frcp1(d0);                                                  // 9
        double irt;
        sqrt_pair(cdn10(u10, d0 * inv_U), rt, irt);
        den = __builtin_fma(rt, xm, 1.0);
        inv_cr = fabs(den) * irt;                                                          // 10a: √C_d = √C_dn / |den|
        const double chk = (z > 0.0 ? L.ly_ch_s : L.ly_ch_u) * 1e-3;
        const double a = __builtin_fma(chk, xh, 1.0), e = __builtin_fma(cek, xh, 1.0);
        const double r = frcp1(a * e);
        th = chk * (r * e);                                                                // 10b, 10c divided by √C_d
        qh = cek * (r * a);
    }
    const double cr = frcp1(inv_cr);
    return Scales{cr * U, th * c.dtheta, qh * c.dq, L.maxiter, L.maxiter};
}

// ---------------------------------------------------------------------------------------------
// Atmosphere–sea-ice interface: the same iteration with a skin temperature inside the loop
// (SkinTemperature(ConductiveFlux): surface energy balance against conduction through the ice,
// limited to ±ΔTmax per iteration and capped at the freshwater melting point), saturation over ice,
// sublimation enthalpy.  The surface state changes every iteration, so its thermodynamics cannot
// be hoisted; roughness lengths follow the runtime kinds (constant in the production presets).
// ---------------------------------------------------------------------------------------------
struct IceParams {  // kernarg
    double hk_min;      // consolidation thickness / conductivity
    double inv_k;       // 1 / conductivity
    double dT_max, T_melt, T_fw, liquidus_slope, eps_sigma, emissivity, albedo, T_offset;
    double semi_implicit;  // 1: upwelling longwave linearised about the previous skin temperature (CF_SKIN_SEMI_IMPLICIT)
    double orbit_shortcut; // 1 (default): an exact period-2 orbit ends the iteration early (CF_OPT_ICE_ORBIT_SHORTCUT)
    double ice_free_zero;  // 1: cells with ℵ = 0 and hᵢ = 0 get zero_interface_state instead of an iteration (CF_OPT_ICE_FREE_CELLS)
};

struct IceConsts {
    double rho, cp, qav, Ls, Ti, hk, Qd, theta_a, pa, du, dv, dU2, dU, alpha_g;
};

__device__ __forceinline__ double svp_ice_from(const DevParams& P, const SvpArg& s) {
    const double dcp = P.cp_v - P.cp_i;
    const double a = dcp * P.inv_R_v, b = (P.LH_s0 - dcp * P.T_0) * P.inv_R_v;
    return P.p_triple * fexp(__builtin_fma(a, s.L, b * s.D));
}
__device__ __forceinline__ double svp_ice_fast(const DevParams& P, const double* logt, double T, double inv_T) {
    return svp_ice_from(P, svp_arg(P, logt, T, inv_T));
}

template <bool COARE>
__device__ __forceinline__ Scales ice_iterate(const DevParams& P, const LoopParams& L, const IceParams& I,
                                              const IceConsts& c, const double* tab, bool active, double& Ts) {
    const double* logt = tab + LOG_OFFSET;
    double us = 1e-4, ts = 1e-4, qq = 1e-4, drift = 0.0;
    int it = 0;
    // The state two iterations ago.  Where the skin-temperature balance does not contract (thick ice in wind: most of
    // the polar cells) the iteration does not wander: clamped by dT_max it falls into a period-2 orbit,
    // and in floating point the orbit becomes EXACT after a few dozen iterations (state(k) = state(k−2) bit for bit;
    // median k = 20 on the synthetic polar surface, 90 % of the cells that never converge).  From there on the
    // remaining iterations up to maxiter are known: the stopped iterate is state(k) or state(k−1) by parity.
    double us_2 = -1.0, ts_2 = 0.0, qq_2 = 0.0, Ts_2 = 0.0;
    int work = 0;
    for (;;) {
        bool go;
        if (L.fixed)
            go = active && it < L.maxiter;
        else
            go = active && ((it == 0) || !(drift < L.tol || it >= L.maxiter));
        if (__ballot(go) == 0ull) break;
        if (go) {
            const double us_1 = us, ts_1 = ts, qq_1 = qq, Ts_1 = Ts;  // state(it)
            // skin temperature from the energy balance with the previous scales
            const double T2 = Ts * Ts;
            const double rho_u = c.rho * us;
            double Tstar;
            if (I.semi_implicit != 0.0) {
                // k (Ti − T★)/h = Q_v + Q_c + Q_d + εσ Ts³ T★ : implicit in one factor of the upwelling longwave
                const double Qrest = -rho_u * c.Ls * qq - rho_u * c.cp * ts + c.Qd;
                Tstar = __builtin_fma(-Qrest, c.hk, c.Ti) * frcp(__builtin_fma(c.hk * I.eps_sigma, T2 * Ts, 1.0));
            } else {
                const double Qnet = -rho_u * c.Ls * qq + I.eps_sigma * T2 * T2 - rho_u * c.cp * ts + c.Qd;
                Tstar = __builtin_fma(-Qnet, c.hk, c.Ti);
            }
            Tstar = (Tstar != Tstar) ? Ts : Tstar;
            const double dT = fmin(fmax(Tstar - Ts, -I.dT_max), I.dT_max);
            Ts = fmin(Ts + dT, I.T_melt);
            const double inv_Ts = frcp(Ts);
            const SvpArg arg_s = svp_arg(P, logt, Ts, inv_Ts);  // one logarithm for both saturation pressures at Ts
            const double qs = svp_ice_from(P, arg_s) * frcp(c.rho * P.R_v * Ts);
            const double dq = c.qav - qs, dtheta = c.theta_a - Ts;
            const double lam_s = liquid_fraction_fast(P, logt, Ts);
            const double pvs_s = svp_equil_from(P, arg_s, lam_s);
            const AirState Sfc = air_state_fast(P, c.pa, Ts, inv_Ts, qs, lam_s, pvs_s);
            const double gTv = P.g * frcp(Sfc.T_virtual);

            const double bstar = gTv * __builtin_fma(ts, 1.0 + P.delta * Sfc.q_vap, P.delta * Sfc.T_virtual * qq);
            const double Jb = -us * bstar;
            double Ug = L.min_gust;
            if (L.beta_gust != 0.0) Ug = fmax(L.beta_gust * fcbrt(fmax(Jb, 0.0) * L.h_bl), L.min_gust);
            const double U = fsqrt1(__builtin_fma(Ug, Ug, c.dU2));

            const double inv_us = frcp1(us);
            double lu, log_lu;
            if (L.m_kind == CF_ROUGHNESS_CONSTANT) {
                lu = L.const_m;
                log_lu = L.log_const_m;
            } else {
                const double lam_nu = P.rm.laminar * air_viscosity(P.rm, Ts);
                const double lR = (us == 0.0) ? L.lm_m : lam_nu * inv_us;
                lu = fmin(__builtin_fma(c.alpha_g * us, us, lR), L.lm_m);
                log_lu = flog(logt, lu);
            }
            double log_lq = L.log_const_q, log_lt = L.log_const_t;
            if (L.q_kind != CF_SCALAR_ROUGHNESS_CONSTANT)
                log_lq = scalar_log_roughness(L.q_kind, L.b_q, L.log_A_q, L.log_lm_q, L.log_const_q, logt, lu, us,
                                              frcp(air_viscosity(P.rq, Ts)));
            if (L.t_kind != CF_SCALAR_ROUGHNESS_CONSTANT)
                log_lt = L.same_scalar ? log_lq
                                       : scalar_log_roughness(L.t_kind, L.b_t, L.log_A_t, L.log_lm_t, L.log_const_t, logt,
                                                              lu, us, frcp(air_viscosity(P.rt, Ts)));

            const double inv_L = (bstar == 0.0) ? 0.0 : (L.kappa * bstar) * (inv_us * inv_us);
            const double2 psi_h2 = psi_eval_pair(tab, psi_arg(L.h_ref * inv_L));
            double Du = L.log_h - log_lu - psi_h2.x;
            double Dq = L.log_h - log_lq - psi_h2.y;
            double Dt = L.log_h - log_lt - psi_h2.y;
            if constexpr (!COARE) {
                Du += psi_eval(tab, 0, psi_arg(lu * inv_L));
                Dq += psi_eval(tab, 1, psi_arg(fexp(log_lq) * inv_L));
                Dt += psi_eval(tab, 1, psi_arg(fexp(log_lt) * inv_L));
            }
            Du = fmax(Du, L.profile_floor);
            Dq = fmax(Dq, L.profile_floor);
            Dt = fmax(Dt, L.profile_floor);
            const double un = L.kappa * frcp1(Du) * U, tn = L.kappa * frcp1(Dt) * dtheta, qn = L.kappa * frcp1(Dq) * dq;
            drift = fabs(un - us) + fabs(tn - ts) + fabs(qn - qq);
            us = un;
            ts = tn;
            qq = qn;
            ++it;
            work = it;
            if (I.orbit_shortcut != 0.0 && us == us_2 && ts == ts_2 && qq == qq_2 && Ts == Ts_2 && !(drift < L.tol)) {
                // exact period 2: jump to the last iteration (an odd number of steps left lands on the other state of the orbit)
                if ((L.maxiter - it) & 1) {
                    us = us_1;
                    ts = ts_1;
                    qq = qq_1;
                    Ts = Ts_1;
                }
                it = L.maxiter;
            }
            us_2 = us_1;
            ts_2 = ts_1;
            qq_2 = qq_1;
            Ts_2 = Ts_1;
        }
    }
    return Scales{us, ts, qq, it, work};
}

__device__ __forceinline__ CellFluxes cell_epilogue(const CellConsts& c, double T_offset, Scales s) {
    CellFluxes R;
    const double inv_dU = (c.dU == 0.0) ? 0.0 : frcp(c.dU);
    const double tau = -s.us * s.us * inv_dU;
    const double rho_u = c.rho * s.us;
    R.Fv = -rho_u * s.qq;
    R.Qv = R.Fv * c.Lv;
    R.Qc = -rho_u * c.cp_m * s.ts;
    R.rho_tau_x = c.rho * tau * c.du;
    R.rho_tau_y = c.rho * tau * c.dv;
    R.Ts_ocean = c.Ts - T_offset;
    R.ustar = s.us;
    R.tstar = s.ts;
    R.qstar = s.qq;
    R.iterations = s.it;
    return R;
}

}  // namespace coflux
