// coflux_lean.hpp — the round-3 body of the atmosphere–ocean Monin–Obukhov iteration (SOLVER_OCEAN_LEAN).
//
// Same fixed point, same initial guess, same update order and same stop rule as mo_iterate<…, SOLVER_OCEAN>
// (coflux_fast.hpp) — the reference's iteration, SURVEY §8 a5–a11; omip_simulation.jl:40-49 — restated so that one
// iteration costs the CU less of the two pipes that bound it (measured, scratch/ubench_valu.hip: an FP64 VALU
// instruction is 4 SIMD cycles, v_rcp/v_rsq_f64 16, a scattered ds_read_b128 10–14 cycles of the CU's one LDS pipe):
//   * the state carries 1/u★ beside u★: u★ ← κU/D_u and 1/u★ ← D_u/(κU) both come out of ONE v_rsq_f64 of U²
//     (√ and 1/√ by a coupled Newton step) — no reciprocal of u★;
//   * 1/D_u and 1/D_q from one reciprocal of D_u·D_q;
//   * κ b★ = θ★·bθ + q★·b_q with κ g/𝒯ᵥ folded into the two per-cell constants; U_G² = (β³ h_bl J_b)^⅔ from a
//     v_log_f32/v_exp_f32 seed and ONE Newton step on y³ = w² whose divisor 1/(3y²) only needs the seed's seven
//     digits (v_rcp_f32) — no FP64 reciprocal, no select for the stable lanes (the floor U_G,min² takes them);
//   * ψ_m, ψ_h(h/L★): degree-6 two-tier tables (coflux_tables.h): seven 16-byte LDS reads and twelve FMAs;
//   * log: 128-entry table + degree-4 log1p (|r| ≤ 2⁻⁸ ⇒ ≤ 1.8e-13 absolute); exp for ℓ_q: 32-entry table +
//     degree-3 (6e-10 relative on a roughness length that enters through ψ_h(ℓ_q/L★) ≈ 1e-3 only);
//   * ψ at the roughness-length arguments: degree 3 below |ζ| < 2⁻¹⁰ (per lane), the general table otherwise.
// Accuracy: every iterate is within ≈ 3e-13 relative of the libm evaluation of the same map; the error budget
// and its effect on the stop rule are measured in scratch/lean_study.py (a 1e-12 perturbation of every iterate
// changes no trip count in 36 664 cells; 1e-11 changes 2e-4 of them) and held by tests/test_full_size.py.
#pragma once
#include "coflux_fast.hpp"

namespace coflux {

// log of a positive normal double from flog_pos_begin's table read: degree-4 log1p
__device__ __forceinline__ double flog_lean_end(const LogHalf& h) {
    // (Estrin form: −½ + r/3 first, then −r²/4 onto it — as a Horner chain the compiler copies the constant ⅓ into the
    // accumulator of a two-address multiply-add in every evaluation)
    const double r = __builtin_fma(h.m, h.ck.x, -1.0);
    const double r2 = r * r;
    const double q = __builtin_fma(r2, -0.25, __builtin_fma(r, 1.0 / 3.0, -0.5));
    return __builtin_fma((double)h.e, 0.6931471805599453094, __builtin_fma(r2, q, r) + h.ck.y);
}

// w^(2/3) for w ≥ 1e-18: f32 seed (≈ 5e-7), one Newton step on y³ = w² in FP64 (⇒ ≈ 3e-13)
__device__ __forceinline__ double pow23_lean(double w) {
    const float wf = (float)w;
    const float y0f = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(wf) * (2.0f / 3.0f));
    const float rf = __builtin_amdgcn_rcpf(3.0f * y0f * y0f);
    const double y0 = (double)y0f;
    const double t = __builtin_fma(-(y0 * y0), y0, w * w);
    return __builtin_fma(t, (double)rf, y0);
}

// g ≈ √s and h ≈ 1/(2√s) for s > 0 from one v_rsq_f64 and one coupled Newton (Goldschmidt) step
__device__ __forceinline__ void sqrt_rsqrt_lean(double s, double& g, double& h) {
    const double r = __builtin_amdgcn_rsq(s);
    const double g0 = s * r, h0 = 0.5 * r;
    const double e = __builtin_fma(-g0, h0, 0.5);
    g = __builtin_fma(g0, e, g0);
    h = __builtin_fma(h0, e, h0);
}

// exp(x), relative ≤ 6e-10: x = (32k' + k)·ln2/32 + r, |r| ≤ ln2/64, 2^(k/32) from LDS, degree 3
__device__ __forceinline__ double fexp_lean(const double* tab, double x) {
    const double kf = __builtin_rint(x * (EXP_SEG * 1.4426950408889634074));
    const double r = __builtin_fma(-kf, 0.6931471805599453094 / EXP_SEG, x);
    const int k = (int)kf;
    const double t = tab[EXP_OFFSET + (k & (EXP_SEG - 1))];
    double p = __builtin_fma(r, 1.0 / 6.0, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(t * p, k >> 5);
}

// max / min against a wave-uniform, NaN-free bound held in SGPRs.  Written as instructions because the C form
// (fmax / fmin under IEEE mode) first canonicalises the bound — v_max_f64 s, s into a VGPR pair — on every call.
__device__ __forceinline__ double vmax_u(double x, double bound) {
    double o;
    asm("v_max_f64 %0, %1, %2" : "=v"(o) : "v"(x), "s"(bound));
    return o;
}
__device__ __forceinline__ double vmin_u(double x, double bound) {
    double o;
    asm("v_min_f64 %0, %1, %2" : "=v"(o) : "v"(x), "s"(bound));
    return o;
}

// exp(x) to ≈ 2e-15 relative: fexp_tab's reduction (32-entry table) with a degree-5 tail — the prologue's saturation
// pressures (a third of fexp's instructions)
__device__ __forceinline__ double fexp_tab5(const double* tab, double x) { return fexp_tab(tab, x); }

// liquid_fraction_fast / svp_equil_from with the table exponential: fexp's twelve Taylor coefficients are literals the
// compiler materialises ahead of the batch loop and then keeps in SCRATCH across the iteration (measured: 12 spilled
// registers reloaded in every prologue); the table form has none.
__device__ __forceinline__ double liquid_fraction_lean(const DevParams& P, const double* tab, double T) {
    const double r = (T - P.T_icenuc) * P.inv_icenuc_span;
    double ramp = r;
    if (P.pow_icenuc != 1.0) ramp = r > 0.0 ? fexp_tab(tab, P.pow_icenuc * flog(tab + LOG_OFFSET, r)) : 0.0;
    return T > P.T_freeze ? 1.0 : (T > P.T_icenuc ? ramp : 0.0);
}
__device__ __forceinline__ double svp_equil_lean(const DevParams& P, const double* tab, const SvpArg& s, double lam) {
    const double LH_0 = lam * P.LH_v0 + (1.0 - lam) * P.LH_s0;
    const double dcp = lam * (P.cp_v - P.cp_l) + (1.0 - lam) * (P.cp_v - P.cp_i);
    const double a = dcp * P.inv_R_v, b = (LH_0 - dcp * P.T_0) * P.inv_R_v;
    return P.p_triple * fexp_tab(tab, __builtin_fma(a, s.L, b * s.D));
}

// Per-cell, iteration-invariant state of the lean ocean path: what the iteration reads, and what the epilogue needs to
// turn the converged scales into fluxes (five numbers instead of the seven of CellConsts: ρ and 1/‖Δu‖ are folded in).
struct LeanCell {
    double bth, bqq, dU2, dtheta, dq, alpha_g, lam_nu, inv_nu_q;  // iteration
    double rdu, rdv, rcp, rho, Lv;                                // epilogue: ρ Δu/‖Δu‖, ρ Δv/‖Δu‖, ρ c_p, ρ, ℒv
    double Ts;                                                    // interface temperature [K] (written before the iteration)
};

// The same relations as cell_prologue / air_state_fast (Thermodynamics.jl PhaseEquil_pTq, Raoult factor of sea water),
// with the reciprocals shared (ten instead of fifteen, one Newton step each: 2e-15), the exponentials from the table
// form and nothing computed that SOLVER_OCEAN_LEAN's preconditions make unnecessary (one scalar roughness, U_G,min > 0).
// `P` should point at the LDS copy of DevParams.
__device__ __forceinline__ LeanCell lean_prologue(const DevParams& P, double kappa, const double* tab, double ua, double va,
                                                  double Ta, double pa, double qa, double uo, double vo, double To, double So) {
    const double* logt = tab + LOG_OFFSET;
    LeanCell c;
    const double Ts = To + P.T_offset;
    // 1/Ta, 1/Ts from one reciprocal; 1/p
    const double rT = frcp1(Ta * Ts);
    const double inv_Ta = rT * Ts, inv_Ts = rT * Ta;
    const double inv_p = frcp1(pa);
    const double tiny = 2.220446049250313e-16;
    // ---- air at the reference height: PhaseEquil_pTq(pa, Ta, qa) ----
    const double lam_a = liquid_fraction_lean(P, tab, Ta);
    const SvpArg arg_a = svp_arg(P, logt, Ta, inv_Ta);
    double pvs_a;
    {
        const double LH_0 = lam_a * P.LH_v0 + (1.0 - lam_a) * P.LH_s0;
        const double dcp = lam_a * (P.cp_v - P.cp_l) + (1.0 - lam_a) * (P.cp_v - P.cp_i);
        const double a = dcp * P.inv_R_v, b = (LH_0 - dcp * P.T_0) * P.inv_R_v;
        pvs_a = P.p_triple * fexp_tab5(tab, __builtin_fma(a, arg_a.L, b * arg_a.D));
    }
    const double q = fmin(fmax(qa, 0.0), 1.0);
    const double dp_a = pa - pvs_a;
    const double qvsp_a = (dp_a >= tiny) ? P.Rd_over_Rv * (1.0 - q) * pvs_a * frcp1(dp_a) : 1.0 / tiny;
    const double qc0_a = fmax(q - qvsp_a, 0.0);
    const double inv_rho = P.R_d * (1.0 + P.delta * q - P.eps * qc0_a) * Ta * inv_p;
    const double rho = frcp1(inv_rho);
    const double qc_a = fmax(q - pvs_a * inv_rho * P.inv_R_v * inv_Ta, 0.0);
    const double ql_a = lam_a * qc_a, qi_a = (1.0 - lam_a) * qc_a;
    const double cp_m = P.cp_d + (P.cp_v - P.cp_d) * q + (P.cp_l - P.cp_v) * ql_a + (P.cp_i - P.cp_v) * qi_a;
    const double qvap_a = fmax(0.0, q - ql_a - qi_a);
    // ---- the sea surface: saturated over salt water at Ts ----
    const SvpArg arg_s = svp_arg(P, logt, Ts, inv_Ts);
    const double pstar_s = P.p_triple * fexp_tab5(tab, __builtin_fma(P.svp_a_liq, arg_s.L, P.svp_b_liq * arg_s.D));
    const double sal = So * 1e-3, fresh = 1.0 - sal;
    const double x_h2o = (P.sw_inv_w * fresh) * frcp1(__builtin_fma(sal, P.sw_inv_mu, P.sw_inv_w * fresh));
    const double qs = x_h2o * pstar_s * (inv_rho * P.inv_R_v * inv_Ts);
    c.dq = qvap_a - qs;
    const double inv_cp = frcp1(cp_m);
    c.dtheta = Ta + (P.g * P.h_ref) * inv_cp - Ts;
    double du = ua, dv = va;
    if (P.velocity_difference == CF_VELOCITY_RELATIVE) {
        du = ua - uo;
        dv = va - vo;
    }
    c.dU2 = du * du + dv * dv;
    // 1/‖Δu‖ (0 in a dead calm) from v_rsq_f64 + one Newton step; ‖Δu‖ = Δu²·(1/‖Δu‖)
    double inv_dU = 0.0, dU = 0.0;
    {
        const double r = __builtin_amdgcn_rsq(c.dU2);
        const double g0 = c.dU2 * r, h0 = 0.5 * r;
        const double e = __builtin_fma(-g0, h0, 0.5);
        const bool moving = c.dU2 > 0.0;
        dU = moving ? __builtin_fma(g0, e, g0) : 0.0;
        inv_dU = moving ? 2.0 * __builtin_fma(h0, e, h0) : 0.0;
    }
    // PhaseEquil_pTq(pa, Ts, qs): virtual temperature and vapour of the surface air
    const double lam_s = liquid_fraction_lean(P, tab, Ts);
    double pvs_s = pstar_s;  // (water below 0 °C — polar cells only: worth a wave-level branch, and the logarithm is shared)
    if (__any(lam_s != 1.0)) pvs_s = (lam_s == 1.0) ? pstar_s : svp_equil_lean(P, tab, arg_s, lam_s);
    const double qss = fmin(fmax(qs, 0.0), 1.0);
    const double dp_s = pa - pvs_s;
    const double qvsp_s = (dp_s >= tiny) ? P.Rd_over_Rv * (1.0 - qss) * pvs_s * frcp1(dp_s) : 1.0 / tiny;
    const double qc0_s = fmax(qss - qvsp_s, 0.0);
    const double inv_rho_s = P.R_d * (1.0 + P.delta * qss - P.eps * qc0_s) * Ts * inv_p;
    const double qc_s = fmax(qss - pvs_s * inv_rho_s * P.inv_R_v * inv_Ts, 0.0);
    const double qvap_s = fmax(0.0, qss - lam_s * qc_s - (1.0 - lam_s) * qc_s);
    const double Tv_s = (1.0 + P.delta * qss - P.eps * qc_s) * Ts;
    const double kg = (kappa * P.g) * frcp1(Tv_s);  // κ g / 𝒯ᵥ
    c.bth = kg * (1.0 + P.delta * qvap_s);
    c.bqq = kg * (P.delta * Tv_s);
    // roughness
    const double nu_m = air_viscosity(P.rm, Ts);
    c.inv_nu_q = frcp1(air_viscosity(P.rq, Ts));
    double alpha = P.rm.charnock;
    if (P.rm.kind == CF_ROUGHNESS_WIND_CHARNOCK) alpha = fmax(P.rm.charnock, P.rm.wind_a1 * fmin(dU, P.rm.wind_umax) + P.rm.wind_a2);
    c.lam_nu = P.rm.laminar * nu_m;
    c.alpha_g = alpha * P.inv_g;
    c.dU2 = __builtin_fma(c.dU2, P.wind2_scale, P.wind2_add);  // from here on: what enters the wind-speed scale (DevParams::wind2_*)
    // epilogue
    const double rho_dir = rho * inv_dU;
    c.rdu = rho_dir * du;
    c.rdv = rho_dir * dv;
    c.rcp = rho * cp_m;
    c.rho = rho;
    c.Lv = P.LH_v0 + (P.cp_v - P.cp_l) * (Ta - P.T_0);
    c.Ts = Ts;
    return c;
}

// All 64 lanes of a wave must call this together (wave64 ballot inside); `active` lanes iterate.
// Preconditions (loop_params selects SOLVER_OCEAN_LEAN only then): U_G,min > 0 (⇒ u★ > 0), Charnock-type momentum
// roughness, identical Reynolds-scaled scalar roughness lengths.  FixedIterations(n) arrives as maxiter = n, tol = 0.
// ONE_COMPARE = false: the loop test as `active && !(drift < tol)` — the certified kernels' exact-path batch, whose launch
// measured 1.5 % slower with the single compare (register allocation around the queue logic; profiles/r05_experiments.md §11)
template <bool COARE, bool ONE_COMPARE = true>
__device__ __forceinline__ Scales mo_iterate_lean(const LoopParams& L, const LeanCell& c, const double* tab, bool active) {
    const double* logt = tab + LOG_OFFSET;
    // θ★ = χ Δθ and q★ = χ Δq share one transfer coefficient χ = κ / D_q from the first iterate on (one scalar
    // roughness length, one stability function), so the state is (u★, 1/u★, χ) and κ b★ = χ·(bθ Δθ + b_q Δq); the drift of
    // the two scalars is |Δχ|·(|Δθ| + |Δq|).  Five FP64 instructions per iteration fewer; the first iterate, which starts
    // from θ★ = q★ = 1e-4, takes the general expressions (a wave-uniform branch: every running lane has the same count).
    const double B = __builtin_fma(c.dtheta, c.bth, c.bqq * c.dq), S = fabs(c.dtheta) + fabs(c.dq);
    double us = 1e-4, ius = 1e4, chi = 0.0, kb = __builtin_fma(1e-4, c.bth, c.bqq * 1e-4);
    // (a lane that does not take part starts BELOW every tolerance and never runs: the loop's test is
    // then ONE compare whose mask is the ballot; `active && …` made the compiler rebuild the mask through v_cndmask / v_cmp_ne
    // and two more VALU → SALU hand-offs per trip)
    double drift = (ONE_COMPARE && !active) ? -__builtin_inf() : __builtin_inf();
    int it = 0;
    double log_A_q = L.log_A_q;  // in a vector register: as the third scalar operand of one multiply-add it would be copied there per iteration
    asm("" : "+v"(log_A_q));
    for (int trip = 0;; ++trip) {
        // (every running lane has it == trip: the trip limit is a scalar test)
        const bool go = (ONE_COMPARE || active) && !(drift < L.tol);
        if (trip >= L.maxiter || __builtin_amdgcn_ballot_w64(go) == 0ull) break;  // the wave leaves the loop together
        if (go) {
            const double inv_L = kb * (ius * ius);  // 1/L★ = κ b★ / u★²
            const double lu = vmin_u(__builtin_fma(c.alpha_g * us, us, c.lam_nu * ius), L.lm_m);
            const LogHalf half_u = flog_pos_begin(logt, lu);
            const LogHalf half_q = flog_pos_begin(logt, lu * us * c.inv_nu_q);
            const PsiArg ah = psi_arg_x(__builtin_fma(L.x_scale, fabs(inv_L), 1.0), kb < 0.0);
            // wind speed scale with gustiness: U² = Δu² + max((β³ h_bl J_b)^⅔, U_G,min²); J_b = −u★ b★ > 0 ⇔ unstable
            double U, rU;  // rU = 1/(2U)
            if (L.beta_gust != 0.0) {
                const double w = fmax(-(us * kb) * L.gust_c, 1e-18);
                sqrt_rsqrt_lean(c.dU2 + vmax_u(pow23_lean(w), L.min_gust2), U, rU);
            } else {
                sqrt_rsqrt_lean(c.dU2 + L.min_gust2, U, rU);
            }
            const double2 ps = psi_eval_pair(tab, ah);
            const double log_lu = flog_lean_end(half_u);
            const double log_lq = vmin_u(__builtin_fma(-L.b_q, flog_lean_end(half_q), log_A_q), L.log_lm_q);
            double Du = (L.log_h - log_lu) - ps.x;
            double Dq = (L.log_h - log_lq) - ps.y;
            if constexpr (!COARE) {
                const double zu = lu * inv_L, zq = fexp_lean(tab, log_lq) * inv_L;
                // per-lane choice (a cell's result never depends on which cells share its wave); a wave whose lanes all
                // agree — four out of five — executes only one side
                double2 pl;
                if (fabs(zu) < SMALL_Z0 && fabs(zq) < SMALL_Z0) {
                    asm volatile("" ::: "memory");
                    pl = psi_small_mh(tab, inv_L < 0.0, zu, zq);
                } else {
                    asm volatile("" ::: "memory");
                    pl = psi_eval_two(tab, psi_arg(zu), psi_arg(zq));
                }
                Du += pl.x;
                Dq += pl.y;
            }
            Du = vmax_u(Du, L.profile_floor);
            Dq = vmax_u(Dq, L.profile_floor);
            const double r = frcp1(Du * Dq);
            const double kU = L.kappa * U, rDq = r * Dq, rDu = r * Du;
            ius = (Du * rU) * L.two_inv_kappa;  // 1/u★ = D_u / (κ U)
            // the differences as fused multiply-subtracts of the OLD state, then the new state over it: no register copies
            // for the loop-carried values (un − u★ differs from the two-step form by half an ulp of u★)
            const double d_u = __builtin_fma(kU, rDq, -us);
            double kU_after = kU;
            asm("" : "+v"(kU_after) : "v"(d_u));  // (orders the product behind the last use of the old u★)
            us = kU_after * rDq;
            if (trip == 0) {
                chi = L.kappa * rDu;
                drift = fabs(d_u) + fabs(chi * c.dtheta - 1e-4) + fabs(chi * c.dq - 1e-4);
            } else {
                const double d_c = __builtin_fma(L.kappa, rDu, -chi);
                chi = L.kappa * rDu;
                drift = __builtin_fma(fabs(d_c), S, fabs(d_u));
            }
            kb = chi * B;
            ++it;
        }
    }
    const bool moved = it > 0;
    return Scales{us, moved ? chi * c.dtheta : 1e-4, moved ? chi * c.dq : 1e-4, it, it};
}

// The same iteration laid out for ONE WAVE PER SIMD (a latitude slab of a strongly scaled run): there a wave
// issues a dependent VALU instruction every ≈ 9 cycles and an independent one every ≈ 4 (scratch/ubench_lat.hip), so what
// counts is how much of a trip is one basic block that tools/gcn_sched.py can re-order for latency after register
// allocation (values local to the block renamed into the registers a 256-VGPR kernel has to spare).  Same arithmetic as
// mo_iterate_lean, expression for expression — results are bitwise the same (tests/test_slab_line.py) —; what differs is
// control flow only: the first trip (general expressions from θ★ = q★ = 1e-4) is peeled off, and the gustiness term is
// unconditional.  Precondition beyond mo_iterate_lean's: β_gust ≠ 0 (every SimilarityTheoryFluxes preset of the reference).
template <bool COARE, bool FIRST>
__device__ __forceinline__ void mo_lean_line_step(const LoopParams& L, const LeanCell& c, const double* tab, const double* logt, double B,
                                                  double S, double log_A_q, double& us, double& ius, double& chi, double& kb,
                                                  double& drift) {
    const double inv_L = kb * (ius * ius);
    const double lu = vmin_u(__builtin_fma(c.alpha_g * us, us, c.lam_nu * ius), L.lm_m);
    const LogHalf half_u = flog_pos_begin(logt, lu);
    const LogHalf half_q = flog_pos_begin(logt, lu * us * c.inv_nu_q);
    const PsiArg ah = psi_arg_x(__builtin_fma(L.x_scale, fabs(inv_L), 1.0), kb < 0.0);
    double U, rU;
    {
        const double w = fmax(-(us * kb) * L.gust_c, 1e-18);
        sqrt_rsqrt_lean(c.dU2 + vmax_u(pow23_lean(w), L.min_gust2), U, rU);
    }
    const double2 ps = psi_eval_pair(tab, ah);
    const double log_lu = flog_lean_end(half_u);
    const double log_lq = vmin_u(__builtin_fma(-L.b_q, flog_lean_end(half_q), log_A_q), L.log_lm_q);
    double Du = (L.log_h - log_lu) - ps.x;
    double Dq = (L.log_h - log_lq) - ps.y;
    if constexpr (!COARE) {
        const double zu = lu * inv_L, zq = fexp_lean(tab, log_lq) * inv_L;
        double2 pl;
        if (fabs(zu) < SMALL_Z0 && fabs(zq) < SMALL_Z0) {
            asm volatile("" ::: "memory");
            pl = psi_small_mh(tab, inv_L < 0.0, zu, zq);
        } else {
            asm volatile("" ::: "memory");
            pl = psi_eval_two(tab, psi_arg(zu), psi_arg(zq));
        }
        Du += pl.x;
        Dq += pl.y;
    }
    Du = vmax_u(Du, L.profile_floor);
    Dq = vmax_u(Dq, L.profile_floor);
    const double r = frcp1(Du * Dq);
    const double kU = L.kappa * U, rDq = r * Dq, rDu = r * Du;
    ius = (Du * rU) * L.two_inv_kappa;
    const double d_u = __builtin_fma(kU, rDq, -us);
    double kU_after = kU;
    asm("" : "+v"(kU_after) : "v"(d_u));
    us = kU_after * rDq;
    if constexpr (FIRST) {
        chi = L.kappa * rDu;
        drift = fabs(d_u) + fabs(chi * c.dtheta - 1e-4) + fabs(chi * c.dq - 1e-4);
    } else {
        const double d_c = __builtin_fma(L.kappa, rDu, -chi);
        chi = L.kappa * rDu;
        drift = __builtin_fma(fabs(d_c), S, fabs(d_u));
    }
    kb = chi * B;
}

template <bool COARE>
__device__ __forceinline__ Scales mo_iterate_lean_line(const LoopParams& L, const LeanCell& c, const double* tab, bool active) {
    const double* logt = tab + LOG_OFFSET;
    const double B = __builtin_fma(c.dtheta, c.bth, c.bqq * c.dq), S = fabs(c.dtheta) + fabs(c.dq);
    double us = 1e-4, ius = 1e4, chi = 0.0, kb = __builtin_fma(1e-4, c.bth, c.bqq * 1e-4);
    double drift = active ? __builtin_inf() : -__builtin_inf();  // (see mo_iterate_lean)
    int it = 0;
    double log_A_q = L.log_A_q;
    asm("" : "+v"(log_A_q));
    if (L.maxiter > 0 && __builtin_amdgcn_ballot_w64(active) != 0ull) {
        if (active) {
            mo_lean_line_step<COARE, true>(L, c, tab, logt, B, S, log_A_q, us, ius, chi, kb, drift);
            it = 1;
        }
        for (int trip = 1;; ++trip) {
            const bool go = !(drift < L.tol);
            if (trip >= L.maxiter || __builtin_amdgcn_ballot_w64(go) == 0ull) break;
            if (go) {
                mo_lean_line_step<COARE, false>(L, c, tab, logt, B, S, log_A_q, us, ius, chi, kb, drift);
                ++it;
            }
        }
    }
    const bool moved = it > 0;
    return Scales{us, moved ? chi * c.dtheta : 1e-4, moved ? chi * c.dq : 1e-4, it, it};
}

// ---------------------------------------------------------------------------------------------
// Atmosphere–sea-ice interface on the lean primitives (SOLVER_SEAICE with constant roughness lengths and U_G,min > 0:
// corrected_/ncar_atmosphere_sea_ice_fluxes, omip_simulation.jl:62-69,105-113).  The same iteration as ice_iterate
// (coflux_fast.hpp) — skin temperature from the energy balance with the previous scales, limited to ±ΔTmax and capped
// at the melting point, saturation over ice and the surface air's phase-equilibrium state at the new skin temperature,
// then the similarity step — with everything that does not depend on the skin temperature hoisted (the three
// log(h/ℓ), the roughness-length ψ arguments' factors, 1/p, 1/(ρ R_v)), one reciprocal for the three profile
// denominators, 1/u★ from the wind scale's reciprocal square root, table exponentials, one Newton step per reciprocal.
// Round 2's body re-derived the constant roughness lengths' exponentials and viscosities every iteration and spent
// ≈ 510 instructions per iteration; this one ≈ 230.
// ---------------------------------------------------------------------------------------------
struct LeanIceConsts {
    double rho, cp, qav, Ls, Ti, hk, Qd, theta_a, pa, inv_pa, inv_rho_Rv, dU2;
};

template <bool COARE>
__device__ __forceinline__ Scales ice_iterate_lean(const DevParams& P, const LoopParams& L, const IceParams& I, const LeanIceConsts& c,
                                                   const double* tab, bool active, double& Ts) {
    const double* logt = tab + LOG_OFFSET;
    double us = 1e-4, ius = 1e4, ts = 1e-4, qq = 1e-4, drift = __builtin_inf();
    int it = 0, work = 0;
    // the state two iterations ago: an exact period-2 orbit ends the iteration early (see ice_iterate)
    // (1/u★ is part of the state here: the orbit is exact only if it repeats too)
    double us_2 = -1.0, ius_2 = 0.0, ts_2 = 0.0, qq_2 = 0.0, Ts_2 = 0.0;
    // iteration-invariant: log(h/ℓ) of the three constant roughness lengths, the lengths themselves
    const double lgu = L.log_h - L.log_const_m, lgq = L.log_h - L.log_const_q, lgt = L.log_h - L.log_const_t;
    const double lu = L.const_m, lq = fexp(L.log_const_q), lt = fexp(L.log_const_t);
    const bool same_scalar = L.log_const_q == L.log_const_t;
    const double dcp_i = P.cp_v - P.cp_i;
    const double a_ice = dcp_i * P.inv_R_v, b_ice = (P.LH_s0 - dcp_i * P.T_0) * P.inv_R_v;
    for (;;) {
        const bool go = active && it < L.maxiter && !(drift < L.tol);
        if (__ballot(go) == 0ull) break;
        if (go) {
            const double us_1 = us, ius_1 = ius, ts_1 = ts, qq_1 = qq, Ts_1 = Ts;  // state(it)
            // skin temperature from the energy balance with the previous scales
            const double T2 = Ts * Ts;
            const double rho_u = c.rho * us;
            double Tstar;
            if (I.semi_implicit != 0.0) {
                const double Qrest = -rho_u * c.Ls * qq - rho_u * c.cp * ts + c.Qd;
                Tstar = __builtin_fma(-Qrest, c.hk, c.Ti) * frcp1(__builtin_fma(c.hk * I.eps_sigma, T2 * Ts, 1.0));
            } else {
                const double Qnet = -rho_u * c.Ls * qq + I.eps_sigma * T2 * T2 - rho_u * c.cp * ts + c.Qd;
                Tstar = __builtin_fma(-Qnet, c.hk, c.Ti);
            }
            Tstar = (Tstar != Tstar) ? Ts : Tstar;
            const double dT = fmin(fmax(Tstar - Ts, -I.dT_max), I.dT_max);
            Ts = fmin(Ts + dT, I.T_melt);
            const double inv_Ts = frcp1(Ts);
            // saturation over ice and the phase-equilibrium saturation pressure at Ts: one logarithm, two table exponentials
            const double Ll = flog_lean_end(flog_pos_begin(logt, Ts * P.inv_T_triple)), Dd = P.inv_T_triple - inv_Ts;
            const double qs = (P.p_triple * fexp_tab(tab, __builtin_fma(a_ice, Ll, b_ice * Dd))) * (c.inv_rho_Rv * inv_Ts);
            const double dq = c.qav - qs, dtheta = c.theta_a - Ts;
            const double lam_s = liquid_fraction_fast(P, logt, Ts);
            double pvs_s;
            {
                const double LH_0 = lam_s * P.LH_v0 + (1.0 - lam_s) * P.LH_s0;
                const double dcp = lam_s * (P.cp_v - P.cp_l) + (1.0 - lam_s) * dcp_i;
                pvs_s = P.p_triple * fexp_tab(tab, __builtin_fma(dcp * P.inv_R_v, Ll, ((LH_0 - dcp * P.T_0) * P.inv_R_v) * Dd));
            }
            // PhaseEquil_pTq(pa, Ts, qs): vapour and virtual temperature of the surface air (air_state_fast's relations)
            const double tiny = 2.220446049250313e-16;
            const double q = fmin(fmax(qs, 0.0), 1.0);
            const double dp = c.pa - pvs_s;
            const double q_vs_p = (dp >= tiny) ? P.Rd_over_Rv * (1.0 - q) * pvs_s * frcp1(dp) : 1.0 / tiny;
            const double q_c0 = fmax(q - q_vs_p, 0.0);
            const double inv_rho_s = P.R_d * (1.0 + P.delta * q - P.eps * q_c0) * Ts * c.inv_pa;
            const double q_c = fmax(q - pvs_s * inv_rho_s * P.inv_R_v * inv_Ts, 0.0);
            const double q_vap = fmax(0.0, q - lam_s * q_c - (1.0 - lam_s) * q_c);
            const double Tv = (1.0 + P.delta * q - P.eps * q_c) * Ts;
            const double kg = (L.kappa * P.g) * frcp1(Tv);
            const double kb = kg * __builtin_fma(ts, 1.0 + P.delta * q_vap, (P.delta * Tv) * qq);  // κ b★
            const double inv_L = kb * (ius * ius);
            // wind speed scale with gustiness
            double U, rU;  // rU = 1/(2U)
            if (L.beta_gust != 0.0) {
                const double w = fmax(-(us * kb) * L.gust_c, 1e-18);
                sqrt_rsqrt_lean(c.dU2 + vmax_u(pow23_lean(w), L.min_gust2), U, rU);
            } else {
                sqrt_rsqrt_lean(c.dU2 + L.min_gust2, U, rU);
            }
            const double2 ps = psi_eval_pair(tab, psi_arg_x(__builtin_fma(L.x_scale, fabs(inv_L), 1.0), kb < 0.0));
            double Du = lgu - ps.x, Dq = lgq - ps.y, Dt = lgt - ps.y;
            if constexpr (!COARE) {
                const double zu = lu * inv_L, zq = lq * inv_L, zt = lt * inv_L;
                if (fabs(zu) < SMALL_Z0 && fabs(zq) < SMALL_Z0 && fabs(zt) < SMALL_Z0) {
                    asm volatile("" ::: "memory");
                    const double2 pl = psi_small_mh(tab, inv_L < 0.0, zu, zq);
                    Du += pl.x;
                    Dq += pl.y;
                    Dt += same_scalar ? pl.y : psi_small_mh(tab, inv_L < 0.0, zu, zt).y;
                } else {
                    asm volatile("" ::: "memory");
                    Du += psi_eval(tab, 0, psi_arg(zu));
                    const double pq = psi_eval(tab, 1, psi_arg(zq));
                    Dq += pq;
                    Dt += same_scalar ? pq : psi_eval(tab, 1, psi_arg(zt));
                }
            }
            Du = vmax_u(Du, L.profile_floor);
            Dq = vmax_u(Dq, L.profile_floor);
            Dt = vmax_u(Dt, L.profile_floor);
            double iDu, iDq, iDt;
            if (same_scalar) {  // (wave-uniform) D_t = D_q: one reciprocal for both profiles
                const double r = frcp1(Du * Dq);
                iDu = r * Dq;
                iDq = iDt = r * Du;
            } else {
                const double DqDt = Dq * Dt;
                const double r = frcp1(Du * DqDt);
                iDu = r * DqDt;
                iDq = r * (Du * Dt);
                iDt = r * (Du * Dq);
            }
            const double un = (L.kappa * U) * iDu, tn = (L.kappa * iDt) * dtheta, qn = (L.kappa * iDq) * dq;
            ius = (Du * rU) * L.two_inv_kappa;
            drift = fabs(un - us) + fabs(tn - ts) + fabs(qn - qq);
            us = un;
            ts = tn;
            qq = qn;
            ++it;
            work = it;
            if (I.orbit_shortcut != 0.0 && us == us_2 && ius == ius_2 && ts == ts_2 && qq == qq_2 && Ts == Ts_2 && !(drift < L.tol)) {
                // exact period 2: jump to the last iteration (an odd number of steps left lands on the other state of the orbit)
                if ((L.maxiter - it) & 1) {
                    us = us_1;
                    ius = ius_1;
                    ts = ts_1;
                    qq = qq_1;
                    Ts = Ts_1;
                }
                it = L.maxiter;
            }
            us_2 = us_1;
            ius_2 = ius_1;
            ts_2 = ts_1;
            qq_2 = qq_1;
            Ts_2 = Ts_1;
        }
    }
    return Scales{us, ts, qq, it, work};
}

__device__ __forceinline__ CellFluxes lean_epilogue(const LeanCell& c, double T_offset, const Scales& s) {
    CellFluxes R;
    const double mu2 = -(s.us * s.us);
    R.Fv = -(c.rho * s.us) * s.qq;
    R.Qv = R.Fv * c.Lv;
    R.Qc = -(c.rcp * s.us) * s.ts;
    R.rho_tau_x = mu2 * c.rdu;
    R.rho_tau_y = mu2 * c.rdv;
    R.Ts_ocean = c.Ts - T_offset;
    R.ustar = s.us;
    R.tstar = s.ts;
    R.qstar = s.qq;
    R.iterations = s.it;
    return R;
}

}  // namespace coflux
