// coflux_certified.hpp — the certified reduced-iteration solve of the atmosphere–ocean Monin–Obukhov fixed point
// (CF_OPT_SOLVER_PATH = CF_SOLVER_PATH_CERTIFIED; SOLVER_OCEAN_LEAN configurations, convergence stop rule only).
//
// The reference iterates x ← G(x) from (u★, θ★, q★) = 1e-4 until |Δu★| + |Δθ★| + |Δq★| < tol (omip_simulation.jl:42-49;
// oracle/coflux_oracle.c).  Its map contracts fast at the fixed point (spectral radius: median 0.11, max 0.36 on the 1/4°
// surface, scratch/certified_study.py) — most of its ≈ 12 evaluations per cell are the way back from a start six orders
// of magnitude off.  This path evaluates the SAME map, FP64 throughout, but
//   * starts from the cell's own neutral profile with an effective stability correction in ζ at that state (below),
//   * works on the two-number state (u★, χ) of mo_iterate_lean (θ★ = χ Δθ, q★ = χ Δq: one scalar roughness length),
//   * takes two plain steps and then Anderson(2) steps — in two dimensions the exact multi-secant (Broyden-type) update,
//     superlinear: 4–6 evaluations per cell instead of 8–20 — and accepts the extrapolated state itself once the relative
//     residual of the last evaluation is below `cert_accept` (1e-7: the accepted state is then within ≈ 2e-8, in the flux
//     metric, of the fixed point).  The Anderson bookkeeping (last step, last residual, the older difference pair) is
//     carried in FP32: eight registers instead of sixteen; an error there perturbs the secant model, i.e. the speed of
//     convergence, never the map or the residual test, which stay in FP64;
//   * CERTIFIES, per lane, that the fixed point is where the reference's iteration would have stopped, to within a stated
//     budget in the metric |Δ flux| ≤ budget · max(|flux|, floor) (floors: 1 W m⁻² for Q_c and Q_v, 1e-6 kg m⁻² s⁻¹ for
//     F_v, 1e-3 N m⁻² for ρτ).  The reference stops at the first iterate whose drift d = x_n − x_{n−1} has
//     |d_u| + (|Δθ| + |Δq|)|d_χ| < tol; in the linear regime x_n − x★ = J (J − I)⁻¹ d with J the map's Jacobian at the
//     fixed point, which the last two secant pairs give for free.  The bound is the maximum of the flux error over all
//     drifts the stop rule admits, times a safety factor for the secant Jacobian's accuracy;
//   * a seventh term holds the vapour flux to budget · max(|F_v − M_p|, floor) as well (CertNetSalt below): the net
//     salinity flux J_S ∝ F_v − M_p can cancel to nothing, and an error bounded relative to |F_v| alone then shows in it
//     amplified (measured: 1.8e-6 of J_S's own scale without the term, 5.4e-7 with it; it doubles the uncertified share);
//   * lanes whose bound exceeds the budget (≈ 2 % at 8e-7: near-neutral cells, where the absolute drift test leaves χ
//     loosely determined, dead-calm cells, cells whose evaporation nearly cancels their precipitation), lanes without a usable secant model and lanes that have not converged in
//     `cert_max_evals` evaluations are NOT certified: the caller sends them down the exact path (mo_iterate_lean, the
//     reference's own iteration) — never a per-wave decision, a cell's result does not depend on its neighbours.
// Per-lane results are a pure function of the cell's inputs (no state carried between calls).
//
// What the certificate is: an empirical estimate, not a guarantee.  J is a secant estimate from two FP32 difference
// pairs; the bound holds in the map's linear regime, which ρ(J) < 0.6 and the determinant test make likely, not certain; the
// safety factor covers the estimate's MEASURED ± 4 %; the accepted extrapolated state is not evaluated again (its residual is
// inferred from the previous iterate's and the scheme's measured superlinear convergence).  A badly conditioned secant pair
// that passes the 1e-4 determinant test could certify a cell outside the budget; none has been observed on the surfaces the
// tests and bench.py check (every field ≤ 6e-7 at the default budget), and nothing in the construction excludes it.  Hence
// opt-in, and hence bench.py's number of record is the exact path's.
#pragma once
#include "coflux_lean.hpp"

namespace coflux {

// iterations reported for a cell the certified path sent down the exact path: CF_CERTIFIED_EXACT_FLAG | reference trips
constexpr int CERT_EXACT_FLAG = 0x100;
// Thresholds of the iteration, compile-time (each would otherwise hold scalar registers across the loop, which the
// kernel does not have: 44 v_readlane per trip were measured with them in LoopParams).  Accept: relative residual below
// 2⁻²³ ≈ 1.2e-7 (a double whose low word is zero: a 32-bit literal operand); at most ten evaluations.
constexpr double CERT_ACCEPT = 0x1p-23;
constexpr int CERT_MAX_EVALS = 10;

// Floor of the net salinity flux J_S in the certificate's seventh term (m s⁻¹ psu; the parity metric's scale of that field,
// tests/util.py::FIELD_SCALE["S"], bench.py::PARITY_SCALE)
constexpr float CERT_JS_FLOOR = 1e-7f;

// What the launch knows of the cell's net salinity flux J_S = −S (F_v − M_p)/ρ_f (compute_net_ocean_fluxes!): evaporation can
// cancel precipitation, and the vapour flux's own budget, relative to |F_v|, then says nothing about J_S.  With the
// precipitation at hand (the launch that assembles the net fluxes has it) the certificate also bounds the vapour flux's error
// by budget · max(|F_v − M_p|, floor_v), floor_v = CERT_JS_FLOOR ρ_f / S.  FP32: it is a threshold, not a result.
struct CertNetSalt {
    float Mp = 0.f;                            // rain + snow, kg m⁻² s⁻¹ (positive down)
    float floor_v = __builtin_inff();          // +inf: the term is off (no net fluxes assembled by this launch)
};

// All 64 lanes of a wave must call this together.  `active` lanes solve; on return `need_exact` is set for the active
// lanes that are not certified (their Scales are meaningless).  Scales.it = evaluations of the map.
template <bool COARE>
__device__ __forceinline__ Scales mo_iterate_certified(const LoopParams& L, const LeanCell& c, const double* tab, bool active, bool& need_exact,
                                                       const CertNetSalt ns = CertNetSalt{}) {
    const double* logt = tab + LOG_OFFSET;
    const double B = __builtin_fma(c.dtheta, c.bth, c.bqq * c.dq);
    // ---- first guess: the neutral profile of THIS cell, then an effective stability correction ----
    // u₁ = c_u U₀ (U₀ = √(Δu² + U_G,min²)) gives the roughness lengths of the neutral column, D_u = log(h/ℓ_u), D_q = log(h/ℓ_q),
    // and ζ₀ = h κ b★ / u★² at that neutral state; the fixed point over the neutral solution is a tight function of ζ₀
    // (± 2 %, scratch/certified_study.py), fitted as effective ψ̃_m, ψ̃_h (a logarithm on the unstable side, a saturating
    // ratio on the stable one — FP32, they only place the start): u★ = κ U₀ / (D_u − ψ̃_m), χ = κ / (D_q − ψ̃_h).
    // Start within 0.6 % (median) / 2 % (90 %) / 7 % (99 %) of the fixed point instead of 4 – 12 %: 4.5 / 4.2 evaluations per cell
    // instead of 5.1, for about a third of one evaluation.  Any start is a valid start: what is accepted and certified is
    // decided on the map itself.  Used by the COARE-profile variant only (solver stage 53.2 → 51.6 µs): the log-profile
    // variant sits at the register cap, the extra code costs it 16 more bytes of scratch and the step 1 % (measured on one
    // box: 0.0775 → 0.0785 ms) — it keeps the plain neutral guess u★ = c_u U₀, χ = κ / log(h / 1e-4 m).
    constexpr bool FITTED_START = COARE;
    double us, ius, chi;
    {
        double U0, rU0;  // rU0 = 1/(2 U0)
        sqrt_rsqrt_lean(c.dU2 + L.min_gust2, U0, rU0);
        const double u1 = L.cert_u0 * U0, iu1 = rU0 * L.cert_two_inv_u0;
        if constexpr (!FITTED_START) {
            us = u1;
            ius = iu1;
            chi = L.cert_chi0;
        } else {
        const double lu = vmin_u(__builtin_fma(c.alpha_g * u1, u1, c.lam_nu * iu1), L.lm_m);
        const LogHalf hu = flog_pos_begin(logt, lu);
        const LogHalf hq = flog_pos_begin(logt, lu * u1 * c.inv_nu_q);
        const double Dun = L.log_h - flog_lean_end(hu);
        const double Dqn = L.log_h - vmin_u(__builtin_fma(-L.b_q, flog_lean_end(hq), L.log_A_q), L.log_lm_q);
        // ζ₀ = h · (κ/D_q) B · (D_u / (κ U₀))² = h B D_u² · 4 rU0² / (κ D_q)
        const double rDq0 = frcp1(Dqn);
        const float z = (float)((L.h_ref * B) * (Dun * Dun) * (4.0 * rU0 * rU0) * (L.inv_kappa * rDq0));
        const float az = fabsf(z);
        float pm, ph;
        if (z < 0.f) {
            pm = (0.777f * 0.6931472f) * __builtin_amdgcn_logf(__builtin_fmaf(3.95f, az, 1.f));   // (v_log_f32 is log2)
            ph = (0.798f * 0.6931472f) * __builtin_amdgcn_logf(__builtin_fmaf(8.90f, az, 1.f));
        } else {
            pm = -6.11f * az * __builtin_amdgcn_rcpf(__builtin_fmaf(0.191f, az, 1.f));
            ph = -4.75f * az;
        }
        const double Du0 = fmax(Dun - (double)pm, 2.0), Dq0 = fmax(Dqn - (double)ph, 2.0);
        const double r0 = frcp1(Du0 * Dq0);
        us = (L.kappa * U0) * (r0 * Dq0);
        ius = (Du0 * rU0) * L.two_inv_kappa;
        chi = L.kappa * (r0 * Du0);
        }
    }
    // Anderson history, FP32: s = the step that led to the current state, fp = the previous residual, (dg2, df2) = the
    // older pair of differences of G and of the residual; ff = the residual of the last evaluation
    float s_u = 0.f, s_c = 0.f, fp_u = 0.f, fp_c = 0.f, dg2_u = 0.f, dg2_c = 0.f, df2_u = 0.f, df2_c = 0.f, ff_u = 0.f, ff_c = 0.f;
    bool done = false, failed = false;
    int it = 0;
    double log_A_q = L.log_A_q;
    asm("" : "+v"(log_A_q));
    for (int trip = 0;; ++trip) {
        const bool go = active && !done;
        if (trip >= CERT_MAX_EVALS || __builtin_amdgcn_ballot_w64(go) == 0ull) break;
        if (go) {
            // ---- one evaluation of the reference's map at (u★, χ): mo_iterate_lean's expressions ----
            const double kb = chi * B;
            const double inv_L = kb * (ius * ius);
            const double lu = vmin_u(__builtin_fma(c.alpha_g * us, us, c.lam_nu * ius), L.lm_m);
            const LogHalf half_u = flog_pos_begin(logt, lu);
            const LogHalf half_q = flog_pos_begin(logt, lu * us * c.inv_nu_q);
            const PsiArg ah = psi_arg_x(__builtin_fma(L.x_scale, fabs(inv_L), 1.0), kb < 0.0);
            double U, rU;
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
                // ψ at the roughness-length arguments.  From the neutral start the iterates stay at |ℓ/L★| < 2⁻¹⁰ in all but
                // ≈ 0.4 % of the cells (the exact path needs the general table for its first two iterates from u★ = 1e-4):
                // the small-argument polynomials for every lane, and — a wave-level branch taken only where a lane needs it —
                // the general table where both arguments share a segment (one chain of 16-byte reads, two accumulators:
                // this loop cannot hold psi_eval_two's fourteen coefficient registers).  A lane whose arguments fall into
                // different segments (|ℓ/L★| ≳ 1e-2) is not certified: the exact path solves it.
                const double zu = lu * inv_L, zq = fexp_lean(tab, log_lq) * inv_L;
                const bool small = fabs(zu) < SMALL_Z0 && fabs(zq) < SMALL_Z0;
                double2 pl = psi_small_mh(tab, inv_L < 0.0, zu, zq);
                if (__builtin_amdgcn_ballot_w64(!small) != 0ull) {
                    const PsiArg a = psi_arg(zu), b = psi_arg(zq);
                    const double2* cf = reinterpret_cast<const double2*>(tab) + (size_t)(a.side * (PSI_DEG + 1)) * PSI_SEG + a.k;
                    double2 v = cf[PSI_DEG * PSI_SEG];
                    double pm = v.x, ph = v.y;
#pragma unroll
                    for (int j = PSI_DEG - 1; j >= 0; --j) {
                        v = cf[j * PSI_SEG];
                        pm = __builtin_fma(pm, a.t, v.x);
                        ph = __builtin_fma(ph, b.t, v.y);
                    }
                    failed |= !small && a.k != b.k;
                    if (!small) pl = make_double2(pm, ph);
                }
                Du += pl.x;
                Dq += pl.y;
            }
            Du = vmax_u(Du, L.profile_floor);
            Dq = vmax_u(Dq, L.profile_floor);
            const double r = frcp1(Du * Dq);
            const double kU = L.kappa * U, rDq = r * Dq, rDu = r * Du;  // 1/D_u, 1/D_q
            const double gu = kU * rDq, gc = L.kappa * rDu;             // G(x)
            const double gius = (Du * rU) * L.two_inv_kappa;            // 1/G_u
            const double fu = __builtin_fma(kU, rDq, -us), fc = __builtin_fma(L.kappa, rDu, -chi);
            // relative residual |f_u|/G_u + |f_χ|/G_χ
            const double res = __builtin_fma(fabs(fu), gius, fabs(fc) * (Dq * L.inv_kappa));
            ff_u = (float)fu;
            ff_c = (float)fc;
            float cu = 0.f, cc = 0.f, df1_u = 0.f, df1_c = 0.f, dg1_u = 0.f, dg1_c = 0.f;
            bool aa = false;
            if (trip >= 1) {  // (wave-uniform: every running lane has the same count)
                df1_u = ff_u - fp_u;
                df1_c = ff_c - fp_c;
                dg1_u = s_u + df1_u;
                dg1_c = s_c + df1_c;
            }
            if (trip >= 2) {
                // Anderson(2): γ solves [df1 df2] γ = f; the next state is G(x) − γ₁ dg1 − γ₂ dg2
                const float p1 = df1_u * df2_c, p2 = df1_c * df2_u;
                const float det = p1 - p2;
                const float rdet = __builtin_amdgcn_rcpf(det);
                const float g1 = (ff_u * df2_c - ff_c * df2_u) * rdet, g2 = (df1_u * ff_c - df1_c * ff_u) * rdet;
                const float tu = g1 * dg1_u + g2 * dg2_u, tc = g1 * dg1_c + g2 * dg2_c;
                // a usable model: the pairs are not parallel (FP32 conditioning), the correction stays inside half of G(x)
                aa = fabsf(det) > 1e-4f * (fabsf(p1) + fabsf(p2)) && fabsf(tu) < 0.5f * (float)gu && fabsf(tc) < 0.5f * (float)gc;
                cu = aa ? tu : 0.f;
                cc = aa ? tc : 0.f;
            }
            us = gu - (double)cu;
            chi = gc - (double)cc;
            ++it;
            if (failed) {
                done = true;
            } else if (aa && res < CERT_ACCEPT) {
                done = true;  // the extrapolated state is the answer; the history stays as it is for the certificate
            } else {
                if (trip >= 1) {
                    dg2_u = dg1_u;
                    dg2_c = dg1_c;
                    df2_u = df1_u;
                    df2_c = df1_c;
                }
                fp_u = ff_u;
                fp_c = ff_c;
                s_u = ff_u - cu;
                s_c = ff_c - cc;
                // 1/u★ of the extrapolated state u★ = G_u (1 − t), t = c_u / G_u: 1/G_u is at hand.  The first Anderson step
                // may be large (one reciprocal); from then on |t| ≲ 1e-2 and three terms of the series leave t⁴ — an
                // inconsistency of the evaluation point that vanishes with the corrections, not an error of the map
                ius = gius;
                if (trip == 2) {
                    ius = aa ? frcp1(us) : gius;
                } else if (trip > 2) {
                    const double t = (double)cu * gius;
                    ius = __builtin_fma(gius * t, __builtin_fma(t, t + 1.0, 1.0), gius);
                    if (fabs(t) > 0x1p-4) ius = frcp1(us);
                }
            }
        }
    }
    // ---- the certificate (FP32): secant Jacobian J [dx1 dx2] = [dg1 dg2] of the last two pairs, M = J (J − I)⁻¹ ----
    bool certified = false;
    {
        const float df1_u = ff_u - fp_u, df1_c = ff_c - fp_c;
        const float dg1_u = s_u + df1_u, dg1_c = s_c + df1_c;
        const float dx2_u = dg2_u - df2_u, dx2_c = dg2_c - df2_c;
        const float q1 = s_u * dx2_c, q2 = s_c * dx2_u;
        const float dX = q1 - q2;
        const bool okJ = fabsf(dX) > 1e-4f * (fabsf(q1) + fabsf(q2));
        const float rX = __builtin_amdgcn_rcpf(dX);
        const float a = (dg1_u * dx2_c - dg2_u * s_c) * rX, b = (dg2_u * s_u - dg1_u * dx2_u) * rX;
        const float cj = (dg1_c * dx2_c - dg2_c * s_c) * rX, d = (dg2_c * s_u - dg1_c * dx2_u) * rX;
        // the linear regime the bound presumes, and a reference that stops on its drift rather than on maxiter: ρ(J) < 0.6
        const float tr = a + d, dj = a * d - b * cj, disc = tr * tr - 4.f * dj;
        const float rho_j = disc >= 0.f ? 0.5f * (fabsf(tr) + __builtin_sqrtf(disc)) : __builtin_sqrtf(dj);
        const float det = (a - 1.f) * (d - 1.f) - b * cj;
        const float rd = __builtin_amdgcn_rcpf(det);
        const float i11 = (d - 1.f) * rd, i12 = -b * rd, i21 = -cj * rd, i22 = (a - 1.f) * rd;
        const float M11 = a * i11 + b * i21, M12 = a * i12 + b * i22, M21 = cj * i11 + d * i21, M22 = cj * i12 + d * i22;
        const float u = (float)us, x = (float)chi, tol = (float)L.tol;
        const float dth = (float)c.dtheta, dq = (float)c.dq, rho = (float)c.rho;
        const float rS = __builtin_amdgcn_rcpf(fmaxf(fabsf(dth) + fabsf(dq), 1e-30f));
        // |e_u| and |χ e_u + u e_χ| maximised over the drifts with |d_u| + S |d_χ| < tol
        const float eu = tol * fmaxf(fabsf(M11), fabsf(M12) * rS);
        const float eq = tol * fmaxf(fabsf(x * M11 + u * M21), fabsf(x * M12 + u * M22) * rS);
        float bound = (2.f * rho * u * eu) * __builtin_amdgcn_rcpf(fmaxf(rho * u * u, 1e-3f));
        const float ux = u * x;
        const float kd_c = fabsf((float)c.rcp * dth), kd_f = fabsf(rho * dq), kd_v = kd_f * (float)c.Lv;
        bound = fmaxf(bound, (kd_c * eq) * __builtin_amdgcn_rcpf(fmaxf(kd_c * ux, 1.f)));
        bound = fmaxf(bound, (kd_v * eq) * __builtin_amdgcn_rcpf(fmaxf(kd_v * ux, 1.f)));
        bound = fmaxf(bound, (kd_f * eq) * __builtin_amdgcn_rcpf(fmaxf(kd_f * ux, 1e-6f)));
        // the net salinity flux: F_v = −ρ u★ χ Δq against the precipitation it may cancel
        bound = fmaxf(bound, (kd_f * eq) * __builtin_amdgcn_rcpf(fmaxf(fabsf(__builtin_fmaf(rho * dq, ux, ns.Mp)), ns.floor_v)));
        certified = done && !failed && okJ && rho_j < 0.6f && bound <= (float)L.cert_budget;  // (NaN anywhere: not certified)
    }
    need_exact = active && !certified;
    return Scales{us, chi * c.dtheta, chi * c.dq, it, it};
}

}  // namespace coflux
