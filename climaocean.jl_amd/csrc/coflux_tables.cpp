// coflux_tables.cpp — host-side construction of the lookup tables the gfx950 solver stages in LDS.
//
// The Monin–Obukhov iteration spends its time in ψ_m(ζ), ψ_h(ζ) and log().  On CDNA4 an FP64
// libm call costs 40–120 instructions (double-double arithmetic), and the stability functions
// need 3 log + 2 atan + cbrt + 2 sqrt each.  Instead, every ψ is tabulated once per context as
// piecewise degree-6 polynomials in  x = 1 + 16|ζ|  (two tiers, coflux_tables.h: eight pieces per binade
// below x = 2^14, one per binade up to 2^34; one table per sign of ζ), which reproduces the analytic functions
// to ≤ 3.3e-12 relative to max(|ψ|, 1) wherever the iteration can stop.  The segment index is the exponent and the
// three top mantissa bits of x and the polynomial variable is u = x − (segment start) (an exact subtraction),
// so an evaluation costs ≈ 10 integer/FP64 instructions + 7 LDS reads + 6 FMAs per function and no logarithm.  Both signs share one instruction stream, so waves that mix stable and unstable cells no
// longer execute both branches.
//
// log() itself uses a 128-entry (1/c, log c) table on the mantissa.
#include <cmath>
#include <utility>
#include <vector>

#include "../../include/coflux.h"
#include "coflux_tables.h"

namespace coflux {

namespace {

const long double PI_L = 3.14159265358979323846264338327950288L;

// --- analytic stability functions (same formulas as coflux_device.hpp's reference branch) -----
long double paulson_m(long double x) {
    return 2.0L * logl((1.0L + x) / 2.0L) + logl((1.0L + x * x) / 2.0L) - 2.0L * atanl(x) + PI_L / 2.0L;
}
long double convective(long double y) {
    const long double r3 = sqrtl(3.0L);
    return 1.5L * logl((1.0L + y + y * y) / 3.0L) - r3 * atanl((1.0L + 2.0L * y) / r3) + PI_L / r3;
}

// `unstable` selects the branch explicitly (ζ = 0 belongs to the stable branch in the reference:
// ifelse(ζ < 0, ψᵤ, ψₛ)); `az` = |ζ|.
long double psi_m_exact(int kind, bool unstable, long double az) {
    if (unstable) {
        long double zm = -az;
        if (kind == CF_STABILITY_EDSON2013) {
            long double p1 = paulson_m(sqrtl(sqrtl(1.0L - 15.0L * zm)));
            long double p2 = convective(cbrtl(1.0L - 10.15L * zm));
            long double f = zm * zm / (1.0L + zm * zm);
            return (1.0L - f) * p1 + f * p2;
        }
        return paulson_m(sqrtl(sqrtl(1.0L - 16.0L * zm)));
    }
    long double zp = az;
    if (kind == CF_STABILITY_EDSON2013) {
        long double dz = fminl(50.0L, 0.35L * zp);
        return -0.7L * zp - 0.75L * (zp - 5.0L / 0.35L) * expl(-dz) - 0.75L * 5.0L / 0.35L;
    }
    if (kind == CF_STABILITY_SHEBA) {
        const long double a = 5.0L, b = 5.0L / 6.5L, r3 = sqrtl(3.0L);
        long double B = cbrtl((1.0L - b) / b), x = cbrtl(1.0L + zp);
        return -3.0L * a / b * (x - 1.0L) +
               a * B / (2.0L * b) *
                   (2.0L * logl((x + B) / (1.0L + B)) - logl((x * x - x * B + B * B) / (1.0L - B + B * B)) +
                    2.0L * r3 * (atanl((2.0L * x - B) / (r3 * B)) - atanl((2.0L - B) / (r3 * B))));
    }
    return -5.0L * zp;
}

long double psi_h_exact(int kind, bool unstable, long double az) {
    if (unstable) {
        long double zm = -az;
        if (kind == CF_STABILITY_EDSON2013) {
            long double p1 = 2.0L * logl((1.0L + sqrtl(1.0L - 15.0L * zm)) / 2.0L);
            long double p2 = convective(cbrtl(1.0L - 34.15L * zm));
            long double f = zm * zm / (1.0L + zm * zm);
            return (1.0L - f) * p1 + f * p2;
        }
        return 2.0L * logl((1.0L + sqrtl(1.0L - 16.0L * zm)) / 2.0L);
    }
    long double zp = az;
    if (kind == CF_STABILITY_EDSON2013) {
        long double dz = fminl(50.0L, 0.35L * zp);
        long double base = 1.0L + 2.0L / 3.0L * zp;
        return -(base * sqrtl(base)) - 2.0L / 3.0L * (zp - 14.28L) * expl(-dz) - 8.525L;
    }
    if (kind == CF_STABILITY_SHEBA) {
        const long double a = 5.0L, b = 5.0L, c = 3.0L;
        long double B = sqrtl(c * c - 4.0L);
        return -b / 2.0L * logl(1.0L + c * zp + zp * zp) +
               (-a / B + b * c / (2.0L * B)) *
                   (logl((2.0L * zp + c - B) / (2.0L * zp + c + B)) - logl((c - B) / (c + B)));
    }
    return -5.0L * zp;
}

// Degree-(PSI_DEG) Chebyshev interpolant of f on [-1, 1], returned as monomial coefficients.
template <class F>
void cheb_fit_monomial_ld(F f, long double* out /* PSI_DEG+1 */) {
    constexpr int N = PSI_DEG + 1;
    long double fx[N], c[N];
    for (int k = 0; k < N; ++k) fx[k] = f(cosl(PI_L * (k + 0.5L) / N));
    for (int j = 0; j < N; ++j) {
        long double s = 0;
        for (int k = 0; k < N; ++k) s += fx[k] * cosl(PI_L * j * (k + 0.5L) / N);
        c[j] = (j == 0 ? 1.0L : 2.0L) * s / N;
    }
    // Σ c_j T_j(t) → Σ m_i t^i via the T recurrence on coefficient vectors
    long double Tm2[N] = {1}, Tm1[N] = {0, 1}, T[N], m[N] = {0};
    for (int i = 0; i < N; ++i) m[i] = 0;
    m[0] += c[0];
    if (N > 1) m[1] += c[1];
    for (int j = 2; j < N; ++j) {
        for (int i = 0; i < N; ++i) T[i] = -Tm2[i] + (i > 0 ? 2.0L * Tm1[i - 1] : 0.0L);
        for (int i = 0; i < N; ++i) {
            m[i] += c[j] * T[i];
            Tm2[i] = Tm1[i];
            Tm1[i] = T[i];
        }
    }
    for (int i = 0; i < N; ++i) out[i] = m[i];
}

// The same interpolant re-expanded in u ∈ [0, 2/alpha):  t = alpha·u − 1.
template <class F>
void cheb_fit_shifted(F f, long double alpha, double* out /* PSI_DEG+1 */) {
    constexpr int N = PSI_DEG + 1;
    long double m[N];
    cheb_fit_monomial_ld(f, m);
    // Σ m_i (αu − 1)^i = Σ_j u^j α^j Σ_{i≥j} m_i C(i,j) (−1)^{i−j}
    long double binom[N][N] = {};
    for (int i = 0; i < N; ++i) {
        binom[i][0] = 1;
        for (int j = 1; j <= i; ++j) binom[i][j] = binom[i - 1][j - 1] + (j <= i - 1 ? binom[i - 1][j] : 0);
    }
    long double apow = 1;
    for (int j = 0; j < N; ++j) {
        long double s = 0;
        for (int i = j; i < N; ++i) s += m[i] * binom[i][j] * (((i - j) & 1) ? -1.0L : 1.0L);
        out[j] = (double)(s * apow);
        apow *= alpha;
    }
}

}  // namespace

// Layout (doubles): side σ ∈ {0: ζ < 0, 1: ζ ≥ 0}, coefficient-major, ψm and ψh interleaved so that one
// 16-byte LDS read returns the coefficient of both functions:
//   psi[((σ * (PSI_DEG+1) + c) * PSI_SEG + k) * 2 + fn],   fn = 0: ψm, 1: ψh
// then the log table  logt[2*k] = 1/c_k, logt[2*k+1] = log c_k.
std::vector<double> build_solver_tables(int stability_kind) {
    std::vector<double> t(TABLE_DOUBLES, 0.0);
    for (int tau = 0; tau < 4; ++tau) {
        const bool scalar = tau >= 2, unstable = (tau % 2) == 0;
        for (int k = 0; k < PSI_SEG; ++k) {
            double coef[PSI_DEG + 1];
            long double width, x0;  // Δ of the segment, its start
            psi_segment(k, &x0, &width);
            auto f = [&](long double tt) {
                long double az = (x0 - 1.0L + 0.5L * (tt + 1.0L) * width) / (long double)PSI_A;
                return scalar ? psi_h_exact(stability_kind, unstable, az) : psi_m_exact(stability_kind, unstable, az);
            };
            cheb_fit_shifted(f, 2.0L / width, coef);
            const int side = unstable ? 0 : 1, fn = scalar ? 1 : 0;
            for (int c = 0; c <= PSI_DEG; ++c)
                t[(((size_t)side * (PSI_DEG + 1) + c) * PSI_SEG + k) * 2 + fn] = coef[c];
        }
    }
    // small-argument polynomials: ψ(±a) = Σ c_j a^j on a ∈ [0, SMALL_Z0], interpolation at Chebyshev nodes in long double
    {
        constexpr int N = SMALL_DEG + 1;
        for (int side = 0; side < 2; ++side)
            for (int fn = 0; fn < 2; ++fn) {
                long double node[N], A[N][N + 1];
                for (int k = 0; k < N; ++k) {
                    node[k] = 0.5L * (cosl(PI_L * (k + 0.5L) / N) + 1.0L);  // a / Z0 ∈ (0, 1)
                    const long double a = node[k] * (long double)SMALL_Z0;
                    long double p = 1.0L;
                    for (int c = 0; c < N; ++c, p *= node[k]) A[k][c] = p;  // Vandermonde row in s = a / Z0
                    A[k][N] = fn ? psi_h_exact(stability_kind, side == 0, a) : psi_m_exact(stability_kind, side == 0, a);
                }
                for (int i = 0; i < N; ++i) {  // Gauss–Jordan with partial pivoting
                    int piv = i;
                    for (int r = i + 1; r < N; ++r)
                        if (fabsl(A[r][i]) > fabsl(A[piv][i])) piv = r;
                    for (int c = 0; c <= N; ++c) std::swap(A[i][c], A[piv][c]);
                    const long double d = A[i][i];
                    for (int c = 0; c <= N; ++c) A[i][c] /= d;
                    for (int r = 0; r < N; ++r)
                        if (r != i) {
                            const long double m = A[r][i];
                            for (int c = 0; c <= N; ++c) A[r][c] -= m * A[i][c];
                        }
                }
                long double zpow = 1.0L;
                for (int c = 0; c < N; ++c, zpow *= (long double)SMALL_Z0)
                    t[SMALL_OFFSET + (side * N + c) * 2 + fn] = (double)(A[c][N] / zpow);
            }
    }
    for (int k = 0; k < EXP_SEG; ++k) t[EXP_OFFSET + k] = (double)exp2l((long double)k / EXP_SEG);
    double* lt = t.data() + LOG_OFFSET;
    for (int k = 0; k < LOG_SEG; ++k) {
        long double c = 1.0L + (k + 0.5L) / LOG_SEG;  // centre of the k-th mantissa interval of [1, 2)
        double inv_c = (double)(1.0L / c);
        lt[2 * k] = inv_c;
        lt[2 * k + 1] = (double)(-logl((long double)inv_c));  // log of the value actually multiplied by
    }
    return t;
}

}  // namespace coflux
