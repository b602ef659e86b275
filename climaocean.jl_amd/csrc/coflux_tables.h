// Shapes of the LDS-resident solver tables (built on the host by coflux_tables.cpp).
#pragma once
#include <vector>

namespace coflux {

// ψ is tabulated against x = 1 + PSI_A·|ζ| ∈ [1, 2^PSI_BINADES) as piecewise polynomials whose segment index is read
// off the floating-point representation of x — no logarithm.  Two tiers:
//   fine   x < 2^PSI_FINE_BINADES (|ζ| < 1024: every state the iteration can stop on): PSI_SUB equal pieces per binade
//          (exponent + 3 mantissa bits), ≤ 3.3e-12 of max(|ψ|, 1);
//   coarse the rest (|ζ| up to 1e9: the first one or two iterates from the 1e-4 guess — the first has |ζ| ≈ 2.4e5 —
//          and extremely stable sea-ice cells): four pieces per binade, ≤ 1.2e-10.  Not coarser: under
//          FixedIterations(n) any iterate can be the stopped one, so no tier may rely on later contraction.
// Round 2 used degree 9 on 4 pieces per binade everywhere (3e-14; 43.5 KB, ten 16-byte LDS reads per ψ pair); the
// solver spends its time issuing FP64 instructions, so the tables buy instructions, not digits: degree 6 = twelve
// FMAs and seven reads per ψ pair, 43 KB.
constexpr int PSI_BINADES = 34;       // |ζ| ≤ 2^34 / 16 = 1.07e9
constexpr int PSI_FINE_BINADES = 14;  // x < 16384
constexpr int PSI_SUB_BITS = 3;
constexpr int PSI_SUB = 1 << PSI_SUB_BITS;  // pieces per fine binade
constexpr int PSI_FINE_SEG = PSI_FINE_BINADES * PSI_SUB;
#ifndef CF_PSI_COARSE_SUB_BITS
#define CF_PSI_COARSE_SUB_BITS 2  // (experiments: 0 = one piece per coarse binade, 39 → 32 KB of tables)
#endif
constexpr int PSI_COARSE_SUB_BITS = CF_PSI_COARSE_SUB_BITS;
constexpr int PSI_COARSE_SUB = 1 << PSI_COARSE_SUB_BITS;  // pieces per coarse binade
constexpr int PSI_SEG = PSI_FINE_SEG + (PSI_BINADES - PSI_FINE_BINADES) * PSI_COARSE_SUB;
constexpr int PSI_DEG = 6;       // polynomial degree per segment, in u = x − (segment start)
constexpr double PSI_A = 16;
constexpr int PSI_TABLE = PSI_SEG * (PSI_DEG + 1);  // doubles per (function, sign) table
constexpr int LOG_SEG = 128;                        // mantissa intervals of the log table
// ψ_m(ℓᵤ/L), ψ_h(ℓ_q/L): the roughness-length arguments are tiny once the iteration has left its first two or
// three iterates (|ζ| < 1e-3 in 99.9 % of converged cells), so below |ζ| < SMALL_Z0 both functions are plain
// degree-SMALL_DEG polynomials in |ζ| (one coefficient set per sign of ζ, ψ_m and ψ_h interleaved), ≤ 1e-12 absolute.
constexpr int SMALL_DEG = 3;
constexpr double SMALL_Z0 = 1.0 / 1024.0;
constexpr int SMALL_DOUBLES = 2 * (SMALL_DEG + 1) * 2;  // [side][coefficient]{ψ_m, ψ_h}
// exp(x) for the scalar roughness length ℓ_q = exp(log ℓ_q): 2^(k/EXP_SEG) table + a short polynomial
constexpr int EXP_SEG = 32;
constexpr int TABLE_PAYLOAD = 4 * PSI_TABLE + 2 * LOG_SEG + SMALL_DOUBLES + EXP_SEG;
constexpr int TABLE_DOUBLES = (TABLE_PAYLOAD + 127) / 128 * 128;  // whole 1 KB pieces for the LDS-DMA stage
constexpr int LOG_OFFSET = 4 * PSI_TABLE;
constexpr int SMALL_OFFSET = LOG_OFFSET + 2 * LOG_SEG;
constexpr int EXP_OFFSET = SMALL_OFFSET + SMALL_DOUBLES;

// segment k → (start of the segment, its width) in x
inline void psi_segment(int k, long double* x0, long double* width) {
    long double one = 1.0L;
    if (k < PSI_FINE_SEG) {
        int b = k / PSI_SUB;
        long double p = one;
        for (int n = 0; n < b; ++n) p *= 2.0L;
        *width = p / PSI_SUB;
        *x0 = p + (k % PSI_SUB) * *width;
    } else {
        int b = PSI_FINE_BINADES + (k - PSI_FINE_SEG) / PSI_COARSE_SUB;
        long double p = one;
        for (int n = 0; n < b; ++n) p *= 2.0L;
        *width = p / PSI_COARSE_SUB;
        *x0 = p + ((k - PSI_FINE_SEG) % PSI_COARSE_SUB) * *width;
    }
}

std::vector<double> build_solver_tables(int stability_kind);

}  // namespace coflux
