// Shapes of the LDS-resident solver tables (built on the host by coflux_tables.cpp).
#pragma once
#include <vector>

namespace coflux {

// ψ is tabulated against x = 1 + PSI_A·|ζ| ∈ [1, 2^PSI_BINADES): every binade of x is cut into PSI_SUB equal
// pieces, so the segment index is just the exponent and the top mantissa bits of x — no logarithm.
constexpr int PSI_BINADES = 34;  // |ζ| ≤ 2^34 / 16 = 1.07e9 (the first iterate from the 1e-4 guess has |ζ| ≈ 2e5)
constexpr int PSI_SUB = 4;       // linear sub-segments per binade (2 mantissa bits)
constexpr int PSI_SEG = PSI_BINADES * PSI_SUB;
constexpr int PSI_DEG = 9;       // polynomial degree per segment, in u = x − (segment start)
constexpr double PSI_A = 16;
constexpr int PSI_TABLE = PSI_SEG * (PSI_DEG + 1);  // doubles per (function, sign) table
constexpr int LOG_SEG = 128;                        // mantissa intervals of the log table
// ψ_m(ℓᵤ/L), ψ_h(ℓ_q/L): the roughness-length arguments are tiny once the iteration has left its first two or
// three iterates (|ζ| < 1e-3 in 99.9 % of converged cells), so below |ζ| < SMALL_Z0 both functions are plain
// degree-SMALL_DEG polynomials in |ζ| (one coefficient set per sign of ζ, ψ_m and ψ_h interleaved): half the LDS
// traffic and a third of the instructions of the general table path, ≤ 1e-16 absolute error.
constexpr int SMALL_DEG = 5;
constexpr double SMALL_Z0 = 1.0 / 1024.0;
constexpr int SMALL_DOUBLES = 2 * (SMALL_DEG + 1) * 2;  // [side][coefficient]{ψ_m, ψ_h}
// exp(x) for the scalar roughness length ℓ_q = exp(log ℓ_q): 2^(k/EXP_SEG) table + degree-5 polynomial
constexpr int EXP_SEG = 32;
constexpr int TABLE_PAYLOAD = 4 * PSI_TABLE + 2 * LOG_SEG + SMALL_DOUBLES + EXP_SEG;
constexpr int TABLE_DOUBLES = (TABLE_PAYLOAD + 127) / 128 * 128;  // whole 1 KB pieces for the LDS-DMA stage
constexpr int LOG_OFFSET = 4 * PSI_TABLE;
constexpr int SMALL_OFFSET = LOG_OFFSET + 2 * LOG_SEG;
constexpr int EXP_OFFSET = SMALL_OFFSET + SMALL_DOUBLES;

std::vector<double> build_solver_tables(int stability_kind);

}  // namespace coflux
