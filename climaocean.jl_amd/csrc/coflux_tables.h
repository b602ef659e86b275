// Shapes of the LDS-resident solver tables (built on the host by coflux_tables.cpp).
#pragma once
#include <vector>

namespace coflux {

// ψ is tabulated against x = 1 + PSI_A·|ζ| ∈ [1, 2^PSI_BINADES): every binade of x is cut into PSI_SUB equal
// pieces, so the segment index is just the exponent and the top mantissa bits of x — no logarithm.
constexpr int PSI_BINADES = 36;  // |ζ| ≤ 2^36 / 16 = 4.3e9
constexpr int PSI_SUB = 4;       // linear sub-segments per binade (2 mantissa bits)
constexpr int PSI_SEG = PSI_BINADES * PSI_SUB;
constexpr int PSI_DEG = 9;       // polynomial degree per segment, in u = x − (segment start)       // polynomial degree per segment, in u = x − (segment start)       // polynomial degree per segment, in u = x − (segment start)
constexpr double PSI_A = 16;
constexpr int PSI_TABLE = PSI_SEG * (PSI_DEG + 1);  // doubles per (function, sign) table
constexpr int LOG_SEG = 128;                        // mantissa intervals of the log table
constexpr int TABLE_DOUBLES = 4 * PSI_TABLE + 2 * LOG_SEG;

std::vector<double> build_solver_tables(int stability_kind);

}  // namespace coflux
