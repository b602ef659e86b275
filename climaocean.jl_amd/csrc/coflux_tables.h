// Shapes of the LDS-resident solver tables (built on the host by coflux_tables.cpp).
#pragma once
#include <vector>

namespace coflux {

constexpr int PSI_SEG = 128;     // segments on w ∈ [0, PSI_WMAX]
constexpr int PSI_DEG = 7;       // polynomial degree per segment
constexpr double PSI_WMAX = 24;  // w = log(1 + PSI_A |ζ|)  ⇒ |ζ| ≤ 1.65e9
constexpr double PSI_A = 16;
constexpr int PSI_TABLE = PSI_SEG * (PSI_DEG + 1);  // doubles per (function, sign) table
constexpr int LOG_SEG = 128;                        // mantissa intervals of the log table
constexpr int TABLE_DOUBLES = 4 * PSI_TABLE + 2 * LOG_SEG;

std::vector<double> build_solver_tables(int stability_kind);

}  // namespace coflux
