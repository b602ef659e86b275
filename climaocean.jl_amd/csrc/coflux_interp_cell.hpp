// coflux_interp_cell.hpp — interpolate_atmosphere_state! for ONE ocean-grid cell, gather form: bilinear in (λ, φ) on the
// JRA55 source grid, linear in time between the two snapshot levels, rain + snow summed, winds rotated into the grid
// frame.  Shared by the stand-alone gather kernel (coflux_interp.hip), the tiled kernel's fallback and the lean ocean
// solver's fused prologue (coflux_solver_lean.hip), so that every path produces the same bits: a source node's two time
// levels are blended first, the four blended corners are interpolated second (the reference interpolates each level
// and blends last; the two orders differ by rounding, ≈ 1e-16 relative).
#pragma once
#include "coflux_kernel_types.hpp"

namespace coflux {

__device__ __forceinline__ int wrap_index(int i, int n) {
    int r = i % n;
    return r < 0 ? r + n : r;
}

__device__ __forceinline__ double blend_levels(float a, float b, double tf) { return (double)b * tf + (double)a * (1.0 - tf); }
__device__ __forceinline__ double bilinear(double w00, double w01, double w10, double w11, double c00, double c01, double c10,
                                           double c11) {
    return w00 * c00 + w01 * c01 + w10 * c10 + w11 * c11;
}

struct InterpCorners {
    unsigned g00, g10, g01, g11;  // offsets of the four corners inside one (level, variable) plane
    double w00, w01, w10, w11;
};

// Oceananigans `interpolator`: i⁻ = trunc(f), i⁺ = i⁻ + sign(f), ξ = mod(f, 1) ∈ [0, 1) — for a negative fractional
// index (a column west of the first source node) that is f − floor(f), not f − trunc(f); periodic in longitude, clamped
// in latitude.
__device__ __forceinline__ InterpCorners interp_corners(int ns_x, int ns_y, double fi, double fj) {
    InterpCorners c;
    const double ti = trunc(fi), tj = trunc(fj);
    const double xi = fi - floor(fi), eta = fj - floor(fj);
    const int i0 = (int)ti, ja = (int)tj;
    const int is0 = wrap_index(i0, ns_x), is1 = wrap_index(i0 + (fi > 0.0 ? 1 : (fi < 0.0 ? -1 : 0)), ns_x);
    const int j0 = min(max(ja, 0), ns_y - 1), j1 = min(max(ja + (fj > 0.0 ? 1 : (fj < 0.0 ? -1 : 0)), 0), ns_y - 1);
    c.g00 = (unsigned)(j0 * ns_x + is0);
    c.g10 = (unsigned)(j0 * ns_x + is1);
    c.g01 = (unsigned)(j1 * ns_x + is0);
    c.g11 = (unsigned)(j1 * ns_x + is1);
    c.w00 = (1.0 - xi) * (1.0 - eta);
    c.w01 = (1.0 - xi) * eta;
    c.w10 = xi * (1.0 - eta);
    c.w11 = xi * eta;
    return c;
}

__device__ __forceinline__ double interp_value(const float* __restrict__ d, unsigned off1, unsigned off2, double tf, const InterpCorners& c) {
    const float* a = d + off1;
    const float* b = d + off2;
    return bilinear(c.w00, c.w01, c.w10, c.w11, blend_levels(a[c.g00], b[c.g00], tf), blend_levels(a[c.g01], b[c.g01], tf),
                    blend_levels(a[c.g10], b[c.g10], tf), blend_levels(a[c.g11], b[c.g11], tf));
}

struct ExchangeCell {
    double u, v, T, p, q, Qs, Ql, Mp;
};

// the eight exchange fields of cell (i, j) (storage index k)
__device__ __forceinline__ ExchangeCell interp_cell(const SourceDesc& S, const WeightDesc& Wt, const GridDesc& G, int i, int j, size_t k) {
    const double fi = Wt.separable ? Wt.fi[i + G.hx] : Wt.fi[k];
    const double fj = Wt.separable ? Wt.fj[j + G.hy] : Wt.fj[k];
    const InterpCorners c = interp_corners(S.ns_x, S.ns_y, fi, fj);
    const unsigned plane = (unsigned)(S.ns_x * S.ns_y);
    const unsigned off1 = (unsigned)S.level1 * plane, off2 = (unsigned)S.level2 * plane;
    ExchangeCell e;
    e.T = interp_value(S.data[CF_JRA55_TAS], off1, off2, S.tf, c);
    e.q = interp_value(S.data[CF_JRA55_HUSS], off1, off2, S.tf, c);
    e.p = interp_value(S.data[CF_JRA55_PSL], off1, off2, S.tf, c);
    double ua = interp_value(S.data[CF_JRA55_UAS], off1, off2, S.tf, c);
    double va = interp_value(S.data[CF_JRA55_VAS], off1, off2, S.tf, c);
    e.Ql = interp_value(S.data[CF_JRA55_RLDS], off1, off2, S.tf, c);
    e.Qs = interp_value(S.data[CF_JRA55_RSDS], off1, off2, S.tf, c);
    e.Mp = interp_value(S.data[CF_JRA55_PRRA], off1, off2, S.tf, c) + interp_value(S.data[CF_JRA55_PRSN], off1, off2, S.tf, c);
    if (Wt.cos_rot != nullptr && Wt.sin_rot != nullptr) {  // intrinsic_vector: geographic (E, N) → grid frame
        const double cs = Wt.cos_rot[k], sn = Wt.sin_rot[k];
        const double ui = ua * cs + va * sn;
        va = -ua * sn + va * cs;
        ua = ui;
    }
    e.u = ua;
    e.v = va;
    return e;
}

__device__ __forceinline__ void store_exchange(const Exchange& E, size_t k, const ExchangeCell& e) {
    E.u[k] = e.u;
    E.v[k] = e.v;
    E.T[k] = e.T;
    E.p[k] = e.p;
    E.q[k] = e.q;
    E.Qs[k] = e.Qs;
    E.Ql[k] = e.Ql;
    E.Mp[k] = e.Mp;
}

}  // namespace coflux
