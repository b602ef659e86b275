// coflux_interp.hip — interpolate_atmosphere_state! on gfx950.
//
// JRA55 window (9 variables × 2 time levels, Float32, 640×320) → 8 Float64 exchange fields on
// the ocean grid: bilinear in (λ, φ), linear in time, rain + snow summed, winds rotated to the
// grid-intrinsic frame.
//
// Design.  The gather (4 corners × 18 planes = 72 dwords per cell) is bound by the vector-memory
// address path, not by HBM: 72 scattered dword loads per lane cost ≈ 16 cycles each in the
// texture-address unit.  So every WAVE stages the source footprint of its own 64 × ROWS cell tile
// in LDS (≈ 31 × 4 source nodes per variable at 1/4°, the two time levels already blended, as doubles)
// with ≈ 9 coalesced loads per cell, and then reads each variable's four corners from there.  Tiles belong to waves, not workgroups: there is no __syncthreads() anywhere, so the
// waves of a CU hide each other's load latency.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "coflux_interp_cell.hpp"
#include "coflux_interp_tiles.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

template <int ROWS>
__global__ __launch_bounds__(64 * IT_WAVES) void interpolate_kernel(SourceDesc S, WeightDesc Wt, GridDesc G,
                                                                     Exchange E, int cap) {
    interpolate_tiles<ROWS>(S, Wt, G, E, cap, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Two independent memory-bound pieces of consecutive steps in ONE launch (cf_time_steps with pipelining on a surface
// that does not fill the device): the face stresses of step n (they need every cell's ρτ of step n: a launch of their own
// behind the solver) and the interpolation of step n + 1's atmosphere state into the OTHER set of exchange fields
// (nothing of step n reads it).  On a latitude slab every launch boundary is ≈ 3 µs of a ≈ 30 µs step; the two kernels'
// workgroups are all resident at once either way.  Workgroups [0, n_interp) interpolate, the rest take 256 stress
// cells each; the arithmetic is the two kernels' own (interpolate_tiles, net_face_stress): same bits.
// ---------------------------------------------------------------------------------------------
struct StressArgs {
    const void* mask;
    const double* rtx;
    const double* rty;
    IceIn I;
    double* tau_x;
    double* tau_y;
};

template <int ROWS>
__global__ __launch_bounds__(64 * IT_WAVES) void interpolate_and_stress_kernel(SourceDesc S, WeightDesc Wt, GridDesc G, Exchange E, int cap,
                                                                                int n_interp, DevParams P, StressArgs A) {
    static_assert(64 * IT_WAVES == 256, "a stress workgroup takes 256 cells");
    if ((int)blockIdx.x < n_interp) {
        interpolate_tiles<ROWS>(S, Wt, G, E, cap, (int)blockIdx.x, n_interp);
        return;
    }
    const int idx = ((int)blockIdx.x - n_interp) * 256 + (int)threadIdx.x;
    if (idx >= G.nx * G.ny) return;
    const int j = idx / G.nx;
    const size_t k = cell_index(G, idx - j * G.nx, j);
    const size_t kw = k - 1, ks = k - (size_t)G.sj;
    const bool wet = cell_is_wet(P, A.mask, k);
    const double aice = A.I.conc ? A.I.conc[k] : 0.0;
    const double tx = net_face_stress(P, A.rtx[kw], A.rtx[k], A.I.conc ? A.I.conc[kw] : 0.0, aice, A.I.txio ? A.I.txio[k] : 0.0);
    const double ty = net_face_stress(P, A.rty[ks], A.rty[k], A.I.conc ? A.I.conc[ks] : 0.0, aice, A.I.tyio ? A.I.tyio[k] : 0.0);
    A.tau_x[k] = wet ? tx : 0.0;
    A.tau_y[k] = wet ? ty : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Background variant: no LDS, ≤ 64 VGPRs, one cell per lane, all 72 corner values gathered from L2.
// Alone it is slower than the tiled kernel (texture-address bound), but it fits into the register and LDS
// budget the flux solver leaves free on every CU (3 × 147 VGPRs, 159 of 160 KB LDS), so on a second stream
// it runs INSIDE the solver of the current step while that kernel is FP64-issue bound
// (cf_prefetch_atmosphere_state).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interpolate_gather_kernel(SourceDesc S, WeightDesc Wt, GridDesc G, Exchange E) {
    const int wx = G.nx + 2 * G.ring, wy = G.ny + 2 * G.ring;
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= wx * wy) return;
    const int jj = idx / wx;
    const int i = idx - jj * wx - G.ring, j = jj - G.ring;
    const unsigned k = (unsigned)((j + G.hy) * G.sj + (i + G.hx));   // surfaces of < 2³² cells
    const double fi = Wt.separable ? Wt.fi[i + G.hx] : Wt.fi[k];
    const double fj = Wt.separable ? Wt.fj[j + G.hy] : Wt.fj[k];
    const double ti = trunc(fi), tj = trunc(fj);
    const double xi = fi - floor(fi), eta = fj - floor(fj);  // ξ = mod(f, 1), see interpolate_kernel
    unsigned g00, g10, g01, g11;   // offsets inside one (level, variable) plane
    {
        const int i0 = (int)ti, ja = (int)tj;
        const int is0 = wrap_index(i0, S.ns_x), is1 = wrap_index(i0 + (fi > 0.0 ? 1 : (fi < 0.0 ? -1 : 0)), S.ns_x);
        const int j0 = min(max(ja, 0), S.ns_y - 1), j1 = min(max(ja + (fj > 0.0 ? 1 : (fj < 0.0 ? -1 : 0)), 0), S.ns_y - 1);
        g00 = (unsigned)(j0 * S.ns_x + is0);
        g10 = (unsigned)(j0 * S.ns_x + is1);
        g01 = (unsigned)(j1 * S.ns_x + is0);
        g11 = (unsigned)(j1 * S.ns_x + is1);
    }
    const unsigned plane = (unsigned)(S.ns_x * S.ns_y);
    const unsigned off1 = (unsigned)S.level1 * plane, off2 = (unsigned)S.level2 * plane;
    const double w00 = (1.0 - xi) * (1.0 - eta), w01 = (1.0 - xi) * eta, w10 = xi * (1.0 - eta), w11 = xi * eta;
    // same operations in the same order as the tiled kernel (bitwise-identical results)
    auto value = [&](int v) {
        const float* a = S.data[v] + off1;
        const float* b = S.data[v] + off2;
        return bilinear(w00, w01, w10, w11, blend_levels(a[g00], b[g00], S.tf), blend_levels(a[g01], b[g01], S.tf),
                        blend_levels(a[g10], b[g10], S.tf), blend_levels(a[g11], b[g11], S.tf));
    };
    // One variable at a time in a rolled loop: ≤ 56 VGPRs is what fits beside three resident solver waves per
    // SIMD (3 × 152 of 512 registers).
    double ua = 0.0, va = 0.0, rain = 0.0;
#pragma unroll 1
    for (int v = 0; v < CF_JRA55_NVARS; ++v) {
        const double x = value(v);
        switch (v) {
            case CF_JRA55_TAS: E.T[k] = x; break;
            case CF_JRA55_PSL: E.p[k] = x; break;
            case CF_JRA55_HUSS: E.q[k] = x; break;
            case CF_JRA55_RSDS: E.Qs[k] = x; break;
            case CF_JRA55_RLDS: E.Ql[k] = x; break;
            case CF_JRA55_PRRA: rain = x; break;
            case CF_JRA55_PRSN: E.Mp[k] = rain + x; break;
            case CF_JRA55_UAS: ua = x; break;
            default: va = x; break;
        }
    }
    if (Wt.cos_rot != nullptr && Wt.sin_rot != nullptr) {
        const double cs = Wt.cos_rot[k], sn = Wt.sin_rot[k];
        const double ui = ua * cs + va * sn;
        va = -ua * sn + va * cs;
        ua = ui;
    }
    E.u[k] = ua;
    E.v[k] = va;
}

// JRA55PrescribedLand: friver + licalvf → one ocean-grid field (same weights and time blend as the atmosphere, gather form)
__global__ __launch_bounds__(256) void interpolate_land_kernel(const float* __restrict__ friver, const float* __restrict__ licalvf,
                                                               int ns_x, int ns_y, int level1, int level2, double tf,
                                                               WeightDesc Wt, GridDesc G, double* __restrict__ out) {
    const int wx = G.nx + 2 * G.ring, wy = G.ny + 2 * G.ring;
    const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (idx >= wx * wy) return;
    const int jj = idx / wx;
    const int i = idx - jj * wx - G.ring, j = jj - G.ring;
    const size_t k = cell_index(G, i, j);
    const double fi = Wt.separable ? Wt.fi[i + G.hx] : Wt.fi[k];
    const double fj = Wt.separable ? Wt.fj[j + G.hy] : Wt.fj[k];
    const double xi = fi - floor(fi), eta = fj - floor(fj);
    const int i0 = (int)trunc(fi), ja = (int)trunc(fj);
    const int is0 = wrap_index(i0, ns_x), is1 = wrap_index(i0 + (fi > 0.0 ? 1 : (fi < 0.0 ? -1 : 0)), ns_x);
    const int j0 = min(max(ja, 0), ns_y - 1), j1 = min(max(ja + (fj > 0.0 ? 1 : (fj < 0.0 ? -1 : 0)), 0), ns_y - 1);
    const size_t g00 = (size_t)j0 * ns_x + is0, g10 = (size_t)j0 * ns_x + is1, g01 = (size_t)j1 * ns_x + is0, g11 = (size_t)j1 * ns_x + is1;
    const size_t plane = (size_t)ns_x * ns_y;
    const double w00 = (1.0 - xi) * (1.0 - eta), w01 = (1.0 - xi) * eta, w10 = xi * (1.0 - eta), w11 = xi * eta;
    auto value = [&](const float* d) {
        const float* a = d + (size_t)level1 * plane;
        const float* b = d + (size_t)level2 * plane;
        return bilinear(w00, w01, w10, w11, blend_levels(a[g00], b[g00], tf), blend_levels(a[g01], b[g01], tf),
                        blend_levels(a[g10], b[g10], tf), blend_levels(a[g11], b[g11], tf));
    };
    out[k] = value(friver) + (licalvf ? value(licalvf) : 0.0);
}

hipError_t launch_interpolate_land(hipStream_t st, const GridDesc& G, const cf_land_source* s, const cf_interp_weights* w, double* out) {
    const int n = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    hipLaunchKernelGGL(interpolate_land_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s->friver, s->licalvf, s->ns_x, s->ns_y,
                       s->level1, s->level2, s->time_fraction, make_weights(w), G, out);
    return hipGetLastError();
}

hipError_t launch_interpolate_background(hipStream_t st, const GridDesc& G, const cf_atmos_source* s,
                                         const cf_interp_weights* w, const cf_exchange_fields* e) {
    const int n = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    hipLaunchKernelGGL(interpolate_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, make_source(s), make_weights(w), G,
                       make_exchange(e));
    return hipGetLastError();
}

// rows per tile and workgroups of the tiled interpolation on this surface
void interpolate_grid(const LaunchCfg& L, const GridDesc& G, int* rows_out, int* blocks_out) {
    // Rows of 64 cells per wave tile.  A tile is one dependent chain (indices → footprint → LDS → 8 stores per row),
    // and on a surface that does not fill the device's wave slots the kernel's time IS that chain: fewer rows per
    // tile then mean more waves and a shorter chain (1440×70: 7.8 → 5.4 µs with one row), while on the full surface
    // four rows amortise the footprint best (measured; thresholds in tiles of four rows, per 256 CUs).
    const int wx = G.nx + 2 * G.ring, wy = G.ny + 2 * G.ring;
    const int tiles_x = (wx + 63) / 64;
    const long tiles4 = (long)tiles_x * ((wy + 3) / 4) * 256 / (L.cu_count > 0 ? L.cu_count : 256);
    const int rows = tiles4 >= 1200 ? 4 : (tiles4 >= 600 ? 2 : 1);
    const int ntiles = tiles_x * ((wy + rows - 1) / rows);
    int blocks = (ntiles + IT_WAVES - 1) / IT_WAVES;
    // Everything resident at once (≤ 4 workgroups per CU: LDS) but not evenly — 817 workgroups on 256 CUs leave 49 CUs
    // with four and the rest with three, and all waves hit the store phase together: a grid of ≈ 78 % of the tiles' waves,
    // a fifth of them taking a second tile, staggers the phases (measured at 1440×560: 817 → 18.0, 704 → 17.4, 640 → 17.2,
    // 600 → 17.3, 560 → 17.5 µs; no effect below 2 workgroups per CU, slightly negative above 4: left alone there)
    const int cus = L.cu_count > 0 ? L.cu_count : 256;
    if (blocks > 2 * cus && blocks <= 4 * cus) blocks = std::max(2 * cus, blocks * 25 / 32);
    static const int blocks_cap = [] {  // (experiments: COFLUX_EXPERIMENTS=1 COFLUX_INTERP_BLOCKS=n, read once)
        const char* cap = experiment_knob("COFLUX_INTERP_BLOCKS");
        return cap ? std::max(1, std::atoi(cap)) : 0;
    }();
    if (blocks_cap > 0) blocks = std::min(blocks, blocks_cap);
    *rows_out = rows;
    *blocks_out = blocks;
}

hipError_t launch_interpolate(hipStream_t st, const LaunchCfg& L, const GridDesc& G, const cf_atmos_source* s,
                              const cf_interp_weights* w, const cf_exchange_fields* e) {
    if (L.interp_cap == 0) return launch_interpolate_background(st, G, s, w, e);  // CF_OPT_INTERP_TILE_CAP = 0
    int rows = 4, blocks = 1;
    interpolate_grid(L, G, &rows, &blocks);
    const size_t lds = (size_t)IT_WAVES * CF_JRA55_NVARS * L.interp_cap * sizeof(double);
#define CF_LAUNCH_INTERP(ROWS_)                                                                                        \
    hipLaunchKernelGGL(interpolate_kernel<ROWS_>, dim3(blocks), dim3(64 * IT_WAVES), lds, st, make_source(s),          \
                       make_weights(w), G, make_exchange(e), L.interp_cap)
    if (rows == 4) CF_LAUNCH_INTERP(4);
    else if (rows == 2) CF_LAUNCH_INTERP(2);
    else CF_LAUNCH_INTERP(1);
#undef CF_LAUNCH_INTERP
    return hipGetLastError();
}

// the face stresses of one step and the interpolation of the next step's atmosphere state in one launch (see the kernel)
hipError_t launch_interpolate_and_stress(hipStream_t st, const LaunchCfg& L, const DevParams& P, const GridDesc& G,
                                         const cf_atmos_source* s, const cf_interp_weights* w, const cf_exchange_fields* e,
                                         const cf_ocean_surface* o, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                                         const cf_net_ocean_fluxes* n) {
    if (L.interp_cap == 0) return hipErrorInvalidValue;
    int rows = 4, blocks = 1;
    interpolate_grid(L, G, &rows, &blocks);
    const size_t lds = (size_t)IT_WAVES * CF_JRA55_NVARS * L.interp_cap * sizeof(double);
    StressArgs A{};
    A.mask = o->mask;
    A.rtx = f->x_momentum;
    A.rty = f->y_momentum;
    if (ice) A.I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress, nullptr};
    A.tau_x = n->u;
    A.tau_y = n->v;
    const int stress_blocks = (G.nx * G.ny + 255) / 256;
#define CF_LAUNCH_BOTH(ROWS_)                                                                                              \
    hipLaunchKernelGGL(interpolate_and_stress_kernel<ROWS_>, dim3(blocks + stress_blocks), dim3(64 * IT_WAVES), lds, st, \
                       make_source(s), make_weights(w), G, make_exchange(e), L.interp_cap, blocks, P, A)
    if (rows == 4) CF_LAUNCH_BOTH(4);
    else if (rows == 2) CF_LAUNCH_BOTH(2);
    else CF_LAUNCH_BOTH(1);
#undef CF_LAUNCH_BOTH
    return hipGetLastError();
}

}  // namespace coflux
