// coflux_interp_tiles.hpp — the tiled interpolation of interpolate_atmosphere_state! as a device routine: one wave stages
// the JRA55 footprint of its tile of 64 × ROWS cells in LDS and reads the corners from there (design notes in
// coflux_interp.hip).  Shared by interpolate_kernel, the merged stress + interpolation launch and the lean solver's
// tail workgroups (coflux_solver_lean.hip): one routine, the same bits wherever it runs.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>

#include "coflux_interp_cell.hpp"
#include "coflux_kernel_types.hpp"

namespace coflux {

constexpr int IT_WAVES = 4;  // waves per workgroup (independent of each other)

// Wave-wide min / max as a wave-uniform value.  Four DPP steps (xor 1, xor 2, mirror within 8, mirror within 16)
// leave every lane with its 16-lane row's result, then one lane per row is read: ≈ 10 short instructions, where the
// __shfl_xor butterfly is six dependent ds_bpermute round trips through LDS (four such reductions open every tile).
template <bool MIN>
__device__ __forceinline__ int wave_reduce(int v) {
    auto op = [](int a, int b) { return MIN ? min(a, b) : max(a, b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return op(op(a, b), op(c, d));
}
__device__ __forceinline__ int wave_min(int v) { return wave_reduce<true>(v); }
__device__ __forceinline__ int wave_max(int v) { return wave_reduce<false>(v); }

// (blend_levels, bilinear, the corner convention: coflux_interp_cell.hpp — shared with the solver's fused prologue)
// LDS traffic of one wave is ordered by issue; this only stops the compiler from moving accesses.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the tiles of workgroup `block` of `nblocks` (the kernel below; the merged stress + interpolation launch)
template <int ROWS>
__device__ __forceinline__ void interpolate_tiles(const SourceDesc& S, const WeightDesc& Wt, const GridDesc& G, const Exchange& E, int cap,
                                                  int block, int nblocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* tile = reinterpret_cast<double*>(smem) + (size_t)wave * CF_JRA55_NVARS * cap;

    const int wx = G.nx + 2 * G.ring, wy = G.ny + 2 * G.ring;
    const int tiles_x = (wx + 63) / 64, tiles_y = (wy + ROWS - 1) / ROWS;
    const int ntiles = tiles_x * tiles_y;
    const size_t plane = (size_t)S.ns_x * S.ns_y;
    const size_t off1 = (size_t)S.level1 * plane, off2 = (size_t)S.level2 * plane;
    const int half = S.ns_x / 2;
    const bool rotate = Wt.cos_rot != nullptr && Wt.sin_rot != nullptr;

    for (int t = block * IT_WAVES + wave; t < ntiles; t += nblocks * IT_WAVES) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int i = tx * 64 + lane - G.ring;
        const int ic = min(i, G.nx + G.ring - 1);  // out-of-window lanes shadow the last column

        // ---- per-row corner indices and weights ---------------------------------------------------
        int d0[ROWS], di[ROWS], j0[ROWS], j1[ROWS];
        double xi[ROWS], eta[ROWS];
        int lo = INT_MAX, hi = INT_MIN, jlo = INT_MAX, jhi = INT_MIN, ref = 0;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int jc = min(ty * ROWS + r - G.ring, G.ny + G.ring - 1);
            const size_t k = cell_index(G, ic, jc);
            const double fi = Wt.separable ? Wt.fi[ic + G.hx] : Wt.fi[k];
            const double fj = Wt.separable ? Wt.fj[jc + G.hy] : Wt.fj[k];
            // Oceananigans `interpolator`: i⁻ = trunc(f), i⁺ = i⁻ + sign(f), ξ = mod(f, 1) ∈ [0, 1) — for a negative
            // fractional index (a column west of the first source node) that is f − floor(f), not f − trunc(f)
            const double ti = trunc(fi), tj = trunc(fj);
            xi[r] = fi - floor(fi);
            eta[r] = fj - floor(fj);
            const int i0 = (int)ti;
            di[r] = fi > 0.0 ? 1 : (fi < 0.0 ? -1 : 0);
            const int ja = (int)tj;
            const int jb = ja + (fj > 0.0 ? 1 : (fj < 0.0 ? -1 : 0));
            j0[r] = min(max(ja, 0), S.ns_y - 1);  // clamped in latitude
            j1[r] = min(max(jb, 0), S.ns_y - 1);
            if (r == 0) ref = __shfl(i0, 0);
            // column offset relative to the tile's reference column, periodic in longitude
            d0[r] = wrap_index(i0 - ref + half, S.ns_x) - half;
            lo = min(lo, min(d0[r], d0[r] + di[r]));
            hi = max(hi, max(d0[r], d0[r] + di[r]));
            jlo = min(jlo, min(j0[r], j1[r]));
            jhi = max(jhi, max(j0[r], j1[r]));
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        jlo = wave_min(jlo);
        jhi = wave_max(jhi);
        const int W = hi - lo + 1, H = jhi - jlo + 1, WH = W * H;
        const bool fits = WH <= cap && W <= S.ns_x;

        // ---- stage the footprint: tile[var][y][x] = the node's value at the time fraction ---------------
        if (fits) {
            const float inv_W = 1.0f / (float)W;
            for (int rem = lane; rem < WH; rem += 64) {
                const int y = (int)(((float)rem + 0.5f) * inv_W);
                const int x = rem - y * W;
                const size_t off = (size_t)(jlo + y) * S.ns_x + wrap_index(ref + lo + x, S.ns_x);
#pragma unroll
                for (int v = 0; v < CF_JRA55_NVARS; ++v)
                    tile[v * cap + rem] = blend_levels(S.data[v][off1 + off], S.data[v][off2 + off], S.tf);
            }
        }
        wave_lds_sync();

        // ---- interpolate ---------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int j = ty * ROWS + r - G.ring;
            const double w00 = (1.0 - xi[r]) * (1.0 - eta[r]), w01 = (1.0 - xi[r]) * eta[r];
            const double w10 = xi[r] * (1.0 - eta[r]), w11 = xi[r] * eta[r];
            double val[CF_JRA55_NVARS];
            if (fits) {
                const int o00 = (j0[r] - jlo) * W + (d0[r] - lo), o01 = (j1[r] - jlo) * W + (d0[r] - lo);
#pragma unroll
                for (int v = 0; v < CF_JRA55_NVARS; ++v) {
                    const double* p = tile + v * cap;
                    val[v] = bilinear(w00, w01, w10, w11, p[o00], p[o01], p[o00 + di[r]], p[o01 + di[r]]);
                }
            } else {  // footprint too large for the tile (coarse target grid): gather from L2
                const int is0 = wrap_index(ref + d0[r], S.ns_x), is1 = wrap_index(ref + d0[r] + di[r], S.ns_x);
                const size_t g00 = (size_t)j0[r] * S.ns_x + is0, g10 = (size_t)j0[r] * S.ns_x + is1;
                const size_t g01 = (size_t)j1[r] * S.ns_x + is0, g11 = (size_t)j1[r] * S.ns_x + is1;
#pragma unroll
                for (int v = 0; v < CF_JRA55_NVARS; ++v) {
                    const float* a = S.data[v] + off1;
                    const float* b = S.data[v] + off2;
                    val[v] = bilinear(w00, w01, w10, w11, blend_levels(a[g00], b[g00], S.tf), blend_levels(a[g01], b[g01], S.tf),
                                      blend_levels(a[g10], b[g10], S.tf), blend_levels(a[g11], b[g11], S.tf));
                }
            }
            if (i < G.nx + G.ring && j < G.ny + G.ring) {
                const size_t k = cell_index(G, i, j);
                double ua = val[CF_JRA55_UAS], va = val[CF_JRA55_VAS];
                if (rotate) {  // intrinsic_vector: geographic (E, N) → grid frame
                    const double cs = Wt.cos_rot[k], sn = Wt.sin_rot[k];
                    const double ui = ua * cs + va * sn;
                    va = -ua * sn + va * cs;
                    ua = ui;
                }
                E.u[k] = ua;
                E.v[k] = va;
                E.T[k] = val[CF_JRA55_TAS];
                E.p[k] = val[CF_JRA55_PSL];
                E.q[k] = val[CF_JRA55_HUSS];
                E.Qs[k] = val[CF_JRA55_RSDS];
                E.Ql[k] = val[CF_JRA55_RLDS];
                E.Mp[k] = val[CF_JRA55_PRRA] + val[CF_JRA55_PRSN];
            }
        }
        wave_lds_sync();  // the tile is rewritten by this wave's next iteration
    }
}

}  // namespace coflux
