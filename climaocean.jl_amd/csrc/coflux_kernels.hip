// coflux_kernels.hip — gfx950 kernels of the surface-flux path.
//
//   interpolate_kernel ....... interpolate_atmosphere_state!   (JRA55 f32 window → 8 exchange fields)
//   ao_flux_fast_kernel ...... compute_atmosphere_ocean_fluxes! (Monin–Obukhov fixed point, LDS tables)
//   fused_fast_kernel ........ the two above in one pass (update_state! path)
//   net_flux_kernel .......... compute_net_ocean_fluxes!        (radiation + partition)
//   ao_flux_libm_kernel ...... cross-check variant of the solver on ocml's libm (CF_SOLVER_LIBM)
//
// All are pointwise / 1-cell-stencil kernels: nothing here is a contraction, so no MFMA.
// Lanes run along i (the contiguous axis) so every field access is a coalesced 512-B wave
// transaction; the JRA55 source tile a workgroup needs is staged once through LDS; the solver's
// ψ / log tables live in LDS; workgroups are persistent (grid = k × 256 CUs) and walk tiles so
// that each XCD keeps to its own latitude band of the JRA55 window (private 4 MB L2 per XCD).
#include <hip/hip_runtime.h>

#include "coflux_device.hpp"
#include "coflux_fast.hpp"
#include "coflux_kernels.h"

namespace coflux {

constexpr int TILE_X = 64;                   // one wave spans 64 consecutive i
constexpr int TILE_Y = 4;                    // 4 waves per workgroup, one row each
constexpr int NPLANES = 2 * CF_JRA55_NVARS;  // (variable, time level)
constexpr int BOX_BYTES = 32;                // 8 ints of tile bookkeeping in dynamic LDS
constexpr int TABLE_BYTES = TABLE_DOUBLES * 8;
constexpr int NUM_XCD = 8;

// ---------------------------------------------------------------------------------------------
// Persistent tile walk.  Workgroup b is (observed to be) placed on XCD b % 8; giving XCD x the
// contiguous tile range [x·chunk, (x+1)·chunk) keeps the JRA55 rows it touches (1/8 of 14.7 MB)
// inside that XCD's own L2.  Placement only affects speed, never results.
// ---------------------------------------------------------------------------------------------
struct TileWalk {
    int tile, end, step;
};
__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
    const int b = blockIdx.x, nb = gridDim.x;
    TileWalk w;
    if (nb % NUM_XCD == 0 && nb >= NUM_XCD) {
        const int chunk = (ntiles + NUM_XCD - 1) / NUM_XCD;
        const int xcd = b % NUM_XCD, slot = b / NUM_XCD;
        w.tile = xcd * chunk + slot;
        w.end = min((xcd + 1) * chunk, ntiles);
        w.step = nb / NUM_XCD;
    } else {
        w.tile = b;
        w.end = ntiles;
        w.step = nb;
    }
    return w;
}

// =============================================================================================
// JRA55 tile staging + bilinear × linear-in-time interpolation
// =============================================================================================
struct InterpCell {
    double v[CF_JRA55_NVARS];
};

struct SourceDesc {
    const float* data[CF_JRA55_NVARS];
    int32_t ns_x, ns_y, level1, level2;
    double tf;
};

struct WeightDesc {
    const double* fi;
    const double* fj;
    const double* cos_rot;
    const double* sin_rot;
    const double* latitude;
    int32_t separable;
};

__device__ __forceinline__ int wrap_index(int i, int n) {
    int r = i % n;
    return r < 0 ? r + n : r;
}

// Stages the (≤ cap floats per plane) source footprint of this workgroup's TILE_X×TILE_Y cells
// into LDS and interpolates the 9 variables for the calling thread's cell.  When the footprint
// does not fit (coarse target grids, folds) the workgroup gathers from global memory instead —
// the window is 14.7 MB and lives in L2 / Infinity Cache.  `box` = 8 ints of dynamic LDS.
// Ends with the tile still live in LDS: callers __syncthreads() before the next tile.
__device__ __forceinline__ InterpCell interpolate_cell(const SourceDesc& S, const WeightDesc& Wt, const GridDesc& G,
                                                       int i, int j, int* box, float* lds, int cap) {
    const int tid = threadIdx.y * TILE_X + threadIdx.x;

    // clamp out-of-window threads onto a valid cell so that they do not widen the footprint
    const int ic = min(max(i, -G.ring), G.nx + G.ring - 1);
    const int jc = min(max(j, -G.ring), G.ny + G.ring - 1);
    const size_t k = cell_index(G, ic, jc);
    const double fi = Wt.separable ? Wt.fi[ic + G.hx] : Wt.fi[k];
    const double fj = Wt.separable ? Wt.fj[jc + G.hy] : Wt.fj[k];

    const double ti = trunc(fi), tj = trunc(fj);
    const double xi = fi - ti, eta = fj - tj;
    const int i0 = (int)ti;
    int j0 = (int)tj;
    const int i1 = i0 + (fi > 0.0 ? 1 : (fi < 0.0 ? -1 : 0));
    int j1 = j0 + (fj > 0.0 ? 1 : (fj < 0.0 ? -1 : 0));
    j0 = min(max(j0, 0), S.ns_y - 1);
    j1 = min(max(j1, 0), S.ns_y - 1);

    if (tid == 0) {
        box[0] = INT_MAX;
        box[1] = INT_MIN;
        box[2] = INT_MAX;
        box[3] = INT_MIN;
        box[4] = i0;
    }
    __syncthreads();
    const int ref = box[4];
    // offsets relative to the tile's reference column, wrapped to [−ns_x/2, ns_x/2)
    const int half = S.ns_x / 2;
    const int d0 = wrap_index(i0 - ref + half, S.ns_x) - half;
    const int d1 = d0 + (i1 - i0);
    atomicMin(&box[0], min(d0, d1));
    atomicMax(&box[1], max(d0, d1));
    atomicMin(&box[2], min(j0, j1));
    atomicMax(&box[3], max(j0, j1));
    __syncthreads();
    const int dmin = box[0], jmin = box[2];
    const int W = box[1] - dmin + 1, H = box[3] - jmin + 1;
    const bool fits = (W * H <= cap) && (W <= S.ns_x);

    const size_t plane_stride = (size_t)S.ns_x * S.ns_y;
    if (fits) {
        const int per_plane = W * H;
        const int total = per_plane * NPLANES;
        for (int e = tid; e < total; e += TILE_X * TILE_Y) {
            const int p = e / per_plane;
            const int r = e - p * per_plane;
            const int y = r / W;
            const int x = r - y * W;
            const int var = p >> 1;
            const int lev = (p & 1) ? S.level2 : S.level1;
            const int is = wrap_index(ref + dmin + x, S.ns_x);
            lds[p * cap + r] = S.data[var][(size_t)lev * plane_stride + (size_t)(jmin + y) * S.ns_x + is];
        }
    }
    __syncthreads();

    InterpCell out;
    const double w00 = (1.0 - xi) * (1.0 - eta), w01 = (1.0 - xi) * eta;
    const double w10 = xi * (1.0 - eta), w11 = xi * eta;
    if (fits) {
        const int o00 = (j0 - jmin) * W + (d0 - dmin), o10 = (j0 - jmin) * W + (d1 - dmin);
        const int o01 = (j1 - jmin) * W + (d0 - dmin), o11 = (j1 - jmin) * W + (d1 - dmin);
#pragma unroll
        for (int var = 0; var < CF_JRA55_NVARS; ++var) {
            const float* a = lds + (2 * var) * cap;
            const float* b = a + cap;
            double v1 = w00 * (double)a[o00] + w01 * (double)a[o01] + w10 * (double)a[o10] + w11 * (double)a[o11];
            double v2 = w00 * (double)b[o00] + w01 * (double)b[o01] + w10 * (double)b[o10] + w11 * (double)b[o11];
            out.v[var] = v2 * S.tf + v1 * (1.0 - S.tf);
        }
    } else {
        const int is0 = wrap_index(i0, S.ns_x), is1 = wrap_index(i1, S.ns_x);
        const size_t g00 = (size_t)j0 * S.ns_x + is0, g10 = (size_t)j0 * S.ns_x + is1;
        const size_t g01 = (size_t)j1 * S.ns_x + is0, g11 = (size_t)j1 * S.ns_x + is1;
#pragma unroll
        for (int var = 0; var < CF_JRA55_NVARS; ++var) {
            const float* a = S.data[var] + (size_t)S.level1 * plane_stride;
            const float* b = S.data[var] + (size_t)S.level2 * plane_stride;
            double v1 = w00 * (double)a[g00] + w01 * (double)a[g01] + w10 * (double)a[g10] + w11 * (double)a[g11];
            double v2 = w00 * (double)b[g00] + w01 * (double)b[g01] + w10 * (double)b[g10] + w11 * (double)b[g11];
            out.v[var] = v2 * S.tf + v1 * (1.0 - S.tf);
        }
    }
    return out;
}

struct Exchange {
    double* u;
    double* v;
    double* T;
    double* p;
    double* q;
    double* Qs;
    double* Ql;
    double* Mp;
};

struct AtmosCell {
    double u, v, T, p, q, Qs, Ql, Mp;
};

__device__ __forceinline__ AtmosCell finish_interp(const InterpCell& c, const WeightDesc& Wt, size_t k) {
    AtmosCell a;
    a.u = c.v[CF_JRA55_UAS];
    a.v = c.v[CF_JRA55_VAS];
    if (Wt.cos_rot != nullptr && Wt.sin_rot != nullptr) {  // geographic (E,N) → grid-intrinsic frame
        double cs = Wt.cos_rot[k], sn = Wt.sin_rot[k];
        double ui = a.u * cs + a.v * sn;
        double vi = -a.u * sn + a.v * cs;
        a.u = ui;
        a.v = vi;
    }
    a.T = c.v[CF_JRA55_TAS];
    a.p = c.v[CF_JRA55_PSL];
    a.q = c.v[CF_JRA55_HUSS];
    a.Qs = c.v[CF_JRA55_RSDS];
    a.Ql = c.v[CF_JRA55_RLDS];
    a.Mp = c.v[CF_JRA55_PRRA] + c.v[CF_JRA55_PRSN];
    return a;
}

__device__ __forceinline__ void store_exchange(const Exchange& E, size_t k, const AtmosCell& a) {
    E.u[k] = a.u;
    E.v[k] = a.v;
    E.T[k] = a.T;
    E.p[k] = a.p;
    E.q[k] = a.q;
    E.Qs[k] = a.Qs;
    E.Ql[k] = a.Ql;
    E.Mp[k] = a.Mp;
}

__global__ __launch_bounds__(TILE_X* TILE_Y) void interpolate_kernel(SourceDesc S, WeightDesc Wt, GridDesc G,
                                                                      Exchange E, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* box = reinterpret_cast<int*>(smem);
    float* lds = reinterpret_cast<float*>(smem + BOX_BYTES);
    const int tiles_x = (G.nx + 2 * G.ring + TILE_X - 1) / TILE_X;
    const int tiles_y = (G.ny + 2 * G.ring + TILE_Y - 1) / TILE_Y;
    for (TileWalk w = tile_walk(tiles_x * tiles_y); w.tile < w.end; w.tile += w.step) {
        const int ty = w.tile / tiles_x, tx = w.tile - ty * tiles_x;
        const int i = tx * TILE_X + (int)threadIdx.x - G.ring;
        const int j = ty * TILE_Y + (int)threadIdx.y - G.ring;
        const bool in_range = (i < G.nx + G.ring) && (j < G.ny + G.ring);
        InterpCell c = interpolate_cell(S, Wt, G, i, j, box, lds, cap);
        if (in_range) {
            size_t k = cell_index(G, i, j);
            store_exchange(E, k, finish_interp(c, Wt, k));
        }
        __syncthreads();  // tile and box are rewritten by the next iteration
    }
}

// =============================================================================================
// Monin–Obukhov solver kernels
// =============================================================================================
struct OceanIn {
    const double* T;
    const double* S;
    const double* u;
    const double* v;
    const void* mask;
};

struct FluxOut {
    double* Qc;
    double* Qv;
    double* Fv;
    double* tx;
    double* ty;
    double* Ts;
    double* ustar;
    double* tstar;
    double* qstar;
    int32_t* iters;
};

__device__ __forceinline__ void store_fluxes(const FluxOut& F, size_t k, const CellFluxes& R) {
    F.Qc[k] = R.Qc;
    F.Qv[k] = R.Qv;
    F.Fv[k] = R.Fv;
    F.tx[k] = R.rho_tau_x;
    F.ty[k] = R.rho_tau_y;
    F.Ts[k] = R.Ts_ocean;
    if (F.ustar) F.ustar[k] = R.ustar;
    if (F.tstar) F.tstar[k] = R.tstar;
    if (F.qstar) F.qstar[k] = R.qstar;
    if (F.iters) F.iters[k] = R.iterations;
}

constexpr int AO_BLOCK = 256;
constexpr int AO_CHUNK = 512;  // cells classified + compacted per pass of a workgroup
constexpr int AO_LDS_BYTES = TABLE_BYTES + AO_CHUNK * 4 + 16;

// ---- production solver: LDS tables, persistent workgroups ------------------------------------
template <bool COARE>
__global__ __launch_bounds__(AO_BLOCK) void ao_flux_fast_kernel(DevParams P, FastConsts C, GridDesc G, OceanIn O,
                                                                Exchange E, FluxOut F,
                                                                const double* __restrict__ g_tab) {
    // Land cells (≈30 % of a global grid) must not occupy lanes for 10–20 iterations: every chunk of
    // AO_CHUNK cells is first compacted to the list of its wet cells (land gets its zeros there and
    // then), and waves then pull 64 list entries at a time from an LDS cursor, so every lane that
    // enters the solver holds an ocean cell.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    int* list = reinterpret_cast<int*>(smem + TABLE_BYTES);
    int* counters = list + AO_CHUNK;  // [0] wet count, [1] cursor
    const int tid = threadIdx.x, lane = tid & 63;
    stage_tables(tab, g_tab, tid, AO_BLOCK);

    const int wx = G.nx + 2 * G.ring;
    const int ncells = wx * (G.ny + 2 * G.ring);
    const int nchunks = (ncells + AO_CHUNK - 1) / AO_CHUNK;
    const bool fixed = P.stop_kind == CF_STOP_FIXED;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (tid < 2) counters[tid] = 0;
        __syncthreads();
        // ---- phase 1: classify, zero land, compact wet cells ------------------------------------
        const int begin = chunk * AO_CHUNK, end = min(begin + AO_CHUNK, ncells);
        for (int base = begin; base < end; base += AO_BLOCK) {
            const int idx = base + tid;
            const bool in_range = idx < end;
            bool wet = false;
            if (in_range) {
                const int jj = idx / wx;
                const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
                wet = cell_is_wet(P, O.mask, k);
                if (!wet) {  // zero_interface_state: all fluxes 0, T = 0 K
                    CellFluxes Z{};
                    Z.Ts_ocean = -P.T_offset;
                    Z.iterations = fixed ? P.maxiter : 0;
                    store_fluxes(F, k, Z);
                }
            }
            const unsigned long long m = __ballot(wet);
            int wave_base = 0;
            if (lane == 0 && m) wave_base = atomicAdd(&counters[0], __popcll(m));
            wave_base = __shfl(wave_base, 0);
            if (wet) list[wave_base + __popcll(m & ((1ull << lane) - 1ull))] = idx;
        }
        __syncthreads();
        const int nwet = counters[0];
        // ---- phase 2: waves pull 64 wet cells at a time ------------------------------------------
        for (;;) {
            int start = 0;
            if (lane == 0) start = atomicAdd(&counters[1], 64);
            start = __shfl(start, 0);
            if (start >= nwet) break;
            const int e = start + lane;
            const bool in_range = e < nwet;
            const int idx = list[in_range ? e : nwet - 1];
            const int jj = idx / wx;
            const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
            // ℑxᶜᵃᵃ u, ℑyᵃᶜᵃ v: cell-centre ocean velocity from the two bracketing faces
            const double uo = 0.5 * (O.u[k] + O.u[k + 1]);
            const double vo = 0.5 * (O.v[k] + O.v[k + (size_t)G.sj]);
            CellFluxes R = solve_cell_fast<COARE>(P, C, tab, E.u[k], E.v[k], E.T[k], E.p[k], E.q[k], uo, vo, O.T[k],
                                                  O.S[k], true, in_range);
            if (in_range) store_fluxes(F, k, R);
        }
        __syncthreads();  // list and counters are reused by the next chunk
    }
}

template <bool COARE>
__global__ __launch_bounds__(TILE_X* TILE_Y) void fused_fast_kernel(DevParams P, FastConsts C, SourceDesc S,
                                                                     WeightDesc Wt, GridDesc G, OceanIn O, Exchange E,
                                                                     FluxOut F, const double* __restrict__ g_tab,
                                                                     int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    int* box = reinterpret_cast<int*>(smem + TABLE_BYTES);
    float* lds = reinterpret_cast<float*>(smem + TABLE_BYTES + BOX_BYTES);
    const int tid = threadIdx.y * TILE_X + threadIdx.x;
    stage_tables(tab, g_tab, tid, TILE_X * TILE_Y);  // visible after the first barrier in interpolate_cell

    const int tiles_x = (G.nx + 2 * G.ring + TILE_X - 1) / TILE_X;
    const int tiles_y = (G.ny + 2 * G.ring + TILE_Y - 1) / TILE_Y;
    for (TileWalk w = tile_walk(tiles_x * tiles_y); w.tile < w.end; w.tile += w.step) {
        const int ty = w.tile / tiles_x, tx = w.tile - ty * tiles_x;
        const int i = tx * TILE_X + (int)threadIdx.x - G.ring;
        const int j = ty * TILE_Y + (int)threadIdx.y - G.ring;
        const bool in_range = (i < G.nx + G.ring) && (j < G.ny + G.ring);
        InterpCell c = interpolate_cell(S, Wt, G, i, j, box, lds, cap);
        const int ic = min(i, G.nx + G.ring - 1), jc = min(j, G.ny + G.ring - 1);
        const size_t k = cell_index(G, ic, jc);
        AtmosCell a = finish_interp(c, Wt, k);
        if (in_range) store_exchange(E, k, a);

        const double uo = 0.5 * (O.u[k] + O.u[k + 1]);
        const double vo = 0.5 * (O.v[k] + O.v[k + (size_t)G.sj]);
        const bool wet = cell_is_wet(P, O.mask, k);
        CellFluxes R = solve_cell_fast<COARE>(P, C, tab, a.u, a.v, a.T, a.p, a.q, uo, vo, O.T[k], O.S[k], wet, in_range);
        if (in_range) store_fluxes(F, k, R);
        __syncthreads();  // tile and box are rewritten by the next iteration
    }
}

// ---- cross-check solver on ocml's libm (CF_SOLVER_LIBM) --------------------------------------
template <int STAB, bool COARE>
__global__ __launch_bounds__(AO_BLOCK) void ao_flux_libm_kernel(DevParams P, GridDesc G, OceanIn O, Exchange E,
                                                                FluxOut F) {
    const int wx = G.nx + 2 * G.ring;
    const int ncells = wx * (G.ny + 2 * G.ring);
    const int idx = (int)blockIdx.x * AO_BLOCK + (int)threadIdx.x;
    const bool in_range = idx < ncells;
    const int cidx = in_range ? idx : ncells - 1;
    const int jj = cidx / wx;
    const int i = cidx - jj * wx - G.ring;
    const int j = jj - G.ring;
    const size_t k = cell_index(G, i, j);
    const double uo = 0.5 * (O.u[k] + O.u[k + 1]);
    const double vo = 0.5 * (O.v[k] + O.v[k + (size_t)G.sj]);
    const bool wet = cell_is_wet(P, O.mask, k);
    CellFluxes R;
    if (P.stop_kind == CF_STOP_FIXED)
        R = solve_cell<STAB, COARE, true>(P, E.u[k], E.v[k], E.T[k], E.p[k], E.q[k], uo, vo, O.T[k], O.S[k], wet, in_range);
    else
        R = solve_cell<STAB, COARE, false>(P, E.u[k], E.v[k], E.T[k], E.p[k], E.q[k], uo, vo, O.T[k], O.S[k], wet, in_range);
    if (in_range) store_fluxes(F, k, R);
}

// =============================================================================================
// Net ocean fluxes: radiation + (1 − ℵ) partition + unit conversion
// =============================================================================================
struct IceIn {
    const double* conc;
    const double* Qio;
    const double* Jsio;
    const double* txio;
    const double* tyio;
};

struct NetOut {
    double* u;
    double* v;
    double* T;
    double* S;
    double* sw;
    double* lw_up;
    double* lw_down;
    double* sw_down;
};

constexpr int NET_BLOCK = 256;

__global__ __launch_bounds__(NET_BLOCK) void net_flux_kernel(DevParams P, GridDesc G, OceanIn O, Exchange E,
                                                             FluxOut F, IceIn I, WeightDesc Wt, NetOut N) {
    const int ncells = G.nx * G.ny;
    const int idx = (int)blockIdx.x * NET_BLOCK + (int)threadIdx.x;
    if (idx >= ncells) return;
    const int j = idx / G.nx;
    const int i = idx - j * G.nx;
    const size_t k = cell_index(G, i, j);
    const size_t kw = k - 1, ks = k - (size_t)G.sj;

    const bool wet = cell_is_wet(P, O.mask, k);
    const double aice = I.conc ? I.conc[k] : 0.0;
    const double aice_w = I.conc ? I.conc[kw] : 0.0;
    const double aice_s = I.conc ? I.conc[ks] : 0.0;
    const double So = O.S[k];
    const double Ts = F.Ts[k] + P.T_offset;
    const double Mp = E.Mp[k], Qs = E.Qs[k], Ql = E.Ql[k];
    const double Qc = F.Qc[k], Qv = F.Qv[k], Mv = F.Fv[k];

    double alb = P.albedo;
    if (P.albedo_kind == CF_ALBEDO_LATITUDE_DEPENDENT) {
        double phi = Wt.separable ? Wt.latitude[j + G.hy] : Wt.latitude[k];
        alb = P.albedo_diffuse - P.albedo_direct * cos(2.0 * phi * (CF_PI / 180.0));
    }
    const double T2 = Ts * Ts;
    const double Qu = P.emissivity * P.sigma * T2 * T2;
    const double Qal = -P.emissivity * Ql;
    const double Qts = -(1.0 - alb) * Qs * (1.0 - aice);
    const double Qss = P.penetrating_sw ? 0.0 : Qts;
    const double SQao = (Qu + Qc + Qv + Qal) * (1.0 - aice) + Qss;

    const double SFao = -Mp * P.rho_f_inv + Mv * P.rho_f_inv;
    const double SFs = (So < P.S_min && SFao < 0.0) ? 0.0 : SFao;

    const double Qio = I.Qio ? I.Qio[k] : 0.0;
    const double Jsio = I.Jsio ? I.Jsio[k] : 0.0;
    const double roc = P.rho_o_inv * P.c_o_inv;
    const double JT = SQao * roc + Qio * roc;
    const double JS = (1.0 - aice) * (-So * SFs) + Jsio;

    const double txao = 0.5 * (F.tx[kw] + F.tx[k]) * P.rho_o_inv;
    const double tyao = 0.5 * (F.ty[ks] + F.ty[k]) * P.rho_o_inv;
    const double ax = 0.5 * (aice_w + aice), ay = 0.5 * (aice_s + aice);
    const double txio = I.txio ? I.txio[k] : 0.0;
    const double tyio = I.tyio ? I.tyio[k] : 0.0;

    const double wf = wet ? 1.0 : 0.0;
    N.u[k] = wf * ((1.0 - ax) * txao + ax * txio);
    N.v[k] = wf * ((1.0 - ay) * tyao + ay * tyio);
    N.T[k] = wf * JT;
    N.S[k] = wf * JS;
    if (N.sw) N.sw[k] = wf * Qts * roc;
    if (N.lw_up) N.lw_up[k] = wf * Qu;
    if (N.lw_down) N.lw_down[k] = wf * (-Qal);
    if (N.sw_down) N.sw_down[k] = wf * (-Qts);
}

// ---------------------------------------------------------------------------------------------
// table / primitive self-test: y[n] = fn(x[n]) with the device's fast primitives (tests only)
// ---------------------------------------------------------------------------------------------
__global__ void debug_eval_kernel(int fn, int n, const double* __restrict__ x, double* __restrict__ y,
                                  const double* __restrict__ g_tab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    stage_tables(tab, g_tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const double* logt = tab + 4 * PSI_TABLE;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const double v = x[k];
        double r;
        switch (fn) {
            case 0: r = flog(logt, v); break;
            case 1: r = fexp(v); break;
            case 2: r = fcbrt(v); break;
            case 3: r = fsqrt(v); break;
            case 4: r = frcp(v); break;
            case 5: r = psi_eval(tab, 0, psi_arg(logt, v)); break;
            case 6: r = psi_eval(tab, 1, psi_arg(logt, v)); break;
            case 7: r = __builtin_amdgcn_rcp(v); break;  // raw v_rcp_f64
            case 8: r = __builtin_amdgcn_rsq(v); break;  // raw v_rsq_f64
            case 9: {
                double q = __builtin_amdgcn_rcp(v);
                r = __builtin_fma(q, __builtin_fma(-v, q, 1.0), q);  // one Newton step
            } break;
            default: r = 0.0;
        }
        y[k] = r;
    }
}

// =============================================================================================
// host-side launchers (called from coflux_abi.cpp)
// =============================================================================================
static SourceDesc make_source(const cf_atmos_source* s) {
    SourceDesc S;
    for (int v = 0; v < CF_JRA55_NVARS; ++v) S.data[v] = s->data[v];
    S.ns_x = s->ns_x;
    S.ns_y = s->ns_y;
    S.level1 = s->level1;
    S.level2 = s->level2;
    S.tf = s->time_fraction;
    return S;
}

static WeightDesc make_weights(const cf_interp_weights* w) {
    WeightDesc W{};
    if (w) {
        W.fi = w->fi;
        W.fj = w->fj;
        W.cos_rot = w->cos_rot;
        W.sin_rot = w->sin_rot;
        W.latitude = w->latitude;
        W.separable = w->separable;
    }
    return W;
}

static Exchange make_exchange(const cf_exchange_fields* e) {
    return Exchange{e->u, e->v, e->T, e->p, e->q, e->Qs, e->Ql, e->Mp};
}
static OceanIn make_ocean(const cf_ocean_surface* o) { return OceanIn{o->T, o->S, o->u, o->v, o->mask}; }
static FluxOut make_fluxes(const cf_interface_fluxes* f) {
    return FluxOut{f->sensible_heat, f->latent_heat,       f->water_vapor,       f->x_momentum,     f->y_momentum,
                   f->temperature,   f->friction_velocity, f->temperature_scale, f->humidity_scale, f->iterations};
}

static int tile_count(const GridDesc& G) {
    return ((G.nx + 2 * G.ring + TILE_X - 1) / TILE_X) * ((G.ny + 2 * G.ring + TILE_Y - 1) / TILE_Y);
}

// persistent grid: a multiple of 8 (one share per XCD), at most `max_blocks`
static int persistent_blocks(int work_items, int max_blocks) {
    if (work_items >= max_blocks) return max_blocks;
    if (work_items >= NUM_XCD) return (work_items / NUM_XCD) * NUM_XCD;
    return work_items > 0 ? work_items : 1;
}

hipError_t launch_interpolate(hipStream_t st, const LaunchCfg& L, const GridDesc& G, const cf_atmos_source* s,
                              const cf_interp_weights* w, const cf_exchange_fields* e) {
    size_t lds = BOX_BYTES + (size_t)NPLANES * L.interp_cap * sizeof(float);
    int blocks = persistent_blocks(tile_count(G), L.max_blocks);
    hipLaunchKernelGGL(interpolate_kernel, dim3(blocks), dim3(TILE_X, TILE_Y), lds, st, make_source(s),
                       make_weights(w), G, make_exchange(e), L.interp_cap);
    return hipGetLastError();
}

hipError_t launch_ao_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const FastConsts& C,
                            const GridDesc& G, const cf_ocean_surface* o, const cf_exchange_fields* e,
                            const cf_interface_fluxes* f) {
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    const int ncells = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    const int chunks = (ncells + AO_BLOCK - 1) / AO_BLOCK;
    if (L.solver == CF_SOLVER_LIBM) {
        dim3 grid(chunks);
        const bool coare = P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC;
#define LIBM_LAUNCH(STAB)                                                                                   \
    if (coare)                                                                                              \
        hipLaunchKernelGGL((ao_flux_libm_kernel<STAB, true>), grid, dim3(AO_BLOCK), 0, st, P, G, O, E, F);  \
    else                                                                                                    \
        hipLaunchKernelGGL((ao_flux_libm_kernel<STAB, false>), grid, dim3(AO_BLOCK), 0, st, P, G, O, E, F);
        switch (P.stability) {
            case CF_STABILITY_EDSON2013: LIBM_LAUNCH(CF_STABILITY_EDSON2013) break;
            case CF_STABILITY_SHEBA: LIBM_LAUNCH(CF_STABILITY_SHEBA) break;
            default: LIBM_LAUNCH(CF_STABILITY_LARGE_YEAGER) break;
        }
#undef LIBM_LAUNCH
        return hipGetLastError();
    }
    // one workgroup per chunk: the hardware dispatcher is the dynamic load balancer (chunks differ in
    // their wet fraction and iteration counts); the 34 KB table stage per workgroup comes from L2.
    dim3 grid(min((ncells + AO_CHUNK - 1) / AO_CHUNK, 1 << 20));
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        hipLaunchKernelGGL((ao_flux_fast_kernel<true>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, P, C, G, O, E, F, L.d_tables);
    else
        hipLaunchKernelGGL((ao_flux_fast_kernel<false>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, P, C, G, O, E, F, L.d_tables);
    return hipGetLastError();
}

hipError_t launch_fused(hipStream_t st, const LaunchCfg& L, const DevParams& P, const FastConsts& C, const GridDesc& G,
                        const cf_atmos_source* s, const cf_interp_weights* w, const cf_ocean_surface* o,
                        const cf_exchange_fields* e, const cf_interface_fluxes* f) {
    SourceDesc S = make_source(s);
    WeightDesc W = make_weights(w);
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    size_t lds = TABLE_BYTES + BOX_BYTES + (size_t)NPLANES * L.interp_cap * sizeof(float);
    dim3 grid(persistent_blocks(tile_count(G), L.max_blocks));
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        hipLaunchKernelGGL((fused_fast_kernel<true>), grid, dim3(TILE_X, TILE_Y), lds, st, P, C, S, W, G, O, E, F,
                           L.d_tables, L.interp_cap);
    else
        hipLaunchKernelGGL((fused_fast_kernel<false>), grid, dim3(TILE_X, TILE_Y), lds, st, P, C, S, W, G, O, E, F,
                           L.d_tables, L.interp_cap);
    return hipGetLastError();
}

hipError_t launch_net_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                             const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                             const cf_interp_weights* w, const cf_net_ocean_fluxes* n) {
    IceIn I{};
    if (ice) I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress};
    NetOut N{n->u, n->v, n->T, n->S, n->shortwave_surface_flux, n->upwelling_longwave, n->downwelling_longwave,
             n->downwelling_shortwave};
    const int ncells = G.nx * G.ny;
    hipLaunchKernelGGL(net_flux_kernel, dim3((ncells + NET_BLOCK - 1) / NET_BLOCK), dim3(NET_BLOCK), 0, st, P, G,
                       make_ocean(o), make_exchange(e), make_fluxes(f), I, make_weights(w), N);
    return hipGetLastError();
}

hipError_t launch_debug_eval(hipStream_t st, const LaunchCfg& L, int fn, int n, const double* x, double* y) {
    hipLaunchKernelGGL(debug_eval_kernel, dim3(64), dim3(256), TABLE_BYTES, st, fn, n, x, y, L.d_tables);
    return hipGetLastError();
}

__global__ void copy_kernel(double2* __restrict__ dst, const double2* __restrict__ src, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

hipError_t launch_copy(hipStream_t st, void* dst, const void* src, size_t bytes) {
    size_t n = bytes / sizeof(double2);
    hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, st, (double2*)dst, (const double2*)src, n);
    return hipGetLastError();
}

}  // namespace coflux
