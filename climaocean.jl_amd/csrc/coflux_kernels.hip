// coflux_kernels.hip — gfx950 kernels of the surface-flux path.
//
//   interpolate_kernel ....... interpolate_atmosphere_state!  (JRA55 f32 window → 8 exchange fields)
//   ao_flux_kernel ........... compute_atmosphere_ocean_fluxes! (Monin–Obukhov fixed point)
//   fused_interp_flux_kernel . the two above in one pass (update_state! path)
//   net_flux_kernel .......... compute_net_ocean_fluxes!       (radiation + partition)
//
// All are pointwise / 1-cell-stencil kernels bounded by HBM, not by MFMA: nothing here is a
// contraction.  Lanes run along i (the contiguous axis) so every field access is a coalesced
// 512-B wave transaction; the JRA55 source tile a workgroup needs is staged once through LDS.
#include <hip/hip_runtime.h>

#include "coflux_device.hpp"
#include "coflux_kernels.h"

namespace coflux {

// =============================================================================================
// JRA55 tile staging + bilinear × linear-in-time interpolation
// =============================================================================================
constexpr int TILE_X = 64;   // one wave spans 64 consecutive i
constexpr int TILE_Y = 4;    // 4 waves per workgroup, one row each
constexpr int NPLANES = 2 * CF_JRA55_NVARS;  // (variable, time level)

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
// the window is 14.7 MB and lives in L2/Infinity Cache.
__device__ __forceinline__ InterpCell interpolate_cell(const SourceDesc& S, const WeightDesc& Wt,
                                                       const GridDesc& G, int i, int j, bool in_range,
                                                       float* lds, int cap) {
    __shared__ int box[5];  // dmin, dmax, jmin, jmax, ref
    const int tid = threadIdx.y * TILE_X + threadIdx.x;

    // clamp out-of-window threads onto a valid cell so that they do not widen the footprint
    int ic = min(max(i, -G.ring), G.nx + G.ring - 1);
    int jc = min(max(j, -G.ring), G.ny + G.ring - 1);
    size_t k = cell_index(G, ic, jc);
    double fi = Wt.separable ? Wt.fi[ic + G.hx] : Wt.fi[k];
    double fj = Wt.separable ? Wt.fj[jc + G.hy] : Wt.fj[k];

    double ti = trunc(fi), tj = trunc(fj);
    double xi = fi - ti, eta = fj - tj;
    int i0 = (int)ti, j0 = (int)tj;
    int i1 = i0 + (fi > 0.0 ? 1 : (fi < 0.0 ? -1 : 0));
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
    // offsets relative to the tile's reference column, wrapped to (−ns_x/2, ns_x/2]
    int d0 = i0 - ref, d1 = i1 - ref;
    const int half = S.ns_x / 2;
    d0 = wrap_index(d0 + half, S.ns_x) - half;
    d1 = d0 + (i1 - i0);
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
            int p = e / per_plane;
            int r = e - p * per_plane;
            int y = r / W;
            int x = r - y * W;
            int var = p >> 1;
            int lev = (p & 1) ? S.level2 : S.level1;
            int is = wrap_index(ref + dmin + x, S.ns_x);
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
    (void)in_range;
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
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = (int)blockIdx.x * TILE_X + (int)threadIdx.x - G.ring;
    const int j = (int)blockIdx.y * TILE_Y + (int)threadIdx.y - G.ring;
    const bool in_range = (i < G.nx + G.ring) && (j < G.ny + G.ring);
    InterpCell c = interpolate_cell(S, Wt, G, i, j, in_range, lds, cap);
    if (in_range) {
        size_t k = cell_index(G, i, j);
        store_exchange(E, k, finish_interp(c, Wt, k));
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

template <int STAB, bool COARE>
__device__ __forceinline__ CellFluxes solve_dispatch(const DevParams& P, const AtmosCell& a, double uo, double vo,
                                                     double To, double So, bool wet, bool in_range) {
    if (P.stop_kind == CF_STOP_FIXED)
        return solve_cell<STAB, COARE, true>(P, a.u, a.v, a.T, a.p, a.q, uo, vo, To, So, wet, in_range);
    return solve_cell<STAB, COARE, false>(P, a.u, a.v, a.T, a.p, a.q, uo, vo, To, So, wet, in_range);
}

constexpr int AO_BLOCK = 256;

template <int STAB, bool COARE>
__global__ __launch_bounds__(AO_BLOCK) void ao_flux_kernel(DevParams P, GridDesc G, OceanIn O, Exchange E,
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

    AtmosCell a;
    a.u = E.u[k];
    a.v = E.v[k];
    a.T = E.T[k];
    a.p = E.p[k];
    a.q = E.q[k];
    // ℑxᶜᵃᵃ u, ℑyᵃᶜᵃ v: cell-centre ocean velocity from the two bracketing faces
    const double uo = 0.5 * (O.u[k] + O.u[k + 1]);
    const double vo = 0.5 * (O.v[k] + O.v[k + (size_t)G.sj]);
    const double To = O.T[k], So = O.S[k];
    const bool wet = cell_is_wet(P, O.mask, k);

    CellFluxes R = solve_dispatch<STAB, COARE>(P, a, uo, vo, To, So, wet, in_range);
    if (in_range) store_fluxes(F, k, R);
}

template <int STAB, bool COARE>
__global__ __launch_bounds__(TILE_X* TILE_Y) void fused_interp_flux_kernel(DevParams P, SourceDesc S, WeightDesc Wt,
                                                                            GridDesc G, OceanIn O, Exchange E,
                                                                            FluxOut F, int cap) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = (int)blockIdx.x * TILE_X + (int)threadIdx.x - G.ring;
    const int j = (int)blockIdx.y * TILE_Y + (int)threadIdx.y - G.ring;
    const bool in_range = (i < G.nx + G.ring) && (j < G.ny + G.ring);
    InterpCell c = interpolate_cell(S, Wt, G, i, j, in_range, lds, cap);
    const int ic = min(i, G.nx + G.ring - 1), jc = min(j, G.ny + G.ring - 1);
    const size_t k = cell_index(G, ic, jc);
    AtmosCell a = finish_interp(c, Wt, k);
    if (in_range) store_exchange(E, k, a);

    const double uo = 0.5 * (O.u[k] + O.u[k + 1]);
    const double vo = 0.5 * (O.v[k] + O.v[k + (size_t)G.sj]);
    const double To = O.T[k], So = O.S[k];
    const bool wet = cell_is_wet(P, O.mask, k);
    CellFluxes R = solve_dispatch<STAB, COARE>(P, a, uo, vo, To, So, wet, in_range);
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

static dim3 tile_grid(const GridDesc& G) {
    return dim3((G.nx + 2 * G.ring + TILE_X - 1) / TILE_X, (G.ny + 2 * G.ring + TILE_Y - 1) / TILE_Y);
}

hipError_t launch_interpolate(hipStream_t st, const GridDesc& G, const cf_atmos_source* s, const cf_interp_weights* w,
                              const cf_exchange_fields* e, int cap) {
    size_t lds = (size_t)NPLANES * cap * sizeof(float);
    hipLaunchKernelGGL(interpolate_kernel, tile_grid(G), dim3(TILE_X, TILE_Y), lds, st, make_source(s),
                       make_weights(w), G, make_exchange(e), cap);
    return hipGetLastError();
}

template <int STAB>
static void launch_ao_t(hipStream_t st, const DevParams& P, const GridDesc& G, const OceanIn& O, const Exchange& E,
                        const FluxOut& F) {
    const int ncells = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    dim3 grid((ncells + AO_BLOCK - 1) / AO_BLOCK);
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        hipLaunchKernelGGL((ao_flux_kernel<STAB, true>), grid, dim3(AO_BLOCK), 0, st, P, G, O, E, F);
    else
        hipLaunchKernelGGL((ao_flux_kernel<STAB, false>), grid, dim3(AO_BLOCK), 0, st, P, G, O, E, F);
}

hipError_t launch_ao_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f) {
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    switch (P.stability) {
        case CF_STABILITY_EDSON2013: launch_ao_t<CF_STABILITY_EDSON2013>(st, P, G, O, E, F); break;
        case CF_STABILITY_SHEBA: launch_ao_t<CF_STABILITY_SHEBA>(st, P, G, O, E, F); break;
        default: launch_ao_t<CF_STABILITY_LARGE_YEAGER>(st, P, G, O, E, F); break;
    }
    return hipGetLastError();
}

template <int STAB>
static void launch_fused_t(hipStream_t st, const DevParams& P, const GridDesc& G, const SourceDesc& S,
                           const WeightDesc& W, const OceanIn& O, const Exchange& E, const FluxOut& F, int cap) {
    size_t lds = (size_t)NPLANES * cap * sizeof(float);
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        hipLaunchKernelGGL((fused_interp_flux_kernel<STAB, true>), tile_grid(G), dim3(TILE_X, TILE_Y), lds, st, P, S,
                           W, G, O, E, F, cap);
    else
        hipLaunchKernelGGL((fused_interp_flux_kernel<STAB, false>), tile_grid(G), dim3(TILE_X, TILE_Y), lds, st, P, S,
                           W, G, O, E, F, cap);
}

hipError_t launch_fused(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_atmos_source* s,
                        const cf_interp_weights* w, const cf_ocean_surface* o, const cf_exchange_fields* e,
                        const cf_interface_fluxes* f, int cap) {
    SourceDesc S = make_source(s);
    WeightDesc W = make_weights(w);
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    switch (P.stability) {
        case CF_STABILITY_EDSON2013: launch_fused_t<CF_STABILITY_EDSON2013>(st, P, G, S, W, O, E, F, cap); break;
        case CF_STABILITY_SHEBA: launch_fused_t<CF_STABILITY_SHEBA>(st, P, G, S, W, O, E, F, cap); break;
        default: launch_fused_t<CF_STABILITY_LARGE_YEAGER>(st, P, G, S, W, O, E, F, cap); break;
    }
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
