// coflux_net.hip — compute_net_ocean_fluxes! on gfx950 (radiation + (1 − ℵ) partition + unit
// conversion), plus the device-copy kernel that calibrates the HBM denominator.
#include <hip/hip_runtime.h>

#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

// =============================================================================================
// Net ocean fluxes: radiation + (1 − ℵ) partition + unit conversion
// =============================================================================================
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

    double alb = P.albedo;
    if (P.albedo_kind == CF_ALBEDO_LATITUDE_DEPENDENT) {
        double phi = Wt.separable ? Wt.latitude[j + G.hy] : Wt.latitude[k];
        alb = P.albedo_diffuse - P.albedo_direct * cos(2.0 * phi * (CF_PI / 180.0));
    }
    const NetCell C = net_cell_local(P, alb, aice, O.S[k], F.Ts[k] + P.T_offset, E.Mp[k], E.Qs[k], E.Ql[k], F.Qc[k], F.Qv[k],
                                     F.Fv[k], I.Qio ? I.Qio[k] : 0.0, I.Jsio ? I.Jsio[k] : 0.0, I.land ? I.land[k] : 0.0);
    const double tx = net_face_stress(P, F.tx[kw], F.tx[k], aice_w, aice, I.txio ? I.txio[k] : 0.0);
    const double ty = net_face_stress(P, F.ty[ks], F.ty[k], aice_s, aice, I.tyio ? I.tyio[k] : 0.0);
    NetCell Z{};
    N.u[k] = wet ? tx : 0.0;
    N.v[k] = wet ? ty : 0.0;
    store_net_cell(N, k, wet ? C : Z);
}

// The stresses alone: what is left of compute_net_ocean_fluxes! when the solver's epilogue has already written the
// cell-local fluxes (cf_update_state's fused path).  Reads ρτ of the cell and of its west / south neighbour.
__global__ __launch_bounds__(NET_BLOCK) void net_stress_kernel(DevParams P, GridDesc G, const void* mask,
                                                               const double* __restrict__ rtx, const double* __restrict__ rty,
                                                               IceIn I, double* __restrict__ tau_x, double* __restrict__ tau_y) {
    const int ncells = G.nx * G.ny;
    const int idx = (int)blockIdx.x * NET_BLOCK + (int)threadIdx.x;
    if (idx >= ncells) return;
    const int j = idx / G.nx;
    const size_t k = cell_index(G, idx - j * G.nx, j);
    const size_t kw = k - 1, ks = k - (size_t)G.sj;
    const bool wet = cell_is_wet(P, mask, k);
    const double aice = I.conc ? I.conc[k] : 0.0;
    const double tx = net_face_stress(P, rtx[kw], rtx[k], I.conc ? I.conc[kw] : 0.0, aice, I.txio ? I.txio[k] : 0.0);
    const double ty = net_face_stress(P, rty[ks], rty[k], I.conc ? I.conc[ks] : 0.0, aice, I.tyio ? I.tyio[k] : 0.0);
    tau_x[k] = wet ? tx : 0.0;
    tau_y[k] = wet ? ty : 0.0;
}

hipError_t launch_net_stress(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                             const cf_interface_fluxes* f, const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* n) {
    IceIn I{};
    if (ice) I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress, nullptr};
    const int ncells = G.nx * G.ny;
    hipLaunchKernelGGL(net_stress_kernel, dim3((ncells + NET_BLOCK - 1) / NET_BLOCK), dim3(NET_BLOCK), 0, st, P, G, o->mask,
                       f->x_momentum, f->y_momentum, I, n->u, n->v);
    return hipGetLastError();
}

// SurfaceFluxRestoring materialised: J_add = v_p (Sₒ − S★) on wet interior cells, 0 elsewhere
__global__ __launch_bounds__(NET_BLOCK) void salinity_restoring_kernel(DevParams P, GridDesc G, const void* mask, double vp,
                                                                       const double* __restrict__ target, const double* __restrict__ S,
                                                                       double* __restrict__ out) {
    const int idx = (int)blockIdx.x * NET_BLOCK + (int)threadIdx.x;
    if (idx >= G.nx * G.ny) return;
    const int j = idx / G.nx;
    const size_t k = cell_index(G, idx - j * G.nx, j);
    out[k] = cell_is_wet(P, mask, k) ? vp * (S[k] - target[k]) : 0.0;
}

hipError_t launch_salinity_restoring(hipStream_t st, const DevParams& P, const GridDesc& G, const void* mask, double vp,
                                     const double* target, const double* S, double* out) {
    const int ncells = G.nx * G.ny;
    hipLaunchKernelGGL(salinity_restoring_kernel, dim3((ncells + NET_BLOCK - 1) / NET_BLOCK), dim3(NET_BLOCK), 0, st, P, G, mask, vp,
                       target, S, out);
    return hipGetLastError();
}

hipError_t launch_net_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                             const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                             const cf_interp_weights* w, const cf_net_ocean_fluxes* n, const double* land) {
    IceIn I{};
    if (ice) I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress, nullptr};
    I.land = land;
    NetOut N{n->u, n->v, n->T, n->S, n->shortwave_surface_flux, n->upwelling_longwave, n->downwelling_longwave,
             n->downwelling_shortwave};
    const int ncells = G.nx * G.ny;
    hipLaunchKernelGGL(net_flux_kernel, dim3((ncells + NET_BLOCK - 1) / NET_BLOCK), dim3(NET_BLOCK), 0, st, P, G,
                       make_ocean(o), make_exchange(e), make_fluxes(f), I, make_weights(w), N);
    return hipGetLastError();
}

// =============================================================================================
// compute_net_sea_ice_fluxes!: top and bottom heat fluxes of the sea ice (pointwise, HBM bound)
// =============================================================================================
__global__ __launch_bounds__(NET_BLOCK) void net_sea_ice_flux_kernel(DevParams P, GridDesc G, const void* mask,
                                                                     const double* __restrict__ conc,
                                                                     const double* __restrict__ albedo_field, double albedo,
                                                                     double emissivity, double eps_sigma, double T_offset,
                                                                     const double* __restrict__ Qs, const double* __restrict__ Ql,
                                                                     const double* __restrict__ Qc, const double* __restrict__ Qv,
                                                                     const double* __restrict__ Ts,
                                                                     const double* __restrict__ Qf, const double* __restrict__ Qi,
                                                                     double* __restrict__ top, double* __restrict__ bottom) {
    const int ncells = G.nx * G.ny;
    const int idx = (int)blockIdx.x * NET_BLOCK + (int)threadIdx.x;
    if (idx >= ncells) return;
    const int j = idx / G.nx;
    const size_t k = cell_index(G, idx - j * G.nx, j);
    double sum_top = 0.0, sum_bottom = 0.0;
    if (cell_is_wet(P, mask, k)) {
        net_sea_ice_cell(albedo_field ? albedo_field[k] : albedo, emissivity, eps_sigma, T_offset, Qs[k], Ql[k], Ts[k], Qc[k], Qv[k], conc[k],
                         Qf ? Qf[k] : 0.0, Qi ? Qi[k] : 0.0, sum_top, sum_bottom);
    }
    top[k] = sum_top;
    bottom[k] = sum_bottom;
}

hipError_t launch_net_sea_ice_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const void* mask,
                                     const cf_sea_ice_state* ice, double albedo, double emissivity, double eps_sigma,
                                     double T_offset, const cf_exchange_fields* e, const cf_interface_fluxes* f,
                                     const double* frazil, const double* interface_heat, const cf_net_sea_ice_fluxes* out) {
    const int ncells = G.nx * G.ny;
    hipLaunchKernelGGL(net_sea_ice_flux_kernel, dim3((ncells + NET_BLOCK - 1) / NET_BLOCK), dim3(NET_BLOCK), 0, st, P, G, mask,
                       ice->concentration, ice->albedo, albedo, emissivity, eps_sigma, T_offset, e->Qs, e->Ql,
                       f->sensible_heat, f->latent_heat, f->temperature, frazil, interface_heat, out->top_heat,
                       out->bottom_heat);
    return hipGetLastError();
}

// =============================================================================================
// SeaIceAlbedo(hi, hs, Ts): the CCSM3 albedo field (pointwise)
// =============================================================================================
__global__ __launch_bounds__(NET_BLOCK) void sea_ice_albedo_kernel(cf_sea_ice_albedo_params A, size_t n, const double* __restrict__ hi,
                                                                   const double* __restrict__ hs, const double* __restrict__ Ts,
                                                                   double* __restrict__ out) {
    const size_t k = (size_t)blockIdx.x * NET_BLOCK + threadIdx.x;
    if (k < n) out[k] = ccsm3_albedo(A, hi[k], hs ? hs[k] : 0.0, Ts[k]);
}

hipError_t launch_sea_ice_albedo(hipStream_t st, const cf_sea_ice_albedo_params& A, const GridDesc& G, const double* hi,
                                 const double* hs, const double* Ts, double* out) {
    const size_t n = (size_t)G.sj * (G.ny + 2 * G.hy);
    hipLaunchKernelGGL(sea_ice_albedo_kernel, dim3((unsigned)((n + NET_BLOCK - 1) / NET_BLOCK)), dim3(NET_BLOCK), 0, st, A, n, hi, hs,
                       Ts, out);
    return hipGetLastError();
}

// =============================================================================================
// compute_sea_ice_ocean_fluxes!: ThreeEquationHeatFlux with momentum-based friction velocity + frazil (pointwise)
// =============================================================================================
__global__ __launch_bounds__(NET_BLOCK) void sea_ice_ocean_flux_kernel(DevParams P, cf_ice_ocean_params Q, GridDesc G, OceanIn O,
                                                                       const double* __restrict__ conc,
                                                                       const double* __restrict__ tx, const double* __restrict__ ty,
                                                                       double* __restrict__ Qio, double* __restrict__ Jsio,
                                                                       double* __restrict__ Qfr, double* __restrict__ ustar_out) {
    const int ncells = G.nx * G.ny;
    const int idx = (int)blockIdx.x * NET_BLOCK + (int)threadIdx.x;
    if (idx >= ncells) return;
    const int j = idx / G.nx;
    const size_t k = cell_index(G, idx - j * G.nx, j);
    double q_io = 0.0, j_io = 0.0, q_fr = 0.0, us = 0.0;
    if (cell_is_wet(P, O.mask, k)) {
        const double rho_o = 1.0 / P.rho_o_inv, c_o = 1.0 / P.c_o_inv;
        const double So = O.S[k];
        double To = O.T[k];
        const double Tf = -Q.liquidus_slope * So;
        if (Q.time_step > 0.0 && To < Tf) {  // frazil: the deficit below freezing is handed to the ice model
            q_fr = rho_o * c_o * Q.top_cell_thickness * (To - Tf) / Q.time_step;
            To = Tf;
        }
        const double a = conc ? conc[k] : 0.0;
        if (a > 0.0) {
            const double txc = tx ? 0.5 * (tx[k] + tx[k + 1]) : 0.0, tyc = ty ? 0.5 * (ty[k] + ty[k + (size_t)G.sj]) : 0.0;
            us = fmax(sqrt(sqrt(txc * txc + tyc * tyc)), Q.minimum_friction_velocity);
            // α_s (S_o − S_b) = (c_o α_h / ℒ)(T_o + m S_b)(S_b − S_i): the positive root of A S_b² + B S_b − C = 0
            const double ah = Q.heat_transfer_coefficient, as = Q.salt_transfer_coefficient, m = Q.liquidus_slope;
            const double g = c_o * ah / Q.latent_heat_of_fusion;
            const double A = g * m, B = g * To - g * m * Q.ice_salinity + as, C = g * To * Q.ice_salinity + as * So;
            const double Sb = (-B + sqrt(B * B + 4.0 * A * C)) / (2.0 * A);
            const double Tb = -m * Sb;
            q_io = a * rho_o * c_o * ah * us * (To - Tb);
            j_io = a * as * us * (So - Sb);
        }
    }
    Qio[k] = q_io;
    Jsio[k] = j_io;
    if (Qfr) Qfr[k] = q_fr;
    if (ustar_out) ustar_out[k] = us;
}

hipError_t launch_sea_ice_ocean_fluxes(hipStream_t st, const DevParams& P, const cf_ice_ocean_params& Q, const GridDesc& G,
                                       const cf_ocean_surface* o, const double* conc, const double* tx, const double* ty,
                                       const cf_ice_ocean_fluxes* out) {
    const int ncells = G.nx * G.ny;
    hipLaunchKernelGGL(sea_ice_ocean_flux_kernel, dim3((ncells + NET_BLOCK - 1) / NET_BLOCK), dim3(NET_BLOCK), 0, st, P, Q, G,
                       make_ocean(o), conc, tx, ty, out->interface_heat, out->salt_flux, out->frazil_heat, out->friction_velocity);
    return hipGetLastError();
}

// =============================================================================================
// NormalizeSalinity: area-weighted mean over wet interior cells, then subtract from the whole parent
// =============================================================================================
constexpr int RED_BLOCK = 256;

// stage 1: per-workgroup partial sums (Σ J·A, Σ A) in a fixed order ⇒ bitwise reproducible
__global__ __launch_bounds__(RED_BLOCK) void salinity_partial_sums_kernel(DevParams P, GridDesc G, const double* __restrict__ flux,
                                                                           const double* __restrict__ additional,
                                                                           const double* __restrict__ area, const void* mask,
                                                                           double2* __restrict__ partial) {
    __shared__ double2 red[RED_BLOCK / 64];
    const int ncells = G.nx * G.ny;
    double sj = 0.0, sa = 0.0;
    for (int idx = (int)blockIdx.x * RED_BLOCK + (int)threadIdx.x; idx < ncells; idx += (int)gridDim.x * RED_BLOCK) {
        const int j = idx / G.nx, i = idx - j * G.nx;
        const size_t k = cell_index(G, i, j);
        if (cell_is_wet(P, mask, k)) {
            const double a = area ? area[k] : 1.0;
            const double v = flux[k] + (additional ? additional[k] : 0.0);
            sj = __builtin_fma(v, a, sj);
            sa += a;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sj += __shfl_xor(sj, m);
        sa += __shfl_xor(sa, m);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_double2(sj, sa);
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 t = red[0];
        for (int w = 1; w < RED_BLOCK / 64; ++w) {
            t.x += red[w].x;
            t.y += red[w].y;
        }
        partial[blockIdx.x] = t;
    }
}

// stage 2: one wave adds the partials in index order → sums[0] = Σ J·A, sums[1] = Σ A
__global__ void salinity_final_sums_kernel(const double2* __restrict__ partial, int n, double* __restrict__ sums) {
    double sj = 0.0, sa = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) {
        sj += partial[k].x;
        sa += partial[k].y;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sj += __shfl_xor(sj, m);
        sa += __shfl_xor(sa, m);
    }
    if (threadIdx.x == 0) {
        sums[0] = sj;
        sums[1] = sa;
    }
}

// stage 3: parent(flux) .-= mean over the whole parent array (halos and land included)
__global__ void salinity_subtract_kernel(double* __restrict__ flux, size_t n, const double* __restrict__ sums,
                                         double* __restrict__ mean_out) {
    const double mean = sums[1] > 0.0 ? sums[0] / sums[1] : 0.0;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) flux[k] -= mean;
    if (mean_out && blockIdx.x == 0 && threadIdx.x == 0) *mean_out = mean;
}

hipError_t launch_salinity_partial_sums(hipStream_t st, const DevParams& P, const GridDesc& G, const double* flux,
                                        const double* additional, const double* area, const void* mask, double* partial,
                                        int nblocks, double* sums) {
    hipLaunchKernelGGL(salinity_partial_sums_kernel, dim3(nblocks), dim3(RED_BLOCK), 0, st, P, G, flux, additional, area, mask,
                       reinterpret_cast<double2*>(partial));
    hipLaunchKernelGGL(salinity_final_sums_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<const double2*>(partial), nblocks,
                       sums);
    return hipGetLastError();
}

hipError_t launch_salinity_subtract(hipStream_t st, const GridDesc& G, double* flux, const double* sums, double* mean_out) {
    const size_t n = (size_t)G.sj * (G.ny + 2 * G.hy);
    hipLaunchKernelGGL(salinity_subtract_kernel, dim3(512), dim3(256), 0, st, flux, n, sums, mean_out);
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
