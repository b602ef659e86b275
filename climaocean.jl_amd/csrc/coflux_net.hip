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
