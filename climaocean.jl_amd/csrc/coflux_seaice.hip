// coflux_seaice.hip — compute_atmosphere_sea_ice_fluxes! on gfx950: the
// Monin–Obukhov iteration with a skin temperature inside the loop (coflux_fast.hpp::ice_iterate).
// Same workgroup structure as the ocean solver: tables / parameters in LDS, wet cells of a chunk
// compacted into a list, waves pull 64 entries at a time and leave the loop on a ballot.
#include <hip/hip_runtime.h>

#include "coflux_fast.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

constexpr int TABLE_BYTES = TABLE_DOUBLES * 8;
constexpr int AI_BLOCK = 256;
constexpr int AI_CHUNK = 512;
constexpr int AI_PARAMS_OFFSET = TABLE_BYTES + AI_CHUNK * 4 + 16;
constexpr int AI_LDS_BYTES = AI_PARAMS_OFFSET + (int)sizeof(DevParams);
static_assert(AI_LDS_BYTES <= 53760, "three workgroups must fit the CU's 160 KB of LDS");

struct IceStateIn {
    const double* thickness;
    const double* top_temperature;
    const double* u;
    const double* v;
    const double* albedo;
};

template <bool COARE>
__global__ __launch_bounds__(AI_BLOCK) void ai_flux_kernel(LoopParams L, IceParams I, GridDesc G, OceanIn O, IceStateIn S,
                                                           Exchange E, FluxOut F, const double* __restrict__ g_tab,
                                                           const DevParams* __restrict__ g_params) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    int* list = reinterpret_cast<int*>(smem + TABLE_BYTES);
    int* counters = list + AI_CHUNK;
    DevParams* lp = reinterpret_cast<DevParams*>(smem + AI_PARAMS_OFFSET);
    const int tid = threadIdx.x, lane = tid & 63;
    stage_tables(tab, g_tab, tid, AI_BLOCK);
    for (int n = tid; n < (int)(sizeof(DevParams) / sizeof(double)); n += AI_BLOCK)
        reinterpret_cast<double*>(lp)[n] = reinterpret_cast<const double*>(g_params)[n];
    const DevParams& P = *lp;
    const double* logt = tab + LOG_OFFSET;

    const int wx = G.nx + 2 * G.ring;
    const int ncells = wx * (G.ny + 2 * G.ring);
    const int nchunks = (ncells + AI_CHUNK - 1) / AI_CHUNK;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (tid < 2) counters[tid] = 0;
        __syncthreads();
        const int begin = chunk * AI_CHUNK, end = min(begin + AI_CHUNK, ncells);
        for (int base = begin; base < end; base += AI_BLOCK) {
            const int idx = base + tid;
            bool wet = false;
            if (idx < end) {
                const int jj = idx / wx;
                const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
                wet = cell_is_wet(P, O.mask, k);
                if (!wet) {  // zero_interface_state
                    CellFluxes Z{};
                    Z.Ts_ocean = -I.T_offset;
                    Z.iterations = L.fixed ? L.maxiter : 0;
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

            // ---- iteration-invariant state ---------------------------------------------------------
            const double ua = E.u[k], va = E.v[k], Ta = E.T[k], pa = E.p[k], qa = E.q[k];
            const double inv_Ta = frcp(Ta);
            const double lam_a = liquid_fraction_fast(P, logt, Ta);
            const AirState A = air_state_fast(P, pa, Ta, inv_Ta, qa, lam_a, svp_equil_fast(P, logt, Ta, inv_Ta, lam_a));
            IceConsts c;
            c.rho = A.rho;
            c.cp = A.cp_m;
            c.qav = A.q_vap;
            c.Ls = P.LH_s0 + (P.cp_v - P.cp_i) * (Ta - P.T_0);
            c.Ti = I.T_fw - I.liquidus_slope * O.S[k];
            c.hk = fmax(S.thickness[k] * I.inv_k, I.hk_min);
            const double alb = S.albedo ? S.albedo[k] : I.albedo;
            c.Qd = -(1.0 - alb) * E.Qs[k] - I.emissivity * E.Ql[k];
            c.theta_a = Ta + P.g * P.h_ref * frcp(A.cp_m);
            c.pa = pa;
            c.du = ua;
            c.dv = va;
            if (P.velocity_difference == CF_VELOCITY_RELATIVE) {
                c.du = ua - (S.u ? S.u[k] : 0.0);
                c.dv = va - (S.v ? S.v[k] : 0.0);
            }
            c.dU2 = c.du * c.du + c.dv * c.dv;
            c.dU = fsqrt(c.dU2);
            double alpha = P.rm.charnock;
            if (P.rm.kind == CF_ROUGHNESS_WIND_CHARNOCK)
                alpha = fmax(P.rm.charnock, P.rm.wind_a1 * fmin(c.dU, P.rm.wind_umax) + P.rm.wind_a2);
            c.alpha_g = alpha * P.inv_g;
            double Ts = S.top_temperature[k] + I.T_offset;

            const Scales s = ice_iterate<COARE>(P, L, I, c, tab, in_range, Ts);
            if (in_range) {
                CellFluxes R;
                const double inv_dU = (c.dU == 0.0) ? 0.0 : frcp(c.dU);
                const double tau = -s.us * s.us * inv_dU;
                const double rho_u = c.rho * s.us;
                R.Fv = -rho_u * s.qq;
                R.Qv = R.Fv * c.Ls;
                R.Qc = -rho_u * c.cp * s.ts;
                R.rho_tau_x = c.rho * tau * c.du;
                R.rho_tau_y = c.rho * tau * c.dv;
                R.Ts_ocean = Ts - I.T_offset;
                R.ustar = s.us;
                R.tstar = s.ts;
                R.qstar = s.qq;
                R.iterations = s.it;
                store_fluxes(F, k, R);
            }
        }
        __syncthreads();
    }
}

hipError_t launch_ai_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const IceParams& I,
                            const GridDesc& G, const cf_sea_ice_state* ice, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f, const double* d_tables,
                            const DevParams* d_params) {
    IceStateIn S{ice->thickness, ice->top_temperature, ice->u, ice->v, ice->albedo};
    const int ncells = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    dim3 grid(min((ncells + AI_CHUNK - 1) / AI_CHUNK, 1 << 20));
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        hipLaunchKernelGGL((ai_flux_kernel<true>), grid, dim3(AI_BLOCK), AI_LDS_BYTES, st, C, I, G, make_ocean(o), S,
                           make_exchange(e), make_fluxes(f), d_tables, d_params);
    else
        hipLaunchKernelGGL((ai_flux_kernel<false>), grid, dim3(AI_BLOCK), AI_LDS_BYTES, st, C, I, G, make_ocean(o), S,
                           make_exchange(e), make_fluxes(f), d_tables, d_params);
    (void)L;
    return hipGetLastError();
}

}  // namespace coflux
