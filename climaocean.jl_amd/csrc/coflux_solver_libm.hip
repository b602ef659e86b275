// coflux_solver_libm.hip — the same Monin–Obukhov iteration on ocml's libm (CF_SOLVER_LIBM).
// ≈ 5× slower than the table solver; kept as an on-device cross-check of the fast primitives.
#include <hip/hip_runtime.h>

#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

constexpr int AO_BLOCK = 256;

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


hipError_t launch_ao_fluxes_libm(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                                 const cf_exchange_fields* e, const cf_interface_fluxes* f) {
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    const int ncells = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    dim3 grid((ncells + AO_BLOCK - 1) / AO_BLOCK);
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

}  // namespace coflux
