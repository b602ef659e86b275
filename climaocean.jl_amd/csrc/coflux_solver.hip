// coflux_solver.hip — compute_atmosphere_ocean_fluxes! on gfx950: the Monin–Obukhov fixed point.
//
// One lane = one ocean cell.  The kernel is FP64-issue bound (≈ 250–480 VALU instructions per
// iteration × 10–20 iterations per cell against 128 algorithmic bytes), so the design goal is to
// waste no issue slot: ψ/log tables in LDS (coflux_fast.hpp), land compacted away before the
// iteration, waves leaving the loop on a wave64 ballot, one workgroup per 512-cell chunk so that
// the hardware dispatcher balances chunks of different wet fraction and trip count.
#include <hip/hip_runtime.h>

#include "coflux_fast.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

constexpr int TABLE_BYTES = TABLE_DOUBLES * 8;

constexpr int AO_BLOCK = 256;
constexpr int AO_CHUNK = 512;  // cells classified + compacted per pass of a workgroup
constexpr int AO_BINS = 32;     // trip-count bins of the per-chunk counting sort
constexpr int AO_PARAMS_OFFSET = TABLE_BYTES + AO_CHUNK * 4 + 16 + 2 * AO_BINS * 4;
constexpr int AO_LDS_BYTES = AO_PARAMS_OFFSET + (int)sizeof(DevParams);

// ---- production solver: LDS tables, persistent workgroups ------------------------------------
template <bool COARE, int SPEC>
__global__ __launch_bounds__(AO_BLOCK) void ao_flux_fast_kernel(LoopParams L, GridDesc G, OceanIn O, Exchange E,
                                                                FluxOut F, const double* __restrict__ g_tab,
                                                                const DevParams* __restrict__ g_params,
                                                                uint8_t* __restrict__ hint) {
    // Land cells (≈30 % of a global grid) must not occupy lanes for 10–20 iterations: every chunk of
    // AO_CHUNK cells is first compacted to the list of its wet cells (land gets its zeros there and
    // then), and waves then pull 64 list entries at a time from an LDS cursor, so every lane that
    // enters the solver holds an ocean cell.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    int* list = reinterpret_cast<int*>(smem + TABLE_BYTES);
    int* counters = list + AO_CHUNK;  // [0] wet count, [1] cursor
    int* hist = counters + 4;
    int* bin_start = hist + AO_BINS;
    DevParams* lp = reinterpret_cast<DevParams*>(smem + AO_PARAMS_OFFSET);
    const int tid = threadIdx.x, lane = tid & 63;
    stage_tables(tab, g_tab, tid, AO_BLOCK);
    for (int n = tid; n < (int)(sizeof(DevParams) / sizeof(double)); n += AO_BLOCK)
        reinterpret_cast<double*>(lp)[n] = reinterpret_cast<const double*>(g_params)[n];
    const DevParams& P = *lp;  // prologue-only parameters live in LDS, not in SGPRs
    const double* logt = tab + 4 * PSI_TABLE;

    const int wx = G.nx + 2 * G.ring;
    const int ncells = wx * (G.ny + 2 * G.ring);
    const int nchunks = (ncells + AO_CHUNK - 1) / AO_CHUNK;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        if (tid < 2) counters[tid] = 0;
        if (tid < AO_BINS) hist[tid] = 0;
        __syncthreads();
        // ---- phase 1: classify, zero land, counting-sort the wet cells by their trip-count hint ----
        // (the hint is the cell's iteration count in the previous call — fields evolve slowly from one
        // coupled step to the next — so lanes of a batch finish together; it only orders the list and
        // cannot change any result)
        const int begin = chunk * AO_CHUNK, end = min(begin + AO_CHUNK, ncells);
        constexpr int PER = AO_CHUNK / AO_BLOCK;
        int my_idx[PER], my_bin[PER], my_rank[PER];
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int idx = begin + p * AO_BLOCK + tid;
            my_idx[p] = idx;
            my_bin[p] = -1;
            my_rank[p] = 0;
            if (idx < end) {
                const int jj = idx / wx;
                const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
                if (cell_is_wet(P, O.mask, k)) {
                    const int bin = hint ? AO_BINS - 1 - min((int)hint[k], AO_BINS - 1) : 0;  // longest first (LPT)
                    my_bin[p] = bin;
                    my_rank[p] = atomicAdd(&hist[bin], 1);
                } else {  // zero_interface_state: all fluxes 0, T = 0 K
                    CellFluxes Z{};
                    Z.Ts_ocean = -P.T_offset;
                    Z.iterations = L.fixed ? L.maxiter : 0;
                    store_fluxes(F, k, Z);
                }
            }
        }
        __syncthreads();
        if (tid < 64) {  // exclusive scan of the AO_BINS (≤ 64) bin counts by one wave
            const int v = lane < AO_BINS ? hist[lane] : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            if (lane < AO_BINS) bin_start[lane] = incl - v;
            if (lane == 63) counters[0] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PER; ++p)
            if (my_bin[p] >= 0) list[bin_start[my_bin[p]] + my_rank[p]] = my_idx[p];
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
            const CellConsts c = cell_prologue(P, L.min_gust, logt, E.u[k], E.v[k], E.T[k], E.p[k], E.q[k], uo, vo,
                                               O.T[k], O.S[k]);
            Scales s;
            if constexpr (SPEC == SOLVER_LY)
                s = ly_iterate(L, c, tab);
            else
                s = mo_iterate<COARE, SPEC>(L, c, tab, in_range);
            if (in_range) {
                store_fluxes(F, k, cell_epilogue(c, P.T_offset, s));
                if (hint) hint[k] = (uint8_t)min(s.it, 255);
            }
        }
        __syncthreads();  // list and counters are reused by the next chunk
    }
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


template <bool COARE>
static void launch_ao_spec(hipStream_t st, dim3 grid, const LaunchCfg& L, const LoopParams& C, const GridDesc& G,
                           const OceanIn& O, const Exchange& E, const FluxOut& F) {
    switch (C.specialization) {
        case SOLVER_OCEAN:
            hipLaunchKernelGGL((ao_flux_fast_kernel<COARE, SOLVER_OCEAN>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G, O,
                               E, F, L.d_tables, L.d_params, L.d_hint);
            break;
        case SOLVER_ICE:
            hipLaunchKernelGGL((ao_flux_fast_kernel<COARE, SOLVER_ICE>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G, O, E,
                               F, L.d_tables, L.d_params, L.d_hint);
            break;
        case SOLVER_LY:
            hipLaunchKernelGGL((ao_flux_fast_kernel<true, SOLVER_LY>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G, O, E,
                               F, L.d_tables, L.d_params, L.d_hint);
            break;
        default:
            hipLaunchKernelGGL((ao_flux_fast_kernel<COARE, SOLVER_GENERIC>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G,
                               O, E, F, L.d_tables, L.d_params, L.d_hint);
    }
}

hipError_t launch_ao_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C,
                            const GridDesc& G, const cf_ocean_surface* o, const cf_exchange_fields* e,
                            const cf_interface_fluxes* f) {
    if (L.solver == CF_SOLVER_LIBM) return launch_ao_fluxes_libm(st, P, G, o, e, f);
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    const int ncells = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    // one workgroup per chunk: the hardware dispatcher is the dynamic load balancer (chunks differ in
    // their wet fraction and iteration counts); the 35 KB table/parameter stage per workgroup comes from L2.
    dim3 grid(min((ncells + AO_CHUNK - 1) / AO_CHUNK, 1 << 20));
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        launch_ao_spec<true>(st, grid, L, C, G, O, E, F);
    else
        launch_ao_spec<false>(st, grid, L, C, G, O, E, F);
    return hipGetLastError();
}

hipError_t launch_debug_eval(hipStream_t st, const LaunchCfg& L, int fn, int n, const double* x, double* y) {
    hipLaunchKernelGGL(debug_eval_kernel, dim3(64), dim3(256), TABLE_BYTES, st, fn, n, x, y, L.d_tables);
    return hipGetLastError();
}

}  // namespace coflux
