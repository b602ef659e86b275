// coflux_solver.hip — compute_atmosphere_ocean_fluxes! on gfx950: the Monin–Obukhov fixed point.
//
// One lane = one ocean cell.  The kernel is FP64-issue bound (≈ 130–200 VALU instructions per
// iteration × 10–20 iterations per cell against 128 algorithmic bytes), so the design goal is to
// waste no issue slot: ψ/log tables in LDS (coflux_fast.hpp), land compacted away before the
// iteration, batches of cells with equal trip counts, waves leaving the loop on a wave64 ballot,
// one workgroup per chunk of a table that gives every workgroup whole batches and fills the
// device's resident-workgroup slots in whole rounds.
#include <hip/hip_runtime.h>

#include "coflux_fast.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

constexpr int TABLE_BYTES = TABLE_DOUBLES * 8;

constexpr int AO_BLOCK = 256;
constexpr int AO_CHUNK = 1024;  // capacity of a workgroup's wet-cell list = the most wet cells a chunk can hold
constexpr int AO_BINS = 32;    // trip-count bins of the per-chunk counting sort
constexpr int AO_PARAMS_OFFSET = TABLE_BYTES + AO_CHUNK * 4 + 16 + 2 * AO_BINS * 4;
constexpr int AO_LDS_BYTES = AO_PARAMS_OFFSET + (int)sizeof(DevParams);
static_assert(AO_LDS_BYTES <= 53760, "three solver workgroups must fit the CU's 160 KB of LDS");

// ---------------------------------------------------------------------------------------------
// Chunk table.  Two quantisation effects cost ≈ 25 % each when every workgroup simply takes 512 surface
// cells: (i) a workgroup's four waves pull 64-cell batches of wet cells, so a chunk is only used well if it
// holds a multiple of 256 wet cells (≈ 370 wet cells = six batches leave two waves — and, with the same
// wave→SIMD placement in every workgroup, two SIMDs of the CU — idle for the second half); (ii) the device
// holds 3 workgroups per CU, and a dispatch "round" that is only half full runs at half throughput.
// The wet mask is static, so the surface is cut ONCE per mask into chunks of prescribed cost with
// wet = AO_WET_COST, land = 1.  The host picks a descending sequence of rounds — each a full set of
// 3·CU chunks of 768, 512 or 256 wet cells, the remainder as a last partial round of the smallest size —
// so that the long workgroups start first and whatever tail is left is short.  An open-ocean chunk holds
// exactly its nominal wet count, a coastal one slightly fewer, a land chunk at most 64× as many cells (it
// only writes zeros).  The table only steers scheduling: the solver re-classifies every cell of its range
// on every call and falls back to smaller pieces if a range holds more wet cells than the list (a mask
// changed in place), so a stale table can cost time, never correctness.
// ---------------------------------------------------------------------------------------------
constexpr int AO_WET_COST = 64;
constexpr int AO_MAX_ROUNDS = 8;
struct ChunkRounds {  // round r covers cost prefixes [base[r], base[r+1]) in chunks of cost[r], ids from first[r]
    int n;
    int base[AO_MAX_ROUNDS + 1];
    int cost[AO_MAX_ROUNDS];
    int first[AO_MAX_ROUNDS];
};

__device__ __forceinline__ int chunk_id(const ChunkRounds& R, int prefix) {
    int r = 0;
    while (r + 1 < R.n && prefix >= R.base[r + 1]) ++r;
    return R.first[r] + (prefix - R.base[r]) / R.cost[r];
}

constexpr int CT_CELLS = 1024;  // cells per block of the table builder (4 per thread)

__device__ __forceinline__ int cell_cost(const DevParams& P, const GridDesc& G, const void* mask, int idx, int ncells) {
    if (idx >= ncells) return 0;
    const int wx = G.nx + 2 * G.ring;
    const int jj = idx / wx;
    return cell_is_wet(P, mask, cell_index(G, idx - jj * wx - G.ring, jj - G.ring)) ? AO_WET_COST : 1;
}

__global__ __launch_bounds__(256) void chunk_block_costs_kernel(const DevParams* __restrict__ g_params, GridDesc G,
                                                                const void* mask, int ncells, int* __restrict__ sums) {
    __shared__ int wave_sum[4];
    const DevParams& P = *g_params;
    const int base = blockIdx.x * CT_CELLS + threadIdx.x * 4;
    int c = 0;
    for (int n = 0; n < 4; ++n) c += cell_cost(P, G, mask, base + n, ncells);
    for (int d = 32; d; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// exclusive scan of the block sums by one wave (a few hundred to a few thousand entries)
__global__ __launch_bounds__(64) void chunk_scan_kernel(int nblocks, int* __restrict__ sums) {
    int carry = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 64) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblocks ? sums[b] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if ((int)threadIdx.x >= d) incl += up;
        }
        if (b < nblocks) sums[b] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
    if (threadIdx.x == 0) sums[nblocks] = carry;
}

// chunk id of a cell = floor(exclusive cost prefix / chunk cost); a cell whose id exceeds its predecessor's
// begins a chunk.  meta[0] = number of chunks.
__global__ __launch_bounds__(256) void chunk_begins_kernel(const DevParams* __restrict__ g_params, GridDesc G,
                                                           const void* mask, int ncells, const int* __restrict__ sums,
                                                           ChunkRounds R, int* __restrict__ begins, int* __restrict__ meta) {
    __shared__ int wave_sum[4];
    const DevParams& P = *g_params;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blockIdx.x * CT_CELLS + threadIdx.x * 4;
    int c[4], mine = 0;
    for (int n = 0; n < 4; ++n) {
        c[n] = cell_cost(P, G, mask, base + n, ncells);
        mine += c[n];
    }
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int prefix = sums[blockIdx.x] + incl - mine;
    for (int w = 0; w < wave; ++w) prefix += wave_sum[w];
    int prev_cost = base > 0 ? cell_cost(P, G, mask, base - 1, ncells) : 0;
    for (int n = 0; n < 4; ++n) {
        const int idx = base + n;
        if (idx < ncells) {
            const int id = chunk_id(R, prefix);
            if (idx == 0)
                begins[0] = 0;
            else if (id != chunk_id(R, prefix - prev_cost))
                begins[id] = idx;
            if (idx == ncells - 1) {
                begins[id + 1] = ncells;
                meta[0] = id + 1;
            }
        }
        prefix += c[n];
        prev_cost = c[n];
    }
}

// Layers.  Workgroups are dealt out in blockIdx order, one per CU before a CU gets its second, so workgroup b
// is the (b / CUs)-th arrival on its CU — and the SIMD arbiter serves the OLDEST wave first (s_setprio does not
// change that: measured).  With three equal workgroups per CU the first finishes at 60 % of the kernel and the
// third runs the last third alone, too few waves to keep the FP64 pipe busy (lifetimes 58 / 73 / 92 µs at
// equal work).  So the work is handed out in proportion to the share each arrival gets: 1024, 768 and 512
// wet cells for the last three layers (all multiples of 256 = whole batches for four waves), 1024 for every
// layer before them (on larger surfaces a new workgroup starts whenever the oldest one retires and the
// pipeline staggers itself; only the tail needs shaping).  Pure host arithmetic (tests/test_abi.py checks it
// without a GPU through cf_debug_chunk_plan); returns the largest chunk size used.
int plan_chunk_rounds(long total, int cu_count, int forced_wet_per_chunk, ChunkRounds* out) {
    const int layer = cu_count > 0 ? cu_count : 256;
    ChunkRounds R{};
    int next_id = 0, largest = 0;
    R.base[0] = 0;
    auto add_round = [&](int w, long count, bool last) {
        if (count <= 0) return;
        const int cost = w * AO_WET_COST;
        R.cost[R.n] = cost;
        R.first[R.n] = next_id;
        R.base[R.n + 1] = last ? (int)total + AO_WET_COST : R.base[R.n] + (int)(count * cost);
        next_id += (int)count;
        largest = w > largest ? w : largest;
        ++R.n;
    };
    auto cap = [&](int w) { return (long)layer * w * AO_WET_COST; };
    auto chunks = [&](long cost_units, int w) {
        return cost_units <= 0 ? 0L : (cost_units + (long)w * AO_WET_COST - 1) / ((long)w * AO_WET_COST);
    };
    const long need = total > 0 ? total : 1;
    if (forced_wet_per_chunk > 0) {
        add_round(forced_wet_per_chunk, chunks(need, forced_wet_per_chunk), true);  // forced uniform size
    } else if (need <= cap(256)) {
        add_round(256, chunks(need, 256), true);
    } else if (need <= cap(512)) {
        add_round(512, chunks(need, 512), true);
    } else {
        // 1024s for everything before the last two layers (none on a surface that fits three layers), then as
        // many 768s as still needed, then 512s
        const long body = need - cap(768) - cap(512);
        const long n1024 = body > 0 ? (body + cap(1024) - 1) / cap(1024) * layer : 0;  // whole layers of 1024s
        add_round(1024, n1024, false);
        const long left = need - n1024 * 1024L * AO_WET_COST;
        const long n768 = left > cap(768) ? layer : chunks(left, 768);                 // a whole layer of 768s if needed
        add_round(768, n768, false);
        add_round(512, chunks(left - n768 * 768L * AO_WET_COST, 512), false);          // the youngest layer takes the rest
        R.base[R.n] = (int)total + AO_WET_COST;  // the last round added absorbs the end
    }
    *out = R;
    return largest;
}

hipError_t build_chunk_table(hipStream_t st, const DevParams* d_params, const GridDesc& G, const void* mask, int cu_count,
                             int wet_per_chunk, int* d_sums, int* d_begins, int* d_meta, int* wet_per_chunk_out,
                             int* nchunks_out) {
    const int ncells = (G.nx + 2 * G.ring) * (G.ny + 2 * G.ring);
    const int nblocks = (ncells + CT_CELLS - 1) / CT_CELLS;
    hipLaunchKernelGGL(chunk_block_costs_kernel, dim3(nblocks), dim3(256), 0, st, d_params, G, mask, ncells, d_sums);
    hipLaunchKernelGGL(chunk_scan_kernel, dim3(1), dim3(64), 0, st, nblocks, d_sums);
    int total = 0;
    hipError_t e = hipMemcpyAsync(&total, d_sums + nblocks, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    ChunkRounds R{};
    const int largest = plan_chunk_rounds(total, cu_count, wet_per_chunk, &R);
    wet_per_chunk = largest;
    hipLaunchKernelGGL(chunk_begins_kernel, dim3(nblocks), dim3(256), 0, st, d_params, G, mask, ncells, d_sums, R, d_begins,
                       d_meta);
    int n = 0;
    if ((e = hipMemcpyAsync(&n, d_meta, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *wet_per_chunk_out = wet_per_chunk;
    *nchunks_out = n;
    return hipGetLastError();
}

int chunk_table_capacity(int ncells) { return (int)(((long)ncells * AO_WET_COST) / (256L * AO_WET_COST)) + 16; }
int chunk_sums_capacity(int ncells) { return (ncells + CT_CELLS - 1) / CT_CELLS + 1; }

// ---- production solver: LDS tables, one workgroup per chunk of the table above -----------------
template <bool COARE, int SPEC>
__global__ __launch_bounds__(AO_BLOCK, 3) void ao_flux_fast_kernel(LoopParams L, GridDesc G, OceanIn O, Exchange E,
                                                                FluxOut F, const double* __restrict__ g_tab,
                                                                const DevParams* __restrict__ g_params,
                                                                uint8_t* __restrict__ hint,
                                                                const int* __restrict__ chunk_begins) {
    // Land cells (≈30 % of a global grid) must not occupy lanes for 10–20 iterations: the chunk is first
    // compacted to the list of its wet cells (land gets its zeros there and then), counting-sorted by the
    // trip-count hint, and waves then pull 64 list entries at a time from an LDS cursor, so every lane that
    // enters the solver holds an ocean cell and the lanes of a batch finish together.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    int* list = reinterpret_cast<int*>(smem + TABLE_BYTES);
    int* counters = list + AO_CHUNK;  // [0] wet count, [1] cursor
    int* hist = counters + 4;
    int* bin_start = hist + AO_BINS;
    DevParams* lp = reinterpret_cast<DevParams*>(smem + AO_PARAMS_OFFSET);
    const int tid = threadIdx.x, lane = tid & 63;
    // Parameters first (ordinary loads), then the 47 KB of tables as LDS-DMA (global_load_lds, 1 KB per wave
    // instruction, no VGPR round trip).  The tables are first read in the batch phase, two barriers later, so
    // the first barrier below deliberately does NOT drain the DMA: the classification's own loads run while the
    // tables stream in.
    for (int n = tid; n < (int)(sizeof(DevParams) / sizeof(double)); n += AO_BLOCK)
        reinterpret_cast<double*>(lp)[n] = reinterpret_cast<const double*>(g_params)[n];
    if (tid < 2) counters[tid] = 0;
    if (tid < AO_BINS) hist[tid] = 0;
    static_assert(TABLE_BYTES % 1024 == 0, "the table stage copies whole 1 KB pieces");
    {
        const char* gb = reinterpret_cast<const char*>(g_tab);
        for (int c = tid >> 6; c < TABLE_BYTES / 1024; c += AO_BLOCK / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + c * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only: vmcnt / expcnt fields left at their maxima
    __builtin_amdgcn_s_barrier();
    const DevParams& P = *lp;  // prologue-only parameters live in LDS, not in SGPRs
    const double* logt = tab + LOG_OFFSET;

    const int wx = G.nx + 2 * G.ring;
    const int chunk = (int)blockIdx.x;  // dispatch order = layer order of the chunk table
    const int range_end = chunk_begins[chunk + 1];
    int begin = chunk_begins[chunk], end = range_end;
    bool first_piece = true;
    for (;;) {
        if (!first_piece) {
            if (tid < 2) counters[tid] = 0;
            if (tid < AO_BINS) hist[tid] = 0;
            __syncthreads();
        }
        first_piece = false;
        // ---- phase 1: classify, zero land, histogram of the trip-count hints -----------------------
        // (the hint is the cell's iteration count in the previous call — fields evolve slowly from one
        // coupled step to the next; it only orders the list and cannot change any result)
        for (int idx = begin + tid; idx < end; idx += AO_BLOCK) {
            const int jj = idx / wx;
            const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
            if (cell_is_wet(P, O.mask, k)) {
                const int bin = hint ? AO_BINS - 1 - min((int)hint[k], AO_BINS - 1) : 0;  // longest first (LPT)
                atomicAdd(&hist[bin], 1);
            } else {  // zero_interface_state: all fluxes 0, T = 0 K
                CellFluxes Z{};
                Z.Ts_ocean = -P.T_offset;
                Z.iterations = L.fixed ? L.maxiter : 0;
                store_fluxes(F, k, Z);
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
        const int nwet = counters[0];
        if (nwet > AO_CHUNK) {  // only with a stale chunk table: retry on a piece that cannot overflow the list
            end = begin + AO_CHUNK;
            __syncthreads();
            continue;
        }
        // ---- phase 2: scatter the wet cells into their bins ----------------------------------------
        for (int idx = begin + tid; idx < end; idx += AO_BLOCK) {
            const int jj = idx / wx;
            const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
            if (cell_is_wet(P, O.mask, k)) {
                const int bin = hint ? AO_BINS - 1 - min((int)hint[k], AO_BINS - 1) : 0;
                list[atomicAdd(&bin_start[bin], 1)] = idx;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's share of the table DMA has landed
        __syncthreads();
        // ---- phase 3: waves pull 64 wet cells at a time --------------------------------------------
        // (requesting the next batch's inputs before iterating the current one was tried: +20 VGPRs cost the
        // third wave per SIMD or spills, 107 → 122–142 µs)
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
        if (end >= range_end) break;
        begin = end;  // stale-table path: the rest of the range
        end = range_end;
        __syncthreads();  // list and counters are reused
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
    const double* logt = tab + LOG_OFFSET;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const double v = x[k];
        double r;
        switch (fn) {
            case 0: r = flog(logt, v); break;
            case 1: r = fexp(v); break;
            case 2: r = fcbrt(v); break;
            case 3: r = fsqrt(v); break;
            case 4: r = frcp(v); break;
            case 5: r = psi_eval(tab, 0, psi_arg(v)); break;
            case 6: r = psi_eval(tab, 1, psi_arg(v)); break;
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
                               E, F, L.d_tables, L.d_params, L.d_hint, L.d_chunk_begins);
            break;
        case SOLVER_ICE:
            hipLaunchKernelGGL((ao_flux_fast_kernel<COARE, SOLVER_ICE>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G, O, E,
                               F, L.d_tables, L.d_params, L.d_hint, L.d_chunk_begins);
            break;
        case SOLVER_LY:
            hipLaunchKernelGGL((ao_flux_fast_kernel<true, SOLVER_LY>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G, O, E,
                               F, L.d_tables, L.d_params, L.d_hint, L.d_chunk_begins);
            break;
        default:
            hipLaunchKernelGGL((ao_flux_fast_kernel<COARE, SOLVER_GENERIC>), grid, dim3(AO_BLOCK), AO_LDS_BYTES, st, C, G,
                               O, E, F, L.d_tables, L.d_params, L.d_hint, L.d_chunk_begins);
    }
}

hipError_t launch_ao_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C,
                            const GridDesc& G, const cf_ocean_surface* o, const cf_exchange_fields* e,
                            const cf_interface_fluxes* f) {
    if (L.solver == CF_SOLVER_LIBM) return launch_ao_fluxes_libm(st, P, G, o, e, f);
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    // one workgroup per chunk of the cost-balanced table: the hardware dispatcher is the dynamic load balancer
    if (!L.d_chunk_begins || L.n_chunks <= 0) return hipErrorInvalidValue;
    dim3 grid(L.n_chunks);
    if (P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC)
        launch_ao_spec<true>(st, grid, L, C, G, O, E, F);
    else
        launch_ao_spec<false>(st, grid, L, C, G, O, E, F);
    return hipGetLastError();
}

// self-test hook: the chunk plan for a given total cost (tests/test_abi.py, no GPU needed).
// out[0] = rounds, then per round: wet cells per chunk, number of chunks
extern "C" int cf_debug_chunk_plan(long long total_cost, int cu_count, int forced_wet_per_chunk, int* out, int capacity) {
    ChunkRounds R{};
    plan_chunk_rounds((long)total_cost, cu_count, forced_wet_per_chunk, &R);
    if (!out || capacity < 1 + 2 * R.n) return -1;
    out[0] = R.n;
    for (int r = 0; r < R.n; ++r) {
        out[1 + 2 * r] = R.cost[r] / AO_WET_COST;
        const long span = (long)R.base[r + 1] - R.base[r];
        out[2 + 2 * r] = (int)((span + R.cost[r] - 1) / R.cost[r]);
    }
    return AO_WET_COST;
}

hipError_t launch_debug_eval(hipStream_t st, const LaunchCfg& L, int fn, int n, const double* x, double* y) {
    hipLaunchKernelGGL(debug_eval_kernel, dim3(64), dim3(256), TABLE_BYTES, st, fn, n, x, y, L.d_tables);
    return hipGetLastError();
}

}  // namespace coflux
