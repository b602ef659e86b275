// coflux_solver.hip — compute_atmosphere_ocean_fluxes! on gfx950: the Monin–Obukhov fixed point.
//
// One lane = one ocean cell.  The kernel is FP64-issue bound (≈ 130–200 VALU instructions per
// iteration × 10–20 iterations per cell against 128 algorithmic bytes), so the design goal is to
// waste no issue slot: ψ/log tables in LDS (coflux_fast.hpp), land compacted away before the
// iteration, batches of cells with equal trip counts, waves leaving the loop on a wave64 ballot,
// one workgroup per chunk of a table that gives every workgroup whole batches and fills the
// device's resident-workgroup slots in whole rounds.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "coflux_solver_shared.hpp"
#include "coflux_lean.hpp"
#include "coflux_lean_kernel.hpp"

namespace coflux {

#ifndef CF_LEAN_WAVES
#define CF_LEAN_WAVES 3  // waves per SIMD the lean ocean solver is compiled for
#endif

// One workgroup geometry: 256 threads, three workgroups per CU (each with its own copy of the tables), chunks of ≤ 1280 wet
// cells in arrival layers.  (Rounds 2-5 also shipped a 768-thread one-workgroup-per-CU geometry, CF_OPT_AO_CHUNK = 3072:
// twelve waves of one age do not stagger themselves as three workgroups of different age do — 3 % slower on the 1/4°
// surface, bitwise the same results; retired in round 6.)
int wet_list_stride();
template <int BLOCK>
struct Geom {
    static_assert(BLOCK == AO_BLOCK, "one workgroup geometry");
    static constexpr int CHUNK = AO_CHUNK;
    static constexpr int COUNTERS_OFFSET = TABLE_DOUBLES * 8 + CHUNK * 4;
    static constexpr int PARAMS_OFFSET = COUNTERS_OFFSET + 16 + 2 * 64 * 4;
    static constexpr int LDS_BYTES = PARAMS_OFFSET + (int)sizeof(DevParams);
};
constexpr int AO_LAYER_1 = AO_CHUNK < 1024 ? AO_CHUNK : 1024, AO_LAYER_2 = AO_CHUNK < 768 ? AO_CHUNK : 768, AO_LAYER_3 = 512;  // wet cells per chunk of the last three arrival layers (plan_chunk_rounds)
static_assert(AO_LAYER_1 <= AO_CHUNK && AO_LAYER_1 >= AO_LAYER_2 && AO_LAYER_2 >= AO_LAYER_3, "layer sizes");
static_assert(AO_BINS == 64, "Geom<>::PARAMS_OFFSET spells the bin count out");
static_assert(Geom<AO_BLOCK>::LDS_BYTES <= 53760, "three narrow solver workgroups must fit the CU's 160 KB of LDS");


// ---------------------------------------------------------------------------------------------
// Chunk table.  Two quantisation effects cost ≈ 25 % each when every workgroup simply takes 512 surface
// cells: (i) a workgroup's four waves pull 64-cell batches of wet cells, so a chunk is only used well if it
// holds a multiple of 256 wet cells (≈ 370 wet cells = six batches leave two waves — and, with the same
// wave→SIMD placement in every workgroup, two SIMDs of the CU — idle for the second half); (ii) the device
// holds 3 workgroups per CU, and a dispatch "round" that is only half full runs at half throughput.
// The wet mask is static, so the surface is cut ONCE per mask into chunks of prescribed cost with
// wet = AO_WET_COST, land = 1.  The host picks a descending sequence of rounds (plan_chunk_rounds below: arrival
// layers of 1024 / 768 / 512 wet cells on a surface that fills the device, 256-cell chunks on one that does not)
// so that the long workgroups start first and whatever tail is left is short.  An open-ocean chunk holds
// exactly its nominal wet count, a coastal one slightly fewer, a land chunk at most 64× as many cells (it
// only writes zeros).  The table and the static wet lists built with it only steer scheduling: the solver checks
// the list against the mask as it is on every call and falls back to classifying its range (in pieces, if a range
// holds more wet cells than the list) when a mask changed in place — a stale table can cost time, never correctness.
// ---------------------------------------------------------------------------------------------
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
// wet cells for the last three layers (all multiples of 256 = whole batches for four waves).  Round 2's kernel, whose
// start phase took 9 µs, did best with 1280/512/512; with round 3's (4 µs start phase, batches in index order)
// the same-box scan reads 1024/768/512 65.6 µs, 1280/512/512 68.1, 1280/768/256 68.0, 1152/640/512 67.5, 1280/640/384
// 68.4, 1024/640/640 70.3, 768/768/768 70.8 (profiles/r03_experiments.md).  1024 for every
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
    const bool tail_plan = forced_wet_per_chunk == AO_PLAN_TAIL;
    if (tail_plan) forced_wet_per_chunk = 0;
    if (forced_wet_per_chunk > 0) {
        add_round(forced_wet_per_chunk, chunks(need, forced_wet_per_chunk), true);  // forced uniform size
    } else if (need <= 3 * cap(256)) {
        // small surface (a slab of a strongly scaled run): every chunk resident at once, one batch per wave — the
        // kernel is as long as its slowest batch, so nothing may queue behind anything
        //
        // More chunks than CUs (a 1440×70 latitude slab: 292 on 256; 1440×140: 561): a wave of the LAST dispatch layer shares
        // its SIMD with older waves, which the issue arbiter serves first — on the 1/8 slab its trips took 0.82 instead of
        // 0.59 µs and the layer ended the kernel 6 µs after the first one (per-wave stamps, profiles/r05_experiments.md §10;
        // issue priority does not change it, and it is not the CU's LDS pipe).  So that layer is cut into the SMALLEST pieces
        // that still give every CU at most one of them — 64 wet cells = one working wave where the excess allows: as few
        // cells as possible in the slow chains, no CU with a whole workgroup more than the others (1440×70: 27.3 → 26.0 µs per
        // step, 1440×130: 33.6 → 32.6, 1440×140: 34.1 → 33.7).  (COFLUX_SLAB_SPLIT=0 with COFLUX_EXPERIMENTS=1: the uniform plan.)
        const long n256 = chunks(need, 256);
        int w2 = 256;
        const long whole = n256 > layer ? (n256 - 1) / layer : 0;  // dispatch layers of whole workgroups ahead of the last one (0: one layer is all)
        if (whole >= 1) {
            static const bool split = [] { const char* e = experiment_knob("COFLUX_SLAB_SPLIT"); return !(e && e[0] == '0'); }();
            const long excess = need - whole * cap(256);
            if (split)
                for (w2 = 64; w2 < 256 && chunks(excess, w2) > layer; w2 += 64) {}
        }
        if (w2 < 256) {
            add_round(256, whole * layer, false);
            add_round(w2, chunks(need - whole * cap(256), w2), true);
        } else {
            add_round(256, n256, true);
        }
    } else if (need <= cap(512)) {
        // (never reached behind the branch above; uniform 512-cell chunks on half a surface were measured: 1440×280 steps
        // in 71.6 µs with them, 65.5 µs with the layered plan below)
        add_round(512, chunks(need, 512), true);
    } else {
        // 1024s for everything before the last two layers (none on a surface that fits three layers), then as
        // many 768s as still needed, then 512s
        int W1 = AO_LAYER_1, W2 = AO_LAYER_2, W3 = AO_LAYER_3;
        // A launch with tail workgroups WANTS its solver workgroups to retire at different times — every slot freed early is
        // taken by the memory-bound riders — so on a surface that fits one dispatch generation the three arrivals get EQUAL
        // chunks (the first retires at ≈ 60 % of the kernel, see above): same box, step with the tail, 1024/768/512 vs
        // 768/768/768: `:corrected` 80.0–80.7 → 75.8–77.1 µs, `:ncar` 60.7–64.6 → 60.8–60.9, `:default` 86.3–87.9 → 85.8–87.9
        // (profiles/r04_tail_layers*.log).  Only where the surface needs all three layers: larger surfaces stagger themselves,
        // smaller ones want the smaller workgroups of the plan below (a 3/4 surface: 80.9 µs per step with equal chunks,
        // 72.4 with the layers; a half surface 64.8 vs 55.4).
        if (tail_plan && need > cap(AO_LAYER_1) + cap(AO_LAYER_2) && need <= cap(AO_LAYER_1) + cap(AO_LAYER_2) + cap(AO_LAYER_3))
            W1 = W2 = W3 = AO_LAYER_2;
        if (const char* env = experiment_knob("COFLUX_LAYERS")) {  // experiments only (with COFLUX_EXPERIMENTS=1): "w1,w2,w3" (multiples of 64, w1 ≥ w2 ≥ w3, w1 ≤ AO_CHUNK)
            int a = 0, b = 0, c = 0;
            if (std::sscanf(env, "%d,%d,%d", &a, &b, &c) == 3 && a <= AO_CHUNK && a >= b && b >= c && c >= 64 && a % 64 == 0 && b % 64 == 0 && c % 64 == 0) {
                W1 = a;
                W2 = b;
                W3 = c;
            }
        }
        const long body = need - cap(W2) - cap(W3);
        const long n1 = body > 0 ? (body + cap(W1) - 1) / cap(W1) * layer : 0;  // whole layers of the largest size
        add_round(W1, n1, false);
        const long left = need - n1 * (long)W1 * AO_WET_COST;
        const long n2 = left > cap(W2) ? layer : chunks(left, W2);                 // a whole middle layer if needed
        add_round(W2, n2, false);
        add_round(W3, chunks(left - n2 * (long)W2 * AO_WET_COST, W3), false);          // the youngest layer takes the rest
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
    const int largest = plan_chunk_rounds(total, cu_count, wet_per_chunk, &R);  // (AO_PLAN_TAIL passes through)
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

// (+ 320: a small surface's second dispatch layer may be cut into up to one 64-cell chunk per CU, plan_chunk_rounds)
int chunk_table_capacity(int ncells) { return (int)(((long)ncells * AO_WET_COST) / (256L * AO_WET_COST)) + 16 + 320; }
int chunk_sums_capacity(int ncells) { return (ncells + CT_CELLS - 1) / CT_CELLS + 1; }

// ---------------------------------------------------------------------------------------------
// Static wet lists.  The wet mask is static, so besides the chunk boundaries the builder also writes, once per
// mask, every chunk's wet cells in index order (wet_pos, 4 B per wet cell) and the prefix of the chunks' wet
// counts (wet_start).  A per-wet-cell byte array (trip) carries each cell's iteration count from one call to the
// next.  The solver's start phase is then two coalesced loads per thread and a counting sort in LDS — no mask
// reads, no index arithmetic, no dependent scattered loads while every workgroup of the device starts at once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chunk_wet_fill_kernel(const DevParams* __restrict__ g_params, GridDesc G,
                                                             const void* mask, const int* __restrict__ begins,
                                                             uint32_t* __restrict__ wet_pos, int* __restrict__ overflow,
                                                             int stride) {
    __shared__ int wave_count[4];
    const DevParams& P = *g_params;
    const int wx = G.nx + 2 * G.ring, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int begin = begins[blockIdx.x], end = begins[blockIdx.x + 1];
    uint32_t* out = wet_pos + (size_t)blockIdx.x * stride;
    int base = 0;
    for (int strip = begin; strip < end; strip += 256) {  // index order: strips in order, waves in order, lanes in order
        const int idx = strip + (int)threadIdx.x;
        bool wet = false;
        if (idx < end) {
            const int jj = idx / wx;
            wet = cell_is_wet(P, mask, cell_index(G, idx - jj * wx - G.ring, jj - G.ring));
        }
        const unsigned long long m = __ballot(wet);
        if (lane == 0) wave_count[wave] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_count[w];
        const int p = off + __popcll(m & ((1ull << lane) - 1ull));
        if (wet && p < stride) out[p] = (uint32_t)idx;
        base += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
        __syncthreads();
    }
    for (int p = base + (int)threadIdx.x; p < stride; p += 256) out[p] = 0xffffffffu;  // sentinel: no cell
    if (threadIdx.x == 0 && base > stride) atomicAdd(overflow, 1);
}

// Chunk c's list occupies entries [c·stride, (c+1)·stride) of wet_pos / trip — a fixed stride per geometry, so the solver
// needs no lookup before it can request its list.  *overflow_out != 0: some chunk holds more wet cells than a list
// (cannot happen with the cost-balanced table; the solver then classifies per call).
hipError_t build_wet_lists(hipStream_t st, const DevParams* d_params, const GridDesc& G, const void* mask, int nchunks,
                           const int* d_begins, uint32_t* d_wet_pos, uint8_t* d_trip, int* d_scratch, int* overflow_out) {
    const int stride = wet_list_stride();
    hipError_t e = hipMemsetAsync(d_scratch, 0, sizeof(int), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(chunk_wet_fill_kernel, dim3(nchunks), dim3(256), 0, st, d_params, G, mask, d_begins, d_wet_pos, d_scratch, stride);
    if ((e = hipMemsetAsync(d_trip, 0, (size_t)nchunks * stride, st)) != hipSuccess) return e;
    int overflow = 0;
    if ((e = hipMemcpyAsync(&overflow, d_scratch, sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *overflow_out = overflow;
    return hipGetLastError();
}

int wet_list_stride() { return AO_CHUNK; }
size_t wet_list_capacity(int ncells) { return (size_t)chunk_table_capacity(ncells) * AO_CHUNK; }


// (second launch bound = waves per SIMD: the lean ocean iteration fits 128 VGPRs — four narrow workgroups per CU, whose
// LDS, tables included, is 39 KB each; everything else keeps three)
// TAIL: workgroups behind the chunk table's interpolate the NEXT step's atmosphere state (see ao_lean_kernel; instantiated for
// CoefficientBasedFluxes with the fused net fluxes, the OMIP-2 standard configuration)
// land interior cell of the sea-ice interface launch with the net sea-ice fluxes in its epilogue: no heat either way
__device__ __forceinline__ void ice_zero_net(SolverArgsPtr K, const GridDesc& G, size_t k, int i, int j) {
    double* top = K->NI.top;
    if (top && i >= 0 && i < G.nx && j >= 0 && j < G.ny) {
        top[k] = 0.0;
        K->NI.bottom[k] = 0.0;
    }
}

// (a device routine: the kernel below, and ice_ocean_kernel, where the ocean solver's workgroups ride behind these)
template <bool COARE, int SPEC, bool FUSE_NET, int BLOCK, bool TAIL = false>
__device__ __forceinline__ void ao_flux_fast_body(SolverArgsPtr K_in, const int block) {
    constexpr int CHUNK = Geom<BLOCK>::CHUNK;
    SolverArgsPtr K = opaque(K_in);
    if constexpr (TAIL) {
        static_assert(BLOCK == 64 * IT_WAVES, "a tail workgroup is an interpolation workgroup");
        const int nch = (int)K->n_chunks;
        if (block >= nch) {
            int b = block - nch;
            const int nb = (int)K->tail_blocks;
            const GridDesc Gt = kread(&K->G);
            if (b < nb) {
                const SourceDesc S = kread(&K->Si);
                const WeightDesc Wt = kread(&K->Wi);
                const Exchange En = kread(&K->E_next);
                const int cap = (int)K->tail_cap, rows = (int)K->tail_rows;
                if (rows == 4) interpolate_tiles<4>(S, Wt, Gt, En, cap, b, nb);
                else if (rows == 2) interpolate_tiles<2>(S, Wt, Gt, En, cap, b, nb);
                else interpolate_tiles<1>(S, Wt, Gt, En, cap, b, nb);
                return;
            }
            // face stresses (net_stress_kernel's arithmetic: net_face_stress, contraction off — the same bits)
            b -= nb;
            const int idx = b * 256 + (int)threadIdx.x;
            if (idx >= Gt.nx * Gt.ny) return;
            const DevParams& Po = *K->stress_params;
            const IceIn I = kread(&K->stress_ice);
            const void* smask = K->stress_mask;
            const double* __restrict__ rtx = K->rtx;
            const double* __restrict__ rty = K->rty;
            const int j = idx / Gt.nx;
            const size_t k = cell_index(Gt, idx - j * Gt.nx, j);
            const size_t kw = k - 1, ks = k - (size_t)Gt.sj;
            const bool wet = cell_is_wet(Po, smask, k);
            const double aice = I.conc ? I.conc[k] : 0.0;
            const double tx = net_face_stress(Po, rtx[kw], rtx[k], I.conc ? I.conc[kw] : 0.0, aice, I.txio ? I.txio[k] : 0.0);
            const double ty = net_face_stress(Po, rty[ks], rty[k], I.conc ? I.conc[ks] : 0.0, aice, I.tyio ? I.tyio[k] : 0.0);
            K->tau_x[k] = wet ? tx : 0.0;
            K->tau_y[k] = wet ? ty : 0.0;
            return;
        }
    }
    const LoopParams L = kread(&K->L);
    const GridDesc G = kread(&K->G);
    const WetLists W = kread(&K->W);
    const double* __restrict__ g_tab = K->g_tab;
    const DevParams* __restrict__ g_params = K->g_params;
    const int* __restrict__ chunk_begins = K->chunk_begins;
    const void* mask = K->O.mask;
    // Land cells (≈30 % of a global grid) must not occupy lanes for 10–20 iterations: a workgroup works on the
    // LIST of its chunk's wet cells, counting-sorted by the trip count of the previous call (longest first), and
    // its waves pull 64 list entries at a time from an LDS cursor, so every lane that enters the solver holds an
    // ocean cell and the lanes of a batch finish together.  The list is static (built with the chunk table) and
    // is checked against the mask as it is NOW in the start phase: a mask rewritten in place makes it stale, a
    // 64-bit fingerprint of the range's wet set gives that away, and the workgroup redoes its range by classifying
    // it — a stale list costs time, never correctness.  Land gets its zeros right behind the start phase.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    unsigned* list = reinterpret_cast<unsigned*>(smem + TABLE_BYTES);
    int* counters = reinterpret_cast<int*>(smem + Geom<BLOCK>::COUNTERS_OFFSET);  // [0] wet count, [1] cursor, [2] wet cells seen, [3] stale
    int* hist = counters + 4;
    int* bin_start = hist + AO_BINS;
    DevParams* lp = reinterpret_cast<DevParams*>(smem + Geom<BLOCK>::PARAMS_OFFSET);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wx = G.nx + 2 * G.ring;
    const unsigned wx_rcp = (unsigned)K->wx_reciprocal;
    const int chunk = block;  // dispatch order = layer order of the chunk table
    bool use_static = W.pos != nullptr;
    // Order of the start phase — everything is REQUESTED before anything is looked at, in straight-line code (a
    // branch around a load makes the compiler wait for all memory at the next join):
    // (1) the parameter block and the 45 KB of tables as LDS-DMA (global_load_lds, 1 KB per wave instruction, no VGPR
    //     round trip): the long transfer first — the compiler drains it before the first LDS access behind the barrier
    //     anyway (it cannot tell the DMA's destination from the lists');
    // (2) my share of the chunk's static list — entries tid, tid + 256, … of a fixed-stride array, so nothing has to be
    //     looked up first: coalesced 4-byte and 1-byte loads;
    // (3) the raw mask words of my share of the chunk's cell range (validation and land, below): they depend on nothing
    //     but the chunk table.  One aligned 4-byte word per cell of a byte mask, the two halves of the double for a
    //     bottom-height mask — the same two load instructions either way, so no branch.
    static_assert(sizeof(DevParams) % 16 == 0 && sizeof(DevParams) <= 1024, "the parameter block is one LDS-DMA piece");
    static_assert(Geom<BLOCK>::PARAMS_OFFSET % 16 == 0, "LDS-DMA destination alignment");
    if (tid < (int)(sizeof(DevParams) / 16))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g_params) + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + Geom<BLOCK>::PARAMS_OFFSET), 16, 0, 0);
    static_assert(TABLE_BYTES % 1024 == 0, "the table stage copies whole 1 KB pieces");
    {
        const char* gb = reinterpret_cast<const char*>(g_tab);
        constexpr int PIECES = TABLE_BYTES / 1024, WAVES = BLOCK / 64;
#pragma unroll
        for (int r = 0; r < (PIECES + WAVES - 1) / WAVES; ++r) {
            const int c = (tid >> 6) + r * WAVES;
            if (c < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + c * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
        }
    }
    constexpr int PER_THREAD = CHUNK / BLOCK;
    constexpr int LAND_UNROLL = 8;  // strips of the range whose mask values are requested up front
    typedef const __attribute__((address_space(1))) unsigned* GlobalWords;
    int my_idx[PER_THREAD], my_trip[PER_THREAD];
    unsigned raw_lo[LAND_UNROLL], raw_hi[LAND_UNROLL], raw_shift = 0;  // raw_shift: 2 bits per strip, the byte within its word
    // (scalar loads through the constant address space: a vector load here would sit behind the table DMA issued above
    // in the in-order vector-memory queue, and its wait would hold back the list and mask requests until the DMA landed)
    const __attribute__((address_space(4))) int* cb = (const __attribute__((address_space(4))) int*)chunk_begins;
    const int range_begin = cb[chunk], range_end = cb[chunk + 1];
    const int mask_kind = (mask == nullptr) ? CF_MASK_NONE : (int)K->mask_kind;
    const double z_surface = K->z_surface;
    const double T_offset = K->T_offset;
    if (use_static) {
        const size_t base = (size_t)chunk * CHUNK + tid;
        const __attribute__((address_space(1))) uint32_t* gpos = (const __attribute__((address_space(1))) uint32_t*)W.pos;
        // no hint array: the bytes are read from the list itself and replaced below
        const __attribute__((address_space(1))) uint8_t* gtrip =
            (const __attribute__((address_space(1))) uint8_t*)(W.trip ? (const void*)W.trip : (const void*)W.pos);
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n) my_idx[n] = (int)gpos[base + n * BLOCK];
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n) my_trip[n] = (int)gtrip[base + n * BLOCK];
        // no mask: the words are read from the list and ignored
        const unsigned long long mbase = mask_kind == CF_MASK_NONE ? (unsigned long long)W.pos : (unsigned long long)mask;
        const unsigned stride = mask_kind == CF_MASK_NONE ? 0u : (mask_kind == CF_MASK_U8 ? 1u : 8u);
        const unsigned hi_step = mask_kind == CF_MASK_BOTTOM_HEIGHT ? 4u : 0u;
#pragma unroll
        for (int n = 0; n < LAND_UNROLL; ++n) {
            const int ic = min(range_begin + tid + n * BLOCK, range_end - 1);
            const int jj = row_of(ic, wx, wx_rcp);
            const unsigned long long a = mbase + (unsigned long long)cell_index(G, ic - jj * wx - G.ring, jj - G.ring) * stride;
            raw_shift |= ((unsigned)a & 3u) << (2 * n);
            raw_lo[n] = *(GlobalWords)(a & ~3ull);
            raw_hi[n] = *(GlobalWords)((a & ~3ull) + hi_step);
        }
    }
    if (tid < 4) counters[tid] = 0;
    if (tid < AO_BINS) hist[tid] = 0;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only: vmcnt / expcnt fields left at their maxima
    __builtin_amdgcn_s_barrier();
    // (the values requested above are not to be looked at before this point: computing with them earlier would put
    // a wait for memory in front of the table DMA's issue)
    if (use_static) {
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n) asm volatile("" : "+v"(my_idx[n]), "+v"(my_trip[n]));
#pragma unroll
        for (int n = 0; n < LAND_UNROLL; ++n) asm volatile("" : "+v"(raw_lo[n]), "+v"(raw_hi[n]));
        if (!W.trip) {
#pragma unroll
            for (int n = 0; n < PER_THREAD; ++n) my_trip[n] = AO_BINS - 1;
        }
    }
    const DevParams& P = *lp;  // prologue-only parameters live in LDS, not in SGPRs (valid once the DMA has drained)
    const double* logt = tab + LOG_OFFSET;

    int nwet = 0;
    bool have_list = false;
    if (use_static) {
        // ---- counting sort of the static list by trip-count bin (all in LDS), longest first (LPT) ---------
        // Alongside, the list is VALIDATED against the mask as it is now: the XOR of a 64-bit hash over the listed
        // cells must equal the XOR over the wet cells of the chunk's range (a mask rewritten in place makes the static
        // list stale; the workgroup then redoes its range by classifying it — a stale list costs time, never
        // correctness).  The mask values were requested before the table DMA, so none of this waits for memory.
        unsigned hx = 0, hy = 0;
        // No hint array (a fixed trip count: CoefficientBasedFluxes): the static list IS the order — it goes to LDS as it
        // stands (wet entries first, index order), counted on the way; no histogram, no scan, no scatter, one barrier.
        const bool unsorted = !W.trip;
        int listed_here = 0;
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n)
            if (my_idx[n] >= 0) {
                if (unsorted) {
                    list[tid + n * BLOCK] = ((unsigned)(tid + n * BLOCK) << AO_LIST_OFFSET_BITS) | (unsigned)(my_idx[n] - range_begin);
                    ++listed_here;
                } else {
                    atomicAdd(&hist[AO_BINS - 1 - trip_bin(my_trip[n])], 1);
                }
                hx ^= cell_hash_lo((unsigned)my_idx[n]);
                hy ^= cell_hash_hi((unsigned)my_idx[n]);
            }
        if (unsorted) {
            for (int d = 32; d; d >>= 1) listed_here += __shfl_xor(listed_here, d);
            if (lane == 0) atomicAdd(&counters[0], listed_here);
        }
        unsigned land = 0;  // bit n: strip n's cell is inside the range and dry — it gets its zeros after the last barrier
#pragma unroll
        for (int n = 0; n < LAND_UNROLL; ++n) {
            const int idx = range_begin + tid + n * BLOCK;
            const bool w = mask_kind == CF_MASK_NONE ? true
                           : mask_kind == CF_MASK_U8 ? ((raw_lo[n] >> (8 * ((raw_shift >> (2 * n)) & 3u))) & 0xffu) != 0
                                                     : !(z_surface <= __hiloint2double((int)raw_hi[n], (int)raw_lo[n]));
            if (idx < range_end) {
                if (w) {
                    hx ^= cell_hash_lo((unsigned)idx);
                    hy ^= cell_hash_hi((unsigned)idx);
                } else {
                    land |= 1u << n;
                }
            }
        }
        // a range longer than LAND_UNROLL strips (a chunk that is mostly land): the rest the plain way, zeros at once
        for (int idx = range_begin + tid + LAND_UNROLL * BLOCK; idx < range_end; idx += BLOCK) {
            const int jj = row_of(idx, wx, wx_rcp);
            const int i = idx - jj * wx - G.ring, j = jj - G.ring;
            const size_t k = cell_index(G, i, j);
            const bool w = mask_kind == CF_MASK_NONE ? true
                           : mask_kind == CF_MASK_U8 ? ((const uint8_t*)mask)[k] != 0 : !(z_surface <= ((const double*)mask)[k]);
            if (w) {
                hx ^= cell_hash_lo((unsigned)idx);
                hy ^= cell_hash_hi((unsigned)idx);
            } else {
                SolverArgsPtr Kz = opaque(K);
                const FluxOut F = kread(&Kz->F);
                const NetOut N = kread(&Kz->N);
                zero_cell<FUSE_NET>(L, T_offset, G, F, N, k, i, j);
                        if constexpr (SPEC == SOLVER_SEAICE || SPEC == SOLVER_SEAICE_LEAN) ice_zero_net(opaque(K), G, k, i, j);
            }
        }
        for (int d = 32; d; d >>= 1) {
            hx ^= (unsigned)__shfl_xor((int)hx, d);
            hy ^= (unsigned)__shfl_xor((int)hy, d);
        }
        if (lane == 0) {
            atomicXor(reinterpret_cast<unsigned*>(&counters[2]), hx);
            atomicXor(reinterpret_cast<unsigned*>(&counters[3]), hy);
        }
        if (!unsorted) {
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
            for (int n = 0; n < PER_THREAD; ++n)
                if (my_idx[n] >= 0) {
                    const int p = atomicAdd(&bin_start[AO_BINS - 1 - trip_bin(my_trip[n])], 1);
                    list[p] = ((unsigned)(tid + n * BLOCK) << AO_LIST_OFFSET_BITS) | (unsigned)(my_idx[n] - range_begin);
                }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's share of the table DMA has landed
        __syncthreads();
        nwet = counters[0];
        have_list = (counters[2] | counters[3]) == 0;  // the list is the range's wet set
        if (have_list) {
            // zero_interface_state of the range's land, issued behind the last barrier: nothing waits for these stores
            // but the first batch's loads, whose latency they share
            if (land) {
                SolverArgsPtr Kz = opaque(K);
                const FluxOut F = kread(&Kz->F);
                const NetOut N = kread(&Kz->N);
#pragma unroll
                for (int n = 0; n < LAND_UNROLL; ++n)
                    if (land & (1u << n)) {
                        const int idx = range_begin + tid + n * BLOCK;
                        const int jj = row_of(idx, wx, wx_rcp);
                        const int i = idx - jj * wx - G.ring, j = jj - G.ring;
                        zero_cell<FUSE_NET>(L, T_offset, G, F, N, cell_index(G, i, j), i, j);
                        if constexpr (SPEC == SOLVER_SEAICE || SPEC == SOLVER_SEAICE_LEAN) ice_zero_net(opaque(K), G, cell_index(G, i, j), i, j);
                    }
            }
        } else {  // stale list: redo the range the slow way
            use_static = false;
            __syncthreads();
            if (tid < 4) counters[tid] = 0;
            __syncthreads();
        }
    }
    if (!use_static) {  // no list at all: the classification below reads the parameter block, which is still in flight
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }
    int begin = range_begin, end = range_end;
    for (;;) {
        if (!have_list) {
            // ---- no (valid) static list: classify the piece [begin, end), zero its land ------------------
            for (int base = begin; base < end; base += BLOCK) {
                const int idx = base + tid;
                bool wet = false;
                if (idx < end) {
                    const int jj = row_of(idx, wx, wx_rcp);
                    const int i = idx - jj * wx - G.ring, j = jj - G.ring;
                    const size_t k = cell_index(G, i, j);
                    wet = cell_is_wet(P, mask, k);
                    if (!wet) {
                        SolverArgsPtr Kz = opaque(K);
                        const FluxOut F = kread(&Kz->F);
                        const NetOut N = kread(&Kz->N);
                        zero_cell<FUSE_NET>(L, T_offset, G, F, N, k, i, j);
                        if constexpr (SPEC == SOLVER_SEAICE || SPEC == SOLVER_SEAICE_LEAN) ice_zero_net(opaque(K), G, k, i, j);
                    }
                }
                const unsigned long long m = __ballot(wet);
                int wave_base = 0;
                if (lane == 0 && m) wave_base = atomicAdd(&counters[0], __popcll(m));
                wave_base = __shfl(wave_base, 0);
                if (wet) {
                    const int p = wave_base + __popcll(m & ((1ull << lane) - 1ull));
                    if (p < CHUNK) list[p] = (unsigned)(idx - range_begin);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the table DMA has landed
            __syncthreads();
            nwet = counters[0];
            if (nwet > CHUNK) {  // more wet cells than the list holds: retry on a piece that cannot overflow it
                end = begin + CHUNK;
                __syncthreads();
                if (tid < 4) counters[tid] = 0;
                __syncthreads();
                continue;
            }
        }
        // ---- waves pull 64 wet cells at a time ---------------------------------------------------------
        for (;;) {
            int start = 0;
            if (lane == 0) start = atomicAdd(&counters[1], 64);
            start = __shfl(start, 0);
            if (start >= nwet) break;
            const int q = start + lane;
            const bool in_range = q < nwet;
            const int qc = in_range ? q : nwet - 1;
            if constexpr (SPEC == SOLVER_SEAICE || SPEC == SOLVER_SEAICE_LEAN) {
                // ---- atmosphere–sea-ice interface: same list, same batches, the skin temperature inside the loop ----
                const IceParams Ice = kread(&K->Ice);
                IceConsts c;
                double Ts;
                bool ice_free = false;
                {
                    const int idx = range_begin + (int)(list[qc] & ((1u << AO_LIST_OFFSET_BITS) - 1u));
                    const int jj = row_of(idx, wx, wx_rcp);
                    const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
                    SolverArgsPtr Kb = opaque(K);
                    const IceStateIn S = kread(&Kb->S);
                    const double ua = Kb->E.u[k], va = Kb->E.v[k], Ta = Kb->E.T[k], pa = Kb->E.p[k], qa = Kb->E.q[k];
                    const double inv_Ta = frcp(Ta);
                    const double lam_a = liquid_fraction_fast(P, logt, Ta);
                    const AirState A = air_state_fast(P, pa, Ta, inv_Ta, qa, lam_a, svp_equil_fast(P, logt, Ta, inv_Ta, lam_a));
                    c.rho = A.rho;
                    c.cp = A.cp_m;
                    c.qav = A.q_vap;
                    c.Ls = P.LH_s0 + (P.cp_v - P.cp_i) * (Ta - P.T_0);
                    c.Ti = Ice.T_fw - Ice.liquidus_slope * Kb->O.S[k];
                    c.hk = fmax(S.thickness[k] * Ice.inv_k, Ice.hk_min);
                    const double alb = S.albedo ? S.albedo[k] : Ice.albedo;
                    c.Qd = -(1.0 - alb) * Kb->E.Qs[k] - Ice.emissivity * Kb->E.Ql[k];
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
                    c.dU2 = __builtin_fma(c.dU2, P.wind2_scale, P.wind2_add);  // (what enters the wind-speed scale: DevParams::wind2_*)
                    double alpha = P.rm.charnock;
                    if (P.rm.kind == CF_ROUGHNESS_WIND_CHARNOCK)
                        alpha = fmax(P.rm.charnock, P.rm.wind_a1 * fmin(c.dU, P.rm.wind_umax) + P.rm.wind_a2);
                    c.alpha_g = alpha * P.inv_g;
                    Ts = S.top_temperature[k] + Ice.T_offset;
                    // CF_OPT_ICE_FREE_CELLS = zero: open water (ℵ = 0 and hᵢ = 0) has no atmosphere–sea-ice interface
#ifndef CF_NO_ICE_FREE_OPTION  // (A/B builds: what the option's test costs the default mode — nothing measurable)
                    if (Ice.ice_free_zero != 0.0 && S.concentration) ice_free = S.concentration[k] == 0.0 && S.thickness[k] == 0.0;
#endif
                }
                Scales s{0.0, 0.0, 0.0, 0, 0};
                const bool solve = in_range && !ice_free;
                if (__ballot(solve) != 0ull) {  // (a batch of open water — most of the surface — skips the solve altogether)
                    if constexpr (SPEC == SOLVER_SEAICE_LEAN) {
                        const LeanIceConsts lc{c.rho, c.cp, c.qav, c.Ls, c.Ti, c.hk, c.Qd, c.theta_a, c.pa, frcp1(c.pa), frcp1(c.rho * P.R_v), c.dU2};
                        s = ice_iterate_lean<COARE>(P, L, Ice, lc, tab, solve, Ts);
                    } else {
                        s = ice_iterate<COARE>(P, L, Ice, c, tab, solve, Ts);
                    }
                }
                if (ice_free) s = Scales{0.0, 0.0, 0.0, 0, 0};  // zero_interface_state: no fluxes; the skin temperature stays the input
                if (in_range) {
                    SolverArgsPtr Ke = opaque(K);
                    const int idx2 = range_begin + (int)(list[qc] & ((1u << AO_LIST_OFFSET_BITS) - 1u));
                    const int jj2 = row_of(idx2, wx, wx_rcp);
                    const size_t k = cell_index(G, idx2 - jj2 * wx - G.ring, jj2 - G.ring);
                    CellFluxes R;
                    const double inv_dU = (c.dU == 0.0) ? 0.0 : frcp(c.dU);
                    const double tau = -s.us * s.us * inv_dU;
                    const double rho_u = c.rho * s.us;
                    R.Fv = -rho_u * s.qq;
                    R.Qv = R.Fv * c.Ls;
                    R.Qc = -rho_u * c.cp * s.ts;
                    R.rho_tau_x = c.rho * tau * c.du;
                    R.rho_tau_y = c.rho * tau * c.dv;
                    R.Ts_ocean = Ts - Ice.T_offset;
                    if (ice_free) R.Ts_ocean = kread(&Ke->S).top_temperature[k];  // (zero_interface_state: the input, bit for bit)
                    R.ustar = s.us;
                    R.tstar = s.ts;
                    R.qstar = s.qq;
                    R.iterations = s.it;
                    const FluxOut F = kread(&Ke->F);
                    store_fluxes(F, k, R);
                    {   // compute_net_sea_ice_fluxes! of the cell (interior only), when the launch carries it
                        const NetIceOut NI = kread(&Ke->NI);
                        const int ci = idx2 - jj2 * wx - G.ring, cj = jj2 - G.ring;
                        if (NI.top && ci >= 0 && ci < G.nx && cj >= 0 && cj < G.ny) {
                            const IceStateIn S2 = kread(&Ke->S);
                            double top, bottom;
                            net_sea_ice_cell(S2.albedo ? S2.albedo[k] : Ice.albedo, Ice.emissivity, Ice.eps_sigma, Ice.T_offset, Ke->E.Qs[k],
                                             Ke->E.Ql[k], R.Ts_ocean, R.Qc, R.Qv, NI.conc[k], NI.frazil ? NI.frazil[k] : 0.0,
                                             NI.interface_heat ? NI.interface_heat[k] : 0.0, top, bottom);
                            NI.top[k] = top;
                            NI.bottom[k] = bottom;
                        }
                    }
                    if (use_static && W.trip) store_hint(&W.trip[(size_t)chunk * CHUNK + (list[qc] >> AO_LIST_OFFSET_BITS)], s.work);
                }
                continue;
            }
            static_assert(SPEC != SOLVER_OCEAN_LEAN, "the ocean presets run in coflux_solver_lean.hip");
            CellConsts c;
            double So;
            {
                const int idx = range_begin + (int)(list[qc] & ((1u << AO_LIST_OFFSET_BITS) - 1u));
                const int jj = row_of(idx, wx, wx_rcp);
                const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
                SolverArgsPtr Kb = opaque(K);  // this batch's view of the arguments: pointers are (re)loaded here, scalar loads
            // ℑxᶜᵃᵃ u, ℑyᵃᶜᵃ v: cell-centre ocean velocity from the two bracketing faces
            const double* __restrict__ Ou = Kb->O.u;
            const double* __restrict__ Ov = Kb->O.v;
            const double uo = 0.5 * (Ou[k] + Ou[k + 1]);
            const double vo = 0.5 * (Ov[k] + Ov[k + (size_t)G.sj]);
                So = Kb->O.S[k];
                c = cell_prologue(P, L.min_gust, logt, Kb->E.u[k], Kb->E.v[k], Kb->E.T[k], Kb->E.p[k], Kb->E.q[k], uo, vo,
                                  Kb->O.T[k], So);
            }
            Scales s;
            if constexpr (SPEC == SOLVER_LY)
                s = ly_iterate(L, c, tab);
            else
                s = mo_iterate<COARE, SPEC>(L, c, tab, in_range);
            if (in_range) {
                SolverArgsPtr Ke = opaque(K);
                // (cell coordinates recomputed from the list entry: cheaper than four registers held across the iteration)
                const int idx2 = range_begin + (int)(list[qc] & ((1u << AO_LIST_OFFSET_BITS) - 1u));
                const int jj2 = row_of(idx2, wx, wx_rcp);
                const int ci = idx2 - jj2 * wx - G.ring, cj = jj2 - G.ring;
                const size_t k = cell_index(G, ci, cj);
                const CellFluxes R = cell_epilogue(c, P.T_offset, s);
                {
                    const FluxOut F = kread(&Ke->F);
                    store_fluxes<FUSE_NET>(F, k, R);
                }
                if (use_static && W.trip) W.trip[(size_t)chunk * CHUNK + (list[qc] >> AO_LIST_OFFSET_BITS)] = (uint8_t)min(s.work, 255);
                if constexpr (FUSE_NET) {
                    // compute_net_ocean_fluxes!, the part that needs no neighbour: interior cells only
                    if (ci >= 0 && ci < G.nx && cj >= 0 && cj < G.ny) {
                        const IceIn I = kread(&Ke->I);
                        const NetOut N = kread(&Ke->N);
                        store_net_cell<true>(N, k, net_cell_local(P, P.albedo, I.conc ? I.conc[k] : 0.0, So, R.Ts_ocean + P.T_offset,
                                                            Ke->E.Mp[k], Ke->E.Qs[k], Ke->E.Ql[k], R.Qc, R.Qv, R.Fv,
                                                            I.Qio ? I.Qio[k] : 0.0, I.Jsio ? I.Jsio[k] : 0.0, I.land ? I.land[k] : 0.0));
                    }
                }
            }
        }
        if (use_static || end >= range_end) break;  // no barrier at the end: a wave that runs out of batches retires
        begin = end;  // classification path: the rest of the range
        end = range_end;
        __syncthreads();  // list and counters are reused
        if (tid < 4) counters[tid] = 0;
        __syncthreads();
    }
}

template <bool COARE, int SPEC, bool FUSE_NET, int BLOCK, bool TAIL = false>
__global__ __launch_bounds__(BLOCK, 3) void ao_flux_fast_kernel(SolverArgs unused_by_name) {
    ao_flux_fast_body<COARE, SPEC, FUSE_NET, BLOCK, TAIL>((SolverArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), (int)blockIdx.x);
}

// Config 3 in ONE solver launch: the workgroups of the atmosphere–sea-ice interface solve (the long ones: ≈ 30 iterations per
// cell, 3 … 100) and of the ocean solve (coflux_lean_kernel.hpp, fused net-flux epilogue) in one dispatch order — the ocean
// solve's first arrival layer, every interface chunk, the rest of the ocean solve (it takes the slots the others free as they
// retire), the next step's interpolation (launch_ai_fluxes lays the segments out).  The two solves are independent of each other (both read the exchange fields and the ocean
// surface); two queues do not overlap them at all (scratch/two_ctx_overlap.py: 256.3 µs against 62.8 + 193.7), one launch does.
// The face stresses need the ocean solve's ρτ everywhere: a launch of their own behind this one.
struct IceOceanArgs {
    SolverArgs A;   // the interface solve (TAIL form: n_chunks, tail_blocks … as in ao_flux_fast_kernel; no stress blocks)
    LeanArgs O;     // the ocean solve
    long long ocean_chunks;
    // dispatch order: up to eight segments of (kind, first, count); kind 0 = interface chunks, 1 = ocean chunks, 2 = interpolation
    int nseg, seg_kind[8], seg_first[8], seg_count[8];
};
template <bool COARE_ICE, bool COARE_OCEAN>
__global__ __launch_bounds__(AO_BLOCK, 3) void ice_ocean_kernel(IceOceanArgs unused_by_name) {
    typedef const IceOceanArgs __attribute__((address_space(4)))* ArgsPtr;
    ArgsPtr K = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(K));
    int block = (int)blockIdx.x;
    const int nch = (int)K->A.n_chunks, noc = (int)K->ocean_chunks;
    (void)noc;
    int kind = 0, idx = 0;
    {
        int rest = block;
        const int ns = K->nseg;
        for (int q = 0; q < ns; ++q) {
            const int c = K->seg_count[q];
            if (rest < c) {
                kind = K->seg_kind[q];
                idx = K->seg_first[q] + rest;
                break;
            }
            rest -= c;
        }
    }
    if (kind == 1) {
        ao_lean_body<COARE_OCEAN, true>((LeanArgsPtr)&K->O, idx);
        return;
    }
    block = kind == 2 ? nch + idx : idx;  // the interpolation's workgroups: numbered as in ao_flux_fast_kernel's tail
    ao_flux_fast_body<COARE_ICE, SOLVER_SEAICE_LEAN, false, AO_BLOCK, true>((SolverArgsPtr)&K->A, block);
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


template <bool COARE, bool FUSE>
static void launch_ao_spec(hipStream_t st, dim3 grid, const LaunchCfg& L, const LoopParams& C, const GridDesc& G,
                           const OceanIn& O, const Exchange& E, const FluxOut& F, const IceIn& I, const NetOut& N,
                           double z_surface, long long mask_kind, double T_offset) {
    // (CoefficientBasedFluxes runs a fixed trip count: no hint bytes to read, sort by or write back)
    const SolverArgs A{C, G, O, E, F, L.d_tables, L.d_params, WetLists{L.d_wet_pos, C.specialization == SOLVER_LY ? nullptr : L.d_trip},
                       L.d_chunk_begins, I, N, IceStateIn{}, IceParams{},
                       z_surface, mask_kind, T_offset, row_reciprocal(G.nx + 2 * G.ring)};
#define CF_LAUNCH(COARE_, SPEC_) \
    hipLaunchKernelGGL((ao_flux_fast_kernel<COARE_, SPEC_, FUSE, AO_BLOCK>), grid, dim3(AO_BLOCK), Geom<AO_BLOCK>::LDS_BYTES, st, A)
    // (SOLVER_OCEAN_LEAN never arrives here: launch_ao_fluxes hands it to coflux_solver_lean.hip)
    switch (C.specialization) {
        case SOLVER_ICE: CF_LAUNCH(COARE, SOLVER_ICE); break;
        case SOLVER_LY: CF_LAUNCH(true, SOLVER_LY); break;
        default: CF_LAUNCH(COARE, SOLVER_GENERIC);
    }
#undef CF_LAUNCH
}

// CoefficientBasedFluxes with the fused net fluxes AND the next step's interpolation in tail workgroups (narrow geometry)
hipError_t launch_ly_fluxes_with_tail(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G,
                                      const cf_ocean_surface* o, const cf_exchange_fields* e, const cf_interface_fluxes* f,
                                      const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net, const double* land,
                                      const cf_atmos_source* next_src, const cf_interp_weights* w, const cf_exchange_fields* next_out,
                                      int tail_rows, int tail_blocks) {
    if (!L.d_chunk_begins || L.n_chunks <= 0 || !net || !next_src || !w || !next_out || L.interp_cap <= 0 || tail_blocks <= 0 ||
        C.specialization != SOLVER_LY)
        return hipErrorInvalidValue;
    if ((size_t)IT_WAVES * CF_JRA55_NVARS * L.interp_cap * sizeof(double) > (size_t)Geom<AO_BLOCK>::LDS_BYTES) return hipErrorInvalidValue;
    IceIn I{};
    if (ice) I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress, nullptr};
    I.land = land;
    const NetOut N{net->u, net->v, net->T, net->S, net->shortwave_surface_flux, net->upwelling_longwave, net->downwelling_longwave,
                   net->downwelling_shortwave};
    SolverArgs A{C, G, make_ocean(o), make_exchange(e), make_fluxes(f), L.d_tables, L.d_params, WetLists{L.d_wet_pos, nullptr},
                 L.d_chunk_begins, I, N, IceStateIn{}, IceParams{}, P.z_surface, P.mask_kind, P.T_offset, row_reciprocal(G.nx + 2 * G.ring)};
    A.Si = make_source(next_src);
    A.Wi = make_weights(w);
    A.E_next = make_exchange(next_out);
    A.n_chunks = L.n_chunks;
    A.tail_blocks = tail_blocks;
    A.tail_rows = tail_rows;
    A.tail_cap = L.interp_cap;
    hipLaunchKernelGGL((ao_flux_fast_kernel<true, SOLVER_LY, true, AO_BLOCK, true>), dim3(L.n_chunks + tail_blocks), dim3(AO_BLOCK),
                       Geom<AO_BLOCK>::LDS_BYTES, st, A);
    return hipGetLastError();
}

// `net` != nullptr: the fused form — the solver's epilogue also writes the cell-local net ocean fluxes (JT, JS,
// shortwave, diagnostics; constant ocean albedo only), leaving the face stresses to launch_net_stress.
hipError_t launch_ao_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C,
                            const GridDesc& G, const cf_ocean_surface* o, const cf_exchange_fields* e,
                            const cf_interface_fluxes* f, const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net,
                            const double* land) {
    if (L.solver == CF_SOLVER_LIBM) return net ? hipErrorInvalidValue : launch_ao_fluxes_libm(st, P, G, o, e, f);
    // the production ocean configurations: the round-3 kernel
    if (C.specialization == SOLVER_OCEAN_LEAN) {
        if (!L.d_lean_info) return hipErrorInvalidValue;
        return launch_ao_fluxes_lean(st, L, P, C, G, o, e, f, ice, net, land);
    }
    OceanIn O = make_ocean(o);
    Exchange E = make_exchange(e);
    FluxOut F = make_fluxes(f);
    IceIn I{};
    if (ice) I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress, nullptr};
    I.land = land;
    NetOut N{};
    if (net)
        N = NetOut{net->u, net->v, net->T, net->S, net->shortwave_surface_flux, net->upwelling_longwave,
                   net->downwelling_longwave, net->downwelling_shortwave};
    // one workgroup per chunk of the cost-balanced table: the hardware dispatcher is the dynamic load balancer
    if (!L.d_chunk_begins || L.n_chunks <= 0) return hipErrorInvalidValue;
    dim3 grid(L.n_chunks);
    const bool coare = P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC;
    if (net) {
        if (coare) launch_ao_spec<true, true>(st, grid, L, C, G, O, E, F, I, N, P.z_surface, P.mask_kind, P.T_offset);
        else launch_ao_spec<false, true>(st, grid, L, C, G, O, E, F, I, N, P.z_surface, P.mask_kind, P.T_offset);
    } else {
        if (coare) launch_ao_spec<true, false>(st, grid, L, C, G, O, E, F, I, N, P.z_surface, P.mask_kind, P.T_offset);
        else launch_ao_spec<false, false>(st, grid, L, C, G, O, E, F, I, N, P.z_surface, P.mask_kind, P.T_offset);
    }
    return hipGetLastError();
}

// compute_atmosphere_sea_ice_fluxes!: the same kernel machinery (static lists, LDS-DMA tables, batches sorted by the
// previous call's trip counts) with the sea-ice iteration; its own formulation block, tables and trip-count array.
hipError_t launch_ai_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const IceParams& Ice,
                            const GridDesc& G, const cf_sea_ice_state* ice, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f, const double* d_tables,
                            const DevParams* d_params, uint8_t* d_trip, const AiTail* tail, const NetIceOut* net_ice) {
    if (!L.d_chunk_begins || L.n_chunks <= 0) return hipErrorInvalidValue;
    SolverArgs A{};
    A.L = C;
    A.G = G;
    A.O = make_ocean(o);
    A.E = make_exchange(e);
    A.F = make_fluxes(f);
    A.g_tab = d_tables;
    A.g_params = d_params;
    A.W = WetLists{L.d_wet_pos, d_trip};
    A.chunk_begins = L.d_chunk_begins;
    A.S = IceStateIn{ice->thickness, ice->top_temperature, ice->u, ice->v, ice->albedo, Ice.ice_free_zero != 0.0 ? ice->concentration : nullptr};
    A.Ice = Ice;
    A.z_surface = P.z_surface;
    A.mask_kind = P.mask_kind;
    A.T_offset = P.T_offset;
    A.wx_reciprocal = row_reciprocal(G.nx + 2 * G.ring);
    dim3 grid(L.n_chunks);
    const bool coare = P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC;
    // constant roughness lengths and a gustiness floor (both production presets): the lean iteration body
    const bool lean = C.specialization == SOLVER_ICE && L.solver == CF_SOLVER_TABLES;
    if (net_ice) A.NI = *net_ice;
    if (tail) {
        // tail workgroups behind the interface solve's (lean iteration, narrow geometry): the next step's interpolation and / or
        // this step's face stresses
        if (!lean) return hipErrorInvalidValue;
        A.n_chunks = L.n_chunks;
        if (tail->next_out) {
            if (!tail->next_src || !tail->w || L.interp_cap <= 0 || tail->interp_blocks <= 0 ||
                (size_t)IT_WAVES * CF_JRA55_NVARS * L.interp_cap * sizeof(double) > (size_t)Geom<AO_BLOCK>::LDS_BYTES)
                return hipErrorInvalidValue;
            A.Si = make_source(tail->next_src);
            A.Wi = make_weights(tail->w);
            A.E_next = make_exchange(tail->next_out);
            A.tail_blocks = tail->interp_blocks;
            A.tail_rows = tail->interp_rows;
            A.tail_cap = L.interp_cap;
        }
        if (tail->stress_net) {
            if (!tail->d_ocean_params || !tail->stress_ocean || !tail->stress_fluxes) return hipErrorInvalidValue;
            A.stress_blocks = (G.nx * G.ny + 255) / 256;
            A.stress_params = tail->d_ocean_params;
            A.stress_mask = tail->stress_ocean->mask;
            A.rtx = tail->stress_fluxes->x_momentum;
            A.rty = tail->stress_fluxes->y_momentum;
            if (tail->stress_ice)
                A.stress_ice = IceIn{tail->stress_ice->concentration, tail->stress_ice->interface_heat, tail->stress_ice->salt_flux,
                                     tail->stress_ice->x_stress, tail->stress_ice->y_stress, nullptr};
            A.tau_x = tail->stress_net->u;
            A.tau_y = tail->stress_net->v;
        }
        if (tail->ocean) {
            // the ocean solve's workgroups behind the interface solve's, the interpolation's behind those (ice_ocean_kernel)
            if (!tail->ocean->valid || tail->stress_net || tail->ocean->n_chunks <= 0) return hipErrorInvalidValue;
            static_assert(sizeof(LeanArgs) <= sizeof(tail->ocean->args), "OceanRider::args holds a LeanArgs");
            IceOceanArgs M{};
            M.A = A;
            memcpy(&M.O, tail->ocean->args, sizeof(LeanArgs));
            M.ocean_chunks = tail->ocean->n_chunks;
            {
                // (experiments: COFLUX_EXPERIMENTS=1 COFLUX_ICE_OCEAN_ORDER="I0:768,O0:768,T" — segments of interface / ocean chunks
                // first:count and the interpolation, in dispatch order; scratch/ice_ocean_orders.sh)
                static const char* order = experiment_knob("COFLUX_ICE_OCEAN_ORDER");
                int n = 0;
                if (order && order[0]) {
                    const char* c = order;
                    while (*c && n < 8) {
                        const char k = *c++;
                        if (k == 'T') {
                            M.seg_kind[n] = 2; M.seg_first[n] = 0; M.seg_count[n] = (int)A.tail_blocks;
                        } else {
                            M.seg_kind[n] = k == 'O' ? 1 : 0;
                            M.seg_first[n] = (int)strtol(c, (char**)&c, 10);
                            if (*c == ':') ++c;
                            M.seg_count[n] = (int)strtol(c, (char**)&c, 10);
                            const int total = k == 'O' ? tail->ocean->n_chunks : L.n_chunks;  // (the knob speaks of 768 chunks: clamp)
                            M.seg_first[n] = std::min(M.seg_first[n], total);
                            M.seg_count[n] = std::min(M.seg_count[n], total - M.seg_first[n]);
                        }
                        ++n;
                        if (*c == ',') ++c;
                    }
                } else {
                    // the ocean solve's FIRST arrival layer (its largest chunks, one per CU) ahead of the interface solve's workgroups,
                    // the rest behind them: the riders left for the end are the short ones (249.1 µs per step against 255.2 with every
                    // ocean chunk behind; its smallest layer first 252.4; profiles/r04_experiments.md §17)
                    const int noc_all = tail->ocean->n_chunks, head = std::min(std::max(L.cu_count, 0), noc_all);
                    M.seg_kind[0] = 1; M.seg_first[0] = 0; M.seg_count[0] = head;
                    M.seg_kind[1] = 0; M.seg_first[1] = 0; M.seg_count[1] = L.n_chunks;
                    M.seg_kind[2] = 1; M.seg_first[2] = head; M.seg_count[2] = noc_all - head;
                    M.seg_kind[3] = 2; M.seg_first[3] = 0; M.seg_count[3] = (int)A.tail_blocks;
                    n = 4;
                }
                M.nseg = n;
            }
            const dim3 mgrid((unsigned)(L.n_chunks + tail->ocean->n_chunks + A.tail_blocks));
            constexpr size_t lds = (size_t)(Geom<AO_BLOCK>::LDS_BYTES > LeanGeom<AO_BLOCK>::LDS_BYTES ? Geom<AO_BLOCK>::LDS_BYTES : LeanGeom<AO_BLOCK>::LDS_BYTES);
            static_assert(lds <= 53760, "three workgroups per CU");
            if (coare) {
                if (tail->ocean->coare) hipLaunchKernelGGL((ice_ocean_kernel<true, true>), mgrid, dim3(AO_BLOCK), lds, st, M);
                else hipLaunchKernelGGL((ice_ocean_kernel<true, false>), mgrid, dim3(AO_BLOCK), lds, st, M);
            } else {
                if (tail->ocean->coare) hipLaunchKernelGGL((ice_ocean_kernel<false, true>), mgrid, dim3(AO_BLOCK), lds, st, M);
                else hipLaunchKernelGGL((ice_ocean_kernel<false, false>), mgrid, dim3(AO_BLOCK), lds, st, M);
            }
            return hipGetLastError();
        }
        const dim3 tgrid((unsigned)(L.n_chunks + A.tail_blocks + A.stress_blocks));
        if (coare) hipLaunchKernelGGL((ao_flux_fast_kernel<true, SOLVER_SEAICE_LEAN, false, AO_BLOCK, true>), tgrid, dim3(AO_BLOCK), Geom<AO_BLOCK>::LDS_BYTES, st, A);
        else hipLaunchKernelGGL((ao_flux_fast_kernel<false, SOLVER_SEAICE_LEAN, false, AO_BLOCK, true>), tgrid, dim3(AO_BLOCK), Geom<AO_BLOCK>::LDS_BYTES, st, A);
        return hipGetLastError();
    }
#define CF_AI_LAUNCH(COARE_, SPEC_, BLOCK_) \
    hipLaunchKernelGGL((ao_flux_fast_kernel<COARE_, SPEC_, false, BLOCK_>), grid, dim3(BLOCK_), Geom<BLOCK_>::LDS_BYTES, st, A)
    if (lean) { if (coare) CF_AI_LAUNCH(true, SOLVER_SEAICE_LEAN, AO_BLOCK); else CF_AI_LAUNCH(false, SOLVER_SEAICE_LEAN, AO_BLOCK); }
    else { if (coare) CF_AI_LAUNCH(true, SOLVER_SEAICE, AO_BLOCK); else CF_AI_LAUNCH(false, SOLVER_SEAICE, AO_BLOCK); }
#undef CF_AI_LAUNCH
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
