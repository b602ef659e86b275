// coflux_lean_kernel.hpp — the round-3 ocean solver kernel (coflux_solver_lean.hip) as a device routine, so that its
// workgroups can also ride in another launch (the ocean solve behind the sea-ice interface solve's workgroups,
// coflux_solver.hip::ice_ocean_kernel).  One body: the same bits wherever it runs.
#pragma once
#include <hip/hip_runtime.h>

#include "coflux_halo_device.hpp"
#include "coflux_interp_tiles.hpp"
#include "coflux_lean.hpp"
#include "coflux_certified.hpp"
#include "coflux_solver_shared.hpp"

#ifndef CF_LEAN_WAVES
#define CF_LEAN_WAVES 3  // waves per SIMD the narrow lean kernel is compiled for (4: ≤ 128 VGPRs; needs ≤ 40 960 B of LDS per workgroup)
#endif
#ifndef CF_LEAN_PREFETCH
#define CF_LEAN_PREFETCH 0  // 1: the next batch's inputs are requested before the current batch iterates (measured: 23 more live registers cost more than the wait, 70.3 vs 68.3 us)
#endif

#ifndef LEAN_STAMP  // (per-wave time stamps: coflux_solver_lean.hip under -DCF_LEAN_STAMPS)
#define LEAN_STAMP(q) do { } while (0)
#define LEAN_STAMP_SET(q, v) do { } while (0)
#endif

namespace coflux {

struct LeanArgs {
    LoopParams L;
    GridDesc G;
    OceanIn O;
    Exchange E;
    FluxOut F;
    const double* g_tab;
    const DevParams* g_params;
    uint32_t* sorted;         // [chunk·AO_CHUNK + p]: offset of the p-th cell (longest trip count first) in the chunk's range; 0xffffffff-padded
    const int* info;          // [chunk·4]: wet cells listed, fingerprint x, fingerprint y, 0
    const int* chunk_begins;
    double z_surface;         // with mask_kind: how the start phase reads wetness before the parameter block is in LDS
    long long mask_kind;
    double T_offset;
    unsigned long long wx_reciprocal;
    long long sort_enabled;   // CF_OPT_TRIP_HINTS
    IceIn I;                  // fused net fluxes only
    NetOut N;
    SourceDesc S;             // tail workgroups: the JRA55 window and the weights of interpolate_atmosphere_state!
    WeightDesc Wt;
    Exchange E_next;          // TAIL: the exchange fields of the NEXT step (another set than E)
    long long n_chunks;       // TAIL: workgroups [0, n_chunks) solve, the tail_blocks behind them interpolate
    long long tail_blocks, tail_rows, tail_cap;
    long long tail_pos;       // TAIL: index of the first interpolation workgroup in dispatch order (n_chunks: behind every solver workgroup)
    HaloRider H;              // HALO: the peer-direct halo rows of this step as rider workgroups at the head of the launch
};
typedef const LeanArgs __attribute__((address_space(4)))* LeanArgsPtr;

// coflux_solver_slab.hip: the exact path's kernels in the one-wave-per-SIMD layout (COARE × plain / fused net fluxes / + tail workgroups)
hipError_t launch_ao_lean_line(hipStream_t st, bool coare, bool fuse, bool tail, bool halo, int blocks, const LeanArgs& A);

__device__ __forceinline__ LeanArgsPtr opaque(LeanArgsPtr p) {
    asm volatile("" : "+s"(p));
    return p;
}

// Fingerprint of a set of cells (window-linear indices): XOR of a 32-bit mix and a wrapping sum of a multiple.
// Only a staleness detector for a mask rewritten in place — six instructions per cell, not a hash against adversaries.
__device__ __forceinline__ unsigned lean_mix(unsigned idx) { return (idx * 0x9e3779b1u) ^ ((idx * 0x85ebca6bu) >> 15); }
__device__ __forceinline__ unsigned lean_sum(unsigned idx) { return idx * 0xc2b2ae35u + 0x27d4eb2fu; }

// LDS: tables | list (sorted cell offsets; a batch's epilogue leaves this call's trip count in the top byte of its
// entries) | histogram | cursors | counters | per-wave fingerprints | DevParams.  Narrow geometry: 52.4 KB — a
// workgroup may use at most 53 760 B for three to fit a CU (the allocation granule eats the rest of 160 KB / 3).
constexpr int LEAN_LIST_OFFSET = TABLE_BYTES;
constexpr int LEAN_OFFSET_BITS = 24;
constexpr unsigned LEAN_OFFSET_MASK = (1u << LEAN_OFFSET_BITS) - 1u;
static_assert((long)AO_CHUNK * AO_WET_COST < (1L << LEAN_OFFSET_BITS), "a chunk's range must fit the offset bits");
template <int BLOCK>
struct LeanGeom {
    static_assert(BLOCK == AO_BLOCK, "one workgroup geometry: 256 threads, three workgroups per CU (the 768-thread one-per-CU geometry of rounds 2-5 measured 3 % slower and was retired in round 6)");
    static constexpr int CHUNK = AO_CHUNK;
    static constexpr int WAVES = BLOCK / 64;
    static constexpr int HIST_OFFSET = LEAN_LIST_OFFSET + CHUNK * 4;
    static constexpr int CURSOR_OFFSET = HIST_OFFSET + AO_BINS * 4;
    static constexpr int COUNTERS_OFFSET = CURSOR_OFFSET + AO_BINS * 4;  // [0] batch cursor, [2] wet count of the classification path
    static constexpr int WAVEHASH_OFFSET = COUNTERS_OFFSET + 16;
    static constexpr int PARAMS_OFFSET = WAVEHASH_OFFSET + WAVES * 8;
    static constexpr int LDS_BYTES = PARAMS_OFFSET + (int)sizeof(DevParams);
    static_assert(PARAMS_OFFSET % 16 == 0 && LEAN_LIST_OFFSET % 1024 == 0 && (CHUNK * 4) % 1024 == 0, "LDS-DMA pieces");
};
static_assert(LeanGeom<AO_BLOCK>::LDS_BYTES <= 53760, "three narrow lean solver workgroups must fit the CU's 160 KB of LDS");
#ifndef CF_SKIP_LDS_ASSERT
static_assert(CF_LEAN_WAVES <= 3 || LeanGeom<AO_BLOCK>::LDS_BYTES <= 40960, "four narrow lean workgroups per CU: 160 KB / 4 in 1280-byte granules");
#endif

// zero_interface_state of a land cell: all fluxes 0, T = 0 K (and, in the fused form, zero net fluxes inside the interior)
template <bool FUSE>
__device__ __forceinline__ void lean_zero_cell(const LoopParams& L, double T_offset, const GridDesc& G, LeanArgsPtr K, size_t k, int i, int j) {
    CellFluxes Z{};
    Z.Ts_ocean = -T_offset;
    Z.iterations = L.fixed ? L.maxiter : 0;
    const FluxOut F = kread(&K->F);
    store_fluxes(F, k, Z);
    if constexpr (FUSE) {
        if (i >= 0 && i < G.nx && j >= 0 && j < G.ny) {
            const NetOut N = kread(&K->N);
            store_net_cell(N, k, NetCell{});
        }
    }
}

// FUSE: the cell-local part of compute_net_ocean_fluxes! (everything but the two face stresses, which need the west /
// south neighbour's ρτ: launch_net_stress) in the epilogue — the same arithmetic as net_flux_kernel, bit for bit
// (net_cell_local, contraction off).  With batches in index order its nine extra accesses per cell are coalesced.
// (Rounds 3-5 also carried interpolate_atmosphere_state! in the batch prologue — CF_OPT_FUSED_INTERP, bitwise the separate
// launch, 0.1007 vs 0.0956 ms per step: 72 scattered gathers per cell inside the FP64-bound kernel — retired in round 6.)
// TAIL: the launch carries tail_blocks more workgroups BEHIND the solver's (dispatch follows the workgroup index, so they
// take the slots the solver's workgroups free as they retire): they interpolate the NEXT step's atmosphere state into the
// other set of exchange fields with the tiled routine of interpolate_kernel — memory-bound work under the solver's
// FP64-bound tail instead of a launch of its own in front of the next solver (cf_time_steps with two exchange sets).
// CERT: the certified reduced-iteration solve (coflux_certified.hpp).  A batch runs mo_iterate_certified; the lanes it
// does not certify (≈ 1.5 % of the cells) put their list position into a queue in LDS (the trip-count histogram's space:
// this mode does not sort) and skip the epilogue.  The first wave of the workgroup to run out of batches closes the queue
// and works it off as one batch of the exact iteration (mo_iterate_lean) — compaction: sixteen batches' cells in one —;
// cells that fail their certificate after the close, entries beyond the queue's capacity, and a workgroup's first
// uncertified cells when its list is already handed out run the exact iteration in their own batch at once.
#ifndef CF_CERT_NET_SALT
#define CF_CERT_NET_SALT 1  // (0: A/B builds without the certificate's net-salinity term)
#endif
// LINE: the iteration in its one-wave-per-SIMD layout (mo_iterate_lean_line; coflux_solver_slab.hip builds those kernels
// through tools/gcn_sched.py); bitwise the same results.
// HALO (with TAIL, exact path): the step's peer-direct halo rows inside this launch.  The first
// H.blocks workgroups are the exchange's riders (peer_halo_rider: one per direction and field); the chunks whose cells read
// halo rows — the south ring row; the last interior row and the north ring row — are dispatched BEHIND every other chunk and
// wait, after their own start phase, for the riders' counter: interior chunks never wait, and the neighbours' latency
// passes under interior work instead of in front of the launch.  The same rows arrive by the same protocol: bitwise the step
// with the stand-alone exchange kernel (tests/test_steps.py).
template <bool COARE, bool FUSE, bool TAIL = false, bool CERT = false, bool LINE = false, bool HALO = false>
__device__ __forceinline__ void ao_lean_body(LeanArgsPtr K_in, int chunk_in) {
    constexpr int BLOCK = AO_BLOCK;
    using Geo = LeanGeom<BLOCK>;
    constexpr int CHUNK = Geo::CHUNK;
    LeanArgsPtr K = opaque(K_in);
    int chunk = chunk_in;  // dispatch order = layer order of the chunk table
    bool wait_south = false, wait_north = false;
    if constexpr (HALO) {
        static_assert(TAIL && !CERT, "the halo riders belong to the stepping loop's exact-path launch");
        const int nh = K->H.blocks;
        if (chunk < nh) {
            extern __shared__ __attribute__((aligned(16))) char smem_h[];
            const int nf = K->H.F.n, dir = chunk / nf, f = chunk - dir * nf;   // (uniform: scalar loads with a computed offset)
            const PeerMailbox M = kread(&K->H.M);
            const GridDesc Gh = kread(&K->G);
            peer_halo_rider(M, K->H.F.ptr[f], f, dir, K->H.rows, K->H.seq, K->H.counters, K->H.expect_sent[dir], K->H.status, Gh, BLOCK,
                            reinterpret_cast<int*>(smem_h));
            return;
        }
        chunk -= nh;
    }
    if constexpr (TAIL) {
        static_assert(BLOCK == 64 * IT_WAVES, "a tail workgroup is an interpolation workgroup");
        const int nb = (int)K->tail_blocks, ipos = (int)K->tail_pos;
        if (chunk >= ipos && chunk < ipos + nb) {
            const SourceDesc S = kread(&K->S);
            const WeightDesc Wt = kread(&K->Wt);
            const GridDesc Gt = kread(&K->G);
            const Exchange En = kread(&K->E_next);
            const int b = chunk - ipos, cap = (int)K->tail_cap, rows = (int)K->tail_rows;
            if (rows == 4) interpolate_tiles<4>(S, Wt, Gt, En, cap, b, nb);
            else if (rows == 2) interpolate_tiles<2>(S, Wt, Gt, En, cap, b, nb);
            else interpolate_tiles<1>(S, Wt, Gt, En, cap, b, nb);
            return;
        }
        if (chunk >= ipos) chunk -= nb;
    }
    if constexpr (HALO) {
        // dispatch position → chunk: the chunks between the two boundary sets first, then the south set, then the north set
        const int cs = K->H.chunk_south, cn = K->H.chunk_north, nc = (int)K->n_chunks;
        if (K->H.blocks > 0) {
            if (cs <= cn) {
                const int inner = cn - cs;
                chunk = chunk < inner ? cs + chunk : (chunk - inner < cs ? chunk - inner : cn + (chunk - inner - cs));
            }
            wait_south = K->H.wait_south != 0 && chunk < cs;
            wait_north = K->H.wait_north != 0 && chunk >= cn && chunk < nc;
        }
    }
    const LoopParams L = kread(&K->L);
    const GridDesc G = kread(&K->G);
    const double* __restrict__ g_tab = K->g_tab;
    const DevParams* __restrict__ g_params = K->g_params;
    const void* mask = K->O.mask;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tab = reinterpret_cast<double*>(smem);
    unsigned* list = reinterpret_cast<unsigned*>(smem + LEAN_LIST_OFFSET);
    int* hist = reinterpret_cast<int*>(smem + Geo::HIST_OFFSET);
    int* cursor = reinterpret_cast<int*>(smem + Geo::CURSOR_OFFSET);
    int* counters = reinterpret_cast<int*>(smem + Geo::COUNTERS_OFFSET);
    unsigned* wavehash = reinterpret_cast<unsigned*>(smem + Geo::WAVEHASH_OFFSET);
    DevParams* lp = reinterpret_cast<DevParams*>(smem + Geo::PARAMS_OFFSET);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LEAN_STAMP(0);
    const int wx = G.nx + 2 * G.ring;
    const unsigned wx_rcp = (unsigned)K->wx_reciprocal;
    const bool use_static = K->sorted != nullptr;
    unsigned long long stamp_iter = 0, stamp_batches = 0, stamp_trips = 0;
    // ---- start phase: everything is REQUESTED before anything is looked at, in straight-line code ----------------
    // (1) LDS-DMA (global_load_lds, 1 KB per wave instruction, no VGPR round trip): the parameter block, the tables,
    //     the chunk's sorted list; (2) the raw mask words of my share of the chunk's cell range (fingerprint, land).
    static_assert(sizeof(DevParams) % 16 == 0 && sizeof(DevParams) <= 1024, "the parameter block is one LDS-DMA piece");
    if (tid < (int)(sizeof(DevParams) / 16))
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g_params) + lane * 16),
                                         (__attribute__((address_space(3))) void*)(smem + Geo::PARAMS_OFFSET), 16, 0, 0);
    static_assert(TABLE_BYTES % 1024 == 0, "the table stage copies whole 1 KB pieces");
    {
        const char* gb = reinterpret_cast<const char*>(g_tab);
        constexpr int PIECES = TABLE_BYTES / 1024, WAVES = Geo::WAVES;
#pragma unroll
        for (int r = 0; r < (PIECES + WAVES - 1) / WAVES; ++r) {
            const int c = wave + r * WAVES;
            if (c < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + c * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + c * 1024), 16, 0, 0);
        }
    }
    if (use_static) {
        const char* gl = reinterpret_cast<const char*>(K->sorted + (size_t)chunk * CHUNK);
        constexpr int PIECES = CHUNK * 4 / 1024, WAVES = Geo::WAVES;
#pragma unroll
        for (int r = 0; r < (PIECES + WAVES - 1) / WAVES; ++r) {
            const int c = wave + r * WAVES;
            if (c < PIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gl + c * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(smem + LEAN_LIST_OFFSET + c * 1024), 16, 0, 0);
        }
    }
    constexpr int LAND_UNROLL = 8;  // strips of the range whose mask values are requested up front
    typedef const __attribute__((address_space(1))) unsigned* GlobalWords;
    unsigned raw_lo[LAND_UNROLL], raw_hi[LAND_UNROLL], raw_shift = 0;  // raw_shift: 2 bits per strip, the byte within its word
    // (scalar loads through the constant address space: a vector load here would sit behind the DMA in the in-order
    // vector-memory queue)
    const __attribute__((address_space(4))) int* cb = (const __attribute__((address_space(4))) int*)K->chunk_begins;
    const int range_begin = cb[chunk], range_end = cb[chunk + 1];
    int listed = 0;
    unsigned want_x = 0, want_y = 0;
    if (use_static) {
        const __attribute__((address_space(4))) int* ci = (const __attribute__((address_space(4))) int*)K->info;
        listed = ci[chunk * 4];
        want_x = (unsigned)ci[chunk * 4 + 1];
        want_y = (unsigned)ci[chunk * 4 + 2];
    }
    const int mask_kind = (mask == nullptr) ? CF_MASK_NONE : (int)K->mask_kind;
    const double z_surface = K->z_surface;
    const double T_offset = K->T_offset;
    const bool sorting = !CERT && use_static && K->sort_enabled != 0;
    // sort_enabled = number of WINDOWS the chunk's list is sorted in: 1 = the whole chunk by trip count (64 bins);
    // 4 = each quarter of the list separately (16 one-count bins each): a batch's cells then stay within a quarter of the
    // chunk's range — a fourth of the lines per access a whole-chunk sort touches — and still run together
    const int sort_windows = (int)K->sort_enabled;
    float inv_window = 0.f;
    auto lean_bin = [&](int q, int trips) -> int {  // ascending bin = taken first; q: position in the current list
        if (sort_windows <= 1) return AO_BINS - 1 - trip_bin(trips);
        const int w = min((int)(((float)q + 0.5f) * inv_window), sort_windows - 1);
        const int bpw = AO_BINS / sort_windows, lo = bpw >= 16 ? 5 : 8;  // one-count bins from `lo` iterations up
        return w * bpw + (bpw - 1 - min(max(trips - lo, 0), bpw - 1));
    };
    {
        const unsigned long long mbase = mask_kind == CF_MASK_NONE ? (unsigned long long)g_tab : (unsigned long long)mask;
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
    if constexpr (CERT) {  // the exact-path queue (hist + cursor): every entry invalid until its producer has written it
        if (tid < 2 * AO_BINS) hist[tid] = -1;
    }
    LEAN_STAMP(1);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): my mask words and my share of the DMA have landed
    LEAN_STAMP(2);
#pragma unroll
    for (int n = 0; n < LAND_UNROLL; ++n) asm volatile("" : "+v"(raw_lo[n]), "+v"(raw_hi[n]));
    // fingerprint of the range's wet set as the mask is NOW; land gets its zeros behind the barrier
    unsigned hx = 0, hy = 0, land = 0;
#pragma unroll
    for (int n = 0; n < LAND_UNROLL; ++n) {
        const int idx = range_begin + tid + n * BLOCK;
        const bool w = mask_kind == CF_MASK_NONE ? true
                       : mask_kind == CF_MASK_U8 ? ((raw_lo[n] >> (8 * ((raw_shift >> (2 * n)) & 3u))) & 0xffu) != 0
                                                 : !(z_surface <= __hiloint2double((int)raw_hi[n], (int)raw_lo[n]));
        if (idx < range_end) {
            if (w) {
                hx ^= lean_mix((unsigned)idx);
                hy += lean_sum((unsigned)idx);
            } else {
                land |= 1u << n;
            }
        }
    }
    // a range longer than LAND_UNROLL strips (a chunk that is mostly land): the rest the plain way, zeros at once
    for (int idx = range_begin + tid + LAND_UNROLL * BLOCK; idx < range_end; idx += BLOCK) {
        const int jj = row_of(idx, wx, wx_rcp);
        const size_t k = cell_index(G, idx - jj * wx - G.ring, jj - G.ring);
        const bool w = mask_kind == CF_MASK_NONE ? true
                       : mask_kind == CF_MASK_U8 ? ((const uint8_t*)mask)[k] != 0 : !(z_surface <= ((const double*)mask)[k]);
        if (w) {
            hx ^= lean_mix((unsigned)idx);
            hy += lean_sum((unsigned)idx);
        } else {
            lean_zero_cell<FUSE>(L, T_offset, G, opaque(K), k, idx - jj * wx - G.ring, jj - G.ring);
        }
    }
    for (int d = 32; d; d >>= 1) {
        hx ^= (unsigned)__shfl_xor((int)hx, d);
        hy += (unsigned)__shfl_xor((int)hy, d);
    }
    if (lane == 0) {
        wavehash[2 * wave] = hx;
        wavehash[2 * wave + 1] = hy;
    }
    __syncthreads();  // the ONE barrier of the start phase: tables, parameters, list and fingerprints are in LDS
    LEAN_STAMP(3);
    if constexpr (HALO) {
        // a boundary chunk: the halo rows its first batch is about to load (nothing before this point reads the ocean state)
        if (wait_south) peer_halo_wait(&opaque(K)->H.counters[2], K->H.expect_done[0]);
        if (wait_north) peer_halo_wait(&opaque(K)->H.counters[3], K->H.expect_done[1]);
    }
    const DevParams& P = *lp;
    bool have_list = false;
    int nwet = 0;
    if (use_static) {
        unsigned got_x = 0, got_y = 0;
#pragma unroll
        for (int w = 0; w < Geo::WAVES; ++w) {
            got_x ^= wavehash[2 * w];
            got_y += wavehash[2 * w + 1];
        }
        have_list = got_x == want_x && got_y == want_y;  // the list is the range's wet set
        nwet = listed;
        if (sort_windows > 1 && nwet > 0)  // windows of whole batches
            inv_window = 1.0f / (float)(((nwet + sort_windows * 64 - 1) / (sort_windows * 64)) * 64);
    }
    if (have_list && land) {
        // zero_interface_state of the range's land: nothing waits for these stores but the first batch's loads
        LeanArgsPtr Kz = opaque(K);
#pragma unroll
        for (int n = 0; n < LAND_UNROLL; ++n)
            if (land & (1u << n)) {
                const int idx = range_begin + tid + n * BLOCK;
                const int jj = row_of(idx, wx, wx_rcp);
                const int i = idx - jj * wx - G.ring, j = jj - G.ring;
                lean_zero_cell<FUSE>(L, T_offset, G, Kz, cell_index(G, i, j), i, j);
            }
    }
    int begin = range_begin, end = range_end;
    for (;;) {
        if (!have_list) {
            // ---- no (valid) sorted list: classify the piece [begin, end), zero its land (every call, unsorted) ---------
            for (int base = begin; base < end; base += BLOCK) {
                const int idx = base + tid;
                bool wet = false;
                if (idx < end) {
                    const int jj = row_of(idx, wx, wx_rcp);
                    const int i = idx - jj * wx - G.ring, j = jj - G.ring;
                    const size_t k = cell_index(G, i, j);
                    wet = cell_is_wet(P, mask, k);
                    if (!wet) lean_zero_cell<FUSE>(L, T_offset, G, opaque(K), k, i, j);
                }
                const unsigned long long m = __ballot(wet);
                int wave_base = 0;
                if (lane == 0 && m) wave_base = atomicAdd(&counters[2], __popcll(m));
                wave_base = __shfl(wave_base, 0);
                if (wet) {
                    const int p = wave_base + __popcll(m & ((1ull << lane) - 1ull));
                    if (p < CHUNK) list[p] = (unsigned)(idx - range_begin);
                }
            }
            __syncthreads();
            nwet = counters[2];
            if (nwet > CHUNK) {  // more wet cells than the list holds: retry on a piece that cannot overflow it
                end = begin + CHUNK;
                __syncthreads();
                if (tid < 4) counters[tid] = 0;
                if constexpr (CERT) {
                    if (tid < 2 * AO_BINS) hist[tid] = -1;
                }
                __syncthreads();
                continue;
            }
        }
        // ---- waves pull 64 wet cells at a time; the NEXT batch's inputs are requested before this batch iterates ------
        // (a batch's eleven loads take ≈ 2 µs to come back and its ≈ 2000 FP64 instructions ≈ 6 µs to issue: requested
        // one batch ahead, the loads cost 23 registers across the iteration and no wait)
        // (results the step's other launches do not read: streamed when this launch assembles the net fluxes itself — gstore_final)
        auto store_result = [&](double* base, unsigned byte_off, double v) {
            if constexpr (FUSE) gstore_final(base, byte_off, v);
            else gstore(base, byte_off, v);
        };
        auto claim = [&]() {
            int st = 0;
            if (lane == 0) st = atomicAdd(&counters[0], 64);
            return __shfl(st, 0);
        };
        // CERT: the queue of list positions whose cells go down the exact path, and whether this wave is working it off
        constexpr int QUEUE_CAP = 2 * AO_BINS;
        constexpr int QUEUE_CLOSED = 1 << 20;
        int* queue = hist;  // (hist and cursor are contiguous)
        static_assert(Geo::CURSOR_OFFSET == Geo::HIST_OFFSET + AO_BINS * 4, "the exact-path queue spans the histogram and the cursors");
        bool straggling = false;
        int limit = nwet;  // entries of the list this wave is claiming from (the chunk's list, or the queue)
        // list position of this lane's cell in the batch that starts at `st`
        auto position_of = [&](int st) -> int {
            const int qq = min(st + lane, limit - 1);
            if constexpr (CERT) return straggling ? queue[qq] : qq;
            return qq;
        };
        auto coords_of = [&](int st, int& ci, int& cj) -> size_t {
            const int idx = range_begin + (int)(list[position_of(st)] & LEAN_OFFSET_MASK);
            const int jj = row_of(idx, wx, wx_rcp);
            ci = idx - jj * wx - G.ring;
            cj = jj - G.ring;
            return cell_index(G, ci, cj);
        };
        auto cell_of = [&](int st) -> size_t {
            int ci, cj;
            return coords_of(st, ci, cj);
        };
        struct Raw {
            double ua, va, Ta, pa, qa, u0, u1, v0, v1, To, So;
        };
        auto request = [&](int st) {
            int ci, cj;
            const size_t k = coords_of(st, ci, cj);
            LeanArgsPtr Kb = opaque(K);  // this batch's view of the arguments: pointers are (re)loaded here, scalar loads
            const double* __restrict__ Ou = Kb->O.u;
            const double* __restrict__ Ov = Kb->O.v;
            const unsigned k8 = (unsigned)k * 8u;
            Raw r;
            r.u0 = gload(Ou, k8);
            r.u1 = gload(Ou, k8 + 8u);
            r.v0 = gload(Ov, k8);
            r.v1 = gload(Ov, k8 + (unsigned)G.sj * 8u);
            r.To = gload(Kb->O.T, k8);
            r.So = gload(Kb->O.S, k8);
            r.ua = gload(Kb->E.u, k8);
            r.va = gload(Kb->E.v, k8);
            r.Ta = gload(Kb->E.T, k8);
            r.pa = gload(Kb->E.p, k8);
            r.qa = gload(Kb->E.q, k8);
            return r;
        };
        int start = claim();
        Raw raw{};
        if (start < nwet) raw = request(start);
        for (;;) {
            if constexpr (CERT) {
                if (!straggling && start >= nwet) {
                    LEAN_STAMP(4);  // (stamped builds: the end phase of a certified wave is its share of the exact-path queue)
                    // no batch left for this wave.  The FIRST wave of the workgroup to get here closes the queue and works it
                    // off: one atomic add of QUEUE_CLOSED onto the fill count returns the number of entries reserved so far —
                    // the closer waits for those to be written — and every later reservation lands beyond the capacity, i.e.
                    // on the in-place path: the (at most three) batches still in flight run the exact iteration on their own
                    // uncertified cells, from registers.  The queue batch — reload, prologue, the slowest of a dozen lanes'
                    // ≈ 15 trips: ≈ 20 µs for a lone wave — then runs BESIDE the workgroup's last certified batches instead of
                    // behind them (the last wave taking the queue: 65.7 µs per launch, the first: 62.2; closing a round
                    // earlier: 70.0 — seven of sixteen batches then pay the in-place price; profiles/r05_experiments.md §1)
                    int filled = 0;
                    if (lane == 0) filled = atomicAdd(&counters[1], QUEUE_CLOSED);
                    filled = __shfl(filled, 0);
                    if (filled <= 0 || filled >= QUEUE_CLOSED) break;
                    limit = min(filled, QUEUE_CAP);
                    // (acquire: pairs with the producers' release store of the entry below)
                    for (int e = lane; e < limit; e += 64)
                        while (__hip_atomic_load(&queue[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 0) __builtin_amdgcn_s_sleep(1);
                    straggling = true;
                    start = 0;
                    raw = request(start);
                }
            }
            if (start >= limit) break;
            const int q = start + lane;
            const bool in_range = q < limit;
            double Mp_now = 0.0;
            const double raw_So = raw.So;
            if constexpr (CERT && CF_CERT_NET_SALT) {
                if (!straggling) Mp_now = gload(opaque(K)->E.Mp, (unsigned)cell_of(start) * 8u);
            }
            // ℑxᶜᵃᵃ u, ℑyᵃᶜᵃ v: cell-centre ocean velocity from the two bracketing faces
            const LeanCell c = lean_prologue(P, L.kappa, tab, raw.ua, raw.va, raw.Ta, raw.pa, raw.qa, 0.5 * (raw.u0 + raw.u1),
                                             0.5 * (raw.v0 + raw.v1), raw.To, raw.So);
            // the interface temperature does not depend on the iteration: written now, not carried across it
            if (in_range) store_result(opaque(K)->F.Ts, (unsigned)cell_of(start) * 8u, c.Ts - T_offset);
            CertNetSalt net_salt;
            if constexpr (CERT && CF_CERT_NET_SALT) {
                // J_S is assembled from the vapour flux (by this launch's epilogue or by net_cell_kernel): the certificate is
                // told what it may cancel against — in every certified launch, so that a cell's bits do not depend on the schedule
                // (two FP32 registers across the iteration; the precipitation is read again, in FP64, where J_S is assembled)
                net_salt.Mp = (float)Mp_now;
                net_salt.floor_v = CERT_JS_FLOOR * __builtin_amdgcn_rcpf((float)raw_So * (float)P.rho_f_inv);
            }
#if CF_LEAN_PREFETCH
            int next;
            if constexpr (CERT) next = straggling ? start + 64 : claim();
            else next = claim();
            if (next < limit) raw = request(next);
#endif
#ifdef CF_LEAN_STAMPS
            const unsigned long long t_it = __builtin_readcyclecounter();
#endif
            Scales s;
            bool finish = in_range;  // lanes whose results are written by this batch
            if constexpr (CERT) {
                bool exact = in_range;  // a batch from the queue: every lane takes the reference's iteration
                if (!straggling) {
                    // (the iteration's scalars are read from the argument block HERE, per batch: held for the kernel's
                    // lifetime beside the exact iteration's and the batch loop's they do not fit the scalar registers —
                    // 36 v_readlane per trip of the iteration were measured that way)
                    const LoopParams Lc = kread(&opaque(K)->L);
                    s = mo_iterate_certified<COARE>(Lc, c, tab, in_range, exact, net_salt);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(exact);
                    // (a workgroup's FIRST uncertified cells, met when its list is already handed out: the exact iteration at once,
                    // from the registers they are in — a queue of one or two cells would cost the workgroup a lone wave's reload,
                    // prologue and ≈ 15 dependent trips at its very end.  With the usual dozen per workgroup the queue has entries
                    // long before that, and compaction — sixteen batches' cells in one — is what makes this path pay:
                    // profiles/r05_experiments.md)
                    const bool in_place = ((volatile int*)counters)[1] == 0 && ((volatile int*)counters)[0] >= nwet;
                    if (m && !in_place) {
                        int slot = 0;
                        if (lane == 0) slot = atomicAdd(&counters[1], __popcll(m));
                        slot = __shfl(slot, 0) + __popcll(m & ((1ull << lane) - 1ull));
                        if (exact && slot < QUEUE_CAP) {
                            // (release at workgroup scope: the closer's acquire load sees a written entry, whatever the compiler
                            // would have liked to do with a plain LDS store between relaxed atomics — ADVICE r5)
                            __hip_atomic_store(&queue[slot], q, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            exact = false;
                            finish = false;
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(exact) != 0ull) {
                    const LoopParams Le = kread(&opaque(K)->L);
                    const Scales e = mo_iterate_lean<COARE, false>(Le, c, tab, exact);
                    if (exact) {
                        s = e;
                        s.it |= CERT_EXACT_FLAG;
                    }
                }
            } else {
                if constexpr (LINE) s = mo_iterate_lean_line<COARE>(L, c, tab, in_range);
                else s = mo_iterate_lean<COARE>(L, c, tab, in_range);
            }
#ifdef CF_LEAN_STAMPS
            stamp_iter += __builtin_readcyclecounter() - t_it;
            ++stamp_batches;
            {
                int m = in_range ? (s.it & 0xff) : 0;
                for (int d = 32; d; d >>= 1) m = max(m, __shfl_xor(m, d));
                stamp_trips += (unsigned long long)m;
            }
#endif
            if (finish) {
                LeanArgsPtr Ke = opaque(K);
                // (cell coordinates recomputed from the list entry: cheaper than registers held across the iteration)
                const size_t k = cell_of(start);
                const CellFluxes R = lean_epilogue(c, T_offset, s);
                const FluxOut F = kread(&Ke->F);
                const unsigned k8 = (unsigned)k * 8u;
                store_result(F.Qc, k8, R.Qc);
                store_result(F.Qv, k8, R.Qv);
                store_result(F.Fv, k8, R.Fv);
                gstore(F.tx, k8, R.rho_tau_x);
                gstore(F.ty, k8, R.rho_tau_y);
                if (F.ustar) gstore_final(F.ustar, k8, R.ustar);
                if (F.tstar) gstore_final(F.tstar, k8, R.tstar);
                if (F.qstar) gstore_final(F.qstar, k8, R.qstar);
                if (F.iters) gstore_i32(F.iters, (unsigned)k * 4u, R.iterations);
                if constexpr (FUSE) {
                    // compute_net_ocean_fluxes!, the part that needs no neighbour: interior cells only
                    const int idx = range_begin + (int)(list[position_of(start)] & LEAN_OFFSET_MASK);
                    const int jj = row_of(idx, wx, wx_rcp);
                    const int ci = idx - jj * wx - G.ring, cj = jj - G.ring;
                    if (ci >= 0 && ci < G.nx && cj >= 0 && cj < G.ny) {
                        const IceIn I = kread(&Ke->I);
                        const NetOut N = kread(&Ke->N);
                        const double Ts_ocean = (gload(Ke->O.T, k8) + P.T_offset) - T_offset;  // what F.Ts holds (written before the iteration)
                        const NetCell C = net_cell_local(P, P.albedo, I.conc ? gload(I.conc, k8) : 0.0, gload(Ke->O.S, k8), Ts_ocean + P.T_offset,
                                                         gload(Ke->E.Mp, k8), gload(Ke->E.Qs, k8), gload(Ke->E.Ql, k8), R.Qc, R.Qv, R.Fv,
                                                         I.Qio ? gload(I.Qio, k8) : 0.0, I.Jsio ? gload(I.Jsio, k8) : 0.0,
                                                         I.land ? gload(I.land, k8) : 0.0);
                        gstore_final(N.T, k8, C.JT);  // (store_net_cell's fields, by offset)
                        gstore_final(N.S, k8, C.JS);
                        if (N.sw) gstore_final(N.sw, k8, C.sw);
                        if (N.lw_up) gstore_final(N.lw_up, k8, C.lw_up);
                        if (N.lw_down) gstore_final(N.lw_down, k8, C.lw_down);
                        if (N.sw_down) gstore_final(N.sw_down, k8, C.sw_down);
                    }
                }
                if (sorting && have_list) {
                    const int w = min(s.work, 255);
                    list[q] = (list[q] & LEAN_OFFSET_MASK) | ((unsigned)w << LEAN_OFFSET_BITS);
                    atomicAdd(&hist[lean_bin(q, w)], 1);
                }
            }
#if !CF_LEAN_PREFETCH
            int next;
            if constexpr (CERT) next = straggling ? start + 64 : claim();
            else next = claim();
            if (next < limit) raw = request(next);
#endif
            start = next;
        }
        if (have_list || end >= range_end) break;
        begin = end;  // classification path: the rest of the range
        end = range_end;
        __syncthreads();  // list and counters are reused
        if (tid < 4) counters[tid] = 0;
        if constexpr (CERT) {
            if (tid < 2 * AO_BINS) hist[tid] = -1;
        }
        __syncthreads();
    }
    if constexpr (!CERT) LEAN_STAMP(4);
    LEAN_STAMP_SET(6, stamp_iter | (stamp_trips << 40));
    LEAN_STAMP_SET(7, stamp_batches | ((unsigned long long)(have_list ? 1 : 0) << 32) | ((unsigned long long)(sorting ? 1 : 0) << 33));
    // ---- end phase: the list in next call's order (counting sort by this call's trip counts, longest first) --------
    // Behind a barrier, by the whole workgroup: a wave that runs out of batches has nothing else to do — every
    // workgroup of the kernel is resident from the start, no slot it could free is waited for — and the workgroup
    // is as old as its last wave either way.
    if (sorting && have_list) {
        __syncthreads();
        if (tid < AO_BINS) {
            const int v = hist[tid];
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d);
                if (lane >= d) incl += up;
            }
            cursor[tid] = incl - v;
        }
        __syncthreads();
        uint32_t* out = opaque(K)->sorted + (size_t)chunk * CHUNK;
        constexpr int PER_THREAD = CHUNK / BLOCK;
        unsigned word[PER_THREAD];
        int at[PER_THREAD];
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n) word[n] = list[min(tid + n * BLOCK, CHUNK - 1)];
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n)
            at[n] = tid + n * BLOCK < nwet ? atomicAdd(&cursor[lean_bin(tid + n * BLOCK, (int)(word[n] >> LEAN_OFFSET_BITS))], 1) : -1;
#pragma unroll
        for (int n = 0; n < PER_THREAD; ++n)
            if (at[n] >= 0) out[at[n]] = word[n] & LEAN_OFFSET_MASK;
    }
    LEAN_STAMP(5);
}

}  // namespace coflux
