// coflux_solver_shared.hpp — what the solver kernels (coflux_solver.hip: every specialization on round 2's outer
// structure; coflux_solver_lean.hip: the round-3 ocean kernel) share: workgroup geometry, the kernarg bundle and the
// small device helpers around it.
#pragma once
#include <hip/hip_runtime.h>

#include "coflux_fast.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"
#include "coflux_interp_tiles.hpp"

namespace coflux {

constexpr int TABLE_BYTES = TABLE_DOUBLES * 8;
constexpr int AO_BLOCK = 256;
#ifndef CF_AO_CHUNK
#define CF_AO_CHUNK 1280
#endif
constexpr int AO_CHUNK = CF_AO_CHUNK;  // capacity of a narrow workgroup's wet-cell list = the most wet cells a chunk can hold
constexpr int AO_BINS = 64;    // trip-count bins of the per-chunk counting sort (one wave scans them)
constexpr int AO_WET_COST = 64;

// zero_interface_state of a land cell: all fluxes 0, T = 0 K (and, in the fused path, zero net fluxes inside the interior)
template <bool FUSE_NET>
__device__ __forceinline__ void zero_cell(const LoopParams& L, double T_offset, const GridDesc& G, const FluxOut& F,
                                          const NetOut& N, size_t k, int i, int j) {
    CellFluxes Z{};
    Z.Ts_ocean = -T_offset;
    Z.iterations = L.fixed ? L.maxiter : 0;
    store_fluxes(F, k, Z);
    if constexpr (FUSE_NET) {
        if (i >= 0 && i < G.nx && j >= 0 && j < G.ny) store_net_cell(N, k, NetCell{});
    }
}

// ---- production solver: LDS tables, one workgroup per chunk of the table above -----------------
// LDS: tables | list (cell offset + list entry per sorted position) | counters, histogram, bin cursors | DevParams
// A list word: the cell's offset from the start of the chunk's range (20 bits: a range costs at most AO_CHUNK wet
// cells' worth of AO_WET_COST = 196 608 cells if it were all land) and its entry in the static list (12 bits).
constexpr int AO_LIST_OFFSET_BITS = 20;
static_assert(AO_CHUNK <= (1 << (32 - AO_LIST_OFFSET_BITS)), "list entry index must fit the upper bits");
static_assert((long)AO_CHUNK * AO_WET_COST < (1L << AO_LIST_OFFSET_BITS), "a chunk's range must fit the lower bits");

struct WetLists {
    const uint32_t* pos;    // wet cells of chunk c in index order at [c·AO_CHUNK, …), 0xffffffff-padded; nullptr: classify per call
    uint8_t* trip;          // iteration count of the previous call per list entry (scheduling hint), or nullptr
};

// Everything the kernel is handed, as ONE by-value argument: the kernarg segment then IS this struct, and the
// kernel reads its ≈ 40 pointers from there with scalar loads where it uses them (through a pointer the compiler
// cannot see through, so that it reloads them per batch instead of hoisting 80 SGPRs' worth out of the loops and
// spilling them into VGPR lanes — 185 SGPR spills and ≈ 300 v_readlane per batch were measured that way).  Only
// the iteration's scalars (LoopParams) are copied into registers for the kernel's lifetime.
struct SolverArgs {
    LoopParams L;
    GridDesc G;
    OceanIn O;
    Exchange E;
    FluxOut F;
    const double* g_tab;
    const DevParams* g_params;
    WetLists W;
    const int* chunk_begins;
    IceIn I;
    NetOut N;
    IceStateIn S;   // SOLVER_SEAICE only
    IceParams Ice;
    double z_surface;    // with mask_kind: how the start phase reads wetness before the parameter block is in LDS
    long long mask_kind;
    double T_offset;     // (zero_interface_state writes −T_offset; same reason)
    unsigned long long wx_reciprocal;  // floor(2³² / (nx + 2·ring)), see row_of
    // TAIL launches only (CF_OPT_MERGED_PREFETCH = 2): the next step's interpolation in the workgroups behind the chunks'
    SourceDesc Si;
    WeightDesc Wi;
    Exchange E_next;
    long long n_chunks, tail_blocks, tail_rows, tail_cap;
    // … and / or the face stresses of the CURRENT step (the sea-ice interface launch: they need the ocean solver's ρτ, which
    // the previous launch wrote): workgroups behind the interpolation's, 256 cells each
    long long stress_blocks;
    const DevParams* stress_params;   // the OCEAN formulation's parameter block (ρₒ, mask kind)
    const void* stress_mask;
    const double* rtx;
    const double* rty;
    IceIn stress_ice;
    double* tau_x;
    double* tau_y;
    NetIceOut NI;   // the sea-ice interface launch: compute_net_sea_ice_fluxes! in its epilogue
};
typedef const SolverArgs __attribute__((address_space(4)))* SolverArgsPtr;

// by-value copy of a member of the kernarg struct: scalar loads of its 8-byte words
template <class T>
__device__ __forceinline__ T kread(const __attribute__((address_space(4))) T* p) {
    static_assert(sizeof(T) % 8 == 0, "argument bundles are made of 8-byte words");
    T v;
    const __attribute__((address_space(4))) unsigned long long* src = (const __attribute__((address_space(4))) unsigned long long*)p;
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(&v);
#pragma unroll
    for (size_t n = 0; n < sizeof(T) / 8; ++n) dst[n] = src[n];
    return v;
}

// Global-memory accesses as (uniform base pointer, 32-bit byte offset per lane): the compiler emits
// `global_load/store … v_off, s[base:base+1]` — no 64-bit address arithmetic per field in vector registers, and no
// `flat_` instruction (a pointer read from the argument block is a generic pointer to the compiler, and flat accesses
// count against the LDS counter as well).  A surface's fields are far below 4 GB.
__device__ __forceinline__ double gload(const double* base, unsigned byte_off) {
    return *(const __attribute__((address_space(1))) double*)((const __attribute__((address_space(1))) char*)base + byte_off);
}
__device__ __forceinline__ void gstore(double* base, unsigned byte_off, double v) {
    *(__attribute__((address_space(1))) double*)((__attribute__((address_space(1))) char*)base + byte_off) = v;
}
// A store of a value that no kernel of this step reads again (the turbulent heat and vapour fluxes, the interface temperature, the
// cell-local net fluxes — not ρτ, which the face-stress launch reads next): a streaming store (`nt`), so that the line does not
// take the place of one the step still needs in the L2.  Same-box A/B, six pairs: the 1/4-degree step 82.8 → 81.7 µs, `:corrected`
// 72.7 → 71.4, the 1/6-degree tripolar surface 219.2 → 214.3, the 1/8 slab 26.05 → 25.82 (profiles/r06_experiments.md §8); ρτ
// streamed as well costs the stress kernel what the solver gains, the exchange fields streamed by the interpolation cost the solver
// more, streaming LOADS of the inputs + 4 … 7 %.  Only where the launch also assembles the net fluxes (FUSE): without them the
// stand-alone net-flux kernel reads these fields next.
__device__ __forceinline__ void gstore_final(double* base, unsigned byte_off, double v) {
    __builtin_nontemporal_store(v, (__attribute__((address_space(1))) double*)((__attribute__((address_space(1))) char*)base + byte_off));
}
__device__ __forceinline__ void gstore_i32(int* base, unsigned byte_off, int v) {
    *(__attribute__((address_space(1))) int*)((__attribute__((address_space(1))) char*)base + byte_off) = v;
}

__device__ __forceinline__ SolverArgsPtr opaque(SolverArgsPtr p) {
    asm volatile("" : "+s"(p));
    return p;
}

// row of a window-linear cell index: idx / wx through the host's floor(2³² / wx) — five instructions where the
// compiler's 32-bit division is ≈ 20, and the kernel does it for every mask word of the start phase and twice per batch.
// The product underestimates the quotient by at most one for idx < 2²⁶ (67 M cells).
__device__ __forceinline__ int row_of(int idx, int wx, unsigned wx_reciprocal) {
    int q = (int)__umulhi((unsigned)idx, wx_reciprocal);
    return idx - q * wx >= wx ? q + 1 : q;
}

// The scheduling hint of a cell: its work in this call, or the previous hint minus one if that is larger.  A cell
// that ran long once keeps a long hint for a while (it decays by one per call) and is scheduled among the first batches
// of its workgroup; hinted short and running long it would sit in a LATE batch and hold the workgroup — and with it the
// kernel — for its whole iteration alone (the sea-ice solve with its hints a step old: 386 → 300 µs; a cell that flips
// between 35 and 100 iterations from one step to the next is all it takes).  An over-hinted lane merely idles.
// Sea ice only: the ocean's counts move by ±1, there the read-modify-write costs more than the bias saves (+0.5 %).
__device__ __forceinline__ void store_hint(uint8_t* hint, int work) {
    const int old = *hint;
    *hint = (uint8_t)min(max(work, old - 1), 255);
}

// Sort bin of a trip count: one bin per count below 56 (the ocean needs 8–20 iterations, most sea-ice cells 10–40),
// eight-count bins above (the sea-ice iteration's orbit cells stop anywhere up to maxiter = 100).
__device__ __forceinline__ int trip_bin(int t) { return t < 56 ? t : min(56 + ((t - 56) >> 3), AO_BINS - 1); }

// (a one-cell-wide window: 2³² does not fit the 32-bit operand; 2³² − 1 gives q = idx − 1 and the correction step adds the one)
inline unsigned long long row_reciprocal(int wx) { return wx <= 1 ? 0xffffffffull : 0x100000000ull / (unsigned long long)wx; }

// two independent 32-bit mixes of a cell's linear index (murmur3's finaliser): XOR-accumulated over a set of cells
// they make a 64-bit fingerprint of the set
__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned cell_hash_lo(unsigned idx) { return mix32(idx + 0x9e3779b9u); }
__device__ __forceinline__ unsigned cell_hash_hi(unsigned idx) { return mix32(idx * 0x01000193u ^ 0x7f4a7c15u); }

}  // namespace coflux
