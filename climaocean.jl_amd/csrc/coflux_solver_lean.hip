// coflux_solver_lean.hip — compute_atmosphere_ocean_fluxes! for the production ocean configurations
// (SOLVER_OCEAN_LEAN: SimilarityTheoryFluxes with Charnock-type momentum roughness and Reynolds-scaled scalars,
// omip_simulation.jl:40-49 and the defaults of README.md:75), round-3 kernel.
//
// What it keeps from coflux_solver.hip: one 256-thread workgroup per chunk of the cost-balanced chunk table, the ψ /
// log tables and the parameter block staged in LDS by LDS-DMA, land compacted away, batches of 64 wet cells, waves
// leaving the iteration together on a wave64 ballot, a mask rewritten in place detected by a fingerprint of the chunk's
// wet set (a stale list costs time, never correctness).
//
// What is new (profiles/r03_experiments.md):
//   * the iteration, its prologue and its epilogue are coflux_lean.hpp's;
//   * the start phase no longer sorts.  Round 2 loaded every chunk's static list and hint bytes, counting-sorted them
//     in LDS (two atomic passes, a scan, three barriers) and hashed every listed cell to validate the list: ≈ 9 µs
//     in which no CU of the device computes (all workgroups start together).  Now the list lives in global memory in
//     the order the batches will take it: it goes straight into LDS by LDS-DMA beside the tables, the chunk's
//     fingerprint is a number computed when the list was built, and one barrier separates the requests from the first
//     batch (≈ 4 µs);
//   * batches are in INDEX order by default (CF_OPT_TRIP_HINTS = 2): sorted by last call's trip counts a batch's cells
//     are scattered over the chunk and its memory phases cost more than the trips the sort saves once the hints are a
//     step old; index order also makes the accesses coalesced (HBM traffic 239 → ≈ 100 MB per launch) and the fused
//     net-flux epilogue cheap.  With CF_OPT_TRIP_HINTS = 1 the order for the NEXT call is produced at the END of the
//     workgroup's life: every batch leaves its lanes' trip counts in the list words' top byte and bumps a 64-bin
//     histogram; behind a barrier the workgroup scans the bins and scatters the list back to global memory;
//   * a batch's loads and stores are uniform base + 32-bit offset (gload / gstore): `global_* v_off, s[base]`.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>


namespace coflux {

// per-wave time stamps (scratch/phases_lean.py; -DCF_LEAN_STAMPS builds only): 0 entry, 1 everything requested, 2 my
// requests have landed, 3 behind the barrier (batches begin), 4 batches done, 5 exit, 6 Σ cycles inside the iteration,
// 7 batches taken
#ifdef CF_LEAN_STAMPS
__device__ unsigned long long g_lean_stamp[4096 * 8];
#define LEAN_STAMP(q) do { if (lane == 0) g_lean_stamp[((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 8 + (q)] = __builtin_readcyclecounter(); } while (0)
#define LEAN_STAMP_SET(q, v) do { if (lane == 0) g_lean_stamp[((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 8 + (q)] = (v); } while (0)
extern "C" int cf_debug_phase_read(unsigned long long* out, int n) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lean_stamp), sizeof(unsigned long long) * n);
    return 0;
}
#else
#define LEAN_STAMP(q) do { } while (0)
#define LEAN_STAMP_SET(q, v) do { } while (0)
#endif
}  // namespace coflux

#include "coflux_lean_kernel.hpp"

namespace coflux {

template <bool COARE, bool FUSE, bool TAIL = false, bool CERT = false>
__global__ __launch_bounds__(AO_BLOCK, CF_LEAN_WAVES) void ao_lean_kernel(LeanArgs unused_by_name) {
    ao_lean_body<COARE, FUSE, TAIL, CERT>((LeanArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), (int)blockIdx.x);
}

// the stepping loop's exact-path launch with the step's peer-direct halo rows as rider workgroups (ao_lean_body, HALO)
template <bool COARE>
__global__ __launch_bounds__(AO_BLOCK, CF_LEAN_WAVES) void ao_lean_halo_kernel(LeanArgs unused_by_name) {
    ao_lean_body<COARE, true, true, false, false, true>((LeanArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Host side: the sorted lists start as the static wet lists of the chunk table (index order); the fingerprint and
// the wet count of every chunk are computed here, once per mask.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lean_list_build_kernel(const uint32_t* __restrict__ wet_pos, const int* __restrict__ begins, int stride,
                                                              uint32_t* __restrict__ sorted, int* __restrict__ info) {  // stride: AO_CHUNK
    __shared__ unsigned sx[4], sy[4];
    __shared__ int sn[4];
    const int chunk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int begin = begins[chunk];
    unsigned hx = 0, hy = 0;
    int n = 0;
    for (int e = threadIdx.x; e < stride; e += 256) {
        const uint32_t idx = wet_pos[(size_t)chunk * stride + e];
        const bool listed = idx != 0xffffffffu;
        sorted[(size_t)chunk * stride + e] = listed ? idx - (uint32_t)begin : LEAN_OFFSET_MASK;
        if (listed) {
            hx ^= lean_mix(idx);
            hy += lean_sum(idx);
            ++n;
        }
    }
    for (int d = 32; d; d >>= 1) {
        hx ^= (unsigned)__shfl_xor((int)hx, d);
        hy += (unsigned)__shfl_xor((int)hy, d);
        n += __shfl_xor(n, d);
    }
    if (lane == 0) {
        sx[wave] = hx;
        sy[wave] = hy;
        sn[wave] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        info[chunk * 4] = sn[0] + sn[1] + sn[2] + sn[3];
        info[chunk * 4 + 1] = (int)(sx[0] ^ sx[1] ^ sx[2] ^ sx[3]);
        info[chunk * 4 + 2] = (int)(sy[0] + sy[1] + sy[2] + sy[3]);
        info[chunk * 4 + 3] = 0;
    }
}

hipError_t build_lean_lists(hipStream_t st, int nchunks, const uint32_t* d_wet_pos, const int* d_begins, uint32_t* d_sorted, int* d_info) {
    hipLaunchKernelGGL(lean_list_build_kernel, dim3(nchunks), dim3(256), 0, st, d_wet_pos, d_begins, AO_CHUNK, d_sorted, d_info);
    return hipGetLastError();
}

bool lean_certified_applies(const LaunchCfg& L, const LoopParams& C) {
    // (maxiter ≥ 40: a cell is certified only where the map's spectral radius is below 0.6 — from the 1e-4 first guess the
    // reference then needs fewer than 40 trips to bring its drift under any tolerance ≥ 1e-9, i.e. it stops on the drift,
    // not on the cap, which the certificate presumes)
    return L.certified && C.specialization == SOLVER_OCEAN_LEAN && !C.fixed && L.lean_hints == 0 &&
           C.cert_max_evals > 2 && C.tol >= 1e-9 && C.maxiter >= 40;
}

bool lean_halo_rides(const LaunchCfg& L, const LoopParams& C) {
    return C.specialization == SOLVER_OCEAN_LEAN && L.solver == CF_SOLVER_TABLES && !lean_certified_applies(L, C);
}

bool lean_line_applies(const LaunchCfg& L, const LoopParams& C, bool coare) {
    // (two workgroups of 256 VGPRs per lane fill a CU: a plan with more would run in rounds; β_gust ≠ 0 is the layout's
    // precondition — mo_iterate_lean_line —; the certified path has its own kernels)
    if (L.latency_layout == 0 || C.specialization != SOLVER_OCEAN_LEAN || C.beta_gust == 0.0) return false;
    if (lean_certified_applies(L, C)) return false;
    // automatic: where it is measured to pay — the COARE profile, whose trip is ONE basic block (1440×70: 24.3 → 22.0 µs per
    // step); with the plain logarithmic profile the slowest waves are the ones that take the general ψ at the roughness
    // lengths behind a per-lane branch, and a lone wave issues that block no faster re-ordered (27.3 µs either way,
    // profiles/r05_experiments.md §9)
    return L.latency_layout == 2 || (coare && L.cu_count > 0 && L.n_chunks <= 2 * L.cu_count);
}

// the argument block of one ocean solve (everything but the tail workgroups' and the fused interpolation's descriptors)
static hipError_t fill_lean_args(const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G, const cf_ocean_surface* o,
                                 const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                                 const cf_net_ocean_fluxes* net, const double* land, LeanArgs& A) {
    if (!L.d_chunk_begins || L.n_chunks <= 0 || !L.d_lean_info) return hipErrorInvalidValue;
    A.L = C;
    A.G = G;
    A.O = make_ocean(o);
    A.E = make_exchange(e);
    A.F = make_fluxes(f);
    A.g_tab = L.d_tables;
    A.g_params = L.d_params;
    A.sorted = L.d_wet_pos ? L.d_lean_sorted : nullptr;  // (no static lists: every workgroup classifies its range per call)
    A.info = L.d_lean_info;
    A.chunk_begins = L.d_chunk_begins;
    A.z_surface = P.z_surface;
    A.mask_kind = P.mask_kind;
    A.T_offset = P.T_offset;
    A.wx_reciprocal = row_reciprocal(G.nx + 2 * G.ring);
    A.sort_enabled = L.lean_hints;
    if (L.lean_hints > 1) {
        static const int windows = [] {  // (experiments: COFLUX_EXPERIMENTS=1 COFLUX_SORT_WINDOWS=1|2|4|8, read once)
            const char* e = experiment_knob("COFLUX_SORT_WINDOWS");
            return e ? std::max(1, std::min(8, std::atoi(e))) : 0;
        }();
        if (windows > 0) A.sort_enabled = windows;
    }
    if (net) {  // the fused form: the epilogue also writes the cell-local net ocean fluxes (constant ocean albedo only)
        if (ice) A.I = IceIn{ice->concentration, ice->interface_heat, ice->salt_flux, ice->x_stress, ice->y_stress, nullptr};
        A.I.land = land;
        A.N = NetOut{net->u, net->v, net->T, net->S, net->shortwave_surface_flux, net->upwelling_longwave, net->downwelling_longwave,
                     net->downwelling_shortwave};
    }
    return hipSuccess;
}

hipError_t make_ocean_rider(const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                            const cf_net_ocean_fluxes* net, const double* land, OceanRider* out) {
    static_assert(sizeof(LeanArgs) <= sizeof(out->args), "OceanRider::args holds a LeanArgs");
    if (!out || !net) return hipErrorInvalidValue;
    LeanArgs A{};
    if (hipError_t err = fill_lean_args(L, P, C, G, o, e, f, ice, net, land, A)) return err;
    memcpy(out->args, &A, sizeof(A));
    out->n_chunks = L.n_chunks;
    out->coare = P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC;
    out->valid = true;
    return hipSuccess;
}

hipError_t launch_ao_fluxes_lean(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G,
                                 const cf_ocean_surface* o, const cf_exchange_fields* e, const cf_interface_fluxes* f,
                                 const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net, const double* land,
                                 const cf_atmos_source* next_src, const cf_interp_weights* w, const cf_exchange_fields* next_out,
                                 int tail_rows, int tail_blocks, int tail_pos, const HaloRider* halo) {
    LeanArgs A{};
    if (hipError_t err = fill_lean_args(L, P, C, G, o, e, f, ice, net, land, A)) return err;
    const bool coare = P.similarity_form == CF_SIMILARITY_COARE_LOGARITHMIC;
    const bool cert = lean_certified_applies(L, C);
    const bool tail = next_out != nullptr;
    int blocks = L.n_chunks;
    if (tail) {
        // tail workgroups: `next_src`, `w` describe the NEXT step's interpolation into `next_out` (with the fused net fluxes)
        if (!net || !next_src || !w || L.interp_cap <= 0 || tail_blocks <= 0) return hipErrorInvalidValue;
        const size_t tile_lds = (size_t)IT_WAVES * CF_JRA55_NVARS * L.interp_cap * sizeof(double);
        if (tile_lds > (size_t)LeanGeom<AO_BLOCK>::LDS_BYTES) return hipErrorInvalidValue;
        A.S = make_source(next_src);
        A.Wt = make_weights(w);
        A.E_next = make_exchange(next_out);
        A.n_chunks = L.n_chunks;
        A.tail_blocks = tail_blocks;
        A.tail_rows = tail_rows;
        A.tail_cap = L.interp_cap;
        A.tail_pos = tail_pos < 0 || tail_pos > L.n_chunks ? L.n_chunks : tail_pos;
        blocks += tail_blocks;
    }
    const bool riders = halo != nullptr && halo->blocks > 0;
    if (riders) {
        if (!tail || !lean_halo_rides(L, C) || halo->F.n < 1 || halo->blocks != 2 * halo->F.n) return hipErrorInvalidValue;
        A.H = *halo;
        blocks += halo->blocks;
    }
    if (lean_line_applies(L, C, coare)) return launch_ao_lean_line(st, coare, net != nullptr, tail, riders, blocks, A);
    if (riders) {
        if (coare) hipLaunchKernelGGL((ao_lean_halo_kernel<true>), dim3(blocks), dim3(AO_BLOCK), LeanGeom<AO_BLOCK>::LDS_BYTES, st, A);
        else hipLaunchKernelGGL((ao_lean_halo_kernel<false>), dim3(blocks), dim3(AO_BLOCK), LeanGeom<AO_BLOCK>::LDS_BYTES, st, A);
        return hipGetLastError();
    }
#define CF_LEAN_LAUNCH(COARE_, FUSE_, TAIL_, CERT_) \
    hipLaunchKernelGGL((ao_lean_kernel<COARE_, FUSE_, TAIL_, CERT_>), dim3(blocks), dim3(AO_BLOCK), LeanGeom<AO_BLOCK>::LDS_BYTES, st, A)
#define CF_LEAN_PICK(FUSE_, TAIL_)                                                       \
    do {                                                                                 \
        if (cert) {                                                                      \
            if (coare) CF_LEAN_LAUNCH(true, FUSE_, TAIL_, true); else CF_LEAN_LAUNCH(false, FUSE_, TAIL_, true);   \
        } else {                                                                         \
            if (coare) CF_LEAN_LAUNCH(true, FUSE_, TAIL_, false); else CF_LEAN_LAUNCH(false, FUSE_, TAIL_, false); \
        }                                                                                \
    } while (0)
    if (tail) CF_LEAN_PICK(true, true);
    else if (net) CF_LEAN_PICK(true, false);
    else CF_LEAN_PICK(false, false);
#undef CF_LEAN_PICK
#undef CF_LEAN_LAUNCH
    return hipGetLastError();
}

}  // namespace coflux
