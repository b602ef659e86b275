// Host-visible launchers of the gfx950 kernels (coflux_interp.hip, coflux_solver.hip,
// coflux_solver_libm.hip, coflux_net.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "coflux_device.hpp"

namespace coflux {

// Experiment knobs (COFLUX_LAYERS, COFLUX_SORT_WINDOWS, COFLUX_INTERP_BLOCKS: scheduling only, never results) are honoured
// only in a process started with COFLUX_EXPERIMENTS=1, and each is read once per process — a stray variable in a
// production environment cannot silently change what a run measures.
inline const char* experiment_knob(const char* name) {
    static const bool enabled = [] {
        const char* e = std::getenv("COFLUX_EXPERIMENTS");
        return e != nullptr && e[0] == '1';
    }();
    return enabled ? std::getenv(name) : nullptr;
}


struct LoopParams;
struct IceParams;
struct HaloRider;

struct LaunchCfg {
    int solver;              // CF_SOLVER_*
    int interp_cap;          // float2 entries per variable of a wave's LDS-staged JRA55 tile
    int ao_chunk;            // wet cells per solver workgroup: 256 / 512 / 768 (0 = automatic)
    int cu_count;            // compute units of the device (sizes the automatic chunk)
    const double* d_tables;  // device copy of the solver tables (coflux_tables.cpp)
    const DevParams* d_params;  // device copy of DevParams (the solver stages it in LDS)
    uint8_t* d_trip;            // per-wet-cell trip count of the previous call (scheduling hint) or NULL
    const int* d_chunk_begins;  // cost-balanced chunk table of the solver (coflux_solver.hip), n_chunks + 1 entries
    int n_chunks;
    int chunk_south, chunk_north;  // chunks [0, chunk_south) read the south halo rows, [chunk_north, n_chunks) the north ones (HaloRider)
    const uint32_t* d_wet_pos;  // static wet lists of the chunks, fixed stride (coflux_solver.hip), or NULL
    uint32_t* d_lean_sorted;    // the lean ocean kernel's lists: every chunk's wet cells ordered by last call's trip counts (coflux_solver_lean.hip)
    const int* d_lean_info;     // per chunk: wet cells listed, fingerprint of the wet set (x, y), 0
    int lean_hints;             // 1: the lean ocean kernel re-orders its lists by trip count at the end of every call
    int certified;              // CF_OPT_SOLVER_PATH: 1 = the certified reduced-iteration solve wherever it applies (lean_certified_applies)
    int latency_layout;         // CF_OPT_LATENCY_LAYOUT: 0 never, 1 automatic (COARE profile on chunk plans of at most two workgroups per CU), 2 always
};

// the exact path's kernels for one or two waves per SIMD (coflux_solver_slab.hip) carry this launch
bool lean_line_applies(const LaunchCfg& L, const LoopParams& C, bool coare);

// CF_SOLVER_PATH_CERTIFIED runs in the lean ocean kernel under the convergence stop rule with index-ordered lists; everywhere
// else the exact path runs.
bool lean_certified_applies(const LaunchCfg& L, const LoopParams& C);

hipError_t launch_interpolate(hipStream_t st, const LaunchCfg& L, const GridDesc& G, const cf_atmos_source* s,
                              const cf_interp_weights* w, const cf_exchange_fields* e);
hipError_t launch_ao_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C,
                            const GridDesc& G, const cf_ocean_surface* o, const cf_exchange_fields* e,
                            const cf_interface_fluxes* f, const cf_sea_ice_fields* ice = nullptr,
                            const cf_net_ocean_fluxes* net = nullptr, const double* land = nullptr);
hipError_t launch_net_stress(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                             const cf_interface_fluxes* f, const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* n);
hipError_t build_wet_lists(hipStream_t st, const DevParams* d_params, const GridDesc& G, const void* mask, int nchunks,
                           const int* d_begins, uint32_t* d_wet_pos, uint8_t* d_trip, int* d_scratch, int* overflow_out);
hipError_t build_lean_lists(hipStream_t st, int nchunks, const uint32_t* d_wet_pos, const int* d_begins, uint32_t* d_sorted, int* d_info);
hipError_t launch_ly_fluxes_with_tail(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G,
                                      const cf_ocean_surface* o, const cf_exchange_fields* e, const cf_interface_fluxes* f,
                                      const cf_sea_ice_fields* ice, const cf_net_ocean_fluxes* net, const double* land,
                                      const cf_atmos_source* next_src, const cf_interp_weights* w, const cf_exchange_fields* next_out,
                                      int tail_rows, int tail_blocks);
hipError_t launch_ao_fluxes_lean(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G,
                                 const cf_ocean_surface* o, const cf_exchange_fields* e, const cf_interface_fluxes* f,
                                 const cf_sea_ice_fields* ice = nullptr, const cf_net_ocean_fluxes* net = nullptr, const double* land = nullptr,
                                 const cf_atmos_source* next_src = nullptr, const cf_interp_weights* w = nullptr,
                                 const cf_exchange_fields* next_out = nullptr, int tail_rows = 0, int tail_blocks = 0, int tail_pos = -1,
                                 const HaloRider* halo = nullptr);
// the step's peer-direct halo rows can ride in the ocean solver's launch (exact path of the lean kernel, with tail workgroups)
bool lean_halo_rides(const LaunchCfg& L, const LoopParams& C);
size_t wet_list_capacity(int ncells);
int wet_list_stride();
hipError_t launch_ao_fluxes_libm(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                                 const cf_exchange_fields* e, const cf_interface_fluxes* f);
hipError_t launch_net_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const cf_ocean_surface* o,
                             const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                             const cf_interp_weights* w, const cf_net_ocean_fluxes* n, const double* land = nullptr);
hipError_t launch_interpolate_land(hipStream_t st, const GridDesc& G, const cf_land_source* s, const cf_interp_weights* w, double* out);
hipError_t launch_salinity_restoring(hipStream_t st, const DevParams& P, const GridDesc& G, const void* mask, double vp,
                                     const double* target, const double* S, double* out);
void interpolate_grid(const LaunchCfg& L, const GridDesc& G, int* rows_out, int* blocks_out);
hipError_t launch_interpolate_and_stress(hipStream_t st, const LaunchCfg& L, const DevParams& P, const GridDesc& G,
                                         const cf_atmos_source* s, const cf_interp_weights* w, const cf_exchange_fields* e,
                                         const cf_ocean_surface* o, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                                         const cf_net_ocean_fluxes* n);
hipError_t launch_interpolate_background(hipStream_t st, const GridDesc& G, const cf_atmos_source* s,
                                         const cf_interp_weights* w, const cf_exchange_fields* e);
// The ocean solve of this step as workgroups of ANOTHER launch (the sea-ice interface solve's: ice_ocean_kernel): the round-3
// ocean kernel's argument block, filled by make_ocean_rider exactly as launch_ao_fluxes_lean fills it (fused net fluxes).
struct OceanRider {
    alignas(16) unsigned char args[1536];
    int n_chunks = 0;
    bool coare = false, valid = false;
};
hipError_t make_ocean_rider(const LaunchCfg& L, const DevParams& P, const LoopParams& C, const GridDesc& G, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f, const cf_sea_ice_fields* ice,
                            const cf_net_ocean_fluxes* net, const double* land, OceanRider* out);
// what may ride in the tail workgroups of the sea-ice interface launch (launch_ai_fluxes)
struct AiTail {
    const OceanRider* ocean = nullptr;              // this step's ocean solve (then the stresses cannot ride: they need its ρτ)
    const cf_atmos_source* next_src = nullptr;      // the next step's interpolation …
    const cf_interp_weights* w = nullptr;
    const cf_exchange_fields* next_out = nullptr;
    int interp_rows = 0, interp_blocks = 0;
    const DevParams* d_ocean_params = nullptr;      // … and this step's face stresses (compute_net_ocean_fluxes!)
    const cf_ocean_surface* stress_ocean = nullptr;
    const cf_interface_fluxes* stress_fluxes = nullptr;
    const cf_sea_ice_fields* stress_ice = nullptr;
    const cf_net_ocean_fluxes* stress_net = nullptr;
};
hipError_t launch_ai_fluxes(hipStream_t st, const LaunchCfg& L, const DevParams& P, const LoopParams& C, const IceParams& I,
                            const GridDesc& G, const cf_sea_ice_state* ice, const cf_ocean_surface* o,
                            const cf_exchange_fields* e, const cf_interface_fluxes* f, const double* d_tables,
                            const DevParams* d_params, uint8_t* d_trip, const AiTail* tail = nullptr, const NetIceOut* net_ice = nullptr);
// build_chunk_table / cf_debug_chunk_plan: `wet_per_chunk` = AO_PLAN_TAIL asks for the automatic plan of a launch that carries
// tail workgroups (CF_OPT_MERGED_PREFETCH = 2)
constexpr int AO_PLAN_TAIL = -2;
hipError_t build_chunk_table(hipStream_t st, const DevParams* d_params, const GridDesc& G, const void* mask, int cu_count,
                             int wet_per_chunk, int* d_sums, int* d_begins, int* d_meta, int* wet_per_chunk_out,
                             int* nchunks_out);
int chunk_table_capacity(int ncells);
int chunk_sums_capacity(int ncells);
hipError_t launch_net_sea_ice_fluxes(hipStream_t st, const DevParams& P, const GridDesc& G, const void* mask,
                                     const cf_sea_ice_state* ice, double albedo, double emissivity, double eps_sigma,
                                     double T_offset, const cf_exchange_fields* e, const cf_interface_fluxes* f,
                                     const double* frazil, const double* interface_heat, const cf_net_sea_ice_fluxes* out);
hipError_t launch_sea_ice_albedo(hipStream_t st, const cf_sea_ice_albedo_params& A, const GridDesc& G, const double* hi,
                                 const double* hs, const double* Ts, double* out);
hipError_t launch_sea_ice_ocean_fluxes(hipStream_t st, const DevParams& P, const cf_ice_ocean_params& Q, const GridDesc& G,
                                       const cf_ocean_surface* o, const double* conc, const double* tx, const double* ty,
                                       const cf_ice_ocean_fluxes* out);
constexpr int SALINITY_PARTIAL_BLOCKS = 512;
hipError_t launch_salinity_partial_sums(hipStream_t st, const DevParams& P, const GridDesc& G, const double* flux,
                                        const double* additional, const double* area, const void* mask, double* partial,
                                        int nblocks, double* sums);
hipError_t launch_salinity_subtract(hipStream_t st, const GridDesc& G, double* flux, const double* sums, double* mean_out);
hipError_t launch_debug_eval(hipStream_t st, const LaunchCfg& L, int fn, int n, const double* x, double* y);
struct PeerMailbox;
struct PeerFields;
struct FoldFields;
hipError_t launch_peer_halo(hipStream_t st, const PeerMailbox& M, const PeerFields& F, const GridDesc& G, int rows,
                            unsigned long long seq, int* d_status);
hipError_t launch_fold_north(hipStream_t st, const FoldFields& F, const GridDesc& G, int rows);
hipError_t launch_copy(hipStream_t st, void* dst, const void* src, size_t bytes);

}  // namespace coflux
