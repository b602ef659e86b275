// coflux_ctx.hpp — the context object behind the C ABI and the error plumbing shared by its translation units
// (coflux_abi.cpp, coflux_window.cpp).  Internal: nothing here crosses the ABI.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/coflux.h"
#include "coflux_fast.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"
#include "coflux_tables.h"

using namespace coflux;

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

struct cf_ctx {
    int device = 0;
    GridDesc grid{};
    cf_flux_params params{};
    DevParams dev{};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    LoopParams fast{};
    DevParams* d_params = nullptr;
    LaunchCfg launch = [] {
        LaunchCfg L{};
        L.solver = CF_SOLVER_TABLES;
        L.interp_cap = 128;
        L.cu_count = 256;
        return L;
    }();
    uint8_t* d_trip = nullptr;       // trip count of the previous call per wet-list entry
    uint32_t* d_wet_pos = nullptr;   // static wet lists of the solver's chunks
    uint32_t* d_lean_sorted = nullptr;  // the lean ocean kernel's sorted lists (same capacity as d_wet_pos)
    int* d_lean_info = nullptr;         // per chunk: listed wet cells + fingerprint (4 ints)
    bool trip_hints = true;
    bool lean_hints = false;         // the lean ocean kernel sorts its lists by trip count only when CF_OPT_TRIP_HINTS = 1
    int merged_prefetch = 0;         // CF_OPT_MERGED_PREFETCH: a requested next-step interpolation rides in the face-stress launch
    double certified_budget = 8e-7;  // CF_OPT_CERTIFIED_BUDGET
    int fused_net = 2;               // cf_update_state: net fluxes in the solver's epilogue + a stress kernel: 0 never, 1 when possible, 2 with the lean ocean kernel
    // cost-balanced chunk table of the solver, rebuilt when the wet mask (pointer / kind / surface z) changes
    int* d_chunk_sums = nullptr;
    int* d_chunk_begins = nullptr;
    int* d_chunk_meta = nullptr;
    const void* chunk_mask = nullptr;
    int chunk_mask_kind = -1;
    double chunk_z_surface = 0.0;
    int chunk_wet = 0;      // wet cells per chunk actually used
    bool chunk_valid = false;
    double* d_reduce = nullptr;  // [2·SALINITY_PARTIAL_BLOCKS partial sums][2 totals]
    // atmosphere–sea-ice formulation (cf_set_sea_ice_formulation)
    bool ice_ready = false;
    cf_flux_params ice_params{};
    cf_sea_ice_params ice_props{};
    DevParams ice_dev{};
    LoopParams ice_loop{};
    IceParams ice_kernel{};
    bool ice_orbit_shortcut = true;  // CF_OPT_ICE_ORBIT_SHORTCUT
    bool ice_free_zero = false;      // CF_OPT_ICE_FREE_CELLS = CF_ICE_FREE_ZERO
    const double* d_land_freshwater = nullptr;   // cf_set_land_freshwater (borrowed)
    bool ice_albedo_ccsm3 = false;   // cf_set_sea_ice_albedo: SeaIceAlbedo(hi, hs, Ts) wherever no albedo field is given
    cf_sea_ice_albedo_params ice_albedo{};
    double* d_ice_albedo = nullptr;  // the albedo field of the current step (computed by the library)
    uint8_t* d_trip_ice = nullptr;   // trip counts of the sea-ice interface solve per wet-list entry
    size_t wet_list_entries = 0;     // entries allocated in d_wet_pos / d_trip / d_trip_ice
    double* d_ice_tables = nullptr;
    DevParams* d_ice_params = nullptr;
    // halo rows travel on their own stream so that they overlap the interpolation kernel, which
    // does not read the ocean state; consumers of the ocean fields wait on ev_comm_done
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_main_idle = nullptr, ev_comm_done = nullptr;
    bool comm_pending = false;
    double* d_tables = nullptr;
    int tables_kind = -1;
    std::string error;
    std::mutex error_mutex;  // cf_window_wait_slot may fail on a reader thread while the stepping thread reads the text
    // auxiliary stream: the next step's interpolation runs here while the current step's solver runs on `stream`
    // (cf_prefetch_atmosphere_state); one record per exchange-field set, matched by cf_update_state
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_aux_gate = nullptr;
    struct Prefetch {
        const double* key = nullptr;  // exchange set, identified by its u pointer
        int level1 = 0, level2 = 0;
        double tf = 0.0;
        hipEvent_t done = nullptr;
        bool valid = false;
        bool on_main = false;         // launched on the main stream (merged with the face stresses): stream order, no event
    } prefetch[2];
    // a prefetch that has been requested but not launched yet: it goes out right AFTER the next solver launch, so
    // that the solver's workgroups are dispatched first and the interpolation only fills what they leave free
    struct Deferred {
        bool valid = false;
        bool gated = false;  // ev_aux_gate was recorded when the request was made (not with merged_prefetch: the request
                             // normally leaves on the main stream, where stream order gates it; a flush records it late)
        cf_atmos_source src{};
        cf_interp_weights w{};
        cf_exchange_fields out{};
    } deferred;
    // peer-direct halo rows (coflux_halo.hip)
    PeerMailbox peer{};
    size_t peer_bytes = 0;
    int peer_max_fields = 0, peer_max_rows = 0;
    bool peer_connected = false;
    bool peer_south_mapped = false, peer_north_mapped = false;  // opened through HIP IPC (to be closed)
    unsigned long long peer_seq = 0;
    int* d_peer_status = nullptr;
    // the peer-direct exchange as riders of the solver launch (CF_OPT_HALO_IN_SOLVER_LAUNCH; coflux_lean_kernel.hpp, HALO)
    int halo_in_launch = 0;
    unsigned long long halo_in_launch_count = 0;            // exchanges that rode in a solver launch (cf_peer_halo_stats)
    unsigned long long* d_halo_counters = nullptr;          // [0,1] fields sent south / north, [2,3] fields received from there
    unsigned long long halo_expect_sent[2] = {0, 0}, halo_expect_done[2] = {0, 0};
    struct HaloRequest {                                    // cf_time_steps asked for this step's rows: the next solver launch carries
        bool valid = false;                                 // them, or cf_update_state issues the stand-alone kernel in front of it
        PeerFields F{};
        int rows = 0;
    } halo_request;
    // RCCL
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    // per-kernel event recorder (cf_profile_enable): 4 events per recorded update_state
    std::vector<hipEvent_t> prof_events;
    int prof_capacity = 0, prof_count = 0;
};

// sets the thread-local and the context's last-error text and returns `code`
int cf_fail(cf_ctx* ctx, int code, const char* fmt, ...);
#define fail cf_fail

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(ctx, CF_ERR_HIP, "%s:%d: %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

#define CHECK(call)            \
    do {                       \
        int rc_ = (call);      \
        if (rc_ != CF_OK) return rc_; \
    } while (0)

// coflux_steps.cpp: the stand-alone peer-direct exchange kernel for `F` (counts the exchange in ctx->peer_seq)
extern "C" __attribute__((visibility("hidden"))) int cf_peer_halo_launch_now(cf_ctx* ctx, const PeerFields* F, int rows);
// coflux_abi.cpp: books ctx->deferred as launched on the main stream (see cf_update_state)
extern "C" __attribute__((visibility("hidden"))) int deferred_went_out_on_main(cf_ctx* ctx);
