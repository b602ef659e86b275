// coflux_steps.cpp — what surrounds update_state! in a multi-step, multi-GPU run (include/coflux.h):
//   * cf_prefetch_atmosphere_state: the next step's interpolation on an auxiliary stream,
//   * cf_peer_halo_*: latitude-slab halo rows through HIP-IPC-mapped mailboxes (coflux_halo.hip),
//   * cf_fold_north_halo: the tripolar fold on the last slab,
//   * cf_time_steps: run!(simulation) of a prescribed-ocean coupled model without returning to the host language.
#include <map>

#include "coflux_ctx.hpp"

// Mailboxes exported by contexts of THIS process: a handle that comes back to the process that made it cannot be
// opened through HIP IPC (and need not be): the pointer is used as is.  Several slabs per process happen in tests
// and in single-process multi-GPU drivers.
static std::mutex g_peer_mutex;
static std::map<std::string, std::pair<char*, int>> g_peer_local;  // handle bytes → (mailbox, device)


static int ensure_aux_stream(cf_ctx* ctx) {
    if (ctx->aux_stream) return CF_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_aux_gate, hipEventDisableTiming));
    for (auto& p : ctx->prefetch) HIP_TRY(ctx, hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    return CF_OK;
}

// Launch the interpolation of `src` into `out` on the auxiliary stream, gated on ev_aux_gate (recorded by the caller
// on the main stream at the point after which `out` is free to be overwritten).
int cf_launch_prefetch(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w, const cf_exchange_fields* out) {
    cf_ctx::Prefetch* rec = nullptr;
    for (auto& p : ctx->prefetch)
        if (p.valid && p.key == out->u) rec = &p;
    if (!rec)
        for (auto& p : ctx->prefetch)
            if (!p.valid) rec = &p;
    if (!rec) return fail(ctx, CF_ERR_INVALID, "two prefetched atmosphere states are already pending");
    if (!ctx->deferred.gated) HIP_TRY(ctx, hipEventRecord(ctx->ev_aux_gate, ctx->stream));  // (a later gate than needed: still ordered)
    ctx->deferred.gated = false;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_aux_gate, 0));
    // the LDS-free gather kernel (≤ 56 VGPRs, no LDS): it is resident BESIDE the solver's workgroups, which fill
    // the CU's LDS and 456 of 512 registers per SIMD; the tiled kernel would wait for them to retire
    HIP_TRY(ctx, launch_interpolate_background(ctx->aux_stream, ctx->grid, src, w, out));
    HIP_TRY(ctx, hipEventRecord(rec->done, ctx->aux_stream));
    rec->key = out->u;
    rec->level1 = src->level1;
    rec->level2 = src->level2;
    rec->tf = src->time_fraction;
    rec->valid = true;
    rec->on_main = false;
    return CF_OK;
}

// A deferred request goes out now (cf_update_state calls this right after its solver launch; cf_sync and a second
// request flush it).  The gate event was recorded when the request was made.
int cf_flush_deferred_prefetch(cf_ctx* ctx) {
    if (!ctx->deferred.valid) return CF_OK;
    ctx->deferred.valid = false;
    return cf_launch_prefetch(ctx, &ctx->deferred.src, &ctx->deferred.w, &ctx->deferred.out);
}

static int request_prefetch(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                            const cf_exchange_fields* out, bool defer);

extern "C" {

int cf_prefetch_atmosphere_state(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                                 const cf_exchange_fields* out) {
    return request_prefetch(ctx, src, w, out, true);
}

}  // extern "C"

static int request_prefetch(cf_ctx* ctx, const cf_atmos_source* src, const cf_interp_weights* w,
                            const cf_exchange_fields* out, bool defer) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    if (!src || !w || !w->fi || !w->fj || !out || !out->u || !out->v || !out->T || !out->p || !out->q || !out->Qs ||
        !out->Ql || !out->Mp)
        return fail(ctx, CF_ERR_INVALID, "cf_prefetch_atmosphere_state: NULL argument");
    for (int v = 0; v < CF_JRA55_NVARS; ++v)
        if (!src->data[v]) return fail(ctx, CF_ERR_INVALID, "atmosphere source variable %d is NULL", v);
    if (src->level1 < 0 || src->level1 >= src->n_levels || src->level2 < 0 || src->level2 >= src->n_levels)
        return fail(ctx, CF_ERR_INVALID, "time levels (%d,%d) outside the %d levels in memory", src->level1, src->level2,
                    src->n_levels);
    CHECK(ensure_aux_stream(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    CHECK(cf_flush_deferred_prefetch(ctx));  // at most one request waits for a solver launch
    // An earlier request into the SAME exchange set that went out on the auxiliary stream and was never consumed (another
    // clock, a switch of CF_OPT_MERGED_PREFETCH in between): this request supersedes it, and whichever stream writes the
    // set next — the main stream in the merged forms — is ordered behind that kernel
    for (auto& p : ctx->prefetch)
        if (p.valid && !p.on_main && p.key == out->u) {
            if (p.done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p.done, 0));
            p.valid = false;
        }
    // the set written was last read by kernels already queued on the main stream: everything queued so far gates it.
    // With CF_OPT_MERGED_PREFETCH the request leaves inside a main-stream launch (stream order is the gate) and an event
    // per step would only put a barrier packet into the queue; the rare flush to the auxiliary stream records it then.
    ctx->deferred.gated = !(defer && ctx->merged_prefetch != 0);
    if (ctx->deferred.gated) HIP_TRY(ctx, hipEventRecord(ctx->ev_aux_gate, ctx->stream));
    if (!defer) return cf_launch_prefetch(ctx, src, w, out);
    ctx->deferred.src = *src;
    ctx->deferred.w = *w;
    ctx->deferred.out = *out;
    ctx->deferred.valid = true;
    return CF_OK;
}

// cf_destroy: the mailbox of a context that goes away must not be handed to a later cf_peer_halo_connect of this process
void cf_peer_forget_local(const char* mailbox) {
    std::lock_guard<std::mutex> lock(g_peer_mutex);
    for (auto it = g_peer_local.begin(); it != g_peer_local.end();)
        it = it->second.first == mailbox ? g_peer_local.erase(it) : std::next(it);
}

extern "C" {

// ---- peer-direct halo rows --------------------------------------------------------------------
int cf_peer_halo_export(cf_ctx* ctx, int max_fields, int max_rows, void* handle_out) {
    if (!ctx || !handle_out) return fail(ctx, CF_ERR_INVALID, "cf_peer_halo_export: NULL argument");
    if (max_fields < 1 || max_fields > PEER_MAX_FIELDS || max_rows < 1 || max_rows > ctx->grid.hy)
        return fail(ctx, CF_ERR_INVALID, "cf_peer_halo_export: %d fields (1…%d), %d rows (1…hy = %d)", max_fields,
                    PEER_MAX_FIELDS, max_rows, ctx->grid.hy);
    static_assert(sizeof(hipIpcMemHandle_t) <= CF_PEER_HANDLE_BYTES, "HIP IPC handle does not fit the ABI blob");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->peer.mine) return fail(ctx, CF_ERR_INVALID, "cf_peer_halo_export: this context already has a mailbox");
    const size_t slot = (size_t)max_fields * max_rows * ctx->grid.sj;
    const size_t bytes = PEER_FLAG_BYTES + 4 * slot * sizeof(double);
    void* p = nullptr;
    // fine-grained: stores arriving over xGMI and the owner's polling loads must meet in memory, not in a cache.
    // Mandatory: on coarse-grained memory the flag poll can spin to its timeout and — worse — the rows read
    // behind the flag can come from stale L2 lines, a silently wrong halo.  The caller falls back to the RCCL exchange.
    {
        const hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, CF_ERR_COMM, "cf_peer_halo_export: no fine-grained device memory for the halo mailbox (%s); use cf_halo_exchange_rows (RCCL)",
                        hipGetErrorString(e));
        }
    }
    HIP_TRY(ctx, hipMemset(p, 0, bytes));
    if (!ctx->d_peer_status) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_peer_status, sizeof(int)));
        HIP_TRY(ctx, hipMemset(ctx->d_peer_status, 0, sizeof(int)));
    }
    HIP_TRY(ctx, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    std::memset(&h, 0, sizeof h);
    hipError_t e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(ctx, CF_ERR_HIP, "hipIpcGetMemHandle on the halo mailbox: %s", hipGetErrorString(e));
    }
    ctx->peer.mine = (char*)p;
    ctx->peer.data_offset = PEER_FLAG_BYTES;
    ctx->peer.slot_doubles = slot;
    ctx->peer_bytes = bytes;
    ctx->peer_max_fields = max_fields;
    ctx->peer_max_rows = max_rows;
    std::memset(handle_out, 0, CF_PEER_HANDLE_BYTES);
    std::memcpy(handle_out, &h, sizeof h);
    {
        std::lock_guard<std::mutex> lock(g_peer_mutex);
        g_peer_local[std::string((const char*)handle_out, CF_PEER_HANDLE_BYTES)] = {(char*)p, ctx->device};
    }
    return CF_OK;
}

static int map_mailbox(cf_ctx* ctx, const void* handle, char** out, bool* opened) {
    *out = nullptr;
    *opened = false;
    if (!handle) return CF_OK;
    {
        std::lock_guard<std::mutex> lock(g_peer_mutex);
        auto it = g_peer_local.find(std::string((const char*)handle, CF_PEER_HANDLE_BYTES));
        if (it != g_peer_local.end()) {  // exported by this very process
            const int dev = it->second.second;
            if (dev != ctx->device) {
                int can = 0;
                HIP_TRY(ctx, hipDeviceCanAccessPeer(&can, ctx->device, dev));
                if (!can) return fail(ctx, CF_ERR_COMM, "device %d cannot access its neighbour's device %d", ctx->device, dev);
                hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(ctx, CF_ERR_HIP, "hipDeviceEnablePeerAccess(%d): %s", dev, hipGetErrorString(e));
                (void)hipGetLastError();
            }
            *out = it->second.first;
            return CF_OK;
        }
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof h);
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(ctx, CF_ERR_COMM, "hipIpcOpenMemHandle on a neighbour's halo mailbox: %s", hipGetErrorString(e));
    *out = (char*)p;
    *opened = true;
    return CF_OK;
}

int cf_peer_halo_connect(cf_ctx* ctx, const void* south_handle, const void* north_handle, int rank, int nranks) {
    if (!ctx || nranks <= 0 || rank < 0 || rank >= nranks) return fail(ctx, CF_ERR_INVALID, "cf_peer_halo_connect: bad arguments");
    if (!ctx->peer.mine) return fail(ctx, CF_ERR_INVALID, "cf_peer_halo_connect: call cf_peer_halo_export first");
    if ((rank > 0) != (south_handle != nullptr) || (rank < nranks - 1) != (north_handle != nullptr))
        return fail(ctx, CF_ERR_INVALID, "cf_peer_halo_connect: rank %d of %d needs %s south and %s north handle", rank, nranks,
                    rank > 0 ? "a" : "no", rank < nranks - 1 ? "a" : "no");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    CHECK(map_mailbox(ctx, south_handle, &ctx->peer.south, &ctx->peer_south_mapped));
    CHECK(map_mailbox(ctx, north_handle, &ctx->peer.north, &ctx->peer_north_mapped));
    ctx->rank = rank;
    ctx->nranks = nranks;
    ctx->peer_connected = true;
    return CF_OK;
}

int cf_halo_exchange_rows_peer(cf_ctx* ctx, double* const* d_fields, int nfields, int rows) {
    if (!ctx || !d_fields || nfields <= 0) return fail(ctx, CF_ERR_INVALID, "cf_halo_exchange_rows_peer: bad arguments");
    if (!ctx->peer_connected) return fail(ctx, CF_ERR_COMM, "cf_peer_halo_connect has not been called");
    const GridDesc& G = ctx->grid;
    if (nfields > ctx->peer_max_fields || rows < 1 || rows > ctx->peer_max_rows || rows > G.ny)
        return fail(ctx, CF_ERR_INVALID, "cf_halo_exchange_rows_peer: %d fields / %d rows exceed the mailbox (%d / %d) or ny = %d",
                    nfields, rows, ctx->peer_max_fields, ctx->peer_max_rows, G.ny);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PeerFields F{};
    F.n = nfields;
    for (int f = 0; f < nfields; ++f) {
        if (!d_fields[f]) return fail(ctx, CF_ERR_INVALID, "cf_halo_exchange_rows_peer: field %d is NULL", f);
        F.ptr[f] = d_fields[f];
    }
    return cf_peer_halo_launch_now(ctx, &F, rows);
}

int cf_peer_halo_launch_now(cf_ctx* ctx, const PeerFields* F, int rows) {
    ++ctx->peer_seq;  // every rank counts its exchanges: the same number names the same step everywhere
    HIP_TRY(ctx, launch_peer_halo(ctx->stream, ctx->peer, *F, ctx->grid, rows, ctx->peer_seq, ctx->d_peer_status));
    return CF_OK;
}

int cf_peer_halo_stats(cf_ctx* ctx, unsigned long long* exchanges, unsigned long long* in_solver_launch) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    if (exchanges) *exchanges = ctx->peer_seq;
    if (in_solver_launch) *in_solver_launch = ctx->halo_in_launch_count;
    return CF_OK;
}

int cf_fold_north_halo(cf_ctx* ctx, double* const* d_fields, const int* locations, const double* signs, int nfields,
                       int rows) {
    if (!ctx || !d_fields || !locations || !signs || nfields <= 0 || nfields > PEER_MAX_FIELDS)
        return fail(ctx, CF_ERR_INVALID, "cf_fold_north_halo: bad arguments (1…%d fields)", PEER_MAX_FIELDS);
    const GridDesc& G = ctx->grid;
    if (rows < 1 || rows > G.hy || rows + 1 > G.ny)
        return fail(ctx, CF_ERR_INVALID, "cf_fold_north_halo: rows = %d outside [1, min(hy, ny − 1)]", rows);
    if (G.nx % 2) return fail(ctx, CF_ERR_INVALID, "cf_fold_north_halo: a tripolar grid has an even number of columns (nx = %d)", G.nx);
    FoldFields F{};
    F.n = nfields;
    for (int f = 0; f < nfields; ++f) {
        if (!d_fields[f] || locations[f] < CF_FOLD_CENTER || locations[f] > CF_FOLD_Y_FACE)
            return fail(ctx, CF_ERR_INVALID, "cf_fold_north_halo: field %d is NULL or has an unknown location", f);
        F.ptr[f] = d_fields[f];
        F.location[f] = locations[f];
        F.sign[f] = signs[f];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_fold_north(ctx->stream, F, G, rows));
    return CF_OK;
}

int cf_discard_prefetched_atmosphere_state(cf_ctx* ctx) {
    if (!ctx) return fail(nullptr, CF_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->deferred.valid = false;   // requested, not launched: never will be
    for (auto& p : ctx->prefetch)
        if (p.valid) {
            // launched on the auxiliary stream: whatever the caller does to that exchange set next is ordered behind it
            if (!p.on_main && p.done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p.done, 0));
            p.valid = false;
        }
    return CF_OK;
}

// ---- run!(simulation) -------------------------------------------------------------------------
int cf_time_steps(cf_ctx* ctx, int64_t first_step, int nsteps, const cf_run_schedule* S, const cf_atmos_source* src,
                  const cf_interp_weights* w, const cf_interface_fluxes* fluxes, const cf_sea_ice_fields* ice,
                  const cf_net_ocean_fluxes* net) {
    if (!ctx || !S || !src) return fail(ctx, CF_ERR_INVALID, "cf_time_steps: NULL argument");
    if (S->struct_size != (int32_t)sizeof(cf_run_schedule))
        return fail(ctx, CF_ERR_INVALID, "cf_run_schedule.struct_size = %d, library expects %zu", S->struct_size, sizeof(cf_run_schedule));
    if (nsteps < 0 || first_step < 0 || S->n_ocean_states < 1 || !S->ocean_states || !S->atmos)
        return fail(ctx, CF_ERR_INVALID, "cf_time_steps: bad schedule");
    if (S->pipeline < 0 || S->pipeline > CF_PIPELINE_CONTINUING) return fail(ctx, CF_ERR_INVALID, "cf_time_steps: pipeline = %d", S->pipeline);
    if (S->n_atmos_sets < 1 || S->n_atmos_sets > 2 || (S->pipeline && S->n_atmos_sets != 2))
        return fail(ctx, CF_ERR_INVALID, "cf_time_steps: %d exchange-field sets (pipelining needs 2)", S->n_atmos_sets);
    if (!(S->time_fraction >= 0.0) || !(S->time_fraction_increment >= 0.0) || S->first_level < 0 || S->first_level >= src->n_levels)
        return fail(ctx, CF_ERR_INVALID, "cf_time_steps: bad clock (fraction %g, increment %g, level %d of %d)", S->time_fraction,
                    S->time_fraction_increment, S->first_level, src->n_levels);
    if (S->halo_backend != CF_HALO_NONE && S->halo_rows != ctx->grid.ring + 1)
        return fail(ctx, CF_ERR_INVALID, "cf_time_steps: halo_rows = %d, but the ring row reads v[j+1]: rows must be ring + 1 = %d",
                    S->halo_rows, ctx->grid.ring + 1);
    auto source_at = [&](int64_t step) {
        cf_atmos_source s = *src;
        const double total = S->time_fraction + (double)step * S->time_fraction_increment;
        const double whole = std::floor(total);
        s.level1 = (int)((S->first_level + (int64_t)whole) % src->n_levels);
        s.level2 = (s.level1 + 1) % src->n_levels;
        s.time_fraction = total - whole;
        return s;
    };
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int n = 0; n < S->n_ocean_states; ++n) {
        if (ctx->dev.mask_kind != CF_MASK_NONE && !S->ocean_states[n].mask) return fail(ctx, CF_ERR_INVALID, "ocean state %d has no mask", n);
        // one static wet mask per schedule: another mask pointer would make cf_update_state rebuild the solver's schedule
        // INSIDE the step loop (a stream synchronisation behind a queued peer-direct halo kernel that waits for its
        // neighbour, and a rebuild per step when states alternate: ADVICE r2)
        if (S->ocean_states[n].mask != S->ocean_states[0].mask)
            return fail(ctx, CF_ERR_INVALID, "cf_time_steps: ocean state %d carries a different mask pointer than state 0; the states of a "
                        "schedule share one wet mask", n);
        if (n == 0) CHECK(cf_ensure_chunk_table(ctx, S->ocean_states[n].mask));
    }
    static const int fold_loc[4] = {CF_FOLD_CENTER, CF_FOLD_CENTER, CF_FOLD_X_FACE, CF_FOLD_Y_FACE};
    static const double fold_sign[4] = {1.0, 1.0, -1.0, -1.0};
    if (S->pipeline && nsteps > 0) {  // the first step's atmosphere, unless a previous call already started it
        const cf_atmos_source s0 = source_at(first_step);
        const cf_exchange_fields* a0 = &S->atmos[first_step % 2];
        bool pending = false;
        for (auto& p : ctx->prefetch)
            pending |= p.valid && p.key == a0->u && p.level1 == s0.level1 && p.level2 == s0.level2 && p.tf == s0.time_fraction;
        if (!pending && ctx->deferred.valid && ctx->deferred.out.u == a0->u) {
            pending = ctx->deferred.src.level1 == s0.level1 && ctx->deferred.src.level2 == s0.level2 &&
                      ctx->deferred.src.time_fraction == s0.time_fraction;
            if (pending) CHECK(cf_flush_deferred_prefetch(ctx));
        }
        if (!pending) {
            if (ctx->merged_prefetch != 0) {
                // the merged forms keep everything on the context's stream: so does the first step's interpolation
                CHECK(cf_flush_deferred_prefetch(ctx));
                HIP_TRY(ctx, launch_interpolate(ctx->stream, ctx->launch, ctx->grid, &s0, w, a0));
                ctx->deferred.src = s0;
                ctx->deferred.w = *w;
                ctx->deferred.out = *a0;
                CHECK(deferred_went_out_on_main(ctx));
            } else {
                CHECK(request_prefetch(ctx, &s0, w, a0, false));
            }
        }
    }
    for (int64_t step = first_step; step < first_step + nsteps; ++step) {
        const cf_ocean_surface* o = &S->ocean_states[step % S->n_ocean_states];
        double* rows[4] = {const_cast<double*>(o->T), const_cast<double*>(o->S), const_cast<double*>(o->u),
                           const_cast<double*>(o->v)};
        if (S->halo_backend == CF_HALO_RCCL)
            CHECK(cf_halo_exchange_rows(ctx, rows, 4, S->halo_rows));
        else if (S->halo_backend == CF_HALO_PEER) {
            if (ctx->halo_in_launch && ctx->peer_connected && 4 <= ctx->peer_max_fields && S->halo_rows <= ctx->peer_max_rows &&
                S->halo_rows <= ctx->grid.ny) {
                // CF_OPT_HALO_IN_SOLVER_LAUNCH: left as a request — this step's solver launch carries the exchange as rider
                // workgroups where it can, cf_update_state issues the exchange kernel in front of the solver where it cannot
                ctx->halo_request.F = PeerFields{};
                ctx->halo_request.F.n = 4;
                for (int f = 0; f < 4; ++f) ctx->halo_request.F.ptr[f] = rows[f];
                ctx->halo_request.rows = S->halo_rows;
                ctx->halo_request.valid = true;
            } else {
                CHECK(cf_halo_exchange_rows_peer(ctx, rows, 4, S->halo_rows));
            }
        }
        if (S->fold_north) CHECK(cf_fold_north_halo(ctx, rows, fold_loc, fold_sign, 4, ctx->grid.ring + 1));
        const cf_atmos_source s = source_at(step);
        const cf_exchange_fields* a = &S->atmos[S->n_atmos_sets == 2 ? step % 2 : 0];
        // CF_PIPELINE_WITHIN_CALL: nothing beyond this call's steps is read or written.  CF_PIPELINE_CONTINUING: the last
        // step also requests step first_step + nsteps (the loop goes on in the next call, which finds that state pending
        // by its key; the caller promises the source levels of that step are resident and final, and that the other
        // exchange set is not read after this call's last step: coflux.h).
        // queued BEFORE this step's kernels: the set it overwrites was last read by the previous step's net fluxes
        if (S->pipeline && (step + 1 < first_step + nsteps || S->pipeline == CF_PIPELINE_CONTINUING)) {
            const cf_atmos_source sn = source_at(step + 1);
            CHECK(cf_prefetch_atmosphere_state(ctx, &sn, w, &S->atmos[(step + 1) % 2]));
        }
        CHECK(cf_update_state(ctx, &s, w, o, a, fluxes, ice, net));
    }
    return CF_OK;
}

}  // extern "C"
