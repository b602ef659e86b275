// coflux_window.cpp — the JRA55 snapshot window in HBM (cf_window_*, include/coflux.h):
// device slots, pinned staging mirrors, a copy stream, and the events that order it against the context's
// compute stream in both directions.
#include "coflux_ctx.hpp"

// JRA55 snapshot window (cf_window_*): n_slots snapshots × nine variables in HBM + pinned staging mirrors.
struct cf_window {
    cf_ctx* ctx = nullptr;
    int ns_x = 0, ns_y = 0, n_slots = 0;
    size_t plane = 0;                       // floats per (slot, variable)
    float* d_data[CF_JRA55_NVARS] = {};     // device, each [n_slots][ns_y][ns_x]
    float* h_data[CF_JRA55_NVARS] = {};     // pinned host mirror, same layout
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_compute = nullptr;        // "everything queued on the compute stream so far"
    std::vector<hipEvent_t> ev_uploaded;    // per slot: its H2D copies have landed
    std::vector<int64_t> time_index;        // per slot: snapshot it holds, or INT64_MIN
    std::vector<char> in_flight;            // per slot: an upload was started and not yet waited for by the host
    std::vector<char> ordered;              // per slot: `ordered_on` already waits for the slot's latest upload
    hipStream_t ordered_on = nullptr;       // the compute stream those waits were queued on
};

extern "C" {

// ---------------------------------------------------------------------------------------------
// JRA55 snapshot window
// ---------------------------------------------------------------------------------------------
int cf_window_create(cf_ctx* ctx, int32_t ns_x, int32_t ns_y, int32_t n_slots, cf_window** out) {
    if (!ctx || !out) return fail(ctx, CF_ERR_INVALID, "cf_window_create: NULL argument");
    *out = nullptr;
    if (ns_x < 2 || ns_y < 2 || n_slots < 2)
        return fail(ctx, CF_ERR_INVALID, "cf_window_create: source grid %dx%d with %d slots (need >= 2 each)", ns_x, ns_y, n_slots);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    cf_window* w = new cf_window;
    w->ctx = ctx;
    w->ns_x = ns_x;
    w->ns_y = ns_y;
    w->n_slots = n_slots;
    w->plane = (size_t)ns_x * ns_y;
    w->time_index.assign(n_slots, INT64_MIN);
    w->in_flight.assign(n_slots, 0);
    w->ordered.assign(n_slots, 0);
    w->ev_uploaded.assign(n_slots, nullptr);
    const size_t bytes = w->plane * n_slots * sizeof(float);
    bool ok = hipStreamCreateWithFlags(&w->copy_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&w->ev_compute, hipEventDisableTiming) == hipSuccess;
    for (int s = 0; ok && s < n_slots; ++s) ok = hipEventCreateWithFlags(&w->ev_uploaded[s], hipEventDisableTiming) == hipSuccess;
    for (int v = 0; ok && v < CF_JRA55_NVARS; ++v)
        ok = hipMalloc((void**)&w->d_data[v], bytes) == hipSuccess &&
             hipHostMalloc((void**)&w->h_data[v], bytes, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        cf_window_destroy(w);
        return fail(ctx, CF_ERR_HIP, "cf_window_create: allocating %d slots of %dx%d failed", n_slots, ns_x, ns_y);
    }
    *out = w;
    return CF_OK;
}

int cf_window_destroy(cf_window* w) {
    if (!w) return CF_OK;
    if (w->copy_stream) (void)hipStreamSynchronize(w->copy_stream);
    for (int v = 0; v < CF_JRA55_NVARS; ++v) {
        if (w->d_data[v]) (void)hipFree(w->d_data[v]);
        if (w->h_data[v]) (void)hipHostFree(w->h_data[v]);
    }
    for (hipEvent_t e : w->ev_uploaded)
        if (e) (void)hipEventDestroy(e);
    if (w->ev_compute) (void)hipEventDestroy(w->ev_compute);
    if (w->copy_stream) (void)hipStreamDestroy(w->copy_stream);
    delete w;
    return CF_OK;
}

float* cf_window_host_buffer(cf_window* w, int32_t slot, int32_t variable) {
    if (!w || slot < 0 || slot >= w->n_slots || variable < 0 || variable >= CF_JRA55_NVARS) return nullptr;
    return w->h_data[variable] + (size_t)slot * w->plane;
}

int cf_window_wait_slot(cf_window* w, int32_t slot) {
    if (!w) return fail(nullptr, CF_ERR_INVALID, "window is NULL");
    if (slot < 0 || slot >= w->n_slots) return fail(w->ctx, CF_ERR_INVALID, "slot %d outside [0, %d)", slot, w->n_slots);
    if (w->in_flight[slot]) {
        HIP_TRY(w->ctx, hipSetDevice(w->ctx->device));  // a reader thread calls this: its current device may differ
        HIP_TRY(w->ctx, hipEventSynchronize(w->ev_uploaded[slot]));
        w->in_flight[slot] = 0;
    }
    return CF_OK;
}

int cf_window_commit(cf_window* w, int32_t slot, int64_t time_index) {
    if (!w) return fail(nullptr, CF_ERR_INVALID, "window is NULL");
    cf_ctx* ctx = w->ctx;
    if (slot < 0 || slot >= w->n_slots) return fail(ctx, CF_ERR_INVALID, "slot %d outside [0, %d)", slot, w->n_slots);
    HIP_TRY(ctx, hipSetDevice(ctx->device));  // one process may drive several contexts / devices
    // An interpolation requested AHEAD that reads this slot (cf_prefetch_atmosphere_state, cf_time_steps with
    // CF_PIPELINE_CONTINUING) was computed — or would be — from the snapshot this commit replaces: it is void.  The step it
    // was meant for then interpolates again, from the window's current contents.
    for (auto& p : ctx->prefetch)
        if (p.valid && (p.level1 == slot || p.level2 == slot)) {
            if (!p.on_main && p.done) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, p.done, 0));  // its writes must not race the redo
            p.valid = false;
        }
    if (ctx->deferred.valid && (ctx->deferred.src.level1 == slot || ctx->deferred.src.level2 == slot)) ctx->deferred.valid = false;
    // the device copy of this slot may only be overwritten once every interpolation already queued has read it
    HIP_TRY(ctx, hipEventRecord(w->ev_compute, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(w->copy_stream, w->ev_compute, 0));
    const size_t off = (size_t)slot * w->plane, bytes = w->plane * sizeof(float);
    for (int v = 0; v < CF_JRA55_NVARS; ++v)
        HIP_TRY(ctx, hipMemcpyAsync(w->d_data[v] + off, w->h_data[v] + off, bytes, hipMemcpyHostToDevice, w->copy_stream));
    HIP_TRY(ctx, hipEventRecord(w->ev_uploaded[slot], w->copy_stream));
    w->time_index[slot] = time_index;
    w->in_flight[slot] = 1;
    w->ordered[slot] = 0;
    return CF_OK;
}

int cf_window_upload(cf_window* w, int64_t time_index, const float* const* host_vars) {
    if (!w || !host_vars) return fail(w ? w->ctx : nullptr, CF_ERR_INVALID, "cf_window_upload: NULL argument");
    const int slot = (int)(((time_index % w->n_slots) + w->n_slots) % w->n_slots);
    CHECK(cf_window_wait_slot(w, slot));
    for (int v = 0; v < CF_JRA55_NVARS; ++v) {
        if (!host_vars[v]) return fail(w->ctx, CF_ERR_INVALID, "cf_window_upload: variable %d is NULL", v);
        std::memcpy(w->h_data[v] + (size_t)slot * w->plane, host_vars[v], w->plane * sizeof(float));
    }
    return cf_window_commit(w, slot, time_index);
}

int cf_window_find(cf_window* w, int64_t time_index) {
    if (!w) return -1;
    const int slot = (int)(((time_index % w->n_slots) + w->n_slots) % w->n_slots);
    return w->time_index[slot] == time_index ? slot : -1;
}

int cf_window_source(cf_window* w, int64_t n1, int64_t n2, double time_fraction, cf_atmos_source* out) {
    if (!w || !out) return fail(w ? w->ctx : nullptr, CF_ERR_INVALID, "cf_window_source: NULL argument");
    cf_ctx* ctx = w->ctx;
    const int s1 = cf_window_find(w, n1), s2 = cf_window_find(w, n2);
    if (s1 < 0 || s2 < 0)
        return fail(ctx, CF_ERR_INVALID, "snapshot %lld is not in the window (time_indices_in_memory = %d)",
                    (long long)(s1 < 0 ? n1 : n2), w->n_slots);
    if (!(time_fraction >= 0.0 && time_fraction <= 1.0))
        return fail(ctx, CF_ERR_INVALID, "time fraction %g outside [0, 1]", time_fraction);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // One wait per upload and stream: stream order carries it to everything queued behind (a wait on an event that has
    // long fired still costs the queue a barrier packet — with two descriptors per step, +24 µs on an 86 µs step, measured)
    if (w->ordered_on != ctx->stream) {
        std::fill(w->ordered.begin(), w->ordered.end(), 0);
        w->ordered_on = ctx->stream;
    }
    for (const int s : {s1, s2})
        if (!w->ordered[s]) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, w->ev_uploaded[s], 0));
            w->ordered[s] = 1;
        }
    for (int v = 0; v < CF_JRA55_NVARS; ++v) out->data[v] = w->d_data[v];
    out->ns_x = w->ns_x;
    out->ns_y = w->ns_y;
    out->n_levels = w->n_slots;
    out->level1 = s1;
    out->level2 = s2;
    out->time_fraction = time_fraction;
    return CF_OK;
}

}  // extern "C"
