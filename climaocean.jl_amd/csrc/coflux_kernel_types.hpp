// Kernel-side argument bundles shared by the .hip translation units (plain device pointers; built
// from the C-ABI structs on the host).
#pragma once
#include <hip/hip_runtime.h>

#include "coflux_device.hpp"

namespace coflux {

constexpr int NUM_XCD = 8;

struct SourceDesc {
    const float* data[CF_JRA55_NVARS];
    int32_t ns_x, ns_y, level1, level2;
    double tf;
};

struct WeightDesc {
    const double* fi;
    const double* fj;
    const double* cos_rot;
    const double* sin_rot;
    const double* latitude;
    int32_t separable;
};

struct Exchange {
    double* u;
    double* v;
    double* T;
    double* p;
    double* q;
    double* Qs;
    double* Ql;
    double* Mp;
};

struct OceanIn {
    const double* T;
    const double* S;
    const double* u;
    const double* v;
    const void* mask;
};

struct FluxOut {
    double* Qc;
    double* Qv;
    double* Fv;
    double* tx;
    double* ty;
    double* Ts;
    double* ustar;
    double* tstar;
    double* qstar;
    int32_t* iters;
};

struct IceIn {
    const double* conc;
    const double* Qio;
    const double* Jsio;
    const double* txio;
    const double* tyio;
    const double* land;   // JRA55PrescribedLand freshwater (kg m⁻² s⁻¹) or nullptr — rides with the partition's inputs
};

struct IceStateIn {   // sea_ice.model fields the atmosphere–sea-ice interface reads (atmosphere.jl:34-39)
    const double* thickness;
    const double* top_temperature;
    const double* u;
    const double* v;
    const double* albedo;
    const double* concentration;  // ℵ: read only with CF_OPT_ICE_FREE_CELLS = zero (null otherwise)
};

struct NetIceOut {   // compute_net_sea_ice_fluxes! inside the interface solve's epilogue (all null: a launch of its own)
    const double* conc;
    const double* frazil;
    const double* interface_heat;
    double* top;
    double* bottom;
};

struct NetOut {
    double* u;
    double* v;
    double* T;
    double* S;
    double* sw;
    double* lw_up;
    double* lw_down;
    double* sw_down;
};

// peer-direct halo rows (coflux_halo.hip)
constexpr int PEER_BLOCK = 1024;
constexpr int PEER_MAX_FIELDS = 8;
constexpr int PEER_FLAG_BYTES = 4 * 64;  // [side][parity] sequence numbers, one 64-byte line each
struct PeerMailbox {
    char* mine;    // this rank's mailbox: flags, then [side][parity][field][row][sj] doubles
    char* south;   // the south / north neighbours' mailboxes mapped into this process (nullptr: none)
    char* north;
    size_t data_offset;   // bytes from the mailbox base to the row storage
    size_t slot_doubles;  // doubles per (side, parity) slot = max_fields · max_rows · sj
};
struct PeerFields {
    double* ptr[PEER_MAX_FIELDS];
    int n;
};
// The peer-direct exchange as RIDER workgroups of the solver launch (coflux_lean_kernel.hpp, HALO): one workgroup per
// (direction, field); `counters` (device memory, monotone): [0], [1] fields SENT towards the south / north neighbour, [2], [3]
// fields RECEIVED from them and copied into the halo rows.  expect_* = the counters' values once this launch's exchange is complete.
struct HaloRider {
    PeerMailbox M;
    PeerFields F;
    unsigned long long* counters;
    unsigned long long seq;
    unsigned long long expect_sent[2], expect_done[2];
    int* status;
    int rows;
    int blocks;       // 2 · F.n rider workgroups at the head of the launch, or 0: this launch carries no exchange
    int chunk_south;  // chunks [0, chunk_south) hold cells that read the south halo rows
    int chunk_north;  // chunks [chunk_north, n_chunks) hold cells that read the north halo rows
    int wait_south, wait_north;   // 1: a neighbour exists on that side
    int pad[2];
};
static_assert(sizeof(HaloRider) % 8 == 0, "argument bundles are made of 8-byte words");

struct FoldFields {
    double* ptr[PEER_MAX_FIELDS];
    double sign[PEER_MAX_FIELDS];
    int location[PEER_MAX_FIELDS];
    int n;
};

inline SourceDesc make_source(const cf_atmos_source* s) {
    SourceDesc S;
    for (int v = 0; v < CF_JRA55_NVARS; ++v) S.data[v] = s->data[v];
    S.ns_x = s->ns_x;
    S.ns_y = s->ns_y;
    S.level1 = s->level1;
    S.level2 = s->level2;
    S.tf = s->time_fraction;
    return S;
}

inline WeightDesc make_weights(const cf_interp_weights* w) {
    WeightDesc W{};
    if (w) {
        W.fi = w->fi;
        W.fj = w->fj;
        W.cos_rot = w->cos_rot;
        W.sin_rot = w->sin_rot;
        W.latitude = w->latitude;
        W.separable = w->separable;
    }
    return W;
}

inline Exchange make_exchange(const cf_exchange_fields* e) {
    return Exchange{e->u, e->v, e->T, e->p, e->q, e->Qs, e->Ql, e->Mp};
}
inline OceanIn make_ocean(const cf_ocean_surface* o) { return OceanIn{o->T, o->S, o->u, o->v, o->mask}; }
inline FluxOut make_fluxes(const cf_interface_fluxes* f) {
    return FluxOut{f->sensible_heat, f->latent_heat,       f->water_vapor,       f->x_momentum,     f->y_momentum,
                   f->temperature,   f->friction_velocity, f->temperature_scale, f->humidity_scale, f->iterations};
}

// FINAL: the launch assembles the net fluxes itself, so nothing in the step reads these fields again (ρτ excepted: the face-stress
// launch does) — streaming stores (coflux_solver_shared.hpp::gstore_final has the measurements)
template <bool FINAL = false>
__device__ __forceinline__ void store_fluxes(const FluxOut& F, size_t k, const CellFluxes& R) {
    auto put = [](double* p, double v) {
        if constexpr (FINAL) __builtin_nontemporal_store(v, p);
        else *p = v;
    };
    put(&F.Qc[k], R.Qc);
    put(&F.Qv[k], R.Qv);
    put(&F.Fv[k], R.Fv);
    F.tx[k] = R.rho_tau_x;
    F.ty[k] = R.rho_tau_y;
    put(&F.Ts[k], R.Ts_ocean);
    if (F.ustar) put(&F.ustar[k], R.ustar);
    if (F.tstar) put(&F.tstar[k], R.tstar);
    if (F.qstar) put(&F.qstar[k], R.qstar);
    if (F.iters) F.iters[k] = R.iterations;
}

// SeaIceAlbedo(hi, hs, Ts) — CCSM3 (include/coflux.h: cf_sea_ice_albedo_params).  `A` are the parameters by value.
__device__ __forceinline__ double ccsm3_albedo(const cf_sea_ice_albedo_params& A, double hi, double hs, double Ts) {
    const double fh = fmin(atan(4.0 * hi) / atan(4.0 * A.reference_thickness), 1.0);
    const double ao = A.ocean_albedo * (1.0 - fh);
    // fT = 0 below (T_melt − ΔT), −1 at the melting point
    const double fT = fmin((A.melting_temperature - Ts) / A.melt_temperature_range - 1.0, 0.0);
    const double ice_v = fmax(A.ice_visible * fh + ao + A.ice_melt_change * fT, A.ocean_albedo);
    const double ice_n = fmax(A.ice_near_infrared * fh + ao + A.ice_melt_change * fT, A.ocean_albedo);
    const double snow_v = A.snow_visible + A.snow_melt_change_visible * fT;
    const double snow_n = A.snow_near_infrared + A.snow_melt_change_near_infrared * fT;
    const double as = hs > 0.0 ? hs / (hs + A.snow_patch_thickness) : 0.0;
    const double v = ice_v * (1.0 - as) + snow_v * as, n = ice_n * (1.0 - as) + snow_n * as;
    return A.visible_fraction * v + (1.0 - A.visible_fraction) * n;
}

// ---------------------------------------------------------------------------------------------
// compute_net_ocean_fluxes!, per cell.  Two kernels evaluate it — net_flux_kernel (coflux_net.hip) and the
// solver's fused epilogue (coflux_solver.hip) — and must agree bit for bit, so the arithmetic lives here with
// floating-point contraction switched off: every product and sum is rounded the same way wherever it is inlined.
// ---------------------------------------------------------------------------------------------
struct NetCell {
    double JT, JS, sw, lw_up, lw_down, sw_down;
};

__device__ __forceinline__ NetCell net_cell_local(const DevParams& P, double alb, double aice, double So, double Ts_kelvin,
                                                  double Mp, double Qs, double Ql, double Qc, double Qv, double Mv,
                                                  double Qio, double Jsio, double Mland = 0.0) {
#pragma clang fp contract(off)
    NetCell C;
    const double T2 = Ts_kelvin * Ts_kelvin;
    const double Qu = P.emissivity * P.sigma * T2 * T2;
    const double Qal = -P.emissivity * Ql;
    const double Qts = -(1.0 - alb) * Qs * (1.0 - aice);
    const double Qss = P.penetrating_sw ? 0.0 : Qts;
    const double SQao = (Qu + Qc + Qv + Qal) * (1.0 - aice) + Qss;
    const double SFao = -Mp * P.rho_f_inv + Mv * P.rho_f_inv;
    const double SFs = (So < P.S_min && SFao < 0.0) ? 0.0 : SFao;
    const double roc = P.rho_o_inv * P.c_o_inv;
    C.JT = SQao * roc + Qio * roc;
    const double SFl = -Mland * P.rho_f_inv;                       // land freshwater (rivers, calving): not ice-masked
    const double SFls = (So < P.S_min && SFl < 0.0) ? 0.0 : SFl;
    C.JS = (1.0 - aice) * (-So * SFs) + Jsio + (-So * SFls);
    C.sw = Qts * roc;
    C.lw_up = Qu;
    C.lw_down = -Qal;
    C.sw_down = -Qts;
    return C;
}

// kinematic stress at a face from the two adjacent cell-centre stresses ρτ and the ice cover on the face
__device__ __forceinline__ double net_face_stress(const DevParams& P, double rho_tau_a, double rho_tau_b, double aice_a,
                                                  double aice_b, double tau_io) {
#pragma clang fp contract(off)
    const double tao = 0.5 * (rho_tau_a + rho_tau_b) * P.rho_o_inv;
    const double a = 0.5 * (aice_a + aice_b);
    return (1.0 - a) * tao + a * tau_io;
}

// compute_net_sea_ice_fluxes! of one wet interior cell: heat into the ice top (where there is ice) and into its bottom.
// One function, contraction off: net_sea_ice_flux_kernel and the interface solve's epilogue give the same bits.
__device__ __forceinline__ void net_sea_ice_cell(double albedo, double emissivity, double eps_sigma, double T_offset, double Qs, double Ql,
                                                 double Ts_celsius, double Qc, double Qv, double conc, double Qf, double Qi, double& top,
                                                 double& bottom) {
#pragma clang fp contract(off)
    const double T = Ts_celsius + T_offset, T2 = T * T;
    const double Qu = eps_sigma * T2 * T2;
    const double Qd = -(1.0 - albedo) * Qs - emissivity * Ql;
    top = conc > 0.0 ? (Qd + Qu + Qc + Qv) : 0.0;
    bottom = Qf + Qi;
}

template <bool FINAL = false>
__device__ __forceinline__ void store_net_cell(const NetOut& N, size_t k, const NetCell& C) {
    auto put = [](double* p, double v) {
        if constexpr (FINAL) __builtin_nontemporal_store(v, p);
        else *p = v;
    };
    put(&N.T[k], C.JT);
    put(&N.S[k], C.JS);
    if (N.sw) put(&N.sw[k], C.sw);
    if (N.lw_up) put(&N.lw_up[k], C.lw_up);
    if (N.lw_down) put(&N.lw_down[k], C.lw_down);
    if (N.sw_down) put(&N.sw_down[k], C.sw_down);
}

}  // namespace coflux
