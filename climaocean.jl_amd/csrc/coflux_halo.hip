// coflux_halo.hip — halo rows of the latitude-slab decomposition without a collective library, and the
// tripolar fold (include/coflux.h: cf_peer_halo_*, cf_fold_north_halo).
//
// Peer-direct exchange.  At ≤ 93 KB per neighbour and step the exchange is pure latency, so it is ONE launch:
// two workgroups (one per direction) each (i) store this rank's boundary rows straight into the neighbour's
// mailbox — fine-grained device memory mapped through HIP IPC, i.e. plain stores that travel over xGMI —,
// (ii) publish the step's sequence number with a system-scope release, (iii) wait for the neighbour's number in
// the own mailbox (one lane polls, bounded, sleeping between polls) and (iv) copy the received rows into the halo
// rows.  No host round trip, no second launch, nothing for the solver to wait on except stream order.
#include <hip/hip_runtime.h>

#include "coflux_halo_device.hpp"
#include "coflux_kernel_types.hpp"
#include "coflux_kernels.h"

namespace coflux {

__global__ __launch_bounds__(PEER_BLOCK) void peer_halo_kernel(PeerMailbox M, PeerFields F, GridDesc G, int rows,
                                                               unsigned long long seq, int* __restrict__ status) {
    const int dir = (int)blockIdx.x;  // 0: south neighbour, 1: north neighbour
    char* remote = dir == 0 ? M.south : M.north;
    if (!remote) return;  // end of the slab ring (or a fold): nothing to exchange in this direction
    const int parity = (int)(seq & 1ull);
    const size_t row_doubles = (size_t)G.sj, per_field = (size_t)rows * row_doubles;
    // (i) my boundary rows → the neighbour's mailbox; seen from THERE they arrive from the opposite side
    {
        double* dst = mailbox_rows(M, remote, 1 - dir, parity);
        const size_t first_row = dir == 0 ? (size_t)G.hy : (size_t)(G.hy + G.ny - rows);
        for (int f = 0; f < F.n; ++f) {
            const double* src = F.ptr[f] + first_row * row_doubles;
            for (size_t n = threadIdx.x; n < per_field; n += PEER_BLOCK) dst[(size_t)f * per_field + n] = src[n];
        }
    }
    // (ii) publish: every lane's stores must be visible system-wide before the flag is
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(mailbox_flag(remote, 1 - dir, parity), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // (iii) wait for the neighbour's rows of the same step
    __shared__ int ok;
    if (threadIdx.x == 0) {
        unsigned long long* flag = mailbox_flag(M.mine, dir, parity);
        unsigned long long spins = 0;
        int good = 1;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > PEER_SPIN_LIMIT) {
                good = 0;
                break;
            }
        }
        ok = good;
        if (!good) atomicExch(status, 1 + dir);  // sticky: reported by the next cf_sync
    }
    __syncthreads();
    if (!ok) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: nothing cached of the mailbox survives the flag
    // (iv) mailbox → my halo rows
    {
        const double* src = mailbox_rows(M, M.mine, dir, parity);
        const size_t first_row = dir == 0 ? (size_t)(G.hy - rows) : (size_t)(G.hy + G.ny);
        for (int f = 0; f < F.n; ++f) {
            double* dst = F.ptr[f] + first_row * row_doubles;
            for (size_t n = threadIdx.x; n < per_field; n += PEER_BLOCK) dst[n] = src[(size_t)f * per_field + n];
        }
    }
}

hipError_t launch_peer_halo(hipStream_t st, const PeerMailbox& M, const PeerFields& F, const GridDesc& G, int rows,
                            unsigned long long seq, int* d_status) {
    hipLaunchKernelGGL(peer_halo_kernel, dim3(2), dim3(PEER_BLOCK), 0, st, M, F, G, rows, seq, d_status);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Tripolar fold: the north halo rows of the last slab from its own mirrored interior rows
// (include/coflux.h: cf_fold_north_halo; Oceananigans zipper boundary condition, UPSTREAM-RECALL).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_north_kernel(FoldFields F, GridDesc G, int rows) {
    const int f = (int)blockIdx.y;
    const int width = G.nx + 2 * G.hx;
    const int n = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (n >= width * rows) return;
    const int r = n / width + 1;        // halo row ny − 1 + r
    const int i = n - (r - 1) * width - G.hx;  // interior-relative column, x-halos included
    int ip = i % G.nx;                  // periodic image inside [0, nx)
    if (ip < 0) ip += G.nx;
    const int loc = F.location[f];
    double sign = F.sign[f];
    int is, js;
    if (loc == CF_FOLD_X_FACE) {
        is = G.nx - ip;
        if (is >= G.nx) {  // the face on the fold axis maps onto itself
            is -= G.nx;
            sign = fabs(sign);
        }
        js = G.ny - 1 - r;
    } else if (loc == CF_FOLD_Y_FACE) {
        is = G.nx - 1 - ip;
        js = G.ny - r;
    } else {
        is = G.nx - 1 - ip;
        js = G.ny - 1 - r;
    }
    double* p = F.ptr[f];
    p[cell_index(G, i, G.ny - 1 + r)] = sign * p[cell_index(G, is, js)];
}

hipError_t launch_fold_north(hipStream_t st, const FoldFields& F, const GridDesc& G, int rows) {
    const int n = (G.nx + 2 * G.hx) * rows;
    hipLaunchKernelGGL(fold_north_kernel, dim3((n + 255) / 256, F.n), dim3(256), 0, st, F, G, rows);
    return hipGetLastError();
}

}  // namespace coflux
