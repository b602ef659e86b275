// coflux_halo_device.hpp — the device side of the peer-direct halo rows (include/coflux.h: cf_peer_halo_*), shared by the
// stand-alone exchange kernel (coflux_halo.hip) and by the rider workgroups of the solver launch (coflux_lean_kernel.hpp, HALO).
#pragma once
#include <hip/hip_runtime.h>

#include "coflux_kernel_types.hpp"

namespace coflux {

// the waiting lane gives up after this many polls (s_sleep 8 ≈ 512 clocks + the load ≈ 1.3 µs each): ≈ 5 s
constexpr unsigned long long PEER_SPIN_LIMIT = 4ull * 1000 * 1000;

__device__ __forceinline__ double* mailbox_rows(const PeerMailbox& M, char* base, int side, int parity) {
    return reinterpret_cast<double*>(base + M.data_offset) + ((size_t)(side * 2 + parity)) * M.slot_doubles;
}
__device__ __forceinline__ unsigned long long* mailbox_flag(char* base, int side, int parity) {
    return reinterpret_cast<unsigned long long*>(base) + (side * 2 + parity) * 8;  // one flag per 64-byte line
}

// One rider workgroup of a solver launch: direction dir = h / F.n (0: south neighbour, 1: north), field f = h mod F.n.
//   (i)   my boundary rows of field f → the neighbour's mailbox (plain stores over xGMI), system-scope fence;
//   (ii)  count myself into counters[dir]; the workgroup that completes the direction's count publishes the step's sequence
//         number in the neighbour's mailbox (system-scope release: every field's rows are visible before the flag);
//   (iii) wait (one lane, bounded, sleeping) for the neighbour's number in my own mailbox;
//   (iv)  its rows of field f → my halo rows; agent-scope release of counters[2 + dir], which the solver workgroups that read
//         those rows wait for (ao_lean_body).  On a timeout the sticky status is set and the counter is released all the same:
//         the launch ends, the next cf_sync reports CF_ERR_COMM.
// Same mailbox protocol, same data movement as peer_halo_kernel — the rows that arrive are the same bits.
// (The rider's arguments arrive as scalars: the caller reads them from the kernel-argument block with uniform indices — a
// by-value HaloRider indexed by `f` would live in scratch.)
__device__ __forceinline__ void peer_halo_rider(const PeerMailbox& M, double* field, int f, int dir, int rows, unsigned long long seq,
                                                unsigned long long* counters, unsigned long long expect_sent, int* status,
                                                const GridDesc& G, int nthreads, int* lds_ok) {
    char* remote = dir == 0 ? M.south : M.north;
    if (!remote) return;  // end of the slab ring (or the fold): nothing to exchange, nobody waits
    const int parity = (int)(seq & 1ull), tid = (int)threadIdx.x;
    const size_t row_doubles = (size_t)G.sj, per_field = (size_t)rows * row_doubles;
    {
        double* dst = mailbox_rows(M, remote, 1 - dir, parity) + (size_t)f * per_field;
        const size_t first_row = dir == 0 ? (size_t)G.hy : (size_t)(G.hy + G.ny - rows);
        const double* src = field + first_row * row_doubles;
        for (size_t n = tid; n < per_field; n += nthreads) dst[n] = src[n];
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned long long before = __hip_atomic_fetch_add(&counters[dir], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1ull == expect_sent)
            __hip_atomic_store(mailbox_flag(remote, 1 - dir, parity), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
        *lds_ok = good;
        if (!good) atomicExch(status, 1 + dir);  // sticky: reported by the next cf_sync
    }
    __syncthreads();
    if (*lds_ok) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: nothing cached of the mailbox survives the flag
        const double* src = mailbox_rows(M, M.mine, dir, parity) + (size_t)f * per_field;
        const size_t first_row = dir == 0 ? (size_t)(G.hy - rows) : (size_t)(G.hy + G.ny);
        double* dst = field + first_row * row_doubles;
        for (size_t n = tid; n < per_field; n += nthreads) dst[n] = src[n];
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&counters[2 + dir], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// A solver workgroup whose cells read halo rows waits here (every wave; lane 0 polls, the wave re-converges behind it) until the
// riders have copied them in.  Bounded like the riders' own wait, plus their bound: a rider that timed out still releases.
__device__ __forceinline__ void peer_halo_wait(const unsigned long long* counter, unsigned long long expect) {
    if ((threadIdx.x & 63) == 0) {
        unsigned long long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > 4ull * PEER_SPIN_LIMIT) break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (every lane: the rows the riders wrote, not what this CU may have cached of them)
}

}  // namespace coflux
