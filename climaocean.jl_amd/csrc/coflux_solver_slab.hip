// coflux_solver_slab.hip — compute_atmosphere_ocean_fluxes! (SOLVER_OCEAN_LEAN, omip_simulation.jl:40-49) for launches
// that leave every SIMD with ONE OR TWO waves: a latitude slab of a strongly scaled run (launch.sh:165 Partition(1,4),
// pbs_launch.sh:51 Partition(1,8) — a 1440×70 slab is 394 workgroups on 256 CUs), small surfaces.
//
// There the solver is not bound by FP64 issue but by ONE wave's dependent chain: ≈ 20 trips of ≈ 170 instructions that
// the hardware issues every ≈ 9 cycles when each waits for its predecessor and every ≈ 4 when it does not
// (scratch/ubench_lat.hip).  hipcc's gfx950 model prices a dependent FP64 instruction at its issue cost and never
// interleaves for latency, and after register allocation the temporaries share so few registers that re-ordering is
// pinned by write-after-read dependences.  So these kernels are
//   * the same body (ao_lean_body) with the iteration in its straight-line layout (mo_iterate_lean_line: first trip
//     peeled off, gustiness unconditional) — big basic blocks;
//   * compiled for two waves per SIMD (256 VGPRs; the production kernels: three, 168);
//   * re-scheduled AFTER register allocation by tools/gcn_sched.py (the Makefile's rule for this file): values local
//     to a block are renamed into the registers the compiler left free, then a critical-path list scheduler with the
//     measured gfx950 latencies orders each block and recomputes its s_waitcnt / hazard no-ops.  Same instructions,
//     same operand values: results are bitwise those of the production kernels (tests/test_slab_line.py).
// launch_ao_fluxes_lean picks them when the chunk plan has at most two workgroups per CU (CF_OPT_LATENCY_LAYOUT).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#define LEAN_STAMP(q) do { } while (0)
#define LEAN_STAMP_SET(q, v) do { } while (0)
#include "coflux_lean_kernel.hpp"

namespace coflux {

template <bool COARE, bool FUSE, bool TAIL, bool HALO = false>
__global__ __launch_bounds__(AO_BLOCK, 2) void ao_lean_line_kernel(LeanArgs unused_by_name) {
    ao_lean_body<COARE, FUSE, TAIL, false, true, HALO>((LeanArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), (int)blockIdx.x);
}

hipError_t launch_ao_lean_line(hipStream_t st, bool coare, bool fuse, bool tail, bool halo, int blocks, const LeanArgs& A) {
    if ((tail && !fuse) || (halo && !tail)) return hipErrorInvalidValue;
#define CF_LINE_LAUNCH(COARE_, FUSE_, TAIL_) \
    hipLaunchKernelGGL((ao_lean_line_kernel<COARE_, FUSE_, TAIL_>), dim3(blocks), dim3(AO_BLOCK), LeanGeom<AO_BLOCK>::LDS_BYTES, st, A)
    if (halo) {
        if (coare) hipLaunchKernelGGL((ao_lean_line_kernel<true, true, true, true>), dim3(blocks), dim3(AO_BLOCK), LeanGeom<AO_BLOCK>::LDS_BYTES, st, A);
        else hipLaunchKernelGGL((ao_lean_line_kernel<false, true, true, true>), dim3(blocks), dim3(AO_BLOCK), LeanGeom<AO_BLOCK>::LDS_BYTES, st, A);
    } else if (tail) {
        if (coare) CF_LINE_LAUNCH(true, true, true); else CF_LINE_LAUNCH(false, true, true);
    } else if (fuse) {
        if (coare) CF_LINE_LAUNCH(true, true, false); else CF_LINE_LAUNCH(false, true, false);
    } else {
        if (coare) CF_LINE_LAUNCH(true, false, false); else CF_LINE_LAUNCH(false, false, false);
    }
#undef CF_LINE_LAUNCH
    return hipGetLastError();
}

}  // namespace coflux
