#!/bin/bash
# Builds scratch/libcoflux_phase.so: the production solver + per-wave time stamps (scratch/phases.py).
# stamps per wave: 0 kernel entry, 4 table DMA issued, 5 list + mask loads issued, 1 after the first barrier, 2 list sorted,
# validated and tables landed (batches begin), 3 = 6 batches done (the wave retires), 7 = (range length, wet count)
set -e
cd "$(dirname "$0")/../climaocean.jl_amd/csrc"
python3 - <<'PY'
s = open('coflux_solver.hip').read()
def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)
rep("namespace coflux {\n", "namespace coflux {\n__device__ unsigned long long g_stamp[4096 * 8];\n#define STAMP(q) do { if (lane == 0) g_stamp[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + (q)] = __builtin_readcyclecounter(); } while (0)\n")
rep("    const int tid = threadIdx.x, lane = tid & 63;\n    const int wx = G.nx + 2 * G.ring;\n", "    const int tid = threadIdx.x, lane = tid & 63;\n    STAMP(0);\n    const int wx = G.nx + 2 * G.ring;\n")
rep("    constexpr int PER_THREAD = CHUNK / BLOCK;\n    constexpr int LAND_UNROLL = 8;", "    STAMP(4);\n    constexpr int PER_THREAD = CHUNK / BLOCK;\n    constexpr int LAND_UNROLL = 8;")
rep("    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only", "    STAMP(5);\n    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only")
rep("    const DevParams& P = *lp;  // prologue-only parameters live in LDS", "    STAMP(1);\n    const DevParams& P = *lp;  // prologue-only parameters live in LDS")
rep("        // ---- waves pull 64 wet cells at a time", "        STAMP(2);\n        if (lane == 0) g_stamp[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + 7] = ((unsigned long long)(end - begin) << 32) | (unsigned)nwet;\n        // ---- waves pull 64 wet cells at a time")
rep("        if (use_static || end >= range_end) break;", "        STAMP(3);\n        STAMP(6);\n        if (use_static || end >= range_end) break;")
rep("hipError_t launch_debug_eval(", "extern \"C\" int cf_debug_phase_read(unsigned long long* out, int n) {\n    hipDeviceSynchronize();\n    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n);\n    return 0;\n}\n\nhipError_t launch_debug_eval(")
open('_solver_phase.hip', 'w').write(s)
PY
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -c _solver_phase.hip -o /tmp/_solver_phase.o
rm -f _solver_phase.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libcoflux_phase.so coflux_interp.o /tmp/_solver_phase.o coflux_solver_lean.o coflux_solver_libm.o coflux_net.o coflux_halo.o coflux_abi.o coflux_window.o coflux_steps.o coflux_tables.o -ldl
echo built scratch/libcoflux_phase.so
