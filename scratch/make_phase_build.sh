#!/bin/bash
# Builds scratch/libcoflux_phase.so: the production solver + per-wave time stamps and hardware ids (scratch/placement.py).
set -e
cd "$(dirname "$0")/../climaocean.jl_amd/csrc"
python3 - <<'PY'
s = open('coflux_solver.hip').read()
s = s.replace("namespace coflux {\n", "namespace coflux {\n__device__ unsigned long long g_stamp[4096 * 8];\n#define STAMP(q) do { if (lane == 0) g_stamp[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + (q)] = __builtin_readcyclecounter(); } while (0)\n", 1)
s = s.replace("    const int tid = threadIdx.x, lane = tid & 63;\n", "    const int tid = threadIdx.x, lane = tid & 63;\n    STAMP(0);\n    if (lane == 0) { g_stamp[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + 4] = __builtin_amdgcn_s_getreg(63492); g_stamp[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + 5] = __builtin_amdgcn_s_getreg(63508); }\n", 1)
s = s.replace("        // ---- phase 1: classify", "        STAMP(1);\n        // ---- phase 1: classify", 1)
s = s.replace("        // ---- phase 3: waves pull", "        STAMP(2);\n        // ---- phase 3: waves pull", 1)
s = s.replace("        __syncthreads();\n        if (tid < 64) {  // exclusive scan", "        STAMP(6);\n        __syncthreads();\n        if (tid < 64) {  // exclusive scan", 1)
s = s.replace("        const int nwet = counters[0];", "        const int nwet = counters[0];\n        if (lane == 0) g_stamp[((size_t)blockIdx.x * 4 + (tid >> 6)) * 8 + 7] = ((unsigned long long)(end - begin) << 32) | (unsigned)nwet;  /* RANGE_LEN */", 1)
s = s.replace("        if (end >= range_end) break;", "        STAMP(3);\n        if (end >= range_end) break;", 1)
s = s.replace("hipError_t launch_debug_eval(", "extern \"C\" int cf_debug_phase_read(unsigned long long* out, int n) {\n    hipDeviceSynchronize();\n    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n);\n    return 0;\n}\n\nhipError_t launch_debug_eval(", 1)
open('/tmp/_solver_phase.hip', 'w').write(s)
PY
cp /tmp/_solver_phase.hip ./_solver_phase.hip
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -c _solver_phase.hip -o /tmp/_solver_phase.o
rm -f _solver_phase.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libcoflux_phase.so coflux_interp.o /tmp/_solver_phase.o coflux_solver_libm.o coflux_seaice.o coflux_net.o coflux_abi.o coflux_window.o coflux_tables.o -ldl
echo built scratch/libcoflux_phase.so
