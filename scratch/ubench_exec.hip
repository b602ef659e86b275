// ubench_exec.hip — does a wave64 FP64 instruction cost less when only part of the wave is enabled?  (scratch)
// v_fma_f64 chains under exec masks: all 64 lanes, lanes < 48 / 32 / 16, lanes 0-15 + 32-47, every 4th lane, one lane.
// Prints cycles per wave-instruction with one wave per SIMD.  build: hipcc -O3 --offload-arch=gfx950 -o ubench_exec ubench_exec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
constexpr int ITER = 4000;
template <int MODE>
__global__ __launch_bounds__(256) void k(long long* out, double seed) {
    const int lane = threadIdx.x & 63;
    double r[8];
    for (int n = 0; n < 8; ++n) r[n] = seed + 1e-3 * (threadIdx.x + n);
    double a = 1.0000001, b = 1e-9;
    asm volatile("" : "+v"(a), "+v"(b));
    bool on = true;
    if (MODE == 1) on = lane < 48;
    if (MODE == 2) on = lane < 32;
    if (MODE == 3) on = lane < 16;
    if (MODE == 4) on = (lane & 16) == 0;
    if (MODE == 5) on = (lane & 3) == 0;
    if (MODE == 6) on = lane == 5;
    if (MODE == 7) on = lane >= 48;
    const long long t0 = __builtin_readcyclecounter();
    if (on) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < 8; ++n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[n]) : "v"(a), "v"(b));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int n = 0; n < 8; ++n) s += r[n];
    if (s == 12345.678) out[0] = 1;
    if (lane == (MODE == 6 ? 5 : (MODE == 7 ? 48 : 0))) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}
template <int MODE>
void run(const char* name, long long* d) {
    const int blocks = 256;
    hipMemset(d, 0, sizeof(long long) * blocks * 4);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), d, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-28s %.2f cycles per wave-instruction (median wave, s_memtime ticks scaled x? see note)\n", name, (double)h[h.size() / 2] / (ITER * 32.0));
}
int main() {
    long long* d;
    hipMalloc(&d, sizeof(long long) * 256 * 4);
    run<0>("all 64 lanes", d);
    run<1>("lanes 0-47", d);
    run<2>("lanes 0-31", d);
    run<3>("lanes 0-15", d);
    run<4>("lanes 0-15 and 32-47", d);
    run<5>("every 4th lane", d);
    run<6>("one lane", d);
    run<7>("lanes 48-63", d);
    return 0;
}
