// anyorder_test.hip — does hipExtAnyOrderLaunch let two consecutive kernels of ONE stream overlap on gfx950?  (scratch)
// kernel A: 256 workgroups busy for ~50 us (one per CU); kernel B: 256 workgroups busy for ~50 us.  Serialized: ~100 us per pair;
// overlapped: ~50 us.   build: hipcc -O3 --offload-arch=gfx950 scratch/anyorder_test.hip -o scratch/anyorder_test
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void busy(double* out, int iters) {
    double a = 1.0 + threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) a = __builtin_fma(a, 1.0000001, 1e-12);
    if (a == 12345.0) out[0] = a;
}
int main() {
    double* d; CHECK(hipMalloc(&d, 8));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int iters = 12000;
    for (int flag : {0, 1}) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(a, st));
            for (int n = 0; n < 50; ++n) {
                hipExtLaunchKernelGGL(busy, dim3(256), dim3(256), 0, st, nullptr, nullptr, 0, d, iters);
                hipExtLaunchKernelGGL(busy, dim3(256), dim3(256), 0, st, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, d, iters);
            }
            CHECK(hipEventRecord(b, st)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            printf("second kernel of each pair %s: %.1f us per pair\n", flag ? "hipExtAnyOrderLaunch" : "in order", ms * 1e3 / 50);
        }
    }
    return 0;
}
