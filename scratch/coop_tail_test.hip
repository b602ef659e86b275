// coop_tail_test.hip — does a consumer workgroup of the SAME launch see a producer workgroup's stores without an L2
// write-back / invalidate, if the stores are write-through (sc0 sc1), the loads system-coherent (sc0 sc1) and the hand-off is
// one agent-scope relaxed atomic counter?  (scratch: the mechanism behind face stresses in the solver launch's tail)
// build: hipcc -O3 --offload-arch=gfx950 scratch/coop_tail_test.hip -o scratch/coop_tail_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ double value_of(unsigned seq, unsigned idx) { return (double)seq * 1e6 + (double)idx * 0.5; }

template <int SC>
__global__ __launch_bounds__(256) void coop_kernel(double* data, unsigned long long* counter, unsigned long long target, int n_prod, unsigned seq,
                                                    int n, unsigned* errors, unsigned* timeouts, int work) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    if ((int)blockIdx.x < n_prod) {
        // producer: some FP64 busy work (different per block: staggered retirement), then its slice
        double acc = 1.0 + tid * 1e-9;
        const int iters = work * (1 + (int)(blockIdx.x % 3));
        for (int it = 0; it < iters; ++it) acc = __builtin_fma(acc, 1.0000001, 1e-12);
        for (int idx = blockIdx.x * 256 + tid; idx < n; idx += n_prod * 256) {
            const double v = value_of(seq, (unsigned)idx) + (acc > 1e300 ? 1.0 : 0.0);
            const unsigned off = (unsigned)idx * 8u;
            if (SC == 3) asm volatile("global_store_dwordx2 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(data) : "memory");
            else if (SC == 2) asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(data) : "memory");
            else data[idx] = v;
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
        if (lane == 0) __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // consumer: wait for every producer wave of this launch, then check a scattered sample of the data
    int ok = 0;
    if (lane == 0) {
        for (int spin = 0; spin < (1 << 20); ++spin) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!ok) atomicAdd(timeouts, 1u);
    }
    ok = __shfl(ok, 0);
    if (!ok) return;
    const int c = (int)blockIdx.x - n_prod, nc = (int)gridDim.x - n_prod;
    unsigned bad = 0;
    for (int q0 = c * 256 + tid; q0 < n; q0 += nc * 256 * 4) {   // four loads in flight per lane, one wait
        double v[4];
        unsigned id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = min(q0 + u * nc * 256, n - 1);
            id[u] = (unsigned)(((unsigned long long)q * 7919ull) % (unsigned long long)n);
            const unsigned off = id[u] * 8u;
            if (SC == 3) asm volatile("global_load_dwordx2 %0, %1, %2 sc0 sc1" : "=v"(v[u]) : "v"(off), "s"(data) : "memory");
            else if (SC == 2) asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(v[u]) : "v"(off), "s"(data) : "memory");
            else v[u] = data[id[u]];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("" : "+v"(v[u]));
            bad += v[u] != value_of(seq, id[u]);
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <int SC>
static void run(const char* name, int launches, int work) {
    const int n = 810404, n_prod = 768, n_cons = 396;
    double* data; unsigned long long* counter; unsigned *errors, *timeouts;
    CHECK(hipMalloc(&data, n * sizeof(double)));
    CHECK(hipMalloc(&counter, 8)); CHECK(hipMalloc(&errors, 4)); CHECK(hipMalloc(&timeouts, 4));
    CHECK(hipMemset(counter, 0, 8)); CHECK(hipMemset(errors, 0, 4)); CHECK(hipMemset(timeouts, 0, 4)); CHECK(hipMemset(data, 0, n * sizeof(double)));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    for (int s = 1; s <= launches; ++s)
        coop_kernel<SC><<<n_prod + n_cons, 256, 52 * 1024>>>(data, counter, (unsigned long long)s * n_prod * 4, n_prod, (unsigned)s, n, errors, timeouts, work);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    unsigned e, t; CHECK(hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost));
    printf("%-34s work %6d: %d launches, %.2f us per launch, %u stale values, %u timed-out waves\n", name, work, launches, ms * 1e3 / launches, e, t);
}

int main() {
    for (int work : {0, 2000}) {
        run<3>("sc0 sc1 stores / loads", 300, work);
        run<2>("sc1 stores / loads", 300, work);
        run<0>("plain stores / loads", 300, work);
    }
    return 0;
}
