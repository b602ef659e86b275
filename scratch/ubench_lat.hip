// ubench_lat.hip — DEPENDENT-chain latency of the gfx950 instructions the flux solver's iteration is made of (scratch).
// One wave per SIMD (a latitude slab's regime: every wave has its SIMD to itself), one dependent chain of 32 × ITER
// instructions, timed with s_memtime; also k independent chains interleaved (k = 2, 3, 4) to see where the issue rate
// takes over, and an LDS pointer chase.  Prints cycles per instruction.
// build: hipcc -O3 --offload-arch=gfx950 scratch/ubench_lat.hip -o scratch/ubench_lat
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 500;

#define KD(NAME, K, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed) {                             \
        double r[K];                                                                                       \
        for (int n = 0; n < K; ++n) r[n] = seed + 1e-3 * (threadIdx.x + n);                                \
        double a = 1.0000001, b = 1e-9;                                                                    \
        asm volatile("" : "+v"(a), "+v"(b));                                                               \
        const long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < ITER; ++it) {                                                                \
            _Pragma("unroll") for (int u = 0; u < 32 / K; ++u) {                                           \
                _Pragma("unroll") for (int n = 0; n < K; ++n) asm volatile(ASM : "+v"(r[n]) : "v"(a), "v"(b)); \
            }                                                                                              \
        }                                                                                                  \
        const long long t1 = __builtin_readcyclecounter();                                                \
        double s = 0;                                                                                      \
        for (int n = 0; n < K; ++n) s += r[n];                                                             \
        if (s == 12345.678) out[0] = 1;                                                                    \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                \
    }
#define KF(NAME, K, ASM)                                                                                   \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed) {                             \
        float r[K];                                                                                        \
        for (int n = 0; n < K; ++n) r[n] = (float)seed + 1e-3f * (threadIdx.x + n);                        \
        float a = 1.0000001f, b = 1e-9f;                                                                   \
        asm volatile("" : "+v"(a), "+v"(b));                                                               \
        const long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < ITER; ++it) {                                                                \
            _Pragma("unroll") for (int u = 0; u < 32 / K; ++u) {                                           \
                _Pragma("unroll") for (int n = 0; n < K; ++n) asm volatile(ASM : "+v"(r[n]) : "v"(a), "v"(b)); \
            }                                                                                              \
        }                                                                                                  \
        const long long t1 = __builtin_readcyclecounter();                                                \
        float s = 0;                                                                                       \
        for (int n = 0; n < K; ++n) s += r[n];                                                             \
        if (s == 12345.678f) out[0] = 1;                                                                   \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                \
    }
// a 64-bit value through a 32-bit one and back (two instructions per step)
#define KDF(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed) {                             \
        double r = seed + 1e-3 * threadIdx.x;                                                              \
        float f = 0.f;                                                                                     \
        const long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < ITER; ++it) {                                                                \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) asm volatile(ASM : "+v"(r), "+v"(f));           \
        }                                                                                                  \
        const long long t1 = __builtin_readcyclecounter();                                                \
        if (r + f == 12345.678) out[0] = 1;                                                                \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;                \
    }

KD(fma64_1, 1, "v_fma_f64 %0, %0, %1, %2")
KD(fma64_2, 2, "v_fma_f64 %0, %0, %1, %2")
KD(fma64_3, 4, "v_fma_f64 %0, %0, %1, %2")
KD(mul64_1, 1, "v_mul_f64 %0, %0, %1")
KD(add64_1, 1, "v_add_f64 %0, %0, %2")
KD(max64_1, 1, "v_max_f64 %0, %0, %2")
KD(rcp64_1, 1, "v_rcp_f64 %0, %0")
KD(rcp64_2, 2, "v_rcp_f64 %0, %0")
KD(rsq64_1, 1, "v_rsq_f64 %0, %0")
KD(rndne64_1, 1, "v_rndne_f64 %0, %0")
KD(ldexp64_1, 1, "v_ldexp_f64 %0, %0, 1")
KD(mov64_1, 1, "v_mov_b64 %0, %0")
KD(rcp_fma_1, 1, "v_rcp_f64 %0, %0\n s_nop 0\n v_fma_f64 %0, %0, %1, %2")
KF(fma32_1, 1, "v_fma_f32 %0, %0, %1, %2")
KF(fma32_2, 2, "v_fma_f32 %0, %0, %1, %2")
KF(mul32_1, 1, "v_mul_f32 %0, %0, %1")
KF(add_u32_1, 1, "v_add_u32 %0, %0, %1")
KF(add_u32_2, 2, "v_add_u32 %0, %0, %1")
KF(and32_1, 1, "v_and_b32 %0, %0, %1")
KF(lshl_add_1, 1, "v_lshl_add_u32 %0, %0, 1, %1")
KF(log32_1, 1, "v_log_f32 %0, %0\n s_nop 0")
KF(exp32_1, 1, "v_exp_f32 %0, %0\n s_nop 0")
KF(rcp32_1, 1, "v_rcp_f32 %0, %0\n s_nop 0")
KF(cmp_cnd_vcc_1, 1, "v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc")
KF(cmp_cnd_sgpr_1, 1, "v_cmp_lt_f32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KDF(cvt_round_1, "v_cvt_f32_f64 %1, %0\n v_cvt_f64_f32 %0, %1")
KDF(cvt_i32_round_1, "v_cvt_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1")
// mixed: one dependent FP64 chain with k independent integer instructions behind every link (do they hide in the shadow?)
KD(fma64_int2, 1, "v_fma_f64 %0, %0, %1, %2\n v_add_u32 v200, v200, v201\n v_add_u32 v202, v202, v201")

// LDS pointer chase: the address of read n+1 is the data of read n (b32), or its low word (b128: 16-byte slots)
template <int BYTES>
__global__ __launch_bounds__(256) void lds_chase(long long* out, double seed) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];  // 32 KB
    for (int n = threadIdx.x; n < 8192; n += 256) lds[n] = ((n * 2654435761u) >> 8) % 2040u * 16u;
    __syncthreads();
    unsigned addr = (threadIdx.x * 37u % 2040u) * 16u;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (BYTES == 16) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 v;
                asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
                addr = v.x;
            } else {
                asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(addr));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (addr == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(long long*, double);
struct Entry { const char* name; kern_t fn; int instr; };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    long long* d_out;
    CHECK(hipMalloc(&d_out, cus * 16 * sizeof(long long)));
    const int N = ITER * 32;
    std::vector<Entry> es = {
        {"v_fma_f64 dependent", fma64_1, N}, {"v_fma_f64 2 chains", fma64_2, N}, {"v_fma_f64 4 chains", fma64_3, N},
        {"v_mul_f64 dependent", mul64_1, N}, {"v_add_f64 dependent", add64_1, N}, {"v_max_f64 dependent", max64_1, N},
        {"v_rcp_f64 dependent", rcp64_1, N}, {"v_rcp_f64 2 chains", rcp64_2, N}, {"v_rsq_f64 dependent", rsq64_1, N},
        {"v_rndne_f64 dependent", rndne64_1, N}, {"v_ldexp_f64 dependent", ldexp64_1, N}, {"v_mov_b64 dependent", mov64_1, N},
        {"v_rcp_f64 + s_nop + v_fma_f64 (per pair)", rcp_fma_1, N},
        {"v_fma_f32 dependent", fma32_1, N}, {"v_fma_f32 2 chains", fma32_2, N}, {"v_mul_f32 dependent", mul32_1, N},
        {"v_add_u32 dependent", add_u32_1, N}, {"v_add_u32 2 chains", add_u32_2, N}, {"v_and_b32 dependent", and32_1, N},
        {"v_lshl_add_u32 dependent", lshl_add_1, N},
        {"v_log_f32 + s_nop dependent", log32_1, N}, {"v_exp_f32 + s_nop dependent", exp32_1, N}, {"v_rcp_f32 + s_nop dependent", rcp32_1, N},
        {"v_cmp vcc + s_nop 1 + v_cndmask (per triple)", cmp_cnd_vcc_1, N}, {"v_cmp sgpr + s_nop 1 + v_cndmask_e64 (per triple)", cmp_cnd_sgpr_1, N},
        {"v_cvt_f32_f64 + v_cvt_f64_f32 (per pair)", cvt_round_1, ITER * 16}, {"v_cvt_i32_f64 + v_cvt_f64_i32 (per pair)", cvt_i32_round_1, ITER * 16},
        {"v_fma_f64 dependent + 2 independent v_add_u32 (per triple)", fma64_int2, N},
        {"ds_read_b32 pointer chase", lds_chase<4>, ITER * 8}, {"ds_read_b128 pointer chase", lds_chase<16>, ITER * 8},
    };
    printf("%-64s %10s %10s   (cycles per instruction / per group; 1 and 3 waves per SIMD)\n", "chain", "1 wave", "3 waves");
    for (const Entry& e : es) {
        printf("%-64s", e.name);
        for (int k : {1, 3}) {
            const int blocks = cus * k, waves = blocks * 4;
            e.fn<<<blocks, 256>>>(d_out, 1.5);
            e.fn<<<blocks, 256>>>(d_out, 1.5);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> t(waves);
            CHECK(hipMemcpy(t.data(), d_out, waves * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(t.begin(), t.end());
            printf(" %10.2f", (double)t[waves / 2] / e.instr);
        }
        printf("\n");
    }
    return 0;
}
