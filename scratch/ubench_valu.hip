// ubench_valu.hip — issue cost of the gfx950 instructions the flux solver is made of (scratch; not product code).
// One op per kernel instantiation, 8 independent register chains (or 1 dependent chain), timed per wave with
// s_memtime, at 1–4 waves per SIMD.  Prints cycles per wave-instruction per SIMD (= wave time ÷ instructions ÷ waves/SIMD)
// build: hipcc -O3 --offload-arch=gfx950 scratch/ubench_valu.hip -o scratch/ubench_valu
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int ITER = 2000;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// 64-bit-register ops: "op %0, %0[, …]" on eight chains
#define KERNEL_D(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed, const double* tab) {        \
        double r[8];                                                                                     \
        for (int n = 0; n < 8; ++n) r[n] = seed + 1e-3 * (threadIdx.x + n);                              \
        double a = 1.0000001, b = 1e-9;                                                                  \
        asm volatile("" : "+v"(a), "+v"(b));                                                             \
        const long long t0 = __builtin_readcyclecounter();                                              \
        for (int it = 0; it < ITER; ++it) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
                _Pragma("unroll") for (int n = 0; n < 8; ++n) asm volatile(ASM : "+v"(r[n]) : "v"(a), "v"(b)); \
            }                                                                                            \
        }                                                                                                \
        const long long t1 = __builtin_readcyclecounter();                                              \
        double s = 0;                                                                                    \
        for (int n = 0; n < 8; ++n) s += r[n];                                                           \
        if (s == 12345.678) out[0] = 1;                                                                  \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;              \
    }

// one dependent chain
#define KERNEL_D1(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed, const double* tab) {        \
        double r = seed + 1e-3 * threadIdx.x;                                                            \
        double a = 1.0000001, b = 1e-9;                                                                  \
        asm volatile("" : "+v"(a), "+v"(b));                                                             \
        const long long t0 = __builtin_readcyclecounter();                                              \
        for (int it = 0; it < ITER; ++it) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 32; ++u) asm volatile(ASM : "+v"(r) : "v"(a), "v"(b)); \
        }                                                                                                \
        const long long t1 = __builtin_readcyclecounter();                                              \
        if (r == 12345.678) out[0] = 1;                                                                  \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;              \
    }

// 32-bit-register ops
#define KERNEL_F(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed, const double* tab) {        \
        float r[8];                                                                                      \
        for (int n = 0; n < 8; ++n) r[n] = (float)seed + 1e-3f * (threadIdx.x + n);                      \
        float a = 1.0000001f, b = 1e-9f;                                                                 \
        asm volatile("" : "+v"(a), "+v"(b));                                                             \
        const long long t0 = __builtin_readcyclecounter();                                              \
        for (int it = 0; it < ITER; ++it) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
                _Pragma("unroll") for (int n = 0; n < 8; ++n) asm volatile(ASM : "+v"(r[n]) : "v"(a), "v"(b)); \
            }                                                                                            \
        }                                                                                                \
        const long long t1 = __builtin_readcyclecounter();                                              \
        float s = 0;                                                                                     \
        for (int n = 0; n < 8; ++n) s += r[n];                                                           \
        if (s == 12345.678f) out[0] = 1;                                                                 \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;              \
    }

// conversions: 64-bit source chain -> 32-bit destination (and back), so that both stay alive
#define KERNEL_DF(NAME, ASM)                                                                             \
    __global__ __launch_bounds__(256) void NAME(long long* out, double seed, const double* tab) {        \
        double r[8];                                                                                     \
        float f[8];                                                                                      \
        for (int n = 0; n < 8; ++n) { r[n] = seed + 1e-3 * (threadIdx.x + n); f[n] = (float)r[n]; }      \
        const long long t0 = __builtin_readcyclecounter();                                              \
        for (int it = 0; it < ITER; ++it) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
                _Pragma("unroll") for (int n = 0; n < 8; ++n) asm volatile(ASM : "+v"(r[n]), "+v"(f[n])); \
            }                                                                                            \
        }                                                                                                \
        const long long t1 = __builtin_readcyclecounter();                                              \
        double s = 0;                                                                                    \
        for (int n = 0; n < 8; ++n) s += r[n] + f[n];                                                    \
        if (s == 12345.678) out[0] = 1;                                                                  \
        if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;              \
    }

// LDS reads: MODE 0 = every lane the same address, 1 = random 16-byte slots (different per lane, fixed per kernel),
// 2 = random among 32 slots only (a few distinct segments per wave)
template <int BYTES, int MODE>
__global__ __launch_bounds__(256) void lds_read_kernel(long long* out, double seed, const double* tab) {
    __shared__ __attribute__((aligned(16))) double lds[4096];  // 32 KB
    for (int n = threadIdx.x; n < 4096; n += 256) lds[n] = seed + n;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    h ^= h >> 15;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    unsigned slot = MODE == 0 ? 7u : (MODE == 1 ? (h % 2040u) : (h % 32u) * 61u);
    unsigned addr = slot * 16u;
    asm volatile("" : "+v"(addr));
    double acc0 = 0, acc1 = 0;
    unsigned iacc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER / 4; ++it) {
        if constexpr (BYTES == 16) {
            u4 v[8];
#pragma unroll
            for (int n = 0; n < 8; ++n)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[n]) : "v"(addr), "n"(n * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                asm volatile("" : "+v"(v[n]));
                iacc ^= v[n].x ^ v[n].w;
            }
        } else {
            double v[8];
#pragma unroll
            for (int n = 0; n < 8; ++n)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[n]) : "v"(addr), "n"(n * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                asm volatile("" : "+v"(v[n]));
                acc0 += v[n];
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (acc0 + acc1 == 12345.678 || iacc == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

// LDS reads beside FP64 FMAs: 8 b128 reads + NF fmas per round — does the LDS pipe run beside the VALU?
template <int NF>
__global__ __launch_bounds__(256) void lds_fma_kernel(long long* out, double seed, const double* tab) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int n = threadIdx.x; n < 4096; n += 256) lds[n] = seed + n;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    h ^= h >> 15;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    unsigned addr = (h % 2040u) * 16u;
    asm volatile("" : "+v"(addr));
    double r[8];
    for (int n = 0; n < 8; ++n) r[n] = seed + 1e-3 * (threadIdx.x + n);
    double a = 1.0000001, b = 1e-9;
    asm volatile("" : "+v"(a), "+v"(b));
    double acc0 = 0;
    unsigned iacc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER / 4; ++it) {
        u4 v[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[n]) : "v"(addr), "n"(n * 16));
#pragma unroll
        for (int u = 0; u < NF / 8; ++u)
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[n]) : "v"(a), "v"(b));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            asm volatile("" : "+v"(v[n]));
            iacc ^= v[n].x ^ v[n].w;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    for (int n = 0; n < 8; ++n) acc0 += r[n];
    if (acc0 == 12345.678 || iacc == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D1(k_fma_f64_dep, "v_fma_f64 %0, %0, %1, %2")
KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL_D(k_max_f64, "v_max_f64 %0, %0, %2")
KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_D(k_rndne_f64, "v_rndne_f64 %0, %0")
KERNEL_D(k_fract_f64, "v_fract_f64 %0, %0")
KERNEL_D(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
KERNEL_D(k_ldexp_f64, "v_ldexp_f64 %0, %0, 1")
KERNEL_D(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_D(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
KERNEL_D(k_lshl_b64, "v_lshlrev_b64 %0, 1, %0")
KERNEL_F(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F(k_rsq_f32, "v_rsq_f32 %0, %0")
KERNEL_F(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F(k_log_f32, "v_log_f32 %0, %0")
KERNEL_F(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL_F(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL_F(k_lshr_b32, "v_lshrrev_b32 %0, 1, %0")
KERNEL_F(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 9")
KERNEL_F(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_F(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_F(k_mov_b32, "v_mov_b32 %0, %1")
KERNEL_F(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
KERNEL_D(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_D(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_DF(k_cvt_f32_f64, "v_cvt_f32_f64 %1, %0")
KERNEL_DF(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL_DF(k_cvt_f64_i32, "v_cvt_f64_i32 %0, %1")
KERNEL_DF(k_cvt_i32_f64, "v_cvt_i32_f64 %1, %0")
KERNEL_DF(k_frexp_exp_f64, "v_frexp_exp_i32_f64 %1, %0")


KERNEL_F(k_cndmask_e64_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL_F(k_cmp_cndmask_f32, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_F(k_cmp_f32, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL_F(k_cmp_e64_cndmask_f32, "v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL_F(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL_F(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
KERNEL_F(k_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
KERNEL_F(k_mul_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL_F(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL_F(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL_F(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL_F(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_F(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL_D(k_fmac_f64, "v_fmac_f64 %0, %1, %2")
KERNEL_D(k_fma_f64_sgpr, "v_fma_f64 %0, %0, s[20:21], %2")
KERNEL_D(k_fma_f64_neg, "v_fma_f64 %0, -%0, %1, |%2|")
KERNEL_D(k_mul_f64_inl, "v_mul_f64 %0, %0, 0.5")

// conflict-free LDS reads: lane l of every 16-lane group reads slot (16*n + l%16)
template <int BYTES>
__global__ __launch_bounds__(256) void lds_cf_kernel(long long* out, double seed, const double* tab) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int n = threadIdx.x; n < 4096; n += 256) lds[n] = seed + n;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    h ^= h >> 15;
    unsigned addr = ((h % 100u) * 16u + (threadIdx.x & 15u)) * 16u;
    asm volatile("" : "+v"(addr));
    unsigned iacc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER / 4; ++it) {
        if constexpr (BYTES == 16) {
            u4 v[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[n]) : "v"(addr), "n"(n * 256));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                asm volatile("" : "+v"(v[n]));
                iacc ^= v[n].x ^ v[n].w;
            }
        } else {
            u2 v[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[n]) : "v"(addr), "n"(n * 256));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                asm volatile("" : "+v"(v[n]));
                iacc ^= v[n].x ^ v[n].y;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (iacc == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

// Horner chains fed from LDS: two chains (m, h) of DEG steps, coefficients read as b128 pairs at random slots
template <int DEG>
__global__ __launch_bounds__(256) void horner_lds_kernel(long long* out, double seed, const double* tab) {
    __shared__ __attribute__((aligned(16))) double lds[4096];
    for (int n = threadIdx.x; n < 4096; n += 256) lds[n] = 1e-3 * n;
    __syncthreads();
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    h ^= h >> 15;
    unsigned addr = (h % 40u) * 16u;
    asm volatile("" : "+v"(addr));
    double t = 0.01 + 1e-4 * threadIdx.x, acc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER / 4; ++it) {
        const double2* c = reinterpret_cast<const double2*>(reinterpret_cast<const char*>(lds) + addr);
        double2 v = c[DEG * 40];
        double pm = v.x, ph = v.y;
#pragma unroll
        for (int j = DEG - 1; j >= 0; --j) {
            v = c[j * 40];
            pm = __builtin_fma(pm, t, v.x);
            ph = __builtin_fma(ph, t, v.y);
        }
        acc += pm + ph;
        asm volatile("" : "+v"(addr), "+v"(t));
    }
    const long long t1 = __builtin_readcyclecounter();
    if (acc == 12345.678) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 256 + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(long long*, double, const double*);
struct Entry {
    const char* name;
    kern_t fn;
    int instr;  // wave-instructions per wave in the timed region
};

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    long long* d_out;
    const int max_waves = cus * 4 * 8;
    CHECK(hipMalloc(&d_out, max_waves * sizeof(long long)));
    const int N32 = ITER * 32;
    std::vector<Entry> es = {
        {"v_fma_f64", k_fma_f64, N32},
        {"v_fma_f64 (1 dependent chain)", k_fma_f64_dep, N32},
        {"v_mul_f64", k_mul_f64, N32},
        {"v_add_f64", k_add_f64, N32},
        {"v_max_f64", k_max_f64, N32},
        {"v_rcp_f64", k_rcp_f64, N32},
        {"v_rsq_f64", k_rsq_f64, N32},
        {"v_sqrt_f64", k_sqrt_f64, N32},
        {"v_rndne_f64", k_rndne_f64, N32},
        {"v_fract_f64", k_fract_f64, N32},
        {"v_frexp_mant_f64", k_frexp_mant_f64, N32},
        {"v_frexp_exp_i32_f64", k_frexp_exp_f64, N32},
        {"v_ldexp_f64", k_ldexp_f64, N32},
        {"v_mov_b64", k_mov_b64, N32},
        {"v_cmp_lt_f64", k_cmp_f64, N32},
        {"v_lshlrev_b64", k_lshl_b64, N32},
        {"v_cvt_f32_f64", k_cvt_f32_f64, N32},
        {"v_cvt_f64_f32", k_cvt_f64_f32, N32},
        {"v_cvt_f64_i32", k_cvt_f64_i32, N32},
        {"v_cvt_i32_f64", k_cvt_i32_f64, N32},
        {"v_fma_f32", k_fma_f32, N32},
        {"v_pk_fma_f32", k_pk_fma_f32, N32},
        {"v_pk_mul_f32", k_pk_mul_f32, N32},
        {"v_rcp_f32", k_rcp_f32, N32},
        {"v_rsq_f32", k_rsq_f32, N32},
        {"v_sqrt_f32", k_sqrt_f32, N32},
        {"v_log_f32", k_log_f32, N32},
        {"v_exp_f32", k_exp_f32, N32},
        {"v_cvt_f32_i32", k_cvt_f32_i32, N32},
        {"v_and_b32", k_and_b32, N32},
        {"v_lshrrev_b32", k_lshr_b32, N32},
        {"v_bfe_u32", k_bfe_u32, N32},
        {"v_and_or_b32", k_and_or_b32, N32},
        {"v_add_u32", k_add_u32, N32},
        {"v_mov_b32", k_mov_b32, N32},
        {"v_cndmask_b32", k_cndmask_b32, N32},
        {"v_cndmask_b32_e64 (sgpr pair mask)", k_cndmask_e64_s, N32},
        {"v_cmp_lt_f32 vcc", k_cmp_f32, N32},
        {"v_cmp_lt_f32 vcc + v_cndmask vcc (per 2)", k_cmp_cndmask_f32, 2 * N32},
        {"v_cmp_e64 sgpr + v_cndmask_e64 (per 2)", k_cmp_e64_cndmask_f32, 2 * N32},
        {"v_max_f32", k_max_f32, N32},
        {"v_med3_f32", k_med3_f32, N32},
        {"v_bfi_b32", k_bfi_b32, N32},
        {"v_mul_lo_u32", k_mul_u32, N32},
        {"v_mov_b32_dpp row_shr", k_mov_dpp, N32},
        {"v_readlane_b32", k_readlane, N32},
        {"v_add_f32", k_add_f32, N32},
        {"v_mul_f32", k_mul_f32, N32},
        {"v_fmac_f32", k_fmac_f32, N32},
        {"v_fmac_f64", k_fmac_f64, N32},
        {"v_fma_f64 sgpr operand", k_fma_f64_sgpr, N32},
        {"v_fma_f64 neg/abs modifiers", k_fma_f64_neg, N32},
        {"v_mul_f64 inline const", k_mul_f64_inl, N32},
        {"ds_read_b128 conflict-free", lds_cf_kernel<16>, ITER / 4 * 8},
        {"ds_read_b64 conflict-free", lds_cf_kernel<8>, ITER / 4 * 8},
        {"horner deg7 x2 chains, 8 b128 LDS (per round)", horner_lds_kernel<7>, ITER / 4},
        {"horner deg9 x2 chains, 10 b128 LDS (per round)", horner_lds_kernel<9>, ITER / 4},
        {"horner deg5 x2 chains, 6 b128 LDS (per round)", horner_lds_kernel<5>, ITER / 4},
        {"ds_read_b128 same address", lds_read_kernel<16, 0>, ITER / 4 * 8},
        {"ds_read_b128 random slots", lds_read_kernel<16, 1>, ITER / 4 * 8},
        {"ds_read_b128 32 distinct slots", lds_read_kernel<16, 2>, ITER / 4 * 8},
        {"ds_read_b64 same address", lds_read_kernel<8, 0>, ITER / 4 * 8},
        {"ds_read_b64 random slots", lds_read_kernel<8, 1>, ITER / 4 * 8},
        {"ds_read_b64 32 distinct slots", lds_read_kernel<8, 2>, ITER / 4 * 8},
        {"8 ds_read_b128 random + 16 v_fma_f64 (per 24)", lds_fma_kernel<16>, ITER / 4 * 24},
        {"8 ds_read_b128 random + 32 v_fma_f64 (per 40)", lds_fma_kernel<32>, ITER / 4 * 40},
        {"8 ds_read_b128 random + 64 v_fma_f64 (per 72)", lds_fma_kernel<64>, ITER / 4 * 72},
    };
    printf("%-48s %10s %10s %10s %10s   (SIMD cycles per wave-instruction at 1/2/3/4 waves per SIMD)\n", "op", "1", "2", "3", "4");
    for (const Entry& e : es) {
        printf("%-48s", e.name);
        for (int k = 1; k <= 4; ++k) {
            const int blocks = cus * k, waves = blocks * 4;
            e.fn<<<blocks, 256>>>(d_out, 1.5, nullptr);  // warm-up
            e.fn<<<blocks, 256>>>(d_out, 1.5, nullptr);
            CHECK(hipDeviceSynchronize());
            std::vector<long long> t(waves);
            CHECK(hipMemcpy(t.data(), d_out, waves * sizeof(long long), hipMemcpyDeviceToHost));
            std::sort(t.begin(), t.end());
            const double med = (double)t[waves / 2];
            printf(" %10.2f", med / e.instr / k);
        }
        printf("\n");
    }
    // wall-clock check of the s_memtime unit: fma kernel at 4 waves/SIMD, events
    {
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        k_fma_f64<<<cus * 4, 256>>>(d_out, 1.5, nullptr);
        CHECK(hipEventRecord(a));
        for (int n = 0; n < 10; ++n) k_fma_f64<<<cus * 4, 256>>>(d_out, 1.5, nullptr);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        std::vector<long long> t(cus * 16);
        CHECK(hipMemcpy(t.data(), d_out, cus * 16 * sizeof(long long), hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end());
        printf("v_fma_f64 x %d per wave, 4 waves/SIMD: %.3f us per launch by events; median wave %lld ticks => %.1f ticks/us\n", N32,
               ms * 100.0, t[t.size() / 2], (double)t[t.size() / 2] / (ms * 100.0));
        printf("  => %.3f G wave-instr/s per SIMD-chip (1024 SIMDs): %.1f\n", 0.0, (double)N32 * cus * 16 / (ms * 1e-3 / 10) / 1e9);
    }
    return 0;
}
