// ubench2.hip -- per-instruction VALU issue rates on gfx950 (development tool; the table
// it prints is quoted in DESIGN.md).  Every kernel runs ITERS x 16 independent copies of one
// instruction per wave; rate = wall time x in-kernel clock / (instructions per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench2.hip -o tools/bin/ubench2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ITERS = 32768;
__device__ unsigned long long g_cyc[256 * 8 * 4];
__device__ unsigned long long g_real[256 * 8 * 4];

#define STAMP0() unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define STAMP1()                                                                                       \
    do {                                                                                               \
        unsigned long long t1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
        if ((threadIdx.x & 63) == 0) {                                                                 \
            g_cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1_ - t0_;                                    \
            g_real[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1_ - r0_;                                   \
        }                                                                                              \
    } while (0)

// r[i] is read-modify-written, x[i]/y[i] are distinct per-copy VGPR sources, sa/sb SGPRs
#define KERNEL(NAME, ASM)                                                                  \
    __global__ void NAME(float* out, float sa, float sb) {                                 \
        float r[16], x[16], y[16];                                                         \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                   \
            r[i] = threadIdx.x * 0.001f + i;                                               \
            x[i] = 1.0f + i * 1e-6f + threadIdx.x * 1e-7f;                                 \
            y[i] = 1e-3f * i;                                                              \
        }                                                                                  \
        STAMP0();                                                                          \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                                 \
                asm volatile(ASM : "+v"(r[i]) : "v"(x[i]), "v"(y[i]), "s"(sa), "s"(sb) : "vcc", "s20", "s21"); \
        }                                                                                  \
        STAMP1();                                                                          \
        float s = 0;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += r[i] + x[i] + y[i];            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                    \
    }

KERNEL(k_add, "v_add_f32 %0, %1, %0")
KERNEL(k_mul, "v_mul_f32 %0, %1, %0")
KERNEL(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_fma_neg, "v_fma_f32 %0, %1, %2, -%0")
KERNEL(k_sub_s, "v_sub_f32 %0, %3, %0")
KERNEL(k_mul_s, "v_mul_f32 %0, %3, %0")
KERNEL(k_fma_s, "v_fma_f32 %0, %1, %3, %0")
KERNEL(k_fma_inl, "v_fma_f32 %0, %1, 2.0, %0")
KERNEL(k_add_e64, "v_add_f32_e64 %0, %1, %0")
KERNEL(k_mul_e64, "v_mul_f32_e64 %0, %1, %0")
KERNEL(k_sub_vv, "v_sub_f32 %0, %1, %0")
KERNEL(k_fma_one, "v_fma_f32 %0, %1, 1.0, %0")
KERNEL(k_fma_mone, "v_fma_f32 %0, %0, -1.0, %1")
KERNEL(k_fma_mul, "v_fma_f32 %0, %1, %0, -%2")
KERNEL(k_max_e64, "v_max_f32_e64 %0, %1, %0")
KERNEL(k_fmac_e64, "v_fmac_f32_e64 %0, %1, %2")
KERNEL(k_mov_e64, "v_mov_b32_e64 %0, %1")
KERNEL(k_cnd_e64, "v_cndmask_b32_e64 %0, %1, %0, vcc")
KERNEL(k_cmp_e64, "v_cmp_lt_f32_e64 vcc, %1, %0")
KERNEL(k_addu_e32, "v_add_u32_e32 %0, %1, %0")
KERNEL(k_addu_e64, "v_add_u32_e64 %0, %1, %0")
KERNEL(k_lshl_e32, "v_lshlrev_b32_e32 %0, 1, %0")
KERNEL(k_lshl_e64, "v_lshlrev_b32_e64 %0, 1, %0")
KERNEL(k_and_e32, "v_and_b32_e32 %0, %1, %0")
KERNEL(k_and_e64, "v_and_b32_e64 %0, %1, %0")
KERNEL(k_rcp_e64, "v_rcp_f32_e64 %0, %0")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %1, 2, %0")
KERNEL(k_max, "v_max_f32 %0, %1, %0")
KERNEL(k_max3, "v_max3_f32 %0, %1, %2, %0")
KERNEL(k_cmp_vcc, "v_cmp_lt_f32 vcc, %1, %0")
KERNEL(k_cmp_s, "v_cmp_lt_f32 s[20:21], %1, %0")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL(k_mov_s, "v_mov_b32 %0, %3")
KERNEL(k_mov_v, "v_mov_b32 %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %0")
KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3")

typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL_PK(NAME, ASM)                                                               \
    __global__ void NAME(float* out, float sa, float sb) {                                 \
        f2 r[16], x[16], y[16];                                                            \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) {                                   \
            r[i].x = threadIdx.x * 0.001f + i; r[i].y = r[i].x + 0.5f;                     \
            x[i].x = 1.0f + i * 1e-6f + threadIdx.x * 1e-7f; x[i].y = x[i].x + 1e-6f;      \
            y[i].x = 1e-3f * i; y[i].y = 2e-3f * i;                                        \
        }                                                                                  \
        STAMP0();                                                                          \
        for (int it = 0; it < ITERS; ++it) {                                               \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                                 \
                asm volatile(ASM : "+v"(r[i]) : "v"(x[i]), "v"(y[i]));                     \
        }                                                                                  \
        STAMP1();                                                                          \
        float s = 0;                                                                       \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y + x[i].x + y[i].y; \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                    \
    }
KERNEL_PK(k_pk_fma, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_PK(k_pk_add, "v_pk_add_f32 %0, %1, %0")
KERNEL_PK(k_pk_mul, "v_pk_mul_f32 %0, %1, %0")

struct Entry { const char* name; void (*fn)(float*, float, float); };

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
    Entry entries[] = {
        {"v_add_f32 v,v", k_add}, {"v_mul_f32 v,v", k_mul}, {"v_fmac_f32 v,v", k_fmac}, {"v_fma_f32 v,v,v", k_fma},
        {"v_fma_f32 v,v,-v", k_fma_neg}, {"v_sub_f32 s,v", k_sub_s}, {"v_mul_f32 s,v", k_mul_s},
        {"v_fma_f32 v,s,v", k_fma_s}, {"v_fma_f32 v,2.0,v", k_fma_inl}, {"v_add_f32_e64", k_add_e64}, {"v_mul_f32_e64", k_mul_e64}, {"v_sub_f32 v,v", k_sub_vv},
        {"v_max_f32_e64", k_max_e64}, {"v_fmac_f32_e64", k_fmac_e64}, {"v_mov_b32_e64 v", k_mov_e64}, {"v_cndmask_e64 vcc", k_cnd_e64},
        {"v_cmp_lt_e64 vcc", k_cmp_e64}, {"v_add_u32_e32", k_addu_e32}, {"v_add_u32_e64", k_addu_e64}, {"v_lshlrev_e32", k_lshl_e32},
        {"v_lshlrev_e64", k_lshl_e64}, {"v_and_b32_e32", k_and_e32}, {"v_and_b32_e64", k_and_e64}, {"v_rcp_f32_e64", k_rcp_e64},
        {"v_lshl_add_u32", k_lshladd},
        {"fma v,1.0,v (=add)", k_fma_one}, {"fma v,-1.0,v (=sub)", k_fma_mone}, {"fma v,v,-v0 (=mul)", k_fma_mul}, {"v_max_f32", k_max}, {"v_max3_f32", k_max3},
        {"v_cmp_lt vcc", k_cmp_vcc}, {"v_cmp_lt sgpr", k_cmp_s}, {"v_cndmask vcc", k_cndmask}, {"v_mov_b32 s", k_mov_s},
        {"v_mov_b32 v", k_mov_v}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_add_f32_dpp", k_add_dpp},
        {"v_readlane_b32", k_readlane}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul},
    };
    printf("%-20s %7s %9s %12s %9s\n", "instruction", "w/SIMD", "ms", "cyc/instr", "GHz");
    for (int wps : {4, 8}) {
        const int blocks = 256 * wps;
        for (auto& e : entries) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> cyc(blocks * 4), real(blocks * 4);
            hipMemcpyFromSymbol(cyc.data(), HIP_SYMBOL(g_cyc), cyc.size() * 8);
            hipMemcpyFromSymbol(real.data(), HIP_SYMBOL(g_real), real.size() * 8);
            std::sort(cyc.begin(), cyc.end()); std::sort(real.begin(), real.end());
            double ghz = (double)cyc[cyc.size() / 2] / ((double)real[real.size() / 2] * 10.0);
            double instr_per_simd = (double)ITERS * 16 * wps;
            printf("%-20s %7d %9.3f %12.3f %9.3f\n", e.name, wps, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    hipFree(out);
    return 0;
}
