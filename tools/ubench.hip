// ubench.hip -- gfx950 issue-rate microbenchmarks that decide the inner-loop design of the
// ray-sphere kernels (development tool; results quoted in DESIGN.md).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o gpurun_out/ubench && gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 65536;
__device__ unsigned long long g_cyc[256 * 8 * 4];
__device__ unsigned long long g_real[256 * 8 * 4];
#define T0() unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
#define T1() do { unsigned long long t1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); if ((threadIdx.x & 63) == 0) { g_cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1_ - t0_; g_real[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1_ - r0_; } } while (0)

// 16 independent v_fma_f32 per iteration
__global__ void k_fma(float* out, float a, float b) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
    T0();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
    }
    T1();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// same with one SGPR operand
__global__ void k_fma_sgpr(float* out, float a, float b) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
    T0();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "s"(a), "v"(b));
    }
    T1();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));

// 16 independent v_pk_fma_f32 per iteration (2 FMAs per lane each)
__global__ void k_pk_fma(float* out, float a, float b) {
    float2v r[16];
    float2v va = {a, a * 0.5f}, vb = {b, b * 0.5f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i].x = threadIdx.x * 0.001f + i; r[i].y = r[i].x + 1.0f; }
    T0();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(va), "v"(vb));
    }
    T1();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_pk_mul / v_pk_add mix
__global__ void k_pk_mul(float* out, float a, float b) {
    float2v r[16];
    float2v va = {a, a * 0.5f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i].x = threadIdx.x * 0.001f + i; r[i].y = r[i].x + 1.0f; }
    T0();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(va));
    }
    T1();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i].x + r[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_cmp + v_max3 + v_mul mix similar to the tail of a test
__global__ void k_max3(float* out, float a, float b) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 0.001f + i;
    T0();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
    }
    T1();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS broadcast reads: every lane reads the same 16 B; 8 reads in flight
__global__ void k_lds_bcast(float* out, int n) {
    __shared__ float4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sh[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    T0();
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float4 v = sh[(it * 16 + i) & 1023];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    T1();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// scalar loads of wave-uniform float4 (K$ path), 16 per iteration, 4 VALU per load
__global__ void k_sload(float* out, const float4* __restrict__ g, int n) {
    float4 acc = make_float4(0, 0, 0, 0);
    T0();
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float4 v = g[(it * 16 + i) & (n - 1)];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    T1();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 32 * 64 * 8 * sizeof(float)));
    float4* g; CHECK(hipMalloc(&g, 4096 * sizeof(float4)));
    CHECK(hipMemset(g, 0, 4096 * sizeof(float4)));
    const int CU = 256;
    printf("%-14s %6s %10s %14s %16s\n", "kernel", "w/SIMD", "ms", "instr/clk/SIMD", "cycles/instr");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = CU * wps;   // 256-thread blocks: 4 waves = one per SIMD
        const double waves_per_simd = wps;
        const double instr = (double)ITERS * 16;
        auto report = [&](const char* name, float ms, double per_wave_instr) {
            std::vector<unsigned long long> cyc(blocks * 4), real(blocks * 4);
            hipMemcpyFromSymbol(cyc.data(), HIP_SYMBOL(g_cyc), cyc.size() * 8);
            hipMemcpyFromSymbol(real.data(), HIP_SYMBOL(g_real), real.size() * 8);
            std::sort(cyc.begin(), cyc.end()); std::sort(real.begin(), real.end());
            double c = (double)cyc[cyc.size() / 2], r = (double)real[real.size() / 2];
            double ipc = per_wave_instr * waves_per_simd / c;
            printf("%-14s %6d %10.4f %14.4f %16.3f   clock %.3f GHz\n", name, wps, ms, ipc, 1.0 / ipc, c / (r * 10.0) );
        };
        report("v_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), instr);
        report("v_fma_f32+sgpr", time_ms([&] { hipLaunchKernelGGL(k_fma_sgpr, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), instr);
        report("v_pk_fma_f32", time_ms([&] { hipLaunchKernelGGL(k_pk_fma, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), instr);
        report("v_pk_mul_f32", time_ms([&] { hipLaunchKernelGGL(k_pk_mul, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), instr);
        report("v_max3_f32", time_ms([&] { hipLaunchKernelGGL(k_max3, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); }), instr);
        report("lds_b128+4add", time_ms([&] { hipLaunchKernelGGL(k_lds_bcast, dim3(blocks), dim3(256), 0, 0, out, 1024); }), (double)(ITERS / 4) * 16);
        report("s_load4+4add", time_ms([&] { hipLaunchKernelGGL(k_sload, dim3(blocks), dim3(256), 0, 0, out, g, 1024); }), (double)(ITERS / 4) * 16);
    }
    hipFree(out); hipFree(g);
    return 0;
}
