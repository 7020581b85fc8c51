// ubench3.hip -- LDS broadcast-read rate on gfx950 next to v_fma_f32 work (development tool).
// Each wave reads a wave-uniform 16-byte record per "test" and spends NF v_fma_f32 on it.
// Tells at which FMA count per record the kernel turns from LDS-bound to VALU-bound.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench3.hip -o tools/bin/ubench3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int ITERS = 8192;   // x 16 records
__device__ unsigned long long g_cyc[256 * 8 * 4];
__device__ unsigned long long g_real[256 * 8 * 4];

template <int NF, int MODE>   // MODE 0: ds_read_b128 uniform, 1: 2x ds_read_b64 uniform, 2: 4x ds_read_b32
__global__ void k_lds(float* out) {
    __shared__ float4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sh[i] = make_float4(1e-3f * i, 1.0f, 0.5f, 0.25f);
    __syncthreads();
    float acc[4] = {0.1f * threadIdx.x, 0.2f, 0.3f, 0.4f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = (it * 16 + i) & 1023;
            if (MODE == 0) v[i] = sh[idx];
            else if (MODE == 1) {
                const float2* p = reinterpret_cast<const float2*>(&sh[idx]);
                float2 a = p[0], b = p[1];
                v[i] = make_float4(a.x, a.y, b.x, b.y);
            } else {
                const float* p = reinterpret_cast<const float*>(&sh[idx]);
                v[i] = make_float4(p[0], p[1], p[2], p[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float src[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int f = 0; f < NF; ++f)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[f & 3]) : "v"(src[f & 3]), "v"(src[(f + 1) & 3]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        g_cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
        g_real[blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int NF, int MODE>
void run(float* out, int wps) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_lds<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_lds<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(blocks * 4), real(blocks * 4);
    hipMemcpyFromSymbol(cyc.data(), HIP_SYMBOL(g_cyc), cyc.size() * 8);
    hipMemcpyFromSymbol(real.data(), HIP_SYMBOL(g_real), real.size() * 8);
    std::sort(cyc.begin(), cyc.end()); std::sort(real.begin(), real.end());
    double ghz = (double)cyc[cyc.size() / 2] / ((double)real[real.size() / 2] * 10.0);
    double rec_per_simd = (double)ITERS * 16 * wps;
    const char* names[] = {"ds_read_b128", "2x ds_read_b64", "4x ds_read_b32"};
    printf("%-15s fma/rec %2d  w/SIMD %d  %8.3f ms  %7.2f cyc/record/SIMD  (%.2f GHz)\n", names[MODE], NF, wps, ms,
           ms * 1e-3 * ghz * 1e9 / rec_per_simd, ghz);
}

int main() {
    float* out;
    if (hipMalloc(&out, 256 * 8 * 256 * sizeof(float)) != hipSuccess) return 1;
    for (int wps : {4, 8}) {
        run<1, 0>(out, wps); run<2, 0>(out, wps); run<4, 0>(out, wps); run<6, 0>(out, wps); run<8, 0>(out, wps); run<12, 0>(out, wps);
        run<1, 1>(out, wps); run<4, 1>(out, wps); run<8, 1>(out, wps);
        run<1, 2>(out, wps); run<4, 2>(out, wps);
    }
    hipFree(out);
    return 0;
}
