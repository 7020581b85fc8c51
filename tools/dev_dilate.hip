// Development experiment, NOT part of the library (tools/build_dev.sh links it into tools/bin/librt355_dev.so only): does a work list
// made from DILATED tile costs -- a tile counts as long as the longest of itself and its four neighbours (x RT355_TRI_DILATE / 8) --
// tolerate motion?  (docs/next.md 1: with the reference's mesh turning an awaited frame takes 0.411 ms of kernel with the previous
// pose's list and 0.360 with its own.)  The library's rt_launch_order_hist is renamed ..._orig in the dev build's object and this
// one stands in front of it: costs dilated in place (through a scratch copy), then the library's two kernels as they are.
// RT355_TRI_GX = tiles per row of the frame (the launcher is not told; 168 for the reference's 1344 x 846).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

hipError_t rt_launch_order_hist_orig(uint32_t* cost, uint32_t* scan, uint32_t* order, uint32_t n_tiles, uint32_t wave_slots,
                                     unsigned long long* counters, unsigned long long* host, uint32_t words, unsigned long long* split_out,
                                     hipStream_t s);

namespace {
__global__ void dilate(const uint32_t* __restrict__ cost, uint32_t* __restrict__ tmp, uint32_t n, uint32_t gx, uint32_t w8) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = i % gx;
    uint32_t m = 0u;
    if (x > 0u) m = max(m, cost[i - 1u]);
    if (x + 1u < gx && i + 1u < n) m = max(m, cost[i + 1u]);
    if (i >= gx) m = max(m, cost[i - gx]);
    if (i + gx < n) m = max(m, cost[i + gx]);
    const unsigned long long scaled = ((unsigned long long)m * w8) >> 3;
    tmp[i] = max(cost[i], scaled > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)scaled);
}
__global__ void copy_back(uint32_t* __restrict__ cost, const uint32_t* __restrict__ tmp, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) cost[i] = tmp[i];
}
}  // namespace

hipError_t rt_launch_order_hist(uint32_t* cost, uint32_t* scan, uint32_t* order, uint32_t n_tiles, uint32_t wave_slots,
                                unsigned long long* counters, unsigned long long* host, uint32_t words, unsigned long long* split_out,
                                hipStream_t s) {
    static uint32_t* tmp = nullptr;
    static uint32_t cap = 0u;
    const char* e = getenv("RT355_TRI_DILATE");
    const char* g = getenv("RT355_TRI_GX");
    const uint32_t w8 = e ? (uint32_t)atoi(e) : 0u, gx = g ? (uint32_t)atoi(g) : 0u;
    if (w8 != 0u && gx != 0u && n_tiles != 0u && n_tiles % gx == 0u) {
        if (cap < n_tiles) {
            (void)hipFree(tmp);
            tmp = nullptr; cap = 0u;
            if (hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)n_tiles * 4u) != hipSuccess) return hipErrorOutOfMemory;
            cap = n_tiles;
        }
        const uint32_t blocks = (n_tiles + 255u) / 256u;
        hipLaunchKernelGGL(dilate, dim3(blocks), dim3(256), 0, s, cost, tmp, n_tiles, gx, w8);
        hipLaunchKernelGGL(copy_back, dim3(blocks), dim3(256), 0, s, cost, tmp, n_tiles);
    }
    return rt_launch_order_hist_orig(cost, scan, order, n_tiles, wave_slots, counters, host, words, split_out, s);
}
