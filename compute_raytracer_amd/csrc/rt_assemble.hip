// De-interleave of the all-gathered row tiles: rank r's compact buffer holds
// tiles r, r+world, r+2*world, ... ; the frame wants them in tile order.  Pure HBM copy:
// 16 B per lane when the row length allows, coalesced on both sides.
#include "rt_types.h"

template <typename T>
__global__ void assemble_frame(const T* __restrict__ gathered, T* __restrict__ frame, uint32_t row_elems,
                               uint32_t H, uint32_t world, uint32_t padded_tiles) {
    const uint32_t y = blockIdx.y;
    if (y >= H) return;
    const uint32_t tile = y >> 3, rank = tile % world, local = tile / world;
    const T* src = gathered + ((size_t)(rank * padded_tiles + local) * 8u + (y & 7u)) * row_elems;
    T* dst = frame + (size_t)y * row_elems;
    for (uint32_t x = blockIdx.x * blockDim.x + threadIdx.x; x < row_elems; x += gridDim.x * blockDim.x)
        dst[x] = src[x];
}

hipError_t rt_launch_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t W, uint32_t H,
                              uint32_t world, uint32_t padded_tiles, hipStream_t s) {
    if (W == 0 || H == 0) return hipSuccess;
    const bool wide = (W % 4u == 0) && ((reinterpret_cast<uintptr_t>(gathered) | reinterpret_cast<uintptr_t>(frame)) % 16u == 0);
    if (wide) {
        const uint32_t elems = W / 4u;
        dim3 grid((elems + 255u) / 256u, H);
        hipLaunchKernelGGL(assemble_frame<uint4>, grid, dim3(256), 0, s, reinterpret_cast<const uint4*>(gathered),
                           reinterpret_cast<uint4*>(frame), elems, H, world, padded_tiles);
    } else {
        dim3 grid((W + 255u) / 256u, H);
        hipLaunchKernelGGL(assemble_frame<uint32_t>, grid, dim3(256), 0, s,
                           reinterpret_cast<const uint32_t*>(gathered), reinterpret_cast<uint32_t*>(frame), W, H,
                           world, padded_tiles);
    }
    return hipGetLastError();
}
