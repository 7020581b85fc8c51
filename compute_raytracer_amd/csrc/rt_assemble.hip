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

// Per-frame instance data of a triangle scene (RR:169-192: BLAS records, BLAS lookup, TLAS nodes): the values travel
// in this kernel's kernarg block -- copied by the runtime when the launch is enqueued, so the host may rewrite its own
// copy at once -- and are stored into the buffer versions the frame behind it on the same stream reads.  One
// workgroup; the three destinations are a few hundred floats.
__global__ __launch_bounds__(256) void apply_instances(const RtInstanceArgs a) {
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < a.n_head_f; i += 256u) a.nodes[i] = a.data[i];
    for (uint32_t i = t; i < a.n_blas_f; i += 256u) a.blas[i] = a.data[31u * 8u + i];
    for (uint32_t i = t; i < a.n_lookup_f; i += 256u) a.lookup[i] = a.data[31u * 8u + 16u * 20u + i];
}

hipError_t rt_launch_apply_instances(const RtInstanceArgs& a, hipStream_t s) {
    if (a.n_head_f + a.n_blas_f + a.n_lookup_f == 0u) return hipSuccess;
    hipLaunchKernelGGL(apply_instances, dim3(1), dim3(256), 0, s, a);
    return hipGetLastError();
}

// The end of a frame on its stream (rt_types.h: rt_frame_epilogue_body).  rt_wait then needs ONE synchronisation, on the event
// behind this kernel: no read-back and memset of its own, whose launch and second wait cost a caller that waits after every
// frame ~0.1 ms per frame (round 3).
__global__ __launch_bounds__(256) void frame_epilogue(unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ host, uint32_t words) {
    rt_frame_epilogue_body(ctr, host, words);
}

hipError_t rt_launch_frame_epilogue(unsigned long long* counters, unsigned long long* host, uint32_t words, hipStream_t s) {
    hipLaunchKernelGGL(frame_epilogue, dim3(1), dim3(256), 0, s, counters, host, words);
    return hipGetLastError();
}
