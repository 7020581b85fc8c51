// rt_types.h -- structures shared by the host layer (rt_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt355.h"

// Relative safety margin of the conservative discriminant filter (see rt_kernels.hip).
#define RT_FILTER_KAPPA 1.52587890625e-05f   // 2^-16
// Extra margin of the expanded form used for per-lane ray origins: eps * (|o|^2 + |c|^2).
#define RT_FILTER_EPS 7.62939453125e-06f     // 2^-17 = 128 * 2^-24
// Filter records hold positions scaled by 2^40 (exact: a power of two) so that the filter's
// discriminant is scaled by 2^80: whenever it is positive it is >= 1 (it is a difference of
// numbers whose granularity exceeds 1, see rt_kernels.hip), and the `clamp` output modifier of
// the v_fma_f32 that forms it turns "positive" into exactly 1.0 at no extra cost.
#define RT_FILTER_SCALE 1099511627776.0f                 // 2^40
#define RT_FILTER_SCALE2 1208925819614629174706176.0f    // 2^80
#define RT_FILTER_UNSCALE 9.094947017729282379150390625e-13f   // 2^-40
// Inner nodes of the sphere hierarchy: stored radius = bound of the members x this (host build rt_bvh_build.h, device
// refit rt_bvh.hip: bvh_refit) -- the slack the node test's proof needs (rt_bvh.hip, header).
#ifndef RT_BVH_SIGMA
#define RT_BVH_SIGMA 1.04
#endif

// The ray counter of a frame is RT_RAY_COUNTERS partial sums, RT_RAY_COUNTER_STRIDE bytes apart
// (rt_device.h: count_rays); the host adds them.
#define RT_RAY_COUNTERS 32u
#define RT_RAY_COUNTER_STRIDE 256u

// One frame's launch arguments (kernarg segment -> SGPRs; everything here is wave-uniform).
struct RtFrameArgs {
    float p[24];               // SceneParameters as RR:157-165 packs them
    uint32_t W, H, N;          // target size, sphere count
    uint32_t N16;              // N rounded up to a multiple of 16 (filter arrays are padded to it)
    uint32_t tile_first;       // first 8-row tile of this rank
    uint32_t tile_step;        // world size (tile stride)
    uint32_t n_local_tiles;    // tiles this launch renders
    uint32_t signed_filter;    // 1: the scene is small enough for the sign-aware filter
    // exact records, [N]: the values the reference arithmetic consumes
    const float4* geo;         // {cx, cy, cz, radius*radius}
    const float4* lgt;         // {L-c (xyz), dot(L-c,L-c) - r*r}   ray origin = light  (shadow rays)
    const float4* cam;         // same for ray origin = camera       (primary rays)
    const float4* col;         // {r, g, b, 0}
    // filter records, [N16]: scaled, inflated copies for the conservative discriminant test
    const float4* geo_f;       // {c * 2^40, (|c|^2 (1-eps) - r*r*(1+kappa)) * 2^80}   pad: w = +inf
    const float4* lgt_f;       // {(L-c) * 2^40, (|L-c|^2 - r*r*(1+kappa)) * 2^80}  pad: w = +inf
    const float4* cam_f;       // same for the camera
    // exact 4th components next to the filter records, [N16] (xyz is recovered exactly as *2^-40)
    const float* geo_w;        // r*r
    const float* lgt_w;        // |L-c|^2 - r*r
    const float* cam_w;
    const uint8_t* face[6];    // cube faces, rgba8unorm
    uint32_t fw[6], fh[6];
    uint32_t sky_flat;         // every face is ONE texel and the six texels agree (the constant sky of C1-C4)
    uint32_t sky_seamless;     // six equal square faces = a WebGPU cube texture: seamless filtering across edges
    uint8_t* out;              // compact tile buffer [n_local_tiles*8][W][4]
    // Textured sky, hierarchy path: bvh_pixels leaves every pixel's end-of-path record here and sky_resolve
    // samples the cube map and composes the pixel (rt_bvh.hip).  2 float4 per pixel slot of `out`:
    // {colour rgb, dist | sign bit = the path ended on a miss}, and for a miss after bounce 0
    // {direction of the missing ray, bounce count as bits}
    float4* fin;
    unsigned long long* rays;  // scene-traversal counter: RT_RAY_COUNTERS partial sums (one atomicAdd per wave)
    // path queue between the first-bounce kernel and the path kernel (two-kernel pipeline):
    // 3 float4 per surviving path {ro.xyz, pixel index}, {rd.xyz, dist}, {color.rgb, 0}
    float4* queue;             // [queue_cap][3]
    uint32_t* qctrl;           // [0] entries appended by the first-bounce kernel, [1] pop cursor,
                               // [2] tile-pair cursor of the first-bounce kernel
    uint32_t queue_cap;
    // bounding-sphere hierarchy of the sphere scene (rt_bvh.hip), depth-first with skip links
    const float4* bvh_rec;     // [bvh_nodes] node records, same layout as geo_f (leaves ARE geo_f records)
    const uint32_t* bvh_link;  // [bvh_nodes] inner node: 4 * (index after its subtree); leaf: 0x80000000 | sphere
    uint32_t bvh_nodes;        // 0: no hierarchy built
    uint32_t grid_share;       // >1: this frame's persistent grid takes 1/grid_share of the chip (frames in flight)
    uint32_t bvh_tail;         // lanes still walking below which a wave leaves the walk for the shading pass (0: never)
};

struct RtPrepArgs {
    float p[24];
    uint32_t N, N16;
    const float* records;      // [N][8] {cx,cy,cz,_, r,g,b, radius}
    float4 *geo, *lgt, *cam, *col;
    float4 *geo_f, *lgt_f, *cam_f;
    float *geo_w, *lgt_w, *cam_w;
};

struct RtLaunchCfg {
    int mode;      // rt_mode
    int variant;   // kernel variant id; 0 = default
};

// The kernel form the last rt_launch_* call on this thread chose (rt_kernel_id of include/rt355.h).
extern thread_local int g_rt_kernel_id;
extern thread_local int g_rt_tri_form;     // rt_triangles.hip: the stack form it launched (rt_stats.tri_form)

// Per-frame instance data of a triangle scene, carried in the kernarg block of apply_instances (rt_assemble.hip).
struct RtInstanceArgs {
    float* nodes;  float* blas;  float* lookup;          // destinations (one version of each buffer); may be null
    uint32_t n_head_f, n_blas_f, n_lookup_f;              // floats to write
    float data[31 * 8 + 16 * 20 + 16];                    // [head | blas | lookup]
};
hipError_t rt_launch_apply_instances(const RtInstanceArgs& a, hipStream_t s);
hipError_t rt_launch_frame_epilogue(unsigned long long* counters, unsigned long long* host, uint32_t words, hipStream_t s);
// The end of a frame on its stream, as 256 threads of ONE workgroup run it (rt_assemble.hip: frame_epilogue is this and nothing
// else; rt_triangles.hip: order_hist runs it in its first workgroup, so that an awaited triangle frame ends with one small kernel
// less in the chain the next frame waits for): the frame's partial ray counters summed, the sum and the frame's fault word
// (rt_device.h: report_fault) stored where the host reads them without a copy -- pinned, device-visible memory --, and the frame's
// counter set and control block (queue counts, pixel cursor) zeroed for the next frame that takes this slot.
#ifdef __HIPCC__
__device__ __forceinline__ void rt_frame_epilogue_body(unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ host, uint32_t words) {
    const uint32_t t = threadIdx.x;
    unsigned long long v = t < RT_RAY_COUNTERS ? ctr[t * (RT_RAY_COUNTER_STRIDE / 8u)] : 0ull;
    const unsigned long long fault = t == 0u ? ctr[1] : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);      // RT_RAY_COUNTERS <= 64: all in the first wave
    __syncthreads();                                                           // every read is done
    for (uint32_t i = t; i < words; i += 256u) ctr[i] = 0ull;
    if (t == 0u) {
        host[0] = v;
        host[1] = fault;
        __threadfence_system();
    }
}
#endif

hipError_t rt_launch_prep(const RtPrepArgs& a, hipStream_t s);
hipError_t rt_launch_trace(const RtFrameArgs& a, const RtLaunchCfg& cfg, hipStream_t s);
hipError_t rt_launch_bvh(const RtFrameArgs& a, hipStream_t s);
hipError_t rt_launch_sky_resolve(const RtFrameArgs& a, hipStream_t s);   // textured sky: composes the end-of-path records in a.fin (rt_bvh.hip)
hipError_t rt_launch_bvh_refit(float4* rec, const uint32_t* link, uint32_t n_nodes, const float* records, hipStream_t s);
hipError_t rt_launch_bvh_fill(float4* rec, const uint32_t* link, uint32_t n_nodes, const float4* geo_f, hipStream_t s);
hipError_t rt_launch_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t W, uint32_t H,
                              uint32_t world, uint32_t padded_tiles, hipStream_t s);
