// rt_types.h -- structures shared by the host layer (rt_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One frame's launch arguments (kernarg segment -> SGPRs; everything here is wave-uniform).
struct RtFrameArgs {
    float p[24];               // SceneParameters as RR:157-165 packs them
    uint32_t W, H, N;          // target size, sphere count
    uint32_t tile_first;       // first 8-row tile of this rank
    uint32_t tile_step;        // world size (tile stride)
    uint32_t n_local_tiles;    // tiles this launch renders
    const float4* geo;         // [N] {cx, cy, cz, radius*radius}
    const float4* lgt;         // [N] {L-c (xyz), dot(L-c,L-c) - r*r}   origin = light (shadow rays)
    const float4* cam;         // [N] same for origin = camera           (primary rays)
    const float4* col;         // [N] {r, g, b, 0}
    const uint8_t* face[6];    // cube faces, rgba8unorm
    uint32_t fw[6], fh[6];
    uint8_t* out;              // compact tile buffer [n_local_tiles*8][W][4]
    unsigned long long* rays;  // scene-traversal counter (atomicAdd once per wave)
};

struct RtPrepArgs {
    float p[24];
    uint32_t N;
    const float* records;      // [N][8] {cx,cy,cz,_, r,g,b, radius}
    float4* geo;
    float4* lgt;
    float4* cam;
    float4* col;
};

// Launch entry points, one pair per arithmetic mode (separate translation units compiled
// with -ffp-contract=off / =fast).
struct RtLaunchCfg {
    int variant;   // kernel variant id; 0 = default
};

hipError_t rt_launch_prep_strict(const RtPrepArgs& a, hipStream_t s);
hipError_t rt_launch_trace_strict(const RtFrameArgs& a, const RtLaunchCfg& cfg, hipStream_t s);
hipError_t rt_launch_trace_fast(const RtFrameArgs& a, const RtLaunchCfg& cfg, hipStream_t s);
hipError_t rt_launch_assemble(const uint8_t* gathered, uint8_t* frame, uint32_t W, uint32_t H,
                              uint32_t world, uint32_t padded_tiles, hipStream_t s);
const char* rt_variant_name(int variant);
