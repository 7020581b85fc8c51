// rt_tri_types.h -- device view of the reference's triangle scene, byte layouts of RR:169-229.
#pragma once
#include "rt_types.h"

// The frame's instance data (RR:169-192: the head of the node buffer = the top-level tree, the BLAS records, the BLAS lookup) as the
// SMALL forms of the triangle kernel take it: by value, in the kernel's own argument block -- what they stage in LDS, in the
// layout they stage it in (word 19 of record k: entry k of the lookup table; word 17: the root's relinked meta).  A frame of these
// forms reads nothing else of the per-frame buffers: no apply_instances kernel in front of it, no version of those buffers to wait for.
constexpr uint32_t kInstHeadNodes = 32u, kInstBlas = 16u;       // = rt_tri_device.h kWideNodes, kWideBlas: the most a small form stages
struct RtTriInst {
    float head[8u * kInstHeadNodes];
    float blas[20u * kInstBlas];
};

struct RtTriScene {
    const float4* nodes;       // [n_nodes][2]  {min.xyz, leftChildIndex}, {max.xyz, primitiveCount}
    const float* blas;         // [n_blas][20]  inverseModel (column-major), rootNodeIndex, pad
    const float* tri;          // [n_tri][40]   RR:198-209
    const float4* corners;     // [n_tri_lookup][3]  the library's own: cornerA/B/C of triangles[u32(tri_lookup[slot])] (rt_triangles.hip: tri_corners)
    const float* tri_lookup;   // [n_tri_lookup] f32 indices
    const float* blas_lookup;  // [n_blas_lookup] f32 indices
    const uint8_t* tex;        // meshTex, rgba8unorm
    uint32_t n_nodes, n_blas, n_tri, n_tri_lookup, n_blas_lookup, tex_w, tex_h;
    uint32_t packed_ok;        // every node's count, child index and lookup slot fits 16 bits: the BLAS stack may hold (count, left)
    uint32_t tlas_small;       // the host walked this frame's top-level tree (rt_tlas_fit.h): 1 = leaves at most 4 levels down, all nodes among
                               // the first 16; 2 = at most 3 levels, 8 nodes, 4 instances; 3 = at most 8 levels, 24 nodes; 4 = 8 levels, 32 nodes;
                               // 0 = none of these
    uint32_t p16_ok;           // ... and (count << 14 | x) fits 16: leaves of at most 3 triangles, at most 16,384 pair records and lookup slots
    // Work list (rt_triangles.hip: order_hist / order_scatter): tile_order[0] tiles are rendered as four quarters, tile_order[1] as
    // sixteen 2x2 blocks, tile_order[2...] is the order of the tiles (null: every tile whole, in index order); every workgroup
    // leaves the time it took in tile_cost[tile] by atomicMax -- the parts of a split tile: the longest of theirs, scaled (x 3 a
    // sixteenth, x 1.5 a quarter) -- (null: nowhere), from which the list of the next frame on this stream is made.
    // Relinked copy of the BLAS trees (rt_flow_build.h), when the scene fits it (rt_api.hip: flow_ok; null otherwise): the two
    // children of an inner node as one 64-byte record with packed (count << 16 | x) metas, and per instance the root's meta.
    const float4* pairs;
    uint32_t root_meta[16];    // instances the tile kernel stages (rt_tri_device.h: kLdsBlas; kWideBlas in the widest small form)
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    uint32_t form;             // the stack form rt_tri_stack_form chose for this frame (0 / 1 / 2 = SMALL of rt_triangles.hip); forms 1 and 2 read `inst`
    RtTriInst inst;
    uint32_t in_flight;        // the caller keeps frames in flight (rt_api.hip: pipelined_hint): throughput over latency
    unsigned long long* dbg;   // development builds (tools/tri_timeline.py): per workgroup {start, end (100 MHz ticks), tile << 8 | part}; null: off
    uint32_t prio;             // development builds: wave priority of the head of the work list (rt_triangles.hip)
    uint32_t roles;            // the frame's work list splits tiles (the caller knows from the pinned word order_hist leaves): forms 1 and 3 then run as trace_roles (rt_triangles.hip)
    uint32_t cost_mul4, cost_mul16;   // ... whose quarters / sixteenths leave their time x this / 8 in tile_cost
    uint32_t xcd_rows;         // set by the launch (no work list): workgroup b renders row (b % 8) + 8 (b / 8 / tiles per row) -- a row per XCD
};

// Which stack form rt_launch_triangles runs the frame as: 0 twenty TLAS slots, 1 four (five waves per SIMD), 2 three (six waves),
// 3 eight (five waves), 4 eight with sixteen staged instances.  The caller stores the answer in t.form before the launch and, for
// 1 - 4, fills t.inst.
int rt_tri_stack_form(const RtTriScene& t, int heatmap);
hipError_t rt_launch_triangles(const RtFrameArgs& a, const RtTriScene& t, int heatmap, hipStream_t s);
uint32_t rt_order_scan_words(void);      // words of scan space rt_launch_order_tiles needs, zeroed once (it leaves them zero)
hipError_t rt_launch_order_hist(uint32_t* cost, uint32_t* scan, uint32_t* order, uint32_t n_tiles, uint32_t wave_slots,
                                unsigned long long* counters, unsigned long long* host, uint32_t words, unsigned long long* split_out,
                                hipStream_t s);   // (+ the frame's epilogue); split_out: pinned word that receives the number of tiles the list splits, or null
hipError_t rt_launch_order_scatter(uint32_t* cost, uint32_t* scan, uint32_t* order, uint32_t n_tiles, hipStream_t s);
hipError_t rt_launch_tri_corners(float4* out, const float* tri, const float* lookup, uint32_t n_slots, uint32_t n_tri, hipStream_t s);
