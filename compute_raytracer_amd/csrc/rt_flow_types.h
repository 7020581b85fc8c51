// rt_flow_types.h -- launch arguments of the persistent triangle kernel (rt_flow.hip), shared with the host layer.
#pragma once
#include "rt_tri_types.h"

constexpr uint32_t kFlowKB = 8u;        // BLAS stack slots a lane keeps in LDS (the reference's stack has 20, RK:71); deeper ones: overflow area
constexpr uint32_t kFlowKT = 2u;        // TLAS stack slots in LDS
constexpr uint32_t kFlowInst = 16u;     // instance records a workgroup stages (= the instances that travel with a frame, rt_ctx.h: kInstMax)
constexpr uint32_t kFlowOvfWords = (20u - kFlowKB) * 64u + (20u - kFlowKT) * 32u;   // 32-bit words of overflow stack per wave

struct RtFlowArgs {
    const float4* pairs;                // [n_pairs][4]: the relinked child pairs (rt_flow_build.h), most visited first
    uint32_t n_pairs;
    uint32_t lds_pairs;                 // records [0, lds_pairs) are staged in LDS (set by the launcher)
    uint32_t thresh;                    // complete rays a wave collects before it runs the shading block
    uint32_t root_meta[kFlowInst];      // per instance: (count << 16 | left) of its root node, inner roots relinked
    uint32_t* ovf;                      // overflow stacks: kFlowOvfWords words per wave of the grid
    uint32_t lds_pairs_cap;             // development knob: at most this many staged records (0: no limit)
    uint32_t static_pct;                // trace_tiles: this share (0 ... 100) of every wave's items is assigned statically, strided over the work list
};

size_t rt_flow_lds_bytes(uint32_t waves, uint32_t lds_pairs);
uint32_t rt_flow_lds_pairs(uint32_t waves, uint32_t per_cu, uint32_t n_pairs);
hipError_t rt_launch_flow(const RtFrameArgs& a, const RtTriScene& t, const RtFlowArgs& f, uint32_t waves, uint32_t per_cu, uint32_t blocks, bool steps, hipStream_t s);
