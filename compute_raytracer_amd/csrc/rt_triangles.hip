// rt_triangles.hip -- the reference's LIVE scene type on gfx950: triangles behind a two-level
// BVH (TLAS over model instances, one BLAS per mesh), SURVEY.md 8(f) row 1, and the heatmap twin
// of the kernel (row 4).  Compiled like rt_kernels.hip with -ffp-contract=off: every floating
// point expression is the reference's, correctly rounded, in the reference's order, so frames are
// bit-identical to oracle/rt_oracle.c (rt_oracle_render_tri / rt_oracle_heatmap_tri).
//
// Reference (citations relative to the reference repository):
//   RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl   traceTLAS RK:168-244, traceBLAS
//        RK:246-341, hitTriangle RK:344-393, hitAABB RK:395-410
//   HK = src/rendering-raycast/shaders/heatmap-kernel.wgsl     main HK:63-83, counters HK:143,242,279
// Buffers arrive in exactly the byte layouts RendererRaytracing writes (RR:169-229): 160-B
// triangles, 32-B nodes, 80-B BLAS records, f32 lookup tables.
//
// CDNA4 mapping: one pixel per lane, 8x8 tile per wave64; the two traversal stacks of a lane
// (`array<u32, 20>` each in the WGSL) live in LDS, slot-major ([slot][thread]) so that a wave's
// push or pop touches 64 different banks; nodes are read as two float4 (the 32-B node is
// exactly {min.xyz, leftChild | max.xyz, count}); interpolated normal, texture coordinate and
// the normal's model transform are formed once for the final nearest hit (they depend only on
// the winning triangle, its barycentrics and its BLAS, so deferring them is bit-exact).
// Out-of-range indices follow the robustness rule the oracle fixes (clamp to the last element).
#include <type_traits>
#include <cstdlib>
#include <algorithm>

#include "rt_device.h"
#include "rt_tri_types.h"


#include "rt_tri_device.h"

namespace rtk {

// ---- kernel: RK main over the triangle scene ---------------------------------------------------------
// What a path carries across a traversal, and where.  The two traversals of a bounce (RK:114, RK:153) are the kernel: the
// registers they need decide how many waves a SIMD holds.  Everything the bounce loop needs AFTER a traversal but not IN it
// is therefore (a) reduced to what will be used -- the shading of a hit is formed before its shadow ray is cast: the light
// term down to the one product that the shadow test switches on or off (RK:160-162), albedo and texture sample (RK:133-134)
// as their sum; the running mean's weights `affect` and `sum` (RK:139-140) are the same for every lane still in the loop
// and live in scalar registers -- and (b) in the SMALL form parked in LDS, one dword per lane and value, slot-major like the
// stacks (a wave's access touches 64 banks).  Round 4's five-wave form left that choice to the register allocator: 10-13
// dwords of scratch per lane, 56 MB of HBM writes per frame of the reference's scene for a 4.5 MB picture.
//   slot 0-2 colour, 3 dist (written once, at bounce 0), 4 the lit intensity, 5-7 albedo, 8-10 the reflected direction;
//   a lane that LEAVES the loop on a miss reuses 4-10 for what its sky sample needs: direction, affect, sum.
// The sky a path escapes into (RK:122-126) is sampled after the loop, by all lanes of the wave that ended on a miss at once
// (inside the loop the filter ran for the lanes that missed at THAT bounce while the others waited), through the same one
// copy of the filter that then samples the fog colour (RK:92): a loop of one or two trips.
// Slots [0, PARK) are in LDS, the rest stay where the compiler puts them (PARK = 0: no LDS to spare; 11: everything parked;
// 9 / 6: the forms whose deeper TLAS stack and larger staged head leave room for nine / six -- the rest ride in registers).
template <int PARK, uint32_t STRIDE> struct Parked {
    static_assert(PARK >= 0 && PARK <= 11, "eleven values are carried");
    typedef volatile __attribute__((address_space(3))) float* lds_f32;   // an LDS address (ds_write_b32 / ds_read_b32 with the slot as
    lds_f32 base;                                                          // immediate offset); volatile: a store is a store, a load a load
    float r[11];
    __device__ __forceinline__ explicit Parked(float* lane_column) : base((lds_f32)lane_column) {}
    template <int K> __device__ __forceinline__ void put(float v) { if (K < PARK) base[K * STRIDE] = v; else r[K] = v; }
    template <int K> __device__ __forceinline__ float get() const { return K < PARK ? (float)base[K * STRIDE] : r[K]; }
};
// one ray per lane that is here: counted per wave, by one lane, in LDS -- not in a register of every lane
__device__ __forceinline__ void count_traversal(uint32_t lane, uint32_t* wave_rays) {
    const uint64_t here = __ballot(1);
    if (lane == (uint32_t)__builtin_amdgcn_readfirstlane((int)lane)) *wave_rays += (uint32_t)__popcll(here);
}
// The frame's arguments, read AGAIN from the kernarg segment (they are the kernel's first parameter: offset 0): what the
// epilogue needs of them -- camera basis, sky faces, sizes, the output pointer -- then does not sit in scalar registers through
// the traversals (the loop's own constants, exec masks of six nested divergent levels and the buffer pointers fill the 102 there are;
// what did not fit went to lanes of a VGPR, and to a scratch frame the dispatcher had to provide although nothing was ever stored in it).
__device__ __forceinline__ const RtFrameArgs& reread_first_kernarg() {
    typedef const __attribute__((address_space(4))) RtFrameArgs* kernarg_ptr;
    kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                                   // opaque: not the pointer the prologue loaded from
    return *(const RtFrameArgs*)p;
}
__device__ __forceinline__ float uniform(float v) {               // a value every ACTIVE lane holds: keep it in a scalar register
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// FLAT: compiled for a one-colour 1x1 sky (A.sky_flat) -- no cube filtering code at all.
// SMALL: the frame's top-level tree passed rt_tlas_fits (rt_tlas_fit.h).  1: kSmallStack TLAS slots, kSmallNodes staged nodes,
// and the LDS that frees is the parking place above (7,364 bytes per wave: five waves per SIMD);  2: kTinyStack slots, kTinyNodes
// nodes, kTinyBlas instance records (6,340 bytes per wave: six waves per SIMD fit a CU's 160 KB);  3: kMidStack slots, kMidNodes
// nodes -- every top-level tree twelve instances can have except a degenerate one (2 M - 1 = 23 nodes; depth 8) --, nine of the
// eleven values parked (7,620 bytes, five waves);  4: the same stack, kWideNodes nodes and kWideBlas instance records -- whatever
// sixteen instances, the most that travel with a frame, can have --, six values parked (7,428 bytes, five waves).
template <int WAVES, typename STK, int OCC, bool FLAT, bool PACKED, bool PAIRS = false, bool P16 = false, int SMALL = 0>
__global__ __launch_bounds__(64 * WAVES, OCC) void trace_triangles(const RtFrameArgs A, const RtTriScene T) {
    typedef typename std::conditional<PACKED && !P16, uint32_t, STK>::type BSTK;
    constexpr uint32_t TS = SMALL >= 3 ? kMidStack : (SMALL == 2 ? kTinyStack : (SMALL == 1 ? kSmallStack : kStack));
    constexpr uint32_t NODES = SMALL == 4 ? kWideNodes : (SMALL == 3 ? kMidNodes : (SMALL == 2 ? kTinyNodes : (SMALL == 1 ? kSmallNodes : kLdsNodes)));
    constexpr uint32_t BLAS = SMALL == 4 ? kWideBlas : (SMALL == 2 ? kTinyBlas : kLdsBlas);
    constexpr int PARK = SMALL == 4 ? 6 : (SMALL == 3 ? 9 : (SMALL ? 11 : 0));
    __shared__ STK tstacks[TS * 64 * WAVES];
    __shared__ BSTK bstacks[kStack * 64 * WAVES];
    STK* tstack = tstacks + threadIdx.x;
    BSTK* bstack = bstacks + threadIdx.x;
    constexpr uint32_t stride = 64 * WAVES;
    __shared__ float4 s_nodes[2 * NODES];
    __shared__ float s_blas[20 * BLAS];
    __shared__ float s_park[PARK ? PARK * 64 * WAVES : 1];
    __shared__ uint32_t s_rays[WAVES];             // traversals of this wave's lanes (the frame's ray counter, RK:114 + RK:153)
    if ((threadIdx.x & 63u) == 0u) s_rays[threadIdx.x >> 6] = 0u;
    Parked<PARK, stride> pk(s_park + threadIdx.x);
    const TriLds L = stage_head<WAVES, NODES, BLAS, (SMALL != 0)>(T, s_nodes, s_blas);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // one workgroup per tile, or per quarter of a tile the previous frame on this stream found long (order_tiles)
    const uint64_t clk0 = wall_clock64();
    const uint32_t groups_x = (A.W + 8u * WAVES - 1u) / (8u * WAVES);
    const uint32_t n_tiles = groups_x * A.n_local_tiles;
    // part 0-3: a 4x4 quarter; 4: the whole tile; 16-31: a 2x2 sixteenth (the very longest tiles: four lanes diverge least, and
    // a frame on its own has idle wave slots to spare)
    uint32_t tile = blockIdx.x, part = 4u;
    if (T.tile_order) {
        const uint32_t s4 = T.tile_order[0], s16 = T.tile_order[1];
        const uint32_t* list = T.tile_order + 2;
        uint32_t i = blockIdx.x;
        if (i < 16u * s16) { tile = list[i >> 4]; part = 16u + (i & 15u); }
        else if ((i -= 16u * s16) < 4u * s4) { tile = list[s16 + (i >> 2)]; part = i & 3u; }
        else if ((i -= 4u * s4) < n_tiles - s16 - s4) tile = list[s16 + s4 + i];
        else return;                                                        // the grid is sized for the most parts there can be
#ifdef RT_TRI_DEV_ENV
        if (T.prio >= 1u && part != 4u) __builtin_amdgcn_s_setprio(3);
        else if (T.prio >= 2u && i < (n_tiles >> 4)) __builtin_amdgcn_s_setprio(2);
#endif
    } else if (T.xcd_rows != 0u) {
        // Workgroups go to the eight XCDs in turn (b % 8) and each XCD has an L2 of its own: in index order every XCD renders
        // every eighth tile of every row and its L2 holds what the whole band of rows in flight touches.  Here XCD x renders rows
        // x, x + 8, ... left to right: the tiles it holds at any time are neighbours (the grid is padded to eight rows).
        const uint32_t k = blockIdx.x >> 3;
        const uint32_t r = (blockIdx.x & 7u) + 8u * (k / groups_x);
        if (r >= A.n_local_tiles) return;
        tile = r * groups_x + (k - (k / groups_x) * groups_x);
    }
    const uint32_t by = tile / groups_x, bx = tile - by * groups_x;
    if ((part < 4u && lane >= 16u) || (part >= 16u && lane >= 4u)) return;
    uint32_t px = lane & 7u, row = lane >> 3;
    if (part < 4u) { px = 4u * (part & 1u) + (lane & 3u); row = 4u * (part >> 1) + (lane >> 2); }
    else if (part >= 16u) { px = 2u * (part & 3u) + (lane & 1u); row = 2u * ((part >> 2) & 3u) + (lane >> 1); }
    const uint32_t x = bx * (8u * WAVES) + wave * 8u + px;
    const uint32_t y = (A.tile_first + by * A.tile_step) * 8u + row;
    if (x >= A.W || y >= A.H) return;          // thread 0 leaves here only with its whole tile or part: its pixel is their first

    const Scene sc = unpack_scene(A);
    float dummy = 0.0f;
    pk.template put<0>(1.0f); pk.template put<1>(1.0f); pk.template put<2>(1.0f);   // color = (1, 1, 1), RK:103
    pk.template put<3>(0.0f);                                                        // dist: 0 unless the primary ray hits (RK:116-118)
    v3 ro = sc.cameraPos, rd = primary_dir(A, sc, x, y);
    float affect = 1.0f, sum = 0.0f;
    bool missed = false;
    for (uint32_t bounce = 0; bounce < sc.bounces; ++bounce) {                       // RK:113
        affect = uniform(affect); sum = uniform(sum);                                // 2^-bounce, 2 - 2^(1-bounce): one value per wave
        const TriHit h = trace_tlas<false, STK, PACKED, PAIRS, P16, TS>(T, L, ro, rd, tstack, bstack, stride, dummy); // RK:114
        count_traversal(lane, &s_rays[wave]);
        if (h.tri < 0) {                                                             // RK:122-126: sampled after the loop
            pk.template put<4>(rd.x); pk.template put<5>(rd.y); pk.template put<6>(rd.z);
            pk.template put<7>(affect); pk.template put<8>(sum);
            missed = true;
            break;
        }
        if (bounce == 0) pk.template put<3>(h.t);                                    // RK:116-118
        const float next = affect + sum;                                             // RK:120
        // (SMALL: every instance record is staged, and the per-frame buffers are not this frame's: rt_tri_types.h RtTriInst)
        const v3 normal = hit_normal(T, h, SMALL ? L.blas + 20u * (uint32_t)h.blas : T.blas + 20u * (size_t)h.blas);
        const Albedo s = hit_albedo(T, h.tri, h.u, h.v);
        ro = add(ro, scale(h.t, rd));                                                // RK:129
        rd = normalize(reflect(rd, normal));                                         // RK:130
        // RK:146-166 lightIntensity: everything but the shadow ray's verdict is known before the ray is cast
        const v3 sdir = normalize(sub(ro, sc.lightPos));                             // RK:147
        const float distance = length(sdir);                                         // RK:148
        {
            const float power = clampf(dot(normal, V(-sdir.x, -sdir.y, -sdir.z)), sc.minIntensity, 1.0f);   // RK:160
            const float cap = sc.lightIntensity / (sc.lightIntensity + distance);                           // RK:161
            pk.template put<4>(power * cap);                                                                // RK:162
            const v3 diffuseColor = scale(s.w, s.rgb);                               // RK:133
            const v3 samplerColor = scale(1.0f - s.w, tex2d_sample(T, s.u, s.v));    // RK:134
            const v3 albedo = add(diffuseColor, samplerColor);                       // RK:135, the sum
            pk.template put<5>(albedo.x); pk.template put<6>(albedo.y); pk.template put<7>(albedo.z);
            pk.template put<8>(rd.x); pk.template put<9>(rd.y); pk.template put<10>(rd.z);
        }
        const TriHit sh = trace_tlas<false, STK, PACKED, PAIRS, P16, TS>(T, L, sc.lightPos, sdir, tstack, bstack, stride, dummy);   // RK:153
        count_traversal(lane, &s_rays[wave]);
        float intensity = sc.minIntensity;                                           // RK:165
        if (sh.tri >= 0) {                                                           // RK:155
            const v3 hp = add(sc.lightPos, scale(sh.t, sdir));                       // RK:156
            const v3 dv = sub(hp, ro);                                               // RK:157-159: see light_term (rt_device.h)
            if (dot(dv, dv) < 0x1.a36e2cp-16f) intensity = pk.template get<4>();
        }
        const v3 albedo = V(pk.template get<5>(), pk.template get<6>(), pk.template get<7>());
        const v3 color = V(pk.template get<0>(), pk.template get<1>(), pk.template get<2>());
        const v3 blended = scale(intensity, albedo);                                 // RK:135
        const v3 mixed = divs(add(scale(sum, color), scale(affect, blended)), next); // RK:136
        pk.template put<0>(mixed.x); pk.template put<1>(mixed.y); pk.template put<2>(mixed.z);
        rd = V(pk.template get<8>(), pk.template get<9>(), pk.template get<10>());
        affect = affect / 2.0f;                                                      // RK:139
        sum = next;                                                                  // RK:140
    }
    v3 color = V(pk.template get<0>(), pk.template get<1>(), pk.template get<2>());
    const float dist = pk.template get<3>();
    // One copy of the cube filter: trip 0 (lanes whose path escaped) the sky along the path's last direction, blended into the
    // running mean (RK:123-125); trip 1 the fog colour, the sky along the primary ray (RK:92) -- its direction is formed again
    // here rather than carried through the traversals.
    v3 fog = V(0.0f, 0.0f, 0.0f);
    v3 dir = V(0.0f, 0.0f, 0.0f);
    float affect_e = 0.0f, sum_e = 0.0f;
    if (missed) {
        dir = V(pk.template get<4>(), pk.template get<5>(), pk.template get<6>());
        affect_e = pk.template get<7>(); sum_e = pk.template get<8>();
    }
    // (the pixel's place is formed again from the lane number, asked of the hardware: nothing of it is carried through the loop)
    const uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t px_e = lane_e & 7u, row_e = lane_e >> 3;
    if (part < 4u) { px_e = 4u * (part & 1u) + (lane_e & 3u); row_e = 4u * (part >> 1) + (lane_e >> 2); }
    else if (part >= 16u) { px_e = 2u * (part & 3u) + (lane_e & 1u); row_e = 2u * ((part >> 2) & 3u) + (lane_e >> 1); }
    const RtFrameArgs& Ae = reread_first_kernarg();
    const Scene sce = unpack_scene(Ae);
    const uint32_t x_e = bx * (8u * WAVES) + wave * 8u + px_e;
    const uint32_t y_e = (Ae.tile_first + by * Ae.tile_step) * 8u + row_e;
#pragma unroll 1
    for (int k = missed ? 0 : 1; k < 2; ++k) {
        if (k == 1) dir = primary_dir(Ae, sce, x_e, y_e);
        const v3 sky = scale(sce.minIntensity, cube_sample<FLAT ? 1 : 0, true>(Ae, dir));
        if (k == 0) color = divs(add(scale(sum_e, color), scale(affect_e, sky)), affect_e + sum_e);   // RK:120, 125
        else fog = sky;
    }
    const uint32_t opix = (by * 8u + row_e) * Ae.W + x_e;
    reinterpret_cast<uint32_t*>(Ae.out)[opix] = compose_pixel_sky(fog, color, dist);   // RK:91-98
    if (lane_e == (uint32_t)__builtin_amdgcn_readfirstlane((int)lane_e)) count_wave_rays(Ae.rays, s_rays[wave]);
    // What a tile leaves for the next frame's work list (10 ns ticks) must not depend on how it was rendered, or the list chases
    // its own tail: a whole tile leaves its time; the parts of a split tile leave the LONGEST of theirs, times what a wave of 64
    // diverging lanes takes longer than its slowest sixteenth (3) or quarter (1.5) alone.  Summed, a split tile looked several times
    // as long as it is and kept its place in the head of the list whatever the other tiles did: the four streams' lists settled in
    // different selections (the reference's scene one frame at a time: 0.43 and 0.49 ms alternating with the streams); the plain
    // maximum made it look short, and every list alternated between splitting a tile and not (0.45 / 0.55).  With the factors:
    // 0.385-0.395 ms, every frame, every stream (profiles/r04/tri_cost_series.log).
#ifdef RT_TRI_DEV_ENV
    if (T.dbg && threadIdx.x == 0u) {
        T.dbg[3u * blockIdx.x] = clk0; T.dbg[3u * blockIdx.x + 1u] = wall_clock64(); T.dbg[3u * blockIdx.x + 2u] = ((unsigned long long)tile << 8) | part;
    }
#endif
    if (T.tile_cost && threadIdx.x == 0u) {
        const uint32_t dt = (uint32_t)(wall_clock64() - clk0);
        atomicMax(&T.tile_cost[tile], part >= 16u ? 3u * dt : (part < 4u ? dt + (dt >> 1) : dt));
    }
}

// ---- the kernel for AWAITED frames with a work list: a split tile's idle lanes trace ahead ---------------------------------
// An awaited frame is as long as its longest waves, and those are the quarters and sixteenths of its longest tiles: 16 or 4
// lanes whose paths are 2 x bounces DEPENDENT traversals each.  But only almost: the shadow ray of bounce b (RK:153) and the
// reflection ray of bounce b + 1 (RK:114) both start from what the hit of bounce b gives -- the point, the normal -- and
// neither needs the other.  In a quarter 48 lanes idle, in a sixteenth 60.  Here lane k + n of a part of n pixels is the HELPER
// of lane k: while the owner walks the shadow ray the helper walks the next reflection ray, through the same call of the
// traversal (one copy of its code, each lane its own stacks), and hands the hit back through LDS -- the helper's own parking
// column is the mailbox: slots 0-5 the ray, 6 whether there is one; then 0-4 the hit.  A path of B bounces is then 1 + B traversals long
// instead of 2 B (each as long as the longer of the two), and so is the wave.  No ray is cast that the reference does not cast
// (the owner posts the next reflection ray only when the bounce limit leaves room for it: the loop of RK:113 would cast it), none is cast twice:
// the ray counter stays exact.  A whole tile has no idle lanes: its owners walk reflection ray and shadow ray in turn, through
// the same one call.  Every floating-point expression is the one of trace_triangles, in its order.
// The shading of a hit before its shadow ray is cast (RK:129-135, RK:147-162; the block of trace_triangles): what survives the
// shadow ray goes to the parked slots 4-10; -> the hit point and the shadow ray's direction.
template <class PK>
__device__ __forceinline__ void shade_hit(const RtTriScene& T, const TriLds& L, const Scene& sc, PK& pk, const TriHit& h, v3 o, v3 d, v3& ro, v3& sdir) {
    const v3 normal = hit_normal(T, h, L.blas + 20u * (uint32_t)h.blas);
    const Albedo s = hit_albedo(T, h.tri, h.u, h.v);
    ro = add(o, scale(h.t, d));                                                  // RK:129
    const v3 rd = normalize(reflect(d, normal));                                 // RK:130
    sdir = normalize(sub(ro, sc.lightPos));                                      // RK:147
    const float distance = length(sdir);                                         // RK:148
    const float power = clampf(dot(normal, V(-sdir.x, -sdir.y, -sdir.z)), sc.minIntensity, 1.0f);   // RK:160
    const float cap = sc.lightIntensity / (sc.lightIntensity + distance);                           // RK:161
    pk.template put<4>(power * cap);                                                                // RK:162
    const v3 diffuseColor = scale(s.w, s.rgb);                                   // RK:133
    const v3 samplerColor = scale(1.0f - s.w, tex2d_sample(T, s.u, s.v));        // RK:134
    const v3 albedo = add(diffuseColor, samplerColor);                           // RK:135, the sum
    pk.template put<5>(albedo.x); pk.template put<6>(albedo.y); pk.template put<7>(albedo.z);
    pk.template put<8>(rd.x); pk.template put<9>(rd.y); pk.template put<10>(rd.z);
}
// The shadow ray's verdict (RK:155-166) and the running mean (RK:120, RK:135-136) into the parked colour.
template <class PK>
__device__ __forceinline__ void blend_bounce(const Scene& sc, PK& pk, const TriHit& sh, v3 sdir, v3 ro, float affect, float sum) {
    float intensity = sc.minIntensity;                                           // RK:165
    if (sh.tri >= 0) {                                                           // RK:155
        const v3 hp = add(sc.lightPos, scale(sh.t, sdir));                       // RK:156
        const v3 dv = sub(hp, ro);                                               // RK:157-159: see light_term (rt_device.h)
        if (dot(dv, dv) < 0x1.a36e2cp-16f) intensity = pk.template get<4>();
    }
    const float next = affect + sum;                                             // RK:120
    const v3 albedo = V(pk.template get<5>(), pk.template get<6>(), pk.template get<7>());
    const v3 color = V(pk.template get<0>(), pk.template get<1>(), pk.template get<2>());
    const v3 blended = scale(intensity, albedo);                                 // RK:135
    const v3 mixed = divs(add(scale(sum, color), scale(affect, blended)), next); // RK:136
    pk.template put<0>(mixed.x); pk.template put<1>(mixed.y); pk.template put<2>(mixed.z);
}

template <typename STK, int OCC, bool FLAT, int SMALL>
__global__ __launch_bounds__(64, OCC) void trace_roles(const RtFrameArgs A, const RtTriScene T) {
    constexpr int WAVES = 1;
    constexpr uint32_t TS = SMALL >= 3 ? kMidStack : (SMALL == 2 ? kTinyStack : (SMALL == 1 ? kSmallStack : kStack));
    constexpr uint32_t NODES = SMALL == 4 ? kWideNodes : (SMALL == 3 ? kMidNodes : (SMALL == 2 ? kTinyNodes : (SMALL == 1 ? kSmallNodes : kLdsNodes)));
    constexpr uint32_t BLAS = SMALL == 4 ? kWideBlas : (SMALL == 2 ? kTinyBlas : kLdsBlas);
    constexpr int PARK = SMALL == 4 ? 6 : (SMALL == 3 ? 9 : (SMALL ? 11 : 0));
    static_assert(PARK >= 7, "the mailbox is seven parked slots of the helper's column");
    __shared__ STK tstacks[TS * 64];
    __shared__ STK bstacks[kStack * 64];
    STK* tstack = tstacks + threadIdx.x;
    STK* bstack = bstacks + threadIdx.x;
    constexpr uint32_t stride = 64;
    __shared__ float4 s_nodes[2 * NODES];
    __shared__ float s_blas[20 * BLAS];
    __shared__ float s_park[PARK * 64];
    __shared__ uint32_t s_rays[1];
    if (threadIdx.x == 0u) s_rays[0] = 0u;
    const TriLds L = stage_head<WAVES, NODES, BLAS, true>(T, s_nodes, s_blas);

    const uint32_t lane = threadIdx.x;
    const uint64_t clk0 = wall_clock64();
    const uint32_t groups_x = (A.W + 7u) / 8u;
    const uint32_t n_tiles = groups_x * A.n_local_tiles;
    uint32_t tile = blockIdx.x, part = 4u;
    if (T.tile_order) {
        const uint32_t s4 = T.tile_order[0], s16 = T.tile_order[1];
        const uint32_t* list = T.tile_order + 2;
        uint32_t i = blockIdx.x;
        if (i < 16u * s16) { tile = list[i >> 4]; part = 16u + (i & 15u); }
        else if ((i -= 16u * s16) < 4u * s4) { tile = list[s16 + (i >> 2)]; part = i & 3u; }
        else if ((i -= 4u * s4) < n_tiles - s16 - s4) tile = list[s16 + s4 + i];
        else return;
    }
    const uint32_t by = tile / groups_x, bx = tile - by * groups_x;
    const bool split = part != 4u;
    const uint32_t owners = part < 4u ? 16u : (part >= 16u ? 4u : 64u);
    if (split && lane >= 2u * owners) return;
    const bool helper = split && lane >= owners;
    const uint32_t ol = helper ? lane - owners : lane;               // the pixel's lane: a helper leaves where its owner leaves
    uint32_t px = ol & 7u, row = ol >> 3;
    if (part < 4u) { px = 4u * (part & 1u) + (ol & 3u); row = 4u * (part >> 1) + (ol >> 2); }
    else if (part >= 16u) { px = 2u * (part & 3u) + (ol & 1u); row = 2u * ((part >> 2) & 3u) + (ol >> 1); }
    const uint32_t x = bx * 8u + px;
    const uint32_t y = (A.tile_first + by * A.tile_step) * 8u + row;
    if (x >= A.W || y >= A.H) return;

    Parked<PARK, stride> pk(s_park + lane);                                             // an owner's carried values (as in trace_triangles)
    Parked<PARK, stride> mb(s_park + (helper ? lane : (split ? lane + owners : lane))); // the pair's mailbox: the helper's column
    const Scene sc = unpack_scene(A);
    float dummy = 0.0f;
    bool missed = false;
    if (!helper) {
        pk.template put<0>(1.0f); pk.template put<1>(1.0f); pk.template put<2>(1.0f);   // color = (1, 1, 1), RK:103
        pk.template put<3>(0.0f);                                                        // dist, RK:116-118
    }
    // One loop, one call of the traversal per trip.  An owner's trips: the primary ray, then per bounce the shadow ray -- and, in a
    // whole tile, where nobody walks it for him, the next reflection ray in a trip of its own.  A helper's: whatever its mailbox holds.
    v3 o = sc.cameraPos, d = V(0.0f, 0.0f, 0.0f), ro = V(0.0f, 0.0f, 0.0f);
    bool act = false, shadow = false;
    uint32_t bounce = 0u;
    float affect = 1.0f, sum = 0.0f;
    if (!helper) {
        d = primary_dir(A, sc, x, y);
        act = sc.bounces > 0u;                                                           // RK:113
        if (split) mb.template put<6>(0.0f);                                             // nothing for the helper yet
    }
    for (;;) {
        __builtin_amdgcn_wave_barrier();
        if (helper) {
            act = mb.template get<6>() != 0.0f;
            if (act) {
                o = V(mb.template get<0>(), mb.template get<1>(), mb.template get<2>());
                d = V(mb.template get<3>(), mb.template get<4>(), mb.template get<5>());
            }
        }
        if (__ballot(act) == 0ull) break;
        TriHit h; h.t = 0.0f; h.u = h.v = 0.0f; h.tri = -1; h.blas = -1;
        if (act) {
            h = trace_tlas<false, STK, true, true, true, TS>(T, L, o, d, tstack, bstack, stride, dummy);   // RK:114 / RK:153
            count_traversal(lane, &s_rays[0]);
        }
        if (helper && act) {                                       // the hit goes back to the owner
            mb.template put<0>(h.t); mb.template put<1>(h.u); mb.template put<2>(h.v);
            mb.template put<3>(__int_as_float(h.tri)); mb.template put<4>(__int_as_float(h.blas));
        }
        __builtin_amdgcn_wave_barrier();
        if (!helper && act) {
            affect = uniform(affect); sum = uniform(sum);          // every owner of the wave is at the same bounce, in the same kind of trip
            bool have = !shadow;                                   // h is the hit of a reflection (or the primary) ray
            if (shadow) {                                          // h is the shadow ray's: RK:155-166, then the blend RK:135-140
                blend_bounce(sc, pk, h, d, ro, affect, sum);
                sum = affect + sum;                                                      // RK:140 (next, RK:120)
                affect = affect / 2.0f;                                                  // RK:139
                bounce += 1u;
                shadow = false;
                o = ro;                                                                  // the next reflection ray (RK:129-130) ...
                d = V(pk.template get<8>(), pk.template get<9>(), pk.template get<10>());
                if (bounce >= sc.bounces) act = false;                                   // RK:113
                else if (split) {                                                        // ... which the helper has walked meanwhile
                    h.t = mb.template get<0>(); h.u = mb.template get<1>(); h.v = mb.template get<2>();
                    h.tri = __float_as_int(mb.template get<3>()); h.blas = __float_as_int(mb.template get<4>());
                    have = true;
                }
            }
            if (have && act) {
                if (h.tri < 0) {                                                         // RK:122-126: sampled after the loop
                    pk.template put<4>(d.x); pk.template put<5>(d.y); pk.template put<6>(d.z);
                    pk.template put<7>(affect); pk.template put<8>(sum);
                    missed = true; act = false;
                    if (split) mb.template put<6>(0.0f);
                } else {
                    if (bounce == 0u) pk.template put<3>(h.t);                           // RK:116-118
                    v3 sdir;
                    shade_hit(T, L, sc, pk, h, o, d, ro, sdir);
                    if (split) {                                                         // the helper's next ray, if the loop of RK:113 casts one
                        if (bounce + 1u < sc.bounces) {
                            mb.template put<0>(ro.x); mb.template put<1>(ro.y); mb.template put<2>(ro.z);
                            mb.template put<3>(pk.template get<8>()); mb.template put<4>(pk.template get<9>()); mb.template put<5>(pk.template get<10>());
                            mb.template put<6>(1.0f);
                        } else mb.template put<6>(0.0f);
                    }
                    o = sc.lightPos; d = sdir; shadow = true;                            // RK:153
                }
            }
        }
    }
    if (helper) return;
    v3 color = V(pk.template get<0>(), pk.template get<1>(), pk.template get<2>());
    const float dist = pk.template get<3>();
    v3 fog = V(0.0f, 0.0f, 0.0f);
    v3 dir = V(0.0f, 0.0f, 0.0f);
    float affect_e = 0.0f, sum_e = 0.0f;
    if (missed) {
        dir = V(pk.template get<4>(), pk.template get<5>(), pk.template get<6>());
        affect_e = pk.template get<7>(); sum_e = pk.template get<8>();
    }
    const uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t px_e = lane_e & 7u, row_e = lane_e >> 3;
    if (part < 4u) { px_e = 4u * (part & 1u) + (lane_e & 3u); row_e = 4u * (part >> 1) + (lane_e >> 2); }
    else if (part >= 16u) { px_e = 2u * (part & 3u) + (lane_e & 1u); row_e = 2u * ((part >> 2) & 3u) + (lane_e >> 1); }
    const RtFrameArgs& Ae = reread_first_kernarg();
    const Scene sce = unpack_scene(Ae);
    const uint32_t x_e = bx * 8u + px_e;
    const uint32_t y_e = (Ae.tile_first + by * Ae.tile_step) * 8u + row_e;
#pragma unroll 1
    for (int k = missed ? 0 : 1; k < 2; ++k) {
        if (k == 1) dir = primary_dir(Ae, sce, x_e, y_e);
        const v3 sky = scale(sce.minIntensity, cube_sample<FLAT ? 1 : 0, true>(Ae, dir));
        if (k == 0) color = divs(add(scale(sum_e, color), scale(affect_e, sky)), affect_e + sum_e);   // RK:120, 125
        else fog = sky;
    }
    const uint32_t opix = (by * 8u + row_e) * Ae.W + x_e;
    reinterpret_cast<uint32_t*>(Ae.out)[opix] = compose_pixel_sky(fog, color, dist);   // RK:91-98
    if (lane_e == (uint32_t)__builtin_amdgcn_readfirstlane((int)lane_e)) count_wave_rays(Ae.rays, s_rays[0]);
#ifdef RT_TRI_DEV_ENV
    if (T.dbg && threadIdx.x == 0u) {
        T.dbg[3u * blockIdx.x] = clk0; T.dbg[3u * blockIdx.x + 1u] = wall_clock64(); T.dbg[3u * blockIdx.x + 2u] = ((unsigned long long)tile << 8) | part;
    }
#endif
    // (what a part leaves for the next list: its time scaled to what the whole tile would take as one wave of 64 owners that walk
    // both rays of a bounce themselves -- T.cost_mul16 / cost_mul4, in eighths)
    if (T.tile_cost && threadIdx.x == 0u) {
        const uint32_t dt = (uint32_t)(wall_clock64() - clk0);
        atomicMax(&T.tile_cost[tile], part >= 16u ? (dt * T.cost_mul16) >> 3 : (part < 4u ? (dt * T.cost_mul4) >> 3 : dt));
    }
}

// ---- the order of the next frame's tiles ------------------------------------------------------------------
// The frame is a grid of independent tiles, one wave each, whose costs differ by more than an order of magnitude (a sky tile:
// one ray per pixel; a tile between two mirrors: 2 x bounces dependent traversals by 64 diverging lanes), and the hardware
// starts them in index order.  Measured with the kernel's own clock (tools/tile_cost_probe.py, profiles/r03/tile_cost.log):
// mean tile 21 us, one in a hundred 156 us, the longest 498 us -- and the 1344x846 frame rendered on its own (the
// reference's await-each-frame loop) took 504 us: a frame is as long as its longest tile, however empty the chip.  So
//   * every tile leaves the time it took (tile_cost; the parts of a split tile: the longest of theirs, scaled -- see the kernel's last lines);
//   * two small kernels turn the costs into the NEXT frame's work list on the same stream: tiles longest first (a counting sort
//     over quarter-octave classes), and the tiles longer than half the frame's throughput time -- sum of all costs / wave
//     slots / 2 --, at most one in sixteen and 1024 (a quarter-wave for every wave slot of the chip), as four 4x4 quarters each: a quarter of the lanes diverge a quarter as much, and
//     the frame's longest wave shrinks accordingly; round 4: the tiles longer than TWICE the throughput time, at most one in 64
//     and 64, as sixteen 2x2 blocks (four lanes each) -- a frame on its own leaves the wave slots for them idle anyway.
//     order[0] = tiles in quarters, order[1] = tiles in sixteenths (the head of the list), order[2...] = the permutation.
// The picture does not depend on any of it: a pixel is rendered by the same code whatever its turn and company; any
// array of costs yields a permutation and a split count within the grid's bound.
// Two launches, many workgroups (round 4's single workgroup took 174 us for the 129,600 tiles of a 4K frame, and awaited 4K
// frames therefore kept a work list four frames old): order_hist -- every workgroup counts its slice of the costs by class in
// LDS and adds its counts to the frame's 128 global bins; the workgroup that finishes last turns the bins into the split counts
// and the classes' first positions --, order_scatter -- every workgroup counts its slice again, reserves a range per class with ONE
// global atomic per class it holds, and places its tiles.  scan: 264 words behind the costs (rt_order_scan_words), zero between frames.
constexpr uint32_t kOrderBlock = 256u, kOrderPerThread = 4u;
__device__ __forceinline__ uint32_t cost_class(uint32_t c) {     // quarter-octave class of a tick count: 0 ... 123, monotone
    if (c < 4u) return c;
    const uint32_t e = 31u - (uint32_t)__clz(c);
    return 4u * e + ((c >> (e - 2u)) & 3u) - 4u;
}
// ctr / host / words: optionally the frame's epilogue (rt_types.h: rt_frame_epilogue_body), run by the first workgroup; not used
// by the product -- the host must hear of the frame's end before these kernels, not after (rt_api.hip: rt_enqueue).
__global__ __launch_bounds__(kOrderBlock) void order_hist(const uint32_t* __restrict__ cost, uint32_t* __restrict__ scan, uint32_t* __restrict__ order,
                                                          uint32_t n, uint32_t wave_slots, uint32_t mult16, uint32_t cap16, uint32_t mult4,
                                                          unsigned long long* __restrict__ ctr, unsigned long long* __restrict__ host, uint32_t words,
                                                          uint32_t multw, unsigned long long* __restrict__ split_out, uint32_t cap4, uint32_t div4) {
    static_assert(kOrderBlock == 256u, "rt_frame_epilogue_body is written for 256 threads");
    __shared__ uint32_t bin[128], start[128];
    __shared__ unsigned long long total;
    __shared__ uint32_t last;
    if (blockIdx.x == 0u && ctr) rt_frame_epilogue_body(ctr, host, words);
    if (threadIdx.x < 128u) bin[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) total = 0ull;
    __syncthreads();
    const uint32_t per = kOrderBlock * kOrderPerThread, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    unsigned long long mine = 0ull;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += kOrderBlock) { const uint32_t c = cost[i]; mine += c; atomicAdd(&bin[cost_class(c)], 1u); }
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x < 128u && bin[threadIdx.x]) atomicAdd(&scan[threadIdx.x], bin[threadIdx.x]);
    if (threadIdx.x == 0u && total) atomicAdd(reinterpret_cast<unsigned long long*>(scan + 128), total);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0u) last = atomicAdd(&scan[130], 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 128u) bin[threadIdx.x] = __hip_atomic_load(&scan[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0u) total = __hip_atomic_load(reinterpret_cast<unsigned long long*>(scan + 128), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // start[k] = tiles in the classes ABOVE k (the longer ones): class k's first position in the list -- every thread its own sum
    if (threadIdx.x < 128u) {
        uint32_t acc = 0u;
        for (uint32_t k = threadIdx.x + 1u; k < 128u; ++k) acc += bin[k];
        start[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0u) {
        const unsigned long long thr = total / (2ull * (wave_slots ? wave_slots : 1u));
        auto cls_of = [&](unsigned long long v) { return cost_class(v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v); };
        const uint32_t kt = cls_of((unsigned long long)mult4 * thr);           // mult4 / 2 times the throughput time: as quarters
        const uint32_t kw = cls_of((unsigned long long)multw * thr);           // multw / 2 times the throughput time (4: twice)
        const uint32_t k16 = cls_of((unsigned long long)mult16 * thr);          // mult16 / 2 times the throughput time: as sixteenths
        uint32_t split = start[kt], split16 = start[k16];      // tiles of the classes above the threshold's: all longer than it
        // some tile takes more than twice the throughput time: the frame waits for it.  (Otherwise quarters only add waves: 4K,
        // 0.77 -> 0.81 ms with them.)
        if (start[kw] == 0u) split = split16 = 0u;
        const uint32_t most = n / div4 < cap4 ? n / div4 : cap4, most16 = n / 64u < cap16 ? n / 64u : cap16;      // (launch_tri sizes the grid for these)
        if (split16 > most16) split16 = most16;
        if (split > most) split = most;
        if (split < split16) split = split16;
        order[0] = split - split16;                // tiles rendered as four quarters ...
        order[1] = split16;                        // ... behind the tiles rendered as sixteen 2x2 blocks: the head of the list
        // ... and, for the host that launches the stream's next frame: does the list split anything?  (pinned memory; read without
        // waiting for it: a frame late is as good)
        if (split_out) __hip_atomic_store(split_out, (unsigned long long)split, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x < 128u) { scan[132u + threadIdx.x] = start[threadIdx.x]; scan[threadIdx.x] = 0u; }   // first positions; bins zero for the next frame
    if (threadIdx.x < 4u) scan[128u + threadIdx.x] = 0u;                                                // total, ticket
}
__global__ __launch_bounds__(kOrderBlock) void order_scatter(uint32_t* __restrict__ cost, uint32_t* __restrict__ scan, uint32_t* __restrict__ order, uint32_t n) {
    __shared__ uint32_t bin[128], base[128];
    if (threadIdx.x < 128u) bin[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t per = kOrderBlock * kOrderPerThread, lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t cls[kOrderPerThread], at[kOrderPerThread];
#pragma unroll
    for (uint32_t k = 0; k < kOrderPerThread; ++k) {
        const uint32_t i = lo + threadIdx.x + k * kOrderBlock;
        cls[k] = i < hi ? cost_class(cost[i]) : 0u;
        at[k] = i < hi ? atomicAdd(&bin[cls[k]], 1u) : 0u;         // this tile's place among the workgroup's tiles of its class
    }
    __syncthreads();
    if (threadIdx.x < 128u) base[threadIdx.x] = bin[threadIdx.x] ? atomicAdd(&scan[132u + threadIdx.x], bin[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kOrderPerThread; ++k) {
        const uint32_t i = lo + threadIdx.x + k * kOrderBlock;
        if (i < hi) {
            order[2u + base[cls[k]] + at[k]] = i;
            cost[i] = 0u;                              // the next frame records its times from zero
        }
    }
}

// ---- kernel: the heatmap twin (HK:63-83) ------------------------------------------------------------
template <int WAVES, typename STK, bool PACKED>
__global__ __launch_bounds__(64 * WAVES) void heatmap_triangles(const RtFrameArgs A, const RtTriScene T) {
    typedef typename std::conditional<PACKED, uint32_t, STK>::type BSTK;
    __shared__ STK tstacks[kStack * 64 * WAVES];
    __shared__ BSTK bstacks[kStack * 64 * WAVES];
    STK* tstack = tstacks + threadIdx.x;
    BSTK* bstack = bstacks + threadIdx.x;
    constexpr uint32_t stride = 64 * WAVES;
    __shared__ float4 s_nodes[2 * kLdsNodes];
    __shared__ float s_blas[20 * kLdsBlas];
    const TriLds L = stage_head<WAVES>(T, s_nodes, s_blas);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * (8u * WAVES) + wave * 8u + (lane & 7u);
    const uint32_t row = lane >> 3;
    const uint32_t y = (A.tile_first + blockIdx.y * A.tile_step) * 8u + row;
    if (x >= A.W || y >= A.H) return;
    const Scene sc = unpack_scene(A);
    const v3 dir0 = primary_dir(A, sc, x, y);                                        // HK:66-76
    float traces = 0.0f;
    (void)trace_tlas<true, STK, PACKED>(T, L, sc.cameraPos, dir0, tstack, bstack, stride, traces);   // HK:96-99, one bounce
    const float g = clampf(traces / 300.0f, 0.0f, 1.0f);                             // HK:79
    const uint32_t q = unorm8(g * 1.0f);                                             // HK:81-82
    const uint32_t opix = (blockIdx.y * 8u + row) * A.W + x;
    reinterpret_cast<uint32_t*>(A.out)[opix] = q | (q << 8) | (q << 16) | 0xFF000000u;
    count_rays(A.rays, 1u);
}

// The corners hitTriangle reads (48 of a triangle's 160 bytes), gathered once into lookup order: entry `slot` holds
// cornerA / B / C of triangles[u32(triangleLookup[slot])] -- the same twelve numbers RK:354-364 reads, so every
// intersection is bit for bit the reference's --, a leaf's triangles sit side by side, and the traversal touches
// neither the lookup table nor the 160-byte records until a hit is shaded.
__global__ void tri_corners(float4* __restrict__ out, const float* __restrict__ tri, const float* __restrict__ lookup,
                            uint32_t n_slots, uint32_t n_tri) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    uint32_t ti = u32f(lookup[slot]);
    if (ti >= n_tri) ti = n_tri - 1u;
    const float4* tr = reinterpret_cast<const float4*>(tri + 40u * (size_t)ti);
    out[3u * (size_t)slot] = tr[0];
    out[3u * (size_t)slot + 1u] = tr[3];
    out[3u * (size_t)slot + 2u] = tr[6];
}

}  // namespace rtk

hipError_t rt_launch_tri_corners(float4* out, const float* tri, const float* lookup, uint32_t n_slots, uint32_t n_tri, hipStream_t s) {
    if (n_slots == 0 || n_tri == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::tri_corners, dim3((n_slots + 255u) / 256u), dim3(256), 0, s, out, tri, lookup, n_slots, n_tri);
    return hipGetLastError();
}

// OCC: waves per SIMD the register allocation is held to.  16-bit stacks leave LDS room for four workgroups per
// CU, and 128 VGPRs (one spilled dword) for the fourth wave per SIMD pay: 4K 0.92 -> 0.79 ms per frame with
// frames in flight; five waves (96 VGPRs, 42 spilled) lose again (profiles/r02/tri_occ2.log).
// WAVES: one wave per workgroup -- a workgroup's LDS and wave slots come free as soon as its own tile is done
// (1 / 2 / 4 / 8 waves: 0.545 / 0.571 / 0.603 / 0.624 ms for the 1344x846 frame one at a time, 0.769 / 0.765 /
// 0.792 / 0.883 ms per 4K frame in flight; profiles/r02/tri_waves.log).
// the most tiles a work list renders in parts: one in kOrderDiv4, at most kOrderCap4 (order_hist caps its selection there, launch_tri
// sizes the grid for it)
static constexpr uint32_t kOrderCap4 = 1024u, kOrderDiv4 = 16u;
static uint32_t order_cap4() {
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_CAP4")) return std::min(8192u, (uint32_t)atoi(e));
#endif
    return kOrderCap4;
}
static uint32_t order_div4() {
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_DIV4")) return std::max(2u, (uint32_t)atoi(e));
#endif
    return kOrderDiv4;
}
template <typename STK, int OCC, bool PACKED, int WAVES = 1, bool PAIRS = false, bool P16 = false, int SMALL = 0>
static void launch_tri(const RtFrameArgs& a, const RtTriScene& t0, int heatmap, hipStream_t s) {
    const dim3 grid((a.W + 8u * WAVES - 1u) / (8u * WAVES), a.n_local_tiles, 1);
    const uint32_t n_tiles = grid.x * grid.y;       // trace_triangles decodes the tile itself; with a work list: room for the quarters
    RtTriScene t = t0;
    t.xcd_rows = (!t.tile_order && !heatmap) ? 1u : 0u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_XCD")) t.xcd_rows = t.xcd_rows && atoi(e) != 0;
#endif
    const uint32_t padded = grid.x * ((grid.y + 7u) & ~7u);
    // (a part of trace_roles takes 0.6-0.7 of the time the same part takes in trace_triangles, and its scaled time must say the same
    // about the whole tile: x 2 a quarter, x 4 a sixteenth, in eighths.  Swept on the reference's scene, nothing moving / its mesh
    // turning: 12 / 24: 0.403 / 0.456 ms per awaited frame, 16 / 32: 0.356 / 0.455, 20 / 40: 0.356 / 0.464, 24 / 48: 0.361 / 0.476,
    // 32 / 64: 0.362 / 0.500; profiles/r05/tri_roles_loop.log)
    t.cost_mul4 = 16u; t.cost_mul16 = 32u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_ROLES")) t.roles = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_CM4")) t.cost_mul4 = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_CM16")) t.cost_mul16 = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_PRIO")) t.prio = (uint32_t)atoi(e);
    const size_t lds_pad = getenv("RT355_TRI_LDSPAD") ? (size_t)atoi(getenv("RT355_TRI_LDSPAD")) : 0u;
#else
    const size_t lds_pad = 0u;
#endif
    const dim3 line(t.tile_order ? n_tiles + 3u * std::min(n_tiles / order_div4(), order_cap4()) + 15u * std::min(n_tiles / 64u, 256u) : (t.xcd_rows ? padded : n_tiles), 1, 1);   // order_tiles: at most that many tiles in quarters / sixteenths
    if constexpr (WAVES == 1 && PAIRS && P16 && (SMALL == 1 || SMALL == 3)) {
        // an awaited frame with a work list: the form whose split tiles' idle lanes trace ahead (trace_roles)
        if (t.tile_order && !heatmap && t.roles) {
            g_rt_kernel_id = RT_KID_TRIANGLES_ROLES;
            // (82 registers: twenty workgroups per CU where the LDS would hold twenty-one; held to 80 -- __launch_bounds__(64, 6) -- the
            // kernel is 3 % slower, profiles/r05/tri_roles.log)
            if (a.sky_flat) hipLaunchKernelGGL((rtk::trace_roles<STK, OCC, true, SMALL>), line, dim3(64), lds_pad, s, a, t);
            else            hipLaunchKernelGGL((rtk::trace_roles<STK, OCC, false, SMALL>), line, dim3(64), lds_pad, s, a, t);
            return;
        }
    }
    if (heatmap) hipLaunchKernelGGL((rtk::heatmap_triangles<WAVES, STK, PACKED>), grid, dim3(64 * WAVES), 0, s, a, t);
    else if (a.sky_flat) hipLaunchKernelGGL((rtk::trace_triangles<WAVES, STK, OCC, true, PACKED, PAIRS, P16, SMALL>), line, dim3(64 * WAVES), lds_pad, s, a, t);
    else                 hipLaunchKernelGGL((rtk::trace_triangles<WAVES, STK, OCC, false, PACKED, PAIRS, P16, SMALL>), line, dim3(64 * WAVES), lds_pad, s, a, t);
}

uint32_t rt_order_scan_words(void) { return 264u; }

// order_hist (with the frame's epilogue in its first workgroup when `counters` is given: the caller records the frame's end event
// behind it) and order_scatter, as two calls
hipError_t rt_launch_order_hist(uint32_t* cost, uint32_t* scan, uint32_t* order, uint32_t n_tiles, uint32_t wave_slots,
                                unsigned long long* counters, unsigned long long* host, uint32_t words, unsigned long long* split_out, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    // Sixteenths from twice the throughput time on, at most 64 tiles (1 / 2 / 4 / 8 times: REF 0.459 / 0.449 / 0.537 / 0.539 ms one at a
    // time, TRI 0.255 / 0.256 / 0.291 / 0.372; 64 against 256 tiles: REF 0.449 against 0.465; a third level of single pixels
    // changes nothing: profiles/r04/tri_split_sweep16.log, tri_split_sweep64.log; again on round 5's kernel: profiles/r05/tri_split_sweep5.log).
    // launch_tri sizes the grid for 256.
    uint32_t mult16 = 4u, cap16 = 64u, mult4 = 1u, multw = 4u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_MULT4")) mult4 = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_MULT16")) mult16 = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_CAP16")) cap16 = std::min(256u, (uint32_t)atoi(e));
    if (const char* e = getenv("RT355_TRI_SLOTS")) wave_slots = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_MULTW")) multw = (uint32_t)atoi(e);
#endif
    const uint32_t per = rtk::kOrderBlock * rtk::kOrderPerThread, blocks = (n_tiles + per - 1u) / per;
    hipLaunchKernelGGL(rtk::order_hist, dim3(blocks), dim3(rtk::kOrderBlock), 0, s, cost, scan, order, n_tiles, wave_slots, mult16, cap16, mult4,
                       counters, host, words, multw, split_out, order_cap4(), order_div4());
    return hipGetLastError();
}
hipError_t rt_launch_order_scatter(uint32_t* cost, uint32_t* scan, uint32_t* order, uint32_t n_tiles, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    const uint32_t per = rtk::kOrderBlock * rtk::kOrderPerThread, blocks = (n_tiles + per - 1u) / per;
    hipLaunchKernelGGL(rtk::order_scatter, dim3(blocks), dim3(rtk::kOrderBlock), 0, s, cost, scan, order, n_tiles);
    return hipGetLastError();
}

static bool tri_p16(const RtTriScene& t) {
    // the relinked pair records (rt_api.hip provides them when the scene fits: every instance staged, 16-bit fields)
    bool p16 = t.p16_ok != 0u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_P16")) p16 = p16 && atoi(e) != 0;
#endif
    return p16;
}

int rt_tri_stack_form(const RtTriScene& t, int heatmap) {
    if (!(t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kWideBlas && tri_p16(t))) return 0;
    // the small forms stage one lookup entry per staged instance record and read nothing else of the per-frame buffers: a lookup
    // table longer than the instance list (entries a leaf could name that are not staged) keeps the twenty-slot form
    if (t.n_blas_lookup > t.n_blas) return 0;
    // Six waves per SIMD (the tiny form) for a caller that keeps frames in flight -- throughput: REF 0.215 -> 0.208 ms per frame,
    // TRI4K 0.542 -> 0.517 --, five (the small form) for one that awaits every frame: such a frame is as long as its longest
    // waves, and those run faster in less company (REF 0.37 against 0.42-0.53 ms; profiles/r05/tri_forms.log).
    uint32_t small = t.tlas_small;
    if (t.n_blas > rtk::kLdsBlas) small = small != 0u ? 4u : 0u;          // 13-16 instances: the form that stages sixteen records (every tree that fits a smaller form fits its 8 levels / 32 nodes)
    if (small == 2u && !t.in_flight) small = 1u;         // (2 implies 1: three levels within 8 nodes are four within 16)
#ifdef RT_TRI_DEV_ENV
    // RT355_TRI_FORM=k: the form to measure, where the frame's tree qualifies for it (a tree good for 2 is good for 1 and 3, one
    // good for 1 is good for 3; 0 is always possible).  (RT355_TRI_SMALL of the earlier logs: the same for 0 / 1 / 2.)
    const char* e = getenv("RT355_TRI_FORM") ? getenv("RT355_TRI_FORM") : getenv("RT355_TRI_SMALL");
    if (e) {
        const uint32_t want = (uint32_t)atoi(e), have = t.tlas_small;
        const bool many = t.n_blas > rtk::kLdsBlas;      // (13-16 instances: 4 or 0 only)
        if (want == 0u || (have != 0u && want == 4u) || (!many && (have == 2u || (have == 1u && want != 2u) || (have == 3u && want == 3u)))) small = want;
    }
#endif
    return (int)small;
}

hipError_t rt_launch_triangles(const RtFrameArgs& a, const RtTriScene& t, int heatmap, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    g_rt_kernel_id = heatmap ? RT_KID_HEATMAP : RT_KID_TRIANGLES;
    const bool p16 = tri_p16(t);
    const uint32_t small = t.form;          // rt_tri_stack_form's answer, which the caller has acted on (t.inst)
    g_rt_tri_form = (int)small;
    if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas && p16 && small == 2u) launch_tri<uint16_t, 6, true, 1, true, true, 2>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas && p16 && small == 1u) launch_tri<uint16_t, 5, true, 1, true, true, 1>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas && p16 && small == 3u) launch_tri<uint16_t, 5, true, 1, true, true, 3>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kWideBlas && p16 && small == 4u) launch_tri<uint16_t, 5, true, 1, true, true, 4>(a, t, heatmap, s);
#ifdef RT_TRI_DEV_ENV
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas && p16 && getenv("RT355_TRI_P16OCC4")) launch_tri<uint16_t, 4, true, 1, true, true>(a, t, heatmap, s);
#endif
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas && p16) launch_tri<uint16_t, 5, true, 1, true, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas) launch_tri<uint16_t, 4, true, 1, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok) launch_tri<uint16_t, 4, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u)           launch_tri<uint16_t, 4, false>(a, t, heatmap, s);
    else                                    launch_tri<uint32_t, 3, false>(a, t, heatmap, s);      // 44 KB of stacks: three workgroups per CU
    return hipGetLastError();
}
