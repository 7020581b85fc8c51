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
// FLAT: compiled for a one-colour 1x1 sky (A.sky_flat) -- no cube filtering code inside the traversal's register budget.
template <int WAVES, typename STK, int OCC, bool FLAT, bool PACKED, bool PAIRS = false, bool P16 = false>
__global__ __launch_bounds__(64 * WAVES, OCC) void trace_triangles(const RtFrameArgs A, const RtTriScene T) {
    typedef typename std::conditional<PACKED && !P16, uint32_t, STK>::type BSTK;
    __shared__ STK tstacks[kStack * 64 * WAVES];
    __shared__ BSTK bstacks[kStack * 64 * WAVES];
    STK* tstack = tstacks + threadIdx.x;
    BSTK* bstack = bstacks + threadIdx.x;
    constexpr uint32_t stride = 64 * WAVES;
    __shared__ float4 s_nodes[2 * kLdsNodes];
    __shared__ float s_blas[20 * kLdsBlas];
    const TriLds L = stage_head<WAVES>(T, s_nodes, s_blas);

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
    uint32_t nrays = 0;
    float dummy = 0.0f;
    float dist = 0.0f;
    v3 color = V(1.0f, 1.0f, 1.0f);
    v3 ro = sc.cameraPos, rd = primary_dir(A, sc, x, y);
    float affect = 1.0f, sum = 0.0f;
    for (uint32_t bounce = 0; bounce < sc.bounces; ++bounce) {                       // RK:113
        const TriHit h = trace_tlas<false, STK, PACKED, PAIRS, P16>(T, L, ro, rd, tstack, bstack, stride, dummy); // RK:114
        ++nrays;
        const bool hit = h.tri >= 0;
        if (bounce == 0) dist = hit ? h.t : 0.0f;                                    // RK:116-118
        const float next = affect + sum;                                             // RK:120
        if (!hit) {                                                                  // RK:122-126
            const v3 sky = scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, rd));
            color = divs(add(scale(sum, color), scale(affect, sky)), next);
            break;
        }
        const v3 normal = hit_normal(T, h);
        const int tri = h.tri;
        const float hu = h.u, hv = h.v;
        ro = add(ro, scale(h.t, rd));                                                // RK:129
        rd = normalize(reflect(rd, normal));                                         // RK:130
        // RK:146-166
        const v3 sdir = normalize(sub(ro, sc.lightPos));
        const float distance = length(sdir);
        const TriHit sh = trace_tlas<false, STK, PACKED, PAIRS, P16>(T, L, sc.lightPos, sdir, tstack, bstack, stride, dummy);   // RK:153
        ++nrays;
        const float intensity = light_term(sc, ro, normal, sdir, distance, sh.tri >= 0, sh.t);
        const Albedo s = hit_albedo(T, tri, hu, hv);
        const v3 diffuseColor = scale(s.w, s.rgb);                                   // RK:133
        const v3 samplerColor = scale(1.0f - s.w, tex2d_sample(T, s.u, s.v));        // RK:134
        const v3 blended = scale(intensity, add(diffuseColor, samplerColor));        // RK:135
        color = divs(add(scale(sum, color), scale(affect, blended)), next);          // RK:136
        affect = affect / 2.0f;                                                      // RK:139
        sum = next;                                                                  // RK:140
    }
    const uint32_t opix = (by * 8u + row) * A.W + x;
    // the fog colour is the sky along the primary ray (RK:93-96): its direction is formed again here rather than
    // carried through both traversals
    reinterpret_cast<uint32_t*>(A.out)[opix] =
        compose_pixel_sky(scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, primary_dir(A, sc, x, y))), color, dist);   // RK:91-98
    count_rays(A.rays, nrays);
    // What a tile leaves for the next frame's work list (10 ns ticks) must not depend on how it was rendered, or the list chases
    // its own tail: a whole tile leaves its time; the parts of a split tile leave the LONGEST of theirs, times what a wave of 64
    // diverging lanes takes longer than its slowest sixteenth (3) or quarter (1.5) alone.  Summed, a split tile looked several times
    // as long as it is and kept its place in the head of the list whatever the other tiles did: the four streams' lists settled in
    // different selections (the reference's scene one frame at a time: 0.43 and 0.49 ms alternating with the streams); the plain
    // maximum made it look short, and every list alternated between splitting a tile and not (0.45 / 0.55).  With the factors:
    // 0.385-0.395 ms, every frame, every stream (profiles/r04/tri_cost_series.log).
    if (T.tile_cost && threadIdx.x == 0u) {
        const uint32_t dt = (uint32_t)(wall_clock64() - clk0);
        atomicMax(&T.tile_cost[tile], part >= 16u ? 3u * dt : (part < 4u ? dt + (dt >> 1) : dt));
    }
}

// ---- the order of the next frame's tiles ------------------------------------------------------------------
// The frame is a grid of independent tiles, one wave each, whose costs differ by more than an order of magnitude (a sky tile:
// one ray per pixel; a tile between two mirrors: 2 x bounces dependent traversals by 64 diverging lanes), and the hardware
// starts them in index order.  Measured with the kernel's own clock (tools/tile_cost_probe.py, profiles/r03/tile_cost.log):
// mean tile 21 us, one in a hundred 156 us, the longest 498 us -- and the 1344x846 frame rendered on its own (the
// reference's await-each-frame loop) took 504 us: a frame is as long as its longest tile, however empty the chip.  So
//   * every tile leaves the time it took (tile_cost; the parts of a split tile: the longest of theirs, scaled -- see the kernel's last lines);
//   * one workgroup turns the costs into the NEXT frame's work list on the same stream: tiles longest first (a counting sort
//     over quarter-octave classes), and the tiles longer than half the frame's throughput time -- sum of all costs / wave
//     slots / 2 --, at most one in sixteen and 1024 (a quarter-wave for every wave slot of the chip), as four 4x4 quarters each: a quarter of the lanes diverge a quarter as much, and
//     the frame's longest wave shrinks accordingly; round 4: the tiles longer than TWICE the throughput time, at most one in 64
//     and 64, as sixteen 2x2 blocks (four lanes each) -- a frame on its own leaves the wave slots for them idle anyway.
//     order[0] = tiles in quarters, order[1] = tiles in sixteenths (the head of the list), order[2...] = the permutation.
// The picture does not depend on any of it: a pixel is rendered by the same code whatever its turn and company; any
// array of costs yields a permutation and a split count within the grid's bound.
__global__ __launch_bounds__(1024) void order_tiles(uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n, uint32_t wave_slots,
                                                    uint32_t mult16, uint32_t cap16) {
    __shared__ uint32_t bin[128];
    __shared__ unsigned long long total;
    if (threadIdx.x < 128u) bin[threadIdx.x] = 0u;
    if (threadIdx.x == 0u) total = 0ull;
    __syncthreads();
    auto cls = [](uint32_t c) -> uint32_t {        // quarter-octave class of a tick count: 0 ... 123, monotone
        if (c < 4u) return c;
        const uint32_t e = 31u - (uint32_t)__clz(c);
        return 4u * e + ((c >> (e - 2u)) & 3u) - 4u;
    };
    unsigned long long mine = 0ull;
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) { const uint32_t c = cost[i]; mine += c; atomicAdd(&bin[cls(c)], 1u); }
    atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0u) {                       // exclusive prefix over the classes, longest class first
        const unsigned long long thr = total / (2ull * (wave_slots ? wave_slots : 1u));
        auto cls_of = [&](unsigned long long v) { return cls(v > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)v); };
        const uint32_t kt = cls_of(thr);
        const uint32_t kw = cls_of(4ull * thr);           // twice the throughput time
        const uint32_t k16 = cls_of((unsigned long long)mult16 * thr);          // mult16 / 2 times the throughput time: as sixteenths
        uint32_t acc = 0u, split = 0u, split16 = 0u;
        bool pays = false;                         // some tile takes more than twice the throughput time: the frame waits for it.
        for (int k = 127; k >= 0; --k) {           // (Otherwise quarters only add waves: 4K, 0.77 -> 0.81 ms with them.)
            const uint32_t v = bin[k];
            bin[k] = acc; acc += v;
            if (k > (int)kt) split = acc;          // tiles of the classes above the threshold's: all longer than it
            if (k > (int)k16) split16 = acc;
            if (k > (int)kw && v) pays = true;
        }
        if (!pays) split = split16 = 0u;
        const uint32_t most = n / 16u < 1024u ? n / 16u : 1024u, most16 = n / 64u < cap16 ? n / 64u : cap16;      // rt_tri_grid
        if (split16 > most16) split16 = most16;
        if (split > most) split = most;
        if (split < split16) split = split16;
        order[0] = split - split16;                // tiles rendered as four quarters ...
        order[1] = split16;                        // ... behind the tiles rendered as sixteen 2x2 blocks: the head of the list
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        order[2u + atomicAdd(&bin[cls(cost[i])], 1u)] = i;
        cost[i] = 0u;                              // the next frame adds its times up from zero
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
template <typename STK, int OCC, bool PACKED, int WAVES = 1, bool PAIRS = false, bool P16 = false>
static void launch_tri(const RtFrameArgs& a, const RtTriScene& t0, int heatmap, hipStream_t s) {
    const dim3 grid((a.W + 8u * WAVES - 1u) / (8u * WAVES), a.n_local_tiles, 1);
    const uint32_t n_tiles = grid.x * grid.y;       // trace_triangles decodes the tile itself; with a work list: room for the quarters
    RtTriScene t = t0;
    t.xcd_rows = (!t.tile_order && !heatmap) ? 1u : 0u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_XCD")) t.xcd_rows = t.xcd_rows && atoi(e) != 0;
#endif
    const uint32_t padded = grid.x * ((grid.y + 7u) & ~7u);
    const dim3 line(t.tile_order ? n_tiles + 3u * std::min(n_tiles / 16u, 1024u) + 15u * std::min(n_tiles / 64u, 256u) : (t.xcd_rows ? padded : n_tiles), 1, 1);   // order_tiles: at most that many tiles in quarters / sixteenths
    if (heatmap) hipLaunchKernelGGL((rtk::heatmap_triangles<WAVES, STK, PACKED>), grid, dim3(64 * WAVES), 0, s, a, t);
    else if (a.sky_flat) hipLaunchKernelGGL((rtk::trace_triangles<WAVES, STK, OCC, true, PACKED, PAIRS, P16>), line, dim3(64 * WAVES), 0, s, a, t);
    else                 hipLaunchKernelGGL((rtk::trace_triangles<WAVES, STK, OCC, false, PACKED, PAIRS, P16>), line, dim3(64 * WAVES), 0, s, a, t);
}

hipError_t rt_launch_order_tiles(uint32_t* cost, uint32_t* order, uint32_t n_tiles, uint32_t wave_slots, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    // Sixteenths from twice the throughput time on, at most 64 tiles (1 / 2 / 4 / 8 times: REF 0.459 / 0.449 / 0.537 / 0.539 ms one at a
    // time, TRI 0.255 / 0.256 / 0.291 / 0.372; 64 against 256 tiles: REF 0.449 against 0.465; a third level of single pixels
    // changes nothing: profiles/r04/tri_split_sweep16.log, tri_split_sweep64.log).  launch_tri sizes the grid for 256.
    uint32_t mult16 = 4u, cap16 = 64u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_MULT16")) mult16 = (uint32_t)atoi(e);
    if (const char* e = getenv("RT355_TRI_CAP16")) cap16 = std::min(256u, (uint32_t)atoi(e));
#endif
    hipLaunchKernelGGL(rtk::order_tiles, dim3(1), dim3(1024), 0, s, cost, order, n_tiles, wave_slots, mult16, cap16);
    return hipGetLastError();
}

hipError_t rt_launch_triangles(const RtFrameArgs& a, const RtTriScene& t, int heatmap, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    g_rt_kernel_id = heatmap ? RT_KID_HEATMAP : RT_KID_TRIANGLES;
    // the relinked pair records (rt_api.hip provides them when the scene fits: every instance staged, 16-bit fields)
    bool p16 = t.p16_ok != 0u;
#ifdef RT_TRI_DEV_ENV
    if (const char* e = getenv("RT355_TRI_P16")) p16 = p16 && atoi(e) != 0;
#endif
    if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas && p16) launch_tri<uint16_t, 5, true, 1, true, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok && t.pairs && !heatmap && t.n_blas <= rtk::kLdsBlas) launch_tri<uint16_t, 4, true, 1, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u && t.packed_ok) launch_tri<uint16_t, 4, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u)           launch_tri<uint16_t, 4, false>(a, t, heatmap, s);
    else                                    launch_tri<uint32_t, 3, false>(a, t, heatmap, s);      // 44 KB of stacks: three workgroups per CU
    return hipGetLastError();
}
