// rt_flow.hip -- the reference's live scene type (triangles behind a TLAS / BLAS hierarchy, RK:168-410) rendered by
// PERSISTENT waves: every lane carries one pixel's path through a small state machine and takes the next pixel from the
// frame's cursor when the path ends; the nested walks of the reference (traceTLAS around traceBLAS around the triangle
// loop) are unrolled into single STEPS, and per trip a wave executes ONE kind of step -- the kind most of its lanes wait
// for -- so that lanes at different bounces, instances and tree levels share every instruction they can.
//
// Why.  rt_triangles.hip keeps the WGSL's shape -- a pixel per lane for the pixel's whole life, a tile per wave -- and
// measured (profiles/r03): a frame on its own takes as long as its longest tile (498 of 504 us), a wave waits on memory
// 47 % of its cycles (a dependent global load per step, four waves per SIMD because two 20-entry stacks per lane fill the
// LDS), and the paths of one tile diverge: 2.3e8 wave-instructions per frame of the reference's scene where its 44 M steps,
// fully packed, need 7e7.  Here
//   * pixels come from a cursor: no wave idles while another still has a tile's worth of paths (the frame ends with its
//     longest PATH, 459 steps on the reference's scene, not its longest tile);
//   * the BLAS trees are read from the library's relinked copy (rt_flow_build.h: 64-byte child PAIRS, most-visited first),
//     whose head the workgroup stages in LDS -- 89-94 % of all inner-node steps of the reference's scene become LDS reads;
//   * the per-lane stacks keep their first 8 (BLAS) / 2 (TLAS) slots in LDS and the rest in a per-wave overflow area in
//     global memory (0.2 % of all pushes on the reference's scene go deeper than 8): 2.3 KB of LDS per wave instead of 7.7;
//   * a step is a block of code of its own -- TLAS (inner node or instance set-up), BNODE (one child pair), TRI (one
//     triangle), DONE (a ray is complete: shade, next ray / next pixel) -- chosen per trip by vote (tools/tri_sched_sim.c
//     is the model: executing every block that has a lane costs 2.5 x the instructions of executing the fullest one).
//
// Bit-exactness is a scheduling argument: a lane's own sequence of operations on its own state is the sequential program
// of RK:101-166 / RK:168-341 statement for statement (the helpers are rt_tri_device.h's, shared with rt_triangles.hip);
// which lane holds which pixel, and when it advances, touches no value.  The relinked copy holds the reference's own box
// corners; indices and counts are converted (u32(f32), clamp to the last element) when it is built instead of per step.
//
// Scenes this form takes (rt_api.hip: flow_ok): up to 16 instances and lookup entries (they travel with the frame), node
// buffer and lookup table within 16 bits, a relinked copy that covers the frame's roots.  Everything else, and the
// heatmap, stays with rt_triangles.hip.
#include "rt_tri_device.h"
#include "rt_flow_types.h"

namespace rtk {

enum : uint32_t { ST_IDLE = 0u, ST_TLAS = 1u, ST_BNODE = 2u, ST_TRI = 3u, ST_RDONE = 4u, ST_SDONE = 5u };

#ifdef RT_FLOW_COUNT
// counting build: [2k] runs of block k, [2k+1] lanes it advanced (k: 0 TLAS, 1 BNODE, 2 TRI, 3 DONE), [8] trips, [9] BNODE steps served from LDS
__device__ unsigned long long g_flow_count[16];
#endif

// LDS is addressed explicitly (address space 3, byte addresses): a pointer that may name LDS or global memory -- "the pair
// record, wherever it lives", "the stack slot, LDS or overflow" -- makes the compiler emit FLAT loads for both, which
// serialise through the address-aperture check; with typed pointers the two sides are ds_read and global_load under their
// own exec masks.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f4v* lds_f4;
typedef __attribute__((address_space(3))) const float* lds_f32;
typedef __attribute__((address_space(3))) uint32_t* lds_u32;
typedef __attribute__((address_space(3))) uint16_t* lds_u16;
__device__ __forceinline__ float4 lds_read4(uint32_t a) { const f4v v = *(lds_f4)(uintptr_t)a; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float lds_read1(uint32_t a) { return *(lds_f32)(uintptr_t)a; }

// node i of the node buffer: from the staged head (LDS byte address a_nodes, n_head nodes) or from global memory; RK:175 etc.
__device__ __forceinline__ NodeR flow_node(const RtTriScene& T, uint32_t a_nodes, uint32_t n_head, uint32_t i) {
    if (i >= T.n_nodes) i = T.n_nodes - 1u;
    float4 a, b;
    if (i < n_head) { a = lds_read4(a_nodes + 32u * i); b = lds_read4(a_nodes + 32u * i + 16u); }
    else { a = T.nodes[2u * (size_t)i]; b = T.nodes[2u * (size_t)i + 1u]; }
    NodeR n;
    n.lo = V(a.x, a.y, a.z); n.left = a.w;
    n.hi = V(b.x, b.y, b.z); n.count = b.w;
    return n;
}

__device__ __forceinline__ uint32_t pack_node(const NodeR& n) {
    const uint32_t c = u32f(n.count), l = u32f(n.left);
    return ((c < 0xFFFFu ? c : 0xFFFFu) << 16) | (l < 0xFFFFu ? l : 0xFFFFu);
}

template <int WAVES, bool FLAT>
__global__ __launch_bounds__(64 * WAVES, 4) void trace_flow(const RtFrameArgs A, const RtTriScene T, const RtFlowArgs F) {
    extern __shared__ float4 lds[];
    constexpr uint32_t stride = 64u * WAVES;
    float4* const s_nodes = lds;                                                        // [2 * kLdsNodes]
    float* const s_blas = reinterpret_cast<float*>(lds + 2u * kLdsNodes);               // [20 * kFlowInst]
    uint32_t* const s_bstack = reinterpret_cast<uint32_t*>(s_blas + 20u * kFlowInst);   // [kFlowKB][stride]
    uint16_t* const s_tstack = reinterpret_cast<uint16_t*>(s_bstack + kFlowKB * stride); // [kFlowKT][stride]
    const float4* const s_pairs = reinterpret_cast<const float4*>(s_tstack + kFlowKT * stride);   // [lds_pairs][4]

    // ---- staging: TLAS head, instance records (+ root meta, + lookup entry), the head of the pair array ----
    TriLds L;
    L.n_nodes = T.n_nodes < kLdsNodes ? T.n_nodes : kLdsNodes;
    L.n_blas = T.n_blas < kFlowInst ? T.n_blas : kFlowInst;
    L.n_lookup = T.n_blas_lookup < L.n_blas ? T.n_blas_lookup : L.n_blas;
    for (uint32_t i = threadIdx.x; i < 2u * L.n_nodes; i += stride) s_nodes[i] = T.nodes[i];
    for (uint32_t i = threadIdx.x; i < 20u * L.n_blas; i += stride) {
        const uint32_t r = i / 20u, k = i % 20u;
        float v = T.blas[i];
        if (k == 17u) v = __uint_as_float(F.root_meta[r]);
        if (k == 19u && r < L.n_lookup) v = T.blas_lookup[r];
        s_blas[i] = v;
    }
    {
        // quarter-major in LDS ([quarter][pair], 16-byte stride within a quarter): lanes reading the same quarter of DIFFERENT
        // records -- what a step does -- spread over all banks; record-major, every lane's 16 bytes would start at bank 0 or 16
        float4* const dst = const_cast<float4*>(s_pairs);
        for (uint32_t i = threadIdx.x; i < 4u * F.lds_pairs; i += stride) dst[(i & 3u) * F.lds_pairs + (i >> 2)] = F.pairs[i];
    }
    __syncthreads();
    L.nodes = s_nodes; L.blas = s_blas;
    // LDS byte addresses of the staged arrays (the low word of a pointer into the LDS aperture is the LDS address)
    const uint32_t a_nodes = (uint32_t)(uintptr_t)s_nodes, a_blas = (uint32_t)(uintptr_t)s_blas, a_pairs = (uint32_t)(uintptr_t)s_pairs;
    const uint32_t a_bst = (uint32_t)(uintptr_t)(s_bstack + threadIdx.x), a_tst = (uint32_t)(uintptr_t)(s_tstack + threadIdx.x);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // overflow stacks of this wave: [kStack - kFlowKB][64] words, then [kStack - kFlowKT][64] half words
    uint32_t* const ovf_b = F.ovf + (size_t)(blockIdx.x * WAVES + wave) * kFlowOvfWords + lane;
    uint16_t* const ovf_t = reinterpret_cast<uint16_t*>(F.ovf + (size_t)(blockIdx.x * WAVES + wave) * kFlowOvfWords + (kStack - kFlowKB) * 64u) + lane;

    const Scene sc = unpack_scene(A);
    const uint32_t tiles_x = (A.W + 7u) / 8u;
    const uint32_t total = A.n_local_tiles * tiles_x * 64u;      // pixel slots, tile-major
    uint32_t cur = 0, end = 0;                                   // wave-uniform chunk of the cursor
    uint32_t chunk_ty = 0, chunk_tx = 0, chunk_first = ~0u;
    bool exhausted = false;
    const uint32_t flow_grid = gridDim.x * (uint32_t)WAVES, flow_wave = blockIdx.x * (uint32_t)WAVES + wave;
    const uint32_t flow_rounds = (uint32_t)(((unsigned long long)((total >> 6) / flow_grid) * F.static_pct) / 100ull);
    uint32_t flow_k = 0;
    // the ray every path starts with enters the TLAS at node 0 (RK:175): its (count, left), once per wave
    const uint32_t root_tnode = pack_node(flow_node(T, a_nodes, L.n_nodes, 0u));

    // ---- per-lane state ----
    uint32_t st = ST_IDLE;
    uint32_t opix = 0, xy = 0, bounce = 0, nrays = 0;
    bool shadow = false;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), color = V(1, 1, 1);
    v3 normal = V(0, 0, 1), sdir = V(0, 0, 1);
    float dist = 0.0f, affect = 1.0f, sum = 0.0f;
    int ptri = -1; float pu = 0.0f, pv = 0.0f;                   // the reflection ray's hit, kept for the albedo (RK:133-134)
    // current ray: nearest hit so far (RK:172), and which triangle / instance it is on
    float nearest = 9999.0f, hu = 0.0f, hv = 0.0f;
    int htri = -1, hblas = -1;
    // TLAS level (RK:168-244)
    uint32_t tnode = 0, ti = 0, sp_t = 0;
    // BLAS level (RK:246-332)
    v3 oo = V(0, 0, 0), od = V(0, 0, 1), inv = V(0, 0, 0);
    float bnear = 9999.0f;
    uint32_t bnode = 0, tk = 0, sp_b = 0, cbi = 0;

    if (sc.bounces == 0u) {
        // RK:113: the loop body never runs -- colour (1,1,1), dist 0 (RK:102-103), composed with the fog colour
        for (;;) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.qctrl[2], 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= total) break;
            const uint32_t ty = (base >> 6) / tiles_x, tx = (base >> 6) - ty * tiles_x;
            const uint32_t x = tx * 8u + (lane & 7u), row = lane >> 3;
            const uint32_t y = (A.tile_first + ty * A.tile_step) * 8u + row;
            if (x < A.W && y < A.H) {
                const uint32_t o = (ty * 8u + row) * A.W + x;
                if (FLAT) reinterpret_cast<uint32_t*>(A.out)[o] =
                    compose_pixel_sky(scale(sc.minIntensity, cube_sample<1>(A, primary_dir(A, sc, x, y))), V(1.0f, 1.0f, 1.0f), 0.0f);
                else A.fin[2u * o] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);      // sky_resolve composes it with the fog colour
            }
        }
        return;
    }

#ifdef RT_FLOW_COUNT
    uint32_t c_runs[4] = {0, 0, 0, 0}, c_lanes[4] = {0, 0, 0, 0}, c_trips = 0, c_lds = 0;
#endif

    // one hybrid-stack access: slot s of this lane, LDS below the split, the wave's overflow area above
    auto bpush = [&](uint32_t s, uint32_t v) {
        if (s < kFlowKB) *(lds_u32)(uintptr_t)(a_bst + s * (stride * 4u)) = v;
        else ovf_b[(s - kFlowKB) * 64u] = v;
    };
    auto bread = [&](uint32_t s) -> uint32_t {
        uint32_t v;
        if (s < kFlowKB) v = *(lds_u32)(uintptr_t)(a_bst + s * (stride * 4u));
        else v = ovf_b[(s - kFlowKB) * 64u];
        return v;
    };
    auto tpush = [&](uint32_t s, uint32_t v) {
        if (s < kFlowKT) *(lds_u16)(uintptr_t)(a_tst + s * (stride * 2u)) = (uint16_t)v;
        else ovf_t[(s - kFlowKT) * 64u] = (uint16_t)v;
    };
    auto tread = [&](uint32_t s) -> uint32_t {
        uint32_t v;
        if (s < kFlowKT) v = (uint32_t)*(lds_u16)(uintptr_t)(a_tst + s * (stride * 2u));
        else v = (uint32_t)ovf_t[(s - kFlowKT) * 64u];
        return v;
    };

    // a new ray of this lane's path enters the TLAS (RK:170-178)
    auto start_ray = [&]() {
        tnode = root_tnode; ti = 0u; sp_t = 0u;
        nearest = 9999.0f; htri = -1; hblas = -1;
        st = ST_TLAS;
    };
    // RK:324-329 / RK:294-298: the BLAS walk needs its next node from the stack, or is over (RK:227-229, then the TLAS leaf's next instance)
    auto blas_pop = [&]() {
        if (sp_b == 0u) {
            nearest = bnear < nearest ? bnear : nearest;
            ti += 1u;
            st = ST_TLAS;
        } else {
            sp_b -= 1u;
            bnode = bread(sclamp(sp_b));
            tk = 0u;
            st = (bnode >> 16) == 0u ? ST_BNODE : ST_TRI;
        }
    };

    for (;;) {
        // ---- which block runs this trip ----
        const uint32_t c_l = (uint32_t)__popcll(__ballot(st == ST_TLAS));
        const uint32_t c_b = (uint32_t)__popcll(__ballot(st == ST_BNODE));
        const uint32_t c_t = (uint32_t)__popcll(__ballot(st == ST_TRI));
        const uint32_t c_d = (uint32_t)__popcll(__ballot(st >= ST_RDONE)) + (exhausted ? 0u : (uint32_t)__popcll(__ballot(st == ST_IDLE)));
        const uint32_t walkers = c_l + c_b + c_t;
        if (walkers == 0u && c_d == 0u) break;
        uint32_t run;                                       // 0 TLAS, 1 BNODE, 2 TRI, 3 DONE
        if (c_d >= F.thresh || walkers == 0u) run = 3u;
        else if (c_b >= c_t && c_b >= c_l) run = 1u;
        else if (c_t >= c_l) run = 2u;
        else run = 0u;
#ifdef RT_FLOW_COUNT
        ++c_trips; ++c_runs[run];
        c_lanes[run] += run == 0u ? c_l : (run == 1u ? c_b : (run == 2u ? c_t : c_d));
#endif

        if (run == 1u) {
            // ---- BNODE: one step of RK:275-307 on the child pair of the current inner node ----
            if (st == ST_BNODE) {
                const uint32_t p = bnode & 0xFFFFu;
                float4 q0, q1, q2, q3;
                if (p < F.lds_pairs) {
                    const uint32_t a = a_pairs + 16u * p, q = 16u * F.lds_pairs;
                    q0 = lds_read4(a); q1 = lds_read4(a + q); q2 = lds_read4(a + 2u * q); q3 = lds_read4(a + 3u * q);
#ifdef RT_FLOW_COUNT
                    ++c_lds;
#endif
                } else {
                    const float4* g = F.pairs + 4u * (size_t)p;
                    q0 = g[0]; q1 = g[1]; q2 = g[2]; q3 = g[3];
                }
                NodeR c1, c2;
                c1.lo = V(q0.x, q0.y, q0.z); c1.hi = V(q1.x, q1.y, q1.z);
                c2.lo = V(q2.x, q2.y, q2.z); c2.hi = V(q3.x, q3.y, q3.z);
                const uint32_t m1 = __float_as_uint(q0.w), m2 = __float_as_uint(q2.w);
                float d1 = hit_aabb(oo, inv, c1);                       // RK:279
                float d2 = hit_aabb(oo, inv, c2);                       // RK:280
                const bool swap = d1 > d2;                              // RK:283-290
                if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; }
                if (d1 > bnear) {                                       // RK:292
                    blas_pop();
                } else {
                    bnode = swap ? m2 : m1;                             // RK:302
                    if (d2 < bnear) {                                   // RK:303-304 (no overflow guard upstream)
                        bpush(sclamp(sp_b), swap ? m1 : m2);
                        sp_b += 1u;
                    }
                    tk = 0u;
                    st = (bnode >> 16) == 0u ? ST_BNODE : ST_TRI;
                }
            }
        } else if (run == 2u) {
            // ---- TRI: one trip of the leaf loop RK:311-322 ----
            if (st == ST_TRI) {
                const uint32_t count = bnode >> 16, left = bnode & 0xFFFFu;
                uint32_t li = left + tk;
                if (li >= T.n_tri_lookup) li = T.n_tri_lookup - 1u;     // RK:314: the lookup itself is folded into T.corners
                float t, u, v;
                if (hit_triangle(T, li, oo, od, bnear, t, u, v)) {      // RK:312-321
                    bnear = t;
                    hu = u; hv = v; htri = (int)li; hblas = (int)cbi;
                }
                tk += 1u;
                if (tk >= count) blas_pop();                            // RK:324-329
            }
        } else if (run == 0u) {
            // ---- TLAS: RK:179-240 up to and including the set-up of the next instance (RK:246-269).  Inner nodes, exhausted
            // leaves and pops are a few instructions each and are walked through here, in a loop of the wave, until every lane
            // of the block stands before an instance or has completed its ray: the expensive part -- the ray's transform and
            // three divisions -- then runs once, for all of them.
            const v3 o = shadow ? sc.lightPos : ro, d = shadow ? sdir : rd;
            bool nav = st == ST_TLAS;
            while (__ballot(nav) != 0ull) {
                if (nav) {
                    const uint32_t count = tnode >> 16, left = tnode & 0xFFFFu;
                    bool pop = false;
                    if (count == 0u) {                                  // RK:183
                        uint32_t i2 = left + 1u;
                        const NodeR c1 = flow_node(T, a_nodes, L.n_nodes, left), c2 = flow_node(T, a_nodes, L.n_nodes, left + 1u);
                        const v3 winv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                        float d1 = hit_aabb(o, winv, c1);               // RK:186
                        float d2 = hit_aabb(o, winv, c2);               // RK:187
                        const bool swap = d1 > d2;                      // RK:190-196
                        if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; i2 = left; }
                        if (d1 > nearest) {                             // RK:198
                            pop = true;
                        } else {
                            tnode = pack_node(swap ? c2 : c1);          // RK:208
                            ti = 0u;
                            if (d2 < nearest) {                         // RK:209
                                tpush(sclamp(sp_t), i2 < T.n_nodes ? i2 : T.n_nodes - 1u);
                                sp_t += 1u;
                                if (sp_t > kStack) sp_t = kStack - 1u;  // RK:212-214 guards with `>`
                            }
                        }
                    } else if (ti < count) {
                        nav = false;                                    // RK:220: instance ti of this leaf is next
                    } else {
                        pop = true;                                     // RK:233
                    }
                    if (pop) {
                        if (sp_t == 0u) {
                            st = shadow ? ST_SDONE : ST_RDONE;          // the ray is complete
                            nav = false;
                        } else {
                            sp_t -= 1u;
                            tnode = pack_node(flow_node(T, a_nodes, L.n_nodes, tread(sclamp(sp_t))));   // RK:237-238
                            ti = 0u;
                        }
                    }
                }
            }
            if (st == ST_TLAS) {
                const uint32_t left = tnode & 0xFFFFu;
                uint32_t li = ti + left;
                if (li >= T.n_blas_lookup) li = T.n_blas_lookup - 1u;
                float bif;
                if (li < L.n_lookup) bif = lds_read1(a_blas + 80u * li + 76u); else bif = T.blas_lookup[li];   // RK:223
                uint32_t bi = u32f(bif);
                if (bi >= T.n_blas) bi = T.n_blas - 1u;
                float m[18];                                            // mat4 column-major, m[4c + r]; every instance is staged (flow_ok)
                {
                    const uint32_t a = a_blas + 80u * bi;
                    const float4 r0 = lds_read4(a), r1 = lds_read4(a + 16u), r2 = lds_read4(a + 32u), r3 = lds_read4(a + 48u);
                    m[0] = r0.x; m[1] = r0.y; m[2] = r0.z; m[3] = r0.w; m[4] = r1.x; m[5] = r1.y; m[6] = r1.z; m[7] = r1.w;
                    m[8] = r2.x; m[9] = r2.y; m[10] = r2.z; m[11] = r2.w; m[12] = r3.x; m[13] = r3.y; m[14] = r3.z; m[15] = r3.w;
                    m[17] = lds_read1(a + 68u);
                }
                oo = V(((m[0] * o.x + m[4] * o.y) + m[8] * o.z) + m[12] * 1.0f,
                       ((m[1] * o.x + m[5] * o.y) + m[9] * o.z) + m[13] * 1.0f,
                       ((m[2] * o.x + m[6] * o.y) + m[10] * o.z) + m[14] * 1.0f);           // RK:254
                od = V(((m[0] * d.x + m[4] * d.y) + m[8] * d.z) + m[12] * 0.0f,
                       ((m[1] * d.x + m[5] * d.y) + m[9] * d.z) + m[13] * 0.0f,
                       ((m[2] * d.x + m[6] * d.y) + m[10] * d.z) + m[14] * 0.0f);           // RK:255
                inv = V(1.0f / od.x, 1.0f / od.y, 1.0f / od.z);         // RK:396
                bnode = __float_as_uint(m[17]);                         // RK:265: the root's (count, left), relinked
                cbi = bi;
                sp_b = 0u;                                              // RK:267
                bnear = nearest;                                        // RK:269
                tk = 0u;
                st = (bnode >> 16) == 0u ? ST_BNODE : ST_TRI;
            }
        } else {
            // ---- DONE: complete rays are shaded (RK:114-141, RK:146-166), finished pixels stored, idle lanes refilled ----
            bool finished = false, missed = false;
            if (st == ST_RDONE) {
                ++nrays;
                const bool hit = htri >= 0;
                if (bounce == 0u) dist = hit ? nearest : 0.0f;                           // RK:116-118
                if (!hit) {
                    missed = true; finished = true;                                      // RK:122-126, blended below
                } else {
                    TriHit h; h.t = nearest; h.u = hu; h.v = hv; h.tri = htri; h.blas = hblas;
                    normal = hit_normal(T, h);
                    ptri = htri; pu = hu; pv = hv;
                    ro = add(ro, scale(nearest, rd));                                    // RK:129
                    rd = normalize(reflect(rd, normal));                                 // RK:130
                    sdir = normalize(sub(ro, sc.lightPos));                              // RK:147
                    shadow = true;                                                       // RK:153
                    start_ray();
                }
            } else if (st == ST_SDONE) {
                ++nrays;
                const float next = affect + sum;                                         // RK:120
                const float distance = length(sdir);                                     // RK:148
                const float intensity = light_term(sc, ro, normal, sdir, distance, htri >= 0, nearest);
                const Albedo s = hit_albedo(T, ptri, pu, pv);
                const v3 diffuseColor = scale(s.w, s.rgb);                               // RK:133
                const v3 samplerColor = scale(1.0f - s.w, tex2d_sample(T, s.u, s.v));    // RK:134
                const v3 blended = scale(intensity, add(diffuseColor, samplerColor));    // RK:135
                color = divs(add(scale(sum, color), scale(affect, blended)), next);      // RK:136
                affect = affect / 2.0f;                                                  // RK:139
                sum = next;                                                              // RK:140
                ++bounce;
                shadow = false;
                if (bounce >= sc.bounces) finished = true;                               // RK:113
                else start_ray();
            }
            // The sky.  FLAT (one colour): sampled here, along the missing ray (RK:123) and, for the fog colour of a finished pixel,
            // along the primary ray (RK:92).  Textured: not sampled in this kernel at all -- the lane leaves the end-of-path
            // record bvh_pixels leaves (rt_bvh.hip) and sky_resolve filters the cube map pixel per lane, full waves, neighbouring
            // directions: same statements, same values, and no cube filter inside this kernel's register budget.
            if (FLAT) {
                if (missed) {
                    const v3 sky = scale(sc.minIntensity, cube_sample<1>(A, rd));
                    const float next = affect + sum;                                     // RK:120
                    color = divs(add(scale(sum, color), scale(affect, sky)), next);      // RK:124
                }
                if (finished)
                    reinterpret_cast<uint32_t*>(A.out)[opix] =
                        compose_pixel_sky(scale(sc.minIntensity, cube_sample<1>(A, primary_dir(A, sc, xy & 0xFFFFu, xy >> 16))), color, dist);   // RK:91-98
            } else if (finished) {
                // dist is +0 or a hit distance > 0.001: its sign bit is free for the flag
                A.fin[2u * opix] = make_float4(color.x, color.y, color.z, __uint_as_float(__float_as_uint(dist) | (missed ? 0x80000000u : 0u)));
                if (missed && bounce != 0u) A.fin[2u * opix + 1u] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(bounce));
            }
            if (finished) st = ST_IDLE;
            // ---- idle lanes take the next pixels of the frame ----
            uint64_t idle = __ballot(st == ST_IDLE);
            while (idle && !exhausted) {
                if (cur == end) {
                    // the next tile: the first F.static_pct per cent of a wave's tiles are w, w + G, w + 2G ... (G waves in the
                    // grid) without asking; the rest comes from the frame's cursor (atomics on one address queue: see trace_tiles)
                    uint32_t base;
                    if (flow_k < flow_rounds) { base = (flow_wave + flow_k * flow_grid) * 64u; ++flow_k; }
                    else {
                        base = 0;
                        if (lane == 0) base = atomicAdd(&A.qctrl[2], 64u);
                        base = __builtin_amdgcn_readfirstlane(base) + flow_rounds * flow_grid * 64u;
                    }
                    if (base >= total) { exhausted = true; break; }
                    cur = base;
                    end = min(base + 64u, total);
                }
                if ((cur & 63u) == 0u || cur == chunk_first) {       // entering a tile: decode it (wave-uniform)
                    chunk_first = cur;
                    chunk_ty = (cur >> 6) / tiles_x;
                    chunk_tx = (cur >> 6) - chunk_ty * tiles_x;
                }
                const uint32_t tile_end = min(end, (cur & ~63u) + 64u);
                const uint32_t take = min((uint32_t)__popcll(idle), tile_end - cur);
                const uint32_t r = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (st == ST_IDLE && r < take) {
                    const uint32_t l = (cur + r) & 63u;
                    const uint32_t x = chunk_tx * 8u + (l & 7u), row = l >> 3;
                    const uint32_t y = (A.tile_first + chunk_ty * A.tile_step) * 8u + row;
                    if (x < A.W && y < A.H) {                        // RR:445: outside the texture: nothing
                        opix = (chunk_ty * 8u + row) * A.W + x;
                        xy = x | (y << 16);
                        ro = sc.cameraPos; rd = primary_dir(A, sc, x, y);
                        color = V(1.0f, 1.0f, 1.0f); dist = 0.0f;    // RK:102-103
                        affect = 1.0f; sum = 0.0f; bounce = 0u;      // RK:106-107
                        shadow = false;
                        start_ray();
                    }
                }
                cur += take;
                idle = __ballot(st == ST_IDLE);
            }
        }
    }
#ifdef RT_FLOW_COUNT
    if (lane == 0u) {
        for (int k = 0; k < 4; ++k) { atomicAdd(&g_flow_count[2 * k], (unsigned long long)c_runs[k]); atomicAdd(&g_flow_count[2 * k + 1], (unsigned long long)c_lanes[k]); }
        atomicAdd(&g_flow_count[8], (unsigned long long)c_trips);
    }
    atomicAdd(&g_flow_count[9], (unsigned long long)c_lds);
#endif
    count_rays(A.rays, nrays);
}


// =====================================================================================================================
// trace_tiles -- the tile-per-wave form of rt_triangles.hip (a pixel per lane for the pixel's whole life, the reference's
// nested walks as nested loops: neighbouring rays keep their lanes in step) on the memory layout of this file: persistent
// 16-wave workgroups take tiles (or quarter tiles) from the frame's cursor, the BLAS walk reads the relinked pair records
// -- LDS for the staged head, global memory below --, metas come packed, the stacks are the hybrid ones.  Measured against
// trace_flow (profiles/r04): the step machine executes 1.5 x the vector instructions of the nested loops at 27 of 64 lanes
// per block run; coherent waves are the cheaper way through this scene, and what the nested loops lacked was the short
// latency of the upper tree levels and an end of the frame that is not one workgroup per tile.
struct TileCtx {
    uint32_t a_nodes, a_blas, a_pairs, a_bst, a_tst, n_head, n_lookup, lds_pairs, q;
    uint32_t* ovf_b; uint16_t* ovf_t;
    const float4* pairs;
};

template <int WAVES>
__device__ __forceinline__ void tiles_bpush(const TileCtx& C, uint32_t s, uint32_t v) {
    if (s < kFlowKB) *(lds_u32)(uintptr_t)(C.a_bst + s * (64u * WAVES * 4u)) = v;
    else C.ovf_b[(s - kFlowKB) * 64u] = v;
}
template <int WAVES>
__device__ __forceinline__ uint32_t tiles_bread(const TileCtx& C, uint32_t s) {
    uint32_t v;
    if (s < kFlowKB) v = *(lds_u32)(uintptr_t)(C.a_bst + s * (64u * WAVES * 4u));
    else v = C.ovf_b[(s - kFlowKB) * 64u];
    return v;
}
template <int WAVES>
__device__ __forceinline__ void tiles_tpush(const TileCtx& C, uint32_t s, uint32_t v) {
    if (s < kFlowKT) *(lds_u16)(uintptr_t)(C.a_tst + s * (64u * WAVES * 2u)) = (uint16_t)v;
    else C.ovf_t[(s - kFlowKT) * 64u] = (uint16_t)v;
}
template <int WAVES>
__device__ __forceinline__ uint32_t tiles_tread(const TileCtx& C, uint32_t s) {
    uint32_t v;
    if (s < kFlowKT) v = (uint32_t)*(lds_u16)(uintptr_t)(C.a_tst + s * (64u * WAVES * 2u));
    else v = (uint32_t)C.ovf_t[(s - kFlowKT) * 64u];
    return v;
}

// RK:246-332 traceBLAS over pair records (the normal transform RK:334-338 is deferred to hit_normal, as in rt_triangles.hip)
template <int WAVES>
__device__ __forceinline__ void tiles_blas(const RtTriScene& T, const TileCtx& C, uint32_t bi, v3 o, v3 d, float& nearest, TriHit& hit) {
    float m[18];                                                    // mat4 column-major, m[4c + r]; m[17]: the root's meta
    {
        const uint32_t a = C.a_blas + 80u * bi;
        const float4 r0 = lds_read4(a), r1 = lds_read4(a + 16u), r2 = lds_read4(a + 32u), r3 = lds_read4(a + 48u);
        m[0] = r0.x; m[1] = r0.y; m[2] = r0.z; m[3] = r0.w; m[4] = r1.x; m[5] = r1.y; m[6] = r1.z; m[7] = r1.w;
        m[8] = r2.x; m[9] = r2.y; m[10] = r2.z; m[11] = r2.w; m[12] = r3.x; m[13] = r3.y; m[14] = r3.z; m[15] = r3.w;
        m[17] = lds_read1(a + 68u);
    }
    const v3 oo = V(((m[0] * o.x + m[4] * o.y) + m[8] * o.z) + m[12] * 1.0f,
                    ((m[1] * o.x + m[5] * o.y) + m[9] * o.z) + m[13] * 1.0f,
                    ((m[2] * o.x + m[6] * o.y) + m[10] * o.z) + m[14] * 1.0f);       // RK:254
    const v3 od = V(((m[0] * d.x + m[4] * d.y) + m[8] * d.z) + m[12] * 0.0f,
                    ((m[1] * d.x + m[5] * d.y) + m[9] * d.z) + m[13] * 0.0f,
                    ((m[2] * d.x + m[6] * d.y) + m[10] * d.z) + m[14] * 0.0f);       // RK:255
    const v3 inv = V(1.0f / od.x, 1.0f / od.y, 1.0f / od.z);        // RK:396
    uint32_t node = __float_as_uint(m[17]);                         // RK:265
    uint32_t sp = 0;                                                // RK:267
    float blasNearest = nearest;                                    // RK:269
    for (;;) {                                                      // RK:271
        const uint32_t count = node >> 16, left = node & 0xFFFFu;   // RK:272-273
        if (count == 0u) {                                          // RK:275
            float4 q0, q1, q2, q3;
            if (left < C.lds_pairs) {
                const uint32_t a = C.a_pairs + 16u * left;
                q0 = lds_read4(a); q1 = lds_read4(a + C.q); q2 = lds_read4(a + 2u * C.q); q3 = lds_read4(a + 3u * C.q);
            } else {
                const float4* g = C.pairs + 4u * (size_t)left;
                q0 = g[0]; q1 = g[1]; q2 = g[2]; q3 = g[3];
            }
            NodeR c1, c2;
            c1.lo = V(q0.x, q0.y, q0.z); c1.hi = V(q1.x, q1.y, q1.z);
            c2.lo = V(q2.x, q2.y, q2.z); c2.hi = V(q3.x, q3.y, q3.z);
            const uint32_t m1 = __float_as_uint(q0.w), m2 = __float_as_uint(q2.w);
            float d1 = hit_aabb(oo, inv, c1);                       // RK:279
            float d2 = hit_aabb(oo, inv, c2);                       // RK:280
            const bool swap = d1 > d2;                              // RK:283-290
            if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; }
            if (d1 > blasNearest) {                                 // RK:292
                if (sp == 0u) break;
                sp -= 1u;
                node = tiles_bread<WAVES>(C, sclamp(sp));           // RK:297-298
            } else {
                node = swap ? m2 : m1;                              // RK:302
                if (d2 < blasNearest) {                             // RK:303-304 (no overflow guard upstream)
                    tiles_bpush<WAVES>(C, sclamp(sp), swap ? m1 : m2);
                    sp += 1u;
                }
            }
        } else {
            for (uint32_t i = 0; i < count; ++i) {                  // RK:311
                uint32_t li = i + left;
                if (li >= T.n_tri_lookup) li = T.n_tri_lookup - 1u;  // RK:314: the lookup itself is folded into T.corners
                float t, u, v;
                if (hit_triangle(T, li, oo, od, blasNearest, t, u, v)) {   // RK:312-321
                    blasNearest = t;
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = (int)li; hit.blas = (int)bi;
                }
            }
            if (sp == 0u) break;                                    // RK:324
            sp -= 1u;
            node = tiles_bread<WAVES>(C, sclamp(sp));               // RK:328-329
        }
    }
    nearest = blasNearest < nearest ? blasNearest : nearest;        // RK:227-229
}

// RK:168-244 traceTLAS
template <int WAVES>
__device__ __forceinline__ TriHit tiles_tlas(const RtTriScene& T, const TileCtx& C, v3 o, v3 d) {
    TriHit hit; hit.t = 0.0f; hit.u = hit.v = 0.0f; hit.tri = -1; hit.blas = -1;   // RK:170-171
    float nearest = 9999.0f;                                        // RK:172
    const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    NodeR node = flow_node(T, C.a_nodes, C.n_head, 0u);             // RK:175
    uint32_t sp = 0;
    for (;;) {                                                      // RK:179
        const uint32_t count = u32f(node.count);                    // RK:180
        const uint32_t left = u32f(node.left);                      // RK:181
        if (count == 0u) {                                          // RK:183
            uint32_t i2 = left + 1u;
            const NodeR c1 = flow_node(T, C.a_nodes, C.n_head, left), c2 = flow_node(T, C.a_nodes, C.n_head, left + 1u);
            float d1 = hit_aabb(o, inv, c1);                        // RK:186
            float d2 = hit_aabb(o, inv, c2);                        // RK:187
            const bool swap = d1 > d2;                              // RK:190-196
            if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; i2 = left; }
            if (d1 > nearest) {                                     // RK:198
                if (sp == 0u) break;
                sp -= 1u;
                node = flow_node(T, C.a_nodes, C.n_head, tiles_tread<WAVES>(C, sclamp(sp)));
            } else {
                node = swap ? c2 : c1;                              // RK:208
                if (d2 < nearest) {                                 // RK:209
                    tiles_tpush<WAVES>(C, sclamp(sp), i2 < T.n_nodes ? i2 : T.n_nodes - 1u);
                    sp += 1u;
                    if (sp > kStack) sp = kStack - 1u;              // RK:212-214 guards with `>`
                }
            }
        } else {
            for (uint32_t i = 0; i < count; ++i) {                  // RK:220
                uint32_t li = i + left;
                if (li >= T.n_blas_lookup) li = T.n_blas_lookup - 1u;
                float bif;
                if (li < C.n_lookup) bif = lds_read1(C.a_blas + 80u * li + 76u); else bif = T.blas_lookup[li];   // RK:223
                uint32_t bi = u32f(bif);
                if (bi >= T.n_blas) bi = T.n_blas - 1u;
                tiles_blas<WAVES>(T, C, bi, o, d, nearest, hit);    // RK:221-230
            }
            if (sp == 0u) break;                                    // RK:233
            sp -= 1u;
            node = flow_node(T, C.a_nodes, C.n_head, tiles_tread<WAVES>(C, sclamp(sp)));   // RK:237-238
        }
    }
    return hit;
}

template <int WAVES, bool FLAT>
__global__ __launch_bounds__(64 * WAVES, 4) void trace_tiles(const RtFrameArgs A, const RtTriScene T, const RtFlowArgs F) {
    extern __shared__ float4 lds[];
    constexpr uint32_t stride = 64u * WAVES;
    float4* const s_nodes = lds;
    float* const s_blas = reinterpret_cast<float*>(lds + 2u * kLdsNodes);
    uint32_t* const s_bstack = reinterpret_cast<uint32_t*>(s_blas + 20u * kFlowInst);
    uint16_t* const s_tstack = reinterpret_cast<uint16_t*>(s_bstack + kFlowKB * stride);
    float4* const s_pairs = reinterpret_cast<float4*>(s_tstack + kFlowKT * stride);
    const uint32_t n_head = T.n_nodes < kLdsNodes ? T.n_nodes : kLdsNodes;
    const uint32_t n_blas = T.n_blas < kFlowInst ? T.n_blas : kFlowInst;
    const uint32_t n_lookup = T.n_blas_lookup < n_blas ? T.n_blas_lookup : n_blas;
    for (uint32_t i = threadIdx.x; i < 2u * n_head; i += stride) s_nodes[i] = T.nodes[i];
    for (uint32_t i = threadIdx.x; i < 20u * n_blas; i += stride) {
        const uint32_t r = i / 20u, k = i % 20u;
        float v = T.blas[i];
        if (k == 17u) v = __uint_as_float(F.root_meta[r]);
        if (k == 19u && r < n_lookup) v = T.blas_lookup[r];
        s_blas[i] = v;
    }
    for (uint32_t i = threadIdx.x; i < 4u * F.lds_pairs; i += stride) s_pairs[(i & 3u) * F.lds_pairs + (i >> 2)] = F.pairs[i];   // quarter-major, see trace_flow
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    TileCtx C;
    C.a_nodes = (uint32_t)(uintptr_t)s_nodes; C.a_blas = (uint32_t)(uintptr_t)s_blas; C.a_pairs = (uint32_t)(uintptr_t)s_pairs;
    C.a_bst = (uint32_t)(uintptr_t)(s_bstack + threadIdx.x); C.a_tst = (uint32_t)(uintptr_t)(s_tstack + threadIdx.x);
    C.n_head = n_head; C.n_lookup = n_lookup; C.lds_pairs = F.lds_pairs; C.q = 16u * F.lds_pairs;
    C.ovf_b = F.ovf + (size_t)(blockIdx.x * WAVES + wave) * kFlowOvfWords + lane;
    C.ovf_t = reinterpret_cast<uint16_t*>(F.ovf + (size_t)(blockIdx.x * WAVES + wave) * kFlowOvfWords + (kStack - kFlowKB) * 64u) + lane;
    C.pairs = F.pairs;

    const Scene sc = unpack_scene(A);
    const uint32_t groups_x = (A.W + 7u) / 8u;
    const uint32_t n_tiles = groups_x * A.n_local_tiles;
    const uint32_t s4 = T.tile_order ? T.tile_order[0] : 0u, s16 = T.tile_order ? T.tile_order[1] : 0u;
    const uint32_t n_items = n_tiles + 3u * s4 + 15u * s16;         // a split tile is four quarter items or sixteen 2x2 items (order_tiles)
    uint32_t nrays = 0;
    // Items.  Atomics on ONE address complete about 12 ns apart: one per tile is a 0.2 ms floor under a 17,800-tile frame and 1.6 ms
    // under a 4K frame, and 4,096 waves asking at once queue behind each other.  So a wave takes the first part of its share
    // WITHOUT asking: items w, w + G, w + 2G ... (G waves in the grid) of the work list -- longest first, so the strided shares
    // are about equal --, F.static_pct per cent of the list; only the rest goes through the frame's cursor, in chunks.
    const uint32_t grid_waves = gridDim.x * (uint32_t)WAVES;
    const uint32_t wave_id = blockIdx.x * (uint32_t)WAVES + wave;
    const uint32_t rounds = (uint32_t)(((unsigned long long)(n_items / grid_waves) * F.static_pct) / 100ull);   // static rounds of G items
    const uint32_t dyn0 = rounds * grid_waves;                      // the cursor hands out items [dyn0, n_items)
    uint32_t k = 0, item = wave_id, item_end = 0;                   // k < rounds: static item wave_id + k G
    for (;;) {
        if (k < rounds) { item = wave_id + k * grid_waves; ++k; }
        else {
            if (item_end == 0u || item + 1u >= item_end) {
                const uint32_t seen = item_end ? item_end : dyn0;   // a lower bound of the cursor
                uint32_t want = n_items > seen ? (n_items - seen) / (4u * grid_waves) : 0u;
                want = want < 1u ? 1u : (want > 16u ? 16u : want);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&A.qctrl[2], want);
                base = __builtin_amdgcn_readfirstlane(base) + dyn0;
                item = base; item_end = base + want;
            } else {
                ++item;
            }
        }
        if (item >= n_items) break;
        const uint64_t clk0 = wall_clock64();
        uint32_t tile = item, part = 4u;                            // part 0-3: a 4x4 quarter; 4: the whole tile; 16-31: a 2x2 sixteenth
        if (T.tile_order) {
            const uint32_t* list = T.tile_order + 2;
            uint32_t i = item;
            if (i < 16u * s16) { tile = list[i >> 4]; part = 16u + (i & 15u); }
            else if ((i -= 16u * s16) < 4u * s4) { tile = list[s16 + (i >> 2)]; part = i & 3u; }
            else tile = list[s16 + s4 + (i - 4u * s4)];
        }
        const uint32_t by = tile / groups_x, bx = tile - by * groups_x;
        uint32_t px = lane & 7u, row = lane >> 3;
        if (part < 4u) { px = 4u * (part & 1u) + (lane & 3u); row = 4u * (part >> 1) + (lane >> 2); }
        else if (part >= 16u) { px = 2u * (part & 3u) + (lane & 1u); row = 2u * ((part >> 2) & 3u) + (lane >> 1); }
        const uint32_t x = bx * 8u + px;
        const uint32_t y = (A.tile_first + by * A.tile_step) * 8u + row;
        if ((part == 4u || (part < 4u && lane < 16u) || (part >= 16u && lane < 4u)) && x < A.W && y < A.H) {
            float dist = 0.0f;
            v3 color = V(1.0f, 1.0f, 1.0f);
            v3 ro = sc.cameraPos, rd = primary_dir(A, sc, x, y);
            float affect = 1.0f, sum = 0.0f;
            for (uint32_t bounce = 0; bounce < sc.bounces; ++bounce) {                       // RK:113
                const TriHit h = tiles_tlas<WAVES>(T, C, ro, rd);                            // RK:114
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
                const TriHit sh = tiles_tlas<WAVES>(T, C, sc.lightPos, sdir);                // RK:153
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
            reinterpret_cast<uint32_t*>(A.out)[opix] =
                compose_pixel_sky(scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, primary_dir(A, sc, x, y))), color, dist);   // RK:91-98
        }
        if (T.tile_cost && lane == 0u) atomicAdd(&T.tile_cost[tile], (uint32_t)(wall_clock64() - clk0));   // 10 ns ticks; quarters add up
    }
    count_rays(A.rays, nrays);
}

}  // namespace rtk

#ifdef RT_FLOW_COUNT
extern "C" __attribute__((visibility("default"))) int rt_debug_flow_counts(unsigned long long* dst, int clear) {
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(rtk::g_flow_count), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && clear) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(rtk::g_flow_count), z, sizeof z);
    }
    return (int)e;
}
#endif

size_t rt_flow_lds_bytes(uint32_t waves, uint32_t lds_pairs) {
    return (size_t)2u * rtk::kLdsNodes * 16u + 20u * kFlowInst * 4u + (size_t)kFlowKB * 64u * waves * 4u + (size_t)kFlowKT * 64u * waves * 2u +
           (size_t)lds_pairs * 64u;
}

// Pair records a workgroup of `waves` waves stages when `per_cu` such workgroups share a CU's 160 KB (handed out in 1280-byte granules).
uint32_t rt_flow_lds_pairs(uint32_t waves, uint32_t per_cu, uint32_t n_pairs) {
    const size_t granules = 128u / per_cu;
    const size_t room = granules * 1280u;
    const size_t fixed = rt_flow_lds_bytes(waves, 0u);
    if (room <= fixed) return 0u;
    const size_t k = (room - fixed) / 64u;
    return (uint32_t)(k < n_pairs ? k : n_pairs);
}

template <int WAVES>
static hipError_t launch_tiles_as(const RtFrameArgs& a, const RtTriScene& t, RtFlowArgs f, uint32_t per_cu, uint32_t blocks, hipStream_t s) {
    f.lds_pairs = rt_flow_lds_pairs(WAVES, per_cu, f.n_pairs);
    if (f.lds_pairs_cap && f.lds_pairs > f.lds_pairs_cap - 1u) f.lds_pairs = f.lds_pairs_cap - 1u;
    const size_t lds = rt_flow_lds_bytes(WAVES, f.lds_pairs);
    auto k = a.sky_flat ? rtk::trace_tiles<WAVES, true> : rtk::trace_tiles<WAVES, false>;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * WAVES), lds, s, a, t, f);
    return hipGetLastError();
}

template <int WAVES>
static hipError_t launch_flow_as(const RtFrameArgs& a, const RtTriScene& t, RtFlowArgs f, uint32_t per_cu, uint32_t blocks, hipStream_t s) {
    f.lds_pairs = rt_flow_lds_pairs(WAVES, per_cu, f.n_pairs);
    const size_t lds = rt_flow_lds_bytes(WAVES, f.lds_pairs);
    auto k = a.sky_flat ? rtk::trace_flow<WAVES, true> : rtk::trace_flow<WAVES, false>;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (!a.sky_flat && !a.fin) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * WAVES), lds, s, a, t, f);
    if (!a.sky_flat) return rt_launch_sky_resolve(a, s);
    return hipGetLastError();
}

// waves: waves per workgroup (16 / 8 / 4); per_cu: workgroups of this frame's launch sized to share a CU.
// steps: the step machine (trace_flow) instead of the tile-per-wave form (trace_tiles).
hipError_t rt_launch_flow(const RtFrameArgs& a, const RtTriScene& t, const RtFlowArgs& f, uint32_t waves, uint32_t per_cu, uint32_t blocks, bool steps, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    if (!steps) {
        g_rt_kernel_id = RT_KID_TRIANGLES_TILES;
        if (waves == 16u) return launch_tiles_as<16>(a, t, f, per_cu, blocks, s);
        if (waves == 8u) return launch_tiles_as<8>(a, t, f, per_cu, blocks, s);
        return launch_tiles_as<4>(a, t, f, per_cu, blocks, s);
    }
    g_rt_kernel_id = RT_KID_TRIANGLES_FLOW;
    if (waves == 16u) return launch_flow_as<16>(a, t, f, per_cu, blocks, s);
    if (waves == 8u) return launch_flow_as<8>(a, t, f, per_cu, blocks, s);
    return launch_flow_as<4>(a, t, f, per_cu, blocks, s);
}
