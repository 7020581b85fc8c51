// rt_tri_device.h -- device-side pieces of the triangle path (rt_triangles.hip: pixel per lane, one tile per wave): the
// reference's node, box, triangle and texture arithmetic (RK:168-410), every expression in the oracle's order.  Include after rt_device.h / rt_tri_types.h in a
// translation unit compiled with -ffp-contract=off.
#pragma once
#include <type_traits>

#include "rt_device.h"
#include "rt_tri_types.h"


namespace rtk {

constexpr uint32_t kStack = 20u;                                  // RK:71

__device__ __forceinline__ uint32_t u32f(float f) {               // WGSL u32(f32)
    if (!(f > 0.0f)) return 0u;
    return f >= 4294967040.0f ? 4294967295u : (uint32_t)f;
}
__device__ __forceinline__ uint32_t sclamp(uint32_t i) { return i > kStack - 1u ? kStack - 1u : i; }
// The TLAS stack of a frame whose top-level tree the host has walked (rt_tlas_fit.h): no leaf deeper than kSmallStack levels,
// every node among the first kSmallNodes -- the stack pointer then never passes kSmallStack, the guard of RK:212 (20) is out
// of reach, and four slots hold what the reference keeps in twenty.  (Indexing clamps to the slots there are all the same.)
constexpr uint32_t kSmallStack = 4u, kSmallNodes = 16u;
constexpr uint32_t kTinyStack = 3u, kTinyNodes = 8u, kTinyBlas = 4u;    // ... and the form for up to four instances
constexpr uint32_t kMidStack = 8u, kMidNodes = 24u;                      // ... and the one for every tree twelve instances can have but a degenerate one
constexpr uint32_t kWideNodes = 32u, kWideBlas = 16u;                    // ... and, with kMidStack slots, for the sixteen instances that travel with a frame
template <uint32_t TS> __device__ __forceinline__ uint32_t tclamp(uint32_t i) { return i > TS - 1u ? TS - 1u : i; }

struct NodeR { v3 lo; float left; v3 hi; float count; };
__device__ __forceinline__ NodeR load_node(const RtTriScene& T, uint32_t i) {
    if (i >= T.n_nodes) i = T.n_nodes - 1u;
    const float4 a = T.nodes[2u * (size_t)i], b = T.nodes[2u * (size_t)i + 1u];
    NodeR n;
    n.lo = V(a.x, a.y, a.z); n.left = a.w;
    n.hi = V(b.x, b.y, b.z); n.count = b.w;
    return n;
}

// The head of the node buffer (the TLAS: RR:184-192 writes it at offset 0) and the BLAS records are
// staged in LDS by every workgroup: the TLAS walk and the per-instance set-up (matrix, root index)
// are a chain of dependent loads that every ray of every pixel pays, sky pixels included.
// (48 nodes + 12 records: with the packed BLAS stack a workgroup then takes 10,176 bytes = 8 LDS granules, sixteen per CU)
constexpr uint32_t kLdsNodes = 48u, kLdsBlas = 12u;
struct TriLds { const float4* nodes; uint32_t n_nodes; const float* blas; uint32_t n_blas; uint32_t n_lookup; };
__device__ __forceinline__ NodeR load_node_head(const RtTriScene& T, const TriLds& L, uint32_t i) {
    if (i >= T.n_nodes) i = T.n_nodes - 1u;
    if (i >= L.n_nodes) return load_node(T, i);
    const float4 a = L.nodes[2u * i], b = L.nodes[2u * i + 1u];
    NodeR n;
    n.lo = V(a.x, a.y, a.z); n.left = a.w;
    n.hi = V(b.x, b.y, b.z); n.count = b.w;
    return n;
}
template <int WAVES, uint32_t NODES = kLdsNodes, uint32_t BLAS = kLdsBlas, bool INST = false>
__device__ __forceinline__ TriLds stage_head(const RtTriScene& T, float4* s_nodes, float* s_blas) {
    TriLds L;
    L.n_nodes = T.n_nodes < NODES ? T.n_nodes : NODES;
    L.n_blas = T.n_blas < BLAS ? T.n_blas : BLAS;
    if constexpr (INST) {
        // the frame's instance data came with the kernel's arguments, laid out as staged (rt_tri_types.h: RtTriInst; the host put
        // the lookup entries and the roots' metas into the records' padding words)
        static_assert(NODES <= kInstHeadNodes && BLAS <= kInstBlas, "RtTriInst holds what the small forms stage");
        L.n_lookup = T.n_blas_lookup < L.n_blas ? T.n_blas_lookup : L.n_blas;
        float* const sn = reinterpret_cast<float*>(s_nodes);
        for (uint32_t i = threadIdx.x; i < 8u * L.n_nodes; i += 64 * WAVES) sn[i] = T.inst.head[i];
        for (uint32_t i = threadIdx.x; i < 20u * L.n_blas; i += 64 * WAVES) s_blas[i] = T.inst.blas[i];
        __syncthreads();
        L.nodes = s_nodes; L.blas = s_blas;
        return L;
    }
    for (uint32_t i = threadIdx.x; i < 2u * L.n_nodes; i += 64 * WAVES) s_nodes[i] = T.nodes[i];
    // (the last padding word of staged record k carries entry k of the BLAS lookup table, RK:223: one dependent global load
    // less per instance and ray)
    L.n_lookup = T.n_blas_lookup < L.n_blas ? T.n_blas_lookup : L.n_blas;
    // (with the relinked pair records, T.pairs: the padding word before it carries the root's (count, left) meta -- the root
    // node itself is then never loaded)
    for (uint32_t i = threadIdx.x; i < 20u * L.n_blas; i += 64 * WAVES) {
        float v = (i % 20u == 19u && i / 20u < L.n_lookup) ? T.blas_lookup[i / 20u] : T.blas[i];
        if (T.pairs && i % 20u == 17u) v = __uint_as_float(T.root_meta[i / 20u]);
        s_blas[i] = v;
    }
    __syncthreads();
    L.nodes = s_nodes; L.blas = s_blas;
    return L;
}

// RK:395-410
__device__ __forceinline__ float hit_aabb(v3 o, v3 inv, const NodeR& n) {
    const v3 t1 = V((n.lo.x - o.x) * inv.x, (n.lo.y - o.y) * inv.y, (n.lo.z - o.z) * inv.z);   // RK:397
    const v3 t2 = V((n.hi.x - o.x) * inv.x, (n.hi.y - o.y) * inv.y, (n.hi.z - o.z) * inv.z);   // RK:398
    const float t_min = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z)); // RK:399,402
    const float t_max = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z)); // RK:400,403
    if (t_min > t_max || t_max < 0.0f) return 99999.0f;                                        // RK:405-407
    return t_min;
}

__device__ __forceinline__ v3 cross3(v3 a, v3 b) {
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}

struct TriHit {       // what survives of RenderState during traversal
    float t, u, v;
    int tri, blas;    // tri: slot of the triangle lookup table (tri_of resolves it), < 0: no hit
};
// RK:314 for the accepted hit: the triangle a lookup slot names (u32() of the f32 index, clamped like every index)
__device__ __forceinline__ uint32_t tri_of(const RtTriScene& T, int slot) {
    uint32_t ti = u32f(T.tri_lookup[slot]);
    return ti >= T.n_tri ? T.n_tri - 1u : ti;
}

// RK:344-381, up to the acceptance test; normal / uv / colour are formed later for the winner.
// `slot` is the position in the triangle lookup table (RK:314: triangles[u32(triangleLookup[i + left])]); the three
// corners of that triangle come from T.corners, the library's own compact copy in lookup order (rt_tri_corners below).
__device__ __forceinline__ bool hit_triangle(const RtTriScene& T, uint32_t slot, v3 o, v3 d, float tMax,
                                             float& t_out, float& u_out, float& v_out) {
    const float4* tr = T.corners + 3u * (size_t)slot;
    const float4 A = tr[0], B = tr[1], C = tr[2];
    const v3 cornerA = V(A.x, A.y, A.z);
    const v3 edge1 = sub(V(B.x, B.y, B.z), cornerA);                // RK:354
    const v3 edge2 = sub(V(C.x, C.y, C.z), cornerA);                // RK:355
    const v3 rayCrossEdge2 = cross3(d, edge2);                      // RK:356
    const float det = dot(edge1, rayCrossEdge2);                    // RK:357
    if (det < 0.00001f) return false;                               // RK:359-362 (back faces culled)
    const v3 s = sub(o, cornerA);                                   // RK:364
    float u = dot(s, rayCrossEdge2);                                // RK:365
    if (u < 0.0f || u > det) return false;                          // RK:366
    const v3 sCrossEdge1 = cross3(s, edge1);                        // RK:370
    float v = dot(d, sCrossEdge1);                                  // RK:371
    if (v < 0.0f || u + v > det) return false;                      // RK:372
    const float invDet = 1.0f / det;                                // RK:376
    const float t = invDet * dot(edge2, sCrossEdge1);               // RK:377
    u = u * invDet;                                                 // RK:378
    v = v * invDet;                                                 // RK:379
    if (t > 0.001f && t < tMax) {                                   // RK:380 (tMin 0.001, RK:315)
        t_out = t; u_out = u; v_out = v;
        return true;
    }
    return false;
}

// RK:246-332 traceBLAS (the normal transform RK:334-338 is deferred to finish_hit)
// STK: the element type of the traversal stacks.  uint16_t when the node buffer has at most 65,536 entries:
// an index is stored clamped to the last node, which is what load_node makes of it anyway.
// PACKED: the loop needs of its current node only `count` and `left` -- the box was tested when the node was a child.
// RK:304 pushes the far child's INDEX and RK:297 / 328 load the node again when it is popped: a dependent global load whose
// result was in registers at push time.  The packed stack keeps (count << 16 | left) of the pushed child instead, in the
// same slot under the same clamping: a pop is one LDS read.  Valid while every count, child index and lookup slot fits 16
// bits, which the host checks when the buffers are written (rt_api.hip: packed_ok).
// PAIRS (with PACKED, and every instance staged): the walk reads the library's relinked pair records (rt_flow_build.h) -- one
// 64-byte record per inner node, metas packed when the copy was built: no u32(f32) per node, no packing at a push, and the
// root's meta comes with the staged instance record instead of a node load.  Same boxes, same order, same decisions.
// P16 (with PAIRS, scenes whose leaves hold at most three triangles and whose pair records and lookup slots number at most
// 16,384 -- what the reference's builder makes of meshes up to that size): a stack entry is (count << 14 | x) in TWO bytes;
// both stacks, the staged heads and the instance records then take 7,616 bytes of LDS per wave and a fifth wave per SIMD fits.
template <bool COUNT, typename STK, bool PACKED, bool PAIRS = false, bool P16 = false>
__device__ __forceinline__ void trace_blas(const RtTriScene& T, const TriLds& L, uint32_t bi, v3 o, v3 d, float& nearest,
                                           TriHit& hit, typename std::conditional<PACKED && !P16, uint32_t, STK>::type* stack,
                                           uint32_t stride, float& traces) {
    typedef typename std::conditional<PACKED && !P16, uint32_t, STK>::type BSTK;
    auto pack16 = [](uint32_t m) -> uint32_t { return ((m >> 16) << 14) | (m & 0x3FFFu); };
    auto unpack16 = [](uint32_t e) -> uint32_t { return ((e >> 14) << 16) | (e & 0x3FFFu); };
    float m[17];                                                    // mat4 column-major, m[4c + r]; m[16] root index
    if (bi < L.n_blas) {
#pragma unroll
        for (int k = 0; k < 17; ++k) m[k] = L.blas[20u * bi + (uint32_t)k];
    } else {
        const float* g = T.blas + 20u * (size_t)bi;
#pragma unroll
        for (int k = 0; k < 17; ++k) m[k] = g[k];
    }
    const v3 oo = V(((m[0] * o.x + m[4] * o.y) + m[8] * o.z) + m[12] * 1.0f,
                    ((m[1] * o.x + m[5] * o.y) + m[9] * o.z) + m[13] * 1.0f,
                    ((m[2] * o.x + m[6] * o.y) + m[10] * o.z) + m[14] * 1.0f);       // RK:254
    const v3 od = V(((m[0] * d.x + m[4] * d.y) + m[8] * d.z) + m[12] * 0.0f,
                    ((m[1] * d.x + m[5] * d.y) + m[9] * d.z) + m[13] * 0.0f,
                    ((m[2] * d.x + m[6] * d.y) + m[10] * d.z) + m[14] * 0.0f);       // RK:255
    const v3 inv = V(1.0f / od.x, 1.0f / od.y, 1.0f / od.z);        // RK:396
    if (PAIRS) {
        uint32_t pnode = __float_as_uint(L.blas[20u * bi + 17u]);   // RK:265: the root's (count, left), relinked
        uint32_t psp = 0;                                           // RK:267
        float pNearest = nearest;                                   // RK:269
        for (;;) {                                                  // RK:271
            const uint32_t count = pnode >> 16, left = pnode & 0xFFFFu;   // RK:272-273
            if (count == 0u) {                                      // RK:275
                if (COUNT) traces += 2.0f;                          // HK:242
                const float4* g = T.pairs + 4u * (size_t)left;
                const float4 q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3];
                NodeR c1, c2;
                c1.lo = V(q0.x, q0.y, q0.z); c1.hi = V(q1.x, q1.y, q1.z);
                c2.lo = V(q2.x, q2.y, q2.z); c2.hi = V(q3.x, q3.y, q3.z);
                const uint32_t m1 = __float_as_uint(q0.w), m2 = __float_as_uint(q2.w);
                float d1 = hit_aabb(oo, inv, c1);                   // RK:279
                float d2 = hit_aabb(oo, inv, c2);                   // RK:280
                const bool swap = d1 > d2;                          // RK:283-290
                if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; }
                if (d1 > pNearest) {                                // RK:292
                    if (psp == 0u) break;
                    psp -= 1u;
                    pnode = P16 ? unpack16(stack[sclamp(psp) * stride]) : (uint32_t)stack[sclamp(psp) * stride];   // RK:297-298
                } else {
                    pnode = swap ? m2 : m1;                         // RK:302
                    if (d2 < pNearest) {                            // RK:303-304 (no overflow guard upstream)
                        stack[sclamp(psp) * stride] = (BSTK)(P16 ? pack16(swap ? m1 : m2) : (swap ? m1 : m2));
                        psp += 1u;
                    }
                }
            } else {
                for (uint32_t i = 0; i < count; ++i) {              // RK:311
                    uint32_t li = i + left;
                    if (li >= T.n_tri_lookup) li = T.n_tri_lookup - 1u;
                    if (COUNT) traces += 1.0f;                      // HK:279
                    float t, u, v;
                    if (hit_triangle(T, li, oo, od, pNearest, t, u, v)) {   // RK:312-321
                        pNearest = t;
                        hit.t = t; hit.u = u; hit.v = v; hit.tri = (int)li; hit.blas = (int)bi;
                    }
                }
                if (psp == 0u) break;                               // RK:324
                psp -= 1u;
                pnode = P16 ? unpack16(stack[sclamp(psp) * stride]) : (uint32_t)stack[sclamp(psp) * stride];       // RK:328-329
            }
        }
        nearest = pNearest < nearest ? pNearest : nearest;          // RK:227-229
        return;
    }
    NodeR node = load_node(T, u32f(m[16]));                         // RK:265
    uint32_t sp = 0;                                                // RK:267
    float blasNearest = nearest;                                    // RK:269
    for (;;) {                                                      // RK:271
        const uint32_t count = u32f(node.count);                    // RK:272
        const uint32_t left = u32f(node.left);                      // RK:273
        if (count == 0u) {                                          // RK:275
            if (COUNT) traces += 2.0f;                              // HK:242
            uint32_t i1 = left, i2 = left + 1u;
            NodeR c1 = load_node(T, left), c2 = load_node(T, left + 1u);
            // `left` and `count` of the children come WITH their boxes: left to itself the compiler loads three
            // components of each corner, decides, and then asks memory again for the fourth of both children -- a
            // second dependent round trip in every step of a traversal that is nothing but such round trips.
            asm volatile("" : "+v"(c1.left), "+v"(c1.count), "+v"(c2.left), "+v"(c2.count));
            float d1 = hit_aabb(oo, inv, c1);                       // RK:279
            float d2 = hit_aabb(oo, inv, c2);                       // RK:280
            const bool swap = d1 > d2;                              // RK:283-290
            if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; i1 = left + 1u; i2 = left; }
            if (d1 > blasNearest) {                                 // RK:292
                if (sp == 0u) break;
                sp -= 1u;
                if (PACKED) { const uint32_t e = stack[sclamp(sp) * stride]; node.count = (float)(e >> 16); node.left = (float)(e & 0xFFFFu); }
                else node = load_node(T, stack[sclamp(sp) * stride]);    // RK:297-298
            } else {
                node = swap ? c2 : c1;                              // RK:302 tree[iChild1]
                (void)i1;
                if (d2 < blasNearest) {                             // RK:303, RK:304 (no overflow guard upstream)
                    if (PACKED) {
                        const uint32_t fc = u32f(swap ? c1.count : c2.count), fl = u32f(swap ? c1.left : c2.left);
                        stack[sclamp(sp) * stride] = ((fc < 0xFFFFu ? fc : 0xFFFFu) << 16) | (fl < 0xFFFFu ? fl : 0xFFFFu);
                    } else {
                        stack[sclamp(sp) * stride] = (STK)(i2 < T.n_nodes ? i2 : T.n_nodes - 1u);
                    }
                    sp += 1u;
                }
            }
        } else {
            for (uint32_t i = 0; i < count; ++i) {                  // RK:311
                uint32_t li = i + left;
                if (li >= T.n_tri_lookup) li = T.n_tri_lookup - 1u;  // RK:314: the lookup itself is folded into T.corners
                if (COUNT) traces += 1.0f;                          // HK:279
                float t, u, v;
                if (hit_triangle(T, li, oo, od, blasNearest, t, u, v)) {   // RK:312-321
                    blasNearest = t;
                    hit.t = t; hit.u = u; hit.v = v; hit.tri = (int)li; hit.blas = (int)bi;
                }
            }
            if (sp == 0u) break;                                    // RK:324
            sp -= 1u;
            if (PACKED) { const uint32_t e = stack[sclamp(sp) * stride]; node.count = (float)(e >> 16); node.left = (float)(e & 0xFFFFu); }
            else node = load_node(T, stack[sclamp(sp) * stride]);        // RK:328-329
        }
    }
    nearest = blasNearest < nearest ? blasNearest : nearest;        // RK:227-229: nearestHit = newRenderState.t on a hit
}

// RK:168-244 traceTLAS.  tstack / bstack: this lane's two LDS stacks.
template <bool COUNT, typename STK, bool PACKED, bool PAIRS = false, bool P16 = false, uint32_t TS = kStack>
__device__ __forceinline__ TriHit trace_tlas(const RtTriScene& T, const TriLds& L, v3 o, v3 d, STK* tstack,
                                             typename std::conditional<PACKED && !P16, uint32_t, STK>::type* bstack, uint32_t stride, float& traces) {
    TriHit hit; hit.t = 0.0f; hit.u = hit.v = 0.0f; hit.tri = -1; hit.blas = -1;   // RK:170-171
    float nearest = 9999.0f;                                        // RK:172
    const v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    NodeR node = load_node_head(T, L, 0u);                          // RK:175
    uint32_t sp = 0;
    for (;;) {                                                      // RK:179
        const uint32_t count = u32f(node.count);                    // RK:180
        const uint32_t left = u32f(node.left);                      // RK:181
        if (count == 0u) {                                          // RK:183
            if (COUNT) traces += 2.0f;                              // HK:143
            uint32_t i2 = left + 1u;
            const NodeR c1 = load_node_head(T, L, left), c2 = load_node_head(T, L, left + 1u);
            float d1 = hit_aabb(o, inv, c1);                        // RK:186
            float d2 = hit_aabb(o, inv, c2);                        // RK:187
            const bool swap = d1 > d2;                              // RK:190-196
            if (swap) { const float tmp = d1; d1 = d2; d2 = tmp; i2 = left; }
            if (d1 > nearest) {                                     // RK:198
                if (sp == 0u) break;
                sp -= 1u;
                node = load_node_head(T, L, tstack[tclamp<TS>(sp) * stride]);
            } else {
                node = swap ? c2 : c1;                              // RK:208
                if (d2 < nearest) {                                 // RK:209
                    tstack[tclamp<TS>(sp) * stride] = (STK)(i2 < T.n_nodes ? i2 : T.n_nodes - 1u);
                    sp += 1u;
                    // RK:212-214 guards with `>`, the heatmap twin with `>=` (HK:168)
                    if (COUNT ? sp >= kStack : sp > kStack) sp = kStack - 1u;
                }
            }
        } else {
            for (uint32_t i = 0; i < count; ++i) {                  // RK:220
                uint32_t li = i + left;
                if (li >= T.n_blas_lookup) li = T.n_blas_lookup - 1u;
                uint32_t bi = u32f(li < L.n_lookup ? L.blas[20u * li + 19u] : T.blas_lookup[li]);   // RK:223
                if (bi >= T.n_blas) bi = T.n_blas - 1u;
                trace_blas<COUNT, STK, PACKED, PAIRS, P16>(T, L, bi, o, d, nearest, hit, bstack, stride, traces);   // RK:221-230
            }
            if (sp == 0u) break;                                    // RK:233
            sp -= 1u;
            node = load_node_head(T, L, tstack[tclamp<TS>(sp) * stride]);   // RK:237-238
        }
    }
    return hit;
}

// What hitTriangle (RK:381-387) and traceBLAS (RK:334-338) attach to the accepted hit -- in two parts, so that
// only the normal (which the reflection needs) is carried across the shadow ray's traversal; texture
// coordinate and colour are read when the bounce is shaded.
// m: the hit instance's record (inverseModel, column-major) -- T.blas + 20 h.blas, or the staged copy of it
__device__ __forceinline__ v3 hit_normal(const RtTriScene& T, const TriHit& h, const float* m) {
    const float* tr = T.tri + 40u * (size_t)tri_of(T, h.tri);
    const float w = 1.0f - h.u - h.v;                                                // RK:381
    const v3 nA = V(tr[4], tr[5], tr[6]), nB = V(tr[16], tr[17], tr[18]), nC = V(tr[28], tr[29], tr[30]);
    const v3 n = add(add(scale(w, nA), scale(h.u, nB)), scale(h.v, nC));             // RK:382
    const v3 tn = V(((m[0] * n.x + m[1] * n.y) + m[2] * n.z) + m[3] * 0.0f,
                    ((m[4] * n.x + m[5] * n.y) + m[6] * n.z) + m[7] * 0.0f,
                    ((m[8] * n.x + m[9] * n.y) + m[10] * n.z) + m[11] * 0.0f);       // RK:335-337
    return normalize(tn);
}
struct Albedo { float u, v; v3 rgb; float w; };
__device__ __forceinline__ Albedo hit_albedo(const RtTriScene& T, int slot, float hu, float hv) {
    const float* tr = T.tri + 40u * (size_t)tri_of(T, slot);
    const float w = 1.0f - hu - hv;                                                  // RK:381
    Albedo s;
    s.u = (tr[8] * w + tr[20] * hu) + tr[32] * hv;                                   // RK:386
    s.v = 1.0f - ((tr[9] * w + tr[21] * hu) + tr[33] * hv);                          // RK:386-387
    s.rgb = V(tr[36], tr[37], tr[38]); s.w = tr[39];                                 // RK:384
    return s;
}

// textureSampleLevel(meshTex, texSamp, uv, 0).rgb with the cube map's sampler (RR:345-347):
// U repeat, V clamp-to-edge, bilinear; arithmetic of oracle/rt_oracle.c:tex2d_sample
__device__ inline v3 tex2d_sample(const RtTriScene& T, float u, float v) {
    const int w = (int)T.tex_w, h = (int)T.tex_h;
    const float x = u * (float)w - 0.5f;
    const float y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float wx = x - fx, wy = y - fy;
    const int x0 = fx >= 2147483520.0f ? 2147483520 : (fx <= -2147483520.0f ? -2147483520 : (int)fx);
    const int y0 = fy >= 2147483520.0f ? 2147483520 : (fy <= -2147483520.0f ? -2147483520 : (int)fy);
    const int xa = ((x0 % w) + w) % w, xb = (((x0 + 1) % w) + w) % w;
    const int ya = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);
    const int yb = y0 + 1 < 0 ? 0 : (y0 + 1 > h - 1 ? h - 1 : y0 + 1);
    const v3 c00 = texel(T.tex, w, h, xa, ya), c10 = texel(T.tex, w, h, xb, ya);
    const v3 c01 = texel(T.tex, w, h, xa, yb), c11 = texel(T.tex, w, h, xb, yb);
    return lerp3(lerp3(c00, c10, wx), lerp3(c01, c11, wx), wy);
}

}  // namespace rtk
