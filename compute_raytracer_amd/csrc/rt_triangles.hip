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

#include "rt_device.h"
#include "rt_tri_types.h"


namespace rtk {

constexpr uint32_t kStack = 20u;                                  // RK:71

__device__ __forceinline__ uint32_t u32f(float f) {               // WGSL u32(f32)
    if (!(f > 0.0f)) return 0u;
    return f >= 4294967040.0f ? 4294967295u : (uint32_t)f;
}
__device__ __forceinline__ uint32_t sclamp(uint32_t i) { return i > kStack - 1u ? kStack - 1u : i; }

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
template <int WAVES>
__device__ __forceinline__ TriLds stage_head(const RtTriScene& T, float4* s_nodes, float* s_blas) {
    TriLds L;
    L.n_nodes = T.n_nodes < kLdsNodes ? T.n_nodes : kLdsNodes;
    L.n_blas = T.n_blas < kLdsBlas ? T.n_blas : kLdsBlas;
    for (uint32_t i = threadIdx.x; i < 2u * L.n_nodes; i += 64 * WAVES) s_nodes[i] = T.nodes[i];
    // (the last padding word of staged record k carries entry k of the BLAS lookup table, RK:223: one dependent global load
    // less per instance and ray)
    L.n_lookup = T.n_blas_lookup < L.n_blas ? T.n_blas_lookup : L.n_blas;
    for (uint32_t i = threadIdx.x; i < 20u * L.n_blas; i += 64 * WAVES)
        s_blas[i] = (i % 20u == 19u && i / 20u < L.n_lookup) ? T.blas_lookup[i / 20u] : T.blas[i];
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
template <bool COUNT, typename STK, bool PACKED>
__device__ __forceinline__ void trace_blas(const RtTriScene& T, const TriLds& L, uint32_t bi, v3 o, v3 d, float& nearest,
                                           TriHit& hit, typename std::conditional<PACKED, uint32_t, STK>::type* stack,
                                           uint32_t stride, float& traces) {
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
template <bool COUNT, typename STK, bool PACKED>
__device__ __forceinline__ TriHit trace_tlas(const RtTriScene& T, const TriLds& L, v3 o, v3 d, STK* tstack,
                                             typename std::conditional<PACKED, uint32_t, STK>::type* bstack, uint32_t stride, float& traces) {
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
                node = load_node_head(T, L, tstack[sclamp(sp) * stride]);
            } else {
                node = swap ? c2 : c1;                              // RK:208
                if (d2 < nearest) {                                 // RK:209
                    tstack[sclamp(sp) * stride] = (STK)(i2 < T.n_nodes ? i2 : T.n_nodes - 1u);
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
                trace_blas<COUNT, STK, PACKED>(T, L, bi, o, d, nearest, hit, bstack, stride, traces);   // RK:221-230
            }
            if (sp == 0u) break;                                    // RK:233
            sp -= 1u;
            node = load_node_head(T, L, tstack[sclamp(sp) * stride]);       // RK:237-238
        }
    }
    return hit;
}

// What hitTriangle (RK:381-387) and traceBLAS (RK:334-338) attach to the accepted hit -- in two parts, so that
// only the normal (which the reflection needs) is carried across the shadow ray's traversal; texture
// coordinate and colour are read when the bounce is shaded.
__device__ __forceinline__ v3 hit_normal(const RtTriScene& T, const TriHit& h) {
    const float* tr = T.tri + 40u * (size_t)tri_of(T, h.tri);
    const float w = 1.0f - h.u - h.v;                                                // RK:381
    const v3 nA = V(tr[4], tr[5], tr[6]), nB = V(tr[16], tr[17], tr[18]), nC = V(tr[28], tr[29], tr[30]);
    const v3 n = add(add(scale(w, nA), scale(h.u, nB)), scale(h.v, nC));             // RK:382
    const float* m = T.blas + 20u * (size_t)h.blas;
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

// ---- kernel: RK main over the triangle scene ---------------------------------------------------------
// FLAT: compiled for a one-colour 1x1 sky (A.sky_flat) -- no cube filtering code inside the traversal's register budget.
template <int WAVES, typename STK, int OCC, bool FLAT, bool PACKED>
__global__ __launch_bounds__(64 * WAVES, OCC) void trace_triangles(const RtFrameArgs A, const RtTriScene T) {
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
    // one workgroup per tile, or per quarter of a tile the previous frame on this stream found long (order_tiles)
    const uint64_t clk0 = wall_clock64();
    const uint32_t groups_x = (A.W + 8u * WAVES - 1u) / (8u * WAVES);
    const uint32_t n_tiles = groups_x * A.n_local_tiles;
    uint32_t tile = blockIdx.x, part = 4u;                                  // part 0-3: a 4x4 quarter; 4: the whole tile
    if (T.tile_order) {
        const uint32_t split = T.tile_order[0];
        if (blockIdx.x < 4u * split) { tile = T.tile_order[1u + (blockIdx.x >> 2)]; part = blockIdx.x & 3u; }
        else if (blockIdx.x - 3u * split < n_tiles) tile = T.tile_order[1u + blockIdx.x - 3u * split];
        else return;                                                        // the grid is sized for the most quarters there can be
    }
    const uint32_t by = tile / groups_x, bx = tile - by * groups_x;
    if (part != 4u && lane >= 16u) return;
    const uint32_t px = part == 4u ? (lane & 7u) : 4u * (part & 1u) + (lane & 3u);
    const uint32_t row = part == 4u ? (lane >> 3) : 4u * (part >> 1) + (lane >> 2);
    const uint32_t x = bx * (8u * WAVES) + wave * 8u + px;
    const uint32_t y = (A.tile_first + by * A.tile_step) * 8u + row;
    if (x >= A.W || y >= A.H) return;          // thread 0 leaves here only with its whole tile or quarter: its pixel is their first

    const Scene sc = unpack_scene(A);
    uint32_t nrays = 0;
    float dummy = 0.0f;
    float dist = 0.0f;
    v3 color = V(1.0f, 1.0f, 1.0f);
    v3 ro = sc.cameraPos, rd = primary_dir(A, sc, x, y);
    float affect = 1.0f, sum = 0.0f;
    for (uint32_t bounce = 0; bounce < sc.bounces; ++bounce) {                       // RK:113
        const TriHit h = trace_tlas<false, STK, PACKED>(T, L, ro, rd, tstack, bstack, stride, dummy); // RK:114
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
        const TriHit sh = trace_tlas<false, STK, PACKED>(T, L, sc.lightPos, sdir, tstack, bstack, stride, dummy);   // RK:153
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
    if (T.tile_cost && threadIdx.x == 0u) atomicAdd(&T.tile_cost[tile], (uint32_t)(wall_clock64() - clk0));   // 10 ns ticks; quarters add up
}

// ---- the order of the next frame's tiles ------------------------------------------------------------------
// The frame is a grid of independent tiles, one wave each, whose costs differ by more than an order of magnitude (a sky tile:
// one ray per pixel; a tile between two mirrors: 2 x bounces dependent traversals by 64 diverging lanes), and the hardware
// starts them in index order.  Measured with the kernel's own clock (tools/tile_cost_probe.py, profiles/r03/tile_cost.log):
// mean tile 21 us, one in a hundred 156 us, the longest 498 us -- and the 1344x846 frame rendered on its own (the
// reference's await-each-frame loop) took 504 us: a frame is as long as its longest tile, however empty the chip.  So
//   * every tile leaves the time it took (tile_cost; quarters add theirs up);
//   * one workgroup turns the costs into the NEXT frame's work list on the same stream: tiles longest first (a counting sort
//     over quarter-octave classes), and the tiles longer than half the frame's throughput time -- sum of all costs / wave
//     slots / 2 --, at most one in sixteen and 1024 (a quarter-wave for every wave slot of the chip), as four 4x4 quarters each: a quarter of the lanes diverge a quarter as much, and
//     the frame's longest wave shrinks accordingly.  order[0] = number of split tiles, order[1...] = the permutation.
// The picture does not depend on any of it: a pixel is rendered by the same code whatever its turn and company; any
// array of costs yields a permutation and a split count within the grid's bound.
__global__ __launch_bounds__(1024) void order_tiles(uint32_t* __restrict__ cost, uint32_t* __restrict__ order, uint32_t n, uint32_t wave_slots) {
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
        const uint32_t kt = cls(thr > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr);
        const unsigned long long whole = 4ull * thr;           // twice the throughput time
        const uint32_t kw = cls(whole > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)whole);
        uint32_t acc = 0u, split = 0u;
        bool pays = false;                         // some tile takes more than twice the throughput time: the frame waits for it.
        for (int k = 127; k >= 0; --k) {           // (Otherwise quarters only add waves: 4K, 0.77 -> 0.81 ms with them.)
            const uint32_t v = bin[k];
            bin[k] = acc; acc += v;
            if (k > (int)kt) split = acc;          // tiles of the classes above the threshold's: all longer than it
            if (k > (int)kw && v) pays = true;
        }
        if (!pays) split = 0u;
        const uint32_t most = n / 16u < 1024u ? n / 16u : 1024u;      // rt_tri_max_split
        order[0] = split < most ? split : most;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
        order[1u + atomicAdd(&bin[cls(cost[i])], 1u)] = i;
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
template <typename STK, int OCC, bool PACKED, int WAVES = 1>
static void launch_tri(const RtFrameArgs& a, const RtTriScene& t, int heatmap, hipStream_t s) {
    const dim3 grid((a.W + 8u * WAVES - 1u) / (8u * WAVES), a.n_local_tiles, 1);
    const uint32_t n_tiles = grid.x * grid.y;       // trace_triangles decodes the tile itself; with a work list: room for the quarters
    const dim3 line(t.tile_order ? n_tiles + 3u * std::min(n_tiles / 16u, 1024u) : n_tiles, 1, 1);   // order_tiles: at most that many tiles in quarters
    if (heatmap) hipLaunchKernelGGL((rtk::heatmap_triangles<WAVES, STK, PACKED>), grid, dim3(64 * WAVES), 0, s, a, t);
    else if (a.sky_flat) hipLaunchKernelGGL((rtk::trace_triangles<WAVES, STK, OCC, true, PACKED>), line, dim3(64 * WAVES), 0, s, a, t);
    else                 hipLaunchKernelGGL((rtk::trace_triangles<WAVES, STK, OCC, false, PACKED>), line, dim3(64 * WAVES), 0, s, a, t);
}

hipError_t rt_launch_order_tiles(uint32_t* cost, uint32_t* order, uint32_t n_tiles, uint32_t wave_slots, hipStream_t s) {
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::order_tiles, dim3(1), dim3(1024), 0, s, cost, order, n_tiles, wave_slots);
    return hipGetLastError();
}

hipError_t rt_launch_triangles(const RtFrameArgs& a, const RtTriScene& t, int heatmap, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    g_rt_kernel_id = heatmap ? RT_KID_HEATMAP : RT_KID_TRIANGLES;
    if (t.n_nodes <= 65536u && t.packed_ok) launch_tri<uint16_t, 4, true>(a, t, heatmap, s);
    else if (t.n_nodes <= 65536u)           launch_tri<uint16_t, 4, false>(a, t, heatmap, s);
    else                                    launch_tri<uint32_t, 3, false>(a, t, heatmap, s);      // 44 KB of stacks: three workgroups per CU
    return hipGetLastError();
}
