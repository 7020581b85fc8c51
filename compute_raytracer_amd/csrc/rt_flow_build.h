// rt_flow_build.h -- host side of the triangle kernel's PAIRS forms (rt_tri_device.h: trace_blas): the library's own copy of
// the BLAS trees, relinked.  Host-only, no HIP: tests/c/flow_build_test.cpp compiles it with g++ under ASan + UBSan
// (tests/test_sanitizers_cpu.py), on trees, one-node buffers and garbage.
//
// The reference's traceBLAS (RK:271-330) walks 32-byte nodes {min.xyz, leftChildIndex | max.xyz, primitiveCount}: an inner
// node (primitiveCount == 0) names its two children by ONE index, they sit side by side at leftChildIndex and
// leftChildIndex + 1, and a step of the walk reads both.  That pair is the unit here: one 64-byte record
//     {c1.min.xyz, meta1 | c1.max.xyz, 0 | c2.min.xyz, meta2 | c2.max.xyz, 0}
// with meta = primitiveCount << 16 | x, x = the first lookup slot of a leaf (leftChildIndex as the reference has it) or,
// for an inner child, the number of ITS pair record.  Boxes are the reference's own twelve floats, the conversions
// u32(f32) and the clamp-to-last-element rule of out-of-range node indices (oracle/rt_oracle.c: load_node) are applied here,
// once, instead of at every step: a walk over pair records visits the same boxes in the same order with the same
// decisions as the walk over nodes -- what a pair is called does not enter the arithmetic.
//
// Records are numbered so that the pairs a ray is most likely to visit come first -- greedy by the surface area of the
// parent's box, from the roots down (a child's box lies inside its parent's in any tree a builder produces, so the order
// is top-down) -- and a workgroup stages records [0, k) in LDS, whatever k its LDS has room for: "pair < k" is the whole
// address decision.  On the reference's own scene 512 records (32 KB) serve 76 % of all inner-node visits, 1,536 serve 89 %
// (tools/tri_path_stats.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <queue>
#include <utility>
#include <vector>

struct RtFlow {
    std::vector<float> pairs;            // 16 floats per record (metas as bit patterns)
    std::vector<uint32_t> pair_of;       // [n_nodes] record of the pair at (clamped) child index L, 0xFFFFFFFF: none
    std::vector<uint32_t> roots;         // the root node indices the build started from (sorted, unique)
    uint32_t n_pairs = 0;
    uint32_t n_nodes = 0;
    uint32_t min_node = 0xFFFFFFFFu;     // smallest node index the build read: a later write at or above it invalidates the copy
    bool ok = false;                     // false: the scene does not fit the 16-bit fields (the caller keeps the node walk)
    uint32_t max_count = 0, max_x = 0;   // largest count and x field of any meta in the records: (count << 14 | x) fits two bytes
                                         // while max_count <= 3 and max_x <= 16383 (the tile kernel's two-byte stack entries)
};

inline uint32_t rt_flow_u32f(float f) {                    // WGSL u32(f32): truncating, saturating, NaN -> 0
    if (!(f > 0.0f)) return 0u;
    return f >= 4294967040.0f ? 4294967295u : (uint32_t)f;
}

// meta of node `i` (already clamped) given the pair numbering; inner children must have a record
inline uint32_t rt_flow_meta(const float* nodes, uint32_t n_nodes, uint32_t i, const std::vector<uint32_t>& pair_of) {
    const float* p = nodes + 8u * (size_t)i;
    const uint32_t count = rt_flow_u32f(p[7]), left = rt_flow_u32f(p[3]);
    if (count == 0u) {
        const uint32_t key = left < n_nodes - 1u ? left : n_nodes - 1u;
        return pair_of[key] & 0xFFFFu;
    }
    return ((count < 0xFFFFu ? count : 0xFFFFu) << 16) | (left < 0xFFFFu ? left : 0xFFFFu);
}

// nodes: the node buffer as the reference writes it (8 floats per node); roots: node indices of the BLAS roots
// (u32(rootNodeIndex) of every instance record, any order, duplicates allowed).
inline void rt_flow_build(const float* nodes, uint32_t n_nodes, const uint32_t* roots, uint32_t n_roots, RtFlow& out) {
    out = RtFlow();
    out.n_nodes = n_nodes;
    if (n_nodes == 0u || n_nodes > 65536u) return;
    out.pair_of.assign(n_nodes, 0xFFFFFFFFu);
    auto clampi = [&](uint32_t i) { return i < n_nodes - 1u ? i : n_nodes - 1u; };
    auto area = [&](uint32_t i) -> double {
        const float* p = nodes + 8u * (size_t)i;
        const double ex = (double)p[4] - (double)p[0], ey = (double)p[5] - (double)p[1], ez = (double)p[6] - (double)p[2];
        const double a = 2.0 * (ex * ey + ey * ez + ex * ez);
        return a == a ? a : 0.0;                                  // NaN orders last among equals
    };
    // (area of the parent, order of discovery) -> key L of the pair; larger area first, earlier discovery first
    typedef std::pair<std::pair<double, int64_t>, uint32_t> item;
    std::priority_queue<item> heap;
    std::vector<uint8_t> queued(n_nodes, 0);
    int64_t serial = 0;
    bool wraps = false;
    auto offer = [&](uint32_t parent) {                           // parent: clamped index of a node; queues its children's pair
        out.min_node = std::min(out.min_node, parent);
        const float* p = nodes + 8u * (size_t)parent;
        if (rt_flow_u32f(p[7]) != 0u) return;                     // a leaf has no pair
        // leftChildIndex = 2^32 - 1 (any f32 >= 4294967040): the reference's `left + 1` wraps to node 0 (RK:277; the node walk
        // does the same) while every other index beyond the buffer clamps to the last node -- a pair keyed by its clamped left
        // index cannot tell the two apart.  Such a buffer keeps the node walk.
        if (rt_flow_u32f(p[3]) == 0xFFFFFFFFu) { wraps = true; return; }
        const uint32_t key = clampi(rt_flow_u32f(p[3]));
        if (queued[key]) return;
        queued[key] = 1;
        heap.push(item(std::make_pair(area(parent), -(serial++)), key));
    };
    std::vector<uint32_t> rs;
    for (uint32_t r = 0; r < n_roots; ++r) rs.push_back(clampi(roots[r]));
    std::sort(rs.begin(), rs.end());
    rs.erase(std::unique(rs.begin(), rs.end()), rs.end());
    out.roots = rs;
    for (uint32_t r : rs) offer(r);
    std::vector<uint32_t> order;                                  // keys in record order
    while (!heap.empty()) {
        const uint32_t key = heap.top().second;
        heap.pop();
        out.pair_of[key] = (uint32_t)order.size();
        order.push_back(key);
        const uint32_t a = key, b = clampi(key + 1u);
        out.min_node = std::min(out.min_node, a);
        offer(a);
        offer(b);
    }
    out.n_pairs = (uint32_t)order.size();
    if (out.n_pairs > 65536u || wraps) return;
    out.pairs.assign((size_t)out.n_pairs * 16u, 0.0f);
    for (uint32_t k = 0; k < out.n_pairs; ++k) {
        const uint32_t a = order[k], b = clampi(order[k] + 1u);
        float* q = out.pairs.data() + 16u * (size_t)k;
        const uint32_t child[2] = {a, b};
        for (int c = 0; c < 2; ++c) {
            const float* p = nodes + 8u * (size_t)child[c];
            if (rt_flow_u32f(p[7]) > 65535u) return;              // a count beyond 16 bits: not ok
            const uint32_t meta = rt_flow_meta(nodes, n_nodes, child[c], out.pair_of);
            out.max_count = std::max(out.max_count, meta >> 16);
            out.max_x = std::max(out.max_x, meta & 0xFFFFu);
            q[8 * c + 0] = p[0]; q[8 * c + 1] = p[1]; q[8 * c + 2] = p[2];
            std::memcpy(&q[8 * c + 3], &meta, 4);
            q[8 * c + 4] = p[4]; q[8 * c + 5] = p[5]; q[8 * c + 6] = p[6];
            q[8 * c + 7] = 0.0f;
        }
    }
    out.ok = true;
}

// Does the copy know every root of `roots`?  (The reference rewrites the instance records before every frame; their root
// indices do not change, but nothing in the interface says so.)
inline bool rt_flow_covers(const RtFlow& f, const uint32_t* roots, uint32_t n_roots) {
    if (!f.ok) return false;
    for (uint32_t r = 0; r < n_roots; ++r) {
        const uint32_t i = roots[r] < f.n_nodes - 1u ? roots[r] : f.n_nodes - 1u;
        if (!std::binary_search(f.roots.begin(), f.roots.end(), i)) return false;
    }
    return true;
}
