// rt_tlas_fit.h -- host side of the triangle kernel's small-stack form (rt_tri_device.h: kSmallStack / kSmallNodes).
// Host-only, no HIP: tests/c/tlas_fit_test.cpp compiles it with g++ under ASan + UBSan.
//
// The reference's traceTLAS (RK:168-244) keeps a stack of twenty node indices per ray and guards it at twenty (RK:212).  A
// walk that takes the nearer child and pushes the farther holds at most one entry per level of the tree above the node it is
// at, so a top-level tree whose leaves lie at most `max_depth` levels below the root never has more than `max_depth` entries
// on the stack: the guard cannot fire, and `max_depth` slots hold everything.  The reference's scene (three instances) has
// depth 2; sixteen instances in a balanced tree have depth 4.  The host walks the tree it is about to hand to a frame -- with
// the kernel's own index arithmetic (u32(f32), wrap of left + 1, clamp to the last node) -- and only a tree that passes
// takes the small form; anything else (deeper, cyclic, children beyond the staged nodes) keeps the twenty slots.
#pragma once
#include <cstdint>

inline uint32_t rt_tlas_u32f(float f) {                    // WGSL u32(f32): truncating, saturating, NaN -> 0
    if (!(f > 0.0f)) return 0u;
    return f >= 4294967040.0f ? 4294967295u : (uint32_t)f;
}

// nodes: the node buffer as the reference writes it (8 floats per node: min.xyz, leftChildIndex, max.xyz, primitiveCount),
// n_nodes of them; the walk starts at node 0 (RK:175).  true: every node the walk can reach is among the first `max_nodes`,
// and no leaf lies deeper than `max_depth` levels.  At most 2^(max_depth + 1) nodes are looked at.
inline bool rt_tlas_fits(const float* nodes, uint32_t n_nodes, uint32_t max_depth, uint32_t max_nodes) {
    if (!nodes || n_nodes == 0u || max_depth == 0u || max_depth > 16u) return false;
    struct Item { uint32_t i, depth; };
    Item stack[40];
    uint32_t sp = 0;
    stack[sp++] = Item{0u, 0u};
    while (sp) {
        const Item it = stack[--sp];
        const uint32_t i = it.i < n_nodes - 1u ? it.i : n_nodes - 1u;      // load_node_head: clamp to the last node
        if (i >= max_nodes) return false;
        const float* p = nodes + 8u * (size_t)i;
        if (rt_tlas_u32f(p[7]) != 0u) continue;                            // a leaf (RK:183): nothing is pushed for it
        if (it.depth >= max_depth) return false;                           // an inner node this deep: a leaf below max_depth, or a cycle
        const uint32_t left = rt_tlas_u32f(p[3]);
        if (sp + 2u > 40u) return false;
        stack[sp++] = Item{left, it.depth + 1u};
        stack[sp++] = Item{left + 1u, it.depth + 1u};                      // (wraps like the kernel's `left + 1u`)
    }
    return true;
}
