// Host-only headers of the triangle path under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizers_cpu.py):
//   rt_flow_build.h -- the relinked pair records of the BLAS trees, built from an ARBITRARY caller node buffer;
//   rt_tlas_fit.h   -- the host's walk of a frame's top-level tree that admits the small-stack kernel forms.
// Inputs: trees a builder would make, one-node buffers, random garbage, NaN / infinite / negative / huge indices and counts,
// cycles, roots beyond the buffer.  Checked: no out-of-bounds access or UB (the sanitizers), the invariants the kernel relies
// on, a walk over the pair records that visits what a walk over the nodes visits, and rt_tlas_fits against an independent
// simulation of the kernel's stack.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#include "../../compute_raytracer_amd/csrc/rt_flow_build.h"
#include "../../compute_raytracer_amd/csrc/rt_tlas_fit.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); ++fails; } } while (0)

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// a proper binary tree over `leaves` leaves laid out as the reference's builder does (children side by side), at node `base`
static void make_tree(std::vector<float>& nodes, uint32_t leaves, std::mt19937& gen, uint32_t lookup_slots) {
    std::uniform_real_distribution<float> U(-5.0f, 5.0f);
    struct Job { uint32_t node, lo, hi; };
    nodes.assign(8, 0.0f);
    std::vector<Job> todo{{0u, 0u, leaves}};
    while (!todo.empty()) {
        const Job j = todo.back(); todo.pop_back();
        float* p = nodes.data() + 8u * (size_t)j.node;
        const float a = U(gen), b = U(gen), c = U(gen);
        p[0] = a; p[1] = b; p[2] = c; p[4] = a + 1.0f + std::fabs(U(gen)); p[5] = b + 1.0f; p[6] = c + 2.0f;
        if (j.hi - j.lo == 1u) { p[3] = (float)((j.lo * 2u) % (lookup_slots ? lookup_slots : 1u)); p[7] = (float)(1u + j.lo % 2u); continue; }
        const uint32_t left = (uint32_t)(nodes.size() / 8u);
        nodes.resize(nodes.size() + 16u, 0.0f);
        p = nodes.data() + 8u * (size_t)j.node;          // (resize may have moved the storage)
        p[3] = (float)left; p[7] = 0.0f;
        const uint32_t mid = j.lo + 1u + gen() % (j.hi - j.lo - 1u);
        todo.push_back({left, j.lo, mid});
        todo.push_back({left + 1u, mid, j.hi});
    }
}

// what a walk visits, as a multiset signature: over the nodes (the reference's way) and over the pair records
static void walk_nodes(const std::vector<float>& nodes, uint32_t root, std::vector<uint64_t>& leaves_seen, uint32_t& inner, uint32_t budget) {
    const uint32_t n = (uint32_t)(nodes.size() / 8u);
    std::vector<uint32_t> st{root < n - 1u ? root : n - 1u};
    while (!st.empty() && budget--) {
        const uint32_t i = st.back(); st.pop_back();
        const float* p = nodes.data() + 8u * (size_t)i;
        const uint32_t count = rt_flow_u32f(p[7]), left = rt_flow_u32f(p[3]);
        if (count) { leaves_seen.push_back(((uint64_t)(count < 0xFFFFu ? count : 0xFFFFu) << 32) | (left < 0xFFFFu ? left : 0xFFFFu)); continue; }
        ++inner;
        const uint32_t a = left < n - 1u ? left : n - 1u, b = (left + 1u) < n - 1u ? left + 1u : n - 1u;
        st.push_back(a); st.push_back(b);
    }
}
static void walk_pairs(const RtFlow& f, uint32_t meta, std::vector<uint64_t>& leaves_seen, uint32_t& inner, uint32_t budget) {
    std::vector<uint32_t> st{meta};
    while (!st.empty() && budget--) {
        const uint32_t m = st.back(); st.pop_back();
        if (m >> 16) { leaves_seen.push_back(((uint64_t)(m >> 16) << 32) | (m & 0xFFFFu)); continue; }
        ++inner;
        const uint32_t k = m & 0xFFFFu;
        if (k >= f.n_pairs) { CHECK(false, "pair index %u beyond %u", k, f.n_pairs); return; }
        st.push_back(bits(f.pairs[16u * (size_t)k + 3u]));
        st.push_back(bits(f.pairs[16u * (size_t)k + 11u]));
    }
}

static void check_build(const std::vector<float>& nodes, const std::vector<uint32_t>& roots, const char* what, bool proper_tree) {
    const uint32_t n = (uint32_t)(nodes.size() / 8u);
    RtFlow f;
    rt_flow_build(nodes.data(), n, roots.data(), (uint32_t)roots.size(), f);
    if (!f.ok) { CHECK(!proper_tree || n > 65536u, "%s: a proper tree was refused", what); return; }
    CHECK(f.pairs.size() == (size_t)f.n_pairs * 16u, "%s: record storage", what);
    CHECK(f.pair_of.size() == n, "%s: pair_of size", what);
    CHECK(f.n_pairs <= n, "%s: more pairs than nodes", what);
    CHECK(rt_flow_covers(f, roots.data(), (uint32_t)roots.size()), "%s: the build does not cover its own roots", what);
    uint32_t mc = 0, mx = 0;
    for (uint32_t k = 0; k < f.n_pairs; ++k)
        for (int c = 0; c < 2; ++c) {
            const uint32_t m = bits(f.pairs[16u * (size_t)k + 3u + 8u * c]);
            mc = std::max(mc, m >> 16); mx = std::max(mx, m & 0xFFFFu);
            if ((m >> 16) == 0u) CHECK((m & 0xFFFFu) < f.n_pairs, "%s: inner meta %u names pair %u of %u", what, m, m & 0xFFFFu, f.n_pairs);
        }
    CHECK(mc == f.max_count && mx == f.max_x, "%s: max_count / max_x", what);
    for (uint32_t r : roots) {
        const uint32_t i = r < n - 1u ? r : n - 1u;
        const uint32_t meta = rt_flow_meta(nodes.data(), n, i, f.pair_of);
        std::vector<uint64_t> a, b;
        uint32_t ia = 0, ib = 0;
        const uint32_t budget = proper_tree ? 1u << 20 : 4096u;       // garbage may be cyclic: compare a bounded prefix of the two walks
        walk_nodes(nodes, i, a, ia, budget);
        walk_pairs(f, meta, b, ib, budget);
        CHECK(a == b && ia == ib, "%s: the walk over pair records differs from the walk over nodes (root %u: %zu/%zu leaves, %u/%u inner)", what, r, a.size(), b.size(), ia, ib);
    }
}

// independent model of the kernel's TLAS stack (rt_tri_device.h: trace_tlas): the deepest the stack pointer can get when every
// box is hit and both children are taken -- by recursion, with its own depth limit
static int tlas_depth(const std::vector<float>& nodes, uint32_t i, uint32_t depth, uint32_t limit, uint32_t& max_index) {
    const uint32_t n = (uint32_t)(nodes.size() / 8u);
    if (i >= n) i = n - 1u;
    max_index = std::max(max_index, i);
    const float* p = nodes.data() + 8u * (size_t)i;
    if (rt_tlas_u32f(p[7]) != 0u) return (int)depth;
    if (depth >= limit) return 1 << 20;
    const uint32_t left = rt_tlas_u32f(p[3]);
    const int a = tlas_depth(nodes, left, depth + 1u, limit, max_index), b = tlas_depth(nodes, left + 1u, depth + 1u, limit, max_index);
    return a > b ? a : b;
}
static void check_fit(const std::vector<float>& nodes, const char* what) {
    const uint32_t n = (uint32_t)(nodes.size() / 8u);
    for (uint32_t d : {1u, 3u, 4u, 16u})
        for (uint32_t m : {1u, 8u, 16u, 31u}) {
            uint32_t mi = 0;
            const int deepest = tlas_depth(nodes, 0u, 0u, d, mi);
            const bool want = deepest <= (int)d && mi < m;
            CHECK(rt_tlas_fits(nodes.data(), n, d, m) == want, "%s: rt_tlas_fits(depth %u, nodes %u) != %d (deepest leaf %d, largest index %u)", what, d, m, (int)want, deepest, mi);
        }
}

int main() {
    std::mt19937 gen(20261005);
    // trees a builder makes: several meshes in one buffer, behind a TLAS head of 31 nodes
    for (uint32_t leaves : {1u, 2u, 3u, 7u, 64u, 1000u, 12174u}) {
        std::vector<float> mesh, all(31u * 8u, 0.0f);
        std::vector<uint32_t> roots;
        for (int k = 0; k < 3; ++k) {
            make_tree(mesh, leaves + (uint32_t)k, gen, 60000u);
            const uint32_t base = (uint32_t)(all.size() / 8u);
            for (size_t i = 0; i < mesh.size() / 8u; ++i)
                if (rt_flow_u32f(mesh[8 * i + 7]) == 0u) mesh[8 * i + 3] += (float)base;
            all.insert(all.end(), mesh.begin(), mesh.end());
            roots.push_back(base);
        }
        roots.push_back(roots[0]);                                     // duplicates allowed
        check_build(all, roots, "trees", true);
    }
    // top-level trees: balanced and degenerate, 1..16 leaves
    for (uint32_t leaves = 1; leaves <= 16u; ++leaves)
        for (int rep = 0; rep < 8; ++rep) {
            std::vector<float> t;
            make_tree(t, leaves, gen, 16u);
            check_fit(t, "tlas");
        }
    {   // a spine: depth = leaves - 1
        std::vector<float> t(8u * 9u, 0.0f);
        for (uint32_t k = 0; k < 4; ++k) { t[8 * (2 * k) + 3] = (float)(2 * k + 1); t[8 * (2 * k) + 7] = 0.0f; t[8 * (2 * k + 1) + 7] = 1.0f; if (k < 3) { } }
        t[8 * 8 + 7] = 1.0f;
        for (uint32_t k = 0; k < 4; ++k) t[8 * (2 * k + 2) + 7] = k < 3 ? 0.0f : 1.0f, t[8 * (2 * k + 2) + 3] = (float)(2 * k + 3);
        check_fit(t, "spine");
    }
    // one node, of either kind
    for (float count : {0.0f, 1.0f, 3.0f}) {
        std::vector<float> one{0, 0, 0, 0, 1, 1, 1, count};
        check_build(one, {0u}, "one node", count != 0.0f);
        check_build(one, {7u, 0u, 4000000000u}, "one node, wild roots", false);
        check_fit(one, "one node");
    }
    // garbage: random bit patterns as floats (NaNs, infinities, denormals, negatives), random counts / indices, cycles
    for (int rep = 0; rep < 400; ++rep) {
        const uint32_t n = 1u + gen() % (rep < 300 ? 40u : 3000u);
        std::vector<float> g(8u * (size_t)n);
        for (float& v : g) {
            const uint32_t kind = gen() % 8u;
            if (kind == 0u) { const uint32_t u = gen(); std::memcpy(&v, &u, 4); }
            else if (kind == 1u) v = NAN;
            else if (kind == 2u) v = (gen() & 1u) ? INFINITY : -INFINITY;
            else if (kind == 3u) v = 4294967296.0f * (float)(gen() % 3u);
            else if (kind == 4u) v = -(float)(gen() % 100u);
            else v = (float)(gen() % (2u * n + 2u));             // plausible indices and small counts, some beyond the buffer
        }
        for (uint32_t i = 0; i < n; ++i) if (gen() % 3u) g[8 * (size_t)i + 7] = (gen() % 4u) ? 0.0f : (float)(1u + gen() % 5u);
        std::vector<uint32_t> roots;
        for (uint32_t k = 0; k < 1u + gen() % 16u; ++k) roots.push_back((gen() % 5u) ? gen() % (n + 3u) : gen());
        check_build(g, roots, "garbage", false);
        check_fit(g, "garbage");
    }
    {   // a count beyond 16 bits: refused, not mangled
        std::vector<float> t{0, 0, 0, 1, 1, 1, 1, 0,   0, 0, 0, 0, 1, 1, 1, 70000.0f,   0, 0, 0, 5, 1, 1, 1, 2};
        RtFlow f;
        const uint32_t r0 = 0u;
        rt_flow_build(t.data(), 3u, &r0, 1u, f);
        CHECK(!f.ok, "a leaf of 70,000 triangles was accepted");
    }
    {   // empty / null
        RtFlow f;
        rt_flow_build(nullptr, 0u, nullptr, 0u, f);
        CHECK(!f.ok && f.n_pairs == 0u, "empty buffer");
        CHECK(!rt_tlas_fits(nullptr, 0u, 4u, 16u), "null tlas");
        CHECK(!rt_flow_covers(f, nullptr, 0u), "covers of a failed build");
    }
    if (fails) { std::printf("%d failures\n", fails); return 1; }
    std::printf("flow build ok\n");
    return 0;
}
