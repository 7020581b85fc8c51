// rt_blocks_build.h -- host side of the BLOCK form of the bounding-sphere hierarchy (rt_blocks.hip walks it).
// Plain C++17, no HIP: included by rt_api.hip / rt_blocks.hip, compiled on its own with g++ -fsanitize by
// tests/test_sanitizers_cpu.py.
//
// Why a second layout.  The threaded tree of rt_bvh_build.h is walked one node per step: a ray is a chain of ~46
// dependent LDS reads, and the kernel waits on that chain (profiles/r02: 39 % of a wave's cycles in SQ_WAIT_ANY).
// Here the unit of a step is a BLOCK: the (up to four) children of one node, stored together --
//
//     block = 4 records {C * 2^40, (|C|^2 (1-eps) - R^2 (1+kappa)) * 2^80}  +  4 links          = 80 bytes
//
// -- so that one step issues five INDEPENDENT reads off one address and tests four children; a ray is ~12
// dependent steps instead of ~46, at the same number of tests.  A block is uniform: its four entries are either all
// spheres (a LEAF block: link = 0x80000000 | sphere index, the record IS the sphere's filter record, written on
// the device from geo_f) or all nodes (an INNER block: link = index of the child block, the record bounds that
// block's whole subtree with the 4 % slack rt_bvh.hip derives).  Unused entries carry a record that never passes
// (w = +inf).  Block 0 is the sentinel: a leaf block of four such entries, where finished lanes idle.
//
// Shape.  Leaf blocks are kept FULL -- every binary split of the build falls on a multiple of four spheres, so
// only one leaf block per scene is partial -- and an inner block is two binary splits deep (a split of the range,
// then of each half), each chosen by the surface-area proxy of rt_bvh_build.h (sum over the sides of squared box
// diagonal x count) among the positions the DEPTH BUDGET allows: the walk keeps the children it has yet to visit
// on a per-lane stack in LDS, at most three per inner level, and the stack is sized for `levels` inner levels;
// a subtree at budget b holds at most 4^(b+1) spheres, which restricts how unbalanced a split may be.  The root
// gets one level more than a full tree would need, which leaves the upper splits free.
// Spheres much larger than the scene (a ground sphere) -- up to four -- are a leaf block of their own that every
// ray visits first; inside a node they would inflate it to cover everything.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#ifndef RT_FILTER_KAPPA
#define RT_FILTER_KAPPA 1.52587890625e-05f
#define RT_FILTER_EPS 7.62939453125e-06f
#define RT_FILTER_SCALE 1099511627776.0f
#define RT_FILTER_SCALE2 1208925819614629174706176.0f
#endif

struct RtBlockTree {
    std::vector<float> rec;         // [n_blocks][4][4]
    std::vector<uint32_t> link;     // [n_blocks][4]
    std::vector<uint32_t> sub_end;  // [n_blocks][4]  inner entry: one past the last block of the child's subtree (blocks are
                                    //                numbered depth-first, a subtree is the range [child, sub_end)); else 0
    uint32_t n_blocks = 0;
    uint32_t first = 0;             // where a ray starts: the block of large spheres if there is one, else the root
    uint32_t then = 0;              // ... and the block it visits after that one's subtree (0: none)
    uint32_t levels = 0;            // inner levels below (and including) the root: the walk's stack holds <= 3 * levels + 1 entries
};

namespace rt_blocks_detail {

constexpr uint32_t kLeafFlag = 0x80000000u, kPad = 0xFFFFFFFFu;

struct Builder {
    const float* rec;              // [n][8] {cx,cy,cz,_, r,g,b, radius}
    RtBlockTree& T;
    std::vector<uint32_t> ids;

    double cx(uint32_t i, int a) const {      // NaN orders as 0 (such a sphere can never be hit), as in rt_bvh_build.h
        const double v = (double)rec[8u * (size_t)i + (size_t)a];
        return v == v ? v : 0.0;
    }
    double rad(uint32_t i) const {
        const double v = std::fabs((double)rec[8u * (size_t)i + 7u]);
        return v == v ? v : 0.0;
    }

    uint32_t new_block() {
        const uint32_t b = T.n_blocks++;
        for (int k = 0; k < 4; ++k) {
            T.rec.insert(T.rec.end(), {0.0f, 0.0f, 0.0f, INFINITY});     // never passes until filled
            T.link.push_back(kPad);
            T.sub_end.push_back(0u);
        }
        return b;
    }

    // spheres a subtree at budget b may hold: a leaf block (b = 0) four, an inner block four children of budget b - 1
    static uint64_t capacity(uint32_t b) { return b >= 15u ? ~0ull : 4ull << (2u * b); }

    // Splits ids[lo, hi) in two at a multiple of four, k in [kmin, kmax] (members left of the cut), by the heuristic of
    // rt_bvh_build.h over all three axes.  Returns the cut position (absolute).
    uint32_t split2(uint32_t lo, uint32_t hi, uint32_t kmin, uint32_t kmax) {
        const uint32_t n = hi - lo;
        auto by_axis = [&](int ax) {
            return [this, ax](uint32_t a, uint32_t b) {
                const double va = cx(a, ax), vb = cx(b, ax);
                return va < vb || (va == vb && a < b);
            };
        };
        double best = INFINITY;
        int best_ax = 0;
        uint32_t best_k = std::min(std::max((n / 2u + 3u) & ~3u, kmin), kmax);
        std::vector<uint32_t> order(n);
        std::vector<double> suffix(n + 1u);
        for (int ax = 0; ax < 3; ++ax) {
            std::copy(ids.begin() + lo, ids.begin() + hi, order.begin());
            std::sort(order.begin(), order.end(), by_axis(ax));
            double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            auto grow = [&](uint32_t i) {
                double d2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    mn[a] = std::min(mn[a], cx(i, a) - rad(i));
                    mx[a] = std::max(mx[a], cx(i, a) + rad(i));
                    d2 += (mx[a] - mn[a]) * (mx[a] - mn[a]);
                }
                return d2;
            };
            for (uint32_t k = n; k-- > 1u;) suffix[k] = grow(order[k]) * (double)(n - k);      // members k .. n-1
            for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
            for (uint32_t k = 1; k < n; ++k) {                                                  // members 0 .. k-1 | k .. n-1
                const double left = grow(order[k - 1u]) * (double)k;
                if ((k & 3u) != 0u || k < kmin || k > kmax) continue;
                const double cost = left + suffix[k];
                if (cost < best) { best = cost; best_ax = ax; best_k = k; }
            }
        }
        std::sort(ids.begin() + lo, ids.begin() + hi, by_axis(best_ax));
        return lo + best_k;
    }

    // bounding sphere of ids[lo, hi): rt_bvh_build.h's construction (box centre, shrink-wrapped, radius about the f32 centre x 1.04)
    void bound(uint32_t lo, uint32_t hi, float out[4]) {
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t k = lo; k < hi; ++k)
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], cx(ids[k], a) - rad(ids[k]));
                mx[a] = std::max(mx[a], cx(ids[k], a) + rad(ids[k]));
            }
        auto radius_at = [&](const double P[3], uint32_t& far) {
            double R = -1.0;
            for (uint32_t k = lo; k < hi; ++k) {
                const uint32_t i = ids[k];
                const double dx = cx(i, 0) - P[0], dy = cx(i, 1) - P[1], dz = cx(i, 2) - P[2];
                const double d = std::sqrt(dx * dx + dy * dy + dz * dz) + rad(i);
                if (d > R) { R = d; far = i; }
            }
            return R;
        };
        double P[3] = {0.5 * (mn[0] + mx[0]), 0.5 * (mn[1] + mx[1]), 0.5 * (mn[2] + mx[2])};
        uint32_t far = ids[lo];
        double Rp = radius_at(P, far);
        for (int it = 0; it < 32; ++it) {
            const double s[3] = {cx(far, 0) - P[0], cx(far, 1) - P[1], cx(far, 2) - P[2]};
            const double len = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
            if (!(len > 1e-12)) break;
            const double Q[3] = {P[0] + s[0] / len * 0.05 * Rp, P[1] + s[1] / len * 0.05 * Rp, P[2] + s[2] / len * 0.05 * Rp};
            uint32_t far_q = far;
            const double Rq = radius_at(Q, far_q);
            if (!(Rq < Rp)) break;
            P[0] = Q[0]; P[1] = Q[1]; P[2] = Q[2]; Rp = Rq; far = far_q;
        }
        const float C[3] = {(float)P[0], (float)P[1], (float)P[2]};
        const double Cd[3] = {C[0], C[1], C[2]};
        uint32_t unused = 0;
        double R = radius_at(Cd, unused);
        R *= 1.04;
        const double c2 = (double)C[0] * C[0] + (double)C[1] * C[1] + (double)C[2] * C[2];
        const double k = c2 * (1.0 - (double)RT_FILTER_EPS) - R * R * (1.0 + (double)RT_FILTER_KAPPA);
        out[0] = C[0] * RT_FILTER_SCALE; out[1] = C[1] * RT_FILTER_SCALE; out[2] = C[2] * RT_FILTER_SCALE;
        out[3] = (float)(k * (double)RT_FILTER_SCALE2);
    }

    uint32_t leaf_block(uint32_t lo, uint32_t hi) {
        const uint32_t b = new_block();
        for (uint32_t k = lo; k < hi; ++k) T.link[4u * b + (k - lo)] = kLeafFlag | ids[k];     // records: filled on the device
        return b;
    }

    // the subtree over ids[lo, hi) with `budget` inner levels at its disposal; returns its root block
    uint32_t subtree(uint32_t lo, uint32_t hi, uint32_t budget) {
        const uint32_t n = hi - lo;
        if (n <= 4u) return leaf_block(lo, hi);
        const uint32_t b = new_block();
        const uint64_t cap = capacity(budget - 1u);          // of one child
        // two binary splits deep: the range, then each half; every cut on a multiple of four, every part within `cap`
        auto bounds = [&](uint32_t m, uint64_t cap_left, uint64_t cap_right, uint32_t& kmin, uint32_t& kmax) {
            const uint64_t lo_k = (uint64_t)m > cap_right ? (uint64_t)m - cap_right : 4u;
            kmin = (uint32_t)((std::max<uint64_t>(lo_k, 4u) + 3u) & ~3ull);
            kmax = (uint32_t)(std::min<uint64_t>(cap_left, (uint64_t)m - 1u) & ~3ull);
            if (kmax < kmin) kmax = kmin;                      // cannot happen while m <= cap_left + cap_right
        };
        uint32_t cut[5] = {lo, hi, hi, hi, hi};
        int parts = 1;
        {
            uint32_t kmin, kmax;
            bounds(n, 2u * cap, 2u * cap, kmin, kmax);
            const uint32_t mid = split2(lo, hi, kmin, kmax);
            const uint32_t a_hi = mid, b_lo = mid;
            parts = 0;
            auto half = [&](uint32_t h_lo, uint32_t h_hi) {
                const uint32_t m = h_hi - h_lo;
                if (m > 4u) {                                  // five to seven spheres: (4, rest) -- two leaf blocks, no node of two entries
                    uint32_t k0, k1;
                    bounds(m, cap, cap, k0, k1);
                    const uint32_t c = split2(h_lo, h_hi, k0, k1);
                    cut[parts++] = h_lo; cut[parts++] = c;
                } else {
                    cut[parts++] = h_lo;
                }
            };
            half(lo, a_hi);
            half(b_lo, hi);
            cut[parts] = hi;
        }
        for (int j = 0; j < parts; ++j) {
            float bs[4];
            bound(cut[j], cut[j + 1], bs);
            const uint32_t child = subtree(cut[j], cut[j + 1], budget - 1u);
            T.link[4u * b + (uint32_t)j] = child;
            T.sub_end[4u * b + (uint32_t)j] = T.n_blocks;
            std::copy(bs, bs + 4, T.rec.begin() + 16u * (size_t)b + 4u * (size_t)j);
        }
        return b;
    }
};

}  // namespace rt_blocks_detail

// Builds the block hierarchy over `n` sphere records (8 floats each, as rt_write_spheres takes them).
inline void rt_blocks_build(const float* records, uint32_t n, RtBlockTree& T) {
    using namespace rt_blocks_detail;
    T = RtBlockTree();
    Builder b{records, T, {}};
    b.new_block();                                         // block 0: the sentinel (four entries that never pass, leaf-typed)
    for (int k = 0; k < 4; ++k) T.link[(size_t)k] = kPad;
    if (n == 0) return;
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.cx(i, a)); mx[a] = std::max(mx[a], b.cx(i, a)); }
    std::vector<double> radii(n);
    for (uint32_t i = 0; i < n; ++i) radii[i] = b.rad(i);
    std::nth_element(radii.begin(), radii.begin() + n / 2u, radii.end());
    const double med = radii[n / 2u];
    const double ext = std::sqrt((mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) + (mx[2] - mn[2]) * (mx[2] - mn[2]));
    std::vector<uint32_t> large;
    for (uint32_t i = 0; i < n; ++i) {
        const bool is_large = n > 8u && b.rad(i) > 8.0 * med && b.rad(i) > 0.125 * ext;     // rt_bvh_build.h's rule
        if (is_large && large.size() < 4u) large.push_back(i); else b.ids.push_back(i);
    }
    uint32_t big = 0;
    if (!large.empty()) {
        big = b.new_block();
        for (size_t k = 0; k < large.size(); ++k) T.link[4u * big + (uint32_t)k] = kLeafFlag | large[k];
    }
    uint32_t root = 0;
    const uint32_t m = (uint32_t)b.ids.size();
    if (m) {
        uint32_t need = 0;                                   // inner levels of a FULL tree over m spheres
        while (Builder::capacity(need) < (uint64_t)m) ++need;
        const uint32_t budget = need + (m > 64u ? 1u : 0u);  // one spare level: the upper splits stay free
        root = b.subtree(0u, m, budget);
        // levels actually used (the stack is sized by them): depth of the deepest inner block
        std::vector<uint32_t> depth(T.n_blocks, 0u);
        uint32_t deepest = 0;
        for (uint32_t blk = root; blk < T.n_blocks; ++blk)
            for (int k = 0; k < 4; ++k) {
                const uint32_t l = T.link[4u * blk + (uint32_t)k];
                if (l == kPad || (l & kLeafFlag)) continue;
                depth[l] = depth[blk] + 1u;
                deepest = std::max(deepest, depth[l]);
            }
        // an inner block at depth d pushes; leaf blocks (the deepest) do not: levels = deepest depth (root = 0 counts when inner)
        const bool root_inner = !(T.link[4u * root] & kLeafFlag);
        T.levels = root_inner ? deepest : 0u;                // leaf blocks sit one below the deepest inner block
        if (root_inner && T.levels == 0u) T.levels = 1u;
    }
    if (big && root) { T.first = big; T.then = root; }
    else { T.first = big ? big : root; T.then = 0u; }
}
