// rt_bvh_build.h -- host side of the bounding-sphere hierarchy (see rt_bvh.hip for the walk and for
// the proof the 4 % radius slack belongs to).  Plain C++17, no HIP: included by rt_bvh.hip, and
// compiled on its own with g++ -fsanitize=address,undefined by tests/test_sanitizers_cpu.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

// constants shared with the device side (rt_types.h); repeated here so that the header stands alone
#ifndef RT_FILTER_KAPPA
#define RT_FILTER_KAPPA 1.52587890625e-05f
#define RT_FILTER_EPS 7.62939453125e-06f
#define RT_FILTER_SCALE 1099511627776.0f
#define RT_FILTER_SCALE2 1208925819614629174706176.0f
#endif
#ifndef RT_BVH_SIGMA
#define RT_BVH_SIGMA 1.04   /* node radius = bound of the members x this: the slack of the node test's proof (rt_bvh.hip, header) */
#endif

// ---- host: hierarchy build ---------------------------------------------------------------------------
namespace {

struct Builder {
    const float* rec;              // [n][8] {cx,cy,cz,_, r,g,b, radius}
    std::vector<float>& out_rec;   // 4 floats per node
    std::vector<uint32_t>& out_link;
    std::vector<uint32_t> ids;
    uint32_t arity = 4;            // children per inner node at most (2..8)

    // NaN coordinates order as 0 (a sphere with a NaN in it can never be hit: every comparison of the
    // literal test is false), so that the sorts below keep a strict weak ordering on any input
    double cx(uint32_t i, int a) const {
        const double v = (double)rec[8u * (size_t)i + (size_t)a];
        return v == v ? v : 0.0;
    }
    double rad(uint32_t i) const {
        const double v = std::fabs((double)rec[8u * (size_t)i + 7u]);
        return v == v ? v : 0.0;
    }

    void leaf(uint32_t sphere) {
        out_rec.insert(out_rec.end(), {0.0f, 0.0f, 0.0f, 0.0f});   // filled on the device from geo_f
        out_link.push_back(0x80000000u | sphere);
    }

    // Splits ids[lo,hi) in two.  Up to 32768 members: the position, over all three axes, that
    // minimises  sum over both sides of (squared diagonal of the members' box) * count  -- a
    // surface-area heuristic with the box diagonal as the proxy for the bounding sphere
    // (17 % fewer node tests per ray than the median split on the BASELINE scenes,
    // tools/bvh_sim.py).  Larger ranges: median of the longest axis.  Ties by sphere index.
    uint32_t split2(uint32_t lo, uint32_t hi) {
        const uint32_t n = hi - lo;
        auto by_axis = [&](int ax) {
            return [this, ax](uint32_t a, uint32_t b) {
                const double va = cx(a, ax), vb = cx(b, ax);
                return va < vb || (va == vb && a < b);
            };
        };
        if (n > 32768u) {
            double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t k = lo; k < hi; ++k)
                for (int a = 0; a < 3; ++a) {
                    const double v = cx(ids[k], a);
                    mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v);
                }
            int ax = 0;
            if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
            if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
            const uint32_t mid = lo + n / 2u;
            std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, by_axis(ax));
            return mid;
        }
        double best = INFINITY;
        int best_ax = 0;
        uint32_t best_k = n / 2u;
        std::vector<uint32_t> order(n);
        std::vector<double> suffix(n);
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
            for (uint32_t k = n; k-- > 1u;) suffix[k] = grow(order[k]) * (double)(n - k);   // members k..n-1
            for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
            for (uint32_t k = 1; k < n; ++k) {                                              // members 0..k-1 | k..n-1
                const double cost = grow(order[k - 1u]) * (double)k + suffix[k];
                if (cost < best) { best = cost; best_ax = ax; best_k = k; }
            }
        }
        std::sort(ids.begin() + lo, ids.begin() + hi, by_axis(best_ax));
        return lo + best_k;
    }

    // bounding sphere of ids[lo,hi): centre of the members' box, then shrink-wrapped -- the centre
    // moves towards the farthest member while that reduces the radius
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
        // ... then towards the farthest member with a shrinking step (Badoiu-Clarkson: converges on the smallest enclosing
        // ball; the greedy walk above stops at its first non-improvement, typically 2 % in radius short of it), keeping
        // the best centre seen.  Any centre is valid -- the radius is measured afterwards --, a smaller ball is passed by fewer rays.
        if (hi - lo > 2u) {
            double Q[3] = {P[0], P[1], P[2]};
            for (int it = 1; it <= 96; ++it) {
                uint32_t fq = far;
                const double Rq = radius_at(Q, fq);
                if (Rq < Rp) { Rp = Rq; P[0] = Q[0]; P[1] = Q[1]; P[2] = Q[2]; }
                const double s[3] = {cx(fq, 0) - Q[0], cx(fq, 1) - Q[1], cx(fq, 2) - Q[2]};
                const double len = std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
                if (!(len > 1e-12)) break;
                // the farthest POINT of the farthest member: its centre pushed out by its radius
                const double k = (1.0 + rad(fq) / len) / (double)(it + 1);
                Q[0] += s[0] * k; Q[1] += s[1] * k; Q[2] += s[2] * k;
            }
        }
        const float C[3] = {(float)P[0], (float)P[1], (float)P[2]};   // the record stores the centre in fp32:
        const double Cd[3] = {C[0], C[1], C[2]};                       // the radius is taken about THAT point
        uint32_t unused = 0;
        double R = radius_at(Cd, unused);
        R *= RT_BVH_SIGMA;            // sigma, see the header
        const double c2 = (double)C[0] * C[0] + (double)C[1] * C[1] + (double)C[2] * C[2];
        const double k = c2 * (1.0 - (double)RT_FILTER_EPS) - R * R * (1.0 + (double)RT_FILTER_KAPPA);
        out[0] = C[0] * RT_FILTER_SCALE; out[1] = C[1] * RT_FILTER_SCALE; out[2] = C[2] * RT_FILTER_SCALE;
        out[3] = (float)(k * (double)RT_FILTER_SCALE2);
    }

    void emit(uint32_t lo, uint32_t hi) {
        if (hi - lo == 1u) { leaf(ids[lo]); return; }
        const size_t me = out_link.size();
        out_rec.insert(out_rec.end(), {0.0f, 0.0f, 0.0f, 0.0f});
        out_link.push_back(0u);
        children(lo, hi);
        float b[4];
        bound(lo, hi, b);
        std::copy(b, b + 4, out_rec.begin() + 4 * me);
        out_link[me] = 4u * (uint32_t)out_link.size();  // skip link: first node after this subtree, as 4 * index
    }

    // up to `arity` children (four: C3 1.50 / 1.55 / 1.61 / 1.65 ms with 4 / 5 / 6 / 8, profiles/r04/bvh_arity_probe.log): the largest
    // part is split until there are that many
    void children(uint32_t lo, uint32_t hi) {
        if (hi - lo <= arity) {
            for (uint32_t k = lo; k < hi; ++k) leaf(ids[k]);
            return;
        }
        uint32_t cut[9] = {lo, hi, 0, 0, 0, 0, 0, 0, 0};     // sorted part boundaries
        int parts = 1;
        while (parts < (int)arity) {
            int big = 0;
            for (int j = 1; j < parts; ++j)
                if (cut[j + 1] - cut[j] > cut[big + 1] - cut[big]) big = j;
            if (cut[big + 1] - cut[big] < 2u) break;
            const uint32_t mid = split2(cut[big], cut[big + 1]);
            for (int j = parts; j > big; --j) cut[j + 1] = cut[j];
            cut[big + 1] = mid;
            ++parts;
        }
        for (int j = 0; j < parts; ++j) emit(cut[j], cut[j + 1]);
    }
};

}  // namespace

// Builds the threaded hierarchy.  Top level: spheres much larger than the scene (a ground
// sphere) as leaves of their own -- inside a node they would inflate it to cover everything --,
// then up to four subtrees over the rest.  Returns the node count n; the arrays hold n + 1
// entries, the last one being the sentinel the traversal loop parks finished lanes on.
inline uint32_t rt_bvh_build(const float* records, uint32_t n, std::vector<float>& rec4, std::vector<uint32_t>& link, uint32_t arity = 4u) {
    rec4.clear(); link.clear();
    if (n == 0) return 0;
    rec4.reserve((size_t)n * 6u); link.reserve((size_t)n * 3u / 2u + 8u);
    Builder b{records, rec4, link, {}};
    b.arity = arity < 2u ? 2u : (arity > 8u ? 8u : arity);
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.cx(i, a)); mx[a] = std::max(mx[a], b.cx(i, a)); }
    // median radius: a sphere is "large" when it exceeds 8 medians AND an eighth of the centres' extent
    std::vector<double> radii(n);
    for (uint32_t i = 0; i < n; ++i) radii[i] = b.rad(i);
    std::nth_element(radii.begin(), radii.begin() + n / 2u, radii.end());
    const double med = radii[n / 2u];
    const double ext = std::sqrt((mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) + (mx[2] - mn[2]) * (mx[2] - mn[2]));
    uint32_t big = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const bool large = n > 8u && b.rad(i) > 8.0 * med && b.rad(i) > 0.125 * ext;
        if (large && big < 64u) { b.leaf(i); ++big; } else b.ids.push_back(i);
    }
    if (!b.ids.empty()) b.children(0u, (uint32_t)b.ids.size());
    const uint32_t nodes = (uint32_t)link.size();
    rec4.insert(rec4.end(), {0.0f, 0.0f, 0.0f, INFINITY});   // sentinel [nodes]: never passes, links to itself
    link.push_back(4u * nodes);
    return nodes;
}

