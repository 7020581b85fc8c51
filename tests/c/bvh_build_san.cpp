// Runs the host-side hierarchy build (compute_raytracer_amd/csrc/rt_bvh_build.h) under
// AddressSanitizer + UndefinedBehaviorSanitizer on ordinary, degenerate and hostile inputs and
// checks the layout invariants the device walk relies on.  Built and run by
// tests/test_sanitizers_cpu.py (sanitizers run on the CPU build only on this pool).
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../compute_raytracer_amd/csrc/rt_bvh_build.h"

static int check(const std::vector<float>& rec, uint32_t n, const char* what) {
    std::vector<float> out;
    std::vector<uint32_t> link;
    const uint32_t m = rt_bvh_build(rec.data(), n, out, link);
    if (n == 0) return (m == 0 && link.empty()) ? 0 : (std::printf("FAIL %s: empty scene\n", what), 1);
    if (link.size() != (size_t)m + 1 || out.size() != 4 * ((size_t)m + 1)) return std::printf("FAIL %s: sizes\n", what), 1;
    if (link[m] != 4u * m || !(out[4 * (size_t)m + 3] > 3e38f)) return std::printf("FAIL %s: sentinel\n", what), 1;
    std::vector<uint32_t> seen(n, 0);
    for (uint32_t i = 0; i < m; ++i) {
        if (link[i] & 0x80000000u) {
            const uint32_t s = link[i] & 0x7FFFFFFFu;
            if (s >= n || seen[s]++) return std::printf("FAIL %s: leaf %u\n", what, s), 1;
        } else if (link[i] % 4u || link[i] / 4u <= i + 1u || link[i] / 4u > m) {
            return std::printf("FAIL %s: link %u\n", what, i), 1;
        }
    }
    for (uint32_t s = 0; s < n; ++s)
        if (seen[s] != 1) return std::printf("FAIL %s: sphere %u missing\n", what, s), 1;
    return 0;
}

int main() {
    std::mt19937 gen(12345);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    int bad = 0;
    for (uint32_t n : {0u, 1u, 2u, 3u, 4u, 5u, 9u, 17u, 64u, 1000u, 5000u, 40000u}) {
        std::vector<float> rec(8 * (size_t)n);
        for (uint32_t i = 0; i < n; ++i) {
            rec[8 * i + 0] = 30.0f * U(gen); rec[8 * i + 1] = 3.0f * U(gen); rec[8 * i + 2] = 30.0f * U(gen);
            rec[8 * i + 7] = 0.05f + 0.3f * std::fabs(U(gen));
        }
        if (n > 8) { rec[1] = -100.0f; rec[7] = 100.0f; }                 // a ground sphere
        bad += check(rec, n, "random");
    }
    {   // coincident centres, zero and negative radii, a line of spheres, huge coordinates
        std::vector<float> rec(8 * 300, 0.0f);
        for (uint32_t i = 0; i < 100; ++i) rec[8 * i + 7] = (i % 3) ? 1.0f : 0.0f;
        for (uint32_t i = 100; i < 200; ++i) { rec[8 * i] = 0.5f * (float)i; rec[8 * i + 7] = -0.2f; }
        for (uint32_t i = 200; i < 300; ++i) { rec[8 * i] = 1e18f * U(gen); rec[8 * i + 1] = 1e-30f; rec[8 * i + 7] = 1e10f; }
        bad += check(rec, 300, "degenerate");
    }
    {   // more than 64 "large" spheres: the surplus joins the tree
        std::vector<float> rec(8 * 400, 0.0f);
        for (uint32_t i = 0; i < 400; ++i) {
            rec[8 * i] = 10.0f * U(gen); rec[8 * i + 2] = 10.0f * U(gen);
            rec[8 * i + 7] = i < 100 ? 50.0f : 0.01f;
        }
        bad += check(rec, 400, "many large");
    }
    {   // infinities and NaNs must not crash the build (the library renders such scenes literally)
        std::vector<float> rec(8 * 50, 1.0f);
        rec[8 * 7] = INFINITY; rec[8 * 9 + 1] = -INFINITY; rec[8 * 11 + 7] = INFINITY;
        bad += check(rec, 50, "infinite");
        for (uint32_t i = 0; i < 50; i += 3) rec[8 * i + (i % 3)] = NAN;
        rec[8 * 20 + 7] = NAN;
        bad += check(rec, 50, "nan");
    }
    std::printf(bad ? "bvh build: %d failures\n" : "bvh build ok\n", bad);
    return bad ? 1 : 0;
}
