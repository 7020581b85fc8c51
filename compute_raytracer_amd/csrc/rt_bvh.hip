// rt_bvh.hip -- sphere scenes through a bounding-sphere hierarchy (variant 4, "hierarchy").
//
// The reference traces sphere scenes by testing every sphere (HK:307-331 inside the loop of
// RK:168-244's predecessor); the brute-force kernels of rt_kernels.hip do the same and are bound
// by the FMA pipe.  This file reaches the same pixels -- bit for bit -- with far fewer tests:
//
//   * the host (rt_bvh_build.h) builds, once per rt_write_spheres, a 4-ary hierarchy of bounding SPHERES over the
//     scene's spheres and stores it in depth-first order with skip links ("threaded" tree): a
//     ray walks the array front to back, `i = pass ? i + 1 : skip[i]`, no stack;
//   * an inner node is tested with the same conservative discriminant filter as a sphere
//     (rt_filter.h: 8 v_fma_f32, one of them clamping, + v_cmp; trace_bvh below), on a record of the same layout
//     {C * 2^40, (|C|^2 (1-eps) - R^2 (1+kappa)) * 2^80}; its radius R covers every member sphere
//     with the slack derived below, so a node can only be skipped when no member can be hit;
//   * a leaf is one sphere: its record IS the filter record of rt_kernels.hip (geo_f), a lane
//     whose ray passes it appends the sphere index to its candidate list in LDS;
//   * after the walk the wave pools its lanes' candidates and evaluates them with the reference's literal
//     arithmetic, full lanes, keeping per ray the lexicographic minimum of (t, index) -- exactly what
//     the reference's in-order loop with `t < nearest` keeps (trace_bvh: drain);
//   * nodes and links are staged in LDS (20 B per node, ~1.35 nodes per sphere); lanes read them
//     with divergent addresses: the kernel sits where VALU issue and the latency of that dependent
//     chain of LDS reads meet, not on the FMA pipe alone like the brute-force kernels.
//
// One persistent kernel renders the frame: every lane carries one path through a small state
// machine (reflection ray -> shadow ray -> next bounce ...) and takes the next pixel from an
// atomic cursor when its path ends, so all 64 lanes of a wave walk the hierarchy together
// whatever their bounce depth; there is no path queue and no second kernel.
//
// Why a node test cannot lose a hit.  Notation: unit direction h, leaf sphere (c, r) with
// L = |o - c|, node (C, R), D = |C - c| <= Rg - r where Rg = max over members (|C - c_i| + r_i);
// rho_x = distance of the ray's LINE from x, T = h.(C - o); u = 2^-24.  The walk scales h by
// (1 + kappa_h), kappa_h = 2^-14 (four times the kappa of rt_filter.h: a leaf test only gets more
// permissive, its proof stands).
//   1. A literal hit (HK:316-318) needs disc > 0 and t > 0.001.  The rounding of the literal
//      discriminant, divided by 4a, is below 20u L^2 + 6u r^2 < 24u (L^2 + r^2) =: d (oc: u per
//      component; dot products: 4u |d||oc| and 5u |oc|^2; b*b: 36u; 4a*c: 40u; the difference: 4u,
//      in units of |d|^2 |oc|^2), so rho_c^2 <= r^2 + d; and t > 0.001 puts c in front of the
//      origin (the host allows the sign-aware form only for reach < 342, where the rounding of
//      h.oc cannot flip that).
//   2. The node test evaluates, in the filter's fused form (rt_filter.h), a value that exceeds
//      (1+kappa_h)^2 T^2 - |o-C|^2 + R^2 by more than its own rounding (the eps terms), so it
//      passes whenever rho_C^2 < R^2 + 2 kappa_h T^2.  The stored radius is R = Rg (1 + sigma),
//      sigma = 0.04:
//      near origins, L < 32 Rg:  sqrt(d) < 0.0383 Rg, so rho_C <= sqrt(r^2 + d) + D < Rg (1 + sigma) = R;
//      far origins,  L >= 32 Rg: rho_C^2 - Rg^2 <= d + 2 D sqrt(d) <= (1.43e-6 + 7.48e-5) L^2, while
//                                T >= 0.967 L gives 2 kappa_h T^2 >= 1.14e-4 L^2.
//   3. Sign-aware form min(b,0)^2 - (|o-C|^2 - R^2) > 0: a computed b < 0 is the unsigned test.  A
//      computed b >= 0 means T <= 7.3e-7 |o-C| while c is ahead; with P the point of closest
//      approach to c, |o-C|^2 = |P-C|^2 - t_c^2 + 2 t_c T <= (D + sqrt(r^2+d))^2 (1 + 2e-6), and here
//      L <= 2.1 Rg, so sqrt(d) < 0.003 Rg and |o-C| < 1.004 Rg < R: the origin is inside the node
//      and the test passes on its second term.
//
// Shadow rays are walked BACKWARDS under the sign-aware test (reversed_shadow_walk below): from just behind the shaded
// point P towards the light L, so that everything beyond P -- behind the walk's origin -- is dropped by min(b, 0) at no
// cost.  The candidates are still evaluated literally with the reference's ray (L, s), s = fl(normalize(P - L)), and the
// reference uses of that ray only this: `lit` iff the NEAREST literal hit t_min has |L + t_min s - P| < 0.005, else
// minIntensity (RK:147-165).  Notation: l = |P - L|, Lam = |L| + |P| + l, kappa_f = 2^-16 = 256u (the r^2 (1 + kappa_f) of
// every record), X = |w - c|^2 + r^2 for the walk's origin w.
//   A. What may be dropped.  P - L = l e for the exact unit vector e, |s - e| < 4u, so |L + t s - P| >= (t - l) - 7ut; forming
//      the hit point, the difference and its length adds < u (2t + |L|) and a relative 4u: a literal t >= l + dA,
//      dA = 0.005001 + 10u (l + |L|), yields diff >= 0.005.  A candidate set that holds every literally accepted sphere with
//      t < l + dA therefore gives the reference's result: a `lit` t_min is in it and is its minimum; otherwise what is left
//      is later than l + dA or nothing, and RK:165 returns the same value for "no hit" as for "hit elsewhere".
//   B. Where the walk starts.  w = fl(P + delta s), delta = 0.0051 + 2^-17 (|P-L|_1 + |L|_1); the walk looks along -s and adds
//      m = 2^-12 (|P-L|_1 + delta)^2 + 2^-21 (|P-L|_1 + |L|_1)^2 to the tested value (folded into the per-ray q; l1 norms
//      bound the Euclidean ones, lam = |w - L| <= |P-L|_1 + delta).  w is within eps_o < 6u Lam of the literal ray's line, at
//      parameter tau_w >= l + delta - 6u Lam.  Take a sphere (c, r) accepted literally with t < l + dA; T_c, rho_c:
//      parameter and distance of c on / from that line, wc^2 = max(0, r^2 - rho_c^2), d = 24u (|L-c|^2 + r^2) as in step 1.
//      Then rho_c^2 <= r^2 + d and, following HK:317 operation by operation, T_c - sqrt(wc^2 + d)(1 + u) <= t + 8ut + 4u |L-c|.
//      Seen from w (primed): rho'_c <= rho_c + eps_o and T'_c = tau_w - T_c +- eps_o.  Either
//        (1) T'_c > 0: c is ahead of w and rho'_c <= r + e1, e1 = sqrt(d) + eps_o; or
//        (2) T'_c <= 0: delta - dA >= 59u Lam pays for every term that depends on the ray alone (eps_o, tau_w, 8ut,
//            4u lam <= 24u Lam), which leaves -T'_c < sqrt(wc^2 + d)(1 + u) + 4u |w-c| and, with rho'_c^2 + wc^2 <=
//            (r + eps_o)^2 + d, |w - c| < r + e2, e2 = eps_o + sqrt(2d + 16u X): w is inside the sphere up to e2.
//      Priced from w: |L-c| <= lam + |w-c| gives d <= 48u lam^2 + 48u X (d <= 120u lam^2 + 30u |w-c|^2 + 24u r^2 where
//      that is tighter).
//   C. The tests pass.  The walk's test passes when rho'^2 < r^2 (1 + kappa_f) + 2 kappa_h T'^2 + m - 14u X (computed b < 0), or
//      when |w-c|^2 < r^2 (1 + kappa_f) + m - 14u X (computed b >= 0: T' <= 7.3e-7 |w-c|, so |w-c|^2 <= rho'^2 (1 + 1e-12)).
//      Leaf, case (1): X <= 2 r^2 + T'^2 (+ e1 terms), so d + 2 eps_o (r + sqrt d) + eps_o^2 + 14u X <=
//        48u lam^2 + 62u X + 40u r^2 + eps_o^2 / 40u <= [164u r^2] + [62u T'^2] + [48u lam^2 + 2u Lam^2]: under kappa_f r^2,
//        2 kappa_h T'^2 = 2048u T'^2 and m >= 4096u lam^2 + 2u Lam^2 (Lam <= 2 (|P-L|_1 + |L|_1)).
//      Leaf, case (2): |w-c|^2 < (r + eps_o)^2 + 2d + 16u X with X < 2 r^2 (1 + 1e-5): 2d + 30u X + 2 eps_o r + eps_o^2 <=
//        [108u + 60u + 40u] r^2 + [240u lam^2 + 2u Lam^2] -- 208u r^2 under kappa_f r^2 = 256u r^2, the rest under m.
//      Inner node (C, R = Rg (1 + sigma)), D = |C - c| <= Rg - r: rho'_C <= rho'_c + D and |w-C| <= |w-c| + D, so the node
//        passes if (Rg + e)^2 <= Rg^2 (1 + 0.0816) + 2 kappa_h T'_C^2 + m - 14u (|w-C|^2 + R^2) with e = e1 or e2 (node centre
//        behind w, member ahead: |w-C|^2 <= (rho'_c + D)^2 (1 + 2e-6) as in step 3).  Case (2): |w-C| <= Rg + e2 and
//        sqrt(X) <= 1.42 r + e2, so e2 <= eps_o + 2.4e-3 lam + 3.7e-3 Rg and 2 Rg e2 + e2^2 <= 0.07 Rg^2 + (2.4e-3 lam)^2 / 0.06
//        + ... = 0.07 Rg^2 + 9.6e-5 lam^2.  Case (1): e1 <= eps_o + 1.69e-3 (lam + sqrt X), sqrt X <= |w-C| + Rg <=
//        2 Rg + e1 + |T'_C|; 2 Rg * 1.69e-3 |T'_C| <= 1.69e-3 (15.4 Rg^2 + T'_C^2 / 15.4) = 0.026 Rg^2 + 1.1e-4 T'_C^2 (under
//        2 kappa_h T'_C^2 = 1.22e-4 T'_C^2 less the 14u); 2 Rg * 1.69e-3 lam <= 0.045 Rg^2 + 6.3e-5 lam^2; the rest
//        6.8e-3 Rg^2 + second-order terms < 0.003 Rg^2: 0.081 Rg^2 + 6.3e-5 lam^2 in all.  m >= 2.44e-4 lam^2 covers both.
// Forward rays (lam = 0, eps_o = 0, m = 0) are steps 1-3 unchanged.
// The argument is checked the only way that counts: frames are compared bit for bit with the
// oracle (tests/test_bvh_gpu.py: golden frames, random scenes over five orders of magnitude), and the walk, restated in
// emulated fp32, against the literal test on rays built to graze (tests/test_hierarchy_walk_cpu.py).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "rt_filter.h"

#define RT_BVH_KAPPA 6.103515625e-05f   /* kappa_h = 2^-14: inflation of the ray direction in the walk */
/* the reversed walk of shadow rays (claims A-C of the header) */
#define RT_BVH_REV_DELTA 0.0051f                    /* the walk starts delta = 0.0051 + 2^-17 (|P-L|_1 + |L|_1) behind P */
#define RT_BVH_REV_DELTA_REL 7.62939453125e-06f     /* 2^-17 */
#define RT_BVH_REV_SLACK 2.44140625e-04f            /* 2^-12 |w-L|^2 ... */
#define RT_BVH_REV_SLACK_ABS 4.76837158203125e-07f  /* ... + 2^-21 (|P-L|_1 + |L|_1)^2 added to the tested discriminant */

// Lanes still walking below which a wave suspends the walk (0: never): a launch argument (launch_bvh_as)
// with these defaults.  Round-1 kernel (tools/ab.py): C3 3.93 / 3.51 / 3.36 / 3.40 / 3.44 ms and C5 40.7 /
// 32.5 / 29.1 / 27.8 / 27.3 ms for 0 / 8 / 16 / 24 / 32 -- the larger the scene, the longer the walk
// relative to the shading a suspension repeats; today's kernel: C3 flat between 8 and 24, C5 best at 32
// (profiles/r02/tail2.log, tail3.log, tail_c5.log).
#ifndef RT_BVH_GRAB
#define RT_BVH_GRAB 256u   /* pixel slots per cursor atomic: a multiple of 64 (whole tiles) */
#endif
#ifndef RT_BVH_TAIL_SMALL
#define RT_BVH_TAIL_SMALL 16   /* 8-wave workgroups (scenes up to ~1300 spheres), frames in flight: 12 / 14 / 16 / 18 / 20 / 24 = 1.574 / 1.565 / 1.563 / 1.565 / 1.572 / 1.598 ms (profiles/r03/tail_sweep.log; round 2, slower steps: 20) */
#endif
#ifndef RT_BVH_TAIL_SERIAL
#define RT_BVH_TAIL_SERIAL 12  /* ... when one frame has the chip to itself */
#endif
#ifndef RT_BVH_TAIL_LARGE
#define RT_BVH_TAIL_LARGE 28   /* 16-wave workgroups and the global-memory form (C5: 20 / 24 / 28 / 32 / 40 = 18.16 / 18.02 / 17.98 / 18.05 / 18.61 ms) */
#endif

namespace rtk {

// LDS layout of a workgroup with the nodes staged (NLDS): the x/255 table first, then the links at byte
// address l0, the records at byte address 4 * l0 -- so that a record's address is its link's address
// shifted left by two, no base to add in the walk --, then the candidate lists.  l0 is the smallest
// 16-byte multiple >= 1024 (the table) whose quadruple clears the end of the link array.
__host__ __device__ inline uint32_t bvh_lds_l0(uint32_t n_nodes, uint32_t base) {
    // 4 * (base + l0) >= base + l0 + 4 * (n + 1)   <=>   3 * (base + l0) >= 4 * (n + 1)
    uint32_t l0 = 1024u;
    const uint32_t need = (4u * (n_nodes + 1u) + 2u) / 3u;
    if (base + l0 < need) l0 = need - base;
    return (l0 + 15u) & ~15u;
}
// The 4x rule leaves [1024, l0) unused: the `best` slots (512 bytes per wave) of as many waves as fit go there -- at C5 thirteen
// of sixteen.
__host__ __device__ inline uint32_t bvh_gap_waves(uint32_t n_nodes, uint32_t waves) {
    const uint32_t k = (bvh_lds_l0(n_nodes, 0u) - 1024u) / 512u;
    return k < waves ? k : waves;
}
// bytes of one wave's candidate lists: nodes in LDS: CAP rows of 64 two-byte entries and the spare row the pooling needs
// (trace_bvh: drain); nodes in global memory: CAP rows of 64 four-byte entries
__host__ __device__ constexpr uint32_t bvh_list_bytes(bool nlds, uint32_t cap) { return nlds ? 128u * (cap + 1u) : 256u * cap; }
__host__ __device__ inline uint32_t bvh_lds_lists(uint32_t n_nodes, uint32_t base) {   // byte offset of the lists
    const uint32_t l0 = bvh_lds_l0(n_nodes, base);
    return 4u * (base + l0) - base + 16u * ((n_nodes + 4u) & ~3u);
}

// ---- device: leaf records = the filter records prep_spheres wrote ------------------------------------
__global__ void bvh_fill_leaves(float4* rec, const uint32_t* link, uint32_t n_nodes, const float4* geo_f) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t l = link[i];
    if (l & 0x80000000u) rec[i] = geo_f[l & 0x7FFFFFFFu];
}

// ---- device: node bounds for moved spheres (same topology) ----------------------------------------------
// The hierarchy's TOPOLOGY -- which spheres a node covers: the links and leaf ids -- stays valid when
// spheres move; only the bounding spheres of the inner nodes must follow.  The reference rebuilds its
// top-level structure every frame (scene-raytracing.ts:138-143); here a frame that follows
// rt_write_spheres with an unchanged sphere count refits the inner nodes on the device -- one wave per
// node, the members (the leaves between the node and its skip link) spread over the lanes -- while the
// host rebuilds the topology for the new positions on a worker thread and hands it over when it is
// done (rt_api.hip: rt_rebuild).  Same construction as rt_bvh_build.h: centre of the members' box,
// shrink-wrapped (the centre walks towards the farthest member while the radius falls), stored in f32,
// radius taken about the STORED centre, times 1.04 (the slack the node test's proof needs, see the header),
// all in f64.  A node bound that contains its members is all exactness asks for.
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(256) void bvh_refit(float4* __restrict__ rec, const uint32_t* __restrict__ link,
                                                  uint32_t n_nodes, const float* __restrict__ records) {
    const uint32_t node = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (node >= n_nodes) return;
    const uint32_t lk = link[node];
    if (lk & 0x80000000u) return;                       // leaves are filled from the filter records (bvh_fill_leaves)
    const uint32_t first = node + 1u, end = lk >> 2;     // the subtree: nodes (node, end)
    auto sphere = [&](uint32_t j, double c[3], double& r) -> bool {     // member j of this lane's stride, if it is a leaf
        if (j >= end) return false;
        const uint32_t l = link[j];
        if (!(l & 0x80000000u)) return false;
        const float* s = records + 8u * (size_t)(l & 0x7FFFFFFFu);
        for (int a = 0; a < 3; ++a) { const double v = (double)s[a]; c[a] = v == v ? v : 0.0; }   // NaN orders as 0 (host build)
        const double rv = fabs((double)s[7]);
        r = rv == rv ? rv : 0.0;
        return true;
    };
    // radius about P (and which member is farthest), over all members
    auto radius_at = [&](const double P[3], double far[3]) -> double {
        double best = -1.0, bc[3] = {0.0, 0.0, 0.0};
        for (uint32_t j = first + lane; j < end; j += 64u) {
            double c[3], r;
            if (!sphere(j, c, r)) continue;
            const double dx = c[0] - P[0], dy = c[1] - P[1], dz = c[2] - P[2];
            const double d = sqrt(dx * dx + dy * dy + dz * dz) + r;
            if (d > best) { best = d; bc[0] = c[0]; bc[1] = c[1]; bc[2] = c[2]; }
        }
        const double R = wave_max_f64(best);
        // the farthest member's centre: the lowest lane that holds the maximum speaks
        const uint64_t who = __ballot(best == R);
        const int src = who ? (int)__builtin_ctzll(who) : 0;
        for (int a = 0; a < 3; ++a) far[a] = __shfl(bc[a], src, 64);
        return R;
    };
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = first + lane; j < end; j += 64u) {
        double c[3], r;
        if (!sphere(j, c, r)) continue;
        for (int a = 0; a < 3; ++a) { mn[a] = fmin(mn[a], c[a] - r); mx[a] = fmax(mx[a], c[a] + r); }
    }
    double P[3];
    for (int a = 0; a < 3; ++a) P[a] = 0.5 * (wave_min_f64(mn[a]) + wave_max_f64(mx[a]));
    double far[3];
    double Rp = radius_at(P, far);
    for (int it = 0; it < 32; ++it) {                   // wave-uniform: every quantity below is
        const double sx = far[0] - P[0], sy = far[1] - P[1], sz = far[2] - P[2];
        const double len = sqrt(sx * sx + sy * sy + sz * sz);
        if (!(len > 1e-12)) break;
        const double Q[3] = {P[0] + sx / len * 0.05 * Rp, P[1] + sy / len * 0.05 * Rp, P[2] + sz / len * 0.05 * Rp};
        double farq[3];
        const double Rq = radius_at(Q, farq);
        if (!(Rq < Rp)) break;
        P[0] = Q[0]; P[1] = Q[1]; P[2] = Q[2]; Rp = Rq;
        far[0] = farq[0]; far[1] = farq[1]; far[2] = farq[2];
    }
    const float C[3] = {(float)P[0], (float)P[1], (float)P[2]};     // the record stores the centre in f32:
    const double Cd[3] = {(double)C[0], (double)C[1], (double)C[2]};   // the radius is taken about THAT point
    double unused[3];
    double R = radius_at(Cd, unused);
    R *= RT_BVH_SIGMA;
    const double c2 = Cd[0] * Cd[0] + Cd[1] * Cd[1] + Cd[2] * Cd[2];
    const double k = c2 * (1.0 - (double)RT_FILTER_EPS) - R * R * (1.0 + (double)RT_FILTER_KAPPA);
    if (lane == 0)
        rec[node] = make_float4(C[0] * RT_FILTER_SCALE, C[1] * RT_FILTER_SCALE, C[2] * RT_FILTER_SCALE,
                                (float)(k * (double)RT_FILTER_SCALE2));
}

#ifdef RT_BVH_COUNT
#define g_steps steps_acc
__device__ unsigned long long g_wave_clock[3 * 8192];      // per wave: start, cursor exhausted, end (100 MHz ticks); mode 13
#endif
__device__ __forceinline__ float min0(float b) {      // min(b, 0) without the IEEE canonicalisation
    float d; asm("v_min_f32 %0, 0, %1" : "=v"(d) : "v"(b)); return d;
}

// literal test of one candidate, lexicographic (t, index) minimum (== in-order `t < nearest`)
__device__ __forceinline__ void exact_any_order(v3 center, float r2, int s, v3 o, v3 d, float fa, float ta,
                                                float& nearest, int& idx) {
    const v3 oc = sub(o, center);
    const float b = 2.0f * dot(d, oc);              // HK:309
    const float c = dot(oc, oc) - r2;               // HK:310
    const float disc = b * b - fa * c;              // HK:311
    if (disc > 0.0f && b < 0.0f) {                  // HK:316; b >= 0 gives t <= 0 (see exact_full)
        const float t = (-b - sqrtf(disc)) / ta;    // HK:317
        if (t > 0.001f && (t < nearest || (t == nearest && idx >= 0 && s < idx))) {   // HK:318
            nearest = t;
            idx = s;
        }
    }
}

// The walk of a SHADOW ray under the sign-aware node test runs backwards.  RK:147-165 casts the ray from the light L
// along s = normalize(P - L) towards the shaded point P and keeps the nearest hit; the result is `lit` only if that hit
// lies within 0.005 of P.  Spheres beyond P cannot change that (claim A in the header), so the walk need not visit
// them: it starts just behind P (w = P + delta s) and looks back along -s, where the sign-aware test drops every node
// behind its origin for free -- the far half of the scene, a fifth of the tests of a shadow ray (C3: 49.7 -> 39.4 per
// ray, candidates 3.2 -> 2.3).  The candidates are then evaluated as ever, literally, with the ray (L, s).  delta and
// the additive slack madd of the tested discriminant are derived in the header (claims A-C); l1 norms because they
// bound the Euclidean ones from above and cost three additions.
__device__ __forceinline__ void reversed_shadow_walk(bool shadow, v3 L, float light_l1, v3 P, v3 s, v3& wo, v3& wd, float& madd) {
    const v3 dl = sub(P, L);
    const float l1 = (__builtin_fabsf(dl.x) + __builtin_fabsf(dl.y)) + __builtin_fabsf(dl.z);
    const float la = l1 + light_l1;                                         // >= |P - L|, and >= |P| by the triangle inequality
    const float delta = __builtin_fmaf(la, RT_BVH_REV_DELTA_REL, RT_BVH_REV_DELTA);
    const float lb = l1 + delta;                                            // >= |w - L|
    const float slack = __builtin_fmaf(lb * lb, RT_BVH_REV_SLACK, (la * la) * RT_BVH_REV_SLACK_ABS);
    if (shadow) {
        wo = V(__builtin_fmaf(delta, s.x, P.x), __builtin_fmaf(delta, s.y, P.y), __builtin_fmaf(delta, s.z, P.z));
        wd = V(-s.x, -s.y, -s.z);
        madd = slack;
    }
}

// Advances the walk of every lane's ray (o, d).  R/L: node records and links (LDS or global),
// n: node count, geo: exact {c, r*r}.  slot: this lane's candidate column (nodes in LDS: two-byte entries, 128 bytes
// apart; nodes in global memory: four-byte entries, [k*64]).
// i: the lane's position in the node array, in and out: 0 starts a ray (the caller then also sets
// nearest = 9999, idx = -1, RK:172), n = no ray / walk complete.  TAIL > 0: once some lanes have
// completed and fewer than TAIL lanes are still walking, the call returns; the stragglers resume
// in the next call, next to the fresh rays of the lanes that completed -- the long tail of a
// wave's slowest rays no longer holds 64 lanes for a handful.
template <bool SGN, bool NLDS, int CAP, bool ROOMY>
__device__ __forceinline__ void trace_bvh(const uint32_t tail, const float4* __restrict__ R, const uint32_t* __restrict__ L, uint32_t n,
                                          const float4* __restrict__ geo, uint32_t* slot, unsigned long long* best,
                                          uint32_t& i, v3 o, v3 d, v3 wo, v3 wd, float madd, float& nearest, int& idx
#ifdef RT_BVH_COUNT
                                          , uint32_t& steps_acc
#endif
                                          ) {
    // (o, d): the ray the candidates are evaluated with, literally.  (wo, wd, madd): the ray the WALK selects them
    // with -- the same ray, madd = 0, except for shadow rays under the sign-aware test (reversed_shadow_walk).
    const float a = dot(wd, wd);         // HK:308
    const float inv = __builtin_amdgcn_rsqf(a) * (1.0f + RT_BVH_KAPPA);
    const v3 h = V(wd.x * inv, wd.y * inv, wd.z * inv);
    const v3 os = V(wo.x * RT_FILTER_SCALE, wo.y * RT_FILTER_SCALE, wo.z * RT_FILTER_SCALE);   // exact
    const v3 m = V(-2.0f * os.x, -2.0f * os.y, -2.0f * os.z);
    const float p = dot(h, os);
    const float q = __builtin_fmaf(-madd, RT_FILTER_SCALE2, dot(os, os) * (1.0f - RT_FILTER_EPS));
    // Sign-aware test on LDS nodes: min(b, 0) comes free as the `clamp` of the FMA that completes -b,
    // once everything is rescaled so that |b| < 1: -b * 2^-62 (|b| < 2^61 inside the filter's validity
    // range), and with it the squared quantities * 2^-124 -- powers of two, so every mantissa, hence every
    // decision, is the one of the unscaled test (the staged records carry w * 2^-124, see the kernel).
    // (v_min_f32 costs about two FMAs on this pipe.)
    constexpr bool CLAMPED = SGN && NLDS;
    constexpr float kT = 2.168404344971009e-19f;            // 2^-62
    constexpr float kT2 = 4.70197740328915e-38f;            // 2^-124
    const v3 hs = V(h.x * kT, h.y * kT, h.z * kT);
    const v3 ms = V(m.x * kT2, m.y * kT2, m.z * kT2);
    const float ps = -p * kT, qs = q * kT2;
    // this lane's candidate column in LDS: entries 128 (two-byte, L16 below) or 256 bytes apart; wa = LDS address of the next free one
    typedef __attribute__((address_space(3))) uint32_t* lds_u32_w;
    typedef __attribute__((address_space(3))) uint16_t* lds_u16_w;
    const uint32_t wa0 = (uint32_t)(uintptr_t)slot;
    uint32_t wa = wa0;

    // Literal evaluation of the wave's candidates, POOLED.  Candidate lists are ragged -- 2.6 entries per ray
    // on average, 6-9 in the longest list of a wave --, so evaluating "entry k of every lane" keeps under a
    // third of the lanes busy (C3: 6.3 M wave-iterations of ~63 instructions per frame for 118 M candidates,
    // a fifth of all VALU work).  Instead the wave compacts its lists into one pool in place (entry k of all
    // lanes, k = 0, 1, ...: positions from a ballot and mbcnt), every lane takes pool items i, i + 64, ...,
    // fetches the OWNER's ray with ds_bpermute, runs the reference's literal test (HK:308-318) and folds the
    // result into the owner's slot of `best` with an LDS 64-bit atomic min on (t bits, sphere index) -- the
    // lexicographic minimum the in-order `t < nearest` loop of the reference keeps (t > 0, so its bit
    // pattern orders like its value; "no hit yet" is index 0xFFFFFFFF, which loses every tie).
    // L16 (every form with its nodes in LDS: a sphere index fits 16 bits there): list entries are two bytes -- a lane's list is
    // twice as long in the same LDS, and the evaluation below is entered half as often (eight -> twelve entries per lane:
    // -5 % on scenes of 2000-3400 spheres, six -> eight: -10 %; profiles/r04/bvh_cap_sweep.log).  The pool's four-byte items
    // (owner, sphere) no longer fit where the rows they come from were, front to back; they are written from the END of the
    // wave's list area downwards while the rows are taken from the LAST one down -- the long rows are the sparse ones --,
    // one spare row above the lists guarantees room for the first, and when the pool would reach a row not yet read
    // (4 (total + row) > 128 (CAP + 1 - k) bytes: all lists full), what is pooled is evaluated first and the pool starts over.
    constexpr bool L16 = NLDS;
    typedef __attribute__((address_space(3))) const uint16_t* lds_u16;
    typedef __attribute__((address_space(3))) const uint32_t* lds_u32r;
    auto drain = [&]() {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t cnt = L16 ? (wa - wa0) >> 7 : (wa - wa0) >> 8;
        uint32_t* const pool = slot - lane;                              // (!L16) this wave's list area, as a linear array
        const uint32_t top = wa0 - 2u * lane + 128u * (uint32_t)(CAP + 1);   // (L16) LDS address of the end of the wave's list area
        uint32_t total = 0;                                              // wave-uniform
        uint32_t lim = cnt;                                              // (L16) rows [0, lim) of this lane's list are still to be taken
        uint32_t again = 0;                                              // (L16) wave-uniform: rows were left for another round
        best[lane] = ((unsigned long long)__float_as_uint(nearest) << 32) | (unsigned long long)(uint32_t)idx;
      for (;;) {
        if (L16) {
            // (unrolled: the row number is a constant of each copy; total, kstop are scalars the compiler keeps scalar)
            total = 0;
            uint32_t kstop = 0;
            bool stopped = false;
#pragma unroll
            for (int k = CAP - 1; k >= 0; --k) {
                const uint64_t m = __ballot((uint32_t)k < lim);
                if (m == 0ull) continue;
                const uint32_t need = (uint32_t)__popcll(m);
                if (!stopped && 4u * (total + need) > 128u * (uint32_t)(CAP + 1 - k)) { stopped = true; kstop = (uint32_t)k + 1u; }   // the pool would reach row k - 1
                if (stopped) continue;
                uint32_t e = 0;
                if ((uint32_t)k < lim) e = *(lds_u16)(uintptr_t)(wa0 + 128u * (uint32_t)k);      // every lane reads row k before any lane writes
                const uint32_t pos = total + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if ((uint32_t)k < lim) *(lds_u32_w)(uintptr_t)(top - 4u - 4u * pos) = (lane << 24) | e;
                total += need;
                asm volatile("" : "+s"(total));                        // (stays in a scalar register between the rows)
            }
            again = stopped ? 1u : 0u;
            if (stopped) lim = lim < kstop ? lim : kstop;                // rows from kstop up are in the pool now
        } else {
        for (uint32_t k = 0;; ++k) {
            const uint64_t m = __ballot(k < cnt);
            if (m == 0ull) break;
            uint32_t e = 0;
            if (k < cnt) e = slot[k * 64u];                              // every lane reads row k before any lane writes
            const uint32_t pos = total + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (k < cnt) pool[pos] = (lane << 24) | (e & 0x00FFFFFFu);   // pos < 64 (k + 1): rows > k are untouched
            total += (uint32_t)__popcll(m);
        }
        }
        for (uint32_t i = lane; __ballot(i < total) != 0ull; i += 64u) {
#ifdef RT_BVH_COUNT
            if (RT_BVH_COUNT == 4) g_steps += lane == 0u ? 1u : 0u;                    // drain iterations (wave)
            if (RT_BVH_COUNT == 5) g_steps += i < total ? 1u : 0u;                     // candidates evaluated (lane)
#endif
            const uint32_t e = i < total ? (L16 ? *(lds_u32r)(uintptr_t)(top - 4u - 4u * i) : pool[i]) : (lane << 24);
            const int owner = (int)(e >> 24);
            const int si = (int)(e & 0x00FFFFFFu);
            // The candidate's exact record is requested BEFORE the six shuffles, not behind them (the compiler sinks an
            // ordinary load into the branch that uses it); lanes without an item ask for sphere 0.  The wait is ours too.
            typedef float f4r __attribute__((ext_vector_type(4)));
            f4r gr;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(gr) : "v"((uint32_t)si << 4), "s"(geo) : "memory");
            const v3 oo = V(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
            const v3 od = V(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
            if (i < total) {
                float a2 = dot(od, od);                           // HK:308
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(gr), "+v"(a2) : : "memory");      // the record: needed from here on
                const float4 g = make_float4(gr.x, gr.y, gr.z, gr.w);
                const v3 oc = sub(oo, V(g.x, g.y, g.z));
                const float b = 2.0f * dot(od, oc);               // HK:309
                const float c = dot(oc, oc) - g.w;                // HK:310
                const float disc = b * b - (4.0f * a2) * c;       // HK:311
                if (disc > 0.0f && b < 0.0f) {                    // HK:316; b >= 0 gives t <= 0
                    const float t = (-b - sqrtf(disc)) / (2.0f * a2);   // HK:317
                    if (t > 0.001f && t < 9999.0f)                // HK:318 with tMin / the initial tMax of RK:315, RK:172
                        atomicMin(&best[owner], ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(uint32_t)si);
                }
            }
        }
        if (!L16 || __builtin_amdgcn_readfirstlane((int)again) == 0) break;
      }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned long long r = best[lane];
        nearest = __uint_as_float((uint32_t)(r >> 32));
        idx = (int)(uint32_t)r;
        wa = wa0;
    };

    // The walk's state j is an address.  Nodes staged in LDS: the LDS address of the node's RECORD, 16 x node index + 4 x
    // (the address of the link array); its link sits at j / 4 (bvh_lds_l0) and inner links are stored in this unit, so
    // between the compare that decides a step and the record load of the next there is ONE instruction, the select --
    // the shift that forms the link's address hangs off the side (round 3; before, the state was the link's address and
    // the shift sat in that chain).  Nodes in global memory: j = 4 x node index, the record at 4 j.  Node n is a sentinel that never passes and links to itself: a lane that is done
    // (or has no ray) idles on it without an exec-mask test per step; the wave leaves the loop
    // when every lane sits there.  Four steps per trip: the loop's own compares and branches are not
    // free on a pipe that is priced per opcode (C3 with frames in flight: 2, 3, 4 steps per trip = 2.22,
    // 2.13, 2.12 ms).
    typedef __attribute__((address_space(3))) const uint32_t* lds_u32;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const f4v* lds_f4;
    const uint32_t l0 = NLDS ? (uint32_t)(uintptr_t)L : 0u;                       // LDS address of L[0]
    // One step in three parts, so that a trip can issue the NEXT node's loads before it appends the current leaf: the
    // append is an exec-masked block of its own, and a wave issues in order -- with the loads behind it, every step
    // waited for six scalar and vector instructions that the next test does not depend on (round 3).
    auto fetch = [&](uint32_t j, float4& g, uint32_t& lk) {
        if (NLDS) {
            const f4v gv = *(lds_f4)(uintptr_t)j;
            lk = *(lds_u32)(uintptr_t)(j >> 2);                   // bvh_lds_l0: the records sit at 4 x the links' address
            g = make_float4(gv.x, gv.y, gv.z, gv.w);
        } else {
            lk = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(L) + j);
            g = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(R) + 4u * (size_t)j);
        }
    };
    auto passes = [&](const float4& g) -> bool {
        if (CLAMPED) {
            float nb;      // max(-b, 0) * 2^-62
            const float part = fma_vvv(hs.y, g.y, fma_vvv(hs.x, g.x, ps));
            asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nb) : "v"(hs.z), "v"(g.z), "v"(part));
            const float cp = fma_vvv(ms.z, g.z, fma_vvv(ms.y, g.y, fma_vvv(ms.x, g.x, g.w)));
            return __builtin_fmaf(nb, nb, -qs) > cp;
        } else {
            const float b = fnma_vvv(h.z, g.z, fnma_vvv(h.y, g.y, fnma_vvv(h.x, g.x, p)));
            const float cp = fma_vvv(m.z, g.z, fma_vvv(m.y, g.y, fma_vvv(m.x, g.x, g.w)));
            const float bm = SGN ? min0(b) : b;
            return __builtin_fmaf(bm, bm, -q) > cp;
        }
    };
    auto append = [&](uint32_t lk) {
        if (L16) {    // the store and the advance in place: the compiler forms the new address in a second register and moves it back
            *(lds_u16_w)(uintptr_t)wa = (uint16_t)lk;                    // the sphere index: the low half of a leaf's link
            asm("v_add_u32_e32 %0, 0x80, %0" : "+v"(wa));
        } else if (NLDS) {
            *(lds_u32_w)(uintptr_t)wa = lk;
            asm("v_add_u32_e32 %0, 0x100, %0" : "+v"(wa));
        } else {
            *(lds_u32_w)(uintptr_t)wa = lk;
            wa += 256u;
        }
    };
    const uint32_t jn = NLDS ? 4u * (4u * n + l0) : 4u * n;
    uint32_t j = NLDS ? 4u * (4u * i + l0) : 4u * i;
    // Loop control in the scalar unit, one exit test per trip: the wave leaves when no lane walks any more, or -- once
    // some lane has finished -- when fewer than `tail` still do (`tail` >= 1 covers the first case whenever the loop was
    // entered; tail == 0, "never suspend", leaves through the loop condition).
    const uint64_t walking0 = __ballot(j != jn);
    uint64_t walking = walking0;
    float4 g;
    uint32_t lk;
    fetch(j, g, lk);
    while (walking != 0ull) {
#ifdef RT_BVH_COUNT   // development statistics: 1 = wave iterations, 2 = lane tests (reported as "rays")
        if (RT_BVH_COUNT == 1) g_steps += (threadIdx.x & 63u) == 0u ? 2u : 0u;
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool pass = passes(g);
            const bool leaf = (int)lk < 0;
#ifdef RT_BVH_COUNT
            if (RT_BVH_COUNT == 2) g_steps += j != jn ? 1u : 0u;
            if (RT_BVH_COUNT == 6) g_steps += leaf ? 1u : 0u;                              // leaf tests (lane)
            if (RT_BVH_COUNT == 7) g_steps += (!leaf && pass) ? 1u : 0u;                   // inner nodes passed (lane)
            if (RT_BVH_COUNT == 8) g_steps += (!leaf && lk != j) ? 1u : 0u;                // inner nodes tested (lane; the sentinel links to itself)
#endif
            const uint32_t here = lk;
            j = (leaf || pass) ? j + (NLDS ? 16u : 4u) : lk;     // staged links are record addresses already
            fetch(j, g, lk);
            if (leaf && pass) append(here);
        }
        if (__builtin_expect(__ballot(wa >= wa0 + (L16 ? 128u : 256u) * (uint32_t)(CAP - 3)) != 0ull, 0)) {   // room for the four entries of the next trip
            drain();
            if (!ROOMY) fetch(j, g, lk);      // again, rather than five registers kept alive across the evaluation (80-VGPR forms: 10 spilled otherwise)
        }
        walking = __ballot(j != jn);
        if (walking != walking0 && (uint32_t)__builtin_popcountll(walking) < tail) break;
    }
    drain();
    i = NLDS ? ((j >> 2) - l0) >> 2 : j >> 2;
}

// ---- kernel ---------------------------------------------------------------------------------------------
// Do `k` workgroups with `bytes` of LDS each fit one CU?  The 160 KB are handed out in 128 granules of 1280 bytes and a
// workgroup's allocation is rounded up to whole granules -- hipOccupancyMaxActiveBlocksPerMultiprocessor divides the
// bytes and is one granule too generous: padding C3's 52,672-byte allocation, the third workgroup per CU stays up to
// 53,760 bytes (42 granules) and is gone at 53,824 (1.855 -> 2.31 ms per frame; tools/knob_ab.py, profiles/r02/lds_cliff.log).
// A scene just past such an edge must take the next form, not lose a third of its waves.
inline bool lds_fits(size_t k, size_t bytes) { return k * ((bytes + 1279u) / 1280u) <= 128u; }
// NLDS: node records and links staged in LDS (else read from global memory / L2: any scene size).
// 8-wave workgroups are held to 80 VGPRs (6 waves per SIMD, three workgroups per CU): the walk is a
// chain of dependent LDS reads, and the extra waves hide it (3.93 vs 4.40 ms at C3); 16-wave
// workgroups serve scenes whose nodes leave room for one workgroup per CU only (4 waves per SIMD).
// FLAT: compiled for a one-colour 1x1 sky (A.sky_flat; C1-C4) -- no cube filtering code in the kernel.
template <int WAVES, bool SGN, bool NLDS, int CAP, bool FLAT>
__global__ __launch_bounds__(64 * WAVES, WAVES == 16 ? 4 : 6) void bvh_pixels(const RtFrameArgs A) {
    extern __shared__ float4 lds[];
    const uint32_t n = A.bvh_nodes;               // the arrays hold n + 1 entries: [n] is the sentinel
    const uint32_t n4 = (n + 4u) & ~3u;
    // NLDS: [x/255 table | links at l0 | records at 4 * (address of the links) | lists]   (bvh_lds_l0)
    char* const lds_b = reinterpret_cast<char*>(lds);
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    const uint32_t l0off = bvh_lds_l0(n, base);
    uint32_t* sL = reinterpret_cast<uint32_t*>(lds_b + (NLDS ? l0off : 0u));
    float4* sR = reinterpret_cast<float4*>(lds_b + (NLDS ? 4u * (base + l0off) - base : 0u));
    uint32_t* lists = reinterpret_cast<uint32_t*>(lds_b + (NLDS ? bvh_lds_lists(n, base) : 1024u));
    (void)n4;
    // the host sized the allocation for dynamic LDS at address 0 (no static LDS in this kernel); anything else is
    // reported, not rendered: rt_wait fails the frame (rt_device.h: report_fault)
    if (NLDS && base != 0u) { report_fault(A.rays, 1ull); return; }
    if (NLDS)
        for (uint32_t i = threadIdx.x; i <= n; i += 64 * WAVES) {
            float4 r = A.bvh_rec[i];
            if (SGN) r.w *= 4.70197740328915e-38f;      // 2^-124: the clamped form of the node test (trace_bvh)
            sR[i] = r;
            const uint32_t lk = A.bvh_link[i];      // inner links become LDS addresses ...
            sL[i] = (int)lk < 0 ? lk : 4u * (lk + (uint32_t)(uintptr_t)sL);      // ... of the target's RECORD (trace_bvh)
        }
    const float4* R = NLDS ? sR : A.bvh_rec;
    const uint32_t* L = NLDS ? sL : A.bvh_link;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // a scalar, and so is all that derives from it
    constexpr uint32_t kListBytes = bvh_list_bytes(NLDS, (uint32_t)CAP);
    uint32_t* slot = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(lists) + wave * kListBytes + lane * (NLDS ? 2u : 4u));   // (NLDS: used as an address only)
    // per wave: 64 x (t bits, sphere index), the running nearest hits of the pooled literal evaluation (trace_bvh: drain);
    // the first bvh_gap_waves() waves' under the links, the others' behind the lists
    const uint32_t gapw = NLDS ? bvh_gap_waves(n, (uint32_t)WAVES) : 0u;
    unsigned long long* best = wave < gapw ? reinterpret_cast<unsigned long long*>(lds_b + 1024u) + wave * 64u
                                           : reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(lists) + WAVES * kListBytes) + (wave - gapw) * 64u;
    // rgba8unorm -> float table for the cube map texels: the reference's x / 255 division done
    // 256 times per workgroup instead of 12 times per sample
    float* lut = reinterpret_cast<float*>(lds_b);                          // the first KiB
    for (uint32_t i = threadIdx.x; i < 256u; i += 64 * WAVES) lut[i] = (float)i / 255.0f;
    __syncthreads();

    const Scene sc = unpack_scene(A);
    // FLAT: every face of the sky is the same single texel c, and a sample is c + (wu * 0 + wv * 0): c unless a weight is
    // NaN (rt_device.h: cube_sample<1> -- major axis, two IEEE divisions, a texel fetch per sample).  The directions this
    // kernel samples are all outputs of normalize() of vectors whose squared length cannot overflow (sums of three unit
    // vectors; fast mode keeps coordinates below 2^20): their components are finite and not all zero, or NaN, or -- |v|
    // underflowed to 0 -- inf / NaN with at least two of them not finite.  For such r the weights are NaN exactly when
    // (r.x * 0 + r.y * 0) + r.z * 0 is: c plus that has the general path's value (tests/test_shortcuts_cpu.py), with the
    // texel read once per wave instead of once per trip.
    v3 sky_c = V(0.0f, 0.0f, 0.0f);
    if (FLAT) {
        const v3 c = texel(A.face[0], 1, 1, 0, 0, lut);
        sky_c = V(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c.x))),
                  __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c.y))),
                  __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(c.z))));
    }
    auto flat_sky = [&](v3 r) -> v3 {
        const float z = (r.x * 0.0f + r.y * 0.0f) + r.z * 0.0f;
        return V(sky_c.x + z, sky_c.y + z, sky_c.z + z);
    };
    const float light_l1 = (__builtin_fabsf(sc.lightPos.x) + __builtin_fabsf(sc.lightPos.y)) + __builtin_fabsf(sc.lightPos.z);
    const uint32_t tiles_x = (A.W + 7u) / 8u;
    const uint32_t total = A.n_local_tiles * tiles_x * 64u;      // pixel slots, tile-major
    uint32_t cur = 0, end = 0;                                   // wave-uniform chunk cursor
    uint32_t chunk_ty = 0, chunk_tx = 0, chunk_first = ~0u;      // the tile the cursor is in
    // Pixel slots a wave reserves per atomic on the frame's cursor.  One 8x8 tile per atomic made the
    // cursor -- 130,000 returning atomics on ONE address per 4K frame, ~13 ns each -- a 1.7 ms floor
    // under the frame time (a 1-bounce frame took as long as a 2-bounce one: 1.62 ms; now 0.74 ms).
    constexpr uint32_t kGrab = RT_BVH_GRAB;
    const uint32_t plenty = gridDim.x * (uint32_t)WAVES * 64u * 16u;  // sixteen tiles per resident wave
    uint32_t grab = 64u, trips = 0u;                                  // wave-uniform
    bool exhausted = false;
#ifdef RT_BVH_COUNT
    const uint64_t clk0 = wall_clock64();      // 100 MHz
    uint64_t clk_x = 0;                        // when this wave found the cursor exhausted
#endif

    // per-lane path state (RK:101-144 unrolled into a state machine)
    bool active = false, shadow = false;
    uint32_t opix = 0, bounce = 0, nrays = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), color = V(1, 1, 1), fog = V(0, 0, 0);
    v3 normal = V(0, 0, 1), sdir = V(0, 0, 1), albedo = V(0, 0, 0);
    float dist = 0.0f, affect = 1.0f, sum = 0.0f, distance = 1.0f;
    uint32_t node = n;           // position of the lane's current ray in the node array; n: none
    float t = 9999.0f;           // nearest hit of the current ray so far (RK:172)
    int idx = -1;

    for (;;) {
        ++trips;
        // ---- idle lanes take the next pixels ----
        uint64_t idle = __ballot(!active);
        while (idle && !exhausted) {
            if (cur == end) {
                // One atomic reserves whole 8x8 tiles.  A wave that used up its last reservation within
                // two trips of the outer loop (cheap pixels: few bounces, sky) doubles the next one, up to
                // kGrab slots, while plenty of the frame is left; one that took more than eight trips goes
                // back to one tile: expensive pixels keep the finest grain, which balances the end of the frame.
                if (trips <= 2u && total - min(end, total) > plenty) grab = min(grab * 2u, kGrab);
                else if (trips > 8u) grab = 64u;
                trips = 0u;
                uint32_t base = 0;
#ifdef RT_BVH_COUNT
                const uint64_t clk_a = wall_clock64();
#endif
                if (lane == 0) base = atomicAdd(&A.qctrl[2], grab);
                base = __builtin_amdgcn_readfirstlane(base);
#ifdef RT_BVH_COUNT
                if (RT_BVH_COUNT == 14) nrays += lane == 0u ? 1u : 0u;                                  // cursor atomics (wave)
                if (RT_BVH_COUNT == 15) nrays += lane == 0u ? (uint32_t)(wall_clock64() - clk_a) : 0u;  // ticks spent on them
#endif
                if (base >= total) {
                    exhausted = true;
#ifdef RT_BVH_COUNT
                    clk_x = wall_clock64();
#endif
                    break;
                }
                cur = base;
                end = min(base + grab, total);
            }
            if ((cur & 63u) == 0u || cur == chunk_first) {       // entering a tile: decode it (wave-uniform)
                chunk_first = cur;
                chunk_ty = (cur >> 6) / tiles_x;
                chunk_tx = (cur >> 6) - chunk_ty * tiles_x;
            }
            const uint32_t tile_end = min(end, (cur & ~63u) + 64u);
            const uint32_t take = min((uint32_t)__popcll(idle), tile_end - cur);
            const uint32_t r = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!active && r < take) {
                const uint32_t l = (cur + r) & 63u;
                const uint32_t ty = chunk_ty, tx = chunk_tx;
                const uint32_t x = tx * 8u + (l & 7u), row = l >> 3;
                const uint32_t y = (A.tile_first + ty * A.tile_step) * 8u + row;
                if (x < A.W && y < A.H) {                        // RR:445: outside the texture: nothing
                    opix = (ty * 8u + row) * A.W + x;
                    ro = sc.cameraPos; rd = primary_dir(A, sc, x, y);
                    if (FLAT && sc.bounces == 0u) fog = scale(sc.minIntensity, flat_sky(rd));   // no ray will be cast
                    color = V(1.0f, 1.0f, 1.0f); dist = 0.0f;    // RK:102-103
                    affect = 1.0f; sum = 0.0f; bounce = 0u;      // RK:106-107
                    shadow = false;
                    active = true;
                    node = 0u; t = 9999.0f; idx = -1;            // primary ray
                    if (sc.bounces == 0u) node = n;              // RK:113: the loop body never runs
                }
            }
            cur += take;
            idle = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;

        bool finished = active && sc.bounces == 0u;
        bool missed = false;
        const bool walking = node != n;
        // the ray the walk selects candidates with (recomputed per trip from the path state: a suspended walk resumes
        // with the same values)
        v3 wo = ro, wd = shadow ? sdir : rd;
        float madd = 0.0f;
        if (SGN) reversed_shadow_walk(shadow, sc.lightPos, light_l1, ro, sdir, wo, wd, madd);
        else if (shadow) wo = sc.lightPos;
#ifdef RT_BVH_COUNT
        if (RT_BVH_COUNT == 3) nrays += lane == 0u ? 1u : 0u;                          // outer iterations (wave)
        trace_bvh<SGN, NLDS, CAP, WAVES == 16>(A.bvh_tail, R, L, n, A.geo, slot, best, node, shadow ? sc.lightPos : ro, shadow ? sdir : rd, wo, wd, madd, t, idx, nrays);
        if (walking && node == n) {
#else
        trace_bvh<SGN, NLDS, CAP, WAVES == 16>(A.bvh_tail, R, L, n, A.geo, slot, best, node, shadow ? sc.lightPos : ro, shadow ? sdir : rd, wo, wd, madd, t, idx);
        if (walking && node == n) {                                      // this lane's ray is complete
            ++nrays;
#endif
            const float next = affect + sum;                             // RK:120
            if (!shadow) {
                // The centre and colour of the sphere a reflection ray has hit are requested HERE, by hand, and awaited
                // where they are used: a wave issues in order, and the miss branch of the other lanes (three IEEE divisions)
                // stands in between -- cycles the two loads used to add to every trip instead of sharing.  Lanes without a
                // hit ask for sphere 0 and ignore the answer.
                typedef float f3r __attribute__((ext_vector_type(3)));
                f3r hit_g, hit_c;
                {
                    const uint32_t off = (uint32_t)(idx < 0 ? 0 : idx) << 4;
                    asm volatile("global_load_dwordx3 %0, %2, %3\n\tglobal_load_dwordx3 %1, %2, %4"
                                 : "=&v"(hit_g), "=&v"(hit_c) : "v"(off), "s"(A.geo), "s"(A.col) : "memory");
                }
                if (bounce == 0u) dist = idx >= 0 ? t : 0.0f;            // RK:116-118
                // One sky sample serves RK:124 (the ray missed) and RK:93-96 (the fog colour of
                // the pixel = the sky along the PRIMARY direction, which is rd at bounce 0).
                // A textured sky (!FLAT) is not sampled here at all: the lane leaves a record, sky_resolve does it.
                v3 sky = V(0, 0, 0);
                if (FLAT && (bounce == 0u || idx < 0)) sky = scale(sc.minIntensity, flat_sky(rd));
                if (FLAT && bounce == 0u) fog = sky;
                if (idx < 0) {                                           // RK:122-126
                    if (FLAT) color = divs(add(scale(sum, color), scale(affect, sky)), next);
                    else missed = true;
                    finished = true;
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(hit_g), "+v"(hit_c) : : "memory");
                if (idx >= 0) {
                    albedo = V(hit_c.x, hit_c.y, hit_c.z);
                    const v3 pos = add(ro, scale(t, rd));                    // RK:129
                    normal = normalize(sub(pos, V(hit_g.x, hit_g.y, hit_g.z)));   // HK:320
                    ro = pos;
                    rd = normalize(reflect(rd, normal));                     // RK:130
                    sdir = normalize(sub(ro, sc.lightPos));                  // RK:147
                    distance = length_of_unit(sdir);                         // RK:148
                    shadow = true;                                           // RK:153 next
                    node = 0u; t = 9999.0f; idx = -1;                        // shadow ray
                }
            } else {
                const float intensity = light_term(sc, ro, normal, sdir, distance, idx >= 0, t);
                const v3 blended = scale(intensity, albedo);                 // RK:133-135
                color = divs(add(scale(sum, color), scale(affect, blended)), next);   // RK:136
                affect = affect / 2.0f;                                      // RK:139
                sum = next;                                                  // RK:140
                ++bounce;
                shadow = false;
                finished = bounce >= sc.bounces;                             // RK:113
                if (!finished) { node = 0u; t = 9999.0f; idx = -1; }         // next reflection ray
            }
        }
        if (finished) {
            if (FLAT) {
                reinterpret_cast<uint32_t*>(A.out)[opix] = compose_pixel_sky(fog, color, dist);   // RK:91-98
            } else {
                // dist is +0 or a hit distance > 0.001: its sign bit is free for the flag
                A.fin[2u * opix] = make_float4(color.x, color.y, color.z, __uint_as_float(__float_as_uint(dist) | (missed ? 0x80000000u : 0u)));
                if (missed && bounce != 0u) A.fin[2u * opix + 1u] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(bounce));
            }
            active = false;
        }
    }
#ifdef RT_BVH_COUNT
    if (RT_BVH_COUNT == 9) nrays = lane == 0u ? (uint32_t)(wall_clock64() - (clk_x ? clk_x : clk0)) : 0u;   // 10 ns ticks after exhaustion, summed over waves
    if (RT_BVH_COUNT == 10) nrays = lane == 0u ? (uint32_t)(wall_clock64() - clk0) : 0u;                     // wave lifetimes
    if (RT_BVH_COUNT == 13 && lane == 0u) {
        const uint32_t w = (blockIdx.x * WAVES + wave) % 8192u;
        g_wave_clock[3 * w] = clk0; g_wave_clock[3 * w + 1] = clk_x ? clk_x : clk0; g_wave_clock[3 * w + 2] = wall_clock64();
    }
#endif
    count_rays(A.rays, nrays);
}

// Textured sky: the second half of every pixel of bvh_pixels<..., FLAT = false>.  In the state machine the
// seamless cube filter (four taps, edge folds, corner rule) ran once per trip of a wave's outer loop for
// however few lanes needed it, and cost the kernel 47 spilled VGPRs (C3 with a 6 x 512^2 sky: 2.78 ms against
// 1.91 with a flat one).  Here it runs pixel per lane, full waves, neighbouring directions: the same
// statements on the same values -- RK:93-96 (fog colour = sky along the primary ray), RK:122-126 (the miss
// blended into the running mean; affect and sum are rebuilt by the RK:139-140 recurrence from the bounce
// count), RK:91-98 (compose, pack) --, 68 bytes of HBM traffic per pixel.
__global__ __launch_bounds__(256) void sky_resolve(const RtFrameArgs A) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
    __syncthreads();
    const Scene sc = unpack_scene(A);
    const uint32_t slots = A.n_local_tiles * 8u * A.W;
    for (uint32_t opix = blockIdx.x * 256u + threadIdx.x; opix < slots; opix += gridDim.x * 256u) {
        const uint32_t lrow = opix / A.W, x = opix - lrow * A.W;
        const uint32_t y = (A.tile_first + (lrow >> 3) * A.tile_step) * 8u + (lrow & 7u);
        if (y >= A.H) continue;                                    // padding rows of the last tile: no pixel (RR:445)
        const float4 r0 = A.fin[2u * opix];
        const v3 fog = scale(sc.minIntensity, cube_sample<2>(A, primary_dir(A, sc, x, y), lut));
        v3 color = V(r0.x, r0.y, r0.z);
        const float dist = __uint_as_float(__float_as_uint(r0.w) & 0x7FFFFFFFu);
        if ((__float_as_uint(r0.w) >> 31) != 0u) {
            // a miss at bounce 0 left dist = +0 (RK:116-118) and no second record; later misses follow a hit at t > 0.001
            float4 r1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (dist != 0.0f) r1 = A.fin[2u * opix + 1u];
            const uint32_t k = min(__float_as_uint(r1.w), sc.bounces);   // a path has bounced fewer than maxBounces times when it misses
            float affect = 1.0f, sum = 0.0f;                        // RK:106-107
            for (uint32_t i = 0; i < k; ++i) { const float next = affect + sum; affect = affect / 2.0f; sum = next; }   // RK:120, 139-140
            const float next = affect + sum;
            // at bounce 0 the missing ray IS the primary ray: one sample serves both, as in the reference's flow
            const v3 sky = k == 0u ? fog : scale(sc.minIntensity, cube_sample<2>(A, V(r1.x, r1.y, r1.z), lut));
            color = divs(add(scale(sum, color), scale(affect, sky)), next);      // RK:124
        }
        reinterpret_cast<uint32_t*>(A.out)[opix] = compose_pixel_sky(fog, color, dist);
    }
}

template <int WAVES, bool SGN, bool NLDS, int CAP, int TAIL>
hipError_t launch_bvh_as(const RtFrameArgs& a0, size_t lds, hipStream_t s) {
    // The walk is left for the shading pass once fewer than `tail` lanes still walk.  A frame that has the
    // chip to itself prefers a lower threshold than frames that share it (C3, tools/knob_ab.py: one frame at a
    // time 8 / 12 / 16 / 20 lanes = 2.32 / 2.27 / 2.29 / 2.31 ms; in flight 16 / 20 / 24 = 1.864 / 1.855 / 1.861):
    // with one frame the end-of-frame tail is paid in full, and shorter trips shorten it.
    RtFrameArgs a = a0;
    a.bvh_tail = (uint32_t)TAIL;
    if (TAIL == RT_BVH_TAIL_SMALL && a.grid_share <= 1u) a.bvh_tail = RT_BVH_TAIL_SERIAL;
    // small shares (a rank of eight of a 4K frame: 1 M pixels) end sooner after they begin: 12 / 16 / 20 / 24 lanes =
    // 0.298 / 0.300 / 0.302 / 0.307 ms per frame in flight (profiles/r02/knobs_w8.log); round 3: 8 / 12 / 16 / 20 = 0.239 / 0.240 / 0.244 / 0.251
    else if (TAIL == RT_BVH_TAIL_SMALL && a.n_local_tiles * 8u * a.W < (1u << 22)) a.bvh_tail = 12u;
#ifdef RT_BVH_DEV_ENV
    if (const char* e = getenv("RT355_BVH_TAIL")) a.bvh_tail = (uint32_t)atoi(e);
#endif
    auto k = a.sky_flat ? bvh_pixels<WAVES, SGN, NLDS, CAP, true> : bvh_pixels<WAVES, SGN, NLDS, CAP, false>;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // resident workgroups per CU: the launch bounds allow 24 (8- and 12-wave) / 16 (16-wave) waves, LDS the rest
    uint32_t per_cu = WAVES == 8 ? 3u : (WAVES == 12 ? 2u : 1u);
    while (per_cu > 1u && !lds_fits(per_cu, lds)) --per_cu;
    const uint32_t pixels = a.n_local_tiles * ((a.W + 7u) / 8u) * 64u;
    uint32_t blocks = 256u * (per_cu ? per_cu : 1u);
    // Frames in flight share the chip: each takes ONE workgroup per CU (256), whatever their number.  Four
    // frames then oversubscribe the 768 resident slots: the workgroups of the youngest frame start as those
    // of the oldest drain, so no slot idles through a frame's tail.  Measured at C3, four frames in flight,
    // workgroups per frame 192 / 224 / 256 / 288 / 320 / 384: 2.31 / 2.42 / 2.22 / 2.50 / 2.52 / 2.50 ms per
    // frame.  Equal grids matter as much as their size: a first frame launched with all 768 workgroups makes
    // the next three start together when it ends, their tails then coincide for the rest of the batch
    // (2.50 ms); the host therefore asks for the share from the first frame on once it has seen frames
    // enqueued back to back (rt_api.hip: pipelined_hint).
    // (the 12- and 16-wave forms, two and one workgroup per CU, do best with exactly their share -- 1300 spheres,
    // 12-wave form, 96 / 128 / 160 / 256 workgroups per frame: 2.77 / 2.24 / 2.26 / 2.45 ms; 16-wave form 64 vs 256:
    // 3.40 vs 3.48 ms at 2500 spheres, no difference at C5)
    if (a.grid_share > 1u) blocks = WAVES == 8 ? std::max(256u, blocks / a.grid_share) : std::max(1u, blocks / a.grid_share);
    const uint32_t need = (pixels + 64u * WAVES - 1u) / (64u * WAVES);
    if (blocks > need) blocks = need;
#ifdef RT_BVH_DEV_ENV
    if (const char* e = getenv("RT355_BVH_BLOCKS")) if (atoi(e) > 0) blocks = std::min((uint32_t)atoi(e), need);
    if (const char* e = getenv("RT355_BVH_LDS_PAD")) {          // extra dynamic LDS: where does the third workgroup per CU go?
        lds += (size_t)atoi(e);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
#endif
    if (!a.sky_flat && !a.fin) return hipErrorInvalidValue;
    g_rt_kernel_id = !NLDS ? RT_KID_HIERARCHY_GLOBAL : (WAVES == 8 ? RT_KID_HIERARCHY_8 : (WAVES == 12 ? RT_KID_HIERARCHY_12 : RT_KID_HIERARCHY_16));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * WAVES), lds, s, a);
    if (!a.sky_flat) return rt_launch_sky_resolve(a, s);
    return hipGetLastError();
}

template <bool SGN>
hipError_t launch_bvh(const RtFrameArgs& a, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    constexpr int CAP = 12;
    // dynamic LDS starts at address 0 of the workgroup's allocation (the kernels declare no static LDS);
    // the kernel re-derives the layout from the address it actually gets
    const size_t nodes = (size_t)bvh_lds_lists(a.bvh_nodes, 0u);        // table + links + records (+ the gap the 4x rule leaves)
    const size_t cap = 160u * 1024u;
    // per wave: the candidate lists (bvh_list_bytes) + 64 eight-byte slots of running nearest hits, those of the first
    // bvh_gap_waves() waves inside `nodes`.  Lists of twelve entries per lane where they fit, of six where that keeps a form
    // with more waves per SIMD (or the LDS form at all).
    auto room = [&](uint32_t waves, uint32_t entries) {
        return nodes + (size_t)waves * bvh_list_bytes(true, entries) + (size_t)(waves - bvh_gap_waves(a.bvh_nodes, waves)) * 512u;
    };
#ifdef RT_BVH_DEV_ENV
    if (const char* e = getenv("RT355_BVH_CAP")) {      // how much do longer lists buy?  (8-wave form)
        if (atoi(e) == 16 && lds_fits(3u, room(8u, 16u))) return launch_bvh_as<8, SGN, true, 16, RT_BVH_TAIL_SMALL>(a, room(8u, 16u), s);
        if (atoi(e) == 8 && lds_fits(3u, room(8u, 8u))) return launch_bvh_as<8, SGN, true, 8, RT_BVH_TAIL_SMALL>(a, room(8u, 8u), s);
        if (atoi(e) == 10 && lds_fits(3u, room(8u, 10u))) return launch_bvh_as<8, SGN, true, 10, RT_BVH_TAIL_SMALL>(a, room(8u, 10u), s);
    }
#endif
    if (lds_fits(3u, room(8u, CAP))) return launch_bvh_as<8, SGN, true, CAP, RT_BVH_TAIL_SMALL>(a, room(8u, CAP), s);
    // two 12-wave workgroups per CU keep six waves per SIMD for scenes between the two forms
    if (lds_fits(2u, room(12u, CAP))) return launch_bvh_as<12, SGN, true, CAP, RT_BVH_TAIL_SMALL>(a, room(12u, CAP), s);
    if (lds_fits(2u, room(12u, 6u))) return launch_bvh_as<12, SGN, true, 6, RT_BVH_TAIL_SMALL>(a, room(12u, 6u), s);
#ifdef RT_BVH_DEV_ENV
    if (const char* e = getenv("RT355_BVH_CAP")) {      // how much do longer lists buy?  (16-wave form, scenes with room)
        if (atoi(e) == 16 && room(16u, 16u) <= cap) return launch_bvh_as<16, SGN, true, 16, RT_BVH_TAIL_LARGE>(a, room(16u, 16u), s);
        if (atoi(e) == 8 && room(16u, 8u) <= cap) return launch_bvh_as<16, SGN, true, 8, RT_BVH_TAIL_LARGE>(a, room(16u, 8u), s);
        if (atoi(e) == 6 && room(16u, 6u) <= cap) return launch_bvh_as<16, SGN, true, 6, RT_BVH_TAIL_LARGE>(a, room(16u, 6u), s);
    }
#endif
    if (room(16u, CAP) <= cap)     return launch_bvh_as<16, SGN, true, CAP, RT_BVH_TAIL_LARGE>(a, room(16u, CAP), s);
    if (room(16u, 6u) <= cap)      return launch_bvh_as<16, SGN, true, 6, RT_BVH_TAIL_LARGE>(a, room(16u, 6u), s);
    return launch_bvh_as<8, SGN, false, 8, RT_BVH_TAIL_LARGE>(a, 1024u + 8u * ((size_t)bvh_list_bytes(false, 8u) + 512u), s);
}

}  // namespace rtk

#ifdef RT_BVH_COUNT
extern "C" __attribute__((visibility("default"))) int rt_debug_wave_clock(unsigned long long* dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rtk::g_wave_clock), n * sizeof(unsigned long long));
}
#endif

hipError_t rt_launch_sky_resolve(const RtFrameArgs& a, hipStream_t s) {
    const uint32_t slots = a.n_local_tiles * 8u * a.W;
    if (slots == 0u) return hipSuccess;
    hipLaunchKernelGGL(rtk::sky_resolve, dim3(std::min((slots + 255u) / 256u, 256u * 8u)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t rt_launch_bvh(const RtFrameArgs& a, hipStream_t s) {
    return a.signed_filter ? rtk::launch_bvh<true>(a, s) : rtk::launch_bvh<false>(a, s);
}

hipError_t rt_launch_bvh_refit(float4* rec, const uint32_t* link, uint32_t n_nodes, const float* records, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::bvh_refit, dim3((n_nodes + 3u) / 4u), dim3(256), 0, s, rec, link, n_nodes, records);
    return hipGetLastError();
}

hipError_t rt_launch_bvh_fill(float4* rec, const uint32_t* link, uint32_t n_nodes, const float4* geo_f, hipStream_t s) {
    if (n_nodes == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::bvh_fill_leaves, dim3((n_nodes + 255u) / 256u), dim3(256), 0, s, rec, link, n_nodes, geo_f);
    return hipGetLastError();
}
