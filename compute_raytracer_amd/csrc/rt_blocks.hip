// rt_blocks.hip -- sphere scenes through the BLOCK form of the bounding-sphere hierarchy: the same frame as
// rt_bvh.hip, bit for bit, with a quarter of the dependent steps per ray.
//
// rt_bvh.hip walks a threaded tree one node per step: load the node, test it, take one of two addresses -- a ray
// is a chain of ~46 dependent LDS reads, and with six waves per SIMD the kernel still waits on that chain 39 % of
// its wave-cycles (profiles/r02/pmc/C3-fast-v0-n1__stall.csv).  Here a step handles a BLOCK (rt_blocks_build.h):
// the four children of one node -- four 16-byte records and four links, 80 bytes at one address, five independent
// reads -- and tests all four with the very filter of rt_bvh.hip (same records, same 8 FMAs + compare per test,
// same conservative margins: the proof in that file's header is about a node's bounding sphere and its members,
// not about the order nodes are visited in).  A ray is ~12 dependent steps.  What a step does with the results:
//
//   * LEAF block (entries are spheres): the ones that pass are appended to the lane's candidate column in LDS,
//     evaluated later with the reference's literal arithmetic, pooled over the wave -- rt_bvh.hip's drain, unchanged;
//   * INNER block (entries are child blocks): the ones that pass are pushed on the lane's stack -- a column of
//     16-bit block indices in LDS --, the last one pushed is taken at once; a block with nothing to descend into
//     pops.  The stack top is fetched at the START of the step, beside the block's own reads, so a pop adds no
//     round trip of its own;
//   * the stack is small (ROWS entries) and cannot overflow: a lane that finds no room for four more entries
//     stops using it and scans EVERY block once, in index order, appending what passes in leaf blocks (a sphere
//     may then be a candidate twice; the minimum over (t, index) does not care).  Exact, ~30 times slower, and
//     as rare as a ray that passes all four children of several nested nodes.
//
// The frame is rendered by the state machine of rt_bvh.hip (one persistent kernel, a path per lane, pixels from an
// atomic cursor, suspended walks): that part of the kernel is the same text with the walk exchanged.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "rt_filter.h"

#define RT_BVH_KAPPA 6.103515625e-05f   /* kappa_h = 2^-14, as rt_bvh.hip */
#ifndef RT_BLK_GRAB
#define RT_BLK_GRAB 256u
#endif

namespace rtk {

constexpr uint32_t kBlockBytes = 80u;

// LDS of a workgroup: [x/255 table 1 KiB | blocks 80 B each | candidate columns WAVES x CAP x 256 B |
//                      pooled-evaluation slots WAVES x 512 B | stacks WAVES x ROWS x 128 B]
__host__ __device__ inline uint32_t blk_lds_lists(uint32_t n_blocks) { return 1024u + kBlockBytes * n_blocks; }
__host__ __device__ inline size_t blk_lds_bytes(uint32_t n_blocks, uint32_t waves, uint32_t cap, uint32_t rows) {
    return (size_t)blk_lds_lists(n_blocks) + (size_t)waves * ((size_t)cap * 256u + 512u + (size_t)rows * 128u);
}

// ---- device: leaf records = the filter records prep_spheres wrote ------------------------------------
__global__ void blk_fill_leaves(float4* rec, const uint32_t* link, uint32_t n_entries, const float4* geo_f) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    const uint32_t l = link[i];
    if ((l & 0x80000000u) && l != 0xFFFFFFFFu) rec[i] = geo_f[l & 0x7FFFFFFFu];
}

// ---- device: bounds of the inner entries for moved spheres (same topology), as rt_bvh.hip: bvh_refit --------
__device__ __forceinline__ double blk_wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double blk_wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(256) void blk_refit(float4* __restrict__ rec, const uint32_t* __restrict__ link,
                                                  const uint32_t* __restrict__ sub_end, uint32_t n_entries,
                                                  const float* __restrict__ records) {
    const uint32_t e = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (e >= n_entries) return;
    const uint32_t lk = link[e];
    if (lk & 0x80000000u) return;                          // spheres and unused entries
    const uint32_t first = 4u * lk, end = 4u * sub_end[e];  // the subtree's entries: blocks [lk, sub_end)
    auto sphere = [&](uint32_t j, double c[3], double& r) -> bool {
        if (j >= end) return false;
        const uint32_t l = link[j];
        if (!(l & 0x80000000u) || l == 0xFFFFFFFFu) return false;
        const float* s = records + 8u * (size_t)(l & 0x7FFFFFFFu);
        for (int a = 0; a < 3; ++a) { const double v = (double)s[a]; c[a] = v == v ? v : 0.0; }
        const double rv = fabs((double)s[7]);
        r = rv == rv ? rv : 0.0;
        return true;
    };
    auto radius_at = [&](const double P[3], double far[3]) -> double {
        double best = -1.0, bc[3] = {0.0, 0.0, 0.0};
        for (uint32_t j = first + lane; j < end; j += 64u) {
            double c[3], r;
            if (!sphere(j, c, r)) continue;
            const double dx = c[0] - P[0], dy = c[1] - P[1], dz = c[2] - P[2];
            const double d = sqrt(dx * dx + dy * dy + dz * dz) + r;
            if (d > best) { best = d; bc[0] = c[0]; bc[1] = c[1]; bc[2] = c[2]; }
        }
        const double R = blk_wave_max(best);
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
    for (int a = 0; a < 3; ++a) P[a] = 0.5 * (blk_wave_min(mn[a]) + blk_wave_max(mx[a]));
    double far[3];
    double Rp = radius_at(P, far);
    for (int it = 0; it < 32; ++it) {
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
    const float C[3] = {(float)P[0], (float)P[1], (float)P[2]};
    const double Cd[3] = {(double)C[0], (double)C[1], (double)C[2]};
    double unused[3];
    double R = radius_at(Cd, unused);
    R *= 1.04;
    const double c2 = Cd[0] * Cd[0] + Cd[1] * Cd[1] + Cd[2] * Cd[2];
    const double k = c2 * (1.0 - (double)RT_FILTER_EPS) - R * R * (1.0 + (double)RT_FILTER_KAPPA);
    if (lane == 0)
        rec[e] = make_float4(C[0] * RT_FILTER_SCALE, C[1] * RT_FILTER_SCALE, C[2] * RT_FILTER_SCALE,
                             (float)(k * (double)RT_FILTER_SCALE2));
}

// The walk's per-lane state: `cur` = index of the block the lane handles next (0: the sentinel -- no ray, or the ray is
// complete), `sp` = LDS address of the top occupied row of its stack column (row 0 holds 0 and is never popped),
// `lin` = the lane has given up its stack and scans every block in index order.
struct BlkWalk { uint32_t cur, sp; bool lin; };

// Advances the walk of every lane's ray (o, d); the caller starts a ray with blk_start.  Returns when every lane is
// through, or -- TAIL -- when some have completed and fewer than `tail` still walk (rt_bvh.hip: suspended walks).
template <bool SGN, int CAP, int ROWS>
__device__ __forceinline__ void trace_blocks(const uint32_t tail, const uint32_t blk_base, const uint32_t n_blocks,
                                             const float4* __restrict__ geo, uint32_t* slot, unsigned long long* best,
                                             const uint32_t sp0, BlkWalk& w, v3 o, v3 d, float& nearest, int& idx
#ifdef RT_BLK_COUNT
                                             , uint32_t& cnt_acc
#endif
                                             ) {
    const float a = dot(d, d);           // HK:308
    const float inv = __builtin_amdgcn_rsqf(a) * (1.0f + RT_BVH_KAPPA);
    const v3 h = V(d.x * inv, d.y * inv, d.z * inv);
    const v3 os = V(o.x * RT_FILTER_SCALE, o.y * RT_FILTER_SCALE, o.z * RT_FILTER_SCALE);   // exact
    const v3 m = V(-2.0f * os.x, -2.0f * os.y, -2.0f * os.z);
    const float p = dot(h, os);
    const float q = dot(os, os) * (1.0f - RT_FILTER_EPS);
    // the sign-aware test through the FMA's `clamp`, everything rescaled by powers of two (rt_bvh.hip: trace_bvh)
    constexpr float kT = 2.168404344971009e-19f;            // 2^-62
    constexpr float kT2 = 4.70197740328915e-38f;            // 2^-124
    const v3 hs = V(h.x * kT, h.y * kT, h.z * kT);
    const v3 ms = V(m.x * kT2, m.y * kT2, m.z * kT2);
    const float ps = -p * kT, qs = q * kT2;
    typedef __attribute__((address_space(3))) uint32_t* lds_u32_w;
    typedef __attribute__((address_space(3))) uint16_t* lds_u16_w;
    typedef __attribute__((address_space(3))) const uint16_t* lds_u16;
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const f4v* lds_f4;
    typedef __attribute__((address_space(3))) const u4v* lds_u4;
    const uint32_t wa0 = (uint32_t)(uintptr_t)slot;
    uint32_t wa = wa0;

    // pooled literal evaluation of the wave's candidates: rt_bvh.hip's drain (the comments are there)
    auto drain = [&]() {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t cnt = (wa - wa0) >> 8;
        uint32_t* const pool = slot - lane;
        uint32_t total = 0;
        for (uint32_t k = 0;; ++k) {
            const uint64_t mk = __ballot(k < cnt);
            if (mk == 0ull) break;
            uint32_t e = 0;
            if (k < cnt) e = slot[k * 64u];
            const uint32_t pos = total + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
            if (k < cnt) pool[pos] = (lane << 24) | (e & 0x00FFFFFFu);
            total += (uint32_t)__popcll(mk);
        }
        best[lane] = ((unsigned long long)__float_as_uint(nearest) << 32) | (unsigned long long)(uint32_t)idx;
        for (uint32_t i = lane; __ballot(i < total) != 0ull; i += 64u) {
            const uint32_t e = i < total ? pool[i] : (lane << 24);
            const int owner = (int)(e >> 24);
            const int si = (int)(e & 0x00FFFFFFu);
            const v3 oo = V(__shfl(o.x, owner, 64), __shfl(o.y, owner, 64), __shfl(o.z, owner, 64));
            const v3 od = V(__shfl(d.x, owner, 64), __shfl(d.y, owner, 64), __shfl(d.z, owner, 64));
            if (i < total) {
                const float4 g = geo[si];
                const float a2 = dot(od, od);                     // HK:308
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
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const unsigned long long r = best[lane];
        nearest = __uint_as_float((uint32_t)(r >> 32));
        idx = (int)(uint32_t)r;
        wa = wa0;
    };

    auto test = [&](const f4v g) -> bool {                    // the node / leaf test of rt_bvh.hip (CLAMPED or plain form)
        if (SGN) {
            float nb;      // max(-b, 0) * 2^-62
            const float part = fma_vvv(hs.y, g.y, fma_vvv(hs.x, g.x, ps));
            asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nb) : "v"(hs.z), "v"(g.z), "v"(part));
            const float cp = fma_vvv(ms.z, g.z, fma_vvv(ms.y, g.y, fma_vvv(ms.x, g.x, g.w)));
            return __builtin_fmaf(nb, nb, -qs) > cp;
        }
        const float b = fnma_vvv(h.z, g.z, fnma_vvv(h.y, g.y, fnma_vvv(h.x, g.x, p)));
        const float cp = fma_vvv(m.z, g.z, fma_vvv(m.y, g.y, fma_vvv(m.x, g.x, g.w)));
        return __builtin_fmaf(b, b, -q) > cp;
    };

    const uint32_t sp_last = sp0 + 128u * (uint32_t)(ROWS - 1);     // the column's last row
    const uint64_t walking0 = __ballot(w.cur != 0u);
    for (;;) {
        const uint64_t walking = __ballot(w.cur != 0u);
        if (walking == 0ull) break;
        if (walking != walking0 && (uint32_t)__popcll(walking) < tail) break;
        // ---- one step: the block at `cur`, and the stack top in case the step ends in a pop ----
        const uint32_t addr = blk_base + w.cur * kBlockBytes;
        const uint32_t top = (uint32_t)*(lds_u16)(uintptr_t)w.sp;
        const f4v g0 = *(lds_f4)(uintptr_t)addr, g1 = *(lds_f4)(uintptr_t)(addr + 16u);
        const f4v g2 = *(lds_f4)(uintptr_t)(addr + 32u), g3 = *(lds_f4)(uintptr_t)(addr + 48u);
        const u4v lk = *(lds_u4)(uintptr_t)(addr + 64u);
        const bool p0 = test(g0), p1 = test(g1), p2 = test(g2), p3 = test(g3);
        const bool inner = (int)lk.x >= 0;                          // uniform blocks: entry 0 tells the type
        // spheres that pass: candidates
        if (p0 && !inner) { *(lds_u32_w)(uintptr_t)wa = lk.x; wa += 256u; }
        if (p1 && !inner) { *(lds_u32_w)(uintptr_t)wa = lk.y; wa += 256u; }
        if (p2 && !inner) { *(lds_u32_w)(uintptr_t)wa = lk.z; wa += 256u; }
        if (p3 && !inner) { *(lds_u32_w)(uintptr_t)wa = lk.w; wa += 256u; }
        // child blocks that pass: pushed; the last one is taken at once (its push is undone below)
        const uint32_t np = (p0 ? 1u : 0u) + (p1 ? 1u : 0u) + (p2 ? 1u : 0u) + (p3 ? 1u : 0u);
        if (inner && !w.lin && w.sp + 128u * np > sp_last) {          // no room: scan every block instead
            w.lin = true; w.sp = sp0; w.cur = 0u;
#ifdef RT_BLK_COUNT
            if (RT_BLK_COUNT == 1) ++cnt_acc;                             // lanes that gave up their stack
#endif
        }
#ifdef RT_BLK_COUNT
        if (RT_BLK_COUNT == 2) cnt_acc += w.cur != 0u || w.lin ? 1u : 0u;       // block steps with a live ray (lane)
        if (RT_BLK_COUNT == 3) cnt_acc += (threadIdx.x & 63u) == 0u ? 1u : 0u;  // block steps (wave)
        if (RT_BLK_COUNT == 4) cnt_acc += np;                                    // entries that pass
        if (RT_BLK_COUNT == 5) { const uint32_t dep = (w.sp - sp0) >> 7; cnt_acc = cnt_acc > dep ? cnt_acc : dep; }   // deepest stack (max, not a sum)
#endif
        const bool push = inner && !w.lin;
        uint32_t nxt = 0xFFFFFFFFu;
        if (p0 && push) { w.sp += 128u; *(lds_u16_w)(uintptr_t)w.sp = (uint16_t)lk.x; nxt = lk.x; }
        if (p1 && push) { w.sp += 128u; *(lds_u16_w)(uintptr_t)w.sp = (uint16_t)lk.y; nxt = lk.y; }
        if (p2 && push) { w.sp += 128u; *(lds_u16_w)(uintptr_t)w.sp = (uint16_t)lk.z; nxt = lk.z; }
        if (p3 && push) { w.sp += 128u; *(lds_u16_w)(uintptr_t)w.sp = (uint16_t)lk.w; nxt = lk.w; }
        if (w.lin) {
            w.cur = w.cur + 1u;                                       // (a lane that has just switched starts at block 1)
            if (w.cur >= n_blocks) { w.cur = 0u; w.lin = false; }
        } else {
            w.cur = nxt != 0xFFFFFFFFu ? nxt : top;                   // descend, or pop
            w.sp = max(w.sp - 128u, sp0);                             // the taken child's push undone / the popped row released
        }
        if (__ballot(wa >= wa0 + 256u * (uint32_t)(CAP - 4)) != 0ull) drain();      // room for the four entries of the next step
    }
    drain();
}

// ---- kernel: the state machine of rt_bvh.hip: bvh_pixels over the block walk ---------------------------------------
template <int WAVES, bool SGN, int CAP, int ROWS, bool FLAT>
__global__ __launch_bounds__(64 * WAVES, WAVES == 16 ? 4 : 6) void blk_pixels(const RtFrameArgs A) {
    extern __shared__ float4 lds[];
    const uint32_t nb = A.blk_blocks;
    char* const lds_b = reinterpret_cast<char*>(lds);
    const uint32_t base = (uint32_t)(uintptr_t)lds;
    const uint32_t blk_base = base + 1024u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* lists = reinterpret_cast<uint32_t*>(lds_b + blk_lds_lists(nb));
    uint32_t* slot = lists + wave * (uint32_t)(CAP * 64) + lane;
    unsigned long long* best = reinterpret_cast<unsigned long long*>(lists + WAVES * CAP * 64) + wave * 64u;
    uint16_t* stacks = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned long long*>(lists + WAVES * CAP * 64) + WAVES * 64u);
    uint16_t* stack = stacks + wave * (uint32_t)(ROWS * 64) + lane;
    const uint32_t sp0 = (uint32_t)(uintptr_t)stack;
    // blocks: records as they are (the sign-aware form takes w * 2^-124, rt_bvh.hip), links as they are
    for (uint32_t i = threadIdx.x; i < 4u * nb; i += 64 * WAVES) {
        float4 r = A.blk_rec[i];
        if (SGN) r.w *= 4.70197740328915e-38f;
        *reinterpret_cast<float4*>(lds_b + 1024u + kBlockBytes * (i >> 2) + 16u * (i & 3u)) = r;
        *reinterpret_cast<uint32_t*>(lds_b + 1024u + kBlockBytes * (i >> 2) + 64u + 4u * (i & 3u)) = A.blk_link[i];
    }
    stack[0] = 0u;                                                           // row 0: the sentinel, never popped
    float* lut = reinterpret_cast<float*>(lds_b);                            // the first KiB: x / 255
    for (uint32_t i = threadIdx.x; i < 256u; i += 64 * WAVES) lut[i] = (float)i / 255.0f;
    __syncthreads();

    const Scene sc = unpack_scene(A);
    const uint32_t tiles_x = (A.W + 7u) / 8u;
    const uint32_t total = A.n_local_tiles * tiles_x * 64u;
    uint32_t cur = 0, end = 0;
    uint32_t chunk_ty = 0, chunk_tx = 0, chunk_first = ~0u;
    constexpr uint32_t kGrab = RT_BLK_GRAB;
    const uint32_t plenty = gridDim.x * (uint32_t)WAVES * 64u * 16u;
    uint32_t grab = 64u, trips = 0u;
    bool exhausted = false;

    bool active = false, shadow = false;
    uint32_t opix = 0, bounce = 0, nrays = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), color = V(1, 1, 1), fog = V(0, 0, 0);
    v3 normal = V(0, 0, 1), sdir = V(0, 0, 1), albedo = V(0, 0, 0);
    float dist = 0.0f, affect = 1.0f, sum = 0.0f, distance = 1.0f;
    BlkWalk w; w.cur = 0u; w.sp = sp0; w.lin = false;
    float t = 9999.0f;
    int idx = -1;
    // a fresh ray: at the block of large spheres if the scene has one (the root waits on the stack), else at the root
    auto start_ray = [&]() {
        w.cur = A.blk_first; w.sp = sp0; w.lin = false;
        if (A.blk_then != 0u) { w.sp = sp0 + 128u; stack[64] = (uint16_t)A.blk_then; }
        t = 9999.0f; idx = -1;                                               // RK:172
    };

    for (;;) {
        ++trips;
        uint64_t idle = __ballot(!active);
        while (idle && !exhausted) {
            if (cur == end) {
                if (trips <= 2u && total - min(end, total) > plenty) grab = min(grab * 2u, kGrab);
                else if (trips > 8u) grab = 64u;
                trips = 0u;
                uint32_t b0 = 0;
                if (lane == 0) b0 = atomicAdd(&A.qctrl[2], grab);
                b0 = __builtin_amdgcn_readfirstlane(b0);
                if (b0 >= total) { exhausted = true; break; }
                cur = b0;
                end = min(b0 + grab, total);
            }
            if ((cur & 63u) == 0u || cur == chunk_first) {
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
                if (x < A.W && y < A.H) {                        // RR:445
                    opix = (ty * 8u + row) * A.W + x;
                    ro = sc.cameraPos; rd = primary_dir(A, sc, x, y);
                    if (FLAT && sc.bounces == 0u) fog = scale(sc.minIntensity, cube_sample<1>(A, rd, lut));
                    color = V(1.0f, 1.0f, 1.0f); dist = 0.0f;    // RK:102-103
                    affect = 1.0f; sum = 0.0f; bounce = 0u;      // RK:106-107
                    shadow = false;
                    active = true;
                    start_ray();                                 // primary ray
                    if (sc.bounces == 0u) w.cur = 0u;            // RK:113: the loop body never runs
                }
            }
            cur += take;
            idle = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;

        bool finished = active && sc.bounces == 0u;
        bool missed = false;
        const bool walking = w.cur != 0u;
#ifdef RT_BLK_COUNT
        trace_blocks<SGN, CAP, ROWS>(A.bvh_tail, blk_base, nb, A.geo, slot, best, sp0, w, shadow ? sc.lightPos : ro, shadow ? sdir : rd, t, idx, nrays);
        if (walking && w.cur == 0u) {
#else
        trace_blocks<SGN, CAP, ROWS>(A.bvh_tail, blk_base, nb, A.geo, slot, best, sp0, w, shadow ? sc.lightPos : ro, shadow ? sdir : rd, t, idx);
        if (walking && w.cur == 0u) {                                        // this lane's ray is complete
            ++nrays;
#endif
            const float next = affect + sum;                             // RK:120
            if (!shadow) {
                if (bounce == 0u) dist = idx >= 0 ? t : 0.0f;            // RK:116-118
                v3 sky = V(0, 0, 0);
                if (FLAT && (bounce == 0u || idx < 0)) sky = scale(sc.minIntensity, cube_sample<1>(A, rd, lut));
                if (FLAT && bounce == 0u) fog = sky;
                if (idx < 0) {                                           // RK:122-126
                    if (FLAT) color = divs(add(scale(sum, color), scale(affect, sky)), next);
                    else missed = true;
                    finished = true;
                } else {
                    const float4 g = A.geo[idx];
                    const float4 cl4 = A.col[idx];
                    albedo = V(cl4.x, cl4.y, cl4.z);
                    const v3 pos = add(ro, scale(t, rd));                    // RK:129
                    normal = normalize(sub(pos, V(g.x, g.y, g.z)));          // HK:320
                    ro = pos;
                    rd = normalize(reflect(rd, normal));                     // RK:130
                    sdir = normalize(sub(ro, sc.lightPos));                  // RK:147
                    distance = length(sdir);                                 // RK:148
                    shadow = true;                                           // RK:153 next
                    start_ray();                                             // shadow ray
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
                if (!finished) start_ray();                                  // next reflection ray
            }
        }
        if (finished) {
            if (FLAT) {
                reinterpret_cast<uint32_t*>(A.out)[opix] = compose_pixel_sky(fog, color, dist);   // RK:91-98
            } else {
                A.fin[2u * opix] = make_float4(color.x, color.y, color.z, __uint_as_float(__float_as_uint(dist) | (missed ? 0x80000000u : 0u)));
                if (missed && bounce != 0u) A.fin[2u * opix + 1u] = make_float4(rd.x, rd.y, rd.z, __uint_as_float(bounce));
            }
            active = false;
        }
    }
    count_rays(A.rays, nrays);
}

template <int WAVES, bool SGN, int CAP, int ROWS>
hipError_t launch_blocks_as(const RtFrameArgs& a0, uint32_t per_cu, uint32_t tail, hipStream_t s) {
    RtFrameArgs a = a0;
    a.bvh_tail = tail;
    if (a.grid_share <= 1u && WAVES != 16) a.bvh_tail = 12u;             // rt_bvh.hip: a frame alone on the chip prefers a lower threshold
    else if (WAVES != 16 && a.n_local_tiles * 8u * a.W < (1u << 22)) a.bvh_tail = 16u;
#ifdef RT_BVH_DEV_ENV
    if (const char* e = getenv("RT355_BVH_TAIL")) a.bvh_tail = (uint32_t)atoi(e);
#endif
    const size_t lds = blk_lds_bytes(a.blk_blocks, WAVES, CAP, ROWS);
    auto k = a.sky_flat ? blk_pixels<WAVES, SGN, CAP, ROWS, true> : blk_pixels<WAVES, SGN, CAP, ROWS, false>;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const uint32_t pixels = a.n_local_tiles * ((a.W + 7u) / 8u) * 64u;
    uint32_t blocks = 256u * per_cu;
    // frames in flight share the chip (rt_bvh.hip: launch_bvh_as): 8-wave workgroups one per CU and frame, the larger forms their exact share
    if (a.grid_share > 1u) blocks = WAVES == 8 ? std::max(256u, blocks / a.grid_share) : std::max(1u, blocks / a.grid_share);
    const uint32_t need = (pixels + 64u * WAVES - 1u) / (64u * WAVES);
    if (blocks > need) blocks = need;
#ifdef RT_BVH_DEV_ENV
    if (const char* e = getenv("RT355_BVH_BLOCKS")) if (atoi(e) > 0) blocks = std::min((uint32_t)atoi(e), need);
#endif
    if (!a.sky_flat && !a.fin) return hipErrorInvalidValue;
    g_rt_kernel_id = WAVES == 8 ? RT_KID_BLOCKS_8 : (WAVES == 12 ? RT_KID_BLOCKS_12 : RT_KID_BLOCKS_16);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * WAVES), lds, s, a);
    if (!a.sky_flat) return rt_launch_sky_resolve(a, s);
    return hipGetLastError();
}

inline bool blk_fits(size_t k, size_t bytes) { return k * ((bytes + 1279u) / 1280u) <= 128u; }   // LDS granules (rt_bvh.hip: lds_fits)

template <bool SGN>
hipError_t launch_blocks(const RtFrameArgs& a, hipStream_t s, bool* taken) {
    *taken = true;
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    const uint32_t nb = a.blk_blocks;
    if (blk_fits(3u, blk_lds_bytes(nb, 8, 8, 12)))   return launch_blocks_as<8, SGN, 8, 12>(a, 3u, 20u, s);
    if (blk_fits(2u, blk_lds_bytes(nb, 12, 8, 12)))  return launch_blocks_as<12, SGN, 8, 12>(a, 2u, 20u, s);
    if (blk_fits(2u, blk_lds_bytes(nb, 12, 6, 10)))  return launch_blocks_as<12, SGN, 6, 10>(a, 2u, 20u, s);
    if (blk_lds_bytes(nb, 16, 8, 12) <= 160u * 1024u) return launch_blocks_as<16, SGN, 8, 12>(a, 1u, 32u, s);
    if (blk_lds_bytes(nb, 16, 6, 10) <= 160u * 1024u) return launch_blocks_as<16, SGN, 6, 10>(a, 1u, 32u, s);
    *taken = false;                                          // beyond a CU's LDS: the threaded form reads its nodes from global memory
    return hipSuccess;
}

}  // namespace rtk

hipError_t rt_launch_blocks(const RtFrameArgs& a, hipStream_t s, bool* taken) {
    return a.signed_filter ? rtk::launch_blocks<true>(a, s, taken) : rtk::launch_blocks<false>(a, s, taken);
}

hipError_t rt_launch_blocks_refit(float4* rec, const uint32_t* link, const uint32_t* sub_end, uint32_t n_blocks, const float* records, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::blk_refit, dim3(n_blocks), dim3(256), 0, s, rec, link, sub_end, 4u * n_blocks, records);
    return hipGetLastError();
}

hipError_t rt_launch_blocks_fill(float4* rec, const uint32_t* link, uint32_t n_blocks, const float4* geo_f, hipStream_t s) {
    if (n_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::blk_fill_leaves, dim3((4u * n_blocks + 255u) / 256u), dim3(256), 0, s, rec, link, 4u * n_blocks, geo_f);
    return hipGetLastError();
}
