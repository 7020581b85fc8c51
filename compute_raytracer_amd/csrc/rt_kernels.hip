// rt_kernels.hip -- the gfx950 ray-trace kernels.  Compiled with -ffp-contract=off and
// -fno-slp-vectorize: every C++ floating-point expression below is evaluated as written
// (correctly rounded fp32, no fused multiply-add, no packed math), so it reproduces
// oracle/rt_oracle.c bit for bit.  The only fused operations are the explicit v_fma_f32 of the
// discriminant FILTER, whose result never reaches a pixel (see trace_filtered).
//
// What is computed follows the reference shader (citations relative to the reference repo):
//   RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl   main RK:73-99, rayColor
//        RK:101-144, lightIntensity RK:146-166
//   HK = src/rendering-raycast/shaders/heatmap-kernel.wgsl     hitSphere HK:307-331
//
// How it is computed is CDNA4-specific (measurements: tools/ubench2.hip, tools/ubench3.hip,
// numbers in DESIGN.md):
//   * sphere records are SoA float4 staged in LDS: one ds_read_b128 with a wave-uniform address
//     feeds one ray-sphere test of the whole wave (an SGPR operand would halve the FMA rate);
//   * only v_fma_f32 issues at ~2 cycles per wave on gfx950 (v_add/v_mul ~3.5, v_max/v_cmp ~4.5):
//     the per-sphere filter is written as v_fma_f32 only -- 9 for a ray with a per-lane origin,
//     6 for rays from the camera or the light, mask update included -- and leaves a per-lane BIT
//     MASK of the spheres that may be hit, built in the FMA pipe (clamp turns "positive" into 1.0;
//     Horner sum);
//   * rays that start at one point for all lanes (primary rays at the camera, shadow rays at
//     the light, RK:151) use records with `origin - center` and `c` precomputed per frame;
//   * exactness: the filter decides only "can this sphere have discriminant > 0 in front of
//     the origin"; it is conservative (margin 2^-16), and every sphere it lets through is
//     re-evaluated later with the reference's literal arithmetic, per lane in index order, so
//     nearest-hit selection, t, normals and colours are the oracle's bits;
//   * the candidates of a lane are queued in LDS and evaluated after the sphere loop with all
//     lanes busy (incoherent rays make almost every 16-sphere batch contain a candidate for
//     SOME lane; evaluating per batch would serialise the wave on 1-2 active lanes);
//   * two-kernel pipeline: bounce 0 per pixel (coherent, cheap hoisted forms), then persistent
//     waves that refill idle lanes from a path queue, so the expensive per-lane-origin trace
//     runs with full waves.
#include "rt_filter.h"

namespace rtk {


// Per-wave candidate list in LDS: CAP entries per lane, entry-major ([entry][lane]) so that the
// 64 lanes of a store hit 64 different banks.  Entry = (first sphere of the batch << 16) | 16-bit
// mask, bit (15-k) = sphere k of the batch.
template <int CAP>
struct CandList {
    uint32_t* slot;   // this lane's column: slot[i * 64]
};

// F: filter records (LDS, [N16]); Wx: exact 4th components ([N16], LDS or global).
// FULL: F = {center*2^30, r2f*2^60}, Wx = r*r.  Hoisted: F = {oc*2^30, cf*2^60}, Wx = c.
template <bool FULL, bool SGN, int CAP>
__device__ __forceinline__ void trace_filtered(const float4* __restrict__ F, const float* __restrict__ Wx,
                                               uint32_t N16, CandList<CAP> cl, v3 o, v3 d,
                                               float& nearest, int& idx) {
    const float a = dot(d, d);           // HK:308
    const float fa = 4.0f * a;           // the (4*a) of HK:311
    const float ta = 2.0f * a;           // HK:317
    const float inv = __builtin_amdgcn_rsqf(a) * (1.0f + RT_FILTER_KAPPA);
    RayF rf;
    rf.negone = opaque_negone();
    rf.h = V(d.x * inv, d.y * inv, d.z * inv);
    {
        const v3 os = V(o.x * RT_FILTER_SCALE, o.y * RT_FILTER_SCALE, o.z * RT_FILTER_SCALE);   // exact
        rf.m = V(-2.0f * os.x, -2.0f * os.y, -2.0f * os.z);
        rf.p = dot(rf.h, os);
        rf.q = dot(os, os) * (1.0f - RT_FILTER_EPS);
    }
    nearest = 9999.0f;                   // RK:172
    idx = -1;
    uint32_t cnt = 0;

    // literal evaluation of this lane's queued candidates, in index order
    auto drain = [&]() {
        uint32_t i = 0, bits = 0, base = 0;
        for (;;) {
            if (bits == 0u && i < cnt) {
                const uint32_t e = cl.slot[i * 64u];
                bits = e & 0xFFFFu;
                base = e >> 16;
                ++i;
            }
            if (__ballot(bits != 0u) == 0ull) break;
            if (bits != 0u) {
                const uint32_t lz = (uint32_t)__clz((int)bits);          // highest bit first = lowest index
                bits &= ~(0x80000000u >> lz);
                const int si = (int)(base + (lz - 16u));
                const float4 g = F[si];
                const float w = Wx[si];
                const v3 p = V(g.x * RT_FILTER_UNSCALE, g.y * RT_FILTER_UNSCALE, g.z * RT_FILTER_UNSCALE);   // exact
                if (FULL) exact_full<true>(p, w, si, o, d, fa, ta, nearest, idx);
                else      exact_hoisted<true>(p, w, si, d, fa, ta, nearest, idx);
            }
        }
        cnt = 0;
    };

    // software pipeline: the 8 ds_read_b128 of the next half-batch are in flight while the
    // current one is evaluated (the arrays carry 8 records of slack past N16)
    float4 g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = F[k];
    for (uint32_t s = 0; s < N16; s += 16u) {
        mask_acc code = mask_zero();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 gn[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) gn[k] = F[s + 8u * (half + 1) + k];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                code = shift_in(code, filter_one<FULL, SGN>(g[k], rf));
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = gn[k];
        }
        const uint32_t bits = mask_bits(code);
        if (bits != 0u) {
            cl.slot[cnt * 64u] = (s << 16) | bits;
            ++cnt;
        }
        if (__ballot(cnt >= (uint32_t)CAP) != 0ull) drain();
    }
    drain();
}

// Two rays per lane that share one origin (two primary rays, or two shadow rays): the hoisted
// filter needs only 6 FMAs per ray and sphere, less than the 16 cycles the LDS takes to deliver
// one record to four SIMDs; evaluating two rays per record read makes the loop VALU-bound.
template <bool SGN, int CAP>
__device__ __forceinline__ void trace_hoisted_pair(const float4* __restrict__ F, const float* __restrict__ Wx,
                                                   uint32_t N16, uint32_t* slot0, uint32_t* slot1,
                                                   v3 d0, v3 d1, bool act0, bool act1,
                                                   float& near0, int& idx0, float& near1, int& idx1) {
    const float a0 = dot(d0, d0), a1 = dot(d1, d1);                    // HK:308
    const float fa0 = 4.0f * a0, fa1 = 4.0f * a1;
    const float ta0 = 2.0f * a0, ta1 = 2.0f * a1;
    const float inv0 = __builtin_amdgcn_rsqf(a0) * (1.0f + RT_FILTER_KAPPA);
    const float inv1 = __builtin_amdgcn_rsqf(a1) * (1.0f + RT_FILTER_KAPPA);
    RayF r0, r1;
    r0.negone = r1.negone = -1.0f;      // unused by the hoisted form
    r0.h = V(d0.x * inv0, d0.y * inv0, d0.z * inv0);
    r1.h = V(d1.x * inv1, d1.y * inv1, d1.z * inv1);
    near0 = near1 = 9999.0f;                                           // RK:172
    idx0 = idx1 = -1;
    uint32_t cnt0 = 0, cnt1 = 0;

    auto drain = [&](uint32_t* slot, uint32_t& cnt, v3 d, float fa, float ta, float& nearest, int& idx) {
        uint32_t i = 0, bits = 0, base = 0;
        for (;;) {
            if (bits == 0u && i < cnt) {
                const uint32_t e = slot[i * 64u];
                bits = e & 0xFFFFu;
                base = e >> 16;
                ++i;
            }
            if (__ballot(bits != 0u) == 0ull) break;
            if (bits != 0u) {
                const uint32_t lz = (uint32_t)__clz((int)bits);
                bits &= ~(0x80000000u >> lz);
                const int si = (int)(base + (lz - 16u));
                const float4 g = F[si];
                const float w = Wx[si];
                const v3 p = V(g.x * RT_FILTER_UNSCALE, g.y * RT_FILTER_UNSCALE, g.z * RT_FILTER_UNSCALE);   // exact
                exact_hoisted<true>(p, w, si, d, fa, ta, nearest, idx);
            }
        }
        cnt = 0;
    };

    // quarter-batches of 4 records keep the register footprint of two rays per lane at 128
    float4 g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = F[k];
    for (uint32_t s = 0; s < N16; s += 16u) {
        mask_acc code0 = mask_zero(), code1 = mask_zero();
#pragma unroll
        for (int quarter = 0; quarter < 4; ++quarter) {
            float4 gn[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) gn[k] = F[s + 4u * (quarter + 1) + k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                code0 = shift_in(code0, filter_one<false, SGN>(g[k], r0));
                code1 = shift_in(code1, filter_one<false, SGN>(g[k], r1));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = gn[k];
        }
        const uint32_t bits0 = mask_bits(code0), bits1 = mask_bits(code1);
        if (act0 && bits0 != 0u) {
            slot0[cnt0 * 64u] = (s << 16) | bits0;
            ++cnt0;
        }
        if (act1 && bits1 != 0u) {
            slot1[cnt1 * 64u] = (s << 16) | bits1;
            ++cnt1;
        }
        if (__ballot(cnt0 >= (uint32_t)CAP || cnt1 >= (uint32_t)CAP) != 0ull) {
            drain(slot0, cnt0, d0, fa0, ta0, near0, idx0);
            drain(slot1, cnt1, d1, fa1, ta1, near1, idx1);
        }
    }
    drain(slot0, cnt0, d0, fa0, ta0, near0, idx0);
    drain(slot1, cnt1, d1, fa1, ta1, near1, idx1);
}

// ---- LDS layout ------------------------------------------------------------------------------------------
struct SceneLds {
    const float4 *Gf, *Lf, *Cf;     // filter records (or the exact records in strict mode)
    const float *Gw, *Lw, *Cw;      // exact 4th components
};

// ---- kernel: pixel per lane ------------------------------------------------------------------------------
// One pixel per lane, one 8x8 pixel tile per wave64 (= one WGSL workgroup, RK:73), WAVES tiles
// side by side per workgroup sharing one LDS copy of the scene.
// FILTER=false: RT_MODE_STRICT, literal loops over the exact records staged in LDS.
// FILTER=true : RT_MODE_FAST, filtered search.
// FIRST=true  : first stage of the two-kernel pipeline: bounce 0 only (primary ray + its shadow
//               ray, both hoisted forms, rays coherent inside the tile); pixels whose path ends
//               here are written, the others are appended to the path queue (one atomicAdd per
//               wave) for trace_paths.  Stages only the light and camera records.
// W_LDS       : exact 4th components in LDS too (else read from global when a candidate is evaluated).
//               In strict mode (no 4th-component arrays) the flag means "camera records in LDS":
//               false keeps them in global memory so that 4096 spheres still fit.
// GS          : "global scene": nothing is staged, the records are read from global memory (L2)
//               with wave-uniform addresses.  The fallback for scenes too large for a CU's LDS
//               (> 4608 spheres); the per-wave candidate lists still live in LDS.
// FLAT (all three pixel kernels): compiled for a one-colour 1x1 sky (A.sky_flat; the launch wrappers choose) -- the
// seamless cube filter is not instantiated, and its registers are not paid for inside the sphere loops.
template <int WAVES, bool FILTER, bool FIRST, bool SGN, bool W_LDS, int CAP, bool GS = false, bool FLAT = false>
__global__ __launch_bounds__(64 * WAVES) void trace_pixels(const RtFrameArgs A) {
    extern __shared__ float4 lds[];
    const uint32_t N = A.N, N16 = A.N16;
    const uint32_t n = GS ? 0u : (FILTER ? N16 : N);
    // float4 arrays first, then the float arrays, then the candidate lists
    constexpr bool CAM_LDS = FILTER || W_LDS;
    const float4* sL = GS ? (FILTER ? A.lgt_f : A.lgt) : lds;
    const float4* sC = GS ? (FILTER ? A.cam_f : A.cam) : lds + n;
    const float4* sG = GS ? (FILTER ? A.geo_f : A.geo) : lds + (CAM_LDS ? 2 : 1) * n;       // unused when FIRST
    float* wbase = reinterpret_cast<float*>(lds + (FIRST ? 2 : 3) * n);
    float* sLw = wbase;
    float* sCw = wbase + n;
    float* sGw = wbase + 2 * n;
    uint32_t* lists = reinterpret_cast<uint32_t*>(wbase + ((FILTER && W_LDS) ? (FIRST ? 2 : 3) * n : 0));
    for (uint32_t i = threadIdx.x; i < n; i += 64 * WAVES) {
        lds[i] = FILTER ? A.lgt_f[i] : A.lgt[i];
        if (CAM_LDS) lds[n + i] = FILTER ? A.cam_f[i] : A.cam[i];
        if (!FIRST) lds[(CAM_LDS ? 2 : 1) * n + i] = FILTER ? A.geo_f[i] : A.geo[i];
        if (FILTER && W_LDS) {
            sLw[i] = A.lgt_w[i];
            sCw[i] = A.cam_w[i];
            if (!FIRST) sGw[i] = A.geo_w[i];
        }
    }
    __syncthreads();
    const float* Lw = (FILTER && W_LDS && !GS) ? sLw : A.lgt_w;
    const float* Cw = (FILTER && W_LDS && !GS) ? sCw : A.cam_w;
    const float* Gw = (FILTER && W_LDS && !GS) ? sGw : A.geo_w;
    const float4* camE = (CAM_LDS || GS) ? sC : A.cam;      // strict mode only

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    CandList<CAP> cl;
    cl.slot = lists + wave * (uint32_t)(CAP * 64) + lane;

    const uint32_t x = blockIdx.x * (8u * WAVES) + wave * 8u + (lane & 7u);
    const uint32_t row = lane >> 3;
    const uint32_t y = (A.tile_first + blockIdx.y * A.tile_step) * 8u + row;
    if (x >= A.W || y >= A.H) return;        // RR:445: threads outside the texture store nothing

    const Scene sc = unpack_scene(A);
    const v3 dir0 = primary_dir(A, sc, x, y);

    // RK:101-144
    uint32_t nrays = 0;
    float dist = 0.0f;
    v3 color = V(1.0f, 1.0f, 1.0f);
    v3 ro = sc.cameraPos, rd = dir0;
    float affect = 1.0f, sum = 0.0f;
    bool alive = true;
    const uint32_t nb = FIRST ? (sc.bounces ? 1u : 0u) : sc.bounces;
    for (uint32_t bounce = 0; bounce < nb; ++bounce) {
        float t; int idx;
        if (FILTER) {
            if (FIRST || bounce == 0) trace_filtered<false, SGN, CAP>(sC, Cw, N16, cl, ro, rd, t, idx);
            else                      trace_filtered<true, SGN, CAP>(sG, Gw, N16, cl, ro, rd, t, idx);
        } else {
            if (FIRST || bounce == 0) trace_literal<false>(camE, N, ro, rd, t, idx);
            else                      trace_literal<true>(sG, N, ro, rd, t, idx);
        }
        ++nrays;
        const bool hit = idx >= 0;
        if (bounce == 0) dist = hit ? t : 0.0f;                  // RK:116-118 (zero-initialised state)
        const float next = affect + sum;                         // RK:120
        if (!hit) {                                              // RK:122-126
            const v3 sky = scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, rd));
            color = divs(add(scale(sum, color), scale(affect, sky)), next);
            alive = false;
            break;
        }
        const float4 g = A.geo[idx];
        const float4 cl4 = A.col[idx];
        const v3 pos = add(ro, scale(t, rd));                    // HK:319 == RK:129
        const v3 normal = normalize(sub(pos, V(g.x, g.y, g.z))); // HK:320
        ro = pos;
        rd = normalize(reflect(rd, normal));                     // RK:130

        // RK:146-153 lightIntensity(ro, normal): shadow ray from the light
        const v3 sdir = normalize(sub(ro, sc.lightPos));         // RK:147
        const float distance = length(sdir);                     // RK:148
        float st; int sidx;
        if (FILTER) trace_filtered<false, SGN, CAP>(sL, Lw, N16, cl, sc.lightPos, sdir, st, sidx);
        else        trace_literal<false>(sL, N, sc.lightPos, sdir, st, sidx);
        ++nrays;
        const float intensity = light_term(sc, ro, normal, sdir, distance, sidx >= 0, st);
        const v3 blended = scale(intensity, V(cl4.x, cl4.y, cl4.z)); // RK:133-135, diffuse.w == 1
        color = divs(add(scale(sum, color), scale(affect, blended)), next); // RK:136
        affect = affect / 2.0f;                                  // RK:139
        sum = next;                                              // RK:140
    }

    const uint32_t opix = (blockIdx.y * 8u + row) * A.W + x;           // index in the compact tile buffer
    bool cont = false;
    if (FIRST) {
        // the path goes on (bounce 1..) in trace_paths: append {ro, pixel}, {rd, dist}, {color}
        cont = alive && sc.bounces > 1u;
        const uint64_t m = __ballot(cont);
        if (m) {
            const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&A.qctrl[0], (uint32_t)__popcll(m));
            base = __shfl(base, (int)leader, 64);
            if (cont) {
                float4* q = A.queue + 3u * (size_t)(base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
                q[0] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(opix));
                q[1] = make_float4(rd.x, rd.y, rd.z, dist);
                q[2] = make_float4(color.x, color.y, color.z, 0.0f);
            }
        }
    }
    if (!cont) reinterpret_cast<uint32_t*>(A.out)[opix] = compose_pixel_sky(scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, dir0)), color, dist);   // RK:91-98
    count_rays(A.rays, nrays);
}

// ---- kernel: first stage of the pipeline, bounce 0 of every pixel ---------------------------------
// Persistent waves (the scene is staged into LDS once per workgroup, not once per 8 tiles); every
// wave takes pairs of horizontally adjacent 8x8 tiles from an atomic counter, two pixels per lane
// (x, y) and (x+8, y), and runs the primary rays and then the shadow rays of both pixels through
// trace_hoisted_pair.  Pixels whose path ends at bounce 0 (miss, or maxBounces == 1) are written;
// the others are appended to the path queue for trace_paths.
template <int WAVES, bool SGN, bool W_LDS, int CAP, bool FLAT>
__global__ __launch_bounds__(64 * WAVES, 4) void first_bounce(const RtFrameArgs A) {   // 4 waves/SIMD: <= 128 VGPRs
    extern __shared__ float4 lds[];
    const uint32_t N16 = A.N16;
    float4* sL = lds;
    float4* sC = lds + N16;
    float* sLw = reinterpret_cast<float*>(lds + 2 * N16);
    float* sCw = sLw + N16;
    uint32_t* lists = reinterpret_cast<uint32_t*>(sLw + (W_LDS ? 2 * N16 : 0));
    for (uint32_t i = threadIdx.x; i < N16; i += 64 * WAVES) {
        sL[i] = A.lgt_f[i];
        sC[i] = A.cam_f[i];
        if (W_LDS) { sLw[i] = A.lgt_w[i]; sCw[i] = A.cam_w[i]; }
    }
    __syncthreads();
    const float* Lw = W_LDS ? sLw : A.lgt_w;
    const float* Cw = W_LDS ? sCw : A.cam_w;

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* slot0 = lists + wave * (uint32_t)(2 * CAP * 64) + lane;
    uint32_t* slot1 = slot0 + CAP * 64;

    const Scene sc = unpack_scene(A);
    const uint32_t pairs_x = (A.W + 15u) / 16u;
    const uint32_t total_pairs = pairs_x * A.n_local_tiles;
    uint32_t nrays = 0;

    for (;;) {
        uint32_t pair = 0;
        if (lane == 0) pair = atomicAdd(&A.qctrl[2], 1u);
        pair = __builtin_amdgcn_readfirstlane(pair);
        if (pair >= total_pairs) break;
        const uint32_t ty = pair / pairs_x, px = pair - ty * pairs_x;
        const uint32_t row = lane >> 3;
        const uint32_t y = (A.tile_first + ty * A.tile_step) * 8u + row;
        const uint32_t xs[2] = {px * 16u + (lane & 7u), px * 16u + 8u + (lane & 7u)};
        bool in[2], hit[2] = {false, false};
        v3 dir0[2], color[2], ro[2], rd[2], normal[2], sdir[2], albedo[2];
        float dist[2], distance[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            in[j] = xs[j] < A.W && y < A.H;                        // RR:445: outside the texture: nothing
            dir0[j] = primary_dir(A, sc, in[j] ? xs[j] : 0u, in[j] ? y : 0u);
            color[j] = V(1.0f, 1.0f, 1.0f);                         // RK:103
            dist[j] = 0.0f;                                         // RK:102
            ro[j] = sc.cameraPos; rd[j] = dir0[j];
            normal[j] = sdir[j] = albedo[j] = V(0.0f, 0.0f, 1.0f);
            distance[j] = 1.0f;
        }
        const bool trace = sc.bounces > 0u;
        if (trace) {
            // ---- RK:114, bounce 0: primary rays (origin = camera for every lane) ----
            float t[2]; int idx[2];
            trace_hoisted_pair<SGN, CAP>(sC, Cw, N16, slot0, slot1, rd[0], rd[1], in[0], in[1], t[0], idx[0], t[1], idx[1]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!in[j]) continue;
                ++nrays;
                hit[j] = idx[j] >= 0;
                dist[j] = hit[j] ? t[j] : 0.0f;                                    // RK:116-118
                if (!hit[j]) {                                                     // RK:122-126, next = 1 + 0
                    const v3 sky = scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, rd[j]));
                    color[j] = divs(add(scale(0.0f, color[j]), scale(1.0f, sky)), 1.0f);
                } else {
                    const float4 g = A.geo[idx[j]];
                    const float4 cl4 = A.col[idx[j]];
                    albedo[j] = V(cl4.x, cl4.y, cl4.z);
                    const v3 pos = add(ro[j], scale(t[j], rd[j]));                 // RK:129
                    normal[j] = normalize(sub(pos, V(g.x, g.y, g.z)));             // HK:320
                    ro[j] = pos;
                    rd[j] = normalize(reflect(rd[j], normal[j]));                  // RK:130
                    sdir[j] = normalize(sub(ro[j], sc.lightPos));                  // RK:147
                    distance[j] = length(sdir[j]);                                 // RK:148
                }
            }
            // ---- RK:153: shadow rays of the pixels that hit (origin = light) ----
            if (__ballot(hit[0] || hit[1]) != 0ull) {
                float st[2]; int sidx[2];
                trace_hoisted_pair<SGN, CAP>(sL, Lw, N16, slot0, slot1, sdir[0], sdir[1], hit[0], hit[1],
                                             st[0], sidx[0], st[1], sidx[1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!hit[j]) continue;
                    ++nrays;
                    const float intensity = light_term(sc, ro[j], normal[j], sdir[j], distance[j], sidx[j] >= 0, st[j]);
                    const v3 blended = scale(intensity, albedo[j]);                // RK:133-135
                    color[j] = divs(add(scale(0.0f, color[j]), scale(1.0f, blended)), 1.0f);   // RK:136, sum 0, affect 1
                }
            }
        }
        // ---- hand over or finish ----
        bool cont[2];
        uint32_t opix[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            opix[j] = (ty * 8u + row) * A.W + xs[j];
            cont[j] = in[j] && hit[j] && sc.bounces > 1u;
        }
        const uint64_t m0 = __ballot(cont[0]), m1 = __ballot(cont[1]);
        if (m0 | m1) {
            const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.qctrl[0], n0 + n1);
            base = __builtin_amdgcn_readfirstlane(base);
            const uint64_t below = (1ull << lane) - 1ull;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!cont[j]) continue;
                const uint32_t e = base + (j ? n0 + (uint32_t)__popcll(m1 & below) : (uint32_t)__popcll(m0 & below));
                float4* q = A.queue + 3u * (size_t)e;
                q[0] = make_float4(ro[j].x, ro[j].y, ro[j].z, __uint_as_float(opix[j]));
                q[1] = make_float4(rd[j].x, rd[j].y, rd[j].z, dist[j]);
                q[2] = make_float4(color[j].x, color[j].y, color[j].z, 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (in[j] && !cont[j])
                reinterpret_cast<uint32_t*>(A.out)[opix[j]] = compose_pixel_sky(scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, dir0[j])), color[j], dist[j]);   // RK:91-98
    }
    count_rays(A.rays, nrays);
}

// ---- kernel: second stage of the pipeline, bounces 1.. with path regeneration -------------------
// Persistent waves: every lane carries one path; a lane whose path ends (miss RK:122-126, or
// the bounce limit) takes the next entry of the path queue, so the per-lane-origin trace (the
// expensive one, 12 FMAs per sphere) runs with all 64 lanes busy while the queue lasts.
// Entries are popped in chunks of 64 with one atomicAdd per chunk; neighbouring entries come
// from the same 8x8 tile of the first stage.  There is no inter-workgroup dependency: a
// workgroup that finds the queue empty exits.
template <int WAVES, bool SGN, bool W_LDS, int CAP, bool FLAT>
__global__ __launch_bounds__(64 * WAVES) void trace_paths(const RtFrameArgs A) {
    extern __shared__ float4 lds[];
    const uint32_t N16 = A.N16;
    float4* sG = lds;
    float4* sL = lds + N16;
    float* sGw = reinterpret_cast<float*>(lds + 2 * N16);
    float* sLw = sGw + N16;
    uint32_t* lists = reinterpret_cast<uint32_t*>(sGw + (W_LDS ? 2 * N16 : 0));
    for (uint32_t i = threadIdx.x; i < N16; i += 64 * WAVES) {
        sG[i] = A.geo_f[i];
        sL[i] = A.lgt_f[i];
        if (W_LDS) { sGw[i] = A.geo_w[i]; sLw[i] = A.lgt_w[i]; }
    }
    __syncthreads();
    const float* Gw = W_LDS ? sGw : A.geo_w;
    const float* Lw = W_LDS ? sLw : A.lgt_w;

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    CandList<CAP> cl;
    cl.slot = lists + wave * (uint32_t)(CAP * 64) + lane;

    const Scene sc = unpack_scene(A);
    const uint32_t total = A.qctrl[0];           // written by the first stage (previous kernel)
    uint32_t cur = 0, end = 0;                    // wave-uniform chunk cursor
    bool exhausted = false;

    bool active = false;
    uint32_t opix = 0, bounce = 0, nrays = 0;
    v3 ro = V(0, 0, 0), rd = V(0, 0, 1), color = V(0, 0, 0);
    float dist = 0.0f, affect = 0.0f, sum = 0.0f;

    for (;;) {
        // ---- refill idle lanes from the queue ----
        uint64_t idle = __ballot(!active);
        while (idle && !exhausted) {
            if (cur == end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&A.qctrl[1], 64u);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total) { exhausted = true; break; }
                cur = base;
                end = min(base + 64u, total);
            }
            const uint32_t nidle = (uint32_t)__popcll(idle);
            const uint32_t take = min(nidle, end - cur);
            const uint32_t r = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!active && r < take) {
                const float4* q = A.queue + 3u * (size_t)(cur + r);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                ro = V(q0.x, q0.y, q0.z); opix = __float_as_uint(q0.w);
                rd = V(q1.x, q1.y, q1.z); dist = q1.w;
                color = V(q2.x, q2.y, q2.z);
                // state after bounce 0: affectFactor 1 -> 1/2, sumFactor 0 -> 1 (RK:139-140)
                affect = 0.5f; sum = 1.0f; bounce = 1u;
                active = true;
            }
            cur += take;
            idle = __ballot(!active);
        }
        if (__ballot(active) == 0ull) break;

        bool finished = false;
        if (active) {
            // ---- RK:114: trace the reflection ray (per-lane origin) ----
            float t; int idx;
            trace_filtered<true, SGN, CAP>(sG, Gw, N16, cl, ro, rd, t, idx);
            ++nrays;
            const float next = affect + sum;                         // RK:120
            if (idx < 0) {                                           // RK:122-126
                const v3 sky = scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, rd));
                color = divs(add(scale(sum, color), scale(affect, sky)), next);
                finished = true;
            } else {
                const float4 g = A.geo[idx];
                const float4 cl4 = A.col[idx];
                const v3 pos = add(ro, scale(t, rd));                    // RK:129
                const v3 normal = normalize(sub(pos, V(g.x, g.y, g.z))); // HK:320
                ro = pos;
                rd = normalize(reflect(rd, normal));                     // RK:130
                // ---- RK:146-153: shadow ray from the light (hoisted form) ----
                const v3 sdir = normalize(sub(ro, sc.lightPos));
                const float distance = length(sdir);
                float st; int sidx;
                trace_filtered<false, SGN, CAP>(sL, Lw, N16, cl, sc.lightPos, sdir, st, sidx);
                ++nrays;
                const float intensity = light_term(sc, ro, normal, sdir, distance, sidx >= 0, st);
                const v3 blended = scale(intensity, V(cl4.x, cl4.y, cl4.z));
                color = divs(add(scale(sum, color), scale(affect, blended)), next);   // RK:136
                affect = affect / 2.0f;                                  // RK:139
                sum = next;                                              // RK:140
                ++bounce;
                finished = bounce >= sc.bounces;                         // RK:113
            }
        }
        if (finished) {
            // RK:91-98: the primary direction is a function of the pixel alone; recompute it
            const uint32_t orow = opix / A.W, x = opix - orow * A.W;
            const uint32_t y = (A.tile_first + (orow >> 3) * A.tile_step) * 8u + (orow & 7u);
            const v3 dir0 = primary_dir(A, sc, x, y);
            reinterpret_cast<uint32_t*>(A.out)[opix] = compose_pixel_sky(scale(sc.minIntensity, cube_sample<FLAT ? 1 : 0>(A, dir0)), color, dist);
            active = false;
        }
    }
    count_rays(A.rays, nrays);
}

// ---- per-frame scene preparation ---------------------------------------------------------------
__global__ void prep_spheres(const RtPrepArgs A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.N16) return;
    if (i >= A.N) {   // padding records: can never pass the filter
        A.geo_f[i] = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
        A.lgt_f[i] = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
        A.cam_f[i] = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
        A.geo_w[i] = 0.0f; A.lgt_w[i] = 0.0f; A.cam_w[i] = 0.0f;
        return;
    }
    const float* r = A.records + 8u * i;
    const v3 c = V(r[0], r[1], r[2]);
    const float radius = r[7];
    const float r2 = radius * radius;                                   // HK:310 `radius * radius`
    const v3 lo = sub(V(A.p[16], A.p[17], A.p[18]), c);                 // origin - center, origin = light
    const v3 co = sub(V(A.p[0], A.p[1], A.p[2]), c);                    // origin = camera
    const float ll = dot(lo, lo), cc = dot(co, co);
    A.geo[i] = make_float4(c.x, c.y, c.z, r2);
    A.lgt[i] = make_float4(lo.x, lo.y, lo.z, ll - r2);                  // HK:310
    A.cam[i] = make_float4(co.x, co.y, co.z, cc - r2);
    A.col[i] = make_float4(r[4], r[5], r[6], 0.0f);
    A.geo_w[i] = r2; A.lgt_w[i] = ll - r2; A.cam_w[i] = cc - r2;
    const float S = RT_FILTER_SCALE, S2 = RT_FILTER_SCALE2;
    const float r2f = r2 * (1.0f + RT_FILTER_KAPPA);
    // expanded form: k = |c|^2 (1-eps) - r^2 (1+kappa), formed in double, rounded once
    const double cd2 = (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z;
    const double kf = cd2 * (1.0 - (double)RT_FILTER_EPS) - (double)r2 * (1.0 + (double)RT_FILTER_KAPPA);
    A.geo_f[i] = make_float4(c.x * S, c.y * S, c.z * S, (float)(kf * (double)S2));
    A.lgt_f[i] = make_float4(lo.x * S, lo.y * S, lo.z * S, (ll - r2f) * S2);
    A.cam_f[i] = make_float4(co.x * S, co.y * S, co.z * S, (cc - r2f) * S2);
}

// ---- launch ------------------------------------------------------------------------------------------------
template <typename K>
hipError_t set_lds(K k, size_t lds) {
    if (lds > 48u * 1024u)
        return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return hipSuccess;
}

constexpr size_t kLdsCap = 160u * 1024u;

// bytes of LDS a pixel-kernel workgroup needs
template <int WAVES, bool FILTER, bool FIRST, bool W_LDS, int CAP>
size_t lds_pixels(const RtFrameArgs& a) {
    const size_t n = FILTER ? a.N16 : a.N;
    const size_t arrays = FIRST ? 2 : ((FILTER || W_LDS) ? 3 : 2);
    return n * arrays * 16u + ((FILTER && W_LDS) ? n * arrays * 4u : 0u) + (FILTER ? (size_t)WAVES * CAP * 256u : 0u);
}

template <int WAVES, bool FILTER, bool FIRST, bool SGN, bool W_LDS, int CAP, bool GS = false>
hipError_t launch_pixels(const RtFrameArgs& a, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    const size_t lds = GS ? (FILTER ? (size_t)WAVES * CAP * 256u : 0u) : lds_pixels<WAVES, FILTER, FIRST, W_LDS, CAP>(a);
    if (lds > kLdsCap) return hipErrorInvalidValue;
    auto k = a.sky_flat ? trace_pixels<WAVES, FILTER, FIRST, SGN, W_LDS, CAP, GS, true> : trace_pixels<WAVES, FILTER, FIRST, SGN, W_LDS, CAP, GS, false>;
    hipError_t e = set_lds(k, lds);
    if (e != hipSuccess) return e;
    dim3 grid((a.W + 8u * WAVES - 1u) / (8u * WAVES), a.n_local_tiles, 1);
    hipLaunchKernelGGL(k, grid, dim3(64 * WAVES), lds, s, a);
    return hipGetLastError();
}

template <int WAVES, bool W_LDS, int CAP>
size_t lds_paths(const RtFrameArgs& a) {
    return (size_t)a.N16 * 32u + (W_LDS ? (size_t)a.N16 * 8u : 0u) + (size_t)WAVES * CAP * 256u;
}

template <int WAVES, bool SGN, bool W_LDS, int CAP>
hipError_t launch_paths(const RtFrameArgs& a, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    const size_t lds = lds_paths<WAVES, W_LDS, CAP>(a);
    if (lds > kLdsCap) return hipErrorInvalidValue;
    auto k = a.sky_flat ? trace_paths<WAVES, SGN, W_LDS, CAP, true> : trace_paths<WAVES, SGN, W_LDS, CAP, false>;
    hipError_t e = set_lds(k, lds);
    if (e != hipSuccess) return e;
    // persistent grid: enough workgroups to fill 256 CUs at the residency LDS allows;
    // surplus workgroups find the queue empty and exit
    const uint32_t per_cu = (uint32_t)min((size_t)(32 / WAVES), kLdsCap / lds);
    const uint32_t pixels = a.n_local_tiles * 8u * a.W;
    uint32_t blocks = 256u * (per_cu ? per_cu : 1u);
    const uint32_t need = (pixels + 64u * WAVES - 1u) / (64u * WAVES);
    if (blocks > need) blocks = need;
    g_rt_kernel_id = RT_KID_BRUTE_PIPELINE;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * WAVES), lds, s, a);
    return hipGetLastError();
}

template <int WAVES, bool W_LDS, int CAP>
size_t lds_first(const RtFrameArgs& a) {
    return (size_t)a.N16 * 32u + (W_LDS ? (size_t)a.N16 * 8u : 0u) + (size_t)WAVES * 2u * CAP * 256u;
}

template <int WAVES, bool SGN, bool W_LDS, int CAP>
hipError_t launch_first(const RtFrameArgs& a, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    const size_t lds = lds_first<WAVES, W_LDS, CAP>(a);
    if (lds > kLdsCap) return hipErrorInvalidValue;
    auto k = a.sky_flat ? first_bounce<WAVES, SGN, W_LDS, CAP, true> : first_bounce<WAVES, SGN, W_LDS, CAP, false>;
    hipError_t e = set_lds(k, lds);
    if (e != hipSuccess) return e;
    const uint32_t per_cu = (uint32_t)min((size_t)(32 / WAVES), kLdsCap / lds);
    const uint32_t pairs = ((a.W + 15u) / 16u) * a.n_local_tiles;
    uint32_t blocks = 256u * (per_cu ? per_cu : 1u);
    const uint32_t need = (pairs + WAVES - 1u) / WAVES;
    if (blocks > need) blocks = need;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * WAVES), lds, s, a);
    return hipGetLastError();
}

// RT_MODE_STRICT: one literal kernel, exact records in LDS
hipError_t launch_strict(const RtFrameArgs& a, hipStream_t s) {
    const size_t bytes = (size_t)a.N * 48u;
    if (bytes <= 56u * 1024u) return launch_pixels<8, false, false, false, true, 1>(a, s);
    if (bytes <= kLdsCap)     return launch_pixels<16, false, false, false, true, 1>(a, s);
    if ((size_t)a.N * 32u <= kLdsCap) return launch_pixels<16, false, false, false, false, 1>(a, s);
    return launch_pixels<4, false, false, false, false, 1, true>(a, s);      // any N: records from global memory
}

template <bool SGN>
hipError_t launch_fast(const RtFrameArgs& a, int variant, hipStream_t s) {
    const size_t rec = (size_t)a.N16 * 16u;
    if (2 * rec + 8u * 8u * 256u > kLdsCap)                                  // > 4608 spheres: no LDS staging
        return launch_pixels<4, true, false, SGN, false, 16, true>(a, s);
    const bool small = 3 * rec + 3 * (rec / 4) + 8u * 16u * 256u <= 80u * 1024u;      // N <= ~800: 2 workgroups of 8 waves per CU
    const bool pipeline = a.queue && a.qctrl && a.N >= 320u;   // measured crossover of the two brute-force forms
    switch (variant) {
        case 0:   // default: two-kernel pipeline for scenes where the sphere loop dominates,
                  // single kernel for small scenes (the queue round trip costs more than it saves)
            if (pipeline) {
                hipError_t e;
                if (lds_paths<8, true, 16>(a) <= kLdsCap) {                      // N <= ~3600
                    e = launch_first<8, SGN, true, 8>(a, s);
                    if (e != hipSuccess) return e;
                    return launch_paths<8, SGN, true, 16>(a, s);
                }
                // larger scenes: exact 4th components from global memory; 16-wave workgroups so
                // that the one workgroup a CU can hold still gives 4 waves per SIMD
                if (lds_paths<16, false, 8>(a) <= kLdsCap) {                     // N <= 4096
                    e = launch_first<16, SGN, false, 4>(a, s);
                    if (e != hipSuccess) return e;
                    return launch_paths<16, SGN, false, 8>(a, s);
                }
                e = launch_first<8, SGN, false, 4>(a, s);                        // N <= 4608
                if (e != hipSuccess) return e;
                return launch_paths<8, SGN, false, 8>(a, s);
            }
            [[fallthrough]];
        case 1:   // single kernel
            if (small) return launch_pixels<8, true, false, SGN, true, 16>(a, s);
            if (lds_pixels<8, true, false, true, 16>(a) <= kLdsCap) return launch_pixels<8, true, false, SGN, true, 16>(a, s);
            return hipErrorInvalidValue;   // > ~2700 spheres need the pipeline
        case 2:   // pipeline, 4-wave workgroups in the path stage
            if (a.queue && a.qctrl && lds_paths<4, true, 16>(a) <= kLdsCap) {
                hipError_t e = launch_pixels<8, true, true, SGN, true, 16>(a, s);
                if (e != hipSuccess) return e;
                return launch_paths<4, SGN, true, 16>(a, s);
            }
            return hipErrorInvalidValue;
        case 3:   // pipeline forced (also for small scenes)
            if (a.queue && a.qctrl && lds_paths<8, true, 16>(a) <= kLdsCap) {
                hipError_t e = launch_pixels<8, true, true, SGN, true, 16>(a, s);
                if (e != hipSuccess) return e;
                return launch_paths<8, SGN, true, 16>(a, s);
            }
            return hipErrorInvalidValue;
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace rtk

hipError_t rt_launch_trace(const RtFrameArgs& a, const RtLaunchCfg& cfg, hipStream_t s) {
    if (cfg.mode == 1) { g_rt_kernel_id = RT_KID_LITERAL; return rtk::launch_strict(a, s); }
    g_rt_kernel_id = RT_KID_BRUTE_SINGLE;          // launch_paths overrides it when the two-kernel pipeline runs
    return a.signed_filter ? rtk::launch_fast<true>(a, cfg.variant, s) : rtk::launch_fast<false>(a, cfg.variant, s);
}

hipError_t rt_launch_prep(const RtPrepArgs& a, hipStream_t s) {
    if (a.N16 == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::prep_spheres, dim3((a.N16 + 255u) / 256u), dim3(256), 0, s, a);
    return hipGetLastError();
}
