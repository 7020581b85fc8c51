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
//   * one pixel per lane, one 8x8 pixel tile per wave64 (= one WGSL workgroup, RK:73), WAVES
//     tiles side by side per workgroup sharing one LDS copy of the scene;
//   * sphere records are SoA float4: one ds_read_b128 with a wave-uniform address feeds one
//     ray-sphere test of the whole wave (an SGPR operand would halve the v_fma_f32 rate);
//   * only v_fma_f32/v_fmac_f32 issue at 2 cycles per wave on gfx950, v_add/v_mul/v_max/v_cmp
//     take about 4: the per-sphere filter is written as v_fma_f32 only (10 for a ray with a
//     per-lane origin, 4 for rays from the camera or the light), plus half a v_max3_f32;
//   * rays that start at one point for all lanes (primary rays at the camera, shadow rays at
//     the light, RK:151) use records with `origin - center` and `c` precomputed per frame;
//   * exactness: the filter decides only "can this sphere have discriminant > 0"; it is
//     conservative (margin 2^-16, see RT_FILTER_KAPPA), and every sphere it lets through is
//     re-evaluated with the reference's literal arithmetic, in index order, so nearest-hit
//     selection, t, normals and colours are the oracle's bits.
#include "rt_device.h"

namespace rtk {

// ---- literal nearest-hit loops (RT_MODE_STRICT; also the definition the filter must match) ----
// Full form (HK:308-318): per-lane ray origin.  G[s] = {cx, cy, cz, r*r}
__device__ __forceinline__ void exact_full(const float4 g, int s, v3 o, v3 d, float fa, float ta,
                                           float& nearest, int& idx) {
    const v3 oc = V(o.x - g.x, o.y - g.y, o.z - g.z);
    const float b = 2.0f * dot(d, oc);              // HK:309
    const float c = dot(oc, oc) - g.w;              // HK:310
    const float disc = b * b - fa * c;              // HK:311
    // b >= 0 makes (-b - sqrt(disc)) <= 0, so t <= 0 fails `t > tMin`: skipping it is exact
    if (disc > 0.0f && b < 0.0f) {                  // HK:316
        const float t = (-b - sqrtf(disc)) / ta;    // HK:317
        if (t > 0.001f && t < nearest) {            // HK:318 with tMin/tMax of RK:315
            nearest = t;
            idx = s;
        }
    }
}
// Hoisted form: P[s] = {o-c, |o-c|^2 - r^2} for the common origin o.
__device__ __forceinline__ void exact_hoisted(const float4 g, int s, v3 d, float fa, float ta,
                                              float& nearest, int& idx) {
    const float b = 2.0f * dot(d, V(g.x, g.y, g.z));
    const float disc = b * b - fa * g.w;
    if (disc > 0.0f && b < 0.0f) {
        const float t = (-b - sqrtf(disc)) / ta;
        if (t > 0.001f && t < nearest) {
            nearest = t;
            idx = s;
        }
    }
}

template <bool FULL>
__device__ __forceinline__ void trace_literal(const float4* __restrict__ E, uint32_t N, v3 o, v3 d,
                                              float& nearest, int& idx) {
    const float a = dot(d, d);           // HK:308
    const float fa = 4.0f * a;           // the (4*a) of HK:311
    const float ta = 2.0f * a;           // HK:317
    nearest = 9999.0f;                   // RK:172
    idx = -1;
#pragma unroll 4
    for (uint32_t s = 0; s < N; ++s) {
        if (FULL) exact_full(E[s], (int)s, o, d, fa, ta, nearest, idx);
        else      exact_hoisted(E[s], (int)s, d, fa, ta, nearest, idx);
    }
}

// ---- filtered nearest-hit loop (RT_MODE_FAST) ---------------------------------------------------
// v_fma_f32 forms (inline asm so that instruction selection is ours: the compiler would turn
// fma(x,1,y) into v_add_f32 and fma(x,y,0) into v_mul_f32, both half rate on gfx950)
__device__ __forceinline__ float fma_vvv(float a, float b, float c) {
    float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ float mul_fma(float a, float b) {          // a*b
    float d; asm("v_fma_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b)); return d;
}
__device__ __forceinline__ float sub_fma(float o, float c) {          // o - c
    float d; asm("v_fma_f32 %0, %1, -1.0, %2" : "=v"(d) : "v"(c), "v"(o)); return d;
}
__device__ __forceinline__ float sq_minus(float b, float c) {         // b*b - c
    float d; asm("v_fma_f32 %0, %1, %1, -%2" : "=v"(d) : "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ float sq_acc(float x, float c) {           // x*x + c
    float d; asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d) : "v"(x), "v"(c)); return d;
}
__device__ __forceinline__ float max3_(float a, float b, float c) {   // no NaN canonicalisation moves
    float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ float sq_sub(float x, float r) {           // x*x - r
    float d; asm("v_fma_f32 %0, %1, %1, -%2" : "=v"(d) : "v"(x), "v"(r)); return d;
}

// Conservative test "can sphere s have discriminant > 0 for this ray".
// With h = d/|d| the reference's condition b^2 - 4a*c > 0 (HK:311,316) is (h.oc)^2 - c > 0.
// The filter evaluates that with fused arithmetic, h scaled by (1+kappa) and r^2 by (1+kappa):
// the value it tests exceeds the real one by >= kappa*(|oc|^2 + r^2)/2 for every sphere whose
// real discriminant is not clearly negative, while the rounding of the filter (<= 14u) and of
// the literal evaluation (<= 8u, u = 2^-24, both relative to |oc|^2 + r^2) together stay below
// 22u = kappa/50.  Spheres with c <= 0 (origin inside or on the sphere) always pass.
template <bool FULL>
__device__ __forceinline__ float filter_one(const float4 g, v3 o, v3 h) {
    if (FULL) {
        const float ocx = sub_fma(o.x, g.x), ocy = sub_fma(o.y, g.y), ocz = sub_fma(o.z, g.z);
        const float b = fma_vvv(h.z, ocz, fma_vvv(h.y, ocy, mul_fma(h.x, ocx)));
        const float c = sq_acc(ocz, sq_acc(ocy, sq_sub(ocx, g.w)));
        return sq_minus(b, c);
    } else {
        const float b = fma_vvv(h.z, g.z, fma_vvv(h.y, g.y, mul_fma(h.x, g.x)));
        return sq_minus(b, g.w);
    }
}

// F: filter records (LDS), padded to N8 = multiple of 8; X: exact records (global, [N]).
template <bool FULL>
__device__ __forceinline__ void trace_filtered(const float4* __restrict__ F, const float4* __restrict__ X,
                                               uint32_t N8, v3 o, v3 d, float& nearest, int& idx) {
    const float a = dot(d, d);
    const float fa = 4.0f * a;
    const float ta = 2.0f * a;
    const float inv = __builtin_amdgcn_rsqf(a) * (1.0f + RT_FILTER_KAPPA);
    const v3 h = V(d.x * inv, d.y * inv, d.z * inv);
    nearest = 9999.0f;
    idx = -1;
    for (uint32_t s = 0; s < N8; s += 8) {
        float4 g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = F[s + k];
        float dd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dd[k] = filter_one<FULL>(g[k], o, h);
        const float m = max3_(max3_(dd[0], dd[1], dd[2]), max3_(dd[3], dd[4], dd[5]), max3_(dd[6], dd[7], dd[7]));
        if (m > 0.0f) {
            // rare: per-lane list of the spheres that passed, visited in index order
            uint32_t mask = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) mask |= dd[k] > 0.0f ? (1u << k) : 0u;
            while (mask) {
                const int si = (int)s + (__ffs((int)mask) - 1);
                mask &= mask - 1u;
                const float4 e = X[si];
                if (FULL) exact_full(e, si, o, d, fa, ta, nearest, idx);
                else      exact_hoisted(e, si, d, fa, ta, nearest, idx);
            }
        }
    }
}

// ---- shading of one bounce after the nearest hit is known (RK:128-140) -----------------------
struct PathState {
    v3 ro, rd, color;
    float affect, sum, dist;
};

// ---- kernel: pixel per lane ----------------------------------------------------------------------
// FILTER=false: RT_MODE_STRICT, literal loops over the exact records staged in LDS.
// FILTER=true : RT_MODE_FAST, filtered loops over the filter records staged in LDS; the exact
//               records of the few spheres that pass come from global memory (L2-resident).
// LDS: 3 arrays (geo, light-hoisted, camera-hoisted) x N8 x 16 B when CAM_LDS, else 2.
template <int WAVES, bool FILTER, bool CAM_LDS>
__global__ __launch_bounds__(64 * WAVES) void trace_pixels(const RtFrameArgs A) {
    extern __shared__ float4 lds[];
    const uint32_t N = A.N, N8 = A.N8;
    float4* sG = lds;
    float4* sL = lds + N8;
    float4* sC = lds + 2 * N8;
    {
        const float4* srcG = FILTER ? A.geo_f : A.geo;
        const float4* srcL = FILTER ? A.lgt_f : A.lgt;
        const float4* srcC = FILTER ? A.cam_f : A.cam;
        const uint32_t n = FILTER ? N8 : N;
        for (uint32_t i = threadIdx.x; i < n; i += 64 * WAVES) {
            sG[i] = srcG[i];
            sL[i] = srcL[i];
            if (CAM_LDS) sC[i] = srcC[i];
        }
        __syncthreads();
    }
    const float4* camRec = CAM_LDS ? sC : (FILTER ? A.cam_f : A.cam);

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
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
    for (uint32_t bounce = 0; bounce < sc.bounces; ++bounce) {
        float t; int idx;
        if (FILTER) {
            if (bounce == 0) trace_filtered<false>(camRec, A.cam, N8, ro, rd, t, idx);
            else             trace_filtered<true>(sG, A.geo, N8, ro, rd, t, idx);
        } else {
            if (bounce == 0) trace_literal<false>(camRec, N, ro, rd, t, idx);
            else             trace_literal<true>(sG, N, ro, rd, t, idx);
        }
        ++nrays;
        const bool hit = idx >= 0;
        if (bounce == 0) dist = hit ? t : 0.0f;                  // RK:116-118 (zero-initialised state)
        const float next = affect + sum;                         // RK:120
        if (!hit) {                                              // RK:122-126
            const v3 sky = scale(sc.minIntensity, cube_sample(A, rd));
            color = divs(add(scale(sum, color), scale(affect, sky)), next);
            break;
        }
        const float4 g = A.geo[idx];
        const float4 cl = A.col[idx];
        const v3 pos = add(ro, scale(t, rd));                    // HK:319 == RK:129
        const v3 normal = normalize(sub(pos, V(g.x, g.y, g.z))); // HK:320
        ro = pos;
        rd = normalize(reflect(rd, normal));                     // RK:130

        // RK:146-153 lightIntensity(ro, normal): shadow ray from the light
        const v3 sdir = normalize(sub(ro, sc.lightPos));         // RK:147
        const float distance = length(sdir);                     // RK:148
        float st; int sidx;
        if (FILTER) trace_filtered<false>(sL, A.lgt, N8, sc.lightPos, sdir, st, sidx);
        else        trace_literal<false>(sL, N, sc.lightPos, sdir, st, sidx);
        ++nrays;
        const float intensity = light_term(sc, ro, normal, sdir, distance, sidx >= 0, st);
        const v3 blended = scale(intensity, V(cl.x, cl.y, cl.z)); // RK:133-135, diffuse.w == 1
        color = divs(add(scale(sum, color), scale(affect, blended)), next); // RK:136
        affect = affect / 2.0f;                                  // RK:139
        sum = next;                                              // RK:140
    }

    const uint32_t packed = compose_pixel(A, sc, dir0, color, dist);   // RK:91-98
    const size_t orow = (size_t)blockIdx.y * 8u + row;
    reinterpret_cast<uint32_t*>(A.out)[orow * A.W + x] = packed;
    count_rays(A.rays, nrays);
}

// ---- per-frame scene preparation ---------------------------------------------------------------
__global__ void prep_spheres(const RtPrepArgs A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.N8) return;
    if (i >= A.N) {   // padding records: can never pass the filter
        A.geo_f[i] = make_float4(0.0f, 0.0f, 0.0f, -INFINITY);
        A.lgt_f[i] = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
        A.cam_f[i] = make_float4(0.0f, 0.0f, 0.0f, INFINITY);
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
    const float r2f = r2 * (1.0f + RT_FILTER_KAPPA);
    A.geo_f[i] = make_float4(c.x, c.y, c.z, r2f);
    A.lgt_f[i] = make_float4(lo.x, lo.y, lo.z, ll - r2f);
    A.cam_f[i] = make_float4(co.x, co.y, co.z, cc - r2f);
}

template <int WAVES, bool FILTER, bool CAM_LDS>
hipError_t launch_pixels(const RtFrameArgs& a, hipStream_t s) {
    if (a.n_local_tiles == 0 || a.W == 0) return hipSuccess;
    const size_t lds = (size_t)a.N8 * (CAM_LDS ? 3u : 2u) * sizeof(float4);
    auto k = trace_pixels<WAVES, FILTER, CAM_LDS>;
    if (lds > 48u * 1024u) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    dim3 grid((a.W + 8u * WAVES - 1u) / (8u * WAVES), a.n_local_tiles, 1);
    hipLaunchKernelGGL(k, grid, dim3(64 * WAVES), lds, s, a);
    return hipGetLastError();
}

template <bool FILTER>
hipError_t launch_mode(const RtFrameArgs& a, int variant, hipStream_t s) {
    const size_t rec = (size_t)a.N8 * sizeof(float4);
    const size_t cap = 160u * 1024u;
    if (2 * rec > cap) return hipErrorInvalidValue;   // > 5120 spheres: not built yet (chunked staging)
    const bool cam = 3 * rec <= cap;
    switch (variant) {
        case 0:
        case 1:   // 8 waves (64x8 px) per workgroup; 16 when the scene takes most of a CU's LDS
            if (3 * rec <= 80u * 1024u) return launch_pixels<8, FILTER, true>(a, s);
            return cam ? launch_pixels<16, FILTER, true>(a, s) : launch_pixels<16, FILTER, false>(a, s);
        case 2:
            return cam ? launch_pixels<4, FILTER, true>(a, s) : launch_pixels<16, FILTER, false>(a, s);
        case 3:
            return cam ? launch_pixels<16, FILTER, true>(a, s) : launch_pixels<16, FILTER, false>(a, s);
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace rtk

hipError_t rt_launch_trace(const RtFrameArgs& a, const RtLaunchCfg& cfg, hipStream_t s) {
    return cfg.mode == 1 ? rtk::launch_mode<false>(a, cfg.variant, s) : rtk::launch_mode<true>(a, cfg.variant, s);
}

hipError_t rt_launch_prep(const RtPrepArgs& a, hipStream_t s) {
    if (a.N8 == 0) return hipSuccess;
    hipLaunchKernelGGL(rtk::prep_spheres, dim3((a.N8 + 255u) / 256u), dim3(256), 0, s, a);
    return hipGetLastError();
}
