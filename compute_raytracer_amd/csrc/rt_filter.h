// rt_filter.h -- device helpers shared by the sphere kernels (rt_kernels.hip, rt_bvh.hip): the
// literal ray-sphere test of the reference and the conservative FMA-only discriminant filter.
// Include only from translation units compiled with -ffp-contract=off -fno-slp-vectorize.
#pragma once
#include "rt_device.h"

namespace rtk {

// ---- literal ray-sphere test (HK:308-318) ------------------------------------------------------------
// SHORTCUT: b >= 0 makes (-b - sqrt(disc)) <= 0, so t <= 0 fails `t > tMin`; skipping the
// square root and the division then is exact.  The strict kernel keeps the literal form.
template <bool SHORTCUT>
__device__ __forceinline__ void exact_full(v3 center, float r2, int s, v3 o, v3 d, float fa, float ta,
                                           float& nearest, int& idx) {
    const v3 oc = sub(o, center);
    const float b = 2.0f * dot(d, oc);              // HK:309
    const float c = dot(oc, oc) - r2;               // HK:310
    const float disc = b * b - fa * c;              // HK:311
    if (disc > 0.0f && (!SHORTCUT || b < 0.0f)) {   // HK:316
        const float t = (-b - sqrtf(disc)) / ta;    // HK:317
        if (t > 0.001f && t < nearest) {            // HK:318 with tMin/tMax of RK:315
            nearest = t;
            idx = s;
        }
    }
}
// Hoisted form: oc = o - center and c = |oc|^2 - r^2 precomputed for the common origin o.
template <bool SHORTCUT>
__device__ __forceinline__ void exact_hoisted(v3 oc, float c, int s, v3 d, float fa, float ta,
                                              float& nearest, int& idx) {
    const float b = 2.0f * dot(d, oc);
    const float disc = b * b - fa * c;
    if (disc > 0.0f && (!SHORTCUT || b < 0.0f)) {
        const float t = (-b - sqrtf(disc)) / ta;
        if (t > 0.001f && t < nearest) {
            nearest = t;
            idx = s;
        }
    }
}

// RT_MODE_STRICT: the literal loop over the exact records.
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
        const float4 g = E[s];
        if (FULL) exact_full<false>(V(g.x, g.y, g.z), g.w, (int)s, o, d, fa, ta, nearest, idx);
        else      exact_hoisted<false>(V(g.x, g.y, g.z), g.w, (int)s, d, fa, ta, nearest, idx);
    }
}

// ---- RT_MODE_FAST: filtered nearest-hit search ---------------------------------------------------------
// v_fma_f32 forms.  RT_FMA_BUILTIN=1 (default): __builtin_fmaf, which hipcc keeps as v_fma_f32 /
// v_fmac_f32 with the neg/abs/clamp modifiers folded in and schedules freely (5 % faster than
// the inline-asm form, RT_FMA_BUILTIN=0, around which it pads s_nop).  Where LLVM would
// canonicalise an FMA into a slower v_sub/v_add (x*(-1)+y), the multiplier is an opaque register.
#ifndef RT_FMA_BUILTIN
#define RT_FMA_BUILTIN 1
#endif
#ifndef RT_CLAMP_BUILTIN
#define RT_CLAMP_BUILTIN 0   /* measured: the inline-asm clamp forms are 4 % faster (7.65 vs 7.97 ms, same box) */
#endif
__device__ __forceinline__ float fma_vvv(float a, float b, float c) {
#if RT_FMA_BUILTIN
    return __builtin_fmaf(a, b, c);
#else
    float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
#endif
}
__device__ __forceinline__ float mul_fma(float a, float b) {          // a*b
#if RT_FMA_BUILTIN
    return __builtin_fmaf(a, b, 0.0f);
#else
    float d; asm("v_fma_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b)); return d;
#endif
}
__device__ __forceinline__ float sub_fma(float o, float c) {          // o - c
    float d; asm("v_fma_f32 %0, %1, -1.0, %2" : "=v"(d) : "v"(c), "v"(o)); return d;
}
__device__ __forceinline__ float sq_acc(float x, float c) {           // x*x + c
    float d; asm("v_fma_f32 %0, %1, %1, %2" : "=v"(d) : "v"(x), "v"(c)); return d;
}
__device__ __forceinline__ float sq_sub(float x, float r) {           // x*x - r
    float d; asm("v_fma_f32 %0, %1, %1, -%2" : "=v"(d) : "v"(x), "v"(r)); return d;
}
// ---- candidate mask of a 16-sphere batch ----------------------------------------------------------------
// RT_MASK_SIGNBITS=1 (default): the mask is built from the SIGN BIT of each filter discriminant,
// shifted in with one v_alignbit_b32 per sphere: code = (code << 1) | sign(x).  A sphere is a
// candidate unless x is negative (x == +0 counts as a candidate: conservative).  Exact for every
// input -- there is no value of x that can disturb a neighbouring bit.
// RT_MASK_SIGNBITS=0: the round-1 form, kept for A/B timing only: the `clamp` modifier of the FMA
// that forms x yields 1.0 / 0.0 and code = 2*code + dd is one more FMA.  That form has a hole:
// 0 < x < 1 (the 2^80-scaled discriminant of a ray that grazes the inflated sphere to within
// 2^-40, constructible: tests/test_filter_fraction_gpu.py) leaves a FRACTION in the sum, whose
// carries can clear the bit of another sphere of the batch.
#ifndef RT_MASK_SIGNBITS
#define RT_MASK_SIGNBITS 1
#endif
#if RT_MASK_SIGNBITS
typedef uint32_t mask_acc;
__device__ __forceinline__ mask_acc mask_zero() { return 0u; }
__device__ __forceinline__ mask_acc shift_in(mask_acc code, float x) {   // (code << 1) | (x < 0 or -0)
    return __builtin_amdgcn_alignbit(code, __float_as_uint(x), 31u);
}
__device__ __forceinline__ uint32_t mask_bits(mask_acc code) { return code ^ 0xFFFFu; }   // after 16 shift_in: bit (15-k) = sphere k may be hit
#else
typedef float mask_acc;
__device__ __forceinline__ mask_acc mask_zero() { return 0.0f; }
__device__ __forceinline__ uint32_t mask_bits(mask_acc code) { return (uint32_t)code; }
#endif
template <bool SGN>
__device__ __forceinline__ float disc_ind(float b, float c) {
#if RT_MASK_SIGNBITS
    return SGN ? __builtin_fmaf(-b, __builtin_fabsf(b), -c) : __builtin_fmaf(b, b, -c);   // the raw discriminant
#endif
#if RT_CLAMP_BUILTIN
    return __builtin_amdgcn_fmed3f(SGN ? __builtin_fmaf(-b, __builtin_fabsf(b), -c) : __builtin_fmaf(b, b, -c), 0.0f, 1.0f);
#endif
    float d;
    if (SGN) asm("v_fma_f32 %0, -%1, |%1|, -%2 clamp" : "=v"(d) : "v"(b), "v"(c));   // -b|b| - c: b*b - c if b < 0
    else     asm("v_fma_f32 %0, %1, %1, -%2 clamp" : "=v"(d) : "v"(b), "v"(c));      // b*b - c
    return d;
}
#if !RT_MASK_SIGNBITS
__device__ __forceinline__ float shift_in(float code, float bit) {    // 2*code + bit
#if RT_FMA_BUILTIN
    return __builtin_fmaf(code, 2.0f, bit);
#endif
    float d; asm("v_fma_f32 %0, %1, 2.0, %2" : "=v"(d) : "v"(code), "v"(bit)); return d;
}
#endif

// Conservative test "can sphere s have discriminant > 0 (and lie in front of the origin)".
// With h = d/|d| the reference's condition b^2 - 4a*c > 0 (HK:311,316) is (h.oc)^2 - c > 0.  The
// filter evaluates that with fused arithmetic on records scaled by 2^40, h scaled by (1+kappa)
// and r^2 by (1+kappa): the value it tests exceeds the real one by >= kappa*(|oc|^2 + r^2)/2
// (times 2^80) for every sphere whose real discriminant is not clearly negative, while the
// rounding of the filter (<= 14u) and of the literal evaluation (<= 8u, u = 2^-24, both
// relative to |oc|^2 + r^2) together stay below 22u = kappa/50.  Spheres with c <= 0 (origin
// inside or on the sphere) always pass.  SGN additionally rejects spheres behind the origin
// (b > 0 and c > 0): a valid hit needs t > 0.001, i.e. h.oc < -0.001, and the host enables SGN
// only when the scene is small enough for the rounding of h.oc (<= 7.3e-7 |oc|) to stay below
// half of that.
//
// Per-lane origin (FULL): the expanded form saves the three `o - c` subtractions.  With the
// per-ray scalars p = h.o, q = |o|^2 (1-eps), m = -2 o and the per-sphere k = |c|^2 (1-eps) - r^2(1+kappa)
//     b = p - h.c                      3 FMAs
//     cp = k + m.c                     3 FMAs      (c = cp + q)
//     e = b*b - q,  dd = e - cp        2 FMAs      (the last one clamps)
// Expanding |o-c|^2 costs cancellation error <= 48u (|o|^2 + |c|^2); eps = 2^-17 = 128u shifts the
// tested value up by eps (|o|^2 + |c|^2), which covers it; the kappa terms cover the rest as above.
struct RayF {     // per-ray constants of the filter (2^40-scaled where a length)
    float negone; // -1.0 in a register the compiler cannot see through (keeps e - cp an FMA)
    v3 h;         // d/|d| * (1+kappa)
    v3 m;         // -2 * o * 2^40        (FULL)
    float p, q;   // h.o * 2^40,  |o|^2 (1-eps) * 2^80   (FULL)
};
__device__ __forceinline__ float fnma_vvv(float a, float b, float c) {   // c - a*b
#if RT_FMA_BUILTIN
    return __builtin_fmaf(-a, b, c);
#endif
    float d; asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d;
}
template <bool SGN>
__device__ __forceinline__ float sq_signed_minus(float b, float q) {     // b*b - q  (SGN: -b|b| - q)
#if RT_FMA_BUILTIN
    return SGN ? __builtin_fmaf(-b, __builtin_fabsf(b), -q) : __builtin_fmaf(b, b, -q);
#endif
    float d;
    if (SGN) asm("v_fma_f32 %0, -%1, |%1|, -%2" : "=v"(d) : "v"(b), "v"(q));
    else     asm("v_fma_f32 %0, %1, %1, -%2" : "=v"(d) : "v"(b), "v"(q));
    return d;
}
__device__ __forceinline__ float opaque_negone() {
    float x; asm volatile("v_mov_b32 %0, -1.0" : "=v"(x)); return x;
}
__device__ __forceinline__ float sub_clamp(float e, float c, float negone) {   // clamp(e - c); sign-bit masks: e - c
#if RT_MASK_SIGNBITS
    return __builtin_fmaf(c, negone, e);    // negone is opaque: stays a v_fma_f32
#endif
#if RT_CLAMP_BUILTIN
    return __builtin_amdgcn_fmed3f(__builtin_fmaf(c, negone, e), 0.0f, 1.0f);
#endif
    float d; asm("v_fma_f32 %0, %1, -1.0, %2 clamp" : "=v"(d) : "v"(c), "v"(e)); return d;
}
template <bool FULL, bool SGN>
__device__ __forceinline__ float filter_one(const float4 g, const RayF& r) {
    if (FULL) {
        const float b = fnma_vvv(r.h.z, g.z, fnma_vvv(r.h.y, g.y, fnma_vvv(r.h.x, g.x, r.p)));
        const float cp = fma_vvv(r.m.z, g.z, fma_vvv(r.m.y, g.y, fma_vvv(r.m.x, g.x, g.w)));
        return sub_clamp(sq_signed_minus<SGN>(b, r.q), cp, r.negone);
    } else {
        const float b = fma_vvv(r.h.z, g.z, fma_vvv(r.h.y, g.y, mul_fma(r.h.x, g.x)));
        return disc_ind<SGN>(b, g.w);
    }
}

}  // namespace rtk
