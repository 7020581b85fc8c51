// rt_device.h -- device-side helpers shared by the gfx950 ray-trace kernels.
//
// Everything here evaluates the reference shader's arithmetic in correctly rounded fp32 with
// no multiply-add fusion (the translation unit is compiled with -ffp-contract=off), in the
// operation order oracle/rt_oracle.c fixes, so that results are bit-identical to the oracle.
// Citations are relative to the reference repository:
//   RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl
//   HK = src/rendering-raycast/shaders/heatmap-kernel.wgsl
#pragma once
#include "rt_types.h"

namespace rtk {

struct v3 { float x, y, z; };
__device__ __forceinline__ v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 scale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float length(v3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ v3 normalize(v3 a) { return divs(a, length(a)); }
// length() of a vector that is itself the output of normalize() (RK:148 takes the length of the normalised shadow
// direction): the squared length is 1 + k ulps with a handful of k, and the correctly rounded root of such a number is
// known without extracting it -- sqrt(1 + k 2^-23) = 1 + k 2^-24 - ..., a float or just under the midpoint of two, so its
// bit pattern is 0x3F800000 + (k >> 1) (arithmetic shift; the same formula below 1, where the ulp halves).  Checked
// against sqrtf for every |k| <= 4096 (tests/test_shortcuts_cpu.py); anything further from 1 (a NaN direction, say)
// takes the square root proper.
__device__ __forceinline__ float length_of_unit(v3 a) {
    const float x = dot(a, a);
    const int k = __float_as_int(x) - 0x3F800000;
    const float quick = __int_as_float(0x3F800000 + (k >> 1));
    const bool near_one = (uint32_t)(k + 4096) <= 8192u;
    if (__builtin_expect(__ballot(!near_one) == 0ull, 1)) return quick;
    return near_one ? quick : sqrtf(x);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// WGSL reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2
__device__ __forceinline__ v3 reflect(v3 e1, v3 e2) { return sub(e1, scale(2.0f * dot(e2, e1), e2)); }

// ---- cube map sample: the arithmetic oracle/rt_oracle.c:cube_sample fixes -------------------
// lut: optional 256-entry table of (float)i / 255.0f (same division, done once per workgroup)
__device__ __forceinline__ v3 texel(const uint8_t* __restrict__ f, int w, int h, int x, int y, const float* lut = nullptr) {
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    const uchar4 p = *reinterpret_cast<const uchar4*>(f + 4u * ((size_t)y * (size_t)w + (size_t)x));
    if (lut) return V(lut[p.x], lut[p.y], lut[p.z]);
    return V((float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f);
}
__device__ __forceinline__ v3 lerp3(v3 a, v3 b, float f) { return add(a, scale(f, sub(b, a))); }

// ---- seamless cube filtering (oracle/rt_oracle.c: cube_fold, cube_tap) -------------------------------
// A WebGPU cube texture is six equal square layers and filters seamlessly across its edges (Vulkan
// "Cube Map Edge Handling"): a bilinear tap one step outside the selected face is the texel of the
// adjacent face that touches the crossed edge at the same position along it.  Integer geometry: the
// texel centre (S, T) = (2i+1-n, 2j+1-n) of `face` as a 3-D point with the face planes at +-n, folded
// over the edge (the coordinate that left the cube becomes the major axis at +-n, the old major axis
// drops to +-(n-1)) and read back through the face table of the new face.
__device__ inline void cube_fold(int face, int i, int j, int n, int& nf, int& ni, int& nj) {
    const int S = 2 * i + 1 - n, T = 2 * j + 1 - n, m = n - 1;
    int x, y, z;
    if (face == 0)      { x = n;  y = -T; z = -S; }    // +X: sc = -z, tc = -y
    else if (face == 1) { x = -n; y = -T; z = S;  }    // -X: sc = +z, tc = -y
    else if (face == 2) { x = S;  y = n;  z = T;  }    // +Y: sc = +x, tc = +z
    else if (face == 3) { x = S;  y = -n; z = -T; }    // -Y: sc = +x, tc = -z
    else if (face == 4) { x = S;  y = -T; z = n;  }    // +Z: sc = +x, tc = -y
    else                { x = -S; y = -T; z = -n; }    // -Z: sc = -x, tc = -y
    const int major = face >> 1;
    const bool ox = major != 0 && (x > m || x < -m);
    const bool oy = major != 1 && (y > m || y < -m);
    if (major == 0) x = x > 0 ? m : -m;
    else if (major == 1) y = y > 0 ? m : -m;
    else z = z > 0 ? m : -m;
    int S2, T2;
    if (ox)      { nf = x > 0 ? 0 : 1; S2 = x > 0 ? -z : z; T2 = -y; }
    else if (oy) { nf = y > 0 ? 2 : 3; S2 = x; T2 = y > 0 ? z : -z; }
    else         { nf = z > 0 ? 4 : 5; S2 = z > 0 ? x : -x; T2 = -y; }
    ni = (S2 + m) >> 1;
    nj = (T2 + m) >> 1;
}

// INDEXED (here and below): `A` is an object in memory -- the kernarg segment, read through a pointer (rt_triangles.hip:
// reread_first_kernarg) --, and a lane's face record is LOADED from it by index, instead of being selected among the six
// pointers and twelve sizes held in scalar registers: thirty SGPRs that the caller then does not need.
template <bool INDEXED = false>
__device__ __forceinline__ v3 cube_texel_at(const RtFrameArgs& A, int face, int n, int x, int y, const float* lut) {
    const uint8_t* f = INDEXED ? A.face[face] : A.face[0];
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (!INDEXED && face == i) f = A.face[i];
    const uchar4 p = *reinterpret_cast<const uchar4*>(f + 4u * ((size_t)y * (size_t)n + (size_t)x));
    if (lut) return V(lut[p.x], lut[p.y], lut[p.z]);
    return V((float)p.x / 255.0f, (float)p.y / 255.0f, (float)p.z / 255.0f);
}

// One bilinear tap.  Beyond a corner no face holds the texel: a + ((b - a) + (c - a)) / 3 with a = this
// face's corner texel and b / c the corner texels of the faces across the u / v edge (Vulkan "Cube Map
// Corner Handling": the mean of the three, and exactly their value when they agree).
template <bool INDEXED = false>
__device__ inline v3 cube_tap(const RtFrameArgs& A, int face, int n, int i, int j, const float* lut) {
    const bool oi = i < 0 || i >= n, oj = j < 0 || j >= n;
    if (!oi && !oj) return cube_texel_at<INDEXED>(A, face, n, i, j, lut);
    const int ci = i < 0 ? 0 : (i >= n ? n - 1 : i), cj = j < 0 ? 0 : (j >= n ? n - 1 : j);
    int f2, i2, j2;
    if (oi && oj) {
        const v3 a = cube_texel_at<INDEXED>(A, face, n, ci, cj, lut);
        cube_fold(face, i, cj, n, f2, i2, j2);
        const v3 b = cube_texel_at<INDEXED>(A, f2, n, i2, j2, lut);
        cube_fold(face, ci, j, n, f2, i2, j2);
        const v3 c = cube_texel_at<INDEXED>(A, f2, n, i2, j2, lut);
        return add(a, divs(add(sub(b, a), sub(c, a)), 3.0f));
    }
    cube_fold(face, i, j, n, f2, i2, j2);
    return cube_texel_at<INDEXED>(A, f2, n, i2, j2, lut);
}

// A.sky_flat (host: every face is one texel and the six texels agree -- the constant sky of the
// BASELINE configs C1-C4): every tap is that texel c, lerp(c, c, w) = c + w * (c - c) is c for finite
// weights and NaN otherwise, and the corner value is c + (0 + 0) / 3 = c -- the sample is formed as
// c + (wu * 0 + wv * 0), the same values with one fetch.
// A.sky_seamless (host: six equal squares): seamless filtering as above; otherwise (not a WebGPU cube)
// the taps clamp to the edge of the selected image.
// SKY: 0 = both decided at run time (wave-uniform flags), 1 = the caller's kernel is compiled for a flat
// sky only (the filtering code is not even instantiated: the hierarchy kernel runs at its VGPR cap and must
// not pay for it in the BASELINE configs C1-C4), 2 = compiled for a textured sky only.
template <int SKY = 0, bool INDEXED = false>
__device__ inline v3 cube_sample(const RtFrameArgs& A, v3 r, const float* lut = nullptr) {
    if (SKY == 1) {
        // Flat sky, compiled in: every face is the same single texel c, so the face does not matter, and the
        // sample c + (wu * 0 + wv * 0) is c unless a weight is NaN.  wu = u - floor(u) is finite whenever
        // sc / ma is (|sc| <= ma rules out infinities), so the NaN-ness of the weights is that of the two
        // quotients: c + ((sc / ma) * 0 + (tc / ma) * 0) has the general path's value for every direction --
        // with the major axis found by three compares and no face table, texel address or floor.
        const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
        float sc, tc, ma;
        if (az >= ax && az >= ay) { sc = r.x; tc = r.y; ma = az; }
        else if (ay >= ax)        { sc = r.x; tc = r.z; ma = ay; }
        else                      { sc = r.z; tc = r.y; ma = ax; }
        const v3 c = texel(A.face[0], 1, 1, 0, 0, lut);
        const float z = (sc / ma) * 0.0f + (tc / ma) * 0.0f;      // signs of sc, tc do not matter: +-0 or NaN
        return V(c.x + z, c.y + z, c.z + z);
    }
    const float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) {
        if (r.z >= 0.0f) { face = 4; sc = r.x;  tc = -r.y; } else { face = 5; sc = -r.x; tc = -r.y; }
        ma = az;
    } else if (ay >= ax) {
        if (r.y >= 0.0f) { face = 2; sc = r.x; tc = r.z; } else { face = 3; sc = r.x; tc = -r.z; }
        ma = ay;
    } else {
        if (r.x >= 0.0f) { face = 0; sc = -r.z; tc = -r.y; } else { face = 1; sc = r.z; tc = -r.y; }
        ma = ax;
    }
    // per-lane face index: select pointer/size without a runtime-indexed register array
    const uint8_t* f = INDEXED ? A.face[face] : A.face[0];
    int w = (int)(INDEXED ? A.fw[face] : A.fw[0]), h = (int)(INDEXED ? A.fh[face] : A.fh[0]);
#pragma unroll
    for (int i = 1; i < 6; ++i)
        if (!INDEXED && face == i) { f = A.face[i]; w = (int)A.fw[i]; h = (int)A.fh[i]; }
    const float s = 0.5f * (sc / ma) + 0.5f;
    const float t = 0.5f * (tc / ma) + 0.5f;
    const float u = s * (float)w - 0.5f;
    const float v = t * (float)h - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float wu = u - fu, wv = v - fv;
    const int x0 = (int)fu, y0 = (int)fv;
    if (SKY == 1 || (SKY == 0 && A.sky_flat)) {
        const v3 c = texel(f, 1, 1, 0, 0, lut);
        const float z = wu * 0.0f + wv * 0.0f;
        return V(c.x + z, c.y + z, c.z + z);
    }
    if (SKY == 1) return V(0, 0, 0);   // not reached
    if (A.sky_seamless && x0 >= 0 && x0 + 1 < w && y0 >= 0 && y0 + 1 < w) {
        // all four taps on the selected face (all but one sample in a few hundred): the tap pairs of a row are
        // adjacent texels, one 8-byte load each; same lerps as below
        // (two 4-byte-aligned texels: memcpy leaves the width of the load to the compiler -- an 8-byte access at an
        // odd texel would be undefined in C++, even where the hardware's unaligned mode serves it)
        uint2 r0, r1;
        __builtin_memcpy(&r0, f + 4u * ((size_t)y0 * (size_t)w + (size_t)x0), 8);
        __builtin_memcpy(&r1, f + 4u * ((size_t)(y0 + 1) * (size_t)w + (size_t)x0), 8);
        auto un = [&](uint32_t p) {
            if (lut) return V(lut[p & 255u], lut[(p >> 8) & 255u], lut[(p >> 16) & 255u]);
            return V((float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f);
        };
        return lerp3(lerp3(un(r0.x), un(r0.y), wu), lerp3(un(r1.x), un(r1.y), wu), wv);
    }
    if (A.sky_seamless && x0 >= -1 && x0 < w && y0 >= -1 && y0 < w) {
        v3 top = V(0, 0, 0), row = V(0, 0, 0);
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {          // one copy of the tap code for both rows
            top = row;
            const v3 a = cube_tap<INDEXED>(A, face, w, x0, y0 + k, lut), b = cube_tap<INDEXED>(A, face, w, x0 + 1, y0 + k, lut);
            row = lerp3(a, b, wu);
        }
        return lerp3(top, row, wv);
    }
    const v3 c00 = texel(f, w, h, x0, y0, lut), c10 = texel(f, w, h, x0 + 1, y0, lut);
    const v3 c01 = texel(f, w, h, x0, y0 + 1, lut), c11 = texel(f, w, h, x0 + 1, y0 + 1, lut);
    return lerp3(lerp3(c00, c10, wu), lerp3(c01, c11, wu), wv);
}

__device__ __forceinline__ uint32_t unorm8(float c) {
    if (!(c == c)) return 0u;
    c = clampf(c, 0.0f, 1.0f);
    return (uint32_t)floorf(c * 255.0f + 0.5f);
}

// ---- frame constants unpacked from the kernarg copy of SceneParameters (RK:2-11) -------------
struct Scene {
    v3 cameraPos, forwards, right, up, lightPos;
    float lightIntensity, minIntensity;
    uint32_t bounces;
};
__device__ __forceinline__ Scene unpack_scene(const RtFrameArgs& A) {
    Scene s;
    s.cameraPos = V(A.p[0], A.p[1], A.p[2]);
    s.forwards = V(A.p[4], A.p[5], A.p[6]);
    s.right = V(A.p[8], A.p[9], A.p[10]);
    s.up = V(A.p[12], A.p[13], A.p[14]);
    s.lightPos = V(A.p[16], A.p[17], A.p[18]);
    s.lightIntensity = A.p[19];
    s.minIntensity = A.p[20];
    const float mb = A.p[21];   // u32(scene.maxBounces), RK:110: truncating, saturating
    s.bounces = 0;
    if (mb > 0.0f) s.bounces = mb >= 4294967040.0f ? 4294967295u : (uint32_t)mb;
    return s;
}

// RK:78-86
__device__ __forceinline__ v3 primary_dir(const RtFrameArgs& A, const Scene& sc, uint32_t x, uint32_t y) {
    const float hc = ((float)(int)x - (float)A.W / 2.0f) / (float)A.W * 2.0f;
    const float vc = ((float)A.H / 2.0f - (float)(int)y) / (float)A.W * 2.0f;
    return normalize(add(add(sc.forwards, scale(hc, sc.right)), scale(vc, sc.up)));
}

// RK:91-98: fog/sky compose and rgba8unorm pack
// sky = minIntensity * cube_sample(primary direction)
__device__ __forceinline__ uint32_t compose_pixel_sky(v3 sky, v3 color, float dist) {
    const float k = clampf((30.0f - dist) / 30.0f, 0.0f, 1.0f);
    const v3 px = add(scale(k, color), scale(1.0f - k, sky));
    return unorm8(px.x) | (unorm8(px.y) << 8) | (unorm8(px.z) << 16) | 0xFF000000u;
}
__device__ __forceinline__ uint32_t compose_pixel(const RtFrameArgs& A, const Scene& sc, v3 dir0, v3 color, float dist) {
    return compose_pixel_sky(scale(sc.minIntensity, cube_sample(A, dir0)), color, dist);
}

// RK:155-165: the tail of lightIntensity once the shadow ray's nearest hit (st, shit) is known
__device__ __forceinline__ float light_term(const Scene& sc, v3 dest, v3 normal, v3 sdir, float distance,
                                            bool shit, float st) {
    if (shit) {
        const v3 hp = add(sc.lightPos, scale(st, sdir));         // RK:156
        // RK:157-159: diff = length(hp - dest), `if (diff < 0.005)`.  The correctly rounded square root is monotone, so
        // sqrt(x) < 0.005f exactly when x < 0x1.a36e2cp-16, the smallest float whose root rounds to 0.005f or more (NaN: false
        // either way) -- the comparison without the sixteen instructions of an IEEE square root.
        const v3 dv = sub(hp, dest);
        if (dot(dv, dv) < 0x1.a36e2cp-16f) {
            const float power = clampf(dot(normal, V(-sdir.x, -sdir.y, -sdir.z)), sc.minIntensity, 1.0f);  // RK:160
            const float cap = sc.lightIntensity / (sc.lightIntensity + distance);                         // RK:161
            return power * cap;                                  // RK:162
        }
    }
    return sc.minIntensity;                                      // RK:165
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One atomicAdd per wave for the scene-traversal counter -- spread over RT_RAY_COUNTERS partial
// counters 256 B apart (the host adds them up): atomics on ONE address complete about 12 ns apart,
// and a 4K frame of the triangle kernels ends 129,600 waves, each with its atomic: 1.57 ms, which
// WAS the frame time of those kernels whatever else changed.
// A kernel that finds it cannot run as the host planned says so in the word behind the frame's first ray counter
// (the partial sums are RT_RAY_COUNTER_STRIDE apart, the bytes between them are free); rt_wait turns a non-zero
// word into RT_ERR_HIP instead of handing out a frame that was never written.
__device__ __forceinline__ void report_fault(unsigned long long* counter, unsigned long long code) {
    if (threadIdx.x == 0) atomicMax(counter + 1, code);
}

// ... for a kernel that has added its wave's rays up already: called by ONE lane of the wave
__device__ __forceinline__ void count_wave_rays(unsigned long long* counter, uint32_t wave_rays) {
    const uint32_t wave_id = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    atomicAdd(counter + (size_t)(wave_id % RT_RAY_COUNTERS) * (RT_RAY_COUNTER_STRIDE / 8u), (unsigned long long)wave_rays);
}

__device__ __forceinline__ void count_rays(unsigned long long* counter, uint32_t nrays) {
    const uint32_t wave_id = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned long long* part = counter + (size_t)(wave_id % RT_RAY_COUNTERS) * (RT_RAY_COUNTER_STRIDE / 8u);
    const uint64_t live = __ballot(1);
    if (live == ~0ull) {
        const uint32_t tot = wave_sum(nrays);
        if ((threadIdx.x & 63u) == 0) atomicAdd(part, (unsigned long long)tot);
    } else {
        atomicAdd(part, (unsigned long long)nrays);
    }
}

}  // namespace rtk
