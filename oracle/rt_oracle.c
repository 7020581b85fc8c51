/*
 * rt_oracle.c -- CPU restatement of the reference's per-pixel ray-trace shader
 * with the sphere primitive.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see
 * rt_oracle.h for both statements and for the arithmetic conventions).
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 *         (see oracle/Makefile).  Never -march=native / -mfma / -ffast-math.
 *
 * Citations are relative to /root/reference/ :
 *   RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl
 *   HK = src/rendering-raycast/shaders/heatmap-kernel.wgsl
 *   RR = src/rendering-raycast/renderer-raytracing.ts
 */
#include "rt_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float length(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { return divs(a, length(a)); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
/* WGSL reflect(e1, e2) = e1 - 2 * dot(e2, e1) * e2 */
static inline v3 reflect(v3 e1, v3 e2) { return sub(e1, scale(2.0f * dot(e2, e1), e2)); }

/* SceneParameters, RK:2-11, float indices as packed by RR:157-165 */
typedef struct {
    v3 cameraPos, forwards, right, up, lightPos;
    float lightIntensity, minIntensity, maxBounces;
} scene_params;

static scene_params unpack(const float p[24]) {
    scene_params s;
    s.cameraPos = V(p[0], p[1], p[2]);
    s.forwards = V(p[4], p[5], p[6]);
    s.right = V(p[8], p[9], p[10]);
    s.up = V(p[12], p[13], p[14]);
    s.lightPos = V(p[16], p[17], p[18]);
    s.lightIntensity = p[19];
    s.minIntensity = p[20];
    s.maxBounces = p[21];
    return s;
}

/* RK:49-56 RenderState, reduced to the fields the sphere path reads.
 * WGSL zero-initialises `var renderState: RenderState;` */
typedef struct {
    float t;
    v3 normal;
    v3 diffuse_rgb; /* diffuse.w is fixed to 1 for spheres */
    int hit;
} render_state;

/* ---- HK:307-331 hitSphere ------------------------------------------------ */
static inline __attribute__((always_inline)) render_state hit_sphere(v3 o, v3 d, const float* s, float tMin, float tMax) {
    render_state rs = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0};
    v3 center = V(s[0], s[1], s[2]);
    float radius = s[7];
    v3 oc = sub(o, center);
    float a = dot(d, d);                                  /* HK:308 */
    float b = 2.0f * dot(d, oc);                          /* HK:309 */
    float c = dot(oc, oc) - radius * radius;              /* HK:310 */
    float discriminant = b * b - 4.0f * a * c;            /* HK:311  (4*a)*c */
    if (discriminant > 0.0f) {                            /* HK:316 */
        float t = (-b - sqrtf(discriminant)) / (2.0f * a);/* HK:317 */
        if (t > tMin && t < tMax) {                       /* HK:318 */
            v3 position = add(o, scale(t, d));            /* HK:319 */
            rs.normal = normalize(sub(position, center)); /* HK:320 */
            rs.t = t;
            rs.diffuse_rgb = V(s[4], s[5], s[6]);
            rs.hit = 1;
            return rs;
        }
    }
    rs.hit = 0;
    return rs;
}

rt_oracle_hit rt_oracle_hit_sphere(const float origin[3], const float dir[3],
                                   const float sphere[8], float t_min, float t_max) {
    render_state rs = hit_sphere(V(origin[0], origin[1], origin[2]), V(dir[0], dir[1], dir[2]),
                                 sphere, t_min, t_max);
    rt_oracle_hit h;
    h.t = rs.t;
    h.normal[0] = rs.normal.x; h.normal[1] = rs.normal.y; h.normal[2] = rs.normal.z;
    h.hit = rs.hit;
    return h;
}

/* ---- scene traversal: the sphere stand-in for traceTLAS (RK:168-244) -------
 * Brute-force loop called the way hitTriangle is called at RK:311-322:
 * tMin = 0.001, tMax = running nearest hit (initial 9999, RK:172). */
static render_state trace_scene(v3 o, v3 d, const float* spheres, uint32_t n) {
    render_state state = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0}; /* RK:170-171 */
    float nearestHit = 9999.0f;                           /* RK:172 */
    for (uint32_t i = 0; i < n; ++i) {
        render_state ns = hit_sphere(o, d, spheres + 8u * i, 0.001f, nearestHit);
        if (ns.hit) {                                     /* RK:318-321 */
            nearestHit = ns.t;
            state = ns;
        }
    }
    return state;
}

/* ---- cube map sample (arithmetic of the WebGPU implementation; a8) -------- */
uint8_t rt_oracle_unorm8(float c) {
    if (!(c == c)) return 0;
    c = clampf(c, 0.0f, 1.0f);
    return (uint8_t)floorf(c * 255.0f + 0.5f);
}

static inline v3 texel(const rt_oracle_face* f, int x, int y) {
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (x > (int)f->w - 1) x = (int)f->w - 1;
    if (y > (int)f->h - 1) y = (int)f->h - 1;
    const uint8_t* p = f->rgba + 4u * ((size_t)y * f->w + (size_t)x);
    return V((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
}

static inline v3 lerp3(v3 a, v3 b, float f) { return add(a, scale(f, sub(b, a))); }

static v3 cube_sample(const rt_oracle_face faces[6], v3 r) {
    float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    int face;
    float sc, tc, ma;
    /* Vulkan 1.3 "Cube Map Face Selection": z wins ties over y over x */
    if (az >= ax && az >= ay) {
        if (r.z >= 0.0f) { face = 4; sc = r.x;  tc = -r.y; }
        else             { face = 5; sc = -r.x; tc = -r.y; }
        ma = az;
    } else if (ay >= ax) {
        if (r.y >= 0.0f) { face = 2; sc = r.x; tc = r.z; }
        else             { face = 3; sc = r.x; tc = -r.z; }
        ma = ay;
    } else {
        if (r.x >= 0.0f) { face = 0; sc = -r.z; tc = -r.y; }
        else             { face = 1; sc = r.z;  tc = -r.y; }
        ma = ax;
    }
    const rt_oracle_face* f = &faces[face];
    float s = 0.5f * (sc / ma) + 0.5f;
    float t = 0.5f * (tc / ma) + 0.5f;
    float u = s * (float)f->w - 0.5f;
    float v = t * (float)f->h - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float wu = u - fu, wv = v - fv;
    int x0 = (int)fu, y0 = (int)fv;
    v3 c00 = texel(f, x0, y0), c10 = texel(f, x0 + 1, y0);
    v3 c01 = texel(f, x0, y0 + 1), c11 = texel(f, x0 + 1, y0 + 1);
    return lerp3(lerp3(c00, c10, wu), lerp3(c01, c11, wu), wv);
}

void rt_oracle_cube_sample(const rt_oracle_face faces[6], const float dir[3], float rgb[3]) {
    v3 c = cube_sample(faces, V(dir[0], dir[1], dir[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ---- RK:146-166 lightIntensity -------------------------------------------- */
static float light_intensity(const scene_params* sc, const float* spheres, uint32_t n,
                             v3 destination, v3 normal, uint64_t* rays) {
    v3 direction = normalize(sub(destination, sc->lightPos));    /* RK:147 */
    float distance = length(direction);                          /* RK:148 (quirk: ~1) */
    render_state result = trace_scene(sc->lightPos, direction, spheres, n); /* RK:150-153 */
    *rays += 1;
    if (result.hit) {                                            /* RK:155 */
        v3 hitPoint = add(sc->lightPos, scale(result.t, direction)); /* RK:156 */
        float diff = length(sub(hitPoint, destination));         /* RK:157 */
        float epsilon = 0.005f;                                  /* RK:158 */
        if (diff < epsilon) {                                    /* RK:159 */
            v3 neg = V(-direction.x, -direction.y, -direction.z);
            float power = clampf(dot(normal, neg), sc->minIntensity, 1.0f);        /* RK:160 */
            float intensityCap = sc->lightIntensity / (sc->lightIntensity + distance); /* RK:161 */
            return power * intensityCap;                         /* RK:162 */
        }
    }
    return sc->minIntensity;                                     /* RK:165 */
}

/* ---- RK:101-144 rayColor --------------------------------------------------- */
static void ray_color(const scene_params* sc, const float* spheres, uint32_t n,
                      const rt_oracle_face faces[6], v3 origin, v3 direction,
                      float out[4], uint64_t* rays) {
    float dist = 0.0f;                                           /* RK:102 */
    v3 color = V(1.0f, 1.0f, 1.0f);                              /* RK:103 */
    v3 ro = origin, rd = direction;                              /* RK:106-108 */
    /* u32(f32): truncation toward zero, saturating (negative / NaN -> 0) */
    uint32_t bounces = 0;                                        /* RK:110 */
    if (sc->maxBounces > 0.0f)
        bounces = sc->maxBounces >= 4294967040.0f ? 4294967295u : (uint32_t)sc->maxBounces;
    float affectFactor = 1.0f, sumFactor = 0.0f;                 /* RK:111-112 */
    for (uint32_t bounce = 0; bounce < bounces; ++bounce) {      /* RK:113 */
        render_state result = trace_scene(ro, rd, spheres, n);   /* RK:114 */
        *rays += 1;
        if (bounce == 0) dist = result.t;                        /* RK:116-118 */
        float nextSumFactor = affectFactor + sumFactor;          /* RK:120 */
        if (!result.hit) {                                       /* RK:122 */
            v3 sky = scale(sc->minIntensity, cube_sample(faces, rd));            /* RK:123 */
            color = divs(add(scale(sumFactor, color), scale(affectFactor, sky)),
                         nextSumFactor);                         /* RK:124 */
            break;                                               /* RK:125 */
        }
        ro = add(ro, scale(result.t, rd));                       /* RK:129 */
        rd = normalize(reflect(rd, result.normal));              /* RK:130 */
        float intensity = light_intensity(sc, spheres, n, ro, result.normal, rays); /* RK:132 */
        /* RK:133-135 with diffuse.w == 1: diffuse.rgb*1 + tex*(1-1) == diffuse.rgb */
        v3 blended = scale(intensity, result.diffuse_rgb);
        color = divs(add(scale(sumFactor, color), scale(affectFactor, blended)),
                     nextSumFactor);                             /* RK:136 */
        affectFactor = affectFactor / 2.0f;                      /* RK:139 */
        sumFactor = nextSumFactor;                               /* RK:140 */
    }
    out[0] = color.x; out[1] = color.y; out[2] = color.z; out[3] = dist;  /* RK:143 */
}

void rt_oracle_ray_color(const float params[24], const float* spheres, uint32_t n,
                         const rt_oracle_face faces[6], const float origin[3], const float dir[3],
                         float rgbd[4], uint64_t* rays) {
    scene_params sc = unpack(params);
    uint64_t r = 0;
    ray_color(&sc, spheres, n, faces, V(origin[0], origin[1], origin[2]),
              V(dir[0], dir[1], dir[2]), rgbd, &r);
    if (rays) *rays += r;
}

/* ---- RK:73-99 main ---------------------------------------------------------- */
static v3 ray_dir(const scene_params* sc, uint32_t W, uint32_t H, uint32_t x, uint32_t y) {
    (void)H;
    float hc = ((float)(int32_t)x - (float)W / 2.0f) / (float)W * 2.0f;          /* RK:78 */
    float vc = ((float)H / 2.0f - (float)(int32_t)y) / (float)W * 2.0f;          /* RK:79 */
    return normalize(add(add(sc->forwards, scale(hc, sc->right)), scale(vc, sc->up))); /* RK:82-86 */
}

void rt_oracle_ray_dir(const float params[24], uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                       float dir[3]) {
    scene_params sc = unpack(params);
    v3 d = ray_dir(&sc, W, H, x, y);
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

static void shade_pixel(const scene_params* sc, const float* spheres, uint32_t n,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t x, uint32_t y, float rgb[3], uint64_t* rays) {
    v3 dir = ray_dir(sc, W, H, x, y);
    float result[4];
    ray_color(sc, spheres, n, faces, sc->cameraPos, dir, result, rays);          /* RK:88-89 */
    v3 rayColor = V(result[0], result[1], result[2]);                            /* RK:91 */
    v3 sky = scale(sc->minIntensity, cube_sample(faces, dir));                   /* RK:92 */
    const float MAX_DISTANCE = 30.0f;                                            /* RK:94 */
    float intensity = clampf((MAX_DISTANCE - result[3]) / MAX_DISTANCE, 0.0f, 1.0f); /* RK:95 */
    v3 pixel = add(scale(intensity, rayColor), scale(1.0f - intensity, sky));    /* RK:96 */
    rgb[0] = pixel.x; rgb[1] = pixel.y; rgb[2] = pixel.z;
}

void rt_oracle_pixel(const float params[24], const float* spheres, uint32_t n,
                     const rt_oracle_face faces[6], uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                     float rgb[3], uint64_t* rays) {
    scene_params sc = unpack(params);
    uint64_t r = 0;
    shade_pixel(&sc, spheres, n, faces, W, H, x, y, rgb, &r);
    if (rays) *rays += r;
}

int rt_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int rt_oracle_render(const float params[24], const float* spheres, uint32_t n,
                     const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                     uint32_t tile_first, uint32_t tile_step,
                     uint8_t* out_rgba8, float* out_rgb, uint64_t* rays_out, int threads) {
    return rt_oracle_render_ex(params, spheres, n, faces, W, H, tile_first, tile_step, out_rgba8,
                               out_rgb, NULL, rays_out, threads);
}

int rt_oracle_render_ex(const float params[24], const float* spheres, uint32_t n,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t tile_first, uint32_t tile_step,
                        uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                        uint64_t* rays_out, int threads) {
    if (!params || !faces || (n && !spheres) || tile_step == 0) return -1;
    for (int f = 0; f < 6; ++f)
        if (!faces[f].rgba || faces[f].w == 0 || faces[f].h == 0) return -2;
    scene_params sc = unpack(params);
    uint32_t ntiles = (H + 7u) / 8u;
    uint64_t total = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    /* RR:445 dispatches ceil(W/8) x ceil(H/8) workgroups; threads outside the
     * texture write nothing (WebGPU out-of-bounds textureStore is dropped). */
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int64_t row = 0; row < (int64_t)H; ++row) {
        uint32_t y = (uint32_t)row;
        uint32_t tile = y / 8u;
        if (tile < tile_first || (tile - tile_first) % tile_step != 0 || tile >= ntiles) continue;
        uint64_t rays = 0;
        for (uint32_t x = 0; x < W; ++x) {
            float rgb[3];
            uint64_t before = rays;
            shade_pixel(&sc, spheres, n, faces, W, H, x, y, rgb, &rays);
            size_t idx = (size_t)y * W + x;
            if (out_rays_px) out_rays_px[idx] = (uint16_t)(rays - before);
            if (out_rgb) {
                out_rgb[3 * idx + 0] = rgb[0];
                out_rgb[3 * idx + 1] = rgb[1];
                out_rgb[3 * idx + 2] = rgb[2];
            }
            if (out_rgba8) {
                out_rgba8[4 * idx + 0] = rt_oracle_unorm8(rgb[0]);
                out_rgba8[4 * idx + 1] = rt_oracle_unorm8(rgb[1]);
                out_rgba8[4 * idx + 2] = rt_oracle_unorm8(rgb[2]);
                out_rgba8[4 * idx + 3] = 255;                                    /* RK:98 alpha 1.0 */
            }
        }
        total += rays;
    }
    if (rays_out) *rays_out = total;
    return 0;
}
