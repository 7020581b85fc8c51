/*
 * rt_oracle.c -- CPU restatement of the reference's per-pixel ray-trace shader
 * with the sphere primitive.  TEST INFRASTRUCTURE ONLY; what pins it -- the reference's screenshot -- and
 * the arithmetic conventions are stated in rt_oracle.h.
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
    float diffuse_w;
    float tex_u, tex_v;
    int hit;
} render_state;

/* what a ray is traced against: the sphere list (brute force, HK:307-331) or the reference's
 * live triangle scene (two-level BVH, RK:168-410) */
typedef struct {
    const float* spheres;
    uint32_t n;
    const rt_oracle_tri_scene* tri;
} oscene;

static render_state trace_tlas(const rt_oracle_tri_scene* T, v3 o, v3 d, uint32_t* steps);
static void tri_work_flush(void);
/* measurement only (rt_oracle_tri_trace_px): the sequence of traversal steps of the current pixel, one byte each */
/* measurement only (rt_oracle_tri_sp_hist): pushes by stack slot, [0] BLAS [1] TLAS -- per thread, folded into the totals with
 * the work counters (tri_work_flush): nothing shared is written from the traversal loops of an OpenMP render */
static __thread uint64_t tl_sp_hist[2][24];
static uint64_t g_sp_hist[2][24];
static uint32_t* g_node_hist;   /* measurement only: visits per inner BLAS node (single-threaded use) */
static __thread uint8_t* tl_trace; static __thread size_t tl_trace_n, tl_trace_cap;
static inline void trace_code(uint8_t c) { if (tl_trace) { if (tl_trace_n < tl_trace_cap) tl_trace[tl_trace_n] = c; ++tl_trace_n; } }
static v3 tex2d_sample(const rt_oracle_face* f, float u, float v);

/* ---- HK:307-331 hitSphere ------------------------------------------------ */
static inline __attribute__((always_inline)) render_state hit_sphere(v3 o, v3 d, const float* s, float tMin, float tMax) {
    render_state rs = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 1.0f, 0.0f, 0.0f, 0};
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
static render_state trace_scene(const oscene* S, v3 o, v3 d) {
    if (S->tri) return trace_tlas(S->tri, o, d, NULL);
    const float* spheres = S->spheres;
    const uint32_t n = S->n;
    render_state state = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 1.0f, 0.0f, 0.0f, 0}; /* RK:170-171 */
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

static inline v3 texel_at(const rt_oracle_face* f, int x, int y) {
    const uint8_t* p = f->rgba + 4u * ((size_t)y * f->w + (size_t)x);
    return V((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f);
}

static inline v3 texel(const rt_oracle_face* f, int x, int y) {   /* clamp-to-edge inside one image */
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (x > (int)f->w - 1) x = (int)f->w - 1;
    if (y > (int)f->h - 1) y = (int)f->h - 1;
    return texel_at(f, x, y);
}

static inline v3 lerp3(v3 a, v3 b, float f) { return add(a, scale(f, sub(b, a))); }

/* A WebGPU cube texture is six SQUARE layers of one size (a 'cube' view needs width == height and
 * 6 array layers), and linear filtering is seamless across its faces on every backend WebGPU runs
 * on (Vulkan 1.3 "Cube Map Edge Handling": a texel beyond an edge of the selected face is taken
 * from the adjacent face; D3D12 and Metal do the same; a sampler's address modes do not apply to
 * the face-local coordinates of a cube lookup).  The C ABI accepts any six images; when they are not
 * six equal squares the lookup is not a valid WebGPU cube and stays inside the selected face
 * (clamp to edge). */
static int cube_is_seamless(const rt_oracle_face faces[6]) {
    for (int i = 0; i < 6; ++i)
        if (faces[i].w != faces[0].w || faces[i].h != faces[0].w) return 0;
    return 1;
}

/* Texel (i, j) of `face`, i or j (not both) one step outside [0, n): the texel of the adjacent face
 * that touches the crossed edge at the same position along it.  Integer geometry, exact: texel
 * centres in units of 1/n with the face planes at +-n are S = 2i+1-n, T = 2j+1-n (|.| <= n-1 inside,
 * n+1 one step outside); the 3-D point of that centre (rows of Vulkan's face table solved for
 * x,y,z) is folded over the edge -- the coordinate that left the cube is clamped to +-n and becomes
 * the major axis, the old major axis drops to +-(n-1), the third coordinate stays -- and read back
 * through the table of the new face. */
static void cube_fold(int face, int i, int j, int n, int* nf, int* ni, int* nj) {
    const int S = 2 * i + 1 - n, T = 2 * j + 1 - n;
    int p[3];   /* x, y, z */
    switch (face) {
        case 0: p[0] = n;  p[1] = -T; p[2] = -S; break;   /* +X: sc = -z, tc = -y */
        case 1: p[0] = -n; p[1] = -T; p[2] = S;  break;   /* -X: sc = +z, tc = -y */
        case 2: p[0] = S;  p[1] = n;  p[2] = T;  break;   /* +Y: sc = +x, tc = +z */
        case 3: p[0] = S;  p[1] = -n; p[2] = -T; break;   /* -Y: sc = +x, tc = -z */
        case 4: p[0] = S;  p[1] = -T; p[2] = n;  break;   /* +Z: sc = +x, tc = -y */
        default: p[0] = -S; p[1] = -T; p[2] = -n; break;  /* -Z: sc = -x, tc = -y */
    }
    const int major = face >> 1;
    int out = 0;
    for (int a = 0; a < 3; ++a)
        if (a != major && (p[a] > n - 1 || p[a] < -(n - 1))) out = a;
    p[major] = p[major] > 0 ? n - 1 : -(n - 1);
    const int pos = p[out] > 0;
    p[out] = pos ? n : -n;
    int S2, T2;
    *nf = 2 * out + (pos ? 0 : 1);
    switch (*nf) {
        case 0: S2 = -p[2]; T2 = -p[1]; break;
        case 1: S2 = p[2];  T2 = -p[1]; break;
        case 2: S2 = p[0];  T2 = p[2];  break;
        case 3: S2 = p[0];  T2 = -p[2]; break;
        case 4: S2 = p[0];  T2 = -p[1]; break;
        default: S2 = -p[0]; T2 = -p[1]; break;
    }
    *ni = (S2 + n - 1) / 2;
    *nj = (T2 + n - 1) / 2;
}

/* One bilinear tap of a seamless cube lookup.  Beyond a CORNER (both coordinates outside) no face
 * holds the texel; Vulkan ("Cube Map Corner Handling") says it should be the average of the three
 * texels that meet at the corner and must equal their common value when they agree: formed here as
 * a + ((b - a) + (c - a)) / 3 with a = this face's corner texel, b / c = the corner texels of the
 * faces across the u / v edge. */
static v3 cube_tap(const rt_oracle_face faces[6], int face, int i, int j, int n) {
    const int oi = i < 0 || i >= n, oj = j < 0 || j >= n;
    if (!oi && !oj) return texel_at(&faces[face], i, j);
    const int ci = i < 0 ? 0 : (i >= n ? n - 1 : i), cj = j < 0 ? 0 : (j >= n ? n - 1 : j);
    int f2, i2, j2;
    if (oi && oj) {
        const v3 a = texel_at(&faces[face], ci, cj);
        cube_fold(face, i, cj, n, &f2, &i2, &j2);
        const v3 b = texel_at(&faces[f2], i2, j2);
        cube_fold(face, ci, j, n, &f2, &i2, &j2);
        const v3 c = texel_at(&faces[f2], i2, j2);
        return add(a, divs(add(sub(b, a), sub(c, a)), 3.0f));
    }
    cube_fold(face, i, j, n, &f2, &i2, &j2);
    return texel_at(&faces[f2], i2, j2);
}

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
    v3 c00, c10, c01, c11;
    if (cube_is_seamless(faces) && x0 >= -1 && x0 < (int)f->w && y0 >= -1 && y0 < (int)f->w) {
        const int n = (int)f->w;
        c00 = cube_tap(faces, face, x0, y0, n);     c10 = cube_tap(faces, face, x0 + 1, y0, n);
        c01 = cube_tap(faces, face, x0, y0 + 1, n); c11 = cube_tap(faces, face, x0 + 1, y0 + 1, n);
    } else {   /* not a WebGPU cube (or a NaN direction): stay inside the selected image */
        c00 = texel(f, x0, y0);     c10 = texel(f, x0 + 1, y0);
        c01 = texel(f, x0, y0 + 1); c11 = texel(f, x0 + 1, y0 + 1);
    }
    return lerp3(lerp3(c00, c10, wu), lerp3(c01, c11, wu), wv);
}

void rt_oracle_cube_sample(const rt_oracle_face faces[6], const float dir[3], float rgb[3]) {
    v3 c = cube_sample(faces, V(dir[0], dir[1], dir[2]));
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ---- RK:146-166 lightIntensity -------------------------------------------- */
static float light_intensity(const scene_params* sc, const oscene* S,
                             v3 destination, v3 normal, uint64_t* rays) {
    v3 direction = normalize(sub(destination, sc->lightPos));    /* RK:147 */
    float distance = length(direction);                          /* RK:148 (quirk: ~1) */
    render_state result = trace_scene(S, sc->lightPos, direction); /* RK:150-153 */
    *rays += 1; trace_code('S');
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
static void ray_color(const scene_params* sc, const oscene* S,
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
        render_state result = trace_scene(S, ro, rd);            /* RK:114 */
        *rays += 1; trace_code('R');
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
        float intensity = light_intensity(sc, S, ro, result.normal, rays); /* RK:132 */
        v3 blended;
        if (S->tri) {
            v3 diffuseColor = scale(result.diffuse_w, result.diffuse_rgb);                /* RK:133 */
            v3 samplerColor = scale(1.0f - result.diffuse_w,
                                    tex2d_sample(&S->tri->mesh_tex, result.tex_u, result.tex_v)); /* RK:134 */
            blended = scale(intensity, add(diffuseColor, samplerColor));                  /* RK:135 */
        } else {
            /* RK:133-135 with diffuse.w == 1: diffuse.rgb*1 + tex*(1-1) == diffuse.rgb */
            blended = scale(intensity, result.diffuse_rgb);
        }
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
    oscene S = {spheres, n, NULL};
    ray_color(&sc, &S, faces, V(origin[0], origin[1], origin[2]),
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

static void shade_pixel(const scene_params* sc, const oscene* S,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t x, uint32_t y, float rgb[3], uint64_t* rays) {
    v3 dir = ray_dir(sc, W, H, x, y);
    float result[4];
    ray_color(sc, S, faces, sc->cameraPos, dir, result, rays);                   /* RK:88-89 */
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
    oscene S = {spheres, n, NULL};
    shade_pixel(&sc, &S, faces, W, H, x, y, rgb, &r);
    if (rays) *rays += r;
}

void rt_oracle_pixel_tri(const float params[24], const rt_oracle_tri_scene* tri,
                         const rt_oracle_face faces[6], uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                         float rgb[3], uint64_t* rays) {
    scene_params sc = unpack(params);
    uint64_t r = 0;
    oscene S = {NULL, 0, tri};
    shade_pixel(&sc, &S, faces, W, H, x, y, rgb, &r);
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

static int render_scene(const float params[24], const oscene* S,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t tile_first, uint32_t tile_step,
                        uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                        uint64_t* rays_out, int threads);

int rt_oracle_render_ex(const float params[24], const float* spheres, uint32_t n,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t tile_first, uint32_t tile_step,
                        uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                        uint64_t* rays_out, int threads) {
    if (n && !spheres) return -1;
    oscene S = {spheres, n, NULL};
    return render_scene(params, &S, faces, W, H, tile_first, tile_step, out_rgba8, out_rgb, out_rays_px,
                        rays_out, threads);
}

int rt_oracle_render_tri(const float params[24], const rt_oracle_tri_scene* tri,
                         const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                         uint32_t tile_first, uint32_t tile_step,
                         uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                         uint64_t* rays_out, int threads) {
    if (!tri || !tri->nodes || tri->n_nodes == 0 || !tri->mesh_tex.rgba) return -1;
    oscene S = {NULL, 0, tri};
    return render_scene(params, &S, faces, W, H, tile_first, tile_step, out_rgba8, out_rgb, out_rays_px,
                        rays_out, threads);
}

static int render_scene(const float params[24], const oscene* S,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t tile_first, uint32_t tile_step,
                        uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                        uint64_t* rays_out, int threads) {
    if (!params || !faces || tile_step == 0) return -1;
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
            shade_pixel(&sc, S, faces, W, H, x, y, rgb, &rays);
            size_t idx = (size_t)y * W + x;
            if (out_rays_px) out_rays_px[idx] = (uint16_t)(rays - before);
            if (x + 1u == W) tri_work_flush();
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

/* =================================================================================
 * The reference's live triangle scene: TLAS -> BLAS -> triangles (RK:168-410).
 * Buffers are the byte layouts RendererRaytracing writes (RR:169-229): indices and counts
 * travel as f32 and are converted with u32() (truncating, saturating) where the WGSL does.
 * ================================================================================= */
#define STACK_SIZE 20u                                       /* RK:71 */

static inline uint32_t u32f(float f) {                       /* WGSL u32(f32) */
    if (!(f > 0.0f)) return 0u;
    return f >= 4294967040.0f ? 4294967295u : (uint32_t)f;
}

/* `var stack: array<u32, 20>` with an index that can run past the end (RK:212-214 guards
 * one too late, RK:303-306 not at all).  WGSL makes out-of-bounds access implementation
 * defined; this restatement takes the robustness transform Tint/Dawn apply: the index is
 * clamped to the last element, for loads and for stores. */
static inline uint32_t sclamp(uint32_t i) { return i > STACK_SIZE - 1u ? STACK_SIZE - 1u : i; }

typedef struct { v3 minCorner; float leftChildIndex; v3 maxCorner; float primitiveCount; } bvh_node;

/* Work counters of the triangle path (measurement only: bench.py prices the GPU kernel's gathers with
 * them): 32-B node loads, 160-B triangle tests, 80-B instance records read.  Per thread, folded into
 * the totals row by row; rt_oracle_tri_counters reads and clears the totals. */
static __thread uint64_t tl_tri_work[3];
static uint64_t g_tri_work[3];
/* per-thread, never flushed: BLAS inner-node visits, BLAS leaf visits (rt_oracle_tri_work_px reads differences) */
static __thread uint64_t tl_tri_visits[2];
static void tri_work_flush(void) {
    for (int k = 0; k < 3; ++k) {
        if (tl_tri_work[k]) __atomic_fetch_add(&g_tri_work[k], tl_tri_work[k], __ATOMIC_RELAXED);
        tl_tri_work[k] = 0;
    }
    for (int k = 0; k < 48; ++k) {
        uint64_t* t = &tl_sp_hist[0][0] + k;
        if (*t) __atomic_fetch_add(&g_sp_hist[0][0] + k, *t, __ATOMIC_RELAXED);
        *t = 0;
    }
}
void rt_oracle_tri_counters(uint64_t out[3]) {
    tri_work_flush();
    for (int k = 0; k < 3; ++k) out[k] = __atomic_exchange_n(&g_tri_work[k], 0, __ATOMIC_RELAXED);
}

static inline bvh_node load_node(const rt_oracle_tri_scene* T, uint32_t i) {
    bvh_node n;
    tl_tri_work[0] += 1;
    if (i >= T->n_nodes) {       /* robust buffer access: out-of-range loads read the last element */
        i = T->n_nodes - 1u;
    }
    const float* p = T->nodes + 8u * (size_t)i;
    n.minCorner = V(p[0], p[1], p[2]); n.leftChildIndex = p[3];
    n.maxCorner = V(p[4], p[5], p[6]); n.primitiveCount = p[7];
    return n;
}

/* RK:395-410 */
static inline float hit_aabb(v3 o, v3 d, const bvh_node* node) {
    v3 inverseDir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);                       /* RK:396 */
    v3 t1 = mul(sub(node->minCorner, o), inverseDir);                            /* RK:397 */
    v3 t2 = mul(sub(node->maxCorner, o), inverseDir);                            /* RK:398 */
    v3 tMin = V(fminf(t1.x, t2.x), fminf(t1.y, t2.y), fminf(t1.z, t2.z));        /* RK:399 */
    v3 tMax = V(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z));        /* RK:400 */
    float t_min = fmaxf(fmaxf(tMin.x, tMin.y), tMin.z);                          /* RK:402 */
    float t_max = fminf(fminf(tMax.x, tMax.y), tMax.z);                          /* RK:403 */
    if (t_min > t_max || t_max < 0.0f) return 99999.0f;                          /* RK:405-407 */
    return t_min;                                                                /* RK:409 */
}

static inline v3 cross3(v3 a, v3 b) {
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);   /* WGSL cross */
}

/* RK:344-393 hitTriangle; tri = 40 f32 (RR:198-209): corner k at 12k: pos @+0, normal @+4,
 * uv @+8; colour vec4 @36 */
static inline render_state hit_triangle(v3 o, v3 d, const float* tri, float tMin, float tMax,
                                        const render_state* old) {
    render_state rs = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0.0f, 0.0f, 0.0f, 0};   /* RK:350-351 */
    rs.tex_u = old->tex_u; rs.tex_v = old->tex_v;                                /* RK:352 */
    v3 cornerA = V(tri[0], tri[1], tri[2]), cornerB = V(tri[12], tri[13], tri[14]), cornerC = V(tri[24], tri[25], tri[26]);
    v3 edge1 = sub(cornerB, cornerA);                                            /* RK:354 */
    v3 edge2 = sub(cornerC, cornerA);                                            /* RK:355 */
    v3 rayCrossEdge2 = cross3(d, edge2);                                         /* RK:356 */
    float det = dot(edge1, rayCrossEdge2);                                       /* RK:357 */
    if (det < 0.00001f) return rs;                                               /* RK:359-362 */
    v3 s = sub(o, cornerA);                                                      /* RK:364 */
    float u = dot(s, rayCrossEdge2);                                             /* RK:365 */
    if (u < 0.0f || u > det) return rs;                                          /* RK:366 */
    v3 sCrossEdge1 = cross3(s, edge1);                                           /* RK:370 */
    float v = dot(d, sCrossEdge1);                                               /* RK:371 */
    if (v < 0.0f || u + v > det) return rs;                                      /* RK:372 */
    float invDet = 1.0f / det;                                                   /* RK:376 */
    float t = invDet * dot(edge2, sCrossEdge1);                                  /* RK:377 */
    u = u * invDet;                                                              /* RK:378 */
    v = v * invDet;                                                              /* RK:379 */
    if (t > tMin && t < tMax) {                                                  /* RK:380 */
        float w = 1.0f - u - v;                                                  /* RK:381 */
        /* mat3x3(nA,nB,nC) * vec3(w,u,v) = nA*w + nB*u + nC*v, summed left to right */
        v3 nA = V(tri[4], tri[5], tri[6]), nB = V(tri[16], tri[17], tri[18]), nC = V(tri[28], tri[29], tri[30]);
        rs.normal = add(add(scale(w, nA), scale(u, nB)), scale(v, nC));          /* RK:382 */
        rs.diffuse_rgb = V(tri[36], tri[37], tri[38]); rs.diffuse_w = tri[39];   /* RK:384 */
        rs.t = t;                                                                /* RK:385 */
        rs.tex_u = (tri[8] * w + tri[20] * u) + tri[32] * v;                     /* RK:386 */
        rs.tex_v = (tri[9] * w + tri[21] * u) + tri[33] * v;
        rs.tex_v = 1.0f - rs.tex_v;                                              /* RK:387 */
        rs.hit = 1;
    }
    return rs;
}

/* RK:246-341 traceBLAS; blas = 20 f32 (RR:169-174): mat4 column-major + rootNodeIndex */
static render_state trace_blas(const rt_oracle_tri_scene* T, v3 o, v3 d, const float* blas,
                               float nearestHit, const render_state* renderState, uint32_t* steps) {
    const float* m = blas;   /* m[4*c + r] */
    tl_tri_work[2] += 1; trace_code('I');
    /* mat4x4 * vec4: sum over columns, left to right (RK:254-255) */
    v3 oo = V(((m[0] * o.x + m[4] * o.y) + m[8] * o.z) + m[12] * 1.0f,
              ((m[1] * o.x + m[5] * o.y) + m[9] * o.z) + m[13] * 1.0f,
              ((m[2] * o.x + m[6] * o.y) + m[10] * o.z) + m[14] * 1.0f);
    v3 od = V(((m[0] * d.x + m[4] * d.y) + m[8] * d.z) + m[12] * 0.0f,
              ((m[1] * d.x + m[5] * d.y) + m[9] * d.z) + m[13] * 0.0f,
              ((m[2] * d.x + m[6] * d.y) + m[10] * d.z) + m[14] * 0.0f);
    render_state brs = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0.0f, 0.0f, 0.0f, 0};     /* RK:258 */
    brs.t = renderState->t; brs.normal = renderState->normal;                    /* RK:259-260 */
    brs.tex_u = renderState->tex_u; brs.tex_v = renderState->tex_v;              /* RK:261 */
    brs.hit = 0;                                                                 /* RK:262 */
    bvh_node node = load_node(T, u32f(blas[16]));                                /* RK:265 */
    uint32_t stack[STACK_SIZE];
    uint32_t stackLocation = 0;                                                  /* RK:267 */
    float blasNearestHit = nearestHit;                                           /* RK:269 */
    for (;;) {                                                                   /* RK:271 */
        uint32_t primitiveCount = u32f(node.primitiveCount);                     /* RK:272 */
        uint32_t leftChildNodeIndex = u32f(node.leftChildIndex);                 /* RK:273 */
        if (primitiveCount == 0) {                                               /* RK:275 */
            tl_tri_visits[0] += 1; trace_code('N');
            if (g_node_hist && leftChildNodeIndex < T->n_nodes) g_node_hist[leftChildNodeIndex] += 1;
            if (steps) *steps += 2;                                              /* HK:242 */
            uint32_t iChild1 = leftChildNodeIndex, iChild2 = leftChildNodeIndex + 1u;
            bvh_node c1 = load_node(T, leftChildNodeIndex), c2 = load_node(T, leftChildNodeIndex + 1u);
            float distance1 = hit_aabb(oo, od, &c1);                             /* RK:279 */
            float distance2 = hit_aabb(oo, od, &c2);                             /* RK:280 */
            if (distance1 > distance2) {                                         /* RK:283-290 */
                float tmp = distance1; distance1 = distance2; distance2 = tmp;
                iChild1 = leftChildNodeIndex + 1u; iChild2 = leftChildNodeIndex;
            }
            if (distance1 > blasNearestHit) {                                    /* RK:292 */
                if (stackLocation == 0) break;
                stackLocation -= 1;
                node = load_node(T, stack[sclamp(stackLocation)]);               /* RK:297-298 */
            } else {
                node = load_node(T, iChild1);                                    /* RK:302 */
                if (distance2 < blasNearestHit) {                                /* RK:303 */
                    stack[sclamp(stackLocation)] = iChild2;                      /* RK:304 (no overflow guard) */
                    tl_sp_hist[0][stackLocation < 23u ? stackLocation : 23u] += 1;
                    stackLocation += 1;
                }
            }
        } else {
            tl_tri_visits[1] += 1;
            for (uint32_t i = 0; i < primitiveCount; ++i) {                      /* RK:311 */
                uint32_t li = i + leftChildNodeIndex;
                if (li >= T->n_tri_lookup) li = T->n_tri_lookup - 1u;
                uint32_t ti = u32f(T->tri_lookup[li]);
                if (ti >= T->n_triangles) ti = T->n_triangles - 1u;
                render_state ns = hit_triangle(oo, od, T->triangles + 40u * (size_t)ti, 0.001f,
                                               blasNearestHit, &brs);            /* RK:312-316 */
                tl_tri_work[1] += 1; trace_code('T');
                if (steps) *steps += 1;                                          /* HK:279 */
                if (ns.hit) { blasNearestHit = ns.t; brs = ns; }                 /* RK:318-321 */
            }
            if (stackLocation == 0) break;                                       /* RK:324 */
            stackLocation -= 1;
            node = load_node(T, stack[sclamp(stackLocation)]);                   /* RK:328-329 */
        }
    }
    if (brs.hit) {                                                               /* RK:334-338 */
        /* transpose(inverseModel) * vec4(n, 0): row r of the transpose = column r of m */
        v3 n = brs.normal;
        v3 tn = V(((m[0] * n.x + m[1] * n.y) + m[2] * n.z) + m[3] * 0.0f,
                  ((m[4] * n.x + m[5] * n.y) + m[6] * n.z) + m[7] * 0.0f,
                  ((m[8] * n.x + m[9] * n.y) + m[10] * n.z) + m[11] * 0.0f);
        brs.normal = normalize(tn);
    }
    return brs;                                                                  /* RK:340 */
}

/* RK:168-244 traceTLAS */
static render_state trace_tlas(const rt_oracle_tri_scene* T, v3 o, v3 d, uint32_t* steps) {
    render_state renderState = {0.0f, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, 0.0f, 0.0f, 0.0f, 0};   /* RK:170-171 */
    float nearestHit = 9999.0f;                                                  /* RK:172 */
    bvh_node node = load_node(T, 0);                                             /* RK:175 */
    uint32_t stack[STACK_SIZE];
    uint32_t stackLocation = 0;
    for (;;) {                                                                   /* RK:179 */
        uint32_t modelCount = u32f(node.primitiveCount);                         /* RK:180 */
        uint32_t leftChildNodeIndex = u32f(node.leftChildIndex);                 /* RK:181 */
        if (modelCount == 0) {                                                   /* RK:183 */
            trace_code('n');
            if (steps) *steps += 2;                                              /* HK:143 */
            uint32_t iChild1 = leftChildNodeIndex, iChild2 = leftChildNodeIndex + 1u;
            bvh_node c1 = load_node(T, leftChildNodeIndex), c2 = load_node(T, leftChildNodeIndex + 1u);
            float distance1 = hit_aabb(o, d, &c1);                               /* RK:186 */
            float distance2 = hit_aabb(o, d, &c2);                               /* RK:187 */
            if (distance1 > distance2) {                                         /* RK:190-196 */
                float tmp = distance1; distance1 = distance2; distance2 = tmp;
                iChild1 = leftChildNodeIndex + 1u; iChild2 = leftChildNodeIndex;
            }
            if (distance1 > nearestHit) {                                        /* RK:198 */
                if (stackLocation == 0) break;
                stackLocation -= 1;
                node = load_node(T, stack[sclamp(stackLocation)]);
            } else {
                node = load_node(T, iChild1);                                    /* RK:208 */
                if (distance2 < nearestHit) {                                    /* RK:209 */
                    stack[sclamp(stackLocation)] = iChild2;
                    tl_sp_hist[1][stackLocation < 23u ? stackLocation : 23u] += 1;
                    stackLocation += 1;
                    /* RK:212-214 guards with `>`; the heatmap twin (steps != NULL) with `>=`, HK:168 */
                    if (steps ? stackLocation >= STACK_SIZE : stackLocation > STACK_SIZE) stackLocation = STACK_SIZE - 1u;
                }
            }
        } else {
            for (uint32_t i = 0; i < modelCount; ++i) {                          /* RK:220 */
                uint32_t li = i + leftChildNodeIndex;
                if (li >= T->n_blas_lookup) li = T->n_blas_lookup - 1u;
                uint32_t bi = u32f(T->blas_lookup[li]);
                if (bi >= T->n_blas) bi = T->n_blas - 1u;
                render_state ns = trace_blas(T, o, d, T->blas + 20u * (size_t)bi, nearestHit,
                                             &renderState, steps);               /* RK:221-225 */
                if (ns.hit) { nearestHit = ns.t; renderState = ns; }             /* RK:227-230 */
            }
            if (stackLocation == 0) break;                                       /* RK:233 */
            stackLocation -= 1;
            node = load_node(T, stack[sclamp(stackLocation)]);                   /* RK:237-238 */
        }
    }
    return renderState;
}

/* textureSampleLevel(meshTex, texSamp, uv, 0).rgb (RK:134).  texSamp is the CUBE MAP's sampler
 * (RR:345-347, cubemap-material.ts:25-32): addressModeU repeat, addressModeV default
 * clamp-to-edge, mag/min linear.  Same lerp form as the cube sample. */
static v3 tex2d_sample(const rt_oracle_face* f, float u, float v) {
    float x = u * (float)f->w - 0.5f;
    float y = v * (float)f->h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float wx = x - fx, wy = y - fy;
    int w = (int)f->w, h = (int)f->h;
    /* floor to integer with saturation so that absurd coordinates stay defined */
    int x0 = fx >= 2147483520.0f ? 2147483520 : (fx <= -2147483520.0f ? -2147483520 : (int)fx);
    int y0 = fy >= 2147483520.0f ? 2147483520 : (fy <= -2147483520.0f ? -2147483520 : (int)fy);
    int xa = ((x0 % w) + w) % w, xb = (((x0 + 1) % w) + w) % w;                  /* repeat */
    int ya = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0);                             /* clamp-to-edge */
    int yb = y0 + 1 < 0 ? 0 : (y0 + 1 > h - 1 ? h - 1 : y0 + 1);
    v3 c00 = texel(f, xa, ya), c10 = texel(f, xb, ya), c01 = texel(f, xa, yb), c11 = texel(f, xb, yb);
    return lerp3(lerp3(c00, c10, wx), lerp3(c01, c11, wx), wy);
}

/* ---- heatmap kernel (HK:63-83): traversal cost of the primary ray -> grey ----------------------
 * HK adds 2 per inner node visited (two hitAABB calls; TLAS HK:143, BLAS HK:242) and 1 per
 * triangle tested (HK:279); one bounce (HK:96); pixel = clamp(traces / 300, 0, 1) (HK:79-82).
 * Its traversal arithmetic is RK's; its struct declarations differ (no light fields HK:1-7,
 * rootNodeIndex as vec4 HK:29-32, colour vec3 HK:19) but read the same bytes of the shared
 * buffers. */

/* Measurement only (tools/tri_path_stats.py): the work of every pixel's path over the triangle scene, out_px[4 * (y*W+x)]
 * = {scene traversals, BLAS inner-node visits, triangle tests, instance records read}.  Single-threaded. */
int rt_oracle_tri_work_px(const float params[24], const rt_oracle_tri_scene* tri, const rt_oracle_face faces[6],
                          uint32_t W, uint32_t H, uint32_t* out_px) {
    if (!tri || !tri->nodes || tri->n_nodes == 0 || !tri->mesh_tex.rgba || !out_px) return -1;
    scene_params sc = unpack(params);
    oscene S = {NULL, 0, tri};
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            float rgb[3];
            uint64_t rays = 0;
            const uint64_t v0 = tl_tri_visits[0], t0 = tl_tri_work[1], i0 = tl_tri_work[2];
            shade_pixel(&sc, &S, faces, W, H, x, y, rgb, &rays);
            uint32_t* o = out_px + 4u * ((size_t)y * W + x);
            o[0] = (uint32_t)rays; o[1] = (uint32_t)(tl_tri_visits[0] - v0);
            o[2] = (uint32_t)(tl_tri_work[1] - t0); o[3] = (uint32_t)(tl_tri_work[2] - i0);
        }
    tri_work_flush();
    return 0;
}


/* Measurement only (tools/tri_sched_sim.py): the step sequence of every pixel's path, one byte per step -- 'n' TLAS inner
 * node, 'I' instance entered, 'N' BLAS inner node, 'T' triangle test, 'R' / 'S' reflection / shadow ray complete --,
 * pixel p's steps at codes[offsets[p] .. offsets[p+1]).  Returns the number of bytes needed (codes may be too small). */
void rt_oracle_tri_sp_hist(uint64_t out[48]) { tri_work_flush(); memcpy(out, g_sp_hist, sizeof g_sp_hist); memset(g_sp_hist, 0, sizeof g_sp_hist); }
void rt_oracle_tri_node_hist(uint32_t* hist) { g_node_hist = hist; }   /* hist[left child index] += 1 per inner-node visit; NULL: off */

uint64_t rt_oracle_tri_trace_px(const float params[24], const rt_oracle_tri_scene* tri, const rt_oracle_face faces[6],
                                uint32_t W, uint32_t H, uint8_t* codes, uint64_t cap, uint64_t* offsets) {
    scene_params sc = unpack(params);
    oscene S = {NULL, 0, tri};
    uint64_t at = 0;
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            float rgb[3];
            uint64_t rays = 0;
            offsets[(size_t)y * W + x] = at;
            tl_trace = codes ? codes + (at < cap ? at : cap) : (uint8_t*)&rays; tl_trace_n = 0; tl_trace_cap = codes && at < cap ? cap - at : 0;
            shade_pixel(&sc, &S, faces, W, H, x, y, rgb, &rays);
            at += tl_trace_n;
        }
    offsets[(size_t)W * H] = at;
    tl_trace = NULL;
    tri_work_flush();
    return at;
}

int rt_oracle_heatmap_tri(const float params[24], const rt_oracle_tri_scene* tri, uint32_t W, uint32_t H,
                          uint8_t* out_rgba8, uint32_t* out_steps, int threads) {
    if (!params || !tri || !tri->nodes || tri->n_nodes == 0) return -1;
    scene_params sc = unpack(params);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t row = 0; row < (int64_t)H; ++row) {
        for (uint32_t x = 0; x < W; ++x) {
            v3 dir = ray_dir(&sc, W, H, x, (uint32_t)row);
            uint32_t steps = 0;
            (void)trace_tlas(tri, sc.cameraPos, dir, &steps);
            size_t idx = (size_t)row * W + x;
            if (out_steps) out_steps[idx] = steps;
            if (out_rgba8) {
                float g = clampf((float)steps / 300.0f, 0.0f, 1.0f);             /* HK:79-82 */
                uint8_t q = rt_oracle_unorm8(g);
                out_rgba8[4 * idx + 0] = q; out_rgba8[4 * idx + 1] = q; out_rgba8[4 * idx + 2] = q;
                out_rgba8[4 * idx + 3] = 255;
            }
        }
    }
    return 0;
}

/* Nearest hit of arbitrary rays against the triangle scene (test hook for the brute-force
 * cross-check in tests/test_triangles_cpu.py): out_t[i] = t or -1, out_tri[i] = triangle index or -1. */
int rt_oracle_trace_tri_rays(const rt_oracle_tri_scene* tri, uint32_t n, const float* origins,
                             const float* dirs, float* out_t, int32_t* out_tri) {
    if (!tri || !origins || !dirs || !out_t) return -1;
    for (uint32_t i = 0; i < n; ++i) {
        render_state rs = trace_tlas(tri, V(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                                     V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), NULL);
        out_t[i] = rs.hit ? rs.t : -1.0f;
        if (out_tri) out_tri[i] = rs.hit ? 0 : -1;
    }
    return 0;
}
