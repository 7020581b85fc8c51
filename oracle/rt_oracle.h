/*
 * rt_oracle.h -- CPU restatement of the reference ray-trace compute shader.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the C-ABI library
 * librt355.so, the Python/Node host layers) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY PINNED BY THE REFERENCE'S ONE HELD OUTPUT.  The reference (GmxMahdi/compute-raytracer) ships no
 * tests, golden images or known-answer vectors, and its WGSL cannot be executed in the build container (no
 * WebGPU implementation).  What it does hold is info/sample_settings.png, a browser screenshot of its own scene.
 * tests/golden/make_ref_scene.py rebuilds that scene from the reference's OBJ files and sky box, finds the scene
 * state the GUI shows only rounded (tools/pin_fit.py) and compares this oracle's frame with the screenshot's canvas:
 * outside the pixels that depend on the one asset the reference does not ship (mousey's diffuse texture) every sky
 * pixel is within one level of 255, 99.98 % of all pixels are, 95 % are identical (tests/golden/ref_pin.json,
 * re-derived by tests/test_ref_pin.py).  That pins ray generation, cube-face order / orientation / filtering, the
 * TLAS / BLAS / triangle traversal, shading, the shadow test, the running-mean bounce weights, fog and
 * quantisation.  Not pinned by it: the sphere primitive (dead code upstream: HK:307-331 is followed to the letter
 * and cross-checked by an independent numpy restatement, oracle/rt_oracle_np.py), the mesh texture path, and the
 * rounding of exact ties in the rgba8unorm store (the fitted scene has none).
 *
 * Reference files followed:
 *   RK = src/rendering-raycast/shaders/raytracer-kernel.wgsl
 *   HK = src/rendering-raycast/shaders/heatmap-kernel.wgsl
 *   RR = src/rendering-raycast/renderer-raytracing.ts
 *
 * Arithmetic conventions (where WGSL leaves precision to the implementation the
 * oracle takes the correctly-rounded IEEE-754 binary32 reading, no FMA
 * contraction, left-to-right evaluation as the WGSL grammar parses it):
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   length(v)     = sqrtf(dot(v,v))
 *   normalize(v)  = v / length(v)            (componentwise IEEE division)
 *   reflect(e1,n) = e1 - (2*dot(n,e1))*n     (WGSL spec definition)
 *   clamp(x,lo,hi)= min(max(x,lo),hi)
 *   rgba8unorm store = floor(clamp(c,0,1)*255 + 0.5), NaN -> 0
 *   cube sample   = Vulkan/WebGPU major-axis face selection, bilinear inside the
 *                   face with clamp-to-edge, lerp written a + (b-a)*f so that a
 *                   constant face returns its colour exactly; texel = byte/255.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    uint32_t w, h;
    const uint8_t* rgba; /* w*h*4 bytes, row-major, row 0 = top */
} rt_oracle_face;

typedef struct {
    float t;
    float normal[3];
    int hit;
} rt_oracle_hit;

/* HK:307-331 (commented-out hitSphere).  sphere = 8 floats {cx,cy,cz,_,r,g,b,radius}
 * (layout of the commented struct Sphere, RK:13-17). */
rt_oracle_hit rt_oracle_hit_sphere(const float origin[3], const float dir[3],
                                   const float sphere[8], float t_min, float t_max);

/* RK:78-86 ray generation for pixel (x,y) of a W x H target. */
void rt_oracle_ray_dir(const float params[24], uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                       float dir[3]);

/* textureSampleLevel(skyTex, texSamp, dir, 0).rgb  (RK:92, RK:123) */
void rt_oracle_cube_sample(const rt_oracle_face faces[6], const float dir[3], float rgb[3]);

/* rgba8unorm quantisation of one channel (RK:58, RK:98). */
uint8_t rt_oracle_unorm8(float c);

/* RK:101-144 rayColor: returns rgb + dist, adds the number of scene traversals
 * (primary/reflection rays RK:114 + shadow rays RK:153) to *rays. */
void rt_oracle_ray_color(const float params[24], const float* spheres, uint32_t n,
                         const rt_oracle_face faces[6], const float origin[3], const float dir[3],
                         float rgbd[4], uint64_t* rays);

/* RK:73-99 for every pixel of the 8-row tiles t = tile_first, tile_first+tile_step, ...
 * (tile t covers rows 8t .. 8t+7).  tile_first=0, tile_step=1 renders the full frame.
 * out_rgba8 : full W*H*4 frame, only the selected rows are written (may be NULL)
 * out_rgb   : full W*H*3 float frame of the pre-quantisation pixelColor (may be NULL)
 * rays_out  : total scene traversals over the rendered pixels (may be NULL)
 * threads   : OpenMP threads (<=0: all)
 */
int rt_oracle_render(const float params[24], const float* spheres, uint32_t n,
                     const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                     uint32_t tile_first, uint32_t tile_step,
                     uint8_t* out_rgba8, float* out_rgb, uint64_t* rays_out, int threads);

/* As rt_oracle_render, additionally writing the number of scene traversals of each rendered
 * pixel to out_rays_px[y*W + x] (may be NULL). */
int rt_oracle_render_ex(const float params[24], const float* spheres, uint32_t n,
                        const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                        uint32_t tile_first, uint32_t tile_step,
                        uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                        uint64_t* rays_out, int threads);

/* RK:73-99 for the single pixel (x,y): pre-quantisation pixelColor; adds its scene
 * traversals to *rays (may be NULL). */
void rt_oracle_pixel(const float params[24], const float* spheres, uint32_t n,
                     const rt_oracle_face faces[6], uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                     float rgb[3], uint64_t* rays);

/* ---- the reference's live triangle scene (RK:168-410): buffers exactly as RR:169-229 packs them */
typedef struct {
    const float* triangles;   uint32_t n_triangles;    /* 40 f32 each: per corner {pos.xyz,_, nrm.xyz,_, uv.xy,_,_}, colour vec4 @36 (RR:198-209) */
    const float* nodes;       uint32_t n_nodes;        /* 8 f32 each: min.xyz, leftChildIndex, max.xyz, primitiveCount (RR:184-192, 212-223) */
    const float* blas;        uint32_t n_blas;         /* 20 f32 each: inverseModel column-major, rootNodeIndex, 3 pad (RR:169-174) */
    const float* tri_lookup;  uint32_t n_tri_lookup;   /* f32 indices (RR:225-229) */
    const float* blas_lookup; uint32_t n_blas_lookup;  /* f32 indices (RR:177-181) */
    rt_oracle_face mesh_tex;                            /* meshTex rgba8unorm (material.ts:61-65) */
} rt_oracle_tri_scene;

/* RK:73-166 over the triangle scene; arguments as rt_oracle_render_ex. */
int rt_oracle_render_tri(const float params[24], const rt_oracle_tri_scene* tri,
                         const rt_oracle_face faces[6], uint32_t W, uint32_t H,
                         uint32_t tile_first, uint32_t tile_step,
                         uint8_t* out_rgba8, float* out_rgb, uint16_t* out_rays_px,
                         uint64_t* rays_out, int threads);
void rt_oracle_pixel_tri(const float params[24], const rt_oracle_tri_scene* tri,
                         const rt_oracle_face faces[6], uint32_t W, uint32_t H, uint32_t x, uint32_t y,
                         float rgb[3], uint64_t* rays);

/* HK:63-83, the heatmap kernel: out_steps (may be NULL) receives the raw `traces` count. */
int rt_oracle_heatmap_tri(const float params[24], const rt_oracle_tri_scene* tri, uint32_t W, uint32_t H,
                          uint8_t* out_rgba8, uint32_t* out_steps, int threads);

/* Nearest hit (RK:168-244) of n arbitrary rays: out_t[i] = t, or -1 when nothing is hit. */
int rt_oracle_trace_tri_rays(const rt_oracle_tri_scene* tri, uint32_t n, const float* origins,
                             const float* dirs, float* out_t, int32_t* out_tri);

/* Work the triangle path did since the last call (all threads): out[0] 32-B node loads, out[1] 160-B
 * triangle tests, out[2] 80-B instance records.  Reading clears.  Measurement only (bench.py --config TRI). */
void rt_oracle_tri_counters(uint64_t out[3]);

/* Measurement only: per pixel {scene traversals, BLAS inner-node visits, triangle tests, instance records read}
 * of the path RK:73-166 traces over the triangle scene (out_px: W*H*4 words).  Single-threaded. */
int rt_oracle_tri_work_px(const float params[24], const rt_oracle_tri_scene* tri, const rt_oracle_face faces[6],
                          uint32_t W, uint32_t H, uint32_t* out_px);

/* Measurement only: every pixel's sequence of traversal steps, one byte each ('n' TLAS inner node, 'I' instance entered,
 * 'N' BLAS inner node, 'T' triangle test, 'R' / 'S' reflection / shadow ray complete); pixel p's at
 * codes[offsets[p] .. offsets[p+1]) (offsets: W*H+1 entries).  Returns the bytes needed.  Single-threaded. */
uint64_t rt_oracle_tri_trace_px(const float params[24], const rt_oracle_tri_scene* tri, const rt_oracle_face faces[6],
                                uint32_t W, uint32_t H, uint8_t* codes, uint64_t cap, uint64_t* offsets);

/* Measurement only: while `hist` is set, every BLAS inner-node visit adds 1 to hist[left child index] (n_nodes words). */
void rt_oracle_tri_node_hist(uint32_t* hist);
/* Measurement only: pushes by stack slot since the last call, out[0..23] traceBLAS, out[24..47] traceTLAS (slot 23: beyond). */
void rt_oracle_tri_sp_hist(uint64_t out[48]);

int rt_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
