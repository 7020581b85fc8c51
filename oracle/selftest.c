/* selftest.c -- drives every entry point of the oracle on tiny inputs; built with
 * -fsanitize=address,undefined by `make -C oracle asan` (sanitizers are a CPU-only tool on this
 * pool).  TEST INFRASTRUCTURE (see rt_oracle.h).  Exit code 0 = ran clean. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_oracle.h"

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static float frand(uint32_t* s, float lo, float hi) { return lo + (hi - lo) * (float)lcg(s) / 16777216.0f; }

int main(void) {
    uint32_t seed = 355;
    float params[24] = {0};
    params[0] = 0.06f; params[1] = 2.7f; params[2] = 3.3f;
    params[5] = -0.2756f; params[6] = -0.9613f;     /* forwards */
    params[8] = 1.0f;                               /* right */
    params[13] = 0.9613f; params[14] = -0.2756f;    /* up */
    params[17] = 5.0f; params[19] = 3.0f; params[20] = 0.3f; params[21] = 4.0f;
    uint8_t skypix[6][4 * 4 * 4];
    rt_oracle_face faces[6];
    for (int f = 0; f < 6; ++f) {
        for (int i = 0; i < 64; ++i) skypix[f][i] = (uint8_t)lcg(&seed);
        faces[f].w = 4; faces[f].h = 4; faces[f].rgba = skypix[f];
    }
    /* spheres */
    enum { N = 37 };
    float sph[N * 8];
    for (int i = 0; i < N; ++i) {
        float* r = sph + 8 * i;
        r[0] = frand(&seed, -6, 6); r[1] = frand(&seed, 0.2f, 3); r[2] = frand(&seed, -14, -3); r[3] = 0;
        r[4] = frand(&seed, 0.2f, 1); r[5] = frand(&seed, 0.2f, 1); r[6] = frand(&seed, 0.2f, 1);
        r[7] = frand(&seed, 0.1f, 1.2f);
    }
    sph[0] = 0; sph[1] = -100; sph[2] = 0; sph[7] = 100;
    const uint32_t W = 45, H = 29;
    uint8_t* img = (uint8_t*)calloc((size_t)W * H, 4);
    float* rgb = (float*)calloc((size_t)W * H, 12);
    uint16_t* cnt = (uint16_t*)calloc((size_t)W * H, 2);
    uint64_t rays = 0, sum = 0;
    if (rt_oracle_render_ex(params, sph, N, faces, W, H, 0, 1, img, rgb, cnt, &rays, 2)) return 1;
    if (rt_oracle_render(params, sph, N, faces, W, H, 1, 3, img, NULL, &rays, 1)) return 1;
    if (rt_oracle_render(params, NULL, 0, faces, W, H, 0, 1, img, NULL, &rays, 1)) return 1;
    float px[3];
    rt_oracle_pixel(params, sph, N, faces, W, H, W - 1, H - 1, px, &rays);
    float d[3];
    rt_oracle_ray_dir(params, W, H, 3, 4, d);
    float o[3] = {params[0], params[1], params[2]}, rgbd[4];
    rt_oracle_ray_color(params, sph, N, faces, o, d, rgbd, &rays);
    rt_oracle_hit h = rt_oracle_hit_sphere(o, d, sph, 0.001f, 9999.0f);
    float c[3];
    rt_oracle_cube_sample(faces, d, c);
    sum += (uint64_t)h.hit + rt_oracle_unorm8(c[0]);

    /* triangle scene: two instances of a 4-triangle pyramid + a floor quad, hand-built buffers */
    float tri[6 * 40];
    memset(tri, 0, sizeof tri);
    const float P[5][3] = {{-1, 0, -1}, {1, 0, -1}, {1, 0, 1}, {-1, 0, 1}, {0, 1.5f, 0}};
    const int F[6][3] = {{0, 4, 1}, {1, 4, 2}, {2, 4, 3}, {3, 4, 0}, {0, 1, 2}, {0, 2, 3}};
    for (int t = 0; t < 6; ++t) {
        for (int k = 0; k < 3; ++k) {
            float* cp = tri + 40 * t + 12 * k;
            cp[0] = P[F[t][k]][0]; cp[1] = P[F[t][k]][1]; cp[2] = P[F[t][k]][2];
            cp[4] = 0; cp[5] = 1; cp[6] = 0;
            cp[8] = (float)k * 0.5f; cp[9] = (float)(t & 1);
        }
        tri[40 * t + 36] = 0.9f; tri[40 * t + 37] = 0.6f; tri[40 * t + 38] = 0.3f; tri[40 * t + 39] = 0.5f;
    }
    /* nodes: [0] TLAS root inner -> children 1,2 (leaves with one BLAS each); [3] BLAS root inner ->
     * 4,5; [4] leaf tris 0..3; [5] leaf tris 4..5 */
    float nodes[6 * 8] = {
        -99, -99, -99, 1, 99, 99, 99, 0,
        -99, -99, -99, 0, 99, 99, 99, 1,
        -99, -99, -99, 1, 99, 99, 99, 1,
        -1, 0, -1, 4, 1, 1.5f, 1, 0,
        -1, 0, -1, 0, 1, 1.5f, 1, 4,
        -1, 0, -1, 4, 1, 0, 1, 2,
    };
    float blas[2 * 20];
    memset(blas, 0, sizeof blas);
    for (int b = 0; b < 2; ++b) {
        float* m = blas + 20 * b;
        m[0] = m[5] = m[10] = m[15] = 1;
        m[12] = b ? 2.5f : -2.5f; m[14] = 6.0f;      /* inverse of a translation to (-/+2.5, 0, -6) */
        m[16] = 3;
    }
    float tl[6] = {0, 1, 2, 3, 4, 5}, bl[2] = {0, 1};
    uint8_t tex[3 * 2 * 4];
    for (int i = 0; i < 24; ++i) tex[i] = (uint8_t)lcg(&seed);
    rt_oracle_tri_scene T;
    T.triangles = tri; T.n_triangles = 6; T.nodes = nodes; T.n_nodes = 6; T.blas = blas; T.n_blas = 2;
    T.tri_lookup = tl; T.n_tri_lookup = 6; T.blas_lookup = bl; T.n_blas_lookup = 2;
    T.mesh_tex.w = 3; T.mesh_tex.h = 2; T.mesh_tex.rgba = tex;
    if (rt_oracle_render_tri(params, &T, faces, W, H, 0, 1, img, rgb, cnt, &rays, 2)) return 1;
    rt_oracle_pixel_tri(params, &T, faces, W, H, W / 2, H / 2, px, &rays);
    uint32_t* steps = (uint32_t*)calloc((size_t)W * H, 4);
    if (rt_oracle_heatmap_tri(params, &T, W, H, img, steps, 1)) return 1;
    float to[6] = {0, 1, 0, -2.5f, 0.5f, 3}, td[6] = {0, 0, -1, 0, 0, -1}, tt[2];
    if (rt_oracle_trace_tri_rays(&T, 2, to, td, tt, NULL)) return 1;
    for (uint32_t i = 0; i < W * H; ++i) sum += img[4 * i] + steps[i] + cnt[i];
    printf("selftest ok rays=%llu sum=%llu t=%.3f %.3f threads=%d\n", (unsigned long long)rays,
           (unsigned long long)sum, tt[0], tt[1], rt_oracle_max_threads());
    free(img); free(rgb); free(cnt); free(steps);
    return 0;
}
