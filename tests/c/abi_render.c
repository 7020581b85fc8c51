/* abi_render.c -- a plain-C host of the C ABI (include/rt355.h): no Python, no Node, no torch.
 *   abi_render in.bin out.rgba
 * in.bin: u32 W, u32 H, u32 N, u32 strict, f32 params[24], f32 spheres[N][8], u8 sky[4]
 * Renders one frame and writes the RGBA8 pixels; prints "rays=<n>".  Built and driven by
 * tests/test_c_abi_gpu.py, which compares the frame with the oracle's. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "rt355.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != RT_OK) {                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rt_last_error(ctx)); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t hdr[4];
    float params[24];
    uint8_t sky[4];
    if (fread(hdr, 4, 4, f) != 4 || fread(params, 4, 24, f) != 24) return 2;
    const uint32_t W = hdr[0], H = hdr[1], N = hdr[2];
    float* spheres = (float*)malloc((size_t)N * 32 + 4);
    if (N && fread(spheres, 32, N, f) != N) return 2;
    if (fread(sky, 1, 4, f) != 4) return 2;
    fclose(f);

    rt_ctx* ctx = NULL;
    CHECK(rt_create(0, &ctx));
    CHECK(rt_resize(ctx, W, H));
    for (int i = 0; i < 6; ++i) CHECK(rt_write_cubemap_face(ctx, i, 1, 1, sky));
    CHECK(rt_write_params(ctx, params));
    CHECK(rt_write_spheres(ctx, spheres, N));
    free(spheres);                                   /* writeBuffer semantics: already copied */
    CHECK(rt_set_mode(ctx, hdr[3] ? RT_MODE_STRICT : RT_MODE_FAST));
    CHECK(rt_render(ctx));
    CHECK(rt_wait(ctx));
    uint8_t* px = (uint8_t*)malloc((size_t)W * H * 4);
    CHECK(rt_read_pixels(ctx, px, (size_t)W * H * 4));
    rt_stats st;
    CHECK(rt_get_stats(ctx, &st));
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(px, 4, (size_t)W * H, o) != (size_t)W * H) return 3;
    fclose(o);
    printf("rays=%llu kernel_ms=%.3f abi=%d\n", (unsigned long long)st.rays, st.kernel_ms, rt_abi_version());
    free(px);
    CHECK(rt_destroy(ctx));
    return 0;
}
