/* abi_group.c -- multi-GPU through the C ABI alone (include/rt355.h): no Python, no torch.
 *   abi_group in.bin out.rgba mode
 * in.bin as for abi_render.c: u32 W, u32 H, u32 N, u32 strict, f32 params[24], f32 spheres[N][8], u8 sky[4]
 * mode "group": rt_group_create(0) -- a context per visible device joined by ncclCommInitAll --, the
 *               scene written to every member, rt_group_render(root 0) twice (frames in flight),
 *               rt_group_wait, rt_read_frame from the root's context;
 * mode "rank":  the process-per-GPU form with a world of one: rt_comm_unique_id, rt_comm_init,
 *               rt_render_gather(-1) (ncclAllGather), rt_read_frame.
 * Prints "devices=<n> rays=<sum over members> gather_ms=<root's>".  Built and driven by
 * tests/test_c_abi_gpu.py, which compares the frame with the golden / the oracle's. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt355.h"

#define CHECK(call)                                                             \
    do {                                                                        \
        int rc_ = (call);                                                       \
        if (rc_ != RT_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rt_last_error(NULL)); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

static int setup(rt_ctx* ctx, uint32_t W, uint32_t H, uint32_t N, int strict, const float* params,
                 const float* spheres, const uint8_t* sky) {
    CHECK(rt_resize(ctx, W, H));
    for (int i = 0; i < 6; ++i) CHECK(rt_write_cubemap_face(ctx, i, 1, 1, sky));
    CHECK(rt_write_params(ctx, params));
    CHECK(rt_write_spheres(ctx, spheres, N));
    CHECK(rt_set_mode(ctx, strict ? RT_MODE_STRICT : RT_MODE_FAST));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
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
    uint8_t* px = (uint8_t*)malloc((size_t)W * H * 4);
    unsigned long long rays = 0;
    float gather_ms = 0.0f;
    int devices = 0;

    if (strcmp(argv[3], "group") == 0) {
        rt_group* g = NULL;
        CHECK(rt_group_create(0, &g));
        devices = rt_group_size(g);
        for (int i = 0; i < devices; ++i)
            if (setup(rt_group_ctx(g, i), W, H, N, (int)hdr[3], params, spheres, sky)) return 1;
        CHECK(rt_group_render(g, 0));
        CHECK(rt_group_render(g, 0));                 /* a second frame in flight */
        CHECK(rt_group_wait(g));
        CHECK(rt_read_frame(rt_group_ctx(g, 0), px, (size_t)W * H * 4));
        for (int i = 0; i < devices; ++i) {
            rt_stats st;
            CHECK(rt_get_stats(rt_group_ctx(g, i), &st));
            rays += st.rays;
            if (i == 0) gather_ms = st.gather_ms;
        }
        /* misuse is refused, not undefined: a member context cannot gather on its own or change its partition */
        if (rt_render_gather(rt_group_ctx(g, 0), 0) != RT_ERR_STATE) return 4;
        if (devices > 1 && rt_set_partition(rt_group_ctx(g, 0), 1, (uint32_t)devices) != RT_ERR_STATE) return 4;
        CHECK(rt_group_destroy(g));
    } else {
        rt_ctx* ctx = NULL;
        uint8_t id[RT355_COMM_ID_BYTES];
        CHECK(rt_create(0, &ctx));
        if (rt_render_gather(ctx, 0) != RT_ERR_STATE) return 4;      /* no communicator yet */
        CHECK(rt_comm_unique_id(id));
        CHECK(rt_comm_init(ctx, id, 0, 1));
        if (setup(ctx, W, H, N, (int)hdr[3], params, spheres, sky)) return 1;
        CHECK(rt_render_gather(ctx, -1));
        CHECK(rt_wait(ctx));
        CHECK(rt_read_frame(ctx, px, (size_t)W * H * 4));
        rt_stats st;
        CHECK(rt_get_stats(ctx, &st));
        rays = st.rays;
        gather_ms = st.gather_ms;
        devices = 1;
        CHECK(rt_comm_destroy(ctx));
        CHECK(rt_render(ctx));                                         /* a plain context again */
        CHECK(rt_wait(ctx));
        CHECK(rt_destroy(ctx));
    }
    free(spheres);
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(px, 4, (size_t)W * H, o) != (size_t)W * H) return 3;
    fclose(o);
    printf("devices=%d rays=%llu gather_ms=%.3f\n", devices, rays, gather_ms);
    free(px);
    return 0;
}
