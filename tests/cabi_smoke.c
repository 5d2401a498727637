/* A consumer of the C ABI written in plain C (C11, -pedantic), built with gcc against
 * include/piet_metal_amd.h ALONE and linked to libpiet_metal_amd.so -- what a host in another
 * language binds to.  It does what TestApp does with the reference: encode the
 * cardioid test scene (make_cardioid, src/lib.rs:257-270) into the renderer's scene buffer through
 * the Encoder entry points, render it, read the frame back -- once as RGBA8, once stored by the
 * kernels as BGRA8 (the reference drawable's format, PietRenderer.m:29) -- and print FNV-1a hashes
 * that tests/test_gpu_parity.py compares with the oracle's (tests/golden/golden.json, "cabi_smoke").
 *
 *   gcc -std=c11 -pedantic -Wall -Wextra -Iinclude tests/cabi_smoke.c -o cabi_smoke -L... -lpiet_metal_amd -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "piet_metal_amd.h"

static uint64_t fnv1a64(const uint8_t *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int st_ = (call);                                                            \
        if (st_ != PM_OK) {                                                          \
            fprintf(stderr, "%s -> %d: %s\n", #call, st_, pm_last_error());          \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    const uint32_t w = argc > 1 ? (uint32_t)atoi(argv[1]) : 1024u;
    const uint32_t h = argc > 2 ? (uint32_t)atoi(argv[2]) : 768u;
    int err = PM_OK;
    if (pm_abi_version() != PM_ABI_VERSION) { /* the struct layouts this file was compiled against */
        fprintf(stderr, "library ABI %u, header %u\n", pm_abi_version(), PM_ABI_VERSION);
        return 1;
    }
    pm_ctx *c = pm_create(0, &err);
    if (!c) {
        fprintf(stderr, "pm_create -> %d: %s\n", err, pm_last_error());
        return 1;
    }
    CHECK(pm_resize(c, w, h));

    /* make_cardioid, through the Encoder, straight into the renderer's pinned scene buffer */
    size_t cap = 0;
    uint8_t *scene = pm_scene_buffer(c, &cap);
    pm_encoder *e = pm_encoder_new(scene, cap);
    if (!scene || !e) return 1;
    const int n = 97;
    const double pi = 3.14159265358979323846; /* std::f64::consts::PI */
    const double dth = pi * 2.0 / (double)n;
    const double cx = 1024.0, cy = 768.0, r = 750.0;
    CHECK(pm_encoder_begin_group(e, (size_t)(n - 1) * 2));
    for (int i = 1; i < n; ++i) {
        const double a0 = (double)i * dth, a1 = (double)((i * 2) % n) * dth;
        const double x0 = cx + cos(a0) * r, y0 = cy + sin(a0) * r; /* Vec2::from_angle */
        const double x1 = cx + cos(a1) * r, y1 = cy + sin(a1) * r;
        CHECK(pm_encoder_circle(e, x0, y0, 8.0));
        CHECK(pm_encoder_stroke_line(e, x0, y0, x1, y1, 2.0f, 0x000080e0u));
    }
    CHECK(pm_encoder_end_group(e));
    const size_t scene_bytes = pm_encoder_bytes_used(e);
    pm_encoder_free(e);
    printf("scene_bytes %zu scene_fnv1a64 %016llx\n", scene_bytes, (unsigned long long)fnv1a64(scene, scene_bytes));

    CHECK(pm_upload_scene(c, scene_bytes));
    uint8_t *px = (uint8_t *)malloc((size_t)w * h * 4);
    if (!px) return 1;
    CHECK(pm_render(c));
    CHECK(pm_sync(c));
    CHECK(pm_read_pixels(c, px, (size_t)w * 4, PM_FMT_RGBA8));
    printf("rgba_fnv1a64 %016llx\n", (unsigned long long)fnv1a64(px, (size_t)w * h * 4));

    CHECK(pm_set_target_format(c, PM_FMT_BGRA8));
    CHECK(pm_render(c));
    CHECK(pm_read_pixels(c, px, (size_t)w * 4, PM_FMT_BGRA8)); /* (pm_read_pixels synchronises) */
    printf("bgra_fnv1a64 %016llx\n", (unsigned long long)fnv1a64(px, (size_t)w * h * 4));

    pm_stats st;
    CHECK(pm_get_stats(c, &st));
    printf("tiles %ux%u items %u queued_tiles %u overflow %u\n", st.tiles_x, st.tiles_y, st.n_items, st.queued_tiles, st.overflow);
    free(px);
    pm_destroy(c);
    return 0;
}
