/* TEST INFRASTRUCTURE ONLY. Drives the CPU oracle (pt_oracle.c) through its
 * main paths in a build with AddressSanitizer and UndefinedBehaviorSanitizer
 * (make SAN=1 check), the counterpart of the reference's sanitized Debug build
 * (CMakeLists.txt:214-219). A small scene with all four materials: brute-force
 * and BVH intersection must agree, a render and a continued render run, the
 * two-level definition runs, and the wide-tree yardstick is computed. Any
 * sanitizer report ends the process with a non-zero status. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pt_oracle.h"

static uint32_t rng = 12345u;
static float frand(void)
{
    rng ^= rng << 13;
    rng ^= rng >> 17;
    rng ^= rng << 5;
    return (float)(rng >> 8) * (1.0f / 16777216.0f);
}

int main(void)
{
    /* a bumpy grid of 12 x 12 quads, instanced four times, one kind each */
    enum { N = 13 };
    float pos[N * N * 3], nrm[N * N * 3];
    uint32_t idx[(N - 1) * (N - 1) * 6];
    for (int i = 0; i < N; ++i)
    {
        for (int j = 0; j < N; ++j)
        {
            float *p = pos + 3 * (i * N + j), *n = nrm + 3 * (i * N + j);
            p[0] = -0.5f + (float)i / (N - 1);
            p[1] = -0.5f + (float)j / (N - 1);
            p[2] = 0.1f * frand();
            n[0] = 0.0f;
            n[1] = 0.0f;
            n[2] = 1.0f;
        }
    }
    uint32_t k = 0;
    for (int i = 0; i + 1 < N; ++i)
    {
        for (int j = 0; j + 1 < N; ++j)
        {
            const uint32_t a = (uint32_t)(i * N + j), b = a + 1, c = a + N, d = c + 1;
            const uint32_t t[6] = {a, c, b, b, c, d};
            memcpy(idx + k, t, sizeof t);
            k += 6;
        }
    }
    const pt_mesh_desc mesh = {pos, nrm, idx, N * N, (N - 1) * (N - 1) * 2};
    pt_instance_desc inst[4];
    for (uint32_t i = 0; i < 4; ++i)
    {
        memset(&inst[i], 0, sizeof inst[i]);
        inst[i].transform[0] = inst[i].transform[5] = inst[i].transform[10] = 1.0f;
        inst[i].transform[3] = -0.9f + 0.6f * (float)i;
        inst[i].transform[11] = -0.2f * (float)i;
        inst[i].mesh = 0;
        inst[i].kind = i;
    }
    enum { EW = 16, EH = 8 };
    float env[EW * EH * 4];
    for (int i = 0; i < EW * EH * 4; ++i)
    {
        env[i] = frand();
    }
    pt_scene *scene = pt_scene_create(&mesh, 1, inst, 4, env, EW, EH);
    if (scene == NULL)
    {
        fprintf(stderr, "san_check: pt_scene_create failed\n");
        return 1;
    }
    enum { R = 4000 };
    float *rays = malloc(sizeof(float) * 6 * R);
    pt_hit *h0 = malloc(sizeof(pt_hit) * R), *h1 = malloc(sizeof(pt_hit) * R);
    for (int i = 0; i < R; ++i)
    {
        float *r = rays + 6 * i;
        r[0] = 3.0f * (frand() - 0.5f);
        r[1] = 2.0f * (frand() - 0.5f);
        r[2] = 2.0f;
        r[3] = 0.6f * (frand() - 0.5f);
        r[4] = 0.6f * (frand() - 0.5f);
        r[5] = -1.0f;
    }
    pt_stats st;
    memset(&st, 0, sizeof st);
    pt_intersect(scene, rays, R, 10000.0f, 0, h0, &st);
    pt_intersect(scene, rays, R, 10000.0f, 1, h1, &st);
    int n_hit = 0;
    for (int i = 0; i < R; ++i)
    {
        if (h0[i].id != h1[i].id || (h0[i].id != PT_NO_HIT && memcmp(&h0[i], &h1[i], sizeof(pt_hit)) != 0))
        {
            fprintf(stderr, "san_check: ray %d: brute force and BVH disagree\n", i);
            return 1;
        }
        n_hit += h0[i].id != PT_NO_HIT;
    }
    pt_push_constants push;
    memset(&push, 0, sizeof push);
    push.samples_per_frame = 2;
    push.camera_position[2] = 3.5f;
    push.camera_dir_x[0] = 1.0f;
    push.camera_dir_y[1] = -1.0f;
    push.camera_dir_z[2] = -1.0f;
    push.sensor_distance = 1.0f;
    push.sensor_half_width = 0.55f;
    push.sensor_half_height = 0.41f;
    push.focus_distance = 3.5f;
    push.aperture_radius = 0.01f;
    enum { W = 40, H = 30 };
    pt_render_params rp;
    memset(&rp, 0, sizeof rp);
    rp.width = W;
    rp.height = H;
    rp.max_depth = 8;
    rp.tmax = 10000.0f;
    rp.n_frames = 2;
    rp.use_bvh = 1;
    rp.n_threads = 2;
    float *accum = calloc(W * H * 4, sizeof(float));
    uint8_t *rgba = calloc(W * H * 4, 1);
    pt_render(scene, &push, &rp, accum, rgba, &st);
    push.sample_count = 4; /* continue the film, other tone map */
    push.global_frame_count = 2;
    rp.tonemap = 1;
    pt_render(scene, &push, &rp, accum, rgba, &st);
    uint32_t pix[5] = {0, 7, W * H / 2, W * H - 2, W * H - 1};
    float pacc[5 * 4];
    uint8_t prgba[5 * 4];
    push.sample_count = 0;
    pt_render_pixels(scene, &push, &rp, pix, 5, pacc, prgba, &st);
    uint32_t *ids = malloc(sizeof(uint32_t) * W * H);
    pt_primary_hits(scene, &push, &rp, ids);
    pt_scene_set_two_level(scene, 1);
    pt_intersect(scene, rays, R, 10000.0f, 0, h1, &st);
    pt_scene_set_two_level(scene, 0);
    uint64_t wide_nodes = 0;
    const double sah = pt_scene_bvh_sah_cost(scene), wide = pt_scene_wide_sah_cost(scene, &wide_nodes);
    double lum = 0.0;
    for (int i = 0; i < W * H; ++i)
    {
        lum += accum[4 * i] + accum[4 * i + 1] + accum[4 * i + 2];
    }
    printf("san_check ok: %d of %d rays hit, mean radiance %.4f, SAH %.2f, wide SAH %.2f (%llu nodes), %llu rays\n", n_hit, R,
           lum / (3.0 * W * H), sah, wide, (unsigned long long)wide_nodes, (unsigned long long)st.rays);
    free(ids);
    free(rgba);
    free(accum);
    free(h1);
    free(h0);
    free(rays);
    pt_scene_destroy(scene);
    return isfinite(lum) && n_hit > 0 ? 0 : 1;
}
