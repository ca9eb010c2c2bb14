/* The CPU oracle under -fsanitize=address,undefined (and, for the threaded frame, -fsanitize=thread):
 * one small synthetic cloud through every stage, the literal radix model, both frame entry points. */
#include "../../oracle/gs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long s = 42;
static float uni(void) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (float)(s >> 40) / 16777216.0f;
}

int main(void) {
    const uint32_t n = 3000, w = 200, h = 120;
    float* aos = (float*)calloc((size_t)n * GSO_FLOATS_PER_GAUSSIAN, sizeof(float));
    for (uint32_t i = 0; i < n; ++i) {
        float* g = aos + (size_t)i * GSO_FLOATS_PER_GAUSSIAN;
        const float d = 0.5f + 19.5f * uni();
        g[0] = d * (w / (float)h) * (-1.5f + 3.0f * uni()); g[1] = d * (-1.5f + 3.0f * uni()); g[2] = d;
        for (int a = 0; a < 3; ++a) g[4 + a] = expf(-3.0f + uni());
        float q[4], l = 0; for (int a = 0; a < 4; ++a) { q[a] = uni() - 0.5f; l += q[a] * q[a]; }
        l = 1.0f / sqrtf(l + 1e-12f); for (int a = 0; a < 4; ++a) g[8 + a] = q[a] * l;
        for (int c = 0; c < 3; ++c) g[12 + c] = -1.5f + 3.0f * uni();
        g[15] = 0.1f + 0.9f * uni();
        for (int k = 16; k < 76; ++k) if ((k & 3) != 3) g[k] = 0.1f * (uni() - 0.5f);
    }
    gso_params p;
    gso_default_params(&p, w, h);
    const float pos[3] = {0, 0, 0};
    gso_camera_matrices(pos, 0.0f, 0.0f, w / (float)h, p.near_plane, p.far_plane, p.view, p.proj);

    uint8_t* a = (uint8_t*)malloc((size_t)w * h * 4);
    uint8_t* b = (uint8_t*)malloc((size_t)w * h * 4);
    double t[5];
    const uint32_t e = gso_frame(&p, aos, n, a, t);
    for (uint32_t th = 1; th <= 8; th += 3) {
        memset(b, 0, (size_t)w * h * 4);
        const uint32_t e2 = gso_frame_mt(&p, aos, n, b, t, th);
        if (e2 != e || memcmp(a, b, (size_t)w * h * 4) != 0) { printf("threaded frame differs (%u threads)\n", th); return 1; }
    }
    /* literal radix model against the stable sort on the emitted list */
    const uint32_t tiles = gso_num_tiles_x(w, 16) * gso_num_tiles_y(h, 16);
    const uint32_t cap = gso_capacity(n, tiles);
    float* color = (float*)calloc((size_t)n * 4, 4);
    float* cov = (float*)calloc((size_t)n * 4, 4);
    uint32_t *lt = malloc((size_t)cap * 4), *ld = malloc((size_t)cap * 4), *li = malloc((size_t)cap * 4);
    uint32_t *mt = malloc((size_t)cap * 4), *md = malloc((size_t)cap * 4), *mi = malloc((size_t)cap * 4);
    const uint64_t counter = gso_init_sort_list(&p, aos, n, cap, color, cov, NULL, lt, ld, li);
    memcpy(mt, lt, (size_t)cap * 4); memcpy(md, ld, (size_t)cap * 4); memcpy(mi, li, (size_t)cap * 4);
    gso_sort_stable(lt, ld, li, (uint32_t)counter);
    gso_radix_sort_literal(mt, md, mi, cap, counter, gso_num_sort_bits(tiles));
    if (memcmp(lt, mt, (size_t)counter * 4) || memcmp(ld, md, (size_t)counter * 4) || memcmp(li, mi, (size_t)counter * 4)) {
        printf("literal radix model differs from the stable sort\n");
        return 1;
    }
    printf("sanitize_oracle ok: E=%u\n", e);
    free(aos); free(a); free(b); free(color); free(cov); free(lt); free(ld); free(li); free(mt); free(md); free(mi);
    return 0;
}
