/*
 * gs_oracle.c -- CPU restatement of the reference's splat hot path.  TEST INFRASTRUCTURE ONLY
 * (see gs_oracle.h for the rules, the pinning status and the numeric contract).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fexcess-precision=standard
 * Every function cites the reference file:line it follows, relative to
 * /root/reference/vkGaussianSplatting/ (S/ = Resources/Shaders/).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include "gs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static inline float clampf(float x, float lo, float hi) { /* GLSL clamp = min(max(x,lo),hi) */
    float t = x > lo ? x : lo;                           /* max(x, lo) */
    return t < hi ? t : hi;                              /* min(t, hi) */
}
static inline float maxf(float a, float b) { return a > b ? a : b; }
static inline int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* GLSL int(float): truncate; out-of-range saturates (hardware behaviour), NaN -> 0. */
static inline int f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}
/* GLSL uint(float): truncate; saturates at 0 / 0xFFFFFFFF, NaN -> 0. */
static inline uint32_t f2u_sat(float x) {
    if (x != x) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    if (x <= 0.0f) return 0u;
    return (uint32_t)x;
}

/* m is column-major: element (row r, col c) = m[c*4+r].  GLSL `M * v`:
 * ((M[0]*v.x + M[1]*v.y) + M[2]*v.z) + M[3]*v.w, per component. */
static inline void mat4_mul_vec4(const float* m, const float v[4], float out[4]) {
    for (int r = 0; r < 4; ++r) {
        float acc = m[0 * 4 + r] * v[0];
        acc = acc + m[1 * 4 + r] * v[1];
        acc = acc + m[2 * 4 + r] * v[2];
        acc = acc + m[3 * 4 + r] * v[3];
        out[r] = acc;
    }
}

/* 3x3 column-major matrices as m[col][row]; C = A*B: C[j][i] = sum_k A[k][i]*B[j][k]. */
typedef struct { float m[3][3]; } mat3;
static inline mat3 mat3_mul(const mat3* a, const mat3* b) {
    mat3 c;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) {
            float acc = a->m[0][i] * b->m[j][0];
            acc = acc + a->m[1][i] * b->m[j][1];
            acc = acc + a->m[2][i] * b->m[j][2];
            c.m[j][i] = acc;
        }
    return c;
}
static inline mat3 mat3_transpose(const mat3* a) {
    mat3 t;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) t.m[j][i] = a->m[i][j];
    return t;
}

/* ------------------------------------------------------------------------------------------ */
/* host formulas                                                                              */
/* ------------------------------------------------------------------------------------------ */

void gso_default_params(gso_params* p, uint32_t width, uint32_t height) {
    memset(p, 0, sizeof(*p));
    p->width = width;
    p->height = height;
    p->near_plane = 0.1f;   /* Camera.cpp:4 */
    p->far_plane = 100.0f;  /* Camera.cpp:5 */
    p->tile_size = 16;      /* Renderer.h:146, Common.glsl:12 */
    p->ndc_cull = 1.3f;     /* Common.glsl:5 */
    p->in_view_limit = 0.8f; /* Common.glsl:9 */
    p->fov_y = 3.1415f * 0.5f; /* Common.glsl:2 */
    p->row_begin = 0;
    p->row_end = gso_num_tiles_y(height, 16);
    for (int i = 0; i < 4; ++i) p->view[i * 5] = p->proj[i * 5] = 1.0f;
}

/* Renderer.cpp:696-701 */
uint32_t gso_num_tiles_x(uint32_t width, uint32_t tile) { return (width + tile - 1) / tile; }
uint32_t gso_num_tiles_y(uint32_t height, uint32_t tile) { return (height + tile - 1) / tile; }

/* Renderer.cpp:703-710 */
uint32_t gso_ceil_pow2(uint32_t x) {
    uint32_t num = 1;
    while (num < x) num *= 2;
    return num;
}

/* Renderer.cpp:725: numSortElements = ceilPow2(numGaussians + 64*16*numTiles) */
uint32_t gso_capacity(uint32_t n, uint32_t num_tiles) {
    return gso_ceil_pow2(n + 64u * 16u * num_tiles);
}

/* RadixSort.cpp:7-16 (getMinNumBits) and 203-204 */
uint32_t gso_num_sort_bits(uint32_t num_tiles) {
    uint32_t x = num_tiles - 1u, min_bits = 0;
    for (int i = 31; i >= 0; --i)
        if ((x >> i) & 1u) { min_bits = (uint32_t)i + 1u; break; }
    uint32_t sort_bits = 32u + min_bits;
    return ((sort_bits + 4u - 1u) / 4u) * 4u;
}

/* Common.glsl:53 `tan(FOV_Y * 0.5f)` -- constant expression, folded on the host. */
float gso_tan_half_fov(float fov_y) { return (float)tan((double)(fov_y * 0.5f)); }

/* Pinned exp (see gs_oracle.h).  exp(x) = 2^t, t = x*log2(e); n = rint(t); r = t-n in
 * [-0.5,0.5]; 2^r by a degree-6 polynomial in Horner form with fused multiply-adds; scale by
 * 2^n exactly.  t is clamped to [-126,126] so the scaling never leaves the normal range
 * (callers only use results for x <= 0, and anything below 2^-126 is far below 1/255). */
float gso_exp(float x) {
    float t = x * 0x1.715476p+0f; /* log2(e) rounded to float */
    t = t > -126.0f ? t : -126.0f;
    t = t < 126.0f ? t : 126.0f;
    float n = rintf(t);
    float r = t - n;
    float p = 0x1.42059ap-13f;
    p = fmaf(p, r, 0x1.5f3e12p-10f);
    p = fmaf(p, r, 0x1.3b2d40p-7f);
    p = fmaf(p, r, 0x1.c6aeeap-5f);
    p = fmaf(p, r, 0x1.ebfbdcp-3f);
    p = fmaf(p, r, 0x1.62e430p-1f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(p, (int)n);
}

/* The form the HIP blend loop evaluates gso_exp in (csrc/gs_render.hip, exp_pinned_live): one max in place of the two
 * clamps, n = rint(t) by adding and subtracting 1.5 * 2^23, ldexp as an integer add of n to the exponent field (the low
 * bits of t + 1.5 * 2^23 are n).  Stated here ONLY so that a test can compare it with gso_exp over every float of the
 * domain the loop uses it on (x <= 0 and NaN): gso_exp_live_mismatches walks the floats in [lo, hi] by bit pattern
 * (every stride-th one). */
static float gso_exp_live(float x) {
    float t = fmaxf(x * 0x1.715476p+0f, -126.0f);
    float tn = t + 12582912.0f;
    float n = tn - 12582912.0f;
    float r = t - n;
    float p = 0x1.42059ap-13f;
    p = fmaf(p, r, 0x1.5f3e12p-10f);
    p = fmaf(p, r, 0x1.3b2d40p-7f);
    p = fmaf(p, r, 0x1.c6aeeap-5f);
    p = fmaf(p, r, 0x1.ebfbdcp-3f);
    p = fmaf(p, r, 0x1.62e430p-1f);
    p = fmaf(p, r, 1.0f);
    uint32_t pb, tb;
    memcpy(&pb, &p, 4); memcpy(&tb, &tn, 4);
    pb += tb << 23;
    memcpy(&p, &pb, 4);
    return p;
}
uint64_t gso_exp_live_mismatches(float lo, float hi, uint32_t stride, float* first_bad) {
    /* lo <= hi <= 0: negative floats are ordered by DEscending bit pattern */
    uint32_t a, b;
    uint64_t bad = 0;
    memcpy(&a, &hi, 4); memcpy(&b, &lo, 4);
    if (hi == 0.0f) a = 0x80000000u;                 /* -0: the first pattern of the negative range */
    for (uint64_t bits = a; bits <= b; bits += stride ? stride : 1u) {
        const uint32_t u = (uint32_t)bits;
        float x, e0, e1;
        memcpy(&x, &u, 4);
        e0 = gso_exp(x); e1 = gso_exp_live(x);
        if (memcmp(&e0, &e1, 4) != 0) { if (!bad && first_bad) *first_bad = x; ++bad; }
    }
    {   /* +0, NaN (a lane whose exponent is NaN counts as live: both forms must send it to 2^-126), -inf */
        const float specials[4] = {0.0f, NAN, -NAN, -INFINITY};
        for (int i = 0; i < 4; ++i) {
            float e0 = gso_exp(specials[i]), e1 = gso_exp_live(specials[i]);
            if (memcmp(&e0, &e1, 4) != 0) { if (!bad && first_bad) *first_bad = specials[i]; ++bad; }
        }
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 1: InitSortList                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* S/Common/Common.glsl:17-30.  rot.x = r (scalar), rot.yzw = x,y,z.  The GLSL mat3x3
 * constructor is COLUMN-major, so the nine expressions fill col0, col1, col2 in turn. */
static mat3 get_rot_mat(const float rot[4]) {
    const float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    mat3 m;
    m.m[0][0] = 1.0f - 2.0f * y * y - 2.0f * z * z;
    m.m[0][1] = 2.0f * x * y - 2.0f * r * z;
    m.m[0][2] = 2.0f * x * z + 2.0f * r * y;
    m.m[1][0] = 2.0f * x * y + 2.0f * r * z;
    m.m[1][1] = 1.0f - 2.0f * x * x - 2.0f * z * z;
    m.m[1][2] = 2.0f * y * z - 2.0f * r * x;
    m.m[2][0] = 2.0f * x * z - 2.0f * r * y;
    m.m[2][1] = 2.0f * y * z + 2.0f * r * x;
    m.m[2][2] = 1.0f - 2.0f * x * x - 2.0f * y * y;
    return m;
}

/* S/Common/Common.glsl:32-78 */
static void get_covariance(const gso_params* p, float tan_fov_y, const float scale[3],
                           const float rot[4], const float pos_v_in[4], float cov[3]) {
    const float width = (float)p->width, height = (float)p->height;
    float pos_v[4] = {pos_v_in[0], pos_v_in[1], pos_v_in[2], pos_v_in[3]};

    mat3 rot_mat = get_rot_mat(rot);                       /* :41 */
    mat3 scale_mat;                                        /* :42-44 */
    memset(&scale_mat, 0, sizeof(scale_mat));
    scale_mat.m[0][0] = scale[0];
    scale_mat.m[1][1] = scale[1];
    scale_mat.m[2][2] = scale[2];
    mat3 rs = mat3_mul(&rot_mat, &scale_mat);              /* :45 */
    mat3 rs_t = mat3_transpose(&rs);
    mat3 sigma = mat3_mul(&rs, &rs_t);                     /* :46 */

    mat3 w;                                                /* :49-51 upper-left 3x3 of view */
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) w.m[c][r] = p->view[c * 4 + r];

    const float tan_fov_x = tan_fov_y * width / height;    /* :54 */
    const float focal_x = width / (2.0f * tan_fov_x);      /* :55 */
    const float focal_y = height / (2.0f * tan_fov_y);     /* :56 */

    const float lim_x = tan_fov_x * p->in_view_limit;      /* :60 */
    const float lim_y = tan_fov_y * p->in_view_limit;
    const float temp_x = pos_v[0] / pos_v[2];              /* :61 */
    const float temp_y = pos_v[1] / pos_v[2];
    pos_v[0] = clampf(temp_x, -lim_x, lim_x) * pos_v[2];   /* :62 */
    pos_v[1] = clampf(temp_y, -lim_y, lim_y) * pos_v[2];   /* :63 */

    mat3 j;                                                /* :65-67, column-major */
    memset(&j, 0, sizeof(j));
    j.m[0][0] = focal_x / pos_v[2];
    j.m[1][1] = focal_y / pos_v[2];
    j.m[2][0] = -(focal_x * pos_v[0]) / (pos_v[2] * pos_v[2]);
    j.m[2][1] = -(focal_y * pos_v[1]) / (pos_v[2] * pos_v[2]);
    mat3 jw = mat3_mul(&j, &w);                            /* :68 */
    mat3 jw_t = mat3_transpose(&jw);
    mat3 tmp = mat3_mul(&jw, &sigma);                      /* :69 left-to-right */
    mat3 sp = mat3_mul(&tmp, &jw_t);

    cov[0] = sp.m[0][0];                                   /* :71 */
    cov[1] = sp.m[0][1];
    cov[2] = sp.m[1][1];
    cov[0] += 0.3f;                                        /* :74-75 */
    cov[2] += 0.3f;
}

/* S/Common/Common.glsl:80-89 (xy only; z,w are unused by every caller) */
static void get_screen_pos(const gso_params* p, const float pos_v[4], float* sx, float* sy) {
    float q[4];
    mat4_mul_vec4(p->proj, pos_v, q);                      /* :82 */
    float x = q[0] / q[3];                                 /* :83 */
    float y = q[1] / q[3];
    y = -y;                                                /* :84 */
    x = (x + 1.0f) * 0.5f;                                 /* :85 */
    y = (y + 1.0f) * 0.5f;
    *sx = x * (float)p->width;                             /* :86 */
    *sy = y * (float)p->height;
}

/* S/Common/Common.glsl:94-138 */
static void sh_eval4(const float dir[3], float sh[16]) {
    const float fX = -dir[0], fY = -dir[1], fZ = dir[2];   /* :99-101 */
    float fC0, fC1, fS0, fS1, fTmpA, fTmpB, fTmpC;
    const float fZ2 = fZ * fZ;

    sh[0] = 0.2820947917738781f;
    sh[2] = 0.4886025119029199f * fZ;
    sh[6] = 0.9461746957575601f * fZ2 + -0.31539156525252f;
    sh[12] = fZ * (1.865881662950577f * fZ2 + -1.119528997770346f);
    fC0 = fX;
    fS0 = fY;

    fTmpA = -0.48860251190292f;
    sh[3] = fTmpA * fC0;
    sh[1] = fTmpA * fS0;
    fTmpB = -1.092548430592079f * fZ;
    sh[7] = fTmpB * fC0;
    sh[5] = fTmpB * fS0;
    fTmpC = -2.285228997322329f * fZ2 + 0.4570457994644658f;
    sh[13] = fTmpC * fC0;
    sh[11] = fTmpC * fS0;
    fC1 = fX * fC0 - fY * fS0;
    fS1 = fX * fS0 + fY * fC0;

    fTmpA = 0.5462742152960395f;
    sh[8] = fTmpA * fC1;
    sh[4] = fTmpA * fS1;
    fTmpB = 1.445305721320277f * fZ;
    sh[14] = fTmpB * fC1;
    sh[10] = fTmpB * fS1;
    fC0 = fX * fC1 - fY * fS1;
    fS0 = fX * fS1 + fY * fC1;

    fTmpC = -0.5900435899266435f;
    sh[15] = fTmpC * fC0;
    sh[9] = fTmpC * fS0;
}

/* S/Common/Common.glsl:141-170.  coeffs = 16 vec4 (xyz used). */
static void sh_color(const float dir[3], const float* coeffs, uint32_t mode, float rgb[3]) {
    float basis[16];
    sh_eval4(dir, basis);
    float res[3] = {0.0f, 0.0f, 0.0f};
    if (mode == 0) {
        for (int i = 0; i < 16; ++i)
            for (int c = 0; c < 3; ++c) res[c] = res[c] + coeffs[i * 4 + c] * basis[i];
    } else if (mode == 1) {
        for (int i = 1; i < 16; ++i)
            for (int c = 0; c < 3; ++c) res[c] = res[c] + coeffs[i * 4 + c] * basis[i];
        for (int c = 0; c < 3; ++c) res[c] = res[c] - 0.5f;
    } else if (mode == 2) {
        for (int c = 0; c < 3; ++c) res[c] = res[c] + coeffs[0 * 4 + c] * basis[0];
    }
    for (int c = 0; c < 3; ++c) {
        res[c] = res[c] + 0.5f;                            /* :165 */
        rgb[c] = maxf(res[c], 0.0f);                       /* :166, no upper clamp */
    }
}

/* S/ComputeShaders/InitSortList.comp:70-80.  float(MAX_UINT32) == 4294967296.0f. */
static uint32_t get_depth_key(const gso_params* p, float view_z) {
    float nd = (-view_z - p->near_plane) / (p->far_plane - p->near_plane);
    nd = clampf(nd, 0.0f, 1.0f);
    return f2u_sat(nd * 4294967296.0f);
}

/* S/ComputeShaders/InitSortList.comp:47-68 */
static void get_tile_extents(const gso_params* p, float sx, float sy, const float cov[3],
                             int grid_w, int grid_h, uint32_t ext[4]) {
    const float ts = (float)p->tile_size;
    float det = cov[0] * cov[2] - cov[1] * cov[1];         /* :49 */
    float m = (cov[0] + cov[2]) * 0.5f;                    /* :53 */
    float lambda0 = m + sqrtf(maxf(m * m - det, 0.0f));    /* :54 */
    float lambda1 = m - sqrtf(maxf(m * m - det, 0.0f));    /* :55 */
    float radius = ceilf(3.0f * sqrtf(maxf(lambda0, lambda1))); /* :56 */
    ext[0] = (uint32_t)clampi(f2i_sat((sx - radius) / ts), 0, grid_w);     /* :61 */
    ext[1] = (uint32_t)clampi(f2i_sat((sy - radius) / ts), 0, grid_h);     /* :62 */
    {
        int t = f2i_sat((sx + radius) / ts);
        ext[2] = (uint32_t)clampi(t == 2147483647 ? t : t + 1, 0, grid_w); /* :63 */
        t = f2i_sat((sy + radius) / ts);
        ext[3] = (uint32_t)clampi(t == 2147483647 ? t : t + 1, 0, grid_h); /* :64 */
    }
}

/* S/ComputeShaders/InitSortList.comp:82-151, threads visited in ascending index (N8). */
uint64_t gso_init_sort_list(const gso_params* p, const float* aos, uint32_t n, uint32_t capacity,
                            float* color, float* cov_out, gso_splat* splats,
                            uint32_t* list_tile, uint32_t* list_depth, uint32_t* list_id) {
    const int grid_w = (int)gso_num_tiles_x(p->width, p->tile_size);   /* :107-110 */
    const int grid_h = (int)gso_num_tiles_y(p->height, p->tile_size);
    const float tan_fov_y = gso_tan_half_fov(p->fov_y);
    uint64_t counter = 0;

    /* Subrenderer.cpp:42-46: list <- 0xFFFFFFFF */
    if (list_tile) memset(list_tile, 0xFF, (size_t)capacity * 4);
    if (list_depth) memset(list_depth, 0xFF, (size_t)capacity * 4);
    if (list_id) memset(list_id, 0xFF, (size_t)capacity * 4);

    for (uint32_t g = 0; g < n; ++g) {
        const float* rec = aos + (size_t)g * GSO_FLOATS_PER_GAUSSIAN;
        const float* pos = rec + 0;
        const float* scale = rec + 4;
        const float* rot = rec + 8;
        const float* sh = rec + 12;
        if (splats) memset(&splats[g], 0, sizeof(gso_splat));

        float world[4] = {pos[0], pos[1], pos[2], 1.0f};
        float view_pos[4];
        mat4_mul_vec4(p->view, world, view_pos);           /* :93 */
        if (-view_pos[2] <= p->near_plane) continue;       /* :94 */

        float ndc[4];
        mat4_mul_vec4(p->proj, view_pos, ndc);             /* :98 */
        const float ndc_x = ndc[0] / ndc[3];               /* :99 */
        const float ndc_y = ndc[1] / ndc[3];
        if (fabsf(ndc_x) > p->ndc_cull || fabsf(ndc_y) > p->ndc_cull) continue; /* :100 */

        const uint32_t depth_key = get_depth_key(p, view_pos[2]); /* :104 */

        float cov[3];
        get_covariance(p, tan_fov_y, scale, rot, view_pos, cov);  /* :113-120 */
        float sx, sy;
        get_screen_pos(p, view_pos, &sx, &sy);
        uint32_t ext[4];
        get_tile_extents(p, sx, sy, cov, grid_w, grid_h, ext);    /* :121 */

        /* :124-127 colour + covariance stored for every non-culled splat (N6) */
        float d[3] = {pos[0] - p->cam_pos[0], pos[1] - p->cam_pos[1], pos[2] - p->cam_pos[2]};
        const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        float dir[3] = {d[0] / len, d[1] / len, d[2] / len};
        float rgb[3];
        sh_color(dir, sh, p->sh_mode, rgb);
        color[(size_t)g * 4 + 0] = rgb[0];
        color[(size_t)g * 4 + 1] = rgb[1];
        color[(size_t)g * 4 + 2] = rgb[2];
        color[(size_t)g * 4 + 3] = sh[3];                  /* shCoeffs[0].a = opacity */
        cov_out[(size_t)g * 4 + 0] = cov[0];
        cov_out[(size_t)g * 4 + 1] = cov[1];
        cov_out[(size_t)g * 4 + 2] = cov[2];

        /* multi-GPU extension (not in the reference): this rank only emits its tile-row band;
         * with row_begin=0,row_end=grid_h this is the identity. */
        uint32_t y0 = ext[1] > p->row_begin ? ext[1] : p->row_begin;
        uint32_t y1 = ext[3] < p->row_end ? ext[3] : p->row_end;
        if (y1 < y0) y1 = y0;

        if (splats) {
            splats[g].visible = 1;
            splats[g].depth_key = depth_key;
            splats[g].min_x = ext[0];
            splats[g].min_y = ext[1];
            splats[g].max_x = ext[2];
            splats[g].max_y = ext[3];
            splats[g].screen_x = sx;
            splats[g].screen_y = sy;
        }

        /* :130-150 */
        const uint32_t num = (ext[2] - ext[0]) * (y1 - y0);
        const uint64_t id_offset = counter;
        counter += num;
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = ext[0]; x < ext[2]; ++x) {
                const uint32_t tile_key = y * (uint32_t)grid_w + x;
                const uint32_t id_local = (y - y0) * (ext[2] - ext[0]) + (x - ext[0]);
                const uint64_t id = id_offset + id_local;
                if (id < capacity && list_tile) {          /* :143 overflow: silently dropped */
                    list_tile[id] = tile_key;
                    list_depth[id] = depth_key;
                    list_id[id] = g;
                }
            }
    }
    return counter;
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 2a: sort semantics (N9): stable by the 64-bit key                                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct { uint64_t key; uint32_t id; uint32_t seq; } sort_rec;

static void merge_sort(sort_rec* a, sort_rec* tmp, size_t n) {
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
            while (i < mid) tmp[k++] = a[i++];
            while (j < hi) tmp[k++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(sort_rec));
    }
}

void gso_sort_stable(uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t e) {
    if (e == 0) return;
    sort_rec* a = (sort_rec*)malloc((size_t)e * sizeof(sort_rec));
    sort_rec* t = (sort_rec*)malloc((size_t)e * sizeof(sort_rec));
    for (uint32_t i = 0; i < e; ++i) {
        a[i].key = ((uint64_t)tile[i] << 32) | depth[i];
        a[i].id = id[i];
        a[i].seq = i;
    }
    merge_sort(a, t, e);
    for (uint32_t i = 0; i < e; ++i) {
        tile[i] = (uint32_t)(a[i].key >> 32);
        depth[i] = (uint32_t)a[i].key;
        id[i] = a[i].id;
    }
    free(a);
    free(t);
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 2b: literal model of the radix shaders, WORK_GROUP_SIZE = 64 (RadixSort.h:38)        */
/* ------------------------------------------------------------------------------------------ */

#define RS_WG 64u
#define RS_BINS 16u

/* digit selection, RadixSortCount.comp:58-73 */
static inline uint32_t rs_digit_count(uint32_t tile, uint32_t depth, uint32_t shift) {
    if (shift < 32u - 4u + 1u) return (depth >> shift) & 15u;
    if (shift >= 32u) return (tile >> (shift - 32u)) & 15u;
    return ((tile >> (shift - 32u)) | (depth >> shift)) & 15u; /* unreachable for 4-bit steps */
}

void gso_radix_sort_literal(uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t cap,
                            uint64_t counter, uint32_t num_sort_bits) {
    /* RadixSortIndirectSetup.comp:25-37 */
    const uint32_t num_elems = counter < cap ? (uint32_t)counter : cap;
    const uint32_t num_groups = (num_elems + RS_WG - 1u) / RS_WG;
    const uint32_t num_reduce_blocks = (num_groups + RS_WG - 1u) / RS_WG;
    const uint32_t num_reduce_elems = num_reduce_blocks * RS_BINS;

    uint32_t* sum_table = (uint32_t*)calloc((size_t)RS_BINS * (num_groups + 1u), 4);
    uint32_t* reduce = (uint32_t*)calloc((size_t)num_reduce_elems + 1u, 4);
    /* ping-pong buffer, cleared to 0xFFFFFFFF every frame (RadixSort.cpp:676-692) */
    uint32_t* pp_tile = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* pp_depth = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* pp_id = (uint32_t*)malloc((size_t)cap * 4);
    memset(pp_tile, 0xFF, (size_t)cap * 4);
    memset(pp_depth, 0xFF, (size_t)cap * 4);
    memset(pp_id, 0xFF, (size_t)cap * 4);

    uint32_t *src_t = tile, *src_d = depth, *src_i = id;
    uint32_t *dst_t = pp_tile, *dst_d = pp_depth, *dst_i = pp_id;

    for (uint32_t shift = 0; shift < num_sort_bits; shift += 4u) { /* RadixSort.cpp:309 */
        /* --- Count (RadixSortCount.comp:40-91): per group 16-bin histogram -> sumTable */
        for (uint32_t g = 0; g < num_groups; ++g) {
            uint32_t hist[RS_BINS] = {0};
            for (uint32_t l = 0; l < RS_WG; ++l) {
                uint32_t t = g * RS_WG + l;
                if (t < num_elems) hist[rs_digit_count(src_t[t], src_d[t], shift)]++;
            }
            for (uint32_t b = 0; b < RS_BINS; ++b) sum_table[b * num_groups + g] = hist[b];
        }
        /* --- Reduce (RadixSortReduce.comp:34-72) */
        for (uint32_t grp = 0; grp < num_reduce_elems; ++grp) {
            uint32_t bin = grp / num_reduce_blocks;
            uint32_t base = (grp % num_reduce_blocks) * RS_WG;
            uint32_t sum = 0;
            for (uint32_t l = 0; l < RS_WG; ++l) {
                uint32_t di = base + l;
                if (di < num_groups) sum += sum_table[bin * num_groups + di];
            }
            reduce[grp] = sum;
        }
        /* --- Scan (RadixSortScan.comp:29-71): exclusive scan in place */
        {
            uint32_t run = 0;
            for (uint32_t i = 0; i < num_reduce_elems; ++i) {
                uint32_t v = reduce[i];
                reduce[i] = run;
                run += v;
            }
        }
        /* --- ScanAdd (RadixSortScanAdd.comp:34-66) */
        for (uint32_t grp = 0; grp < num_reduce_elems; ++grp) {
            uint32_t bin = grp / num_reduce_blocks;
            uint32_t base = (grp % num_reduce_blocks) * RS_WG;
            uint32_t run = reduce[grp];
            for (uint32_t l = 0; l < RS_WG; ++l) {
                uint32_t di = base + l;
                if (di < num_groups) {
                    uint32_t v = sum_table[bin * num_groups + di];
                    sum_table[bin * num_groups + di] = run;
                    run += v;
                }
            }
        }
        /* --- Scatter (RadixSortScatter.comp:58-171): stable in-group rank + global offset.
         * The two 2-bit split rounds (:91-135) implement a stable sort of the 64 keys by
         * digit; padding keys (~0, digit 15 at every shift) sort last and are dropped by
         * totalOffset < numSortElements (:163). */
        for (uint32_t g = 0; g < num_groups; ++g) {
            uint64_t key[RS_WG];
            uint32_t val[RS_WG], dig[RS_WG], order[RS_WG];
            uint32_t hist[RS_BINS] = {0}, pre[RS_BINS];
            for (uint32_t l = 0; l < RS_WG; ++l) {
                uint32_t t = g * RS_WG + l;
                key[l] = t < num_elems ? (((uint64_t)src_t[t] << 32) | src_d[t]) : ~(uint64_t)0;
                val[l] = t < num_elems ? src_i[t] : 0u;
                dig[l] = (uint32_t)(key[l] >> shift) & 15u;
                hist[dig[l]]++;
            }
            uint32_t run = 0;
            for (uint32_t b = 0; b < RS_BINS; ++b) { pre[b] = run; run += hist[b]; }
            {   /* stable counting sort of lanes by digit == result of the split rounds */
                uint32_t fill[RS_BINS];
                memcpy(fill, pre, sizeof(fill));
                for (uint32_t l = 0; l < RS_WG; ++l) order[fill[dig[l]]++] = l;
            }
            for (uint32_t li = 0; li < RS_WG; ++li) {       /* li = localIndex after re-arrange */
                uint32_t l = order[li];
                uint32_t d = dig[l];
                uint32_t global_off = sum_table[d * num_groups + g];   /* :72, :153 */
                uint32_t local_off = li - pre[d];                      /* :157 */
                uint32_t total = global_off + local_off;               /* :160 */
                if (total < num_elems) {                               /* :163 */
                    dst_t[total] = (uint32_t)(key[l] >> 32);
                    dst_d[total] = (uint32_t)key[l];
                    dst_i[total] = val[l];
                }
            }
        }
        /* ping-pong (RadixSort.cpp:638-641) */
        uint32_t* s;
        s = src_t; src_t = dst_t; dst_t = s;
        s = src_d; src_d = dst_d; dst_d = s;
        s = src_i; src_i = dst_i; dst_i = s;
    }
    /* After the loop `src` names the buffer the last Scatter wrote; the reference's
     * shared_ptr swaps (RadixSort.cpp:644-651) make gaussiansSortListSBO name that buffer. */
    if (src_t != tile) {
        memcpy(tile, src_t, (size_t)cap * 4);
        memcpy(depth, src_d, (size_t)cap * 4);
        memcpy(id, src_i, (size_t)cap * 4);
    }
    free(sum_table);
    free(reduce);
    free(pp_tile);
    free(pp_depth);
    free(pp_id);
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 3: FindRanges                                                                        */
/* ------------------------------------------------------------------------------------------ */

void gso_find_ranges(const uint32_t* tile, uint32_t n, uint32_t num_tiles, uint32_t* ranges,
                     int literal) {
    memset(ranges, 0, (size_t)num_tiles * 2 * 4);          /* Subrenderer.cpp:55-60 */
    if (n == 0) return;
    if (literal) {
        /* FindRanges.comp:42-71 with numSortElements = capacity (Subrenderer.cpp:205) */
        for (uint32_t i = 0; i < n; ++i) {
            if (i > 0 && i < n - 1u) {
                uint32_t t0 = tile[i - 1], t1 = tile[i];
                if (t0 != t1) {
                    if (t0 != 0xFFFFFFFFu) ranges[t0 * 2 + 1] = i;
                    if (t1 != 0xFFFFFFFFu) ranges[t1 * 2 + 0] = i;
                }
            } else if (i == 0) {
                uint32_t t0 = tile[0];
                if (t0 != 0xFFFFFFFFu) ranges[t0 * 2 + 0] = 0;
            } else if (i == n - 1u) {
                uint32_t t0 = tile[i];
                if (t0 != 0xFFFFFFFFu) ranges[t0 * 2 + 1] = i;  /* quirk Q1 */
            }
        }
    } else {
        /* product form: n = E valid entries, no sentinel, last end = E */
        ranges[tile[0] * 2 + 0] = 0;
        for (uint32_t i = 1; i < n; ++i)
            if (tile[i - 1] != tile[i]) {
                ranges[tile[i - 1] * 2 + 1] = i;
                ranges[tile[i] * 2 + 0] = i;
            }
        ranges[tile[n - 1] * 2 + 1] = n;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Stage 4: RenderGaussians                                                                   */
/* ------------------------------------------------------------------------------------------ */

typedef float (*exp_fn)(float);

static void render_impl(const gso_params* p, const float* aos, const float* color,
                        const float* cov, const uint32_t* sorted_id, const uint32_t* ranges,
                        uint8_t* rgba_out, exp_fn ex) {
    const uint32_t ts = p->tile_size;
    const uint32_t grid_w = gso_num_tiles_x(p->width, ts);
    const uint32_t grid_h = gso_num_tiles_y(p->height, ts);
    const uint32_t row_end = p->row_end < grid_h ? p->row_end : grid_h;

    /* per-(tile,splat) values of RenderGaussians.comp:88-107 depend only on the splat, so they
     * are computed per tile entry here exactly as the shader does (same expressions). */
    for (uint32_t ty = p->row_begin; ty < row_end; ++ty)
        for (uint32_t tx = 0; tx < grid_w; ++tx) {
            const uint32_t tile_index = ty * grid_w + tx;  /* :74-76 */
            const uint32_t start = ranges[tile_index * 2 + 0];
            const uint32_t end = ranges[tile_index * 2 + 1];
            const uint32_t cnt = end > start ? end - start : 0;
            float* sd = (float*)malloc((size_t)(cnt ? cnt : 1) * 9 * sizeof(float));
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t gi = sorted_id[start + k];   /* :88 */
                const float* rec = aos + (size_t)gi * GSO_FLOATS_PER_GAUSSIAN;
                float world[4] = {rec[0], rec[1], rec[2], 1.0f}, pv[4];
                mat4_mul_vec4(p->view, world, pv);          /* :89 */
                float* o = sd + (size_t)k * 9;
                get_screen_pos(p, pv, &o[0], &o[1]);        /* :90 */
                o[2] = color[(size_t)gi * 4 + 0];           /* :92 */
                o[3] = color[(size_t)gi * 4 + 1];
                o[4] = color[(size_t)gi * 4 + 2];
                o[5] = color[(size_t)gi * 4 + 3];
                const float cx = cov[(size_t)gi * 4 + 0], cy = cov[(size_t)gi * 4 + 1],
                            cz = cov[(size_t)gi * 4 + 2];
                const float det = cx * cz - cy * cy;        /* :96 */
                if (det != 0.0f) {
                    const float det_inv = 1.0f / det;       /* :99 */
                    o[6] = cz * det_inv;                    /* :100 */
                    o[7] = -cy * det_inv;
                    o[8] = cx * det_inv;
                } else {
                    o[6] = o[7] = o[8] = 0.0f;
                    o[5] = 0.0f;                            /* :104 */
                }
            }
            for (uint32_t ly = 0; ly < ts; ++ly)
                for (uint32_t lx = 0; lx < ts; ++lx) {
                    const uint32_t px = tx * ts + lx, py = ty * ts + ly;
                    if (!(px < p->width && py < p->height)) continue; /* :147, no side effects */
                    float col[3] = {0.0f, 0.0f, 0.0f};
                    float Ti = 1.0f;
                    const float fpx = (float)px, fpy = (float)py;     /* R1: integer coords */
                    for (uint32_t k = 0; k < cnt; ++k) {    /* :81,:112 batches are unobservable */
                        const float* o = sd + (size_t)k * 9;
                        float ex_x = o[0] - fpx;            /* :119 */
                        float ex_y = o[1] - fpy;
                        ex_y = -ex_y;                       /* :120 */
                        const float f = -0.5f * (o[6] * ex_x * ex_x + o[8] * ex_y * ex_y) -
                                        o[7] * ex_x * ex_y; /* :123 */
                        const float alpha = o[5] * ex(f);   /* :124 */
                        if (f > 0.0f || alpha < 1.0f / 255.0f) continue; /* :127 */
                        const float wgt = Ti * alpha;       /* :131 */
                        col[0] = col[0] + wgt * o[2];
                        col[1] = col[1] + wgt * o[3];
                        col[2] = col[2] + wgt * o[4];
                        const float next_t = Ti * (1.0f - alpha);        /* :133 */
                        if (next_t < 0.0001f) break;        /* :136-140 add-then-test */
                        Ti = next_t;                        /* :142 */
                    }
                    uint8_t* out = rgba_out + ((size_t)py * p->width + px) * 4;
                    for (int c = 0; c < 3; ++c) {           /* :149-150 + UNORM8 store */
                        float v = clampf(col[c], 0.0f, 1.0f);
                        out[c] = (uint8_t)(v * 255.0f + 0.5f);
                    }
                    out[3] = 255;
                }
            free(sd);
        }
}

void gso_render(const gso_params* p, const float* aos, const float* color, const float* cov,
                const uint32_t* sorted_id, const uint32_t* ranges, uint8_t* rgba_out) {
    render_impl(p, aos, color, cov, sorted_id, ranges, rgba_out, gso_exp);
}

void gso_render_libm_exp(const gso_params* p, const float* aos, const float* color,
                         const float* cov, const uint32_t* sorted_id, const uint32_t* ranges,
                         uint8_t* rgba_out) {
    render_impl(p, aos, color, cov, sorted_id, ranges, rgba_out, expf);
}

/* ------------------------------------------------------------------------------------------ */
/* Whole frame, timed with the reference's buckets (Renderer.cpp:471-475)                     */
/* ------------------------------------------------------------------------------------------ */

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

uint32_t gso_frame(const gso_params* p, const float* aos, uint32_t n, uint8_t* rgba_out,
                   double* timings_ms) {
    const uint32_t grid_w = gso_num_tiles_x(p->width, p->tile_size);
    const uint32_t grid_h = gso_num_tiles_y(p->height, p->tile_size);
    const uint32_t num_tiles = grid_w * grid_h;
    const uint32_t cap = gso_capacity(n, num_tiles);
    float* color = (float*)calloc((size_t)n * 4, 4);
    float* cov = (float*)calloc((size_t)n * 4, 4);
    uint32_t* lt = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* ld = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* li = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* ranges = (uint32_t*)malloc((size_t)num_tiles * 8);

    double t0 = now_ms();
    uint64_t counter = gso_init_sort_list(p, aos, n, cap, color, cov, NULL, lt, ld, li);
    uint32_t e = counter < cap ? (uint32_t)counter : cap;
    double t1 = now_ms();
    gso_sort_stable(lt, ld, li, e);
    double t2 = now_ms();
    gso_find_ranges(lt, e, num_tiles, ranges, 0);
    double t3 = now_ms();
    gso_render(p, aos, color, cov, li, ranges, rgba_out);
    double t4 = now_ms();
    if (timings_ms) {
        timings_ms[0] = t1 - t0;
        timings_ms[1] = t2 - t1;
        timings_ms[2] = t3 - t2;
        timings_ms[3] = t4 - t3;
        timings_ms[4] = t4 - t0;
    }
    free(color); free(cov); free(lt); free(ld); free(li); free(ranges);
    return e;
}

/* ------------------------------------------------------------------------------------------ */
/* Whole frame on several host threads (the all-cores CPU baseline).  The     */
/* reference has no CPU path; this is the same restatement, split the way its GPU stages are  */
/* data-parallel: splats for InitSortList (count, scan, emit -- the deterministic order N8),  */
/* a parallel stable LSD radix (8-bit digits over the 4P key bits) for the sort, tile rows    */
/* for RenderGaussians.  Every per-element value comes from the single-thread functions       */
/* above, so the image and the sorted list equal gso_frame()'s (tests/test_oracle.py).        */
/* ------------------------------------------------------------------------------------------ */

typedef struct mt_ctx {
    const gso_params* p;
    const float* aos;
    uint32_t n, threads, e, bits;
    float *color, *cov;
    gso_splat* splats;
    uint64_t* offsets;                 /* exclusive scan of per-splat element counts */
    uint32_t cap;
    uint32_t *t[2], *d[2], *i[2];      /* ping-pong sort list (SoA) */
    uint32_t* hist;                    /* [threads][256] */
    uint32_t shift, src;
    const uint32_t* ranges;
    uint8_t* rgba;
    volatile uint32_t next_row;        /* dynamic tile-row queue for the render phase */
    pthread_mutex_t lock;
} mt_ctx;

typedef struct mt_arg { mt_ctx* c; uint32_t tid; void (*fn)(mt_ctx*, uint32_t); } mt_arg;

static void* mt_tramp(void* a) {
    mt_arg* m = (mt_arg*)a;
    m->fn(m->c, m->tid);
    return NULL;
}

static void mt_run(mt_ctx* c, void (*fn)(mt_ctx*, uint32_t)) {
    pthread_t th[GSO_MAX_THREADS];
    mt_arg args[GSO_MAX_THREADS];
    for (uint32_t k = 0; k < c->threads; ++k) {
        args[k].c = c; args[k].tid = k; args[k].fn = fn;
        if (k + 1 == c->threads) mt_tramp(&args[k]);            /* caller works too */
        else pthread_create(&th[k], NULL, mt_tramp, &args[k]);
    }
    for (uint32_t k = 0; k + 1 < c->threads; ++k) pthread_join(th[k], NULL);
}

static void mt_chunk(uint32_t total, uint32_t parts, uint32_t k, uint32_t* b, uint32_t* e) {
    const uint64_t per = ((uint64_t)total + parts - 1) / parts;
    uint64_t lo = per * k, hi = lo + per;
    if (lo > total) lo = total;
    if (hi > total) hi = total;
    *b = (uint32_t)lo; *e = (uint32_t)hi;
}

/* phase 1: project + count (gso_init_sort_list without a list = count only) */
static void mt_project(mt_ctx* c, uint32_t tid) {
    uint32_t b, e;
    mt_chunk(c->n, c->threads, tid, &b, &e);
    if (e > b)
        gso_init_sort_list(c->p, c->aos + (size_t)b * GSO_FLOATS_PER_GAUSSIAN, e - b, 0,
                           c->color + (size_t)b * 4, c->cov + (size_t)b * 4, c->splats + b, NULL, NULL, NULL);
}

static void splat_band(const gso_params* p, const gso_splat* s, uint32_t* y0, uint32_t* y1) {
    *y0 = s->min_y > p->row_begin ? s->min_y : p->row_begin;     /* as gso_init_sort_list */
    *y1 = s->max_y < p->row_end ? s->max_y : p->row_end;
    if (*y1 < *y0) *y1 = *y0;
}

/* phase 2: emit at the scanned offsets (InitSortList.comp:130-150, ascending splat index) */
static void mt_emit(mt_ctx* c, uint32_t tid) {
    uint32_t b, e;
    mt_chunk(c->n, c->threads, tid, &b, &e);
    const uint32_t grid_w = gso_num_tiles_x(c->p->width, c->p->tile_size);
    for (uint32_t g = b; g < e; ++g) {
        const gso_splat* s = &c->splats[g];
        if (!s->visible) continue;
        uint32_t y0, y1;
        splat_band(c->p, s, &y0, &y1);
        uint64_t id = c->offsets[g];
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = s->min_x; x < s->max_x; ++x, ++id)
                if (id < c->cap) {                                /* :143 */
                    c->t[0][id] = y * grid_w + x;
                    c->d[0][id] = s->depth_key;
                    c->i[0][id] = g;
                }
    }
}

static inline uint32_t mt_digit(uint32_t tile, uint32_t depth, uint32_t shift) {
    return (shift < 32 ? depth >> shift : tile >> (shift - 32)) & 255u;
}

static void mt_hist(mt_ctx* c, uint32_t tid) {
    uint32_t b, e;
    mt_chunk(c->e, c->threads, tid, &b, &e);
    uint32_t* h = c->hist + (size_t)tid * 256;
    memset(h, 0, 256 * sizeof(uint32_t));
    const uint32_t *t = c->t[c->src], *d = c->d[c->src];
    for (uint32_t k = b; k < e; ++k) ++h[mt_digit(t[k], d[k], c->shift)];
}

static void mt_scatter(mt_ctx* c, uint32_t tid) {                 /* hist now holds start offsets */
    uint32_t b, e;
    mt_chunk(c->e, c->threads, tid, &b, &e);
    uint32_t* h = c->hist + (size_t)tid * 256;
    const uint32_t s = c->src;
    for (uint32_t k = b; k < e; ++k) {
        const uint32_t pos = h[mt_digit(c->t[s][k], c->d[s][k], c->shift)]++;
        c->t[s ^ 1][pos] = c->t[s][k];
        c->d[s ^ 1][pos] = c->d[s][k];
        c->i[s ^ 1][pos] = c->i[s][k];
    }
}

static void mt_render(mt_ctx* c, uint32_t tid) {
    (void)tid;
    const uint32_t grid_h = gso_num_tiles_y(c->p->height, c->p->tile_size);
    const uint32_t row_end = c->p->row_end < grid_h ? c->p->row_end : grid_h;
    for (;;) {
        pthread_mutex_lock(&c->lock);
        const uint32_t row = c->next_row++;
        pthread_mutex_unlock(&c->lock);
        if (row >= row_end) break;
        gso_params q = *c->p;
        q.row_begin = row;
        q.row_end = row + 1;
        render_impl(&q, c->aos, c->color, c->cov, c->i[c->src], c->ranges, c->rgba, gso_exp);
    }
}

static uint32_t clamp_threads(uint32_t threads) {
    if (threads < 1) threads = 1;
    if (threads > GSO_MAX_THREADS) threads = GSO_MAX_THREADS;
    return threads;
}

/* Stage 1 on several threads: same outputs as gso_init_sort_list (tests/test_oracle.py). */
uint64_t gso_init_sort_list_mt(const gso_params* p, const float* aos, uint32_t n, uint32_t capacity,
                               float* color, float* cov, gso_splat* splats, uint32_t* list_tile,
                               uint32_t* list_depth, uint32_t* list_id, uint32_t threads) {
    mt_ctx c;
    memset(&c, 0, sizeof c);
    c.p = p; c.aos = aos; c.n = n; c.threads = clamp_threads(threads);
    c.cap = capacity; c.color = color; c.cov = cov;
    c.splats = splats ? splats : (gso_splat*)calloc((size_t)n + 1, sizeof(gso_splat));
    c.offsets = (uint64_t*)malloc(((size_t)n + 1) * sizeof(uint64_t));
    c.t[0] = list_tile; c.d[0] = list_depth; c.i[0] = list_id;
    /* Subrenderer.cpp:42-46: list <- 0xFFFFFFFF */
    if (list_tile) memset(list_tile, 0xFF, (size_t)capacity * 4);
    if (list_depth) memset(list_depth, 0xFF, (size_t)capacity * 4);
    if (list_id) memset(list_id, 0xFF, (size_t)capacity * 4);
    mt_run(&c, mt_project);
    uint64_t counter = 0;
    for (uint32_t g = 0; g < n; ++g) {                            /* the scan that replaces the atomic */
        c.offsets[g] = counter;
        if (c.splats[g].visible) {
            uint32_t y0, y1;
            splat_band(p, &c.splats[g], &y0, &y1);
            counter += (uint64_t)(c.splats[g].max_x - c.splats[g].min_x) * (y1 - y0);
        }
    }
    if (list_tile) mt_run(&c, mt_emit);
    if (!splats) free(c.splats);
    free(c.offsets);
    return counter;
}

/* Stage 2a on several threads: stable LSD radix, 8 bits per pass over all 64 key bits (passes whose
 * digit is the same for every element are skipped) == gso_sort_stable's order. */
void gso_sort_stable_mt(uint32_t* tile, uint32_t* depth, uint32_t* id, uint32_t e, uint32_t threads) {
    if (e == 0) return;
    mt_ctx c;
    memset(&c, 0, sizeof c);
    c.threads = clamp_threads(threads);
    c.e = e;
    c.t[0] = tile; c.d[0] = depth; c.i[0] = id;
    c.t[1] = (uint32_t*)malloc((size_t)e * 4);
    c.d[1] = (uint32_t*)malloc((size_t)e * 4);
    c.i[1] = (uint32_t*)malloc((size_t)e * 4);
    c.hist = (uint32_t*)malloc((size_t)c.threads * 256 * sizeof(uint32_t));
    for (c.shift = 0; c.shift < 64; c.shift += 8) {
        mt_run(&c, mt_hist);
        uint32_t run = 0, nonzero = 0;
        for (uint32_t dg = 0; dg < 256; ++dg) {
            uint32_t tot = 0;
            for (uint32_t k = 0; k < c.threads; ++k) tot += c.hist[(size_t)k * 256 + dg];
            nonzero += tot != 0;
        }
        if (nonzero <= 1) continue;                               /* nothing to reorder in this pass */
        for (uint32_t dg = 0; dg < 256; ++dg)
            for (uint32_t k = 0; k < c.threads; ++k) {
                const uint32_t v = c.hist[(size_t)k * 256 + dg];
                c.hist[(size_t)k * 256 + dg] = run;
                run += v;
            }
        mt_run(&c, mt_scatter);
        c.src ^= 1;
    }
    if (c.src) {
        memcpy(tile, c.t[1], (size_t)e * 4);
        memcpy(depth, c.d[1], (size_t)e * 4);
        memcpy(id, c.i[1], (size_t)e * 4);
    }
    free(c.t[1]); free(c.d[1]); free(c.i[1]); free(c.hist);
}

/* Stage 4 on several threads (tile rows handed out dynamically): same pixels as gso_render. */
void gso_render_mt(const gso_params* p, const float* aos, const float* color, const float* cov,
                   const uint32_t* sorted_id, const uint32_t* ranges, uint8_t* rgba_out, uint32_t threads) {
    mt_ctx c;
    memset(&c, 0, sizeof c);
    c.p = p; c.aos = aos; c.threads = clamp_threads(threads); c.rgba = rgba_out;
    c.color = (float*)color; c.cov = (float*)cov;
    c.i[0] = (uint32_t*)sorted_id; c.src = 0;
    c.ranges = ranges;
    c.next_row = p->row_begin;
    pthread_mutex_init(&c.lock, NULL);
    mt_run(&c, mt_render);
    pthread_mutex_destroy(&c.lock);
}

uint32_t gso_frame_mt(const gso_params* p, const float* aos, uint32_t n, uint8_t* rgba_out,
                      double* timings_ms, uint32_t threads) {
    threads = clamp_threads(threads);
    const uint32_t grid_w = gso_num_tiles_x(p->width, p->tile_size);
    const uint32_t grid_h = gso_num_tiles_y(p->height, p->tile_size);
    const uint32_t num_tiles = grid_w * grid_h;
    const uint32_t cap = gso_capacity(n, num_tiles);
    float* color = (float*)calloc((size_t)n * 4 + 4, 4);
    float* cov = (float*)calloc((size_t)n * 4 + 4, 4);
    uint32_t* lt = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* ld = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* li = (uint32_t*)malloc((size_t)cap * 4);
    uint32_t* ranges = (uint32_t*)malloc((size_t)num_tiles * 8);

    double t0 = now_ms();
    const uint64_t counter = gso_init_sort_list_mt(p, aos, n, cap, color, cov, NULL, lt, ld, li, threads);
    const uint32_t e = counter < cap ? (uint32_t)counter : cap;   /* IndirectSetup.comp:28 */
    double t1 = now_ms();
    gso_sort_stable_mt(lt, ld, li, e, threads);
    double t2 = now_ms();
    gso_find_ranges(lt, e, num_tiles, ranges, 0);
    double t3 = now_ms();
    gso_render_mt(p, aos, color, cov, li, ranges, rgba_out, threads);
    double t4 = now_ms();
    if (timings_ms) {
        timings_ms[0] = t1 - t0; timings_ms[1] = t2 - t1; timings_ms[2] = t3 - t2;
        timings_ms[3] = t4 - t3; timings_ms[4] = t4 - t0;
    }
    free(color); free(cov); free(lt); free(ld); free(li); free(ranges);
    return e;
}

/* ------------------------------------------------------------------------------------------ */
/* Camera (Engine/Graphics/Camera.cpp:7-48) over glm 0.9.9.8 formulas:                        */
/* lookAtRH  glm/ext/matrix_transform.inl:99-119, perspectiveRH_ZO matrix_clip_space.inl:233-246, */
/* normalize = v * (1/sqrt(dot)) glm/detail/func_geometric.inl:88, func_exponential.inl.      */
/* ------------------------------------------------------------------------------------------ */

static void v3_normalize(float v[3]) {
    float d = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float inv = 1.0f / sqrtf(d);
    v[0] = v[0] * inv; v[1] = v[1] * inv; v[2] = v[2] * inv;
}
static void v3_cross(const float a[3], const float b[3], float o[3]) {
    o[0] = a[1] * b[2] - b[1] * a[2];
    o[1] = a[2] * b[0] - b[2] * a[0];
    o[2] = a[0] * b[1] - b[0] * a[1];
}
static float v3_dot(const float a[3], const float b[3]) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

void gso_camera_matrices(const float pos[3], float yaw, float pitch, float aspect,
                         float near_plane, float far_plane, float* view, float* proj) {
    /* Camera.cpp:10-16 (double sin/cos, cast to float, then normalize) */
    float fwd[3] = {(float)(sin((double)yaw) * cos((double)pitch)), (float)sin((double)pitch),
                    (float)(cos((double)yaw) * cos((double)pitch))};
    v3_normalize(fwd);
    /* Camera.cpp:34-38: lookAt(position, position + forwardDir, (0,1,0)) */
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float center[3] = {pos[0] + fwd[0], pos[1] + fwd[1], pos[2] + fwd[2]};
    float f[3] = {center[0] - pos[0], center[1] - pos[1], center[2] - pos[2]};
    v3_normalize(f);
    float s[3];
    v3_cross(f, up, s);
    v3_normalize(s);
    float u[3];
    v3_cross(s, f, u);
    memset(view, 0, 16 * sizeof(float));
    view[0 * 4 + 0] = s[0]; view[1 * 4 + 0] = s[1]; view[2 * 4 + 0] = s[2];
    view[0 * 4 + 1] = u[0]; view[1 * 4 + 1] = u[1]; view[2 * 4 + 1] = u[2];
    view[0 * 4 + 2] = -f[0]; view[1 * 4 + 2] = -f[1]; view[2 * 4 + 2] = -f[2];
    view[3 * 4 + 0] = -v3_dot(s, pos);
    view[3 * 4 + 1] = -v3_dot(u, pos);
    view[3 * 4 + 2] = v3_dot(f, pos);
    view[3 * 4 + 3] = 1.0f;
    /* Camera.cpp:41-46: perspective(radians(90), aspect, near, far), GLM_FORCE_DEPTH_ZERO_TO_ONE */
    const float fovy = 90.0f * 0.01745329251994329576923690768489f; /* glm::radians */
    const float tan_half = tanf(fovy / 2.0f);
    memset(proj, 0, 16 * sizeof(float));
    proj[0 * 4 + 0] = 1.0f / (aspect * tan_half);
    proj[1 * 4 + 1] = 1.0f / tan_half;
    proj[2 * 4 + 2] = far_plane / (near_plane - far_plane);
    proj[2 * 4 + 3] = -1.0f;
    proj[3 * 4 + 2] = -(far_plane * near_plane) / (far_plane - near_plane);
}

/* Engine/SMath.h:10-34 */
static uint32_t morton_part_by2(uint32_t x) {
    x &= 0x000003ffu;
    x = (x ^ (x << 16)) & 0xff0000ffu;
    x = (x ^ (x << 8)) & 0x0300f00fu;
    x = (x ^ (x << 4)) & 0x030c30c3u;
    x = (x ^ (x << 2)) & 0x09249249u;
    return x;
}
uint32_t gso_morton(uint32_t x, uint32_t y, uint32_t z) {
    return (morton_part_by2(z) << 2) + (morton_part_by2(y) << 1) + morton_part_by2(x);
}
