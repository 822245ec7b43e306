/*
 * pt_oracle.c -- scalar CPU restatement of the reference's device programs.
 * TEST INFRASTRUCTURE ONLY (see pt_oracle.h). Build: oracle/Makefile
 * (-O2 -ffp-contract=off -mfma, no fast-math).
 */
#include "pt_oracle.h"
#include "pt_math.h"

#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_math_mode = PT_MATH_PORTABLE;

void pt_set_math_mode(int mode)
{
    g_math_mode = mode;
}

static inline void m_sincos(float theta, float *s, float *c)
{
    if (g_math_mode == PT_MATH_LIBM)
    {
        *s = sinf(theta);
        *c = cosf(theta);
    }
    else
    {
        pt_sincos_portable(theta, s, c);
    }
}
static inline float m_log(float x)
{
    return g_math_mode == PT_MATH_LIBM ? logf(x) : pt_log_portable(x);
}
static inline float m_acos(float x)
{
    if (g_math_mode == PT_MATH_LIBM)
    {
        return acosf(pt_min(pt_max(x, -1.0f), 1.0f));
    }
    return pt_acos_portable(x);
}
static inline float m_atan2(float y, float x)
{
    return g_math_mode == PT_MATH_LIBM ? atan2f(y, x) : pt_atan2_portable(y, x);
}

void pt_math_eval(int fn, int mode, const float *a, const float *b, float *out,
                  float *out2, size_t n)
{
    const int saved = g_math_mode;
    g_math_mode = mode;
    for (size_t i = 0; i < n; ++i)
    {
        switch (fn)
        {
        case 0: m_sincos(a[i], &out[i], &out2[i]); break;
        case 1: out[i] = m_log(a[i]); break;
        case 2: out[i] = m_acos(a[i]); break;
        default: out[i] = m_atan2(a[i], b[i]); break;
        }
    }
    g_math_mode = saved;
}

/* ------------------------------------------------------------------------ */
/* RNG (shader_common.glsl:15-23, shader.rgen:34-42, :174-175)              */
/* ------------------------------------------------------------------------ */

uint32_t pt_hash(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}

float pt_random(uint32_t *rng_state)
{
    uint32_t s = *rng_state;
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    *rng_state = s;
    /* uint -> float is round-to-nearest-even, so this can return 1.0 */
    return (float)s * (1.0f / 4294967296.0f);
}

uint32_t pt_seed(uint32_t pixel_index, uint32_t global_frame_count)
{
    return pt_hash(pixel_index) + pt_hash(global_frame_count + 1u);
}

/* ------------------------------------------------------------------------ */
/* Camera ray (shader.rgen:45-58, :181-191)                                  */
/* ------------------------------------------------------------------------ */

static void sample_gaussian(uint32_t *rng, float *gx, float *gy)
{
    const float float_min = 1.175494351e-38f;
    const float r = sqrtf(-2.0f * m_log(pt_max(float_min, pt_random(rng))));
    const float theta = PT_TWO_PI * pt_random(rng);
    float s, c;
    m_sincos(theta, &s, &c);
    *gx = r * c;
    *gy = r * s;
}

static void sample_disk(uint32_t *rng, float *dx, float *dy)
{
    const float r = sqrtf(pt_random(rng));
    const float theta = PT_TWO_PI * pt_random(rng);
    float s, c;
    m_sincos(theta, &s, &c);
    *dx = r * c;
    *dy = r * s;
}

static inline pt_vec3 ld3(const float *p)
{
    return pt_v3(p[0], p[1], p[2]);
}
static inline void st3(float *p, pt_vec3 v)
{
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

static void camera_ray(const pt_push_constants *push, uint32_t width,
                       uint32_t height, uint32_t px, uint32_t py,
                       uint32_t *rng, pt_vec3 *origin, pt_vec3 *dir)
{
    float gx, gy;
    sample_gaussian(rng, &gx, &gy);
    const float offx = 0.5f + 0.375f * gx;
    const float offy = 0.5f + 0.375f * gy;
    const float uvx = 2.0f * ((float)px + offx) / (float)width - 1.0f;
    const float uvy = 2.0f * ((float)py + offy) / (float)height - 1.0f;

    float dx, dy;
    sample_disk(rng, &dx, &dy);
    const float defx = push->aperture_radius * dx;
    const float defy = push->aperture_radius * dy;
    const pt_vec3 cdx = ld3(push->camera_dir_x);
    const pt_vec3 cdy = ld3(push->camera_dir_y);
    const pt_vec3 cdz = ld3(push->camera_dir_z);
    const pt_vec3 defocus_offset =
        pt_add(pt_scale(cdx, defx), pt_scale(cdy, defy));
    *origin = pt_add(ld3(push->camera_position), defocus_offset);
    pt_vec3 t = pt_scale(cdz, push->sensor_distance);
    t = pt_add(t, pt_scale(pt_scale(cdx, push->sensor_half_width), uvx));
    t = pt_add(t, pt_scale(pt_scale(cdy, push->sensor_half_height), uvy));
    t = pt_scale(t, push->focus_distance); /* focus * (...) is commutative */
    *dir = pt_normalize(pt_sub(t, defocus_offset));
}

void pt_camera_ray(const pt_push_constants *push, uint32_t width,
                   uint32_t height, uint32_t px, uint32_t py,
                   uint32_t *rng_state, float origin[3], float dir[3])
{
    pt_vec3 o, d;
    camera_ray(push, width, height, px, py, rng_state, &o, &d);
    st3(origin, o);
    st3(dir, d);
}

/* ------------------------------------------------------------------------ */
/* Closest-hit helpers (closest_hit_common.glsl)                             */
/* ------------------------------------------------------------------------ */

static inline float offset_component(float p, float n)
{
    const int32_t of_i = (int32_t)(256.0f * n);
    const uint32_t bits = pt_f2u(p);
    const uint32_t moved =
        (p < 0.0f) ? bits - (uint32_t)of_i : bits + (uint32_t)of_i;
    const float p_i = pt_u2f(moved);
    const float origin = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    return fabsf(p) < origin ? p + float_scale * n : p_i;
}

/* closest_hit_common.glsl:25-44 */
static pt_vec3 offset_position_along_normal(pt_vec3 p, pt_vec3 n)
{
    return pt_v3(offset_component(p.x, n.x), offset_component(p.y, n.y),
                 offset_component(p.z, n.z));
}

void pt_offset_position_along_normal(const float p[3], const float n[3],
                                     float out[3])
{
    st3(out, offset_position_along_normal(ld3(p), ld3(n)));
}

/* closest_hit_common.glsl:46-52 */
static pt_vec3 sample_sphere(uint32_t *rng)
{
    const float theta = PT_TWO_PI * pt_random(rng);
    const float z = 2.0f * pt_random(rng) - 1.0f;
    const float r = sqrtf(1.0f - z * z);
    float s, c;
    m_sincos(theta, &s, &c);
    return pt_v3(r * c, r * s, z);
}

/* closest_hit_common.glsl:54-57 */
static pt_vec3 reflect_diffuse(pt_vec3 normal, uint32_t *rng)
{
    return pt_normalize(pt_add(normal, sample_sphere(rng)));
}

typedef struct
{
    pt_vec3 color;
    pt_vec3 emissivity;
    pt_vec3 origin;
    pt_vec3 dir;
    float attenuation;
    uint32_t rng;
    int hit_sky;
} payload_t;

/* The four closest-hit programs after get_hit(): diffuse.rchit:5-14,
 * specular.rchit:5-14, emissive.rchit:5-14, dielectric.rchit:5-52.
 * dir is gl_WorldRayDirectionEXT, pos/nrm the Hit. */
static void material_scatter(uint32_t kind, pt_vec3 dir, pt_vec3 pos,
                             pt_vec3 nrm, payload_t *P)
{
    P->hit_sky = 0;
    switch (kind)
    {
    case PT_KIND_DIFFUSE:
        P->origin = offset_position_along_normal(pos, nrm);
        P->dir = reflect_diffuse(nrm, &P->rng);
        P->color = pt_v3((nrm.x + 1.0f) * 0.5f, (nrm.y + 1.0f) * 0.5f,
                         (nrm.z + 1.0f) * 0.5f);
        P->emissivity = pt_v3(0.0f, 0.0f, 0.0f);
        break;
    case PT_KIND_SPECULAR:
        P->origin = offset_position_along_normal(pos, nrm);
        P->dir = pt_reflect(dir, nrm);
        P->color = pt_v3(0.75f, 0.75f, 0.75f);
        P->emissivity = pt_v3(0.0f, 0.0f, 0.0f);
        break;
    case PT_KIND_EMISSIVE:
        P->origin = offset_position_along_normal(pos, nrm);
        P->dir = reflect_diffuse(nrm, &P->rng);
        P->color = pt_v3(0.75f, 0.75f, 0.75f);
        P->emissivity = pt_v3(5.0f, 5.0f, 5.0f);
        break;
    default: /* PT_KIND_DIELECTRIC */
    {
        P->color = pt_v3(1.0f, 1.0f, 1.0f);
        P->emissivity = pt_v3(0.0f, 0.0f, 0.0f);
        const int into = pt_dot(dir, nrm) < 0.0f;
        const pt_vec3 normal = into ? nrm : pt_neg(nrm);
        const pt_vec3 reflected_dir = pt_reflect(dir, normal);
        const float n_air = 1.0f;
        const float n_glass = 1.5f;
        const float n_ratio = into ? n_air / n_glass : n_glass / n_air;
        const float ddn = pt_dot(dir, normal);
        const float cos2t = 1.0f - n_ratio * n_ratio * (1.0f - ddn * ddn);
        if (cos2t < 0.0f)
        {
            P->origin = offset_position_along_normal(pos, normal);
            P->dir = reflected_dir;
            return;
        }
        const float k = (into ? 1.0f : -1.0f) * (ddn * n_ratio + sqrtf(cos2t));
        const pt_vec3 transmitted_dir =
            pt_normalize(pt_sub(pt_scale(dir, n_ratio), pt_scale(nrm, k)));
        const float a = n_glass - n_air;
        const float b = n_glass + n_air;
        const float R0 = a * a / (b * b);
        const float c = 1.0f - (into ? -ddn : pt_dot(transmitted_dir, nrm));
        const float Re = R0 + (1.0f - R0) * c * c * c * c * c;
        const float Tr = 1.0f - Re;
        const float Pr = 0.25f + 0.5f * Re;
        const float RP = Re / Pr;
        const float TP = Tr / (1.0f - Pr);
        if (pt_random(&P->rng) < Pr)
        {
            P->attenuation = RP;
            P->origin = offset_position_along_normal(pos, normal);
            P->dir = reflected_dir;
        }
        else
        {
            P->attenuation = TP;
            P->origin = offset_position_along_normal(pos, pt_neg(normal));
            P->dir = transmitted_dir;
        }
        break;
    }
    }
}

void pt_material_scatter(uint32_t kind, const float dir[3], const float pos[3],
                         const float nrm[3], uint32_t *rng_state,
                         float out_origin[3], float out_dir[3],
                         float out_color[3], float out_emissivity[3],
                         float *out_attenuation)
{
    payload_t P;
    memset(&P, 0, sizeof P);
    P.attenuation = 1.0f;
    P.rng = *rng_state;
    material_scatter(kind, ld3(dir), ld3(pos), ld3(nrm), &P);
    *rng_state = P.rng;
    st3(out_origin, P.origin);
    st3(out_dir, P.dir);
    st3(out_color, P.color);
    st3(out_emissivity, P.emissivity);
    *out_attenuation = P.attenuation;
}

/* ------------------------------------------------------------------------ */
/* Tone map + UNORM8 store (shader.rgen:141-162, :218-223;                   */
/* render target is R8G8B8A8_UNORM, src/renderer.cpp:1759-1760)              */
/* ------------------------------------------------------------------------ */

static pt_vec3 tonemap_pbr_neutral(pt_vec3 color)
{
    const float start_compression = 0.8f - 0.04f;
    const float desaturation = 0.15f;

    const float x = pt_min(color.x, pt_min(color.y, color.z));
    const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    color = pt_v3(color.x - offset, color.y - offset, color.z - offset);

    const float peak = pt_max(color.x, pt_max(color.y, color.z));
    if (peak < start_compression)
    {
        return color;
    }
    const float d = 1.0f - start_compression;
    const float new_peak = 1.0f - d * d / (peak + d - start_compression);
    color = pt_scale(color, new_peak / peak);
    const float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    /* mix(x, y, a) = x * (1 - a) + y * a */
    const float one_minus_g = 1.0f - g;
    return pt_v3(color.x * one_minus_g + new_peak * g,
                 color.y * one_minus_g + new_peak * g,
                 color.z * one_minus_g + new_peak * g);
}

void pt_tonemap_pbr_neutral(const float rgb[3], float out[3])
{
    st3(out, tonemap_pbr_neutral(ld3(rgb)));
}

/* shader.rgen:107-135, the alternative the reference keeps under "#else" at
 * :220-221. GLSL mat3 constructors take columns, so "input_mat * color" has
 * x = 0.59719 r + 0.35458 g + 0.04823 b, and so on. The multiply-adds are
 * evaluated as fma chains (GLSL leaves contraction to the compiler). */
static float aces_fit(float v)
{
    const float a = v * (v + 0.0245786f) - 0.000090537f;
    const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}

static float clamp01(float x)
{
    return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
}

static pt_vec3 tonemap_aces(pt_vec3 c)
{
    const float x = fmaf(0.04823f, c.z, fmaf(0.35458f, c.y, 0.59719f * c.x));
    const float y = fmaf(0.01566f, c.z, fmaf(0.90834f, c.y, 0.07600f * c.x));
    const float z = fmaf(0.83777f, c.z, fmaf(0.13383f, c.y, 0.02840f * c.x));
    const float fx = aces_fit(x), fy = aces_fit(y), fz = aces_fit(z);
    const float r = fmaf(-0.07367f, fz, fmaf(-0.53108f, fy, 1.60475f * fx));
    const float g = fmaf(-0.00605f, fz, fmaf(1.10813f, fy, -0.10208f * fx));
    const float b = fmaf(1.07602f, fz, fmaf(-0.07276f, fy, -0.00327f * fx));
    return pt_v3(clamp01(r), clamp01(g), clamp01(b));
}

void pt_tonemap_aces(const float rgb[3], float out[3])
{
    st3(out, tonemap_aces(ld3(rgb)));
}

/* float -> UNORM8: clamp to [0,1] (NaN -> 0), scale by 255, round to nearest
 * even */
uint8_t pt_unorm8(float x)
{
    if (!(x > 0.0f))
    {
        return 0;
    }
    if (x > 1.0f)
    {
        x = 1.0f;
    }
    return (uint8_t)rintf(x * 255.0f);
}

/* ------------------------------------------------------------------------ */
/* Scene                                                                     */
/* ------------------------------------------------------------------------ */

typedef struct
{
    float *positions;
    float *normals;
    uint32_t *indices;
    uint32_t n_vertices, n_triangles;
} mesh_t;

typedef struct
{
    float o2w[12];
    float w2o[12];
    uint32_t mesh, kind;
    uint32_t first_flat;
} inst_t;

typedef struct
{
    pt_vec3 v0, e1, e2;
} wtri_t;

typedef struct
{
    float lo[3], hi[3];
    uint32_t left, right; /* children (internal nodes) */
    uint32_t first;       /* first leaf slot in prim_order (leaves) */
    uint32_t count;       /* 0 for internal, else number of prims */
} bnode_t;

struct pt_scene
{
    mesh_t *meshes;
    uint32_t n_meshes;
    inst_t *insts;
    uint32_t n_insts;
    wtri_t *tris; /* flattened, world space */
    uint32_t *tri_inst;
    uint32_t n_tris;
    float *env;
    uint32_t env_w, env_h;
    /* oracle BVH */
    bnode_t *nodes;
    uint32_t n_nodes;
    uint32_t *prim_order; /* leaf slots -> flat triangle index */
    double sah_cost;
    /* two-level mode: closest hits are defined in object space, instance by
     * instance (see intersect_two_level) */
    int two_level;
};

/* Inverse of a 3x4 affine matrix: cofactor formula evaluated in double,
 * rounded once to float. The driver's own gl_WorldToObjectEXT computation is
 * not specified; this is the oracle's definition. */
void pt_inverse_affine(const float m[12], float out[12])
{
    const double a = m[0], b = m[1], c = m[2], tx = m[3];
    const double d = m[4], e = m[5], f = m[6], ty = m[7];
    const double g = m[8], h = m[9], i = m[10], tz = m[11];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) +
                       c * (d * h - e * g);
    const double r = 1.0 / det;
    const double i00 = (e * i - f * h) * r, i01 = (c * h - b * i) * r,
                 i02 = (b * f - c * e) * r;
    const double i10 = (f * g - d * i) * r, i11 = (a * i - c * g) * r,
                 i12 = (c * d - a * f) * r;
    const double i20 = (d * h - e * g) * r, i21 = (b * g - a * h) * r,
                 i22 = (a * e - b * d) * r;
    out[0] = (float)i00; out[1] = (float)i01; out[2] = (float)i02;
    out[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out[4] = (float)i10; out[5] = (float)i11; out[6] = (float)i12;
    out[7] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out[8] = (float)i20; out[9] = (float)i21; out[10] = (float)i22;
    out[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
}

static void build_bvh(pt_scene *S);

pt_scene *pt_scene_create(const pt_mesh_desc *meshes, uint32_t n_meshes,
                          const pt_instance_desc *instances,
                          uint32_t n_instances, const float *env_rgba,
                          uint32_t env_width, uint32_t env_height)
{
    pt_scene *S = (pt_scene *)calloc(1, sizeof *S);
    S->n_meshes = n_meshes;
    S->meshes = (mesh_t *)calloc(n_meshes ? n_meshes : 1, sizeof(mesh_t));
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        mesh_t *M = &S->meshes[m];
        M->n_vertices = meshes[m].n_vertices;
        M->n_triangles = meshes[m].n_triangles;
        const size_t vb = (size_t)M->n_vertices * 12, ib =
                                                          (size_t)M->n_triangles * 12;
        M->positions = (float *)malloc(vb ? vb : 4);
        M->normals = (float *)malloc(vb ? vb : 4);
        M->indices = (uint32_t *)malloc(ib ? ib : 4);
        memcpy(M->positions, meshes[m].positions, vb);
        memcpy(M->normals, meshes[m].normals, vb);
        memcpy(M->indices, meshes[m].indices, ib);
    }
    S->n_insts = n_instances;
    S->insts = (inst_t *)calloc(n_instances ? n_instances : 1, sizeof(inst_t));
    uint32_t total = 0;
    for (uint32_t i = 0; i < n_instances; ++i)
    {
        inst_t *I = &S->insts[i];
        memcpy(I->o2w, instances[i].transform, sizeof I->o2w);
        pt_inverse_affine(I->o2w, I->w2o);
        I->mesh = instances[i].mesh;
        I->kind = instances[i].kind;
        I->first_flat = total;
        total += S->meshes[I->mesh].n_triangles;
    }
    S->n_tris = total;
    S->tris = (wtri_t *)malloc((total ? total : 1) * sizeof(wtri_t));
    S->tri_inst = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n_instances; ++i)
    {
        const inst_t *I = &S->insts[i];
        const mesh_t *M = &S->meshes[I->mesh];
        for (uint32_t p = 0; p < M->n_triangles; ++p)
        {
            const uint32_t i0 = M->indices[3 * p], i1 = M->indices[3 * p + 1],
                           i2 = M->indices[3 * p + 2];
            const pt_vec3 a = pt_xform_point(I->o2w, ld3(M->positions + 3 * i0));
            const pt_vec3 b = pt_xform_point(I->o2w, ld3(M->positions + 3 * i1));
            const pt_vec3 c = pt_xform_point(I->o2w, ld3(M->positions + 3 * i2));
            wtri_t *T = &S->tris[I->first_flat + p];
            T->v0 = a;
            T->e1 = pt_sub(b, a);
            T->e2 = pt_sub(c, a);
            S->tri_inst[I->first_flat + p] = i;
        }
    }
    if (env_rgba && env_width && env_height)
    {
        const size_t eb = (size_t)env_width * env_height * 16;
        S->env = (float *)malloc(eb);
        memcpy(S->env, env_rgba, eb);
        S->env_w = env_width;
        S->env_h = env_height;
    }
    build_bvh(S);
    return S;
}

void pt_scene_destroy(pt_scene *S)
{
    if (!S)
    {
        return;
    }
    for (uint32_t m = 0; m < S->n_meshes; ++m)
    {
        free(S->meshes[m].positions);
        free(S->meshes[m].normals);
        free(S->meshes[m].indices);
    }
    free(S->meshes);
    free(S->insts);
    free(S->tris);
    free(S->tri_inst);
    free(S->env);
    free(S->nodes);
    free(S->prim_order);
    free(S);
}

uint32_t pt_scene_triangle_count(const pt_scene *S)
{
    return S->n_tris;
}

void pt_scene_triangle_ids(const pt_scene *S, uint32_t flat,
                           uint32_t *instance, uint32_t *primitive)
{
    const uint32_t i = S->tri_inst[flat];
    *instance = i;
    *primitive = flat - S->insts[i].first_flat;
}

void pt_scene_world_triangles(const pt_scene *S, float *out)
{
    memcpy(out, S->tris, (size_t)S->n_tris * sizeof(wtri_t));
}

double pt_scene_bvh_sah_cost(const pt_scene *S)
{
    return S->sah_cost;
}

/* ------------------------------------------------------------------------ */
/* Miss program (shader.rmiss:11-29)                                         */
/* ------------------------------------------------------------------------ */

static inline int wrap_index(int i, int n)
{
    int r = i % n;
    return r < 0 ? r + n : r;
}

/* Bilinear, repeat addressing, no mips (src/renderer.cpp:861-878); texel
 * centres at (i + 0.5) / size; float weights. */
static pt_vec3 env_lookup(const pt_scene *S, float u, float v)
{
    if (!(u == u))
    {
        u = 0.0f;
    }
    if (!(v == v))
    {
        v = 0.0f;
    }
    const float x = u * (float)S->env_w - 0.5f;
    const float y = v * (float)S->env_h - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const int x0 = wrap_index((int)x0f, (int)S->env_w);
    const int x1 = wrap_index((int)x0f + 1, (int)S->env_w);
    const int y0 = wrap_index((int)y0f, (int)S->env_h);
    const int y1 = wrap_index((int)y0f + 1, (int)S->env_h);
    const float *t00 = S->env + 4 * ((size_t)y0 * S->env_w + x0);
    const float *t10 = S->env + 4 * ((size_t)y0 * S->env_w + x1);
    const float *t01 = S->env + 4 * ((size_t)y1 * S->env_w + x0);
    const float *t11 = S->env + 4 * ((size_t)y1 * S->env_w + x1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    float c[3];
    for (int k = 0; k < 3; ++k)
    {
        const float top = t00[k] * gx + t10[k] * fx;
        const float bot = t01[k] * gx + t11[k] * fx;
        c[k] = top * gy + bot * fy;
    }
    return pt_v3(c[0], c[1], c[2]);
}

static pt_vec3 miss_color(const pt_scene *S, pt_vec3 d)
{
    if (!S->env)
    {
        /* the "#if 0" branch, shader.rmiss:13-18 */
        return d.y >= 0.0f ? pt_v3(0.75f, 0.75f, 0.75f)
                           : pt_v3(0.1f, 0.1f, 0.1f);
    }
    const float azimuth = m_atan2(d.z, d.x);
    const float inclination = m_acos(d.y);
    const float u = (azimuth + PT_PI) / PT_TWO_PI;
    const float v = inclination / PT_PI;
    return env_lookup(S, u, v);
}

void pt_miss(const pt_scene *S, const float dir[3], float out_color[3])
{
    st3(out_color, miss_color(S, ld3(dir)));
}

/* ------------------------------------------------------------------------ */
/* Intersection                                                              */
/* ------------------------------------------------------------------------ */

/* Moller-Trumbore on a world-space triangle stored as (v0, e1, e2). No
 * culling. Accept iff det != 0, u >= 0, v >= 0, u + v <= 1, 0 < t < tmax.
 * This exact operation order is part of the hit definition. */
static inline int mt_test(pt_vec3 o, pt_vec3 d, const wtri_t *T, float tmax,
                          float *t, float *u, float *v)
{
    const pt_vec3 p = pt_cross(d, T->e2);
    const float det = pt_dot(T->e1, p);
    if (det == 0.0f)
    {
        return 0;
    }
    const float inv = 1.0f / det;
    const pt_vec3 tv = pt_sub(o, T->v0);
    const float uu = pt_dot(tv, p) * inv;
    const pt_vec3 q = pt_cross(tv, T->e1);
    const float vv = pt_dot(d, q) * inv;
    const float tt = pt_dot(T->e2, q) * inv;
    if (uu >= 0.0f && vv >= 0.0f && (uu + vv) <= 1.0f && tt > 0.0f &&
        tt < tmax)
    {
        *t = tt;
        *u = uu;
        *v = vv;
        return 1;
    }
    return 0;
}

static inline void consider(pt_hit *best, float t, float u, float v,
                            uint32_t id)
{
    if (t < best->t || (t == best->t && id < best->id))
    {
        best->t = t;
        best->u = u;
        best->v = v;
        best->id = id;
    }
}

static pt_hit intersect_brute(const pt_scene *S, pt_vec3 o, pt_vec3 d,
                              float tmax, pt_stats *st)
{
    pt_hit best = {tmax, 0.0f, 0.0f, PT_NO_HIT};
    for (uint32_t g = 0; g < S->n_tris; ++g)
    {
        float t, u, v;
        if (mt_test(o, d, &S->tris[g], tmax, &t, &u, &v))
        {
            consider(&best, t, u, v, g);
        }
    }
    if (st)
    {
        st->tris_tested += S->n_tris;
    }
    return best;
}

/* Two-level closest hit, the shape of the reference's TLAS of instances over
 * one BLAS per mesh (src/renderer.cpp:1044-1281): for every instance the ray
 * is taken to object space with the instance's world-to-object matrix
 * (origin as a point, direction as a vector, not re-normalised, so t keeps its
 * meaning), and Moller-Trumbore runs on the mesh's own vertices (v0, e1 = v1 -
 * v0, e2 = v2 - v0 in object space). The hit is the lexicographic minimum over
 * (t, flattened triangle index) as in the flattened definition; the two
 * definitions agree up to the rounding of the transforms. Brute force only. */
static pt_hit intersect_two_level(const pt_scene *S, pt_vec3 o, pt_vec3 d,
                                  float tmax, pt_stats *st)
{
    pt_hit best = {tmax, 0.0f, 0.0f, PT_NO_HIT};
    for (uint32_t i = 0; i < S->n_insts; ++i)
    {
        const inst_t *I = &S->insts[i];
        const mesh_t *M = &S->meshes[I->mesh];
        const pt_vec3 ro = pt_xform_point(I->w2o, o);
        const pt_vec3 rd = pt_xform_vector(I->w2o, d);
        for (uint32_t f = 0; f < M->n_triangles; ++f)
        {
            const pt_vec3 a = ld3(M->positions + 3 * M->indices[3 * f + 0]);
            const pt_vec3 b = ld3(M->positions + 3 * M->indices[3 * f + 1]);
            const pt_vec3 c = ld3(M->positions + 3 * M->indices[3 * f + 2]);
            wtri_t T;
            T.v0 = a;
            T.e1 = pt_sub(b, a);
            T.e2 = pt_sub(c, a);
            float t, u, v;
            if (mt_test(ro, rd, &T, tmax, &t, &u, &v))
            {
                consider(&best, t, u, v, I->first_flat + f);
            }
        }
        if (st)
        {
            st->tris_tested += M->n_triangles;
        }
    }
    return best;
}

void pt_scene_set_two_level(pt_scene *S, int enabled)
{
    S->two_level = enabled != 0;
}

/* --- oracle BVH: binned SAH, binary, float boxes ---------------------------*/

#define BVH_BINS 32
#define BVH_LEAF_MAX 4

typedef struct
{
    float lo[3], hi[3];
} box_t;

static inline void box_empty(box_t *b)
{
    for (int k = 0; k < 3; ++k)
    {
        b->lo[k] = INFINITY;
        b->hi[k] = -INFINITY;
    }
}
static inline void box_grow_pt(box_t *b, const float *p)
{
    for (int k = 0; k < 3; ++k)
    {
        if (p[k] < b->lo[k]) b->lo[k] = p[k];
        if (p[k] > b->hi[k]) b->hi[k] = p[k];
    }
}
static inline void box_grow(box_t *b, const box_t *o)
{
    for (int k = 0; k < 3; ++k)
    {
        if (o->lo[k] < b->lo[k]) b->lo[k] = o->lo[k];
        if (o->hi[k] > b->hi[k]) b->hi[k] = o->hi[k];
    }
}
static inline double box_area(const box_t *b)
{
    const double dx = (double)b->hi[0] - b->lo[0];
    const double dy = (double)b->hi[1] - b->lo[1];
    const double dz = (double)b->hi[2] - b->lo[2];
    if (dx < 0 || dy < 0 || dz < 0)
    {
        return 0.0;
    }
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}

typedef struct
{
    pt_scene *S;
    box_t *pbox;
    float *pcen; /* 3 per prim */
    uint32_t cap;
    float pad; /* absolute box padding covering Moller-Trumbore rounding slop */
    uint32_t leaf_max; /* most triangles a leaf may hold */
} build_ctx;

static uint32_t build_rec(build_ctx *B, uint32_t first, uint32_t count)
{
    pt_scene *S = B->S;
    const uint32_t me = S->n_nodes++;
    box_t nb, cb;
    box_empty(&nb);
    box_empty(&cb);
    for (uint32_t i = first; i < first + count; ++i)
    {
        const uint32_t p = S->prim_order[i];
        box_grow(&nb, &B->pbox[p]);
        box_grow_pt(&cb, &B->pcen[3 * p]);
    }
    memcpy(S->nodes[me].lo, nb.lo, 12);
    memcpy(S->nodes[me].hi, nb.hi, 12);
    const double pa = box_area(&nb);
    int best_axis = -1, best_bin = 0;
    /* leaf cost (Ct = 1, Cn = 1); nodes too big for a leaf must split */
    double best_cost = count > B->leaf_max ? INFINITY : (double)count;
    if (count > 1)
    {
        for (int ax = 0; ax < 3; ++ax)
        {
            const float c0 = cb.lo[ax], c1 = cb.hi[ax];
            if (!(c1 > c0))
            {
                continue;
            }
            box_t bb[BVH_BINS];
            uint32_t bc[BVH_BINS];
            for (int b = 0; b < BVH_BINS; ++b)
            {
                box_empty(&bb[b]);
                bc[b] = 0;
            }
            const float sc = (float)BVH_BINS / (c1 - c0);
            for (uint32_t i = first; i < first + count; ++i)
            {
                const uint32_t p = S->prim_order[i];
                int b = (int)((B->pcen[3 * p + ax] - c0) * sc);
                if (b >= BVH_BINS) b = BVH_BINS - 1;
                if (b < 0) b = 0;
                box_grow(&bb[b], &B->pbox[p]);
                bc[b]++;
            }
            double ra[BVH_BINS];
            uint32_t rc[BVH_BINS];
            box_t acc;
            box_empty(&acc);
            uint32_t n = 0;
            for (int b = BVH_BINS - 1; b > 0; --b)
            {
                box_grow(&acc, &bb[b]);
                n += bc[b];
                ra[b] = box_area(&acc);
                rc[b] = n;
            }
            box_empty(&acc);
            n = 0;
            for (int b = 0; b < BVH_BINS - 1; ++b)
            {
                box_grow(&acc, &bb[b]);
                n += bc[b];
                if (n == 0 || rc[b + 1] == 0)
                {
                    continue;
                }
                const double cost =
                    1.0 + (box_area(&acc) * n + ra[b + 1] * rc[b + 1]) /
                              (pa > 0 ? pa : 1.0);
                if (cost < best_cost)
                {
                    best_cost = cost;
                    best_axis = ax;
                    best_bin = b;
                }
            }
        }
    }
    if (best_axis < 0 && count > B->leaf_max)
    {
        /* all centroids coincide: split in the middle */
        const uint32_t half = count / 2;
        const uint32_t l = build_rec(B, first, half);
        const uint32_t r = build_rec(B, first + half, count - half);
        S->nodes[me].left = l;
        S->nodes[me].right = r;
        S->nodes[me].count = 0;
        return me;
    }
    if (best_axis < 0)
    {
        S->nodes[me].first = first;
        S->nodes[me].count = count;
        return me;
    }
    /* partition */
    const float c0 = cb.lo[best_axis], c1 = cb.hi[best_axis];
    const float sc = (float)BVH_BINS / (c1 - c0);
    uint32_t i = first, j = first + count;
    while (i < j)
    {
        const uint32_t p = S->prim_order[i];
        int b = (int)((B->pcen[3 * p + best_axis] - c0) * sc);
        if (b >= BVH_BINS) b = BVH_BINS - 1;
        if (b < 0) b = 0;
        if (b <= best_bin)
        {
            ++i;
        }
        else
        {
            --j;
            S->prim_order[i] = S->prim_order[j];
            S->prim_order[j] = p;
        }
    }
    const uint32_t nl = i - first;
    const uint32_t l = build_rec(B, first, nl);
    const uint32_t r = build_rec(B, first + nl, count - nl);
    S->nodes[me].left = l;
    S->nodes[me].right = r;
    S->nodes[me].count = 0;
    return me;
}

/* padded boxes and centres of all flattened triangles */
static void prepare_prims(pt_scene *S, build_ctx *B)
{
    const uint32_t n = S->n_tris;
    B->S = S;
    B->pbox = (box_t *)malloc(n * sizeof(box_t));
    B->pcen = (float *)malloc((size_t)n * 12);
    box_t sb;
    box_empty(&sb);
    for (uint32_t g = 0; g < n; ++g)
    {
        const wtri_t *T = &S->tris[g];
        const float a[3] = {T->v0.x, T->v0.y, T->v0.z};
        const float b[3] = {T->v0.x + T->e1.x, T->v0.y + T->e1.y,
                            T->v0.z + T->e1.z};
        const float c[3] = {T->v0.x + T->e2.x, T->v0.y + T->e2.y,
                            T->v0.z + T->e2.z};
        box_t *pb = &B->pbox[g];
        box_empty(pb);
        box_grow_pt(pb, a);
        box_grow_pt(pb, b);
        box_grow_pt(pb, c);
        for (int k = 0; k < 3; ++k)
        {
            B->pcen[3 * g + k] = 0.5f * (pb->lo[k] + pb->hi[k]);
        }
        box_grow(&sb, pb);
        S->prim_order[g] = g;
    }
    float mag = 0.0f;
    for (int k = 0; k < 3; ++k)
    {
        mag = pt_max(mag, pt_max(fabsf(sb.lo[k]), fabsf(sb.hi[k])));
    }
    B->pad = mag * (1.0f / 262144.0f);
    for (uint32_t g = 0; g < n; ++g)
    {
        for (int k = 0; k < 3; ++k)
        {
            B->pbox[g].lo[k] -= B->pad;
            B->pbox[g].hi[k] += B->pad;
        }
    }
}

static void build_bvh(pt_scene *S)
{
    const uint32_t n = S->n_tris;
    S->nodes = (bnode_t *)calloc(2 * (size_t)(n ? n : 1), sizeof(bnode_t));
    S->prim_order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    S->n_nodes = 0;
    S->sah_cost = 0.0;
    if (n == 0)
    {
        return;
    }
    build_ctx B;
    prepare_prims(S, &B);
    B.leaf_max = BVH_LEAF_MAX;
    build_rec(&B, 0, n);
    /* SAH cost: sum over internal nodes of A/Aroot + leaves A/Aroot * count */
    box_t rb;
    memcpy(rb.lo, S->nodes[0].lo, 12);
    memcpy(rb.hi, S->nodes[0].hi, 12);
    const double ra = box_area(&rb);
    double cost = 0.0;
    for (uint32_t i = 0; i < S->n_nodes; ++i)
    {
        box_t b;
        memcpy(b.lo, S->nodes[i].lo, 12);
        memcpy(b.hi, S->nodes[i].hi, 12);
        const double a = box_area(&b) / (ra > 0 ? ra : 1.0);
        cost += S->nodes[i].count ? a * S->nodes[i].count : a;
    }
    S->sah_cost = cost;
    free(B.pbox);
    free(B.pcen);
}

/* --- yardstick for a device-built wide tree ---------------------------------
 * The shipped library builds an 8-wide tree (bottom-up clustering, collapse,
 * leaf slots of <= 3 triangles). To judge its quality on the same scale, the
 * oracle builds a top-down binned-SAH binary tree down to single triangles,
 * collapses it by the same rule (open the child of largest area until 8 slots
 * are used; children with <= 3 triangles become leaf slots) and evaluates the
 * same cost: 1 for the root, A/A_root per internal slot, A/A_root * triangles
 * per leaf slot, with the exact (unquantised) float boxes. */

static uint32_t subtree_count(const bnode_t *N, uint32_t i, uint32_t *cnt)
{
    if (N[i].count)
    {
        cnt[i] = N[i].count;
    }
    else
    {
        cnt[i] = subtree_count(N, N[i].left, cnt) + subtree_count(N, N[i].right, cnt);
    }
    return cnt[i];
}

static double node_area(const bnode_t *n)
{
    box_t b;
    memcpy(b.lo, n->lo, 12);
    memcpy(b.hi, n->hi, 12);
    return box_area(&b);
}

static double wide_cost_rec(const bnode_t *N, const uint32_t *cnt, uint32_t root, double root_area,
                            uint32_t leaf_max, uint64_t *n_wide)
{
    uint32_t child[8];
    uint32_t n = 0;
    ++*n_wide;
    if (N[root].count)
    {
        child[n++] = root;
    }
    else
    {
        child[n++] = N[root].left;
        child[n++] = N[root].right;
        while (n < 8)
        {
            int pick = -1;
            double best = -1.0;
            for (uint32_t k = 0; k < n; ++k)
            {
                const double a = N[child[k]].count ? -1.0 : node_area(&N[child[k]]);
                if (a > best)
                {
                    best = a;
                    pick = (int)k;
                }
            }
            if (pick < 0)
            {
                break;
            }
            const uint32_t c = child[pick];
            child[pick] = N[c].left;
            child[n++] = N[c].right;
        }
    }
    double cost = 0.0;
    for (uint32_t k = 0; k < n; ++k)
    {
        const double a = node_area(&N[child[k]]) / root_area;
        if (cnt[child[k]] > leaf_max)
        {
            cost += a + wide_cost_rec(N, cnt, child[k], root_area, leaf_max, n_wide);
        }
        else
        {
            cost += a * cnt[child[k]];
        }
    }
    return cost;
}

double pt_scene_wide_sah_cost(const pt_scene *S, uint64_t *n_wide_nodes)
{
    const uint32_t n = S->n_tris;
    if (n_wide_nodes)
    {
        *n_wide_nodes = 0;
    }
    if (n == 0)
    {
        return 1.0;
    }
    pt_scene T = *S; /* shares the triangles; owns its own tree */
    T.nodes = (bnode_t *)calloc(2 * (size_t)n, sizeof(bnode_t));
    T.prim_order = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    T.n_nodes = 0;
    build_ctx B;
    prepare_prims(&T, &B);
    B.leaf_max = 1;
    build_rec(&B, 0, n);
    uint32_t *cnt = (uint32_t *)calloc(T.n_nodes, sizeof(uint32_t));
    subtree_count(T.nodes, 0, cnt);
    const double ra = node_area(&T.nodes[0]);
    uint64_t n_wide = 0;
    const double cost = 1.0 + wide_cost_rec(T.nodes, cnt, 0, ra > 0 ? ra : 1.0, 3, &n_wide);
    if (n_wide_nodes)
    {
        *n_wide_nodes = n_wide;
    }
    free(cnt);
    free(B.pbox);
    free(B.pcen);
    free(T.nodes);
    free(T.prim_order);
    return cost;
}

/* --- traversal of the oracle's own BVH -------------------------------------*/

#define PT_DIR_EPS 8.673617379884035e-19f /* 2^-60 */
#define PT_SLAB_SLACK (1.0f / 262144.0f)  /* per-axis slack on slab distances */
#define PT_TLIMIT_PAD 1.00006103515625f   /* 1 + 2^-14: pad on the culling t */

static inline float clamp_dir(float d)
{
    return fabsf(d) < PT_DIR_EPS ? copysignf(PT_DIR_EPS, d) : d;
}

static pt_hit intersect_bvh(const pt_scene *S, pt_vec3 o, pt_vec3 d, float tmax,
                            pt_stats *st)
{
    pt_hit best = {tmax, 0.0f, 0.0f, PT_NO_HIT};
    if (S->n_nodes == 0)
    {
        return best;
    }
    const float od[3] = {o.x, o.y, o.z};
    const float idir[3] = {1.0f / clamp_dir(d.x), 1.0f / clamp_dir(d.y),
                           1.0f / clamp_dir(d.z)};
    uint32_t stack[256];
    int sp = 0;
    stack[sp++] = 0;
    uint64_t nodes = 0, tris = 0;
    while (sp)
    {
        const bnode_t *N = &S->nodes[stack[--sp]];
        ++nodes;
        float tn = 0.0f, tf = best.t * PT_TLIMIT_PAD;
        for (int k = 0; k < 3; ++k)
        {
            const float t0 = (N->lo[k] - od[k]) * idir[k];
            const float t1 = (N->hi[k] - od[k]) * idir[k];
            const float e = pt_max(fabsf(t0), fabsf(t1)) * PT_SLAB_SLACK;
            tn = pt_max(tn, pt_min(t0, t1) - e);
            tf = pt_min(tf, pt_max(t0, t1) + e);
        }
        if (!(tn <= tf))
        {
            continue;
        }
        if (N->count)
        {
            for (uint32_t i = N->first; i < N->first + N->count; ++i)
            {
                const uint32_t g = S->prim_order[i];
                float t, u, v;
                ++tris;
                if (mt_test(o, d, &S->tris[g], tmax, &t, &u, &v))
                {
                    consider(&best, t, u, v, g);
                }
            }
        }
        else
        {
            stack[sp++] = N->left;
            stack[sp++] = N->right;
        }
    }
    if (st)
    {
        st->nodes_visited += nodes;
        st->tris_tested += tris;
    }
    return best;
}

static inline pt_hit intersect_any(const pt_scene *S, pt_vec3 o, pt_vec3 d,
                                   float tmax, int use_bvh, pt_stats *st)
{
    if (S->two_level)
    {
        return intersect_two_level(S, o, d, tmax, st);
    }
    return use_bvh ? intersect_bvh(S, o, d, tmax, st)
                   : intersect_brute(S, o, d, tmax, st);
}

void pt_intersect(const pt_scene *S, const float *rays, size_t n, float tmax,
                  int use_bvh, pt_hit *out, pt_stats *stats)
{
    pt_stats total;
    memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        pt_stats local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 256)
        for (long long i = 0; i < (long long)n; ++i)
        {
            out[i] = intersect_any(S, ld3(rays + 6 * i), ld3(rays + 6 * i + 3),
                                   tmax, use_bvh, &local);
            local.rays++;
        }
#pragma omp critical
        {
            total.rays += local.rays;
            total.nodes_visited += local.nodes_visited;
            total.tris_tested += local.tris_tested;
        }
    }
    if (stats)
    {
        *stats = total;
    }
}

/* --- the shipped library's wide BVH format, traversed on the CPU ------------*/

/* 80-byte node: 8 child slots with boxes quantised to 8 bits on a per-node
 * power-of-two grid. Internal children are contiguous from child_base in slot
 * order; the triangles of all leaf slots are contiguous from tri_base.
 * meta[s]: 0 empty; internal 0x20 | (24 + s); leaf (unary count << 5) | offset,
 * unary count in {1, 3, 7} for 1, 2, 3 triangles. */
typedef struct
{
    float origin[3];
    uint8_t exp[3];
    uint8_t imask;
    uint32_t child_base;
    uint32_t tri_base;
    uint8_t meta[8];
    uint8_t qlo[3][8];
    uint8_t qhi[3][8];
} node8_t;

/* 48-byte triangle */
typedef struct
{
    float v0[3];
    uint32_t instance;
    float e1[3];
    uint32_t primitive;
    float e2[3];
    uint32_t flat;
} tri8_t;

typedef struct
{
    uint32_t base, bits;
} group_t;

static pt_hit bvh8_trace(const node8_t *nodes, const tri8_t *tris, pt_vec3 o,
                         pt_vec3 d, float tmax, pt_stats *st)
{
    pt_hit best = {tmax, 0.0f, 0.0f, PT_NO_HIT};
    const float od[3] = {o.x, o.y, o.z};
    const float dc[3] = {clamp_dir(d.x), clamp_dir(d.y), clamp_dir(d.z)};
    const float idir[3] = {1.0f / dc[0], 1.0f / dc[1], 1.0f / dc[2]};
    const int neg[3] = {signbit(dc[0]) != 0, signbit(dc[1]) != 0,
                        signbit(dc[2]) != 0};
    const uint32_t oct = (uint32_t)(neg[0] | (neg[1] << 1) | (neg[2] << 2));
    const uint32_t octinv = 7u - oct;
    group_t stack[64];
    int sp = 0;
    uint32_t node_idx = 0;
    uint64_t n_nodes = 0, n_tris = 0;
    for (;;)
    {
        const node8_t *N = &nodes[node_idx];
        ++n_nodes;
        float s[3], bn[3], bf[3];
        for (int k = 0; k < 3; ++k)
        {
            const float scale = pt_u2f((uint32_t)N->exp[k] << 23);
            s[k] = scale * idir[k];
            const float b = (N->origin[k] - od[k]) * idir[k];
            const float e = fmaf(255.0f, fabsf(s[k]), fabsf(b)) * PT_SLAB_SLACK;
            bn[k] = b - e;
            bf[k] = b + e;
        }
        const float tlim = best.t * PT_TLIMIT_PAD;
        uint32_t hitmask = 0;
        for (uint32_t c = 0; c < 8; ++c)
        {
            const uint32_t meta = N->meta[c];
            if (meta == 0)
            {
                continue;
            }
            float tn = 0.0f, tf = tlim;
            for (int k = 0; k < 3; ++k)
            {
                const float qn = (float)(neg[k] ? N->qhi[k][c] : N->qlo[k][c]);
                const float qf = (float)(neg[k] ? N->qlo[k][c] : N->qhi[k][c]);
                tn = pt_max(tn, fmaf(qn, s[k], bn[k]));
                tf = pt_min(tf, fmaf(qf, s[k], bf[k]));
            }
            if (tn <= tf)
            {
                if ((N->imask >> c) & 1u)
                {
                    hitmask |= 1u << (24u + (c ^ octinv));
                }
                else
                {
                    hitmask |= (meta >> 5) << (meta & 31u);
                }
            }
        }
        group_t ng = {N->child_base, (hitmask & 0xff000000u) | N->imask};
        uint32_t tbits = hitmask & 0x00ffffffu;
        while (tbits)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(tbits);
            tbits &= tbits - 1u;
            const tri8_t *T = &tris[N->tri_base + i];
            wtri_t W;
            W.v0 = ld3(T->v0);
            W.e1 = ld3(T->e1);
            W.e2 = ld3(T->e2);
            float t, u, v;
            ++n_tris;
            if (mt_test(o, d, &W, tmax, &t, &u, &v))
            {
                consider(&best, t, u, v, T->flat);
            }
        }
        if (!(ng.bits & 0xff000000u))
        {
            if (sp == 0)
            {
                break;
            }
            ng = stack[--sp];
        }
        const uint32_t bit = 31u - (uint32_t)__builtin_clz(ng.bits);
        ng.bits &= ~(1u << bit);
        const uint32_t slot = (bit - 24u) ^ octinv;
        if (ng.bits & 0xff000000u)
        {
            stack[sp++] = ng;
        }
        node_idx =
            ng.base + (uint32_t)__builtin_popcount(ng.bits & 0xffu &
                                                   ((1u << slot) - 1u));
    }
    if (st)
    {
        st->nodes_visited += n_nodes;
        st->tris_tested += n_tris;
    }
    return best;
}

void pt_bvh8_intersect(const void *nodes, uint32_t n_nodes, const void *tris,
                       uint32_t n_tris, const float *rays, size_t n,
                       float tmax, pt_hit *out, pt_stats *stats)
{
    (void)n_nodes;
    (void)n_tris;
    pt_stats total;
    memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        pt_stats local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 256)
        for (long long i = 0; i < (long long)n; ++i)
        {
            out[i] = bvh8_trace((const node8_t *)nodes, (const tri8_t *)tris,
                                ld3(rays + 6 * i), ld3(rays + 6 * i + 3), tmax,
                                &local);
            local.rays++;
        }
#pragma omp critical
        {
            total.rays += local.rays;
            total.nodes_visited += local.nodes_visited;
            total.tris_tested += local.tris_tested;
        }
    }
    if (stats)
    {
        *stats = total;
    }
}

typedef struct
{
    const node8_t *nodes;
    uint32_t n_nodes;
    const tri8_t *tris;
    uint32_t n_tris;
    const float *world; /* 9 floats per flat triangle, may be NULL */
    uint8_t *seen;
    uint8_t *node_seen;
    double root_area;
    double cost;
    int err;
} val_ctx;

static void decode_box(const node8_t *N, uint32_t c, box_t *b)
{
    for (int k = 0; k < 3; ++k)
    {
        const float scale = pt_u2f((uint32_t)N->exp[k] << 23);
        b->lo[k] = fmaf((float)N->qlo[k][c], scale, N->origin[k]);
        b->hi[k] = fmaf((float)N->qhi[k][c], scale, N->origin[k]);
    }
}

static int box_inside(const box_t *in, const box_t *out)
{
    for (int k = 0; k < 3; ++k)
    {
        if (in->lo[k] < out->lo[k] || in->hi[k] > out->hi[k])
        {
            return 0;
        }
    }
    return 1;
}

/* returns the exact bounds of the triangles below node idx */
static box_t validate_node(val_ctx *V, uint32_t idx, int depth, box_t *decoded)
{
    box_t exact;
    box_empty(&exact);
    box_empty(decoded);
    if (idx >= V->n_nodes || depth > 60)
    {
        V->err = -1;
        return exact;
    }
    if (V->node_seen[idx])
    {
        V->err = -2;
        return exact;
    }
    V->node_seen[idx] = 1;
    const node8_t *N = &V->nodes[idx];
    uint32_t n_internal = 0;
    for (uint32_t c = 0; c < 8; ++c)
    {
        const uint32_t meta = N->meta[c];
        const int internal = (N->imask >> c) & 1;
        if (meta == 0)
        {
            if (internal)
            {
                V->err = -3;
            }
            continue;
        }
        box_t cb;
        decode_box(N, c, &cb);
        box_grow(decoded, &cb);
        box_t sub;
        box_empty(&sub);
        if (internal)
        {
            if (meta != (0x20u | (24u + c)))
            {
                V->err = -4;
            }
            box_t child_decoded;
            sub = validate_node(V, N->child_base + n_internal, depth + 1,
                                &child_decoded);
            ++n_internal;
            const double a = box_area(&cb);
            V->cost += V->root_area > 0 ? a / V->root_area : 0.0;
        }
        else
        {
            const uint32_t un = meta >> 5, off = meta & 31u;
            const uint32_t cnt = un == 1 ? 1 : un == 3 ? 2 : un == 7 ? 3 : 0;
            if (cnt == 0 || off + cnt > 24)
            {
                V->err = -5;
                continue;
            }
            for (uint32_t i = 0; i < cnt; ++i)
            {
                const uint32_t ti = N->tri_base + off + i;
                if (ti >= V->n_tris)
                {
                    V->err = -6;
                    continue;
                }
                const tri8_t *T = &V->tris[ti];
                if (T->flat >= V->n_tris || V->seen[T->flat])
                {
                    V->err = -7;
                    continue;
                }
                V->seen[T->flat] = 1;
                if (V->world &&
                    (memcmp(V->world + 9 * (size_t)T->flat, T->v0, 12) ||
                     memcmp(V->world + 9 * (size_t)T->flat + 3, T->e1, 12) ||
                     memcmp(V->world + 9 * (size_t)T->flat + 6, T->e2, 12)))
                {
                    V->err = -8;
                }
                const float a[3] = {T->v0[0], T->v0[1], T->v0[2]};
                const float b[3] = {T->v0[0] + T->e1[0], T->v0[1] + T->e1[1],
                                    T->v0[2] + T->e1[2]};
                const float cc[3] = {T->v0[0] + T->e2[0], T->v0[1] + T->e2[1],
                                     T->v0[2] + T->e2[2]};
                box_grow_pt(&sub, a);
                box_grow_pt(&sub, b);
                box_grow_pt(&sub, cc);
            }
            const double a = box_area(&cb);
            V->cost += V->root_area > 0 ? a / V->root_area * cnt : 0.0;
        }
        if (!box_inside(&sub, &cb))
        {
            V->err = -9;
        }
        box_grow(&exact, &sub);
    }
    return exact;
}

int pt_bvh8_validate(const void *nodes, uint32_t n_nodes, const void *tris,
                     uint32_t n_tris, const float *world_tris,
                     double *sah_cost)
{
    val_ctx V;
    memset(&V, 0, sizeof V);
    V.nodes = (const node8_t *)nodes;
    V.n_nodes = n_nodes;
    V.tris = (const tri8_t *)tris;
    V.n_tris = n_tris;
    V.world = world_tris;
    V.seen = (uint8_t *)calloc(n_tris ? n_tris : 1, 1);
    V.node_seen = (uint8_t *)calloc(n_nodes ? n_nodes : 1, 1);
    if (n_nodes == 0)
    {
        free(V.seen);
        free(V.node_seen);
        return -1;
    }
    /* root area from the root's decoded child boxes */
    box_t rb;
    box_empty(&rb);
    for (uint32_t c = 0; c < 8; ++c)
    {
        if (V.nodes[0].meta[c])
        {
            box_t cb;
            decode_box(&V.nodes[0], c, &cb);
            box_grow(&rb, &cb);
        }
    }
    V.root_area = box_area(&rb);
    V.cost = 1.0; /* the root node itself */
    box_t decoded;
    validate_node(&V, 0, 0, &decoded);
    for (uint32_t i = 0; i < n_tris && !V.err; ++i)
    {
        if (!V.seen[i])
        {
            V.err = -10;
        }
    }
    for (uint32_t i = 0; i < n_nodes && !V.err; ++i)
    {
        if (!V.node_seen[i])
        {
            V.err = -11;
        }
    }
    if (sah_cost)
    {
        *sah_cost = V.cost;
    }
    free(V.seen);
    free(V.node_seen);
    return V.err;
}

/* ------------------------------------------------------------------------ */
/* get_hit (closest_hit_common.glsl:59-89) + path loop (shader.rgen:60-100)   */
/* ------------------------------------------------------------------------ */

static void closest_hit(const pt_scene *S, payload_t *P, pt_vec3 ray_dir,
                        const pt_hit *h)
{
    const uint32_t ii = S->tri_inst[h->id];
    const inst_t *I = &S->insts[ii];
    const mesh_t *M = &S->meshes[I->mesh];
    const uint32_t prim = h->id - I->first_flat;
    const uint32_t i0 = M->indices[3 * prim + 0];
    const uint32_t i1 = M->indices[3 * prim + 1];
    const uint32_t i2 = M->indices[3 * prim + 2];
    const pt_vec3 v0 = ld3(M->positions + 3 * i0);
    const pt_vec3 v1 = ld3(M->positions + 3 * i1);
    const pt_vec3 v2 = ld3(M->positions + 3 * i2);
    const pt_vec3 n0 = ld3(M->normals + 3 * i0);
    const pt_vec3 n1 = ld3(M->normals + 3 * i1);
    const pt_vec3 n2 = ld3(M->normals + 3 * i2);
    const float bx = 1.0f - h->u - h->v, by = h->u, bz = h->v;
    const pt_vec3 object_position =
        pt_add(pt_add(pt_scale(v0, bx), pt_scale(v1, by)), pt_scale(v2, bz));
    const pt_vec3 object_normal =
        pt_add(pt_add(pt_scale(n0, bx), pt_scale(n1, by)), pt_scale(n2, bz));
    const pt_vec3 world_normal =
        pt_normalize(pt_xform_normal_rowvec(I->w2o, object_normal));
    const pt_vec3 world_position = pt_xform_point(I->o2w, object_position);
    const pt_vec3 nrm = pt_faceforward(world_normal, ray_dir, world_normal);
    material_scatter(I->kind, ray_dir, world_position, nrm, P);
}

/* shader.rgen:60-100; *bounces_out is the loop counter the shader returns
 * through its out parameter: the iteration the path ended in, max_depth for a
 * path that ran through all of them */
static pt_vec3 radiance(const pt_scene *S, payload_t *P,
                        const pt_render_params *rp, pt_stats *st,
                        uint32_t *bounces_out)
{
    pt_vec3 accumulated_color = pt_v3(0.0f, 0.0f, 0.0f);
    pt_vec3 accumulated_reflectance = pt_v3(1.0f, 1.0f, 1.0f);
    uint32_t bounces;
    for (bounces = 0; bounces < rp->max_depth; ++bounces)
    {
        P->attenuation = 1.0f;
        const pt_vec3 ray_dir = P->dir;
        const pt_hit h =
            intersect_any(S, P->origin, ray_dir, rp->tmax, rp->use_bvh, st);
        st->rays++;
        if (h.id == PT_NO_HIT)
        {
            const pt_vec3 c = miss_color(S, ray_dir);
            P->color = c;
            P->emissivity = c;
            P->hit_sky = 1;
        }
        else
        {
            closest_hit(S, P, ray_dir, &h);
        }
        accumulated_color =
            pt_add(accumulated_color, pt_mul(accumulated_reflectance, P->emissivity));
        if (P->hit_sky)
        {
            *bounces_out = bounces;
            return accumulated_color;
        }
        pt_vec3 hit_color = P->color;
        const float p = pt_max(hit_color.x, pt_max(hit_color.y, hit_color.z));
        if (pt_random(&P->rng) < p)
        {
            hit_color = pt_divs(hit_color, p);
        }
        else
        {
            *bounces_out = bounces;
            return accumulated_color;
        }
        accumulated_reflectance = pt_mul(
            accumulated_reflectance, pt_scale(hit_color, P->attenuation));
    }
    *bounces_out = bounces;
    return accumulated_color;
}

/* one invocation of shader.rgen main() for pixel (px, py) */
static void raygen(const pt_scene *S, const pt_push_constants *push,
                   const pt_render_params *rp, uint32_t px, uint32_t py,
                   float *accum_px, uint8_t *rgba8_px, pt_stats *st)
{
    const uint32_t pixel_index = py * rp->width + px;
    payload_t P;
    memset(&P, 0, sizeof P);
    P.rng = pt_seed(pixel_index, push->global_frame_count);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t s = 0; s < push->samples_per_frame; ++s)
    {
        camera_ray(push, rp->width, rp->height, px, py, &P.rng, &P.origin,
                   &P.dir);
        uint32_t bounces = 0;
        pt_vec3 c = radiance(S, &P, rp, st, &bounces);
        if (rp->debug_view == 1)
        {
            /* the branch under "#else" at shader.rgen:197-198 */
            const float b = (float)bounces / 16.0f;
            c = pt_v3(b, b, b);
        }
        st->paths++;
        acc[0] += c.x;
        acc[1] += c.y;
        acc[2] += c.z;
        acc[3] += 1.0f;
    }
    float avg[4];
    if (rp->accumulate_sum)
    {
        for (int k = 0; k < 4; ++k)
        {
            avg[k] = push->sample_count > 0 ? accum_px[k] + acc[k] : acc[k];
        }
    }
    else if (push->sample_count > 0)
    {
        const float sc = (float)push->sample_count;
        const float tot = (float)(push->sample_count + push->samples_per_frame);
        for (int k = 0; k < 4; ++k)
        {
            avg[k] = (accum_px[k] * sc + acc[k]) / tot;
        }
    }
    else
    {
        const float spf = (float)push->samples_per_frame;
        for (int k = 0; k < 4; ++k)
        {
            avg[k] = acc[k] / spf;
        }
    }
    memcpy(accum_px, avg, 16);
    if (rgba8_px)
    {
        pt_vec3 lin = pt_v3(avg[0], avg[1], avg[2]);
        if (rp->accumulate_sum)
        {
            lin = pt_divs(lin, avg[3]);
        }
        const pt_vec3 rc = rp->tonemap == 1 ? tonemap_aces(lin) : tonemap_pbr_neutral(lin);
        rgba8_px[0] = pt_unorm8(rc.x);
        rgba8_px[1] = pt_unorm8(rc.y);
        rgba8_px[2] = pt_unorm8(rc.z);
        rgba8_px[3] = 255;
    }
}

void pt_render(const pt_scene *S, const pt_push_constants *push,
               const pt_render_params *rp, float *accum, uint8_t *rgba8,
               pt_stats *stats)
{
    pt_stats total;
    memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (rp->n_threads > 0)
    {
        omp_set_num_threads(rp->n_threads);
    }
#endif
    const long long n_px = (long long)rp->width * rp->height;
#pragma omp parallel
    {
        pt_stats local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 64)
        for (long long i = 0; i < n_px; ++i)
        {
            const uint32_t px = (uint32_t)(i % rp->width);
            const uint32_t py = (uint32_t)(i / rp->width);
            pt_push_constants pc = *push;
            for (uint32_t f = 0; f < rp->n_frames; ++f)
            {
                raygen(S, &pc, rp, px, py, accum + 4 * i,
                       rgba8 ? rgba8 + 4 * i : NULL, &local);
                pc.global_frame_count += 1u;
                pc.sample_count += pc.samples_per_frame;
            }
        }
#pragma omp critical
        {
            total.rays += local.rays;
            total.paths += local.paths;
            total.nodes_visited += local.nodes_visited;
            total.tris_tested += local.tris_tested;
        }
    }
    if (stats)
    {
        *stats = total;
    }
}

/* The same dispatches for chosen pixels only: pixel_indices[k] = py * width + px
 * of a width x height frame; accum / rgba8 hold n entries in that order. A
 * raygen invocation depends on nothing but its own pixel, so this is exactly
 * what pt_render() would leave in those pixels. */
void pt_render_pixels(const pt_scene *S, const pt_push_constants *push,
                      const pt_render_params *rp, const uint32_t *pixel_indices,
                      size_t n, float *accum, uint8_t *rgba8, pt_stats *stats)
{
    pt_stats total;
    memset(&total, 0, sizeof total);
#pragma omp parallel
    {
        pt_stats local;
        memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 16)
        for (long long k = 0; k < (long long)n; ++k)
        {
            const uint32_t px = pixel_indices[k] % rp->width;
            const uint32_t py = pixel_indices[k] / rp->width;
            pt_push_constants pc = *push;
            for (uint32_t f = 0; f < rp->n_frames; ++f)
            {
                raygen(S, &pc, rp, px, py, accum + 4 * k,
                       rgba8 ? rgba8 + 4 * k : NULL, &local);
                pc.global_frame_count += 1u;
                pc.sample_count += pc.samples_per_frame;
            }
        }
#pragma omp critical
        {
            total.rays += local.rays;
            total.paths += local.paths;
            total.nodes_visited += local.nodes_visited;
            total.tris_tested += local.tris_tested;
        }
    }
    if (stats)
    {
        *stats = total;
    }
}

void pt_primary_hits(const pt_scene *S, const pt_push_constants *push,
                     const pt_render_params *rp, uint32_t *ids)
{
    const long long n_px = (long long)rp->width * rp->height;
#pragma omp parallel for schedule(dynamic, 64)
    for (long long i = 0; i < n_px; ++i)
    {
        const uint32_t px = (uint32_t)(i % rp->width);
        const uint32_t py = (uint32_t)(i / rp->width);
        uint32_t rng = pt_seed((uint32_t)i, push->global_frame_count);
        pt_vec3 o, d;
        camera_ray(push, rp->width, rp->height, px, py, &rng, &o, &d);
        const pt_hit h = intersect_any(S, o, d, rp->tmax, rp->use_bvh, NULL);
        ids[i] = h.id;
    }
}
