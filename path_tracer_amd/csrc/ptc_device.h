// Device programs of the path tracer: RNG, camera rays, wide-BVH traversal with
// Moller-Trumbore, the four material programs, the sky, the running-mean
// accumulation and the tone map. Each function cites the reference shader it
// stands in for (paths relative to the reference repository).
#pragma once

#include "ptc_device_math.h"

namespace ptc
{

// ---- data resident in HBM ---------------------------------------------------

struct DInstance // 128 bytes
{
    float o2w[12];      // object-to-world, 3x4 row-major
    float w2o[12];      // inverse of o2w
    uint32_t kind;      // PTC_KIND_*
    uint32_t vtx_offset; // first vertex of the mesh in the shared vertex arrays
    uint32_t tri_offset; // first triangle of the mesh in the shared index array
    uint32_t first_flat; // flattened index of this instance's primitive 0
    uint32_t n_tris;
    uint32_t blas_root;  // two-level scenes: root node of the mesh's own tree (kNoBlas: no triangles)
    uint32_t pad[2];
};
constexpr uint32_t kNoBlas = 0xffffffffu;
constexpr uint32_t kNoInstance = 0xffffffffu;

struct DScene
{
    const uint4 *__restrict__ nodes;   // 5 x uint4 per 80-byte node
    const float4 *__restrict__ tris;   // 3 x float4 per 48-byte triangle
    const float *__restrict__ positions; // 3 floats per vertex, all meshes
    const float *__restrict__ normals;
    const uint32_t *__restrict__ indices; // 3 per triangle, mesh-local vertex ids
    // What shading a hit reads, gathered per mesh triangle: 5 x float4 = positions
    // of the three corners (9 floats), their normals (9 floats), 2 unused.
    const float4 *__restrict__ shade;
    const DInstance *__restrict__ instances;
    // Two-level scenes only (nullptr otherwise): a binary tree over the world
    // boxes of the instances, n_inst_nodes nodes in depth-first order, two
    // float4 each: (box lo, index of the node to go on with when the box is
    // missed) and (box hi, instance index, or kNoInstance for an inner node).
    // nodes / tris then hold one tree per mesh in object space and rays enter an
    // instance through its w2o matrix.
    const float4 *__restrict__ inst_box;
    uint32_t n_inst_nodes;
    // two-level scenes: the same padded world boxes by instance index, two float4
    // (lo, hi) each, for a ray that knows which instance it wants to test again
    const float4 *__restrict__ inst_bounds;
    // flattened scenes: the (padded) box of the whole tree, for rays that start
    // outside it; lo > hi when the scene has no triangles
    float4 box_lo, box_hi;
    const float4 *__restrict__ env; // RGBA32F equirect map or nullptr
    uint32_t env_w, env_h;
    uint32_t n_instances;
    uint32_t n_tris;
    uint32_t n_nodes; // entries of nodes
};

// Push_constants of src/renderer.cpp:27-42 plus the build's explicit knobs.
struct DFrame
{
    uint32_t global_frame_count;
    uint32_t sample_count;
    uint32_t samples_per_frame;
    vec3 camera_position;
    vec3 camera_dir_x;
    vec3 camera_dir_y;
    vec3 camera_dir_z;
    float sensor_distance;
    float sensor_half_width;
    float sensor_half_height;
    float focus_distance;
    float aperture_radius;
    uint32_t width, height;
    uint32_t max_depth;
    float tmax;
    uint32_t n_frames;
    uint32_t frame_stride;
    uint32_t accumulate_sum;
    uint32_t tonemap; // PTC_TONEMAP_*
    uint32_t wavefront_bounces; // bounces run as trace / shade launch pairs before the drain launch
    uint32_t max_paths_in_flight; // workspace bound of the queue-based integrators
    uint32_t debug_view; // 1: a sample contributes bounces / 16 instead of its radiance (shader.rgen:195-199)
    uint32_t regroup;    // wavefront integrator: the queue is handed to the persistent launch sorted by ray origin and octant
};

struct Hit
{
    float t, u, v;
    uint32_t flat;     // PTC_NO_HIT on a miss
    uint32_t instance; // valid when flat != NO_HIT
    uint32_t primitive;
};

constexpr uint32_t kNoHit = 0xffffffffu;

// ---- RNG: shader_common.glsl:15-23, shader.rgen:34-42, :174-175 -------------

PTC_DEV uint32_t hash_u32(uint32_t x)
{
    x += x << 10;
    x ^= x >> 6;
    x += x << 3;
    x ^= x >> 11;
    x += x << 15;
    return x;
}

PTC_DEV float random_(uint32_t &s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    // uint -> float rounds to nearest even, so this can return 1.0
    return (float)s * (1.0f / 4294967296.0f);
}

PTC_DEV uint32_t seed_(uint32_t pixel_index, uint32_t global_frame_count)
{
    return hash_u32(pixel_index) + hash_u32(global_frame_count + 1u);
}

// ---- camera: shader.rgen:45-58, :181-191 ------------------------------------

PTC_DEV void sample_gaussian(uint32_t &rng, float &gx, float &gy)
{
    const float float_min = 1.175494351e-38f;
    const float r = __builtin_sqrtf(-2.0f * log_(fmax_(float_min, random_(rng))));
    const float theta = kTwoPi * random_(rng);
    float s, c;
    sincos_(theta, s, c);
    gx = r * c;
    gy = r * s;
}

PTC_DEV void sample_disk(uint32_t &rng, float &dx, float &dy)
{
    const float r = __builtin_sqrtf(random_(rng));
    const float theta = kTwoPi * random_(rng);
    float s, c;
    sincos_(theta, s, c);
    dx = r * c;
    dy = r * s;
}

PTC_DEV void camera_ray(const DFrame &f, uint32_t px, uint32_t py, uint32_t &rng,
                        vec3 &origin, vec3 &dir)
{
    float gx, gy;
    sample_gaussian(rng, gx, gy);
    const float offx = 0.5f + 0.375f * gx;
    const float offy = 0.5f + 0.375f * gy;
    const float uvx = 2.0f * ((float)px + offx) / (float)f.width - 1.0f;
    const float uvy = 2.0f * ((float)py + offy) / (float)f.height - 1.0f;
    float defx, defy;
    if (f.aperture_radius == 0.0f)
    {
        // The reference's default (src/camera.cpp:87): the lens sample is drawn all
        // the same (shader.rgen:184) and multiplied by zero. What survives is the
        // sign of each zero, which reaches the ray through sums with other zeros:
        // the two draws and the signs of cos / sin, without the square root and
        // the polynomials. (The disk radius sqrt(u) is +0 or positive, so a
        // product's sign is the cosine's or the sine's; the aperture is +0 or -0.)
        (void)random_(rng);
        bool s_negative, c_negative;
        sincos_signs(kTwoPi * random_(rng), s_negative, c_negative);
        const uint32_t zero = __float_as_uint(f.aperture_radius);
        defx = __uint_as_float(zero ^ (c_negative ? 0x80000000u : 0u));
        defy = __uint_as_float(zero ^ (s_negative ? 0x80000000u : 0u));
    }
    else
    {
        float dx, dy;
        sample_disk(rng, dx, dy);
        defx = f.aperture_radius * dx;
        defy = f.aperture_radius * dy;
    }
    const vec3 defocus_offset = f.camera_dir_x * defx + f.camera_dir_y * defy;
    origin = f.camera_position + defocus_offset;
    vec3 t = f.camera_dir_z * f.sensor_distance;
    t = t + (f.camera_dir_x * f.sensor_half_width) * uvx;
    t = t + (f.camera_dir_y * f.sensor_half_height) * uvy;
    t = t * f.focus_distance;
    dir = normalize(t - defocus_offset);
}

// ---- closest-hit helpers: closest_hit_common.glsl ---------------------------

PTC_DEV float offset_component(float p, float n)
{
    const int32_t of_i = (int32_t)(256.0f * n);
    const uint32_t bits = __float_as_uint(p);
    const uint32_t moved = (p < 0.0f) ? bits - (uint32_t)of_i : bits + (uint32_t)of_i;
    const float p_i = __uint_as_float(moved);
    const float origin = 1.0f / 32.0f;
    const float float_scale = 1.0f / 65536.0f;
    return __builtin_fabsf(p) < origin ? p + float_scale * n : p_i;
}

// closest_hit_common.glsl:25-44
PTC_DEV vec3 offset_position_along_normal(vec3 p, vec3 n)
{
    return v3(offset_component(p.x, n.x), offset_component(p.y, n.y),
              offset_component(p.z, n.z));
}

// closest_hit_common.glsl:46-52
PTC_DEV vec3 sample_sphere(uint32_t &rng)
{
    const float theta = kTwoPi * random_(rng);
    const float z = 2.0f * random_(rng) - 1.0f;
    const float r = __builtin_sqrtf(1.0f - z * z);
    float s, c;
    sincos_(theta, s, c);
    return v3(r * c, r * s, z);
}

// closest_hit_common.glsl:54-57
PTC_DEV vec3 reflect_diffuse(vec3 normal, uint32_t &rng)
{
    return normalize(normal + sample_sphere(rng));
}

// Ray_payload of shader_common.glsl:4-13. Material and miss programs build one
// and return it by value; every field is assigned exactly once on every path,
// which keeps the whole payload in registers.
struct Payload
{
    vec3 color;
    vec3 emissivity;
    vec3 origin;
    vec3 dir;
    float attenuation;
    uint32_t rng;
    bool hit_sky;
};

// diffuse.rchit:5-14, specular.rchit:5-14, emissive.rchit:5-14,
// dielectric.rchit:5-52, after get_hit() produced pos / nrm.
PTC_DEV Payload material_scatter(uint32_t kind, vec3 dir, vec3 pos, vec3 nrm, uint32_t rng)
{
    vec3 origin, out_dir, color, emissivity;
    float attenuation = 1.0f;
    // (diffuse.rchit and emissive.rchit differ in two constants only: one branch
    // for both, so that a wave with lanes of either kind samples the sphere once)
    if (kind == 0u || kind == 2u)
    {
        origin = offset_position_along_normal(pos, nrm);
        out_dir = reflect_diffuse(nrm, rng);
        const float e = kind == 2u ? 5.0f : 0.0f;
        color = kind == 2u ? v3(0.75f, 0.75f, 0.75f)
                           : v3((nrm.x + 1.0f) * 0.5f, (nrm.y + 1.0f) * 0.5f, (nrm.z + 1.0f) * 0.5f);
        emissivity = v3(e, e, e);
    }
    else if (kind == 1u)
    {
        origin = offset_position_along_normal(pos, nrm);
        out_dir = reflect(dir, nrm);
        color = v3(0.75f, 0.75f, 0.75f);
        emissivity = v3(0.0f, 0.0f, 0.0f);
    }
    else
    {
        color = v3(1.0f, 1.0f, 1.0f);
        emissivity = v3(0.0f, 0.0f, 0.0f);
        const bool into = dot(dir, nrm) < 0.0f;
        const vec3 normal = into ? nrm : -nrm;
        const vec3 reflected_dir = reflect(dir, normal);
        const float n_air = 1.0f;
        const float n_glass = 1.5f;
        const float n_ratio = into ? n_air / n_glass : n_glass / n_air;
        const float ddn = dot(dir, normal);
        const float cos2t = 1.0f - n_ratio * n_ratio * (1.0f - ddn * ddn);
        if (cos2t < 0.0f)
        {
            origin = offset_position_along_normal(pos, normal);
            out_dir = reflected_dir;
        }
        else
        {
            const float k = (into ? 1.0f : -1.0f) * (ddn * n_ratio + __builtin_sqrtf(cos2t));
            const vec3 transmitted_dir = normalize(dir * n_ratio - nrm * k);
            const float a = n_glass - n_air;
            const float b = n_glass + n_air;
            const float R0 = a * a / (b * b);
            const float c = 1.0f - (into ? -ddn : dot(transmitted_dir, nrm));
            const float Re = R0 + (1.0f - R0) * c * c * c * c * c;
            const float Tr = 1.0f - Re;
            const float Pr = 0.25f + 0.5f * Re;
            const float RP = Re / Pr;
            const float TP = Tr / (1.0f - Pr);
            const bool reflects = random_(rng) < Pr;
            attenuation = reflects ? RP : TP;
            origin = offset_position_along_normal(pos, reflects ? normal : -normal);
            out_dir = reflects ? reflected_dir : transmitted_dir;
        }
    }
    Payload P;
    P.color = color;
    P.emissivity = emissivity;
    P.origin = origin;
    P.dir = out_dir;
    P.attenuation = attenuation;
    P.rng = rng;
    P.hit_sky = false;
    return P;
}

PTC_DEV vec3 load3(const float *__restrict__ p)
{
    return v3(p[0], p[1], p[2]);
}

// get_hit() of closest_hit_common.glsl:59-89 followed by the material program
// selected by the instance's kind (the SBT record offset).
PTC_DEV Payload closest_hit(const DScene &S, vec3 ray_dir, const Hit &h, uint32_t rng)
{
    // The instance record (transforms, first triangle, kind). Most scenes have
    // a handful of instances, so the lanes of a wave usually want the same one:
    // fetch it once through the scalar cache then, per lane otherwise.
    float o2w[12], w2o[12];
    uint32_t kind, tri_offset;
    const uint32_t first_inst = __builtin_amdgcn_readfirstlane(h.instance);
    if (__builtin_amdgcn_ballot_w64(h.instance != first_inst) == 0ull)
    {
        typedef const __attribute__((address_space(4))) uint32_t *scalar_words;
        const scalar_words w = (scalar_words)(uintptr_t)(S.instances + first_inst);
#pragma unroll
        for (int i = 0; i < 12; ++i)
        {
            o2w[i] = __uint_as_float(w[i]);
            w2o[i] = __uint_as_float(w[12 + i]);
        }
        kind = w[24];
        tri_offset = w[26];
    }
    else
    {
        const DInstance *__restrict__ I = &S.instances[h.instance];
#pragma unroll
        for (int i = 0; i < 12; ++i)
        {
            o2w[i] = I->o2w[i];
            w2o[i] = I->w2o[i];
        }
        kind = I->kind;
        tri_offset = I->tri_offset;
    }
    // the triangle's shading record: positions and normals of its three corners
    // (what the reference gathers through the index buffer, closest_hit_common.glsl:61-69)
    const float4 *__restrict__ rec = S.shade + 5u * (size_t)(tri_offset + h.primitive);
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
    const vec3 v0 = v3(r0.x, r0.y, r0.z);
    const vec3 v1 = v3(r0.w, r1.x, r1.y);
    const vec3 v2 = v3(r1.z, r1.w, r2.x);
    const vec3 n0 = v3(r2.y, r2.z, r2.w);
    const vec3 n1 = v3(r3.x, r3.y, r3.z);
    const vec3 n2 = v3(r3.w, r4.x, r4.y);
    const float bx = 1.0f - h.u - h.v, by = h.u, bz = h.v;
    const vec3 object_position = (v0 * bx + v1 * by) + v2 * bz;
    const vec3 object_normal = (n0 * bx + n1 * by) + n2 * bz;
    const vec3 world_normal = normalize(xform_normal_rowvec(w2o, object_normal));
    const vec3 world_position = xform_point(o2w, object_position);
    const vec3 nrm = faceforward(world_normal, ray_dir, world_normal);
    return material_scatter(kind, ray_dir, world_position, nrm, rng);
}

// ---- miss program: shader.rmiss:11-29 ----------------------------------------

// i modulo n into [0, n) (repeat addressing). Texel coordinates of directions on
// the unit sphere land in [-1, n], so the division is a rare fallback.
PTC_DEV int wrap_index(int i, int n)
{
    if (__builtin_expect(i < -n || i >= 2 * n, 0))
    {
        const int r = i % n;
        return r < 0 ? r + n : r;
    }
    return i < 0 ? i + n : (i >= n ? i - n : i);
}

// bilinear, repeat addressing, no mips (src/renderer.cpp:861-878)
PTC_DEV vec3 env_lookup(const DScene &S, float u, float v)
{
    if (!(u == u))
    {
        u = 0.0f;
    }
    if (!(v == v))
    {
        v = 0.0f;
    }
    const float x = u * (float)S.env_w - 0.5f;
    const float y = v * (float)S.env_h - 0.5f;
    const float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    const int x0 = wrap_index((int)x0f, (int)S.env_w);
    const int x1 = wrap_index((int)x0f + 1, (int)S.env_w);
    const int y0 = wrap_index((int)y0f, (int)S.env_h);
    const int y1 = wrap_index((int)y0f + 1, (int)S.env_h);
    // (32-bit texel numbers: ptc_scene_create bounds the map at 2^28 texels)
    const uint32_t row0 = (uint32_t)y0 * S.env_w, row1 = (uint32_t)y1 * S.env_w;
    const float4 t00 = S.env[row0 + (uint32_t)x0];
    const float4 t10 = S.env[row0 + (uint32_t)x1];
    const float4 t01 = S.env[row1 + (uint32_t)x0];
    const float4 t11 = S.env[row1 + (uint32_t)x1];
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    vec3 c;
    c.x = (t00.x * gx + t10.x * fx) * gy + (t01.x * gx + t11.x * fx) * fy;
    c.y = (t00.y * gx + t10.y * fx) * gy + (t01.y * gx + t11.y * fx) * fy;
    c.z = (t00.z * gx + t10.z * fx) * gy + (t01.z * gx + t11.z * fx) * fy;
    return c;
}

PTC_DEV vec3 miss_color(const DScene &S, vec3 d)
{
    if (S.env == nullptr)
    {
        // the two-tone sky kept under "#if 0", shader.rmiss:13-18
        return d.y >= 0.0f ? v3(0.75f, 0.75f, 0.75f) : v3(0.1f, 0.1f, 0.1f);
    }
    const float azimuth = atan2_(d.z, d.x);
    const float inclination = acos_(d.y);
    const float u = (azimuth + kPi) / kTwoPi;
    const float v = inclination / kPi;
    return env_lookup(S, u, v);
}

// What the closest-hit or the miss program leaves in the payload for a ray
// that went in direction ray_dir: shader.rgen:68-78 with shader.rmiss:25-27.
PTC_DEV Payload bounce_payload(const DScene &S, const Hit &h, vec3 ray_origin, vec3 ray_dir, uint32_t rng)
{
    Payload P;
    if (h.flat == kNoHit)
    {
        const vec3 c = miss_color(S, ray_dir);
        P.color = c;
        P.emissivity = c;
        P.origin = ray_origin;
        P.dir = ray_dir;
        P.attenuation = 1.0f;
        P.rng = rng;
        P.hit_sky = true;
    }
    else
    {
        P = closest_hit(S, ray_dir, h, rng);
    }
    return P;
}

// A path between bounces, and the part of the path loop after the trace call,
// shader.rgen:80-96: adds the emission, ends the path at the sky or by Russian
// roulette, updates the throughput. Returns true when the path continues from
// (origin, dir).
struct PathState
{
    vec3 origin, dir;
    vec3 throughput, radiance;
    uint32_t rng;
};

PTC_DEV bool path_bounce(const DScene &S, const Hit &h, PathState &ps)
{
    const Payload P = bounce_payload(S, h, ps.origin, ps.dir, ps.rng);
    uint32_t rng = P.rng;
    const vec3 radiance = ps.radiance + ps.throughput * P.emissivity;
    vec3 hit_color = P.color;
    const float p = fmax_(hit_color.x, fmax_(hit_color.y, hit_color.z));
    bool alive = !P.hit_sky;
    vec3 throughput = ps.throughput;
    if (alive)
    {
        alive = random_(rng) < p;
        hit_color = hit_color / p;
        throughput = throughput * (hit_color * P.attenuation);
    }
    ps.origin = P.origin;
    ps.dir = P.dir;
    ps.throughput = throughput;
    ps.radiance = radiance;
    ps.rng = rng;
    return alive;
}

// ---- tone map + UNORM8: shader.rgen:141-162, :218-223 ------------------------

PTC_DEV vec3 tonemap_pbr_neutral(vec3 color)
{
    const float start_compression = 0.8f - 0.04f;
    const float desaturation = 0.15f;
    const float x = fmin_(color.x, fmin_(color.y, color.z));
    const float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
    color = v3(color.x - offset, color.y - offset, color.z - offset);
    const float peak = fmax_(color.x, fmax_(color.y, color.z));
    if (peak < start_compression)
    {
        return color;
    }
    const float d = 1.0f - start_compression;
    const float new_peak = 1.0f - d * d / (peak + d - start_compression);
    color = color * (new_peak / peak);
    const float g = 1.0f - 1.0f / (desaturation * (peak - new_peak) + 1.0f);
    const float one_minus_g = 1.0f - g;
    return v3(color.x * one_minus_g + new_peak * g, color.y * one_minus_g + new_peak * g,
              color.z * one_minus_g + new_peak * g);
}

// float -> UNORM8 (render target is R8G8B8A8_UNORM, src/renderer.cpp:1759-1760):
// clamp to [0,1] (NaN -> 0), scale by 255, round to nearest even
PTC_DEV uint32_t unorm8(float x)
{
    if (!(x > 0.0f))
    {
        return 0u;
    }
    if (x > 1.0f)
    {
        x = 1.0f;
    }
    return (uint32_t)__builtin_rintf(x * 255.0f);
}

// shader.rgen:107-135 (compiled out in the reference, "#if 1" at :218): the ACES
// fit of S. Hill. GLSL mat3 constructors are column-major, so the products
// below are input_mat * color and output_mat * color.
PTC_DEV float aces_fit(float v)
{
    const float a = v * (v + 0.0245786f) - 0.000090537f;
    const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}

PTC_DEV float clamp01(float x)
{
    return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
}

PTC_DEV vec3 tonemap_aces(vec3 c)
{
    const float x = __builtin_fmaf(0.04823f, c.z, __builtin_fmaf(0.35458f, c.y, 0.59719f * c.x));
    const float y = __builtin_fmaf(0.01566f, c.z, __builtin_fmaf(0.90834f, c.y, 0.07600f * c.x));
    const float z = __builtin_fmaf(0.83777f, c.z, __builtin_fmaf(0.13383f, c.y, 0.02840f * c.x));
    const float fx = aces_fit(x), fy = aces_fit(y), fz = aces_fit(z);
    const float r = __builtin_fmaf(-0.07367f, fz, __builtin_fmaf(-0.53108f, fy, 1.60475f * fx));
    const float g = __builtin_fmaf(-0.00605f, fz, __builtin_fmaf(1.10813f, fy, -0.10208f * fx));
    const float b = __builtin_fmaf(1.07602f, fz, __builtin_fmaf(-0.07276f, fy, -0.00327f * fx));
    return v3(clamp01(r), clamp01(g), clamp01(b));
}

PTC_DEV uint32_t pack_rgba8(vec3 lin, uint32_t tonemap)
{
    const vec3 rc = tonemap == 1u ? tonemap_aces(lin) : tonemap_pbr_neutral(lin);
    return unorm8(rc.x) | (unorm8(rc.y) << 8) | (unorm8(rc.z) << 16) | 0xff000000u;
}

// ---- wide-BVH traversal + Moller-Trumbore ------------------------------------
//
// Node (80 bytes = 5 x uint4):
//   w0: origin.x, origin.y, origin.z (float bits), exp_x | exp_y<<8 | exp_z<<16 | imask<<24
//   w1: child_base, tri_base, meta[0..3], meta[4..7]
//   w2: qlo_x[0..3], qlo_x[4..7], qlo_y[0..3], qlo_y[4..7]
//   w3: qlo_z[0..3], qlo_z[4..7], qhi_x[0..3], qhi_x[4..7]
//   w4: qhi_y[0..3], qhi_y[4..7], qhi_z[0..3], qhi_z[4..7]
// Triangle (48 bytes = 3 x float4): (v0, instance), (e1, primitive), (e2, flat).
//
// Closest hit = lexicographic minimum over (t, flat) of all triangles passing
// the Moller-Trumbore test below with 0 < t < tmax. Box tests only decide
// which triangles get tested. They are made conservative three ways: leaf boxes
// are padded by 2^-18 of the scene's largest coordinate at build time; every
// slab distance gets a per-axis slack of 2^-18 of the magnitudes it is computed
// from; and the current best t is padded by 2^-14 before it culls. Together
// these cover the rounding of the slab arithmetic and of the triangle test, so
// no triangle the test would accept is ever culled.

constexpr float kDirEps = 8.673617379884035e-19f; // 2^-60
constexpr float kSlabSlack = 1.0f / 262144.0f;     // 2^-18
constexpr float kTLimitPad = 1.00006103515625f;    // 1 + 2^-14
constexpr int kStackTotal = 64; // traversal stack entries per lane, LDS + scratch
constexpr int kLdsStack = 8;    // default number of entries per lane kept in LDS

PTC_DEV float clamp_dir(float d)
{
    return __builtin_fabsf(d) < kDirEps ? __builtin_copysignf(kDirEps, d) : d;
}

struct TraceCounters
{
    uint32_t nodes;
    uint32_t tris;
    // node visits served by a fetch shared with the rest of the wave (through the
    // scalar cache), and the number of such fetches: a wave of 64 lanes at one
    // node adds 64 to the first and 1 to the second
    uint32_t uniform_nodes;
    uint32_t uniform_fetches;
};

PTC_DEV float ubyte_f(uint32_t w, int i)
{
    return (float)((w >> (8 * i)) & 0xffu);
}

// Moller-Trumbore on (v0, e1, e2); accepts iff det != 0, u >= 0, v >= 0,
// u + v <= 1, 0 < t < tmax. The operation order is part of the hit definition.
PTC_DEV bool mt_test(vec3 o, vec3 d, vec3 v0, vec3 e1, vec3 e2, float tmax, float &t,
                     float &u, float &v)
{
    // straight-line code (det == 0 is part of the final condition, not an early
    // exit): a branch here makes the compiler sink part of the triangle load
    // below it, which costs a second memory round trip per test
    const vec3 p = cross(d, e2);
    const float det = dot(e1, p);
    const float inv = 1.0f / det;
    const vec3 tv = o - v0;
    const float uu = dot(tv, p) * inv;
    const vec3 q = cross(tv, e1);
    const float vv = dot(d, q) * inv;
    const float tt = dot(e2, q) * inv;
    t = tt;
    u = uu;
    v = vv;
    return det != 0.0f && uu >= 0.0f && vv >= 0.0f && (uu + vv) <= 1.0f && tt > 0.0f && tt < tmax;
}

// Per-ray constants of a traversal.
struct RayCtx
{
    vec3 o, d;
    float idx, idy, idz; // 1 / direction, direction clamped away from zero
    uint32_t octinv;     // 7 - octant; octant bit a set = direction negative on axis a
    float tmax;
};

PTC_DEV RayCtx make_ray_ctx(vec3 o, vec3 d, float tmax)
{
    RayCtx R;
    R.o = o;
    R.d = d;
    const float dcx = clamp_dir(d.x), dcy = clamp_dir(d.y), dcz = clamp_dir(d.z);
    R.idx = 1.0f / dcx;
    R.idy = 1.0f / dcy;
    R.idz = 1.0f / dcz;
    const uint32_t oct = (__float_as_uint(dcx) >> 31) | ((__float_as_uint(dcy) >> 31) << 1) |
                         ((__float_as_uint(dcz) >> 31) << 2);
    R.octinv = 7u - oct;
    R.tmax = tmax;
    return R;
}

PTC_DEV Hit no_hit(float tmax)
{
    Hit h;
    h.t = tmax;
    h.u = 0.0f;
    h.v = 0.0f;
    h.flat = kNoHit;
    h.instance = 0u;
    h.primitive = 0u;
    return h;
}

// A ray with a non-finite origin or direction (a path that met a degenerate
// normal, say) can hit nothing: every comparison of the triangle test fails on
// a NaN, and an infinite coordinate makes the barycentrics infinite or NaN. It
// must not enter the tree either: the box test takes hardware minima / maxima,
// which drop NaNs, so it would pass every box and the ray would walk all nodes
// and triangles (one such ray among 10^8 made a launch twenty times longer).
PTC_DEV bool ray_is_finite(vec3 o, vec3 d)
{
    // x * 0 is 0 for finite x and NaN for an infinity or a NaN
    const float z = ((o.x * 0.0f + o.y * 0.0f) + (o.z * 0.0f + d.x * 0.0f)) + (d.y * 0.0f + d.z * 0.0f);
    return z == 0.0f;
}

// A node group is (first internal child, hit bits 31..24 | imask 7..0): bit
// 24 + (slot ^ octinv) set = that slot's box was hit and is still to be
// visited; the highest bit is the nearest. A triangle group is (first triangle,
// pending bits 23..0). The root is the group (0, 0x80000000).
constexpr uint32_t kNodeBits = 0xff000000u;

// Pops the nearest pending child off a node group; returns its node index.
PTC_DEV uint32_t next_child(uint2 &ng, uint32_t octinv)
{
    const uint32_t bit = 31u - (uint32_t)__builtin_clz(ng.y);
    ng.y &= ~(1u << bit);
    const uint32_t slot = (bit - 24u) ^ octinv;
    return ng.x + (uint32_t)__builtin_popcount(ng.y & 0xffu & ((1u << slot) - 1u));
}

// Tests the 8 child boxes of one wide node (its five 16-byte words w0..w4)
// against the ray; yields the group of hit internal children and the group of
// triangles of hit leaf slots.
// (octinv is passed beside R: a caller that knows the node words and the octant
// to be the same in every lane hands in wave-uniform values, and everything that
// depends on them alone — exponent decode, near / far plane selection, the hit
// bits of each slot — is then computed once on the scalar unit.)
PTC_DEV void node_eval(const uint4 w0, const uint4 w1, const uint4 w2, const uint4 w3, const uint4 w4,
                       const RayCtx &R, uint32_t octinv, float best_t, uint2 &ng, uint2 &tg)
{
    const uint32_t imask = w0.w >> 24;
    const bool negx = !(octinv & 1u), negy = !(octinv & 2u), negz = !(octinv & 4u);
    const float sx = __uint_as_float((w0.w & 0xffu) << 23) * R.idx;
    const float sy = __uint_as_float(((w0.w >> 8) & 0xffu) << 23) * R.idy;
    const float sz = __uint_as_float(((w0.w >> 16) & 0xffu) << 23) * R.idz;
    const float bx = (__uint_as_float(w0.x) - R.o.x) * R.idx;
    const float by = (__uint_as_float(w0.y) - R.o.y) * R.idy;
    const float bz = (__uint_as_float(w0.z) - R.o.z) * R.idz;
    // per-axis slack: 2^-18 of the magnitudes entering each slab distance,
    // subtracted from entry distances and added to exit distances
    const float ex = __builtin_fmaf(255.0f, __builtin_fabsf(sx), __builtin_fabsf(bx)) * kSlabSlack;
    const float ey = __builtin_fmaf(255.0f, __builtin_fabsf(sy), __builtin_fabsf(by)) * kSlabSlack;
    const float ez = __builtin_fmaf(255.0f, __builtin_fabsf(sz), __builtin_fabsf(bz)) * kSlabSlack;
    const float bnx = bx - ex, bny = by - ey, bnz = bz - ez;
    const float bfx = bx + ex, bfy = by + ey, bfz = bz + ez;
    const float tlim = best_t * kTLimitPad;
    // near / far plane bytes per axis, chosen by the ray's direction sign
    const uint32_t nx0 = negx ? w3.z : w2.x, nx1 = negx ? w3.w : w2.y;
    const uint32_t fx0 = negx ? w2.x : w3.z, fx1 = negx ? w2.y : w3.w;
    const uint32_t ny0 = negy ? w4.x : w2.z, ny1 = negy ? w4.y : w2.w;
    const uint32_t fy0 = negy ? w2.z : w4.x, fy1 = negy ? w2.w : w4.y;
    const uint32_t nz0 = negz ? w4.z : w3.x, nz1 = negz ? w4.w : w3.y;
    const uint32_t fz0 = negz ? w3.x : w4.z, fz1 = negz ? w3.y : w4.w;
    // Per slot, the bits a hit contributes: (meta >> 5) << (meta & 31), where an
    // internal slot's index 24 + s is first XORed with octinv. Done on four
    // packed meta bytes at a time: internal slots are the ones with bits 4 and 3
    // of meta both set (leaf triangle offsets stay below 24).
    const uint32_t oct4 = octinv * 0x01010101u;
    uint32_t bits_of[8];
#pragma unroll
    for (int half = 0; half < 2; ++half)
    {
        const uint32_t meta4 = half == 0 ? w1.z : w1.w;
        const uint32_t inner4 = ((meta4 & (meta4 << 1)) & 0x10101010u) >> 4; // 1 per internal byte
        const uint32_t index4 = (meta4 ^ (oct4 & (inner4 * 0xffu))) & 0x1f1f1f1fu;
        const uint32_t count4 = (meta4 >> 5) & 0x07070707u;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            bits_of[4 * half + k] = ((count4 >> (8 * k)) & 0xffu) << ((index4 >> (8 * k)) & 0xffu);
        }
    }
    uint32_t hitmask = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c)
    {
        const int k = c & 3;
        const float tnx = __builtin_fmaf(ubyte_f(c < 4 ? nx0 : nx1, k), sx, bnx);
        const float tny = __builtin_fmaf(ubyte_f(c < 4 ? ny0 : ny1, k), sy, bny);
        const float tnz = __builtin_fmaf(ubyte_f(c < 4 ? nz0 : nz1, k), sz, bnz);
        const float tfx = __builtin_fmaf(ubyte_f(c < 4 ? fx0 : fx1, k), sx, bfx);
        const float tfy = __builtin_fmaf(ubyte_f(c < 4 ? fy0 : fy1, k), sy, bfy);
        const float tfz = __builtin_fmaf(ubyte_f(c < 4 ? fz0 : fz1, k), sz, bfz);
        // (slab distances are finite for finite input, so the hardware min / max
        // and the compare-select forms agree)
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, 0.0f));
        const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, tlim));
        hitmask |= tn <= tf ? bits_of[c] : 0u;
    }
    ng.x = w1.x;
    ng.y = (hitmask & kNodeBits) | imask;
    tg.x = w1.y;
    tg.y = hitmask & ~kNodeBits;
}

// Loads one wide node and tests it. kTryUniform = false skips the check for a
// wave-uniform node (kernels whose waves hold unrelated rays almost never pass it).
template <bool kTryUniform = true>
PTC_DEV void node_test(const DScene &S, const RayCtx &R, float best_t, uint32_t node_idx, uint2 &ng,
                       uint2 &tg, TraceCounters *ctr = nullptr)
{
    // When every active lane of the wave visits the same node (the root always;
    // camera rays of one pixel nearly always), fetch it once through the scalar
    // cache instead of 64 times through the vector memory path; if the lanes also
    // share the ray octant, the per-node part of the test runs on the scalar unit.
    const uint32_t first_idx = __builtin_amdgcn_readfirstlane(node_idx);
    if (kTryUniform && __builtin_amdgcn_ballot_w64(node_idx != first_idx) == 0ull)
    {
        // constant address space + wave-uniform address = scalar (SMEM) loads
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(4))) u32x4 *scalar_ptr;
        const scalar_ptr N = (scalar_ptr)(uintptr_t)(S.nodes + 5u * (size_t)first_idx);
        const u32x4 s0 = N[0], s1 = N[1], s2 = N[2], s3 = N[3], s4 = N[4];
        if (ctr != nullptr)
        {
            ctr->uniform_nodes++;
            const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
            const uint32_t me = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            ctr->uniform_fetches += me == (uint32_t)__builtin_ctzll(active) ? 1u : 0u;
        }
        const uint4 w0 = make_uint4(s0.x, s0.y, s0.z, s0.w), w1 = make_uint4(s1.x, s1.y, s1.z, s1.w);
        const uint4 w2 = make_uint4(s2.x, s2.y, s2.z, s2.w), w3 = make_uint4(s3.x, s3.y, s3.z, s3.w);
        const uint4 w4 = make_uint4(s4.x, s4.y, s4.z, s4.w);
        const uint32_t first_oct = __builtin_amdgcn_readfirstlane(R.octinv);
        if (__builtin_amdgcn_ballot_w64(R.octinv != first_oct) == 0ull)
        {
            node_eval(w0, w1, w2, w3, w4, R, first_oct, best_t, ng, tg);
        }
        else
        {
            node_eval(w0, w1, w2, w3, w4, R, R.octinv, best_t, ng, tg);
        }
    }
    else
    {
        const uint4 *__restrict__ N = S.nodes + 5u * (size_t)node_idx;
        const uint4 w0 = N[0], w1 = N[1], w2 = N[2], w3 = N[3], w4 = N[4];
        node_eval(w0, w1, w2, w3, w4, R, R.octinv, best_t, ng, tg);
    }
}

// Tests one triangle (its three 16-byte words) and keeps it if it beats the
// best hit in (t, flat) order.
PTC_DEV void tri_eval(const float4 a, const float4 b, const float4 c, const RayCtx &R, Hit &best)
{
    float t = 0.0f, u = 0.0f, v = 0.0f;
    const bool hit = mt_test(R.o, R.d, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), R.tmax, t, u, v);
    // selects, not a branch: every word of the record is then used on every
    // path, which keeps the record at three 16-byte loads (behind a branch the
    // compiler splits the ids off into loads of their own)
    const uint32_t flat = __float_as_uint(c.w);
    const bool better = hit && (t < best.t || (t == best.t && flat < best.flat));
    best.t = better ? t : best.t;
    best.u = better ? u : best.u;
    best.v = better ? v : best.v;
    best.flat = better ? flat : best.flat;
    best.instance = better ? __float_as_uint(a.w) : best.instance;
    best.primitive = better ? __float_as_uint(b.w) : best.primitive;
}

PTC_DEV void tri_test(const DScene &S, const RayCtx &R, uint32_t tri_index, Hit &best)
{
    const float4 *__restrict__ T = S.tris + 3u * (size_t)tri_index;
    tri_eval(T[0], T[1], T[2], R, best);
}

// Traversal stack: the first kLdsStack entries of a lane live in LDS (entry e
// of the lane at lds[e * stride]), deeper ones in a private spill array. The
// pieces are passed separately (not as one struct) so that the depth counter
// stays in a register and the LDS pointer keeps its address space.
// (lds always points into a __shared__ array. It is cast to the LDS address
// space explicitly: with a generic pointer the compiler folds the LDS and the
// spill access into one flat-pointer access, and flat loads take the slower
// vector-memory path on every pop.)
typedef uint32_t stack_entry_bits __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) stack_entry_bits *lds_stack_ptr;

template <int kLds>
PTC_DEV void stack_push(uint2 *lds, uint32_t stride, uint2 *spill, int &sp, uint2 e)
{
    if (sp < kLds)
    {
        const stack_entry_bits v = {e.x, e.y};
        ((lds_stack_ptr)lds)[sp * stride] = v;
    }
    else if (sp < kStackTotal)
    {
        spill[sp - kLds] = e;
    }
    ++sp;
}

template <int kLds>
PTC_DEV uint2 stack_pop(uint2 *lds, uint32_t stride, const uint2 *spill, int &sp)
{
    --sp;
    if (sp < kLds)
    {
        const stack_entry_bits v = ((lds_stack_ptr)lds)[sp * stride];
        return make_uint2(v.x, v.y);
    }
    return spill[sp - kLds];
}

// Closest hit of one ray, in a fixed order: visit the nearest pending child,
// test the triangles of its hit leaf slots at once, descend. (This order is the
// one a scalar re-implementation can follow, so node and triangle counts are
// reproducible.)
template <bool kCount, int kLds = kLdsStack>
PTC_DEV Hit trace_closest(const DScene &S, vec3 o, vec3 d, float tmax, uint2 *lds_stack,
                          uint32_t stride, TraceCounters *ctr, uint32_t root = 0u)
{
    Hit best = no_hit(tmax);
    if (S.n_tris == 0u || !ray_is_finite(o, d))
    {
        return best;
    }
    const RayCtx R = make_ray_ctx(o, d, tmax);
    uint2 spill[kStackTotal - kLds];
    int sp = 0;
    uint2 ng = make_uint2(root, 0x80000000u);
    for (;;)
    {
        const uint32_t node_idx = next_child(ng, R.octinv);
        if (ng.y & kNodeBits)
        {
            stack_push<kLds>(lds_stack, stride, spill, sp, ng);
        }
        uint2 tg;
        node_test(S, R, best.t, node_idx, ng, tg, kCount ? ctr : nullptr);
        if (kCount)
        {
            ctr->nodes++;
        }
        while (tg.y)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(tg.y);
            tg.y &= tg.y - 1u;
            tri_test(S, R, tg.x + i, best);
            if (kCount)
            {
                ctr->tris++;
            }
        }
        if (!(ng.y & kNodeBits))
        {
            if (sp == 0)
            {
                break;
            }
            ng = stack_pop<kLds>(lds_stack, stride, spill, sp);
        }
    }
    return best;
}

// Same closest hit, different schedule: every loop iteration runs at most one
// node visit and one triangle test per lane, and triangle groups found while
// older ones are pending wait on the stack. Lanes of a wave then stay in step
// (nobody waits through another lane's triangle loop), at the price of a
// later-shrinking t limit, so this order visits slightly more nodes than
// trace_closest(). The result is identical: the closest hit does not depend on
// the visit order.
template <int kLds = kLdsStack>
PTC_DEV Hit trace_closest_interleaved(const DScene &S, vec3 o, vec3 d, float tmax, uint2 *lds_stack,
                                      uint32_t stride, uint32_t root = 0u)
{
    Hit best = no_hit(tmax);
    if (S.n_tris == 0u || !ray_is_finite(o, d))
    {
        return best;
    }
    const RayCtx R = make_ray_ctx(o, d, tmax);
    uint2 spill[kStackTotal - kLds];
    int sp = 0;
    uint2 ng = make_uint2(root, 0x80000000u), tg = make_uint2(0u, 0u);
    for (;;)
    {
        if (!(ng.y & kNodeBits) && tg.y == 0u)
        {
            if (sp == 0)
            {
                break;
            }
            const uint2 e = stack_pop<kLds>(lds_stack, stride, spill, sp);
            if (e.y & kNodeBits)
            {
                ng = e;
            }
            else
            {
                tg = e;
            }
        }
        if (ng.y & kNodeBits)
        {
            const uint32_t node_idx = next_child(ng, R.octinv);
            if (ng.y & kNodeBits)
            {
                stack_push<kLds>(lds_stack, stride, spill, sp, ng);
            }
            if (tg.y)
            {
                stack_push<kLds>(lds_stack, stride, spill, sp, tg);
            }
            node_test(S, R, best.t, node_idx, ng, tg);
        }
        if (tg.y)
        {
            const uint32_t i = (uint32_t)__builtin_ctz(tg.y);
            tg.y &= tg.y - 1u;
            tri_test(S, R, tg.x + i, best);
        }
    }
    return best;
}

// Two-level scenes, the structure of the reference's TLAS over one BLAS per
// mesh (src/renderer.cpp:1044-1281): a ray that meets an instance's world box
// is taken to object space with the instance's w2o matrix (origin as a point,
// direction as a vector and not re-normalised, so t keeps its meaning) and
// traverses the mesh's own tree with the current best t as its limit (see
// instance_t_limit() for ties between instances). The instances sit in a tree
// over their boxes; a ray takes the ones it can meet nearest first when a
// short list holds them, in the tree's order otherwise.
// Conservative slab test of a ray against a padded box.
PTC_DEV bool ray_meets_box(float4 lo, float4 hi, vec3 o, vec3 d, float best_t)
{
    const float idx = 1.0f / clamp_dir(d.x), idy = 1.0f / clamp_dir(d.y), idz = 1.0f / clamp_dir(d.z);
    const float ax = (lo.x - o.x) * idx, bx = (hi.x - o.x) * idx;
    const float ay = (lo.y - o.y) * idy, by = (hi.y - o.y) * idy;
    const float az = (lo.z - o.z) * idz, bz = (hi.z - o.z) * idz;
    const float tn = fmax_(fmax_(fmin_(ax, bx), fmin_(ay, by)), fmin_(az, bz));
    const float tf = fmin_(fmin_(fmax_(ax, bx), fmax_(ay, by)), fmax_(az, bz));
    const float slack = (__builtin_fabsf(tn) + __builtin_fabsf(tf)) * kSlabSlack;
    return tn - slack <= tf + slack && tf + slack >= 0.0f && tn - slack <= best_t * kTLimitPad;
}

// The same test with hardware reciprocals (1 ulp) instead of divisions: their
// error, 2^-23 of each slab distance, disappears in the test's slack of 2^-18,
// so the test stays conservative; it decides only whether an instance is
// entered, never a hit.
PTC_DEV bool ray_meets_box_fast(float4 lo, float4 hi, vec3 o, vec3 d, float best_t)
{
    const float idx = __builtin_amdgcn_rcpf(clamp_dir(d.x)), idy = __builtin_amdgcn_rcpf(clamp_dir(d.y)),
                idz = __builtin_amdgcn_rcpf(clamp_dir(d.z));
    const float ax = (lo.x - o.x) * idx, bx = (hi.x - o.x) * idx;
    const float ay = (lo.y - o.y) * idy, by = (hi.y - o.y) * idy;
    const float az = (lo.z - o.z) * idz, bz = (hi.z - o.z) * idz;
    const float tn = fmax_(fmax_(fmin_(ax, bx), fmin_(ay, by)), fmin_(az, bz));
    const float tf = fmin_(fmin_(fmax_(ax, bx), fmax_(ay, by)), fmax_(az, bz));
    const float slack = (__builtin_fabsf(tn) + __builtin_fabsf(tf)) * kSlabSlack;
    return tn - slack <= tf + slack && tf + slack >= 0.0f && tn - slack <= best_t * kTLimitPad;
}

// The same test with the reciprocal direction at hand (a caller that tests many
// boxes against one ray divides once); *entry receives the conservative entry
// distance, clamped to 0 for a ray that starts inside the box.
PTC_DEV bool ray_meets_box_inv(float4 lo, float4 hi, vec3 o, float idx, float idy, float idz, float best_t,
                               float *entry)
{
    const float ax = (lo.x - o.x) * idx, bx = (hi.x - o.x) * idx;
    const float ay = (lo.y - o.y) * idy, by = (hi.y - o.y) * idy;
    const float az = (lo.z - o.z) * idz, bz = (hi.z - o.z) * idz;
    const float tn = fmax_(fmax_(fmin_(ax, bx), fmin_(ay, by)), fmin_(az, bz));
    const float tf = fmin_(fmin_(fmax_(ax, bx), fmax_(ay, by)), fmax_(az, bz));
    const float slack = (__builtin_fabsf(tn) + __builtin_fabsf(tf)) * kSlabSlack;
    *entry = fmax_(tn - slack, 0.0f);
    return tn - slack <= tf + slack && tf + slack >= 0.0f && tn - slack <= best_t * kTLimitPad;
}

// Walks the instance tree from node pos on to the next instance whose box the
// ray meets; returns its index (kNoInstance: none left) and leaves pos behind it.
// Needs no stack: a missed node names the node to go on with.
PTC_DEV uint32_t next_instance(const DScene &S, uint32_t &pos, vec3 o, vec3 d, float best_t)
{
    while (pos < S.n_inst_nodes)
    {
        const float4 lo = S.inst_box[2u * pos], hi = S.inst_box[2u * pos + 1u];
        if (ray_meets_box(lo, hi, o, d, best_t))
        {
            ++pos;
            if (__float_as_uint(hi.w) != kNoInstance)
            {
                return __float_as_uint(hi.w);
            }
        }
        else
        {
            pos = __float_as_uint(lo.w);
        }
    }
    return kNoInstance;
}

// The instances a ray can meet, nearest first: one walk of the whole instance
// tree that keeps the leaves whose boxes the ray meets, ordered by the distance
// at which it enters them (a ray that starts inside several boxes takes them in
// tree order). Up to four instances fit the list, one byte each, the nearest in
// the low byte, 0xff = no entry; a fifth one or an instance index beyond 254
// makes the function return false, and the caller walks the tree in its own
// order with next_instance() instead. Visiting near instances first lets the
// hit found there cull the far ones' boxes (once it has a hit, the caller tests
// an instance's box again, S.inst_bounds, before it enters the instance).
constexpr uint32_t kNoCandidates = 0xffffffffu;
PTC_DEV bool nearest_instances(const DScene &S, vec3 o, vec3 d, float tmax, uint32_t &list)
{
    const float idx = 1.0f / clamp_dir(d.x), idy = 1.0f / clamp_dir(d.y), idz = 1.0f / clamp_dir(d.z);
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f; // entry distances of the listed leaves
    uint32_t n = 0u;
    list = kNoCandidates;
    bool fits = true;
    for (uint32_t pos = 0u; pos < S.n_inst_nodes;)
    {
        const float4 lo = S.inst_box[2u * pos], hi = S.inst_box[2u * pos + 1u];
        float t;
        if (!ray_meets_box_inv(lo, hi, o, idx, idy, idz, tmax, &t))
        {
            pos = __float_as_uint(lo.w);
            continue;
        }
        if (__float_as_uint(hi.w) != kNoInstance)
        {
            const uint32_t id = __float_as_uint(hi.w);
            if (n == 4u || id >= 255u)
            {
                fits = false;
                break;
            }
            // insert behind every entry that is not farther (stable for equal distances)
            const uint32_t at = (n > 0u && t0 <= t ? 1u : 0u) + (n > 1u && t1 <= t ? 1u : 0u) + (n > 2u && t2 <= t ? 1u : 0u);
            const uint32_t keep = at == 0u ? 0u : (0xffffffffu >> (32u - 8u * at));
            list = (list & keep) | (id << (8u * at)) | ((list & ~keep) << 8);
            t3 = at <= 2u ? t2 : t;
            t2 = at <= 1u ? t1 : (at == 2u ? t : t2);
            t1 = at == 0u ? t0 : (at == 1u ? t : t1);
            t0 = at == 0u ? t : t0;
            ++n;
        }
        ++pos;
    }
    (void)t3;
    return fits;
}

// The t limit of a ray inside instance inst. Instances are met in the tree's
// order, not in index order, so the (t, instance, primitive) order of the hit
// definition takes a rule: a hit at exactly the best t wins against a best hit
// from an instance of larger index (limit one ulp above the best t), and loses
// against one of smaller index (the strict limit).
PTC_DEV float instance_t_limit(const Hit &best, uint32_t inst)
{
    const bool ties_win = best.flat != kNoHit && inst < best.instance;
    return ties_win ? __uint_as_float(__float_as_uint(best.t) + 1u) : best.t;
}

template <bool kCount, int kLds = kLdsStack>
PTC_DEV Hit trace_two_level(const DScene &S, vec3 o, vec3 d, float tmax, uint2 *lds_stack, uint32_t stride,
                            TraceCounters *ctr)
{
    Hit best = no_hit(tmax);
    // The instances the ray can meet, nearest first, so that a hit in a near one
    // rules out the boxes of the far ones; when the list cannot hold them, the
    // instance tree in its own order.
    uint32_t list = kNoCandidates, pos = S.n_inst_nodes; // (a non-finite ray meets no instance)
    bool listed = true;
    if (ray_is_finite(o, d) && !nearest_instances(S, o, d, tmax, list))
    {
        listed = false;
        pos = 0u;
    }
    for (;;)
    {
        uint32_t i = kNoInstance;
        if (listed)
        {
            while (list != kNoCandidates && i == kNoInstance)
            {
                const uint32_t id = list & 0xffu;
                list = (list >> 8) | 0xff000000u;
                // (without a hit so far the box passes as it did when the list was made)
                if (best.flat == kNoHit || ray_meets_box_fast(S.inst_bounds[2u * id], S.inst_bounds[2u * id + 1u], o, d, best.t))
                {
                    i = id;
                }
            }
        }
        else
        {
            i = next_instance(S, pos, o, d, best.t);
        }
        if (i == kNoInstance)
        {
            break;
        }
        const DInstance *__restrict__ I = &S.instances[i];
        const vec3 ro = xform_point(I->w2o, o);
        const vec3 rd = xform_vector(I->w2o, d);
        const float limit = instance_t_limit(best, i);
        const Hit h = kCount ? trace_closest<true, kLds>(S, ro, rd, limit, lds_stack, stride, ctr, I->blas_root)
                             : trace_closest_interleaved<kLds>(S, ro, rd, limit, lds_stack, stride, I->blas_root);
        if (h.flat != kNoHit)
        {
            // below the limit: nearer than the best so far, or as near in an instance of smaller index
            best = h;
            best.instance = i;
            best.flat = I->first_flat + h.primitive;
        }
    }
    return best;
}

// The closest-hit query of the kind the scene was built for. (A compile-time
// choice: kernels are instantiated for both kinds and picked by the host, so
// the flattened path carries none of the other's registers or code.)
template <bool kCount, bool kTwoLevel, int kLds = kLdsStack>
PTC_DEV Hit trace_scene(const DScene &S, vec3 o, vec3 d, float tmax, uint2 *lds_stack, uint32_t stride,
                        TraceCounters *ctr)
{
    if (kTwoLevel)
    {
        return trace_two_level<kCount, kLds>(S, o, d, tmax, lds_stack, stride, ctr);
    }
    return kCount ? trace_closest<true, kLds>(S, o, d, tmax, lds_stack, stride, ctr)
                  : trace_closest_interleaved<kLds>(S, o, d, tmax, lds_stack, stride);
}

// ---- path loop: shader.rgen:60-100 -------------------------------------------

// What a finished sample adds to its pixel: its radiance, or with the debug view
// of shader.rgen:195-199 (the branch under "#else") the value of the path loop's
// counter when the path ended, over 16. `bounces` is that counter: the iteration
// the path ended in, or max_depth for a path that ran through all of them.
PTC_DEV vec3 sample_value(uint32_t debug_view, vec3 radiance, uint32_t bounces)
{
    const float b = (float)bounces / 16.0f;
    return debug_view == 1u ? v3(b, b, b) : radiance;
}

template <bool kCount, bool kTwoLevel>
PTC_DEV vec3 radiance(const DScene &S, const DFrame &f, vec3 origin, vec3 dir, uint32_t &rng_io, uint2 *lds_stack,
                      uint32_t stride, uint32_t &rays, TraceCounters *ctr)
{
    PathState ps;
    ps.origin = origin;
    ps.dir = dir;
    ps.throughput = v3(1.0f, 1.0f, 1.0f);
    ps.radiance = v3(0.0f, 0.0f, 0.0f);
    ps.rng = rng_io;
    uint32_t bounces;
    for (bounces = 0; bounces < f.max_depth; ++bounces)
    {
        // counting runs use the reproducible visit order, plain runs the faster one
        const Hit h = trace_scene<kCount, kTwoLevel>(S, ps.origin, ps.dir, f.tmax, lds_stack, stride, ctr);
        ++rays;
        if (!path_bounce(S, h, ps))
        {
            break;
        }
    }
    rng_io = ps.rng;
    return sample_value(f.debug_view, ps.radiance, bounces);
}

// running mean of shader.rgen:202-216 (or the plain sum used when samples are
// sharded over devices)
PTC_DEV float4 accumulate(float4 old, float4 acc, uint32_t sample_count, uint32_t spf,
                          bool sum_mode)
{
    float4 r;
    if (sum_mode)
    {
        if (sample_count > 0u)
        {
            r = make_float4(old.x + acc.x, old.y + acc.y, old.z + acc.z, old.w + acc.w);
        }
        else
        {
            r = acc;
        }
    }
    else if (sample_count > 0u)
    {
        const float sc = (float)sample_count;
        const float tot = (float)(sample_count + spf);
        r = make_float4((old.x * sc + acc.x) / tot, (old.y * sc + acc.y) / tot,
                        (old.z * sc + acc.z) / tot, (old.w * sc + acc.w) / tot);
    }
    else
    {
        const float n = (float)spf;
        r = make_float4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
    }
    return r;
}

} // namespace ptc
