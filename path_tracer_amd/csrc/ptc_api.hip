// C ABI of libptcore (include/ptc.h): contexts, scene upload, films, render
// dispatch bookkeeping, read-back and PNG output.
#include "ptc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

namespace ptc
{

static thread_local std::string g_error;

void set_error(const std::string &msg)
{
    g_error = msg;
}

namespace
{

// Inverse of a 3x4 affine matrix: cofactor formula in double, rounded once to
// float. Stands in for the driver's gl_WorldToObjectEXT.
// Returns false for a singular (or non-finite) linear part.
bool inverse_affine(const float m[12], float out[12])
{
    const double a = m[0], b = m[1], c = m[2], tx = m[3];
    const double d = m[4], e = m[5], f = m[6], ty = m[7];
    const double g = m[8], h = m[9], i = m[10], tz = m[11];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double r = 1.0 / det;
    if (!(std::fabs(det) > 0.0) || !std::isfinite(r))
    {
        return false;
    }
    const double i00 = (e * i - f * h) * r, i01 = (c * h - b * i) * r, i02 = (b * f - c * e) * r;
    const double i10 = (f * g - d * i) * r, i11 = (a * i - c * g) * r, i12 = (c * d - a * f) * r;
    const double i20 = (d * h - e * g) * r, i21 = (b * g - a * h) * r, i22 = (a * e - b * d) * r;
    out[0] = (float)i00;
    out[1] = (float)i01;
    out[2] = (float)i02;
    out[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out[4] = (float)i10;
    out[5] = (float)i11;
    out[6] = (float)i12;
    out[7] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out[8] = (float)i20;
    out[9] = (float)i21;
    out[10] = (float)i22;
    out[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
    for (int k = 0; k < 12; ++k)
    {
        if (!std::isfinite(out[k]))
        {
            return false;
        }
    }
    return true;
}

vec3 to_v3(const ptc_vec3 &v)
{
    vec3 r;
    r.x = v.x;
    r.y = v.y;
    r.z = v.z;
    return r;
}

// Push_constants as draw_frame() fills them (src/renderer.cpp:2082-2094).
DFrame make_frame(const ptc_camera &cam, uint32_t width, uint32_t height)
{
    DFrame F{};
    F.camera_position = to_v3(cam.position);
    F.camera_dir_x = to_v3(cam.direction_x);
    F.camera_dir_y = to_v3(cam.direction_y);
    F.camera_dir_z = to_v3(cam.direction_z);
    F.sensor_distance = cam.sensor_distance;
    F.sensor_half_width = cam.sensor_half_width;
    F.sensor_half_height = cam.sensor_half_height;
    F.focus_distance = cam.focus_distance;
    F.aperture_radius = cam.aperture_radius;
    F.width = width;
    F.height = height;
    F.max_depth = 32;
    F.tmax = 10000.0f;
    F.n_frames = 1;
    F.frame_stride = 1;
    F.samples_per_frame = 1;
    F.tonemap = PTC_TONEMAP_PBR_NEUTRAL;
    F.wavefront_bounces = 0;
    F.max_paths_in_flight = 1u << 27;
    return F;
}

int use_device(const ptc_context *ctx)
{
    PTC_HIP_CHECK(hipSetDevice(ctx->device));
    return PTC_OK;
}

// ---- minimal PNG writer (8-bit RGBA, stored deflate blocks) -------------------

uint32_t crc_table[256];
bool crc_ready = false;

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    if (!crc_ready)
    {
        for (uint32_t i = 0; i < 256; ++i)
        {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k)
            {
                c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            }
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    for (size_t i = 0; i < n; ++i)
    {
        crc = crc_table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    }
    return crc;
}

void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data)
{
    put_be32(png, (uint32_t)data.size());
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    const uint32_t crc = crc32_update(0xffffffffu, png.data() + start, png.size() - start) ^ 0xffffffffu;
    put_be32(png, crc);
}

bool write_png_rgba8(const char *file_name, const uint8_t *rgba, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (4 * (size_t)w + 1));
    for (uint32_t y = 0; y < h; ++y)
    {
        raw.push_back(0); // filter: none
        raw.insert(raw.end(), rgba + (size_t)y * w * 4, rgba + (size_t)(y + 1) * w * 4);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    size_t pos = 0;
    do
    {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff));
        z.push_back((uint8_t)((~n >> 8) & 0xff));
        for (size_t i = 0; i < n; ++i)
        {
            const uint8_t b = raw[pos + i];
            z.push_back(b);
            s1 = (s1 + b) % 65521u;
            s2 = (s2 + s1) % 65521u;
        }
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (s2 << 16) | s1);

    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    ihdr.push_back(8); // bit depth
    ihdr.push_back(6); // RGBA
    ihdr.push_back(0);
    ihdr.push_back(0);
    ihdr.push_back(0);
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    FILE *f = std::fopen(file_name, "wb");
    if (!f)
    {
        return false;
    }
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    return std::fclose(f) == 0 && ok;
}

} // namespace
} // namespace ptc

using namespace ptc;

namespace
{
int render_impl(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const ptc_camera *camera,
                const ptc_render_params *params, bool static_scratch);

// Launches the dispatches a film has recorded (ptc_render_deferred) as one
// ptc_render; nothing to do when there are none. static_scratch: as one
// ptc_render_static instead (the caller is one and must not allocate).
int flush_film(ptc_film *film, bool static_scratch = false)
{
    if (film->pending.params.n_frames == 0u)
    {
        return PTC_OK;
    }
    const ptc_film::Pending job = film->pending;
    const bool reset_after = film->reset_pending;
    film->pending.params.n_frames = 0u;
    film->pending.scene = nullptr;
    film->reset_pending = false;
    const int rc = render_impl(film->ctx, job.scene, film, &job.camera, &job.params, static_scratch);
    if (rc == PTC_E_LIMIT || rc == PTC_E_NOMEM)
    {
        // refused before anything was launched (the run does not fit the static
        // workspace, or the workspace could not be allocated): the run stays
        // recorded, so the film's sample count still tells the truth and another
        // call (ptc_film_flush after a ptc_render_static, say) can launch it
        film->pending = job;
        film->reset_pending = reset_after;
        return rc;
    }
    if (reset_after)
    {
        film->sample_count = 0; // the reset that came after the run was recorded
    }
    return rc;
}

// A dispatch is about to overwrite a film that was reset with a run recorded:
// the run need not be launched any more.
void drop_reset_run(ptc_film *film)
{
    if (film->reset_pending)
    {
        film->pending.params.n_frames = 0u;
        film->pending.scene = nullptr;
        film->reset_pending = false;
        film->sample_count = 0;
    }
}

// Every film of the context; with a scene given, only the films whose recorded
// dispatches use it.
int flush_context(ptc_context *ctx, const ptc_scene *only_scene = nullptr)
{
    int rc = PTC_OK;
    for (ptc_film *film : ctx->films)
    {
        if (only_scene == nullptr || film->pending.scene == only_scene)
        {
            const int r = flush_film(film);
            rc = rc == PTC_OK ? r : rc;
        }
    }
    return rc;
}

// Device staging memory of at least `bytes` for a query call; calls on one
// context are ordered on its stream and end with a synchronisation, so one
// buffer serves them all.
int stage(ptc_context *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->stage_bytes)
    {
        PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->stage_buf);
        ctx->stage_buf = nullptr;
        ctx->stage_bytes = 0;
        if (hipMalloc(&ctx->stage_buf, bytes) != hipSuccess)
        {
            ctx->stage_buf = nullptr;
            set_error("out of device memory for " + std::to_string(bytes) + " bytes of staging");
            return PTC_E_NOMEM;
        }
        ctx->stage_bytes = bytes;
    }
    *out = ctx->stage_buf;
    return PTC_OK;
}

// PTC_INTEGRATOR_AUTO by the call's shape: wavefront for one sample per dispatch
// or 16+ dispatches in flight per pass, stream otherwise.
uint32_t resolve_integrator(const DFrame &F, uint32_t integrator)
{
    if (integrator != PTC_INTEGRATOR_AUTO)
    {
        return integrator;
    }
    // dispatches the wavefront integrator would hold in flight per pass
    const uint64_t per_dispatch = std::max<uint64_t>((uint64_t)F.width * F.height * F.samples_per_frame, 1u);
    const uint64_t in_flight = std::min<uint64_t>(F.n_frames, std::max<uint64_t>(F.max_paths_in_flight / per_dispatch, 1u));
    return F.samples_per_frame == 1u || in_flight >= 16u ? PTC_INTEGRATOR_WAVEFRONT : PTC_INTEGRATOR_STREAM;
}

uint32_t pending_samples(const ptc_film *film)
{
    return film->pending.params.n_frames * film->pending.params.samples_per_frame;
}

// The film's sample count as the caller sees it: recorded dispatches included,
// zero after a reset (whether or not the run recorded before it has been launched).
uint64_t visible_samples(const ptc_film *film)
{
    return film->reset_pending ? 0u : (uint64_t)film->sample_count + pending_samples(film);
}
} // namespace

extern "C" {

const char *ptc_last_error(void)
{
    return g_error.c_str();
}

int ptc_context_create(int device_ordinal, ptc_context **out)
{
    if (!out)
    {
        set_error("ptc_context_create: out is NULL");
        return PTC_E_INVALID;
    }
    *out = nullptr;
    int n_dev = 0;
    PTC_HIP_CHECK(hipGetDeviceCount(&n_dev));
    if (device_ordinal < 0 || device_ordinal >= n_dev)
    {
        set_error("ptc_context_create: no such device");
        return PTC_E_INVALID;
    }
    PTC_HIP_CHECK(hipSetDevice(device_ordinal));
    ptc_context *ctx = new (std::nothrow) ptc_context();
    if (!ctx)
    {
        set_error("out of host memory");
        return PTC_E_NOMEM;
    }
    ctx->device = device_ordinal;
    // every later failure releases what was created so far
    auto attempt = [&]() -> int {
        hipDeviceProp_t prop;
        PTC_HIP_CHECK(hipGetDeviceProperties(&prop, device_ordinal));
        ctx->n_cus = prop.multiProcessorCount;
        PTC_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
        PTC_HIP_CHECK(hipMalloc(&ctx->d_stats, sizeof(DeviceStats)));
        PTC_HIP_CHECK(hipMemsetAsync(ctx->d_stats, 0, sizeof(DeviceStats), ctx->stream));
        PTC_HIP_CHECK(hipEventCreate(&ctx->ev_start));
        PTC_HIP_CHECK(hipEventCreate(&ctx->ev_stop));
        PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // device address of the static workspace of ptc_render_static on this device
        return wavefront_static_init(ctx);
    };
    const int rc = attempt();
    if (rc != PTC_OK)
    {
        const std::string msg = g_error; // the teardown below may overwrite it
        if (ctx->ev_start)
        {
            (void)hipEventDestroy(ctx->ev_start);
        }
        if (ctx->ev_stop)
        {
            (void)hipEventDestroy(ctx->ev_stop);
        }
        (void)hipFree(ctx->d_stats);
        if (ctx->own_stream)
        {
            (void)hipStreamDestroy(ctx->stream);
        }
        delete ctx;
        set_error(msg);
        return rc;
    }
    *out = ctx;
    return PTC_OK;
}

void ptc_context_destroy(ptc_context *ctx)
{
    if (!ctx)
    {
        return;
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->wf_buf);
    (void)hipFree(ctx->stage_buf);
    (void)hipEventDestroy(ctx->ev_start);
    (void)hipEventDestroy(ctx->ev_stop);
    for (auto &p : ctx->ev_pool)
    {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto &e : ctx->phase_pool)
    {
        (void)hipEventDestroy(e.start);
        (void)hipEventDestroy(e.stop);
    }
    if (ctx->own_stream)
    {
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

int ptc_context_set_stream(ptc_context *ctx, void *hip_stream)
{
    if (!ctx)
    {
        set_error("ptc_context_set_stream: ctx is NULL");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream)
    {
        (void)hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return PTC_OK;
}

int ptc_synchronize(ptc_context *ctx)
{
    if (!ctx)
    {
        set_error("ptc_synchronize: ctx is NULL");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    const int rc = flush_context(ctx);
    if (rc != PTC_OK)
    {
        return rc;
    }
    PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PTC_OK;
}

namespace
{
constexpr float kMaxCoordinate = 1e9f; // of a vertex and of a transform entry
}

int ptc_scene_create(ptc_context *ctx, const ptc_mesh_desc *meshes, uint32_t n_meshes,
                     const ptc_instance_desc *instances, uint32_t n_instances,
                     const ptc_envmap_desc *envmap, const ptc_build_options *opts, ptc_scene **out)
{
    if (!ctx || !out || (n_meshes && !meshes) || (n_instances && !instances))
    {
        set_error("ptc_scene_create: NULL argument");
        return PTC_E_INVALID;
    }
    *out = nullptr;
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    // validate and lay the meshes out back to back
    std::vector<uint32_t> vtx_off(n_meshes + 1, 0), tri_off(n_meshes + 1, 0);
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        const ptc_mesh_desc &M = meshes[m];
        if ((M.n_vertices && (!M.positions || !M.normals)) || (M.n_triangles && !M.indices))
        {
            set_error("ptc_scene_create: mesh with NULL arrays");
            return PTC_E_INVALID;
        }
        for (size_t k = 0; k < 3 * (size_t)M.n_triangles; ++k)
        {
            if (M.indices[k] >= M.n_vertices)
            {
                set_error("ptc_scene_create: vertex index out of range");
                return PTC_E_INVALID;
            }
        }
        // box areas are products of two extents: beyond this bound they overflow binary32
        for (size_t k = 0; k < 3 * (size_t)M.n_vertices; ++k)
        {
            if (!(std::fabs(M.positions[k]) <= kMaxCoordinate))
            {
                set_error("ptc_scene_create: vertex coordinate not finite or beyond 1e9");
                return PTC_E_INVALID;
            }
        }
        const uint64_t v = (uint64_t)vtx_off[m] + M.n_vertices, t = (uint64_t)tri_off[m] + M.n_triangles;
        if (v > 0x7fffffffu || t > 0x7fffffffu)
        {
            set_error("ptc_scene_create: scene too large");
            return PTC_E_LIMIT;
        }
        vtx_off[m + 1] = (uint32_t)v;
        tri_off[m + 1] = (uint32_t)t;
    }
    std::vector<DInstance> inst(n_instances);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_instances; ++i)
    {
        const ptc_instance_desc &I = instances[i];
        if (I.mesh >= n_meshes || I.kind > PTC_KIND_DIELECTRIC)
        {
            set_error("ptc_scene_create: instance with bad mesh index or material kind");
            return PTC_E_INVALID;
        }
        for (int k = 0; k < 12; ++k)
        {
            if (!(std::fabs(I.transform[k]) <= kMaxCoordinate))
            {
                set_error("ptc_scene_create: instance transform not finite or beyond 1e9");
                return PTC_E_INVALID;
            }
        }
        DInstance &D = inst[i];
        std::memset(&D, 0, sizeof D);
        std::memcpy(D.o2w, I.transform, sizeof D.o2w);
        if (!inverse_affine(D.o2w, D.w2o))
        {
            // a zero-scale axis would send rays and normals through an inverse full of inf / NaN
            set_error("ptc_scene_create: instance transform is singular");
            return PTC_E_INVALID;
        }
        D.kind = I.kind;
        D.vtx_offset = vtx_off[I.mesh];
        D.tri_offset = tri_off[I.mesh];
        D.first_flat = (uint32_t)total;
        D.n_tris = meshes[I.mesh].n_triangles;
        total += D.n_tris;
        if (total > 0x7ffffff0u)
        {
            set_error("ptc_scene_create: more than 2^31 flattened triangles");
            return PTC_E_LIMIT;
        }
    }
    if (envmap && envmap->rgba && (envmap->width == 0 || envmap->height == 0))
    {
        set_error("ptc_scene_create: empty environment map");
        return PTC_E_INVALID;
    }
    if (envmap && envmap->rgba && (uint64_t)envmap->width * envmap->height > (1ull << 28))
    {
        set_error("ptc_scene_create: environment map beyond 2^28 texels");
        return PTC_E_LIMIT;
    }

    ptc_scene *S = new (std::nothrow) ptc_scene();
    if (!S)
    {
        set_error("out of host memory");
        return PTC_E_NOMEM;
    }
    S->ctx = ctx;
    for (const DInstance &D : inst)
    {
        S->first_flat.push_back(D.first_flat);
    }
    int rc = PTC_OK;
    auto fail = [&](int code) {
        ptc_scene_destroy(S);
        return code;
    };
    const size_t nv = vtx_off[n_meshes], nt = tri_off[n_meshes];
    hipStream_t st = ctx->stream;
#define PTC_TRY(expr)                                                                     \
    do                                                                                    \
    {                                                                                     \
        const hipError_t err_ = (expr);                                                   \
        if (err_ != hipSuccess)                                                           \
        {                                                                                 \
            set_error(std::string(#expr) + ": " + hipGetErrorString(err_));               \
            return fail(PTC_E_DEVICE);                                                    \
        }                                                                                 \
    } while (0)
    PTC_TRY(hipMalloc(&S->d_positions, std::max<size_t>(nv, 1) * 12));
    PTC_TRY(hipMalloc(&S->d_normals, std::max<size_t>(nv, 1) * 12));
    PTC_TRY(hipMalloc(&S->d_indices, std::max<size_t>(nt, 1) * 12));
    PTC_TRY(hipMalloc(&S->d_shade, std::max<size_t>(nt, 1) * 80));
    PTC_TRY(hipMalloc(&S->d_instances, std::max<size_t>(n_instances, 1) * sizeof(DInstance)));
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        const ptc_mesh_desc &M = meshes[m];
        if (M.n_vertices)
        {
            PTC_TRY(hipMemcpyAsync(S->d_positions + 3 * (size_t)vtx_off[m], M.positions,
                                   (size_t)M.n_vertices * 12, hipMemcpyHostToDevice, st));
            PTC_TRY(hipMemcpyAsync(S->d_normals + 3 * (size_t)vtx_off[m], M.normals,
                                   (size_t)M.n_vertices * 12, hipMemcpyHostToDevice, st));
        }
        if (M.n_triangles)
        {
            PTC_TRY(hipMemcpyAsync(S->d_indices + 3 * (size_t)tri_off[m], M.indices,
                                   (size_t)M.n_triangles * 12, hipMemcpyHostToDevice, st));
        }
    }
    if (n_instances)
    {
        PTC_TRY(hipMemcpyAsync(S->d_instances, inst.data(), n_instances * sizeof(DInstance),
                               hipMemcpyHostToDevice, st));
    }
    if (envmap && envmap->rgba)
    {
        const size_t eb = (size_t)envmap->width * envmap->height * 16;
        PTC_TRY(hipMalloc(&S->d_env, eb));
        PTC_TRY(hipMemcpyAsync(S->d_env, envmap->rgba, eb, hipMemcpyHostToDevice, st));
        S->d.env = S->d_env;
        S->d.env_w = envmap->width;
        S->d.env_h = envmap->height;
    }
    PTC_TRY(hipStreamSynchronize(st)); // host arrays may be freed by the caller after return
#undef PTC_TRY
    S->d.positions = S->d_positions;
    S->d.normals = S->d_normals;
    S->d.indices = S->d_indices;
    S->d.shade = S->d_shade;
    S->d.instances = S->d_instances;
    S->d.n_instances = n_instances;
    S->d.n_tris = (uint32_t)total;
    S->info.n_triangles = (uint32_t)total;
    S->info.n_instances = n_instances;

    rc = build_shade_records(ctx, S, vtx_off, tri_off);
    if (rc != PTC_OK)
    {
        return fail(rc);
    }
    BuildParams build_params;
    if (opts)
    {
        build_params.search_radius = opts->search_radius;
        build_params.collapse = opts->collapse;
        build_params.tri_cost = opts->tri_cost;
        build_params.optimize_rounds = opts->optimize_rounds;
        if (opts->collapse > 1u || !(opts->tri_cost >= 0.0f) || opts->tri_cost > 64.0f)
        {
            set_error("ptc_scene_create: unknown collapse rule or triangle cost outside [0, 64]");
            ptc_scene_destroy(S);
            return PTC_E_INVALID;
        }
    }
    if (const char *env = getenv("PTC_TRI_COST")) // measurement aid: overrides the triangle cost of the collapse
    {
        build_params.tri_cost = (float)atof(env);
    }
    if (const char *env = getenv("PTC_OPT_ROUNDS")) // measurement aid: rounds of reinsertion on the binary tree
    {
        const int r = atoi(env);
        build_params.optimize_rounds = r <= 0 ? kNoOptimize : (uint32_t)r;
    }
    if (opts && opts->two_level)
    {
        // object-space box of every mesh, from the host arrays
        std::vector<float> mesh_box(6 * std::max<size_t>(n_meshes, 1), 0.0f);
        for (uint32_t m = 0; m < n_meshes; ++m)
        {
            float *b = &mesh_box[6 * (size_t)m];
            b[0] = b[1] = b[2] = INFINITY;
            b[3] = b[4] = b[5] = -INFINITY;
            for (size_t v = 0; v < 3 * (size_t)meshes[m].n_vertices; ++v)
            {
                b[v % 3] = std::min(b[v % 3], meshes[m].positions[v]);
                b[3 + v % 3] = std::max(b[3 + v % 3], meshes[m].positions[v]);
            }
        }
        std::vector<uint32_t> mesh_of(n_instances);
        for (uint32_t i = 0; i < n_instances; ++i)
        {
            mesh_of[i] = instances[i].mesh;
        }
        rc = build_two_level(ctx, S, inst, vtx_off, tri_off, mesh_box, mesh_of, build_params);
    }
    else
    {
        rc = build_bvh(ctx, S, inst, (uint32_t)total, build_params);
    }
    if (rc != PTC_OK)
    {
        return fail(rc);
    }
    S->info.n_bvh_triangles = (uint32_t)(S->d_inst_box ? nt : total);
    S->info.device_bytes = (uint64_t)S->info.n_nodes * 80 + (uint64_t)(S->d_inst_box ? nt : total) * 48 + nv * 24 +
                           nt * (12 + 80) + (S->d_inst_box ? (uint64_t)n_instances * 96 : 0) +
                           n_instances * sizeof(DInstance) +
                           (S->d_env ? (uint64_t)S->d.env_w * S->d.env_h * 16 : 0);
    *out = S;
    return PTC_OK;
}

void ptc_scene_destroy(ptc_scene *scene)
{
    if (!scene)
    {
        return;
    }
    (void)hipSetDevice(scene->ctx->device);
    (void)flush_context(scene->ctx, scene); // recorded dispatches that use this scene run first
    (void)hipStreamSynchronize(scene->ctx->stream);
    (void)hipFree(scene->d_nodes);
    (void)hipFree(scene->d_tris);
    (void)hipFree(scene->d_positions);
    (void)hipFree(scene->d_normals);
    (void)hipFree(scene->d_indices);
    (void)hipFree(scene->d_shade);
    (void)hipFree(scene->d_instances);
    (void)hipFree(scene->d_inst_box);
    (void)hipFree(scene->d_env);
    delete scene;
}

int ptc_scene_get_info(const ptc_scene *scene, ptc_scene_info *out)
{
    if (!scene || !out)
    {
        set_error("ptc_scene_get_info: NULL argument");
        return PTC_E_INVALID;
    }
    *out = scene->info;
    return PTC_OK;
}

int ptc_scene_read_bvh(const ptc_scene *scene, ptc_bvh8_node *nodes, size_t n_nodes,
                       ptc_bvh_triangle *triangles, size_t n_triangles)
{
    if (!scene || !nodes || (!triangles && n_triangles))
    {
        set_error("ptc_scene_read_bvh: NULL argument");
        return PTC_E_INVALID;
    }
    if (n_nodes != scene->info.n_nodes || n_triangles != scene->info.n_bvh_triangles)
    {
        set_error("ptc_scene_read_bvh: buffer sizes do not match n_nodes / n_bvh_triangles of ptc_scene_get_info");
        return PTC_E_INVALID;
    }
    if (use_device(scene->ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    hipStream_t st = scene->ctx->stream;
    PTC_HIP_CHECK(hipMemcpyAsync(nodes, scene->d_nodes, n_nodes * 80, hipMemcpyDeviceToHost, st));
    if (n_triangles)
    {
        PTC_HIP_CHECK(hipMemcpyAsync(triangles, scene->d_tris, n_triangles * 48, hipMemcpyDeviceToHost, st));
    }
    PTC_HIP_CHECK(hipStreamSynchronize(st));
    return PTC_OK;
}

int ptc_film_create(ptc_context *ctx, uint32_t width, uint32_t height, void *accum_dev,
                    void *rgba8_dev, ptc_film **out)
{
    if (!ctx || !out || width == 0 || height == 0 || (uint64_t)width * height > 0x10000000u)
    {
        set_error("ptc_film_create: bad argument");
        return PTC_E_INVALID;
    }
    *out = nullptr;
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    ptc_film *F = new (std::nothrow) ptc_film();
    if (!F)
    {
        set_error("out of host memory");
        return PTC_E_NOMEM;
    }
    F->ctx = ctx;
    F->width = width;
    F->height = height;
    const size_t n = (size_t)width * height;
    if (accum_dev)
    {
        F->d_accum = static_cast<float4 *>(accum_dev);
    }
    else
    {
        if (hipMalloc(&F->d_accum, n * 16) != hipSuccess)
        {
            set_error("ptc_film_create: hipMalloc failed");
            delete F;
            return PTC_E_NOMEM;
        }
        F->own_accum = true;
        // draw_frame clears the storage image to (0,0,0,1) (src/renderer.cpp:2023-2048)
        (void)hipMemsetAsync(F->d_accum, 0, n * 16, ctx->stream);
    }
    if (rgba8_dev)
    {
        F->d_rgba8 = static_cast<uint32_t *>(rgba8_dev);
    }
    else
    {
        if (hipMalloc(&F->d_rgba8, n * 4) != hipSuccess)
        {
            set_error("ptc_film_create: hipMalloc failed");
            ptc_film_destroy(F);
            return PTC_E_NOMEM;
        }
        F->own_rgba8 = true;
        (void)hipMemsetAsync(F->d_rgba8, 0, n * 4, ctx->stream);
    }
    ctx->films.push_back(F);
    *out = F;
    return PTC_OK;
}

void ptc_film_destroy(ptc_film *film)
{
    if (!film)
    {
        return;
    }
    (void)hipSetDevice(film->ctx->device);
    (void)hipStreamSynchronize(film->ctx->stream);
    // (dispatches recorded and not launched are dropped with the film)
    auto &films = film->ctx->films;
    films.erase(std::remove(films.begin(), films.end(), film), films.end());
    if (film->own_accum)
    {
        (void)hipFree(film->d_accum);
    }
    if (film->own_rgba8)
    {
        (void)hipFree(film->d_rgba8);
    }
    delete film;
}

int ptc_film_reset(ptc_film *film)
{
    if (!film)
    {
        set_error("ptc_film_reset: film is NULL");
        return PTC_E_INVALID;
    }
    if (film->pending.params.n_frames != 0u)
    {
        // Dispatches recorded and not launched yet: in the reference they have been
        // submitted, so a read-back before the next dispatch shows them. They stay
        // recorded; what reads the film launches them, the next dispatch drops them.
        film->reset_pending = true;
    }
    else
    {
        film->sample_count = 0;
    }
    return PTC_OK;
}

uint32_t ptc_film_sample_count(const ptc_film *film)
{
    return film ? (uint32_t)visible_samples(film) : 0u;
}

int ptc_film_set_sample_count(ptc_film *film, uint32_t sample_count)
{
    if (!film)
    {
        set_error("ptc_film_set_sample_count: film is NULL");
        return PTC_E_INVALID;
    }
    if (sample_count == visible_samples(film))
    {
        return PTC_OK; // already there, recorded dispatches included
    }
    const int rc = flush_film(film);
    film->sample_count = sample_count;
    return rc;
}

int ptc_film_read_accum(ptc_film *film, float *rgba_out)
{
    if (!film || !rgba_out)
    {
        set_error("ptc_film_read_accum: NULL argument");
        return PTC_E_INVALID;
    }
    if (use_device(film->ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_film(film); rc != PTC_OK)
    {
        return rc;
    }
    const size_t n = (size_t)film->width * film->height;
    PTC_HIP_CHECK(hipMemcpyAsync(rgba_out, film->d_accum, n * 16, hipMemcpyDeviceToHost, film->ctx->stream));
    PTC_HIP_CHECK(hipStreamSynchronize(film->ctx->stream));
    return PTC_OK;
}

int ptc_film_read_rgba8(ptc_film *film, uint8_t *rgba_out)
{
    if (!film || !rgba_out)
    {
        set_error("ptc_film_read_rgba8: NULL argument");
        return PTC_E_INVALID;
    }
    if (use_device(film->ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_film(film); rc != PTC_OK)
    {
        return rc;
    }
    const size_t n = (size_t)film->width * film->height;
    PTC_HIP_CHECK(hipMemcpyAsync(rgba_out, film->d_rgba8, n * 4, hipMemcpyDeviceToHost, film->ctx->stream));
    PTC_HIP_CHECK(hipStreamSynchronize(film->ctx->stream));
    return PTC_OK;
}

void *ptc_film_accum_device_ptr(ptc_film *film)
{
    if (!film)
    {
        return nullptr;
    }
    // like the read-backs: what the pointer shows includes every recorded dispatch
    // (launched on the context's stream; work the caller enqueues there afterwards
    // is ordered behind it). If that launch fails there is no pointer to a film
    // worth reading: NULL, with the reason in ptc_last_error().
    if (use_device(film->ctx) != PTC_OK || flush_film(film) != PTC_OK)
    {
        return nullptr;
    }
    return film->d_accum;
}

int ptc_film_resolve(ptc_film *film, int sum_mode)
{
    if (!film)
    {
        set_error("ptc_film_resolve: film is NULL");
        return PTC_E_INVALID;
    }
    if (use_device(film->ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_film(film); rc != PTC_OK)
    {
        return rc;
    }
    return launch_resolve(film->ctx, film, sum_mode);
}

} // extern "C"

namespace
{
// ptc_render and ptc_render_static: the same dispatch bookkeeping and kernels;
// the second takes its workspace from the library's static device array and
// creates no events, so nothing in it allocates.
int render_impl(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const ptc_camera *camera,
                const ptc_render_params *params, bool static_scratch)
{
    if (!ctx || !scene || !film || !camera || !params)
    {
        set_error("ptc_render: NULL argument");
        return PTC_E_INVALID;
    }
    if (scene->ctx != ctx || film->ctx != ctx)
    {
        set_error("ptc_render: scene or film belongs to another context");
        return PTC_E_INVALID;
    }
    if (params->samples_per_frame == 0 || params->n_frames == 0)
    {
        set_error("ptc_render: samples_per_frame and n_frames must be >= 1");
        return PTC_E_INVALID;
    }
    if (params->integrator > PTC_INTEGRATOR_AUTO || params->accumulate > PTC_ACCUM_SUM)
    {
        set_error("ptc_render: unknown integrator or accumulation mode");
        return PTC_E_INVALID;
    }
    const uint64_t added = (uint64_t)params->samples_per_frame * params->n_frames;
    if (visible_samples(film) + added > 0xffffffffull)
    {
        set_error("ptc_render: sample count overflow");
        return PTC_E_LIMIT;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    // dispatches recorded on this film come first (never the case when this is the flush itself),
    // unless the film was reset after them: this call overwrites what they would render
    drop_reset_run(film);
    if (const int rc = flush_film(film, static_scratch); rc != PTC_OK)
    {
        return rc;
    }
    DFrame F = make_frame(*camera, film->width, film->height);
    F.global_frame_count = params->global_frame_count;
    F.sample_count = film->sample_count;
    F.samples_per_frame = params->samples_per_frame;
    F.n_frames = params->n_frames;
    F.frame_stride = params->frame_stride ? params->frame_stride : 1u;
    F.max_depth = params->max_depth ? params->max_depth : 32u;
    F.tmax = params->tmax > 0.0f ? params->tmax : 10000.0f;
    F.accumulate_sum = params->accumulate == PTC_ACCUM_SUM ? 1u : 0u;
    if (params->tonemap > PTC_TONEMAP_ACES)
    {
        set_error("ptc_render: unknown tonemap");
        return PTC_E_INVALID;
    }
    F.tonemap = params->tonemap;
    film->tonemap = params->tonemap;
    F.wavefront_bounces = params->wavefront_bounces;
    F.max_paths_in_flight = params->max_paths_in_flight ? params->max_paths_in_flight : kDefaultPathsInFlight;
    if (params->debug_view > 1u || params->regroup > 1u)
    {
        set_error("ptc_render: unknown debug_view or regroup value");
        return PTC_E_INVALID;
    }
    F.debug_view = params->debug_view;
    F.regroup = params->regroup;
    // one event pair per call, kept until ptc_render_times() collects them
    // (the static variant records none: creating events allocates)
    constexpr size_t kMaxTimed = 256;
    if (!static_scratch && ctx->ev_used == ctx->ev_pool.size() && ctx->ev_pool.size() < kMaxTimed)
    {
        hipEvent_t a, b;
        PTC_HIP_CHECK(hipEventCreate(&a));
        PTC_HIP_CHECK(hipEventCreate(&b));
        ctx->ev_pool.emplace_back(a, b);
    }
    const bool timed = !static_scratch && ctx->ev_used < ctx->ev_pool.size();
    hipEvent_t ev0 = timed ? ctx->ev_pool[ctx->ev_used].first : ctx->ev_start;
    hipEvent_t ev1 = timed ? ctx->ev_pool[ctx->ev_used].second : ctx->ev_stop;
    if (!static_scratch)
    {
        PTC_HIP_CHECK(hipEventRecord(ev0, ctx->stream));
    }
    int rc;
    const uint32_t integrator = resolve_integrator(F, params->integrator);
    ctx->no_events = static_scratch;
    if (integrator != PTC_INTEGRATOR_MEGAKERNEL)
    {
        rc = launch_render_wavefront(ctx, scene, film, F, integrator == PTC_INTEGRATOR_STREAM, static_scratch,
                                     params->instrument != 0);
    }
    else
    {
        rc = launch_render_mega(ctx, scene, film, F, params->instrument != 0);
    }
    ctx->no_events = false;
    if (rc != PTC_OK)
    {
        return rc;
    }
    if (!static_scratch)
    {
        PTC_HIP_CHECK(hipEventRecord(ev1, ctx->stream));
    }
    if (timed)
    {
        ++ctx->ev_used;
    }
    film->sample_count += (uint32_t)added;
    return PTC_OK;
}
} // namespace

extern "C" {

int ptc_render(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const ptc_camera *camera,
               const ptc_render_params *params)
{
    return render_impl(ctx, scene, film, camera, params, false);
}

int ptc_render_static(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const ptc_camera *camera,
                      const ptc_render_params *params)
{
    return render_impl(ctx, scene, film, camera, params, true);
}

int ptc_render_workspace(uint32_t width, uint32_t height, const ptc_render_params *params, int static_workspace,
                         uint64_t *bytes_out, uint32_t *dispatches_per_batch_out)
{
    if (!params || width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28))
    {
        set_error("ptc_render_workspace: NULL parameters or a film of 0 or more than 2^28 pixels");
        return PTC_E_INVALID;
    }
    if (params->samples_per_frame == 0 || params->n_frames == 0 || params->integrator > PTC_INTEGRATOR_AUTO)
    {
        set_error("ptc_render_workspace: samples_per_frame and n_frames must be >= 1, the integrator a known one");
        return PTC_E_INVALID;
    }
    DFrame F{};
    F.width = width;
    F.height = height;
    F.samples_per_frame = params->samples_per_frame;
    F.n_frames = params->n_frames;
    F.max_depth = params->max_depth ? params->max_depth : 32u;
    F.wavefront_bounces = params->wavefront_bounces;
    F.max_paths_in_flight = params->max_paths_in_flight ? params->max_paths_in_flight : kDefaultPathsInFlight;
    const uint32_t integrator = resolve_integrator(F, params->integrator);
    WavefrontPlan plan;
    if (integrator == PTC_INTEGRATOR_MEGAKERNEL)
    {
        plan.dispatches_per_batch = params->n_frames; // all dispatches of a pixel run in one lane; no workspace
    }
    else if (const int rc = plan_wavefront(F, integrator == PTC_INTEGRATOR_STREAM, static_workspace != 0, &plan);
             rc != PTC_OK)
    {
        return rc;
    }
    if (bytes_out)
    {
        *bytes_out = plan.bytes;
    }
    if (dispatches_per_batch_out)
    {
        *dispatches_per_batch_out = plan.dispatches_per_batch;
    }
    return PTC_OK;
}

int ptc_render_deferred(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const ptc_camera *camera,
                        const ptc_render_params *params)
{
    if (!ctx || !scene || !film || !camera || !params)
    {
        set_error("ptc_render_deferred: NULL argument");
        return PTC_E_INVALID;
    }
    if (scene->ctx != ctx || film->ctx != ctx)
    {
        set_error("ptc_render_deferred: scene or film belongs to another context");
        return PTC_E_INVALID;
    }
    if (params->samples_per_frame == 0 || params->n_frames == 0)
    {
        set_error("ptc_render_deferred: samples_per_frame and n_frames must be >= 1");
        return PTC_E_INVALID;
    }
    if (params->integrator > PTC_INTEGRATOR_AUTO || params->accumulate > PTC_ACCUM_SUM ||
        params->tonemap > PTC_TONEMAP_ACES || params->debug_view > 1u || params->regroup > 1u)
    {
        set_error("ptc_render_deferred: unknown integrator, accumulation mode, tonemap, debug_view or regroup value");
        return PTC_E_INVALID;
    }
    const uint64_t added = (uint64_t)params->samples_per_frame * params->n_frames;
    if (visible_samples(film) + added > 0xffffffffull)
    {
        set_error("ptc_render_deferred: sample count overflow");
        return PTC_E_LIMIT;
    }
    // A film over memory of the caller's (ptc_film_create with accum_dev or
    // rgba8_dev) is launched at once: such a caller orders its own work after the
    // dispatch by the stream (a reduction of the accumulator enqueued behind it, an
    // event, a stream synchronisation), and a dispatch that is only recorded would
    // not be on the stream yet.
    if (!film->own_accum || !film->own_rgba8)
    {
        return render_impl(ctx, scene, film, camera, params, false);
    }
    drop_reset_run(film); // this dispatch overwrites a film that was reset after its recorded run
    ptc_render_params want = *params;
    want.frame_stride = want.frame_stride ? want.frame_stride : 1u;
    ptc_film::Pending &pend = film->pending;
    // A call continues the recorded run when it differs from it in nothing but
    // the frame count, and that is the run's next one.
    bool continues = pend.params.n_frames != 0u && pend.scene == scene &&
                     std::memcmp(&pend.camera, camera, sizeof(ptc_camera)) == 0;
    if (continues)
    {
        ptc_render_params a = pend.params, b = want;
        const uint32_t next = a.global_frame_count + a.n_frames * a.frame_stride;
        continues = b.global_frame_count == next;
        a.global_frame_count = b.global_frame_count = 0u;
        a.n_frames = b.n_frames = 0u;
        continues = continues && std::memcmp(&a, &b, sizeof a) == 0 &&
                    (uint64_t)pend.params.n_frames + want.n_frames <= 0xffffffffull;
    }
    if (continues)
    {
        pend.params.n_frames += want.n_frames;
    }
    else
    {
        if (const int rc = flush_film(film); rc != PTC_OK)
        {
            return rc;
        }
        pend.scene = scene;
        pend.camera = *camera;
        pend.params = want;
    }
    // as many dispatches as one batch of the queue-based integrators holds
    // (max_paths_in_flight); more would only wait for the next batch
    const uint64_t per_dispatch = (uint64_t)film->width * film->height * want.samples_per_frame;
    const uint64_t budget = want.max_paths_in_flight ? want.max_paths_in_flight : kDefaultPathsInFlight;
    uint64_t bound = std::max<uint64_t>(budget / std::max<uint64_t>(per_dispatch, 1u), 1u);
    // (whole groups of 64 or 16 dispatches, which the primary pass deals to the lanes of a wave)
    bound = bound >= 64u ? bound & ~63ull : (bound >= 16u ? bound & ~15ull : bound);
    if (pend.params.n_frames >= bound)
    {
        return flush_film(film);
    }
    return PTC_OK;
}

int ptc_film_flush(ptc_film *film)
{
    if (!film)
    {
        set_error("ptc_film_flush: film is NULL");
        return PTC_E_INVALID;
    }
    if (use_device(film->ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    return flush_film(film);
}

int ptc_primary_hits(ptc_context *ctx, const ptc_scene *scene, const ptc_camera *camera,
                     uint32_t width, uint32_t height, uint32_t global_frame_count, float tmax,
                     uint32_t *ids_out)
{
    if (!ctx || !scene || !camera || !ids_out || width == 0 || height == 0)
    {
        set_error("ptc_primary_hits: bad argument");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    DFrame F = make_frame(*camera, width, height);
    F.global_frame_count = global_frame_count;
    F.tmax = tmax > 0.0f ? tmax : 10000.0f;
    void *mem = nullptr;
    const size_t n = (size_t)width * height;
    if (const int rs = stage(ctx, n * 4, &mem); rs != PTC_OK)
    {
        return rs;
    }
    uint32_t *d_ids = static_cast<uint32_t *>(mem);
    int rc = launch_primary_hits(ctx, scene, F, d_ids);
    if (rc == PTC_OK)
    {
        hipError_t e = hipMemcpyAsync(ids_out, d_ids, n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess)
        {
            e = hipStreamSynchronize(ctx->stream);
        }
        if (e != hipSuccess)
        {
            set_error(std::string("ptc_primary_hits: ") + hipGetErrorString(e));
            rc = PTC_E_DEVICE;
        }
    }
    return rc;
}

int ptc_scene_triangle_ids(const ptc_scene *scene, const uint32_t *flat, size_t n, uint32_t *instance_out,
                           uint32_t *primitive_out)
{
    if (!scene || (n && (!flat || !instance_out || !primitive_out)))
    {
        set_error("ptc_scene_triangle_ids: NULL argument");
        return PTC_E_INVALID;
    }
    const std::vector<uint32_t> &first = scene->first_flat;
    for (size_t i = 0; i < n; ++i)
    {
        if (flat[i] == PTC_NO_HIT || flat[i] >= scene->info.n_triangles)
        {
            instance_out[i] = primitive_out[i] = PTC_NO_HIT;
            continue;
        }
        // the last instance that starts at or before the index (instances without triangles share a start)
        const size_t k = (size_t)(std::upper_bound(first.begin(), first.end(), flat[i]) - first.begin()) - 1u;
        instance_out[i] = (uint32_t)k;
        primitive_out[i] = flat[i] - first[k];
    }
    return PTC_OK;
}

int ptc_trace_rays_device(ptc_context *ctx, const ptc_scene *scene, const void *rays_dev, size_t n_rays,
                          float tmax, void *hits_dev)
{
    if (!ctx || !scene || (n_rays && (!rays_dev || !hits_dev)))
    {
        set_error("ptc_trace_rays_device: NULL argument");
        return PTC_E_INVALID;
    }
    if (n_rays > 0x7fffffffull * 128)
    {
        set_error("ptc_trace_rays_device: too many rays for one launch");
        return PTC_E_LIMIT;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    return launch_trace(ctx, scene, static_cast<const float *>(rays_dev), n_rays,
                        tmax > 0.0f ? tmax : 10000.0f, static_cast<ptc_hit *>(hits_dev), false);
}

int ptc_trace_rays(ptc_context *ctx, const ptc_scene *scene, const float *rays, size_t n_rays,
                   float tmax, ptc_hit *hits_out, ptc_stats *counters)
{
    if (!ctx || !scene || (n_rays && (!rays || !hits_out)))
    {
        set_error("ptc_trace_rays: NULL argument");
        return PTC_E_INVALID;
    }
    if (n_rays == 0)
    {
        return PTC_OK;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    void *mem = nullptr;
    const size_t rays_bytes = (n_rays * 24 + 255) & ~(size_t)255;
    if (const int rs = stage(ctx, rays_bytes + n_rays * sizeof(ptc_hit), &mem); rs != PTC_OK)
    {
        return rs;
    }
    float *d_rays = static_cast<float *>(mem);
    ptc_hit *d_hits = reinterpret_cast<ptc_hit *>(static_cast<char *>(mem) + rays_bytes);
    hipStream_t st = ctx->stream;
    int rc = PTC_OK;
    DeviceStats before{}, after{};
    hipError_t e = hipMemcpyAsync(d_rays, rays, n_rays * 24, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && counters)
    {
        e = hipMemcpyAsync(&before, ctx->d_stats, sizeof before, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
    {
        rc = launch_trace(ctx, scene, d_rays, n_rays, tmax > 0.0f ? tmax : 10000.0f, d_hits,
                          counters != nullptr);
    }
    if (e == hipSuccess && rc == PTC_OK)
    {
        e = hipMemcpyAsync(hits_out, d_hits, n_rays * sizeof(ptc_hit), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess && rc == PTC_OK && counters)
    {
        e = hipMemcpyAsync(&after, ctx->d_stats, sizeof after, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
    {
        e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess)
    {
        set_error(std::string("ptc_trace_rays: ") + hipGetErrorString(e));
        return PTC_E_DEVICE;
    }
    if (rc == PTC_OK && counters)
    {
        counters->rays += n_rays;
        counters->nodes_visited += after.nodes - before.nodes;
        counters->tris_tested += after.tris - before.tris;
        counters->nodes_wave_uniform += after.uniform_nodes - before.uniform_nodes;
        counters->uniform_fetches += after.uniform_fetches - before.uniform_fetches;
    }
    return rc;
}

int ptc_get_stats(ptc_context *ctx, ptc_stats *out)
{
    if (!ctx || !out)
    {
        set_error("ptc_get_stats: NULL argument");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_context(ctx); rc != PTC_OK)
    {
        return rc;
    }
    DeviceStats s{};
    PTC_HIP_CHECK(hipMemcpyAsync(&s, ctx->d_stats, sizeof s, hipMemcpyDeviceToHost, ctx->stream));
    PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->ev_used > 0)
    {
        float ms = 0.0f;
        const auto &p = ctx->ev_pool[ctx->ev_used - 1];
        if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess)
        {
            ctx->last_render_ms = ms;
        }
    }
    out->rays = s.rays;
    out->paths = s.paths;
    out->nodes_visited = s.nodes;
    out->tris_tested = s.tris;
    out->nodes_wave_uniform = s.uniform_nodes;
    out->uniform_fetches = s.uniform_fetches;
    out->render_ms = ctx->last_render_ms;
    return PTC_OK;
}

int ptc_render_times(ptc_context *ctx, double *ms_out, size_t capacity, size_t *n_out)
{
    if (!ctx || !n_out || (capacity && !ms_out))
    {
        set_error("ptc_render_times: NULL argument");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_context(ctx); rc != PTC_OK)
    {
        return rc;
    }
    PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    size_t n = 0;
    for (size_t i = 0; i < ctx->ev_used && n < capacity; ++i)
    {
        float ms = 0.0f;
        PTC_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
        ms_out[n++] = ms;
    }
    if (ctx->ev_used > 0)
    {
        ctx->last_render_ms = n ? ms_out[n - 1] : ctx->last_render_ms;
    }
    ctx->ev_used = 0;
    *n_out = n;
    return PTC_OK;
}

int ptc_phase_times(ptc_context *ctx, double *ms_out, uint32_t *launches_out)
{
    if (!ctx || !ms_out || !launches_out)
    {
        set_error("ptc_phase_times: NULL argument");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_context(ctx); rc != PTC_OK)
    {
        return rc;
    }
    PTC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (uint32_t k = 0; k < PTC_N_PHASES; ++k)
    {
        ms_out[k] = 0.0;
        launches_out[k] = 0;
    }
    for (size_t i = 0; i < ctx->phase_used; ++i)
    {
        const auto &e = ctx->phase_pool[i];
        float ms = 0.0f;
        PTC_HIP_CHECK(hipEventElapsedTime(&ms, e.start, e.stop));
        if (e.tag < PTC_N_PHASES)
        {
            ms_out[e.tag] += ms;
            launches_out[e.tag] += 1;
        }
    }
    ctx->phase_used = 0;
    return PTC_OK;
}

int ptc_reset_stats(ptc_context *ctx)
{
    if (!ctx)
    {
        set_error("ptc_reset_stats: ctx is NULL");
        return PTC_E_INVALID;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    if (const int rc = flush_context(ctx); rc != PTC_OK)
    {
        return rc;
    }
    PTC_HIP_CHECK(hipMemsetAsync(ctx->d_stats, 0, sizeof(DeviceStats), ctx->stream));
    return PTC_OK;
}

int ptc_write_png(ptc_film *film, const char *file_name)
{
    if (!film || !file_name)
    {
        set_error("ptc_write_png: NULL argument");
        return PTC_E_INVALID;
    }
    std::vector<uint8_t> px((size_t)film->width * film->height * 4);
    const int rc = ptc_film_read_rgba8(film, px.data());
    if (rc != PTC_OK)
    {
        return rc;
    }
    if (!write_png_rgba8(file_name, px.data(), film->width, film->height))
    {
        // message text of write_png(), src/utility.cpp:101-105
        set_error(std::string("Failed to write PNG image to \"") + file_name + "\"");
        return PTC_E_IO;
    }
    return PTC_OK;
}

int ptc_math_eval(ptc_context *ctx, int fn, const float *a, const float *b, float *out, float *out2,
                  size_t n)
{
    if (!ctx || !a || !b || !out || !out2)
    {
        set_error("ptc_math_eval: NULL argument");
        return PTC_E_INVALID;
    }
    if (n == 0)
    {
        return PTC_OK;
    }
    if (use_device(ctx) != PTC_OK)
    {
        return PTC_E_DEVICE;
    }
    void *mem = nullptr;
    if (const int rs = stage(ctx, n * 16, &mem); rs != PTC_OK)
    {
        return rs;
    }
    float *d = static_cast<float *>(mem);
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(d, a, n * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
    {
        e = hipMemcpyAsync(d + n, b, n * 4, hipMemcpyHostToDevice, st);
    }
    int rc = PTC_OK;
    if (e == hipSuccess)
    {
        rc = launch_math_eval(ctx, fn, d, d + n, d + 2 * n, d + 3 * n, n);
    }
    if (e == hipSuccess && rc == PTC_OK)
    {
        e = hipMemcpyAsync(out, d + 2 * n, n * 4, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess && rc == PTC_OK)
    {
        e = hipMemcpyAsync(out2, d + 3 * n, n * 4, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess)
    {
        e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess)
    {
        set_error(std::string("ptc_math_eval: ") + hipGetErrorString(e));
        return PTC_E_DEVICE;
    }
    return rc;
}

} // extern "C"
