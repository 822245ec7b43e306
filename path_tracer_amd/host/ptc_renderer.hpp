// C++ host-side mirror of the reference renderer interface, over libptcore's
// C ABI (include/ptc.h). A host that today calls the reference's
//   create_context / create_render_resources / draw_frame / reset_render /
//   write_to_png                       (src/renderer.hpp:256-278)
//   create_camera / camera_set_distance (src/camera.hpp:61-67)
// can include this header instead: same names, same argument meaning, same
// counters on the resources struct (samples_to_render, sample_count,
// samples_per_frame; src/renderer.hpp:251-253), exceptions on failure like the
// reference's Vulkan-Hpp calls, and write_to_png returning "" or a message.
// The scene crosses as plain arrays (Scene_arrays) because Assimp's aiScene is
// not a dependency here; the adapter from aiScene copies mesh 0's vertices,
// normals and face indices into one Scene_arrays::Mesh and lists the instances.
#ifndef PTC_RENDERER_HPP
#define PTC_RENDERER_HPP

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ptc.h"

namespace ptc_host
{

using vec3 = ptc_vec3;
using Camera = ptc_camera; // byte-identical to the reference's Camera

namespace detail
{
inline vec3 sub(const vec3 &a, const vec3 &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 mul(const vec3 &a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3 &a, const vec3 &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length(const vec3 &a) { return std::sqrt(dot(a, a)); }
inline vec3 unit(const vec3 &a) { return mul(a, 1.0f / length(a)); }
inline void check(int rc, const char *what)
{
    if (rc != PTC_OK)
    {
        throw std::runtime_error(std::string(what) + ": " + ptc_last_error());
    }
}
} // namespace detail

// create_camera (src/camera.cpp:65-90): look-at frame with x right, y down,
// z forward; focus at the target, pinhole aperture.
inline Camera create_camera(const vec3 &position, const vec3 &target, float sensor_distance,
                            float sensor_half_width, float sensor_half_height)
{
    using namespace detail;
    const vec3 up {0.0f, 1.0f, 0.0f};
    const vec3 to_target = sub(target, position);
    Camera c {};
    c.distance = length(to_target);
    c.position = position;
    c.target = target;
    c.direction_z = unit(to_target);
    c.direction_x = unit(cross(c.direction_z, up));
    c.direction_y = unit(cross(c.direction_z, c.direction_x));
    c.sensor_distance = sensor_distance;
    c.sensor_half_width = sensor_half_width;
    c.sensor_half_height = sensor_half_height;
    c.focus_distance = c.distance;
    c.aperture_radius = 0.0f;
    return c;
}

// camera_set_distance (src/camera.cpp:92-96)
inline void camera_set_distance(Camera &camera, float distance)
{
    using namespace detail;
    camera.distance = distance;
    camera.position = sub(camera.target, mul(unit(camera.direction_z), distance));
}

// What create_render_resources() reads out of aiScene (src/renderer.cpp:1818-1838),
// generalised from "mesh 0, four hard-coded instances" to a list of meshes and
// instances with a material kind each.
struct Scene_arrays
{
    struct Mesh
    {
        std::vector<float> positions; // 3 per vertex
        std::vector<float> normals;   // 3 per vertex
        std::vector<std::uint32_t> indices; // 3 per triangle
    };
    std::vector<Mesh> meshes;
    std::vector<ptc_instance_desc> instances;
    std::vector<float> envmap_rgba; // empty = two-tone sky
    std::uint32_t envmap_width {0}, envmap_height {0};
};

struct Render_context
{
    ptc_context *handle {nullptr};
    std::uint32_t global_frame_count {0}; // Vulkan_context::global_frame_count

    Render_context() = default;
    Render_context(const Render_context &) = delete;
    Render_context &operator=(const Render_context &) = delete;
    Render_context(Render_context &&o) noexcept { *this = std::move(o); }
    Render_context &operator=(Render_context &&o) noexcept
    {
        std::swap(handle, o.handle);
        std::swap(global_frame_count, o.global_frame_count);
        return *this;
    }
    ~Render_context() { ptc_context_destroy(handle); }
};

struct Render_resources
{
    ptc_scene *scene {nullptr};
    ptc_film *film {nullptr};
    std::uint32_t width {0}, height {0};
    std::uint32_t samples_to_render {1000}; // src/renderer.cpp:1950-1952
    std::uint32_t sample_count {0};
    std::uint32_t samples_per_frame {1};
    std::uint32_t max_depth {32};            // literal of shader.rgen:65
    std::uint32_t integrator {PTC_INTEGRATOR_AUTO};
    std::uint32_t tonemap {PTC_TONEMAP_PBR_NEUTRAL}; // shader.rgen:218-221
    std::uint32_t debug_view {0};            // 1 = bounce counts / 16, the "#else" branch of shader.rgen:195-199

    Render_resources() = default;
    Render_resources(const Render_resources &) = delete;
    Render_resources &operator=(const Render_resources &) = delete;
    Render_resources(Render_resources &&o) noexcept { *this = std::move(o); }
    Render_resources &operator=(Render_resources &&o) noexcept
    {
        std::swap(scene, o.scene);
        std::swap(film, o.film);
        std::swap(width, o.width);
        std::swap(height, o.height);
        std::swap(samples_to_render, o.samples_to_render);
        std::swap(sample_count, o.sample_count);
        std::swap(samples_per_frame, o.samples_per_frame);
        std::swap(max_depth, o.max_depth);
        std::swap(integrator, o.integrator);
        std::swap(tonemap, o.tonemap);
        std::swap(debug_view, o.debug_view);
        return *this;
    }
    ~Render_resources()
    {
        ptc_film_destroy(film);
        ptc_scene_destroy(scene);
    }
};

// The import half of open_scene() (src/application.cpp:150-200): the meshes of
// a Wavefront OBJ or Stanford PLY file after the post-processing the reference
// configures Assimp with. Instances and the sky are left to the caller. Throws
// with the importer's message ("Scene has no meshes", ...) on failure.
[[nodiscard]] inline Scene_arrays import_scene(const char *file_name, const ptc_import_options *options = nullptr)
{
    ptc_mesh_file *file = nullptr;
    detail::check(ptc_mesh_file_open(file_name, options, &file), "open_scene");
    Scene_arrays scene;
    for (std::uint32_t i = 0; i < ptc_mesh_file_mesh_count(file); ++i)
    {
        ptc_mesh_desc d {};
        ptc_mesh_file_get_mesh(file, i, &d);
        Scene_arrays::Mesh m;
        m.positions.assign(d.positions, d.positions + 3 * static_cast<std::size_t>(d.n_vertices));
        m.normals.assign(d.normals, d.normals + 3 * static_cast<std::size_t>(d.n_vertices));
        m.indices.assign(d.indices, d.indices + 3 * static_cast<std::size_t>(d.n_triangles));
        scene.meshes.push_back(std::move(m));
    }
    ptc_mesh_file_close(file);
    return scene;
}

// Image<float> of src/utility.hpp:16-22 and read_hdr_image (src/utility.cpp:75-92):
// RGBA32F, rows top to bottom; throws with the decoder's reason on failure.
struct Image_f32
{
    std::vector<float> data;
    std::uint32_t width {0}, height {0};
};

[[nodiscard]] inline Image_f32 read_hdr_image(const char *file_name)
{
    Image_f32 image;
    detail::check(ptc_read_hdr(file_name, &image.width, &image.height, nullptr, 0), "read_hdr_image");
    image.data.resize(static_cast<std::size_t>(4) * image.width * image.height);
    detail::check(ptc_read_hdr(file_name, &image.width, &image.height, image.data.data(), image.data.size()),
                  "read_hdr_image");
    return image;
}

// create_context (src/renderer.hpp:256); the window argument has no meaning here.
[[nodiscard]] inline Render_context create_context(int device_ordinal = 0)
{
    Render_context ctx;
    detail::check(ptc_context_create(device_ordinal, &ctx.handle), "create_context");
    return ctx;
}

// create_render_resources (src/renderer.cpp:1738-1955). build_options->two_level
// keeps one tree per mesh plus an instance list, the shape of the reference's
// acceleration structures (src/renderer.cpp:1044-1281), instead of one tree
// over all instanced triangles.
[[nodiscard]] inline Render_resources create_render_resources(const Render_context &context,
                                                              std::uint32_t render_width,
                                                              std::uint32_t render_height,
                                                              const Scene_arrays &scene,
                                                              const ptc_build_options *build_options = nullptr)
{
    std::vector<ptc_mesh_desc> meshes;
    for (const auto &m : scene.meshes)
    {
        if (m.normals.size() != m.positions.size())
        {
            throw std::runtime_error("create_render_resources: mesh without per-vertex normals");
        }
        meshes.push_back({m.positions.data(), m.normals.data(), m.indices.data(),
                          static_cast<std::uint32_t>(m.positions.size() / 3),
                          static_cast<std::uint32_t>(m.indices.size() / 3)});
    }
    const ptc_envmap_desc env {scene.envmap_rgba.empty() ? nullptr : scene.envmap_rgba.data(), scene.envmap_width,
                               scene.envmap_height};
    Render_resources rr;
    rr.width = render_width;
    rr.height = render_height;
    detail::check(ptc_scene_create(context.handle, meshes.data(), static_cast<std::uint32_t>(meshes.size()),
                                   scene.instances.data(), static_cast<std::uint32_t>(scene.instances.size()),
                                   &env, build_options, &rr.scene),
                  "create_render_resources");
    detail::check(ptc_film_create(context.handle, render_width, render_height, nullptr, nullptr, &rr.film),
                  "create_render_resources");
    return rr;
}

// The traced part of draw_frame (src/renderer.cpp:2015-2130, :2224). Like the
// reference's, which submits its command buffer and returns with up to
// frames_in_flight frames outstanding (src/renderer.hpp:208,
// src/renderer.cpp:2170-2172), the call returns once its dispatch is recorded:
// consecutive calls with one camera are launched together (ptc_render_deferred),
// at the latest when something reads the film (write_to_png, a read-back).
inline void draw_frame(Render_context &context, Render_resources &rr, const Camera &camera)
{
    if (rr.sample_count == 0)
    {
        detail::check(ptc_film_reset(rr.film), "draw_frame");
    }
    if (rr.sample_count < rr.samples_to_render)
    {
        const std::uint32_t samples_this_frame =
            std::min(rr.samples_to_render - rr.sample_count, rr.samples_per_frame);
        ptc_render_params p {};
        p.global_frame_count = context.global_frame_count;
        p.samples_per_frame = samples_this_frame;
        p.n_frames = 1;
        p.max_depth = rr.max_depth;
        p.integrator = rr.integrator;
        p.tonemap = rr.tonemap;
        p.debug_view = rr.debug_view;
        p.accumulate = PTC_ACCUM_RUNNING_MEAN;
        detail::check(ptc_film_set_sample_count(rr.film, rr.sample_count), "draw_frame");
        if (const int rc = ptc_render_deferred(context.handle, rr.scene, rr.film, &camera, &p); rc != PTC_OK)
        {
            // the call may have launched earlier dispatches and failed in them: the film knows what it holds
            rr.sample_count = ptc_film_sample_count(rr.film);
            detail::check(rc, "draw_frame");
        }
        rr.sample_count += samples_this_frame;
    }
    ++context.global_frame_count;
}

// n consecutive draw_frame() calls with one camera, as one call into the
// library: the same film and the same counters afterwards (dispatch k uses
// global_frame_count + k and continues the sample count), but the dispatches
// that render a full samples_per_frame are in flight together, which is what
// the wavefront integrator needs to reach its rate. The last, partial dispatch
// and calls beyond samples_to_render go through draw_frame() itself.
inline void draw_frames(Render_context &context, Render_resources &rr, const Camera &camera, std::uint32_t n)
{
    const std::uint32_t spf = std::max<std::uint32_t>(rr.samples_per_frame, 1u);
    const std::uint32_t remaining = rr.sample_count < rr.samples_to_render ? rr.samples_to_render - rr.sample_count : 0u;
    const std::uint32_t n_full = std::min(n, remaining / spf);
    if (n_full > 1u)
    {
        if (rr.sample_count == 0)
        {
            detail::check(ptc_film_reset(rr.film), "draw_frames");
        }
        ptc_render_params p {};
        p.global_frame_count = context.global_frame_count;
        p.samples_per_frame = spf;
        p.n_frames = n_full;
        p.max_depth = rr.max_depth;
        p.integrator = rr.integrator;
        p.tonemap = rr.tonemap;
        p.debug_view = rr.debug_view;
        p.accumulate = PTC_ACCUM_RUNNING_MEAN;
        detail::check(ptc_film_set_sample_count(rr.film, rr.sample_count), "draw_frames");
        if (const int rc = ptc_render(context.handle, rr.scene, rr.film, &camera, &p); rc != PTC_OK)
        {
            rr.sample_count = ptc_film_sample_count(rr.film);
            detail::check(rc, "draw_frames");
        }
        rr.sample_count += n_full * spf;
        context.global_frame_count += n_full;
        n -= n_full;
    }
    for (; n != 0u; --n)
    {
        draw_frame(context, rr, camera);
    }
}

// reset_render (src/renderer.cpp:2236-2239)
inline void reset_render(Render_resources &rr)
{
    rr.sample_count = 0;
}

// write_to_png (src/renderer.cpp:2241-2327): "" on success, else the message.
[[nodiscard]] inline std::string write_to_png(const Render_context &, const Render_resources &rr,
                                              const char *file_name)
{
    return ptc_write_png(rr.film, file_name) == PTC_OK ? std::string {} : std::string {ptc_last_error()};
}

} // namespace ptc_host

#endif // PTC_RENDERER_HPP
