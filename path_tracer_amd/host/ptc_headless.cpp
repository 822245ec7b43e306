// Headless stand-in for the reference's entry point `path_tracer [<file>]`
// (src/main.cpp:8-36, src/application.cpp:933-997): loads a scene, renders it
// progressively with draw_frame() until samples_to_render is reached, and
// writes the PNG that the reference's "save" dialog would write.
//
//   ptc_headless [<file.obj|.ply>] [--width W] [--height H] [--spp N] [--spf N]
//                [--depth D] [--megakernel] [--env sky.hdr] [--out image.png]
//                [--device N]
//
// --env loads a Radiance .hdr file as the environment map, as the reference
// does with its hard-coded "powerplant.hdr" (src/renderer.cpp:1842); without it
// the sky is the two-tone one kept under "#if 0" in shader.rmiss:13-18.
//
// Without a file it renders a built-in box scene. With a Wavefront OBJ or
// Stanford PLY file it does what the reference does at HEAD: imports it with
// the post-processing of src/application.cpp:150-189 (normalised to [-1, 1],
// smooth normals where the file has none, identical vertices joined), takes
// mesh 0 and instantiates it four times with the transforms and material kinds
// of src/renderer.cpp:1147-1174.
#include "ptc_renderer.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace ptc_host;

namespace
{

void add_quad(Scene_arrays::Mesh &m, const float q[4][3])
{
    const auto base = static_cast<std::uint32_t>(m.positions.size() / 3);
    const vec3 a {q[0][0], q[0][1], q[0][2]}, b {q[1][0], q[1][1], q[1][2]}, c {q[2][0], q[2][1], q[2][2]};
    const vec3 n = detail::unit(detail::cross(detail::sub(b, a), detail::sub(c, a)));
    for (int k = 0; k < 4; ++k)
    {
        m.positions.insert(m.positions.end(), {q[k][0], q[k][1], q[k][2]});
        m.normals.insert(m.normals.end(), {n.x, n.y, n.z});
    }
    m.indices.insert(m.indices.end(), {base, base + 1, base + 2, base, base + 2, base + 3});
}

ptc_instance_desc instance(std::uint32_t mesh, std::uint32_t kind, float s = 1.0f, float tx = 0.0f,
                           float ty = 0.0f, float tz = 0.0f)
{
    ptc_instance_desc i {};
    const float t[12] = {s, 0, 0, tx, 0, s, 0, ty, 0, 0, s, tz};
    std::memcpy(i.transform, t, sizeof t);
    i.mesh = mesh;
    i.kind = kind;
    return i;
}

// A small closed room: five diffuse walls, an emissive panel, one mirror and one
// glass block (unit cube corners scaled), opening towards +z.
Scene_arrays builtin_box()
{
    Scene_arrays s;
    Scene_arrays::Mesh walls, light, cube;
    const float w[5][4][3] = {
        {{-1, -1, 1}, {1, -1, 1}, {1, -1, -1}, {-1, -1, -1}},  // floor
        {{-1, 1, -1}, {1, 1, -1}, {1, 1, 1}, {-1, 1, 1}},      // ceiling
        {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}},  // back
        {{-1, -1, 1}, {-1, -1, -1}, {-1, 1, -1}, {-1, 1, 1}},  // left
        {{1, -1, -1}, {1, -1, 1}, {1, 1, 1}, {1, 1, -1}},      // right
    };
    for (const auto &q : w)
    {
        add_quad(walls, q);
    }
    const float l[4][3] = {{-0.3f, 0.995f, -0.3f}, {0.3f, 0.995f, -0.3f}, {0.3f, 0.995f, 0.3f}, {-0.3f, 0.995f, 0.3f}};
    add_quad(light, l);
    const float c[6][4][3] = {
        {{-1, -1, -1}, {-1, 1, -1}, {1, 1, -1}, {1, -1, -1}}, {{-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}},
        {{-1, -1, -1}, {-1, -1, 1}, {-1, 1, 1}, {-1, 1, -1}}, {{1, -1, -1}, {1, 1, -1}, {1, 1, 1}, {1, -1, 1}},
        {{-1, 1, -1}, {-1, 1, 1}, {1, 1, 1}, {1, 1, -1}},     {{-1, -1, -1}, {1, -1, -1}, {1, -1, 1}, {-1, -1, 1}},
    };
    for (const auto &q : c)
    {
        add_quad(cube, q);
    }
    s.meshes = {walls, light, cube};
    s.instances = {instance(0, PTC_KIND_DIFFUSE), instance(1, PTC_KIND_EMISSIVE),
                   instance(2, PTC_KIND_SPECULAR, 0.3f, -0.45f, -0.7f, -0.3f),
                   instance(2, PTC_KIND_DIELECTRIC, 0.25f, 0.45f, -0.75f, 0.2f)};
    return s;
}

} // namespace

int main(int argc, char **argv)
{
    try
    {
        const char *file = nullptr;
        std::string out = "render.png", env;
        std::uint32_t width = 640, height = 480, spp = 64, spf = 1, depth = 32; // src/application.cpp:206-207
        std::uint32_t preview_every = 0, batch = 64;
        int device = 0;
        bool megakernel = false, two_level = false, bounce_view = false, fast_build = false;
        for (int i = 1; i < argc; ++i)
        {
            const std::string a = argv[i];
            auto next = [&]() -> const char * {
                if (i + 1 >= argc)
                {
                    throw std::runtime_error("missing value after " + a);
                }
                return argv[++i];
            };
            if (a == "--width") width = std::strtoul(next(), nullptr, 10);
            else if (a == "--height") height = std::strtoul(next(), nullptr, 10);
            else if (a == "--spp") spp = std::strtoul(next(), nullptr, 10);
            else if (a == "--spf") spf = std::strtoul(next(), nullptr, 10);
            else if (a == "--depth") depth = std::strtoul(next(), nullptr, 10);
            else if (a == "--out") out = next();
            else if (a == "--env") env = next();
            else if (a == "--device") device = std::atoi(next());
            else if (a == "--megakernel") megakernel = true;
            else if (a == "--two-level") two_level = true;
            else if (a == "--fast-build") fast_build = true;
            else if (a == "--bounce-view") bounce_view = true;
            else if (a == "--preview-every") preview_every = std::strtoul(next(), nullptr, 10);
            else if (a == "--batch") batch = std::max(1ul, std::strtoul(next(), nullptr, 10));
            else if (a[0] == '-')
            {
                std::cerr << "Usage: " << argv[0] << " [<file.obj|.ply>] [--width W] [--height H] [--spp N] [--spf N] "
                          << "[--depth D] [--megakernel] [--two-level] [--fast-build] [--bounce-view] [--preview-every N] [--batch N] [--env sky.hdr] [--out image.png] [--device N]\n";
                return EXIT_FAILURE;
            }
            else file = argv[i];
        }

        Scene_arrays scene;
        if (file)
        {
            // the reference uploads mesh 0 only (src/renderer.cpp:1818-1838)
            scene.meshes.push_back(std::move(import_scene(file).meshes.front()));
            // the four instances of src/renderer.cpp:1147-1159, kind = i % 4
            scene.instances = {instance(0, 0, 1.0f, -1.0f), instance(0, 1, 0.7f, 1.0f), instance(0, 2, 0.2f),
                               instance(0, 3, 0.8f, 0.0f, 0.0f, 0.8f)};
        }
        else
        {
            scene = builtin_box();
        }

        if (!env.empty())
        {
            Image_f32 sky = read_hdr_image(env.c_str());
            scene.envmap_width = sky.width;
            scene.envmap_height = sky.height;
            scene.envmap_rgba = std::move(sky.data);
        }

        Render_context context = create_context(device);
        ptc_build_options build {};
        build.two_level = two_level ? 1u : 0u;
        build.optimize_rounds = fast_build ? PTC_OPTIMIZE_NONE : 0u; // (0: the library's default number of reinsertion rounds)
        Render_resources rr = create_render_resources(context, width, height, scene, &build);
        rr.samples_to_render = spp;
        rr.samples_per_frame = spf;
        rr.max_depth = depth;
        rr.integrator = megakernel ? PTC_INTEGRATOR_MEGAKERNEL : PTC_INTEGRATOR_AUTO;
        rr.debug_view = bounce_view ? 1u : 0u; // the view under "#else" at src/shaders/shader.rgen:195-199

        // camera of open_scene(), src/application.cpp:209-222
        const float vertical_fov = 45.0f / 180.0f * 3.14159265358979323846f;
        const float shh = std::tan(vertical_fov * 0.5f);
        const Camera camera = create_camera({0.0f, 0.0f, 3.5f}, {0.0f, 0.0f, 0.0f}, 1.0f,
                                            static_cast<float>(width) / static_cast<float>(height) * shh, shh);

        const auto t0 = std::chrono::steady_clock::now();
        // The reference shows the target image after every draw_frame()
        // (src/application.cpp:690-694, :978); without a window the picture so far
        // is written over the output file every preview_every samples instead.
        std::uint32_t next_preview = preview_every;
        while (rr.sample_count < rr.samples_to_render)
        {
            // up to `batch` draw_frame() calls at a time (the same picture as one by
            // one), stopping at the next preview
            std::uint32_t calls = batch;
            if (preview_every != 0 && next_preview > rr.sample_count)
            {
                const std::uint32_t spf_now = std::max<std::uint32_t>(rr.samples_per_frame, 1u);
                calls = std::min(calls, (next_preview - rr.sample_count + spf_now - 1u) / spf_now);
            }
            draw_frames(context, rr, camera, std::max<std::uint32_t>(calls, 1u));
            if (preview_every != 0 && rr.sample_count >= next_preview && rr.sample_count < rr.samples_to_render)
            {
                const std::string msg = write_to_png(context, rr, out.c_str());
                if (!msg.empty())
                {
                    std::cerr << msg << '\n';
                    return EXIT_FAILURE;
                }
                std::cout << "preview at " << rr.sample_count << " samples -> " << out << '\n';
                next_preview = rr.sample_count + preview_every;
            }
        }
        ptc_synchronize(context.handle);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ptc_stats st {};
        ptc_get_stats(context.handle, &st);
        ptc_scene_info info {};
        ptc_scene_get_info(rr.scene, &info);
        const std::string err = write_to_png(context, rr, out.c_str());
        if (!err.empty())
        {
            std::cerr << err << '\n';
            return EXIT_FAILURE;
        }
        std::cout << "triangles " << info.n_triangles << " bvh_nodes " << info.n_nodes << " build_ms " << info.build_ms
                  << " samples " << rr.sample_count << " rays " << st.rays << " ms " << ms << " Mrays/s "
                  << static_cast<double>(st.rays) / ms / 1e3 << " -> " << out << '\n';
        return EXIT_SUCCESS;
    }
    catch (const std::exception &e)
    {
        std::cerr << "Exception thrown: " << e.what() << '\n';
        return EXIT_FAILURE;
    }
}
