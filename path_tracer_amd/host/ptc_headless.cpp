// Headless stand-in for the reference's entry point `path_tracer [<file>]`
// (src/main.cpp:8-36, src/application.cpp:933-997): loads a scene, renders it
// progressively with draw_frame() until samples_to_render is reached, and
// writes the PNG that the reference's "save" dialog would write.
//
//   ptc_headless [<file.obj>] [--width W] [--height H] [--spp N] [--spf N]
//                [--depth D] [--megakernel] [--out image.png] [--device N]
//
// Without a file it renders a built-in Cornell box. With a Wavefront OBJ file it
// does what the reference does at HEAD: takes the first mesh, normalises it to
// [-1, 1] (AI_CONFIG_PP_PTV_NORMALIZE, src/application.cpp:161), generates
// smooth vertex normals when the file has none, and instantiates it four times
// with the transforms and material kinds of src/renderer.cpp:1147-1174.
#include "ptc_renderer.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

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

// Minimal Wavefront OBJ reader: v, vn (optional) and f records; polygons are
// fan-triangulated; all groups are merged into one mesh ("mesh 0").
Scene_arrays::Mesh load_obj(const char *file_name)
{
    std::ifstream in(file_name);
    if (!in)
    {
        throw std::runtime_error(std::string("cannot open ") + file_name);
    }
    Scene_arrays::Mesh m;
    std::string line;
    while (std::getline(in, line))
    {
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v")
        {
            float x, y, z;
            ls >> x >> y >> z;
            m.positions.insert(m.positions.end(), {x, y, z});
        }
        else if (tag == "f")
        {
            std::vector<std::uint32_t> idx;
            std::string tok;
            const long nv = static_cast<long>(m.positions.size() / 3);
            while (ls >> tok)
            {
                long i = std::strtol(tok.c_str(), nullptr, 10);
                i = i < 0 ? nv + i : i - 1;
                if (i < 0 || i >= nv)
                {
                    throw std::runtime_error("OBJ face index out of range");
                }
                idx.push_back(static_cast<std::uint32_t>(i));
            }
            for (std::size_t k = 2; k < idx.size(); ++k)
            {
                m.indices.insert(m.indices.end(), {idx[0], idx[k - 1], idx[k]});
            }
        }
    }
    if (m.indices.empty())
    {
        throw std::runtime_error("Scene has no meshes"); // message of src/application.cpp:200
    }
    // normalise to [-1, 1] about the bounding-box centre
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    for (std::size_t v = 0; v < m.positions.size(); v += 3)
    {
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = std::min(lo[k], m.positions[v + k]);
            hi[k] = std::max(hi[k], m.positions[v + k]);
        }
    }
    const float half = 0.5f * std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]});
    for (std::size_t v = 0; v < m.positions.size(); v += 3)
    {
        for (int k = 0; k < 3; ++k)
        {
            m.positions[v + k] = (m.positions[v + k] - 0.5f * (lo[k] + hi[k])) / half;
        }
    }
    // area-weighted smooth vertex normals
    m.normals.assign(m.positions.size(), 0.0f);
    for (std::size_t t = 0; t < m.indices.size(); t += 3)
    {
        const std::uint32_t i[3] = {m.indices[t], m.indices[t + 1], m.indices[t + 2]};
        const vec3 a {m.positions[3 * i[0]], m.positions[3 * i[0] + 1], m.positions[3 * i[0] + 2]};
        const vec3 b {m.positions[3 * i[1]], m.positions[3 * i[1] + 1], m.positions[3 * i[1] + 2]};
        const vec3 c {m.positions[3 * i[2]], m.positions[3 * i[2] + 1], m.positions[3 * i[2] + 2]};
        const vec3 n = detail::cross(detail::sub(b, a), detail::sub(c, a));
        for (int k = 0; k < 3; ++k)
        {
            m.normals[3 * i[k]] += n.x;
            m.normals[3 * i[k] + 1] += n.y;
            m.normals[3 * i[k] + 2] += n.z;
        }
    }
    for (std::size_t v = 0; v < m.normals.size(); v += 3)
    {
        const vec3 n {m.normals[v], m.normals[v + 1], m.normals[v + 2]};
        const float len = detail::length(n);
        const vec3 u = len > 0.0f ? detail::mul(n, 1.0f / len) : vec3 {0.0f, 1.0f, 0.0f};
        m.normals[v] = u.x;
        m.normals[v + 1] = u.y;
        m.normals[v + 2] = u.z;
    }
    return m;
}

} // namespace

int main(int argc, char **argv)
{
    try
    {
        const char *file = nullptr;
        std::string out = "render.png";
        std::uint32_t width = 640, height = 480, spp = 64, spf = 1, depth = 32; // src/application.cpp:206-207
        int device = 0;
        bool megakernel = false;
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
            else if (a == "--device") device = std::atoi(next());
            else if (a == "--megakernel") megakernel = true;
            else if (a[0] == '-')
            {
                std::cerr << "Usage: " << argv[0] << " [<file.obj>] [--width W] [--height H] [--spp N] [--spf N] "
                          << "[--depth D] [--megakernel] [--out image.png] [--device N]\n";
                return EXIT_FAILURE;
            }
            else file = argv[i];
        }

        Scene_arrays scene;
        if (file)
        {
            scene.meshes.push_back(load_obj(file));
            // the four instances of src/renderer.cpp:1147-1159, kind = i % 4
            scene.instances = {instance(0, 0, 1.0f, -1.0f), instance(0, 1, 0.7f, 1.0f), instance(0, 2, 0.2f),
                               instance(0, 3, 0.8f, 0.0f, 0.0f, 0.8f)};
        }
        else
        {
            scene = builtin_box();
        }

        Render_context context = create_context(device);
        Render_resources rr = create_render_resources(context, width, height, scene);
        rr.samples_to_render = spp;
        rr.samples_per_frame = spf;
        rr.max_depth = depth;
        rr.integrator = megakernel ? PTC_INTEGRATOR_MEGAKERNEL : PTC_INTEGRATOR_WAVEFRONT;

        // camera of open_scene(), src/application.cpp:209-222
        const float vertical_fov = 45.0f / 180.0f * 3.14159265358979323846f;
        const float shh = std::tan(vertical_fov * 0.5f);
        const Camera camera = create_camera({0.0f, 0.0f, 3.5f}, {0.0f, 0.0f, 0.0f}, 1.0f,
                                            static_cast<float>(width) / static_cast<float>(height) * shh, shh);

        const auto t0 = std::chrono::steady_clock::now();
        while (rr.sample_count < rr.samples_to_render)
        {
            draw_frame(context, rr, camera);
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
