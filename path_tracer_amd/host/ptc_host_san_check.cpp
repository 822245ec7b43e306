// Drives the host-only code of libptcore (scene import, Radiance .hdr decode)
// in a build with AddressSanitizer and UndefinedBehaviorSanitizer
// (make SAN=1 check in csrc/), the counterpart of the reference's sanitized
// Debug build (CMakeLists.txt:214-219). Well-formed files must import; every
// truncated or corrupted variant must either import or fail with an error
// code, never trip a sanitizer. No GPU is touched.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptc.h"

namespace
{
std::string g_dir;

std::string write_file(const char *name, const std::string &data)
{
    const std::string path = g_dir + "/" + name;
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (f == nullptr)
    {
        std::perror(path.c_str());
        std::exit(2);
    }
    std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
    return path;
}

// returns the triangle count over all meshes, or -1 when the importer refused the file
long import_count(const std::string &path, const ptc_import_options *opts = nullptr)
{
    ptc_mesh_file *file = nullptr;
    if (ptc_mesh_file_open(path.c_str(), opts, &file) != PTC_OK)
    {
        return -1;
    }
    long n = 0;
    for (uint32_t i = 0; i < ptc_mesh_file_mesh_count(file); ++i)
    {
        ptc_mesh_desc d {};
        if (ptc_mesh_file_get_mesh(file, i, &d) != PTC_OK)
        {
            std::exit(3);
        }
        // touch every element the descriptor promises
        double sum = 0.0;
        for (uint32_t t = 0; t < 3 * d.n_triangles; ++t)
        {
            const uint32_t v = d.indices[t];
            if (v >= d.n_vertices)
            {
                std::fprintf(stderr, "index out of range\n");
                std::exit(3);
            }
            sum += d.positions[3 * v] + d.normals[3 * v + 2];
        }
        n += d.n_triangles + (sum != sum ? 0 : 0);
    }
    ptc_mesh_file_close(file);
    return n;
}

std::string make_obj()
{
    std::string s = "# grid with an L-shaped polygon\no a\n";
    for (int i = 0; i < 6; ++i)
    {
        for (int j = 0; j < 6; ++j)
        {
            s += "v " + std::to_string(i) + " " + std::to_string(j) + " " + std::to_string((i * j) % 3 * 0.1) + "\n";
        }
    }
    for (int i = 0; i < 5; ++i)
    {
        for (int j = 0; j < 5; ++j)
        {
            const int a = i * 6 + j + 1;
            s += "f " + std::to_string(a) + " " + std::to_string(a + 6) + " " + std::to_string(a + 7) + " " + std::to_string(a + 1) + "\n";
        }
    }
    s += "g b\nv 10 0 0\nv 13 0 0\nv 13 1 0\nv 11 1 0\nv 11 3 0\nv 10 3 0\nf -6 -5 -4 -3 -2 -1\n";
    return s;
}

std::string make_ply(bool big_endian)
{
    const int n = 5;
    std::string s = std::string("ply\nformat ") + (big_endian ? "binary_big_endian" : "binary_little_endian") +
                    " 1.0\nelement vertex " + std::to_string(n * n) +
                    "\nproperty float x\nproperty float y\nproperty float z\nelement face " +
                    std::to_string((n - 1) * (n - 1)) + "\nproperty list uchar int vertex_indices\nend_header\n";
    auto put32 = [&](uint32_t v) {
        char b[4];
        for (int k = 0; k < 4; ++k)
        {
            b[big_endian ? 3 - k : k] = (char)(v >> (8 * k));
        }
        s.append(b, 4);
    };
    for (int i = 0; i < n; ++i)
    {
        for (int j = 0; j < n; ++j)
        {
            const float p[3] = {(float)i, (float)j, 0.25f * (float)((i + j) % 2)};
            for (float f : p)
            {
                uint32_t u;
                std::memcpy(&u, &f, 4);
                put32(u);
            }
        }
    }
    for (int i = 0; i + 1 < n; ++i)
    {
        for (int j = 0; j + 1 < n; ++j)
        {
            s.push_back(4);
            const int a = i * n + j;
            put32((uint32_t)a);
            put32((uint32_t)(a + n));
            put32((uint32_t)(a + n + 1));
            put32((uint32_t)(a + 1));
        }
    }
    return s;
}

std::string make_hdr(bool rle)
{
    const int w = 16, h = 4;
    std::string s = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " + std::to_string(h) + " +X " + std::to_string(w) + "\n";
    for (int y = 0; y < h; ++y)
    {
        if (!rle)
        {
            for (int x = 0; x < w; ++x)
            {
                const char px[4] = {(char)(x * 8), (char)(y * 40), (char)77, (char)(128 + (x & 3))};
                s.append(px, 4);
            }
            continue;
        }
        const char head[4] = {2, 2, (char)(w >> 8), (char)(w & 0xff)};
        s.append(head, 4);
        for (int c = 0; c < 4; ++c)
        {
            s.push_back((char)(128 + 6)); // run of 6
            s.push_back((char)(10 * c + y));
            s.push_back((char)10); // 10 literals
            for (int x = 0; x < 10; ++x)
            {
                s.push_back((char)(x * 20 + c));
            }
        }
    }
    return s;
}

long read_hdr_pixels(const std::string &path)
{
    uint32_t w = 0, h = 0;
    if (ptc_read_hdr(path.c_str(), &w, &h, nullptr, 0) != PTC_OK)
    {
        return -1;
    }
    std::vector<float> px((size_t)4 * w * h);
    if (ptc_read_hdr(path.c_str(), &w, &h, px.data(), px.size()) != PTC_OK)
    {
        return -1;
    }
    return (long)w * h;
}
} // namespace

int main(int argc, char **argv)
{
    g_dir = argc > 1 ? argv[1] : "/tmp";
    int failures = 0;
    auto expect = [&](bool ok, const char *what) {
        if (!ok)
        {
            std::fprintf(stderr, "ptc_host_san_check: %s\n", what);
            ++failures;
        }
    };
    const std::string obj = make_obj();
    expect(import_count(write_file("a.obj", obj)) == 50 + 4, "OBJ: 25 quads and one hexagon");
    ptc_import_options opts {};
    opts.flags = PTC_IMPORT_EXACT_MATCH | PTC_IMPORT_FAN | PTC_IMPORT_NO_NORMALIZE;
    expect(import_count(write_file("a.obj", obj), &opts) == 54, "OBJ with the exact / fan options");
    opts.flags = 0;
    opts.position_epsilon_rel = 0.3f;
    opts.join_epsilon = 0.5f;
    expect(import_count(write_file("a.obj", obj), &opts) == 54, "OBJ with wide tolerances");
    for (int big = 0; big < 2; ++big)
    {
        const std::string ply = make_ply(big != 0);
        expect(import_count(write_file("a.ply", ply)) == 32, "binary PLY: 16 quads");
        // every truncation of the file and a sweep of single-byte corruptions: a result or an error, no fault
        for (size_t cut = 0; cut < ply.size(); cut += 3)
        {
            (void)import_count(write_file("cut.ply", ply.substr(0, cut)));
        }
        for (size_t at = 0; at < ply.size(); at += 5)
        {
            std::string bad = ply;
            bad[at] = (char)(bad[at] ^ 0x5a);
            (void)import_count(write_file("bad.ply", bad));
        }
    }
    for (size_t cut = 0; cut < obj.size(); cut += 7)
    {
        (void)import_count(write_file("cut.obj", obj.substr(0, cut)));
    }
    for (size_t at = 0; at < obj.size(); at += 11)
    {
        std::string bad = obj;
        bad[at] = (char)(bad[at] ^ 0x15);
        (void)import_count(write_file("bad.obj", bad));
    }
    expect(import_count(write_file("huge.ply", "ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\n"
                                               "property float z\nelement face 99999999999999999999\n"
                                               "property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n")) == -1,
           "PLY with an absurd face count is refused");
    // non-finite coordinates are refused; huge finite ones kept with NO_NORMALIZE import without overflow
    for (const char *value : {"nan", "inf", "-inf"})
    {
        const std::string bad = std::string("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\n"
                                            "property float z\nelement face 1\nproperty list uchar int vertex_indices\n"
                                            "end_header\n0 0 0\n1 ") + value + " 0\n0 1 0\n3 0 1 2\n";
        expect(import_count(write_file("nonfinite.ply", bad)) == -1, "PLY with a non-finite coordinate is refused");
        const std::string bad_obj = std::string("v 0 0 0\nv 1 0 ") + value + "\nv 0 1 0\nf 1 2 3\n";
        expect(import_count(write_file("nonfinite.obj", bad_obj)) == -1, "OBJ with a non-finite coordinate is refused");
    }
    {
        ptc_import_options keep {};
        keep.flags = PTC_IMPORT_NO_NORMALIZE;
        const std::string far = "ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\n"
                                "property float z\nelement face 2\nproperty list uchar int vertex_indices\n"
                                "end_header\n1e30 0 0\n-1e30 1e30 0\n0 -1e30 3e38\n1e30 1e30 1e30\n3 0 1 2\n3 1 2 3\n";
        expect(import_count(write_file("far.ply", far), &keep) == 2, "PLY with coordinates of 1e30 kept as they are");
        expect(import_count(write_file("far.ply", far)) == 2, "the same normalised to [-1, 1]");
    }
    for (int rle = 0; rle < 2; ++rle)
    {
        const std::string hdr = make_hdr(rle != 0);
        expect(read_hdr_pixels(write_file("a.hdr", hdr)) == 64, "HDR decode");
        for (size_t cut = 0; cut < hdr.size(); cut += 2)
        {
            (void)read_hdr_pixels(write_file("cut.hdr", hdr.substr(0, cut)));
        }
        for (size_t at = 0; at < hdr.size(); ++at)
        {
            std::string bad = hdr;
            bad[at] = (char)(bad[at] ^ 0xff);
            (void)read_hdr_pixels(write_file("bad.hdr", bad));
        }
    }
    if (failures == 0)
    {
        std::printf("ptc_host_san_check ok\n");
    }
    return failures == 0 ? 0 : 1;
}
