// Mesh ingestion: the data format on the input side of the scene. The
// reference imports its scene with Assimp (src/application.cpp:150-189) and
// reads positions, normals and triangle indices of mesh 0 out of the result
// (src/renderer.cpp:1818-1838). Assimp is a third-party dependency pinned at
// v5.3.1 (CMakeLists.txt:89-92) and absent from the reference tree, so this
// file restates the published behaviour of the post-processing steps the
// reference asks for, in Assimp's fixed step order:
//   PreTransformVertices + AI_CONFIG_PP_PTV_NORMALIZE  scene bbox -> [-1, 1]
//   Triangulate                                         quads from their concave
//                                                       corner, larger polygons by
//                                                       ear clipping
//   FindDegenerates (area check off) + SortByPType      drop triangles with two
//     removing lines and points                         equal corner positions
//   FixInfacingNormals                                  bbox-growth heuristic
//   GenSmoothNormals, max angle 80 degrees              only if the file has none;
//                                                       corners within 1e-4 of the
//                                                       mesh's bbox diagonal share
//   JoinIdenticalVertices                               position and normal within
//                                                       1e-5 of an earlier vertex
// for Wavefront OBJ and Stanford PLY (ascii, binary little / big endian) files.
// Where Assimp's result depends on the order in which its spatial sort or its
// hash map hands out candidates, this importer goes by ascending corner index
// (sums) and by the earliest vertex (joins). PTC_IMPORT_EXACT_MATCH and
// PTC_IMPORT_FAN select bit-exact position matching and fan triangulation.
// Host code only: no device work happens here.
#include "ptc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

struct ptc_mesh_file
{
    struct Mesh
    {
        std::vector<float> positions, normals;
        std::vector<uint32_t> indices;
    };
    std::vector<Mesh> meshes;
};

namespace
{

struct V3
{
    float x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normalize_safe(V3 a)
{
    const float len = std::sqrt(dot(a, a));
    return len > 0.0f ? V3 {a.x / len, a.y / len, a.z / len} : a;
}

// One imported mesh before post-processing: a polygon soup with separately
// indexed positions and (optional) normals.
struct RawMesh
{
    std::vector<V3> positions, normals;
    std::vector<uint32_t> face_start; // into corners; last entry = corners.size()
    std::vector<uint32_t> corner_pos, corner_nrm; // corner_nrm empty or parallel to corner_pos
    bool has_normals() const { return !corner_nrm.empty(); }
};

struct ParseError
{
    std::string what;
};

std::vector<char> read_file(const char *file_name)
{
    std::FILE *f = std::fopen(file_name, "rb");
    if (f == nullptr)
    {
        throw ParseError {"Unable to open file \"" + std::string(file_name) + "\"."};
    }
    std::vector<char> data;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0)
    {
        data.insert(data.end(), buf, buf + n);
    }
    std::fclose(f);
    return data;
}

// ---- Wavefront OBJ ------------------------------------------------------------

void parse_obj(const std::vector<char> &data, std::vector<RawMesh> &out)
{
    std::vector<V3> positions, normals;
    struct Group
    {
        std::vector<uint32_t> face_start, corner_pos, corner_nrm;
        bool any_normal = false, all_normal = true;
    };
    std::vector<Group> groups(1);
    const char *p = data.data(), *end = p + data.size();
    auto skip_space = [&]() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r'))
        {
            ++p;
        }
    };
    auto read_float = [&]() -> float {
        skip_space();
        char *q = nullptr;
        const float v = std::strtof(p, &q);
        if (q == p)
        {
            throw ParseError {"OBJ: malformed number"};
        }
        p = q;
        return v;
    };
    std::vector<char> text; // strtof / strtol need a terminator
    text.assign(data.begin(), data.end());
    text.push_back('\n');
    text.push_back('\0');
    p = text.data();
    end = p + text.size() - 1;
    while (p < end)
    {
        skip_space();
        const char *line = p;
        const char *eol = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        if (eol == nullptr)
        {
            eol = end;
        }
        if (line[0] == 'v' && (line[1] == ' ' || line[1] == '\t'))
        {
            p += 1;
            const float x = read_float(), y = read_float(), z = read_float();
            positions.push_back({x, y, z});
        }
        else if (line[0] == 'v' && line[1] == 'n' && (line[2] == ' ' || line[2] == '\t'))
        {
            p += 2;
            const float x = read_float(), y = read_float(), z = read_float();
            normals.push_back({x, y, z});
        }
        else if (line[0] == 'f' && (line[1] == ' ' || line[1] == '\t'))
        {
            p += 1;
            Group &g = groups.back();
            g.face_start.push_back((uint32_t)g.corner_pos.size());
            for (;;)
            {
                skip_space();
                if (p >= eol)
                {
                    break;
                }
                char *q = nullptr;
                long vi = std::strtol(p, &q, 10), ni = 0;
                if (q == p)
                {
                    throw ParseError {"OBJ: malformed face"};
                }
                p = q;
                bool has_n = false;
                if (*p == '/')
                {
                    ++p;
                    if (*p != '/')
                    {
                        (void)std::strtol(p, &q, 10); // texture coordinate index, unused
                        p = q;
                    }
                    if (*p == '/')
                    {
                        ++p;
                        ni = std::strtol(p, &q, 10);
                        has_n = q != p;
                        p = q;
                    }
                }
                vi = vi < 0 ? (long)positions.size() + vi : vi - 1;
                if (vi < 0 || vi >= (long)positions.size())
                {
                    throw ParseError {"OBJ: vertex index out of range"};
                }
                if (has_n)
                {
                    ni = ni < 0 ? (long)normals.size() + ni : ni - 1;
                    if (ni < 0 || ni >= (long)normals.size())
                    {
                        throw ParseError {"OBJ: normal index out of range"};
                    }
                }
                g.corner_pos.push_back((uint32_t)vi);
                g.corner_nrm.push_back((uint32_t)ni);
                g.any_normal |= has_n;
                g.all_normal &= has_n;
            }
        }
        else if ((line[0] == 'o' && (line[1] == ' ' || line[1] == '\t')) || std::strncmp(line, "usemtl", 6) == 0)
        {
            // a new object or material starts a new mesh (once the current one has faces)
            if (!groups.back().face_start.empty())
            {
                groups.emplace_back();
            }
        }
        p = eol + 1;
    }
    for (Group &g : groups)
    {
        if (g.face_start.empty())
        {
            continue;
        }
        RawMesh m;
        // compact the position / normal tables to what this mesh references
        std::unordered_map<uint32_t, uint32_t> pmap, nmap;
        for (uint32_t &c : g.corner_pos)
        {
            auto it = pmap.find(c);
            if (it == pmap.end())
            {
                it = pmap.emplace(c, (uint32_t)m.positions.size()).first;
                m.positions.push_back(positions[c]);
            }
            c = it->second;
        }
        if (g.any_normal && g.all_normal)
        {
            for (uint32_t &c : g.corner_nrm)
            {
                auto it = nmap.find(c);
                if (it == nmap.end())
                {
                    it = nmap.emplace(c, (uint32_t)m.normals.size()).first;
                    m.normals.push_back(normals[c]);
                }
                c = it->second;
            }
            m.corner_nrm = std::move(g.corner_nrm);
        }
        g.face_start.push_back((uint32_t)g.corner_pos.size());
        m.face_start = std::move(g.face_start);
        m.corner_pos = std::move(g.corner_pos);
        out.push_back(std::move(m));
    }
}

// ---- Stanford PLY -------------------------------------------------------------

enum PlyType { kI8, kU8, kI16, kU16, kI32, kU32, kF32, kF64, kBad };

PlyType ply_type(const std::string &s)
{
    static const struct { const char *a, *b; PlyType t; } names[] = {
        {"char", "int8", kI8},     {"uchar", "uint8", kU8},     {"short", "int16", kI16}, {"ushort", "uint16", kU16},
        {"int", "int32", kI32},    {"uint", "uint32", kU32},    {"float", "float32", kF32}, {"double", "float64", kF64}};
    for (const auto &n : names)
    {
        if (s == n.a || s == n.b)
        {
            return n.t;
        }
    }
    return kBad;
}

int ply_size(PlyType t)
{
    static const int sizes[] = {1, 1, 2, 2, 4, 4, 4, 8, 0};
    return sizes[t];
}

struct PlyProperty
{
    std::string name;
    bool is_list = false;
    PlyType count_type = kBad, type = kBad;
};
struct PlyElement
{
    std::string name;
    size_t count = 0;
    std::vector<PlyProperty> props;
};

struct PlyCursor
{
    const char *p, *end;
    int format; // 0 ascii, 1 little endian, 2 big endian
    double read(PlyType t)
    {
        if (format == 0)
        {
            while (p < end && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n'))
            {
                ++p;
            }
            char *q = nullptr;
            const double v = std::strtod(p, &q);
            if (q == p)
            {
                throw ParseError {"PLY: malformed number"};
            }
            p = q;
            return v;
        }
        const int n = ply_size(t);
        if (end - p < n)
        {
            throw ParseError {"PLY: unexpected end of file"};
        }
        unsigned char b[8];
        std::memcpy(b, p, (size_t)n);
        p += n;
        if (format == 2)
        {
            std::reverse(b, b + n);
        }
        switch (t)
        {
        case kI8: { int8_t v; std::memcpy(&v, b, 1); return v; }
        case kU8: { uint8_t v; std::memcpy(&v, b, 1); return v; }
        case kI16: { int16_t v; std::memcpy(&v, b, 2); return v; }
        case kU16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
        case kI32: { int32_t v; std::memcpy(&v, b, 4); return v; }
        case kU32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
        case kF32: { float v; std::memcpy(&v, b, 4); return v; }
        default: { double v; std::memcpy(&v, b, 8); return v; }
        }
    }
};

void parse_ply(const std::vector<char> &data, std::vector<RawMesh> &out)
{
    std::vector<char> text(data);
    text.push_back('\0');
    const char *p = text.data(), *end = p + data.size();
    auto next_line = [&]() -> std::string {
        const char *eol = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        if (eol == nullptr)
        {
            throw ParseError {"PLY: header not terminated"};
        }
        std::string s(p, eol);
        p = eol + 1;
        while (!s.empty() && (s.back() == '\r' || s.back() == ' '))
        {
            s.pop_back();
        }
        return s;
    };
    auto words = [](const std::string &s) {
        std::vector<std::string> w;
        size_t i = 0;
        while (i < s.size())
        {
            while (i < s.size() && (s[i] == ' ' || s[i] == '\t'))
            {
                ++i;
            }
            size_t j = i;
            while (j < s.size() && s[j] != ' ' && s[j] != '\t')
            {
                ++j;
            }
            if (j > i)
            {
                w.push_back(s.substr(i, j - i));
            }
            i = j;
        }
        return w;
    };
    if (next_line() != "ply")
    {
        throw ParseError {"PLY: bad magic"};
    }
    int format = -1;
    std::vector<PlyElement> elements;
    for (;;)
    {
        const auto w = words(next_line());
        if (w.empty() || w[0] == "comment" || w[0] == "obj_info")
        {
            continue;
        }
        if (w[0] == "end_header")
        {
            break;
        }
        if (w[0] == "format" && w.size() >= 2)
        {
            format = w[1] == "ascii" ? 0 : w[1] == "binary_little_endian" ? 1 : w[1] == "binary_big_endian" ? 2 : -1;
        }
        else if (w[0] == "element" && w.size() == 3)
        {
            PlyElement e;
            e.name = w[1];
            e.count = std::strtoull(w[2].c_str(), nullptr, 10);
            elements.push_back(e);
        }
        else if (w[0] == "property" && !elements.empty())
        {
            PlyProperty pr;
            if (w.size() == 5 && w[1] == "list")
            {
                pr.is_list = true;
                pr.count_type = ply_type(w[2]);
                pr.type = ply_type(w[3]);
                pr.name = w[4];
            }
            else if (w.size() == 3)
            {
                pr.type = ply_type(w[1]);
                pr.name = w[2];
            }
            if (pr.type == kBad || (pr.is_list && pr.count_type == kBad))
            {
                throw ParseError {"PLY: unsupported property"};
            }
            elements.back().props.push_back(pr);
        }
    }
    if (format < 0)
    {
        throw ParseError {"PLY: unsupported format"};
    }
    PlyCursor c {p, end, format};
    RawMesh m;
    std::vector<V3> vertex_normals;
    for (const PlyElement &e : elements)
    {
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        if (e.props.empty())
        {
            continue; // nothing stored per entry, whatever the header's count says
        }
        // An entry takes at least one byte per property (binary: the property's
        // or its list count's size), so the bytes left bound the count a header
        // may claim; a larger one is refused before anything is reserved.
        size_t min_entry = 0;
        for (const PlyProperty &pr : e.props)
        {
            min_entry += format == 0 ? 1u : (size_t)ply_size(pr.is_list ? pr.count_type : pr.type);
        }
        if (e.count > (size_t)(c.end - c.p) / min_entry)
        {
            throw ParseError {"PLY: element count exceeds the file size"};
        }
        if (is_face)
        {
            m.face_start.reserve(e.count + 1);
        }
        for (size_t i = 0; i < e.count; ++i)
        {
            V3 pos {0, 0, 0}, nrm {0, 0, 0};
            bool got_n = false;
            for (const PlyProperty &pr : e.props)
            {
                if (!pr.is_list)
                {
                    const double v = c.read(pr.type);
                    if (is_vertex)
                    {
                        if (pr.name == "x") pos.x = (float)v;
                        else if (pr.name == "y") pos.y = (float)v;
                        else if (pr.name == "z") pos.z = (float)v;
                        else if (pr.name == "nx") { nrm.x = (float)v; got_n = true; }
                        else if (pr.name == "ny") nrm.y = (float)v;
                        else if (pr.name == "nz") nrm.z = (float)v;
                    }
                    continue;
                }
                const long n = (long)c.read(pr.count_type);
                const bool indices = is_face && (pr.name == "vertex_indices" || pr.name == "vertex_index");
                if (indices)
                {
                    m.face_start.push_back((uint32_t)m.corner_pos.size());
                }
                for (long k = 0; k < n; ++k)
                {
                    const double v = c.read(pr.type);
                    if (indices)
                    {
                        if (v < 0 || v >= (double)m.positions.size())
                        {
                            throw ParseError {"PLY: vertex index out of range"};
                        }
                        m.corner_pos.push_back((uint32_t)v);
                    }
                }
            }
            if (is_vertex)
            {
                m.positions.push_back(pos);
                if (got_n)
                {
                    vertex_normals.push_back(nrm);
                }
            }
        }
    }
    m.face_start.push_back((uint32_t)m.corner_pos.size());
    if (!vertex_normals.empty() && vertex_normals.size() == m.positions.size())
    {
        m.normals = std::move(vertex_normals);
        m.corner_nrm = m.corner_pos;
    }
    if (m.face_start.size() > 1)
    {
        out.push_back(std::move(m));
    }
}

// ---- post-processing ------------------------------------------------------------

struct Corner
{
    V3 p, n;
};

inline uint32_t bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u == 0x80000000u ? 0u : u; // -0 and +0 are the same position
}

struct Key3
{
    uint32_t a, b, c;
    bool operator==(const Key3 &o) const { return a == o.a && b == o.b && c == o.c; }
};
struct Key6
{
    uint32_t v[6];
    bool operator==(const Key6 &o) const { return std::memcmp(v, o.v, sizeof v) == 0; }
};
struct KeyHash
{
    size_t operator()(const Key3 &k) const
    {
        uint64_t h = k.a * 0x9e3779b97f4a7c15ull;
        h = (h ^ k.b) * 0xbf58476d1ce4e5b9ull;
        h = (h ^ k.c) * 0x94d049bb133111ebull;
        return (size_t)(h ^ (h >> 29));
    }
    size_t operator()(const Key6 &k) const
    {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        for (uint32_t x : k.v)
        {
            h = (h ^ x) * 0xbf58476d1ce4e5b9ull;
            h ^= h >> 31;
        }
        return (size_t)h;
    }
};

// Positions within a distance: a grid of cells as wide as the distance over the
// distinct positions (most corners of a mesh share their position with others
// bit for bit, so the search runs once per distinct position).
struct Cell
{
    int64_t x, y, z;
    bool operator==(const Cell &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellHash
{
    size_t operator()(const Cell &c) const
    {
        uint64_t h = (uint64_t)c.x * 0x9e3779b97f4a7c15ull;
        h = (h ^ (uint64_t)c.y) * 0xbf58476d1ce4e5b9ull;
        h = (h ^ (uint64_t)c.z) * 0x94d049bb133111ebull;
        return (size_t)(h ^ (h >> 29));
    }
};

struct PositionIndex
{
    std::vector<V3> unique;               // distinct positions, in order of first use
    std::vector<uint32_t> unique_of;      // per input position: its distinct position
    std::vector<std::vector<uint32_t>> near; // per distinct position: distinct positions within the distance (itself first)

    // radius <= 0: bit-exact matching only
    PositionIndex(const std::vector<V3> &pos, float radius)
    {
        std::unordered_map<Key3, uint32_t, KeyHash> ids;
        ids.reserve(pos.size() / 2);
        unique_of.resize(pos.size());
        for (size_t i = 0; i < pos.size(); ++i)
        {
            const Key3 k {bits(pos[i].x), bits(pos[i].y), bits(pos[i].z)};
            auto it = ids.find(k);
            if (it == ids.end())
            {
                it = ids.emplace(k, (uint32_t)unique.size()).first;
                unique.push_back(pos[i]);
            }
            unique_of[i] = it->second;
        }
        near.resize(unique.size());
        for (uint32_t u = 0; u < unique.size(); ++u)
        {
            near[u].push_back(u);
        }
        if (!(radius > 0.0f))
        {
            return;
        }
        const double inv = 1.0 / (double)radius;
        // (cell numbers are held within +-2^62: coordinates that large against this radius -- file
        // coordinates kept with PTC_IMPORT_NO_NORMALIZE -- share the outermost cells, which costs time,
        // not correctness, and neither the cast nor the neighbour offsets below can overflow)
        auto cell_axis = [&](float v) {
            const double c = std::floor((double)v * inv);
            constexpr double kEdge = 4611686018427387904.0; // 2^62
            return (int64_t)std::max(-kEdge, std::min(kEdge, c));
        };
        auto cell_of = [&](V3 v) { return Cell {cell_axis(v.x), cell_axis(v.y), cell_axis(v.z)}; };
        std::unordered_map<Cell, std::vector<uint32_t>, CellHash> grid;
        grid.reserve(unique.size());
        for (uint32_t u = 0; u < unique.size(); ++u)
        {
            grid[cell_of(unique[u])].push_back(u);
        }
        const float r2 = radius * radius;
        for (uint32_t u = 0; u < unique.size(); ++u)
        {
            const Cell c = cell_of(unique[u]);
            for (int dz = -1; dz <= 1; ++dz)
            {
                for (int dy = -1; dy <= 1; ++dy)
                {
                    for (int dx = -1; dx <= 1; ++dx)
                    {
                        auto it = grid.find(Cell {c.x + dx, c.y + dy, c.z + dz});
                        if (it == grid.end())
                        {
                            continue;
                        }
                        for (uint32_t w : it->second)
                        {
                            const V3 d = unique[w] - unique[u];
                            if (w != u && dot(d, d) < r2)
                            {
                                near[u].push_back(w);
                            }
                        }
                    }
                }
            }
        }
    }
};

double area2(double ax, double ay, double bx, double by, double cx, double cy)
{
    return (bx - ax) * (cy - ay) - (cx - ax) * (by - ay); // twice the signed area of (a, b, c)
}

// Splits one polygon (corner indices c[0..n)) into triangles, as local corner
// numbers. Triangles stay; a quad is cut from its concave corner when it has
// one (the angles its two edges make with the diagonal there add up to more
// than pi), else from corner 0; larger polygons are ear-clipped in the plane
// their Newell normal points out of: going round from the last ear, the first
// convex corner whose triangle holds none of the other corners is cut off. A
// polygon without an ear (not simple) falls back to the fan from corner 0.
void split_polygon(const V3 *pts, uint32_t n, bool fan, std::vector<uint32_t> &out)
{
    out.clear();
    if (n < 3)
    {
        return;
    }
    auto fan_from = [&](uint32_t start) {
        for (uint32_t k = 2; k < n; ++k)
        {
            out.insert(out.end(), {start, (start + k - 1) % n, (start + k) % n});
        }
    };
    if (n == 3 || fan)
    {
        fan_from(0);
        return;
    }
    if (n == 4)
    {
        uint32_t start = 0;
        for (uint32_t i = 0; i < 4; ++i)
        {
            const V3 v = pts[i];
            const V3 left = normalize_safe(pts[(i + 3) % 4] - v), diag = normalize_safe(pts[(i + 2) % 4] - v),
                     right = normalize_safe(pts[(i + 1) % 4] - v);
            auto clamped = [](float x) { return x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x); };
            const float angle = std::acos(clamped(dot(left, diag))) + std::acos(clamped(dot(right, diag)));
            if (angle > 3.14159265358979323846f)
            {
                start = i;
                break;
            }
        }
        fan_from(start);
        return;
    }
    // Newell normal, then the projection that drops its largest component
    double nx = 0.0, ny = 0.0, nz = 0.0;
    for (uint32_t i = 0; i < n; ++i)
    {
        const V3 a = pts[i], b = pts[(i + 1) % n];
        nx += ((double)a.y - b.y) * ((double)a.z + b.z);
        ny += ((double)a.z - b.z) * ((double)a.x + b.x);
        nz += ((double)a.x - b.x) * ((double)a.y + b.y);
    }
    const double ax = std::fabs(nx), ay = std::fabs(ny), az = std::fabs(nz);
    int drop = az >= ax && az >= ay ? 2 : (ay >= ax ? 1 : 0);
    const double sign = drop == 2 ? nz : (drop == 1 ? ny : nx);
    std::vector<double> px(n), py(n);
    for (uint32_t i = 0; i < n; ++i)
    {
        const double c[3] = {pts[i].x, pts[i].y, pts[i].z};
        double u = c[(drop + 1) % 3], v = c[(drop + 2) % 3];
        if (sign < 0.0)
        {
            std::swap(u, v); // keeps the polygon counter-clockwise in the plane
        }
        px[i] = u;
        py[i] = v;
    }
    std::vector<char> done(n, 0);
    uint32_t left = n, ear = 0;
    auto next_of = [&](uint32_t i) {
        do
        {
            i = (i + 1) % n;
        } while (done[i]);
        return i;
    };
    auto prev_of = [&](uint32_t i) {
        do
        {
            i = (i + n - 1) % n;
        } while (done[i]);
        return i;
    };
    while (left > 3)
    {
        bool found = false;
        uint32_t e = ear;
        for (uint32_t tries = 0; tries < left; ++tries, e = next_of(e))
        {
            const uint32_t a = prev_of(e), b = next_of(e);
            if (!(area2(px[a], py[a], px[e], py[e], px[b], py[b]) > 0.0))
            {
                continue; // reflex or flat corner
            }
            bool holds_other = false;
            for (uint32_t t = 0; t < n && !holds_other; ++t)
            {
                if ((px[t] == px[a] && py[t] == py[a]) || (px[t] == px[e] && py[t] == py[e]) ||
                    (px[t] == px[b] && py[t] == py[b]))
                {
                    continue; // several corners may name one position
                }
                holds_other = area2(px[a], py[a], px[e], py[e], px[t], py[t]) >= 0.0 &&
                              area2(px[e], py[e], px[b], py[b], px[t], py[t]) >= 0.0 &&
                              area2(px[b], py[b], px[a], py[a], px[t], py[t]) >= 0.0;
            }
            if (!holds_other)
            {
                out.insert(out.end(), {a, e, b});
                done[e] = 1;
                --left;
                ear = b;
                found = true;
                break;
            }
        }
        if (!found)
        {
            out.clear();
            fan_from(0);
            return;
        }
    }
    const uint32_t a = next_of(n - 1), b = next_of(a), c = next_of(b);
    out.insert(out.end(), {a, b, c});
}

// Triangulate + FindDegenerates: triangle corners in "verbose" form (one vertex
// per corner), triangles with two coincident corners dropped.
std::vector<Corner> triangulate(const RawMesh &m, bool fan)
{
    std::vector<Corner> tri;
    std::vector<V3> pts;
    std::vector<uint32_t> local;
    for (size_t f = 0; f + 1 < m.face_start.size(); ++f)
    {
        const uint32_t b = m.face_start[f], e = m.face_start[f + 1];
        pts.clear();
        for (uint32_t k = b; k < e; ++k)
        {
            pts.push_back(m.positions[m.corner_pos[k]]);
        }
        split_polygon(pts.data(), e - b, fan, local);
        for (size_t t = 0; t + 2 < local.size(); t += 3)
        {
            Corner v[3];
            for (int j = 0; j < 3; ++j)
            {
                const uint32_t c = b + local[t + j];
                v[j].p = m.positions[m.corner_pos[c]];
                v[j].n = m.has_normals() ? m.normals[m.corner_nrm[c]] : V3 {0, 0, 0};
            }
            auto same = [](V3 a, V3 b2) { return a.x == b2.x && a.y == b2.y && a.z == b2.z; };
            if (same(v[0].p, v[1].p) || same(v[1].p, v[2].p) || same(v[0].p, v[2].p))
            {
                continue;
            }
            tri.insert(tri.end(), v, v + 3);
        }
    }
    return tri;
}

// The distance within which GenSmoothNormals takes two corners to be at one
// position: rel (1e-4) of the diagonal of the mesh's bounding box.
float position_epsilon(const std::vector<Corner> &tri, float rel)
{
    V3 lo {1e10f, 1e10f, 1e10f}, hi {-1e10f, -1e10f, -1e10f};
    for (const Corner &c : tri)
    {
        lo = {std::min(lo.x, c.p.x), std::min(lo.y, c.p.y), std::min(lo.z, c.p.z)};
        hi = {std::max(hi.x, c.p.x), std::max(hi.y, c.p.y), std::max(hi.z, c.p.z)};
    }
    const V3 d = hi - lo;
    return std::sqrt(dot(d, d)) * rel;
}

// GenSmoothNormals: every corner gets the normalised sum of the face normals of
// all corners within `radius` of its position (0: at its position bit for bit)
// whose face normal is within max_angle of its own, summed in corner order.
void smooth_normals(std::vector<Corner> &tri, float max_angle_deg, float radius)
{
    const size_t n = tri.size();
    std::vector<V3> face(n), pos(n);
    for (size_t t = 0; t < n; t += 3)
    {
        const V3 fn = normalize_safe(cross(tri[t + 1].p - tri[t].p, tri[t + 2].p - tri[t].p));
        face[t] = face[t + 1] = face[t + 2] = fn;
        pos[t] = tri[t].p;
        pos[t + 1] = tri[t + 1].p;
        pos[t + 2] = tri[t + 2].p;
    }
    const PositionIndex index(pos, radius);
    std::vector<std::vector<uint32_t>> corners_at(index.unique.size()); // ascending
    for (size_t i = 0; i < n; ++i)
    {
        corners_at[index.unique_of[i]].push_back((uint32_t)i);
    }
    const float limit = std::cos(max_angle_deg * 3.14159265358979323846f / 180.0f);
    const bool unlimited = max_angle_deg >= 175.0f;
    std::vector<uint32_t> merged;
    for (size_t i = 0; i < n; ++i)
    {
        const std::vector<uint32_t> &nb = index.near[index.unique_of[i]];
        const std::vector<uint32_t> *list = &corners_at[nb[0]];
        if (nb.size() > 1)
        {
            merged.clear();
            for (uint32_t u : nb)
            {
                merged.insert(merged.end(), corners_at[u].begin(), corners_at[u].end());
            }
            std::sort(merged.begin(), merged.end());
            list = &merged;
        }
        const V3 vr = face[i];
        V3 sum {0, 0, 0};
        for (uint32_t j : *list)
        {
            const V3 v = face[j];
            if (v.x != v.x)
            {
                continue;
            }
            if (unlimited || j == i || dot(v, vr) >= limit)
            {
                sum = sum + v;
            }
        }
        tri[i].n = normalize_safe(sum);
    }
}

// FixInfacingNormals: if the bounding box of (position + normal) is smaller
// than the bounding box of the positions, the normals point inwards: flip them
// and the winding. Skipped for planar meshes and inconsistent boxes.
void fix_infacing_normals(std::vector<Corner> &tri)
{
    if (tri.empty())
    {
        return;
    }
    V3 lo0 {1e10f, 1e10f, 1e10f}, hi0 {-1e10f, -1e10f, -1e10f}, lo1 = lo0, hi1 = hi0;
    for (const Corner &c : tri)
    {
        const V3 q = c.p + c.n;
        lo1 = {std::min(lo1.x, c.p.x), std::min(lo1.y, c.p.y), std::min(lo1.z, c.p.z)};
        hi1 = {std::max(hi1.x, c.p.x), std::max(hi1.y, c.p.y), std::max(hi1.z, c.p.z)};
        lo0 = {std::min(lo0.x, q.x), std::min(lo0.y, q.y), std::min(lo0.z, q.z)};
        hi0 = {std::max(hi0.x, q.x), std::max(hi0.y, q.y), std::max(hi0.z, q.z)};
    }
    const V3 d0 = hi0 - lo0, d1 = hi1 - lo1;
    if ((d0.x > 0.0f) != (d1.x > 0.0f) || (d0.y > 0.0f) != (d1.y > 0.0f) || (d0.z > 0.0f) != (d1.z > 0.0f))
    {
        return;
    }
    if (d1.x < 0.05f * std::sqrt(d1.y * d1.z) || d1.y < 0.05f * std::sqrt(d1.z * d1.x) ||
        d1.z < 0.05f * std::sqrt(d1.y * d1.x))
    {
        return;
    }
    if (std::fabs(d0.x * d0.y * d0.z) < std::fabs(d1.x * d1.y * d1.z))
    {
        for (size_t t = 0; t < tri.size(); t += 3)
        {
            for (int j = 0; j < 3; ++j)
            {
                V3 &nn = tri[t + j].n;
                nn = {-nn.x, -nn.y, -nn.z};
            }
            std::swap(tri[t], tri[t + 2]);
        }
    }
}

// JoinIdenticalVertices: a corner becomes the earliest vertex made so far whose
// position and normal are both within `eps` of its own (eps 0: equal bit for
// bit), else a new vertex.
void join_vertices(const std::vector<Corner> &tri, bool join, float eps, ptc_mesh_file::Mesh &out)
{
    out.indices.reserve(tri.size());
    auto emit = [&](const Corner &c) {
        const uint32_t id = (uint32_t)(out.positions.size() / 3);
        out.positions.insert(out.positions.end(), {c.p.x, c.p.y, c.p.z});
        out.normals.insert(out.normals.end(), {c.n.x, c.n.y, c.n.z});
        return id;
    };
    if (!join)
    {
        for (const Corner &c : tri)
        {
            out.indices.push_back(emit(c));
        }
        return;
    }
    if (!(eps > 0.0f))
    {
        std::unordered_map<Key6, uint32_t, KeyHash> seen;
        seen.reserve(tri.size() / 4);
        for (const Corner &c : tri)
        {
            const Key6 k {{bits(c.p.x), bits(c.p.y), bits(c.p.z), bits(c.n.x), bits(c.n.y), bits(c.n.z)}};
            auto it = seen.find(k);
            if (it == seen.end())
            {
                it = seen.emplace(k, emit(c)).first;
            }
            out.indices.push_back(it->second);
        }
        return;
    }
    std::vector<V3> pos(tri.size());
    for (size_t i = 0; i < tri.size(); ++i)
    {
        pos[i] = tri[i].p;
    }
    const PositionIndex index(pos, eps);
    std::vector<std::vector<uint32_t>> made_at(index.unique.size()); // vertices made at a distinct position, ascending
    const float e2 = eps * eps;
    for (size_t i = 0; i < tri.size(); ++i)
    {
        const Corner &c = tri[i];
        const uint32_t u = index.unique_of[i];
        uint32_t id = 0xffffffffu;
        for (uint32_t w : index.near[u])
        {
            for (uint32_t v : made_at[w])
            {
                const V3 dn = V3 {out.normals[3 * (size_t)v], out.normals[3 * (size_t)v + 1], out.normals[3 * (size_t)v + 2]} - c.n;
                if (v < id && !(dot(dn, dn) > e2))
                {
                    id = v;
                }
            }
        }
        if (id == 0xffffffffu)
        {
            id = emit(c);
            made_at[u].push_back(id);
        }
        out.indices.push_back(id);
    }
}

bool ends_with_nocase(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    if (s.size() < n)
    {
        return false;
    }
    for (size_t i = 0; i < n; ++i)
    {
        if (std::tolower((unsigned char)s[s.size() - n + i]) != suffix[i])
        {
            return false;
        }
    }
    return true;
}

} // namespace

extern "C" int ptc_mesh_file_open(const char *file_name, const ptc_import_options *options, ptc_mesh_file **out)
{
    if (file_name == nullptr || out == nullptr)
    {
        ptc::set_error("ptc_mesh_file_open: null argument");
        return PTC_E_INVALID;
    }
    *out = nullptr;
    const float max_angle = options != nullptr && options->max_smoothing_angle_deg > 0.0f
                                ? options->max_smoothing_angle_deg
                                : 80.0f; // src/application.cpp:155
    const uint32_t flags = options != nullptr ? options->flags : 0u;
    const bool exact = (flags & PTC_IMPORT_EXACT_MATCH) != 0u;
    // ComputePositionEpsilon() of Assimp's smoothing step and the epsilon of its join step
    const float smooth_rel = exact ? 0.0f : (options != nullptr && options->position_epsilon_rel > 0.0f ? options->position_epsilon_rel : 1e-4f);
    const float join_eps = exact ? 0.0f : (options != nullptr && options->join_epsilon > 0.0f ? options->join_epsilon : 1e-5f);
    try
    {
        const std::vector<char> data = read_file(file_name);
        std::vector<RawMesh> raw;
        const std::string name(file_name);
        if (ends_with_nocase(name, ".ply") || (data.size() >= 3 && std::memcmp(data.data(), "ply", 3) == 0))
        {
            parse_ply(data, raw);
        }
        else if (ends_with_nocase(name, ".obj"))
        {
            parse_obj(data, raw);
        }
        else
        {
            throw ParseError {"No suitable reader found for the file format of file \"" + name + "\"."};
        }
        // A NaN or an infinity among the coordinates (a binary PLY can carry any bit
        // pattern) would poison the bounding box, the smoothing and joining grids and
        // finally the tree: such a file is refused here.
        for (const RawMesh &m : raw)
        {
            for (const std::vector<V3> *list : {&m.positions, &m.normals})
            {
                for (const V3 &v : *list)
                {
                    if (!std::isfinite(v.x) || !std::isfinite(v.y) || !std::isfinite(v.z))
                    {
                        throw ParseError {"Non-finite vertex coordinate or normal in file \"" + name + "\"."};
                    }
                }
            }
        }
        // PreTransformVertices with AI_CONFIG_PP_PTV_NORMALIZE: the bounding box of
        // the whole scene goes to [-1, 1] along its longest axis, centred
        if (!(flags & PTC_IMPORT_NO_NORMALIZE))
        {
            V3 lo {1e10f, 1e10f, 1e10f}, hi {-1e10f, -1e10f, -1e10f};
            for (const RawMesh &m : raw)
            {
                for (const V3 &v : m.positions)
                {
                    lo = {std::min(lo.x, v.x), std::min(lo.y, v.y), std::min(lo.z, v.z)};
                    hi = {std::max(hi.x, v.x), std::max(hi.y, v.y), std::max(hi.z, v.z)};
                }
            }
            V3 d = hi - lo;
            const float div = std::max(d.x, std::max(d.y, d.z)) * 0.5f;
            d = {lo.x + d.x * 0.5f, lo.y + d.y * 0.5f, lo.z + d.z * 0.5f};
            if (div > 0.0f)
            {
                for (RawMesh &m : raw)
                {
                    for (V3 &v : m.positions)
                    {
                        v = {(v.x - d.x) / div, (v.y - d.y) / div, (v.z - d.z) / div};
                    }
                }
            }
        }
        std::unique_ptr<ptc_mesh_file> file(new ptc_mesh_file);
        for (const RawMesh &m : raw)
        {
            std::vector<Corner> tri = triangulate(m, (flags & PTC_IMPORT_FAN) != 0u);
            if (tri.empty())
            {
                continue;
            }
            if (!m.has_normals())
            {
                smooth_normals(tri, max_angle, smooth_rel > 0.0f ? position_epsilon(tri, smooth_rel) : 0.0f);
            }
            fix_infacing_normals(tri);
            ptc_mesh_file::Mesh mesh;
            join_vertices(tri, !(flags & PTC_IMPORT_NO_JOIN), join_eps, mesh);
            file->meshes.push_back(std::move(mesh));
        }
        if (file->meshes.empty())
        {
            ptc::set_error("Scene has no meshes"); // src/application.cpp:200
            return PTC_E_IO;
        }
        *out = file.release();
        return PTC_OK;
    }
    catch (const ParseError &e)
    {
        ptc::set_error(e.what);
        return PTC_E_IO;
    }
    catch (const std::bad_alloc &)
    {
        ptc::set_error("ptc_mesh_file_open: out of memory");
        return PTC_E_NOMEM;
    }
    catch (const std::exception &e)
    {
        // no exception may leave an extern "C" function
        ptc::set_error(std::string("ptc_mesh_file_open: ") + e.what());
        return PTC_E_IO;
    }
}

extern "C" uint32_t ptc_mesh_file_mesh_count(const ptc_mesh_file *file)
{
    return file != nullptr ? (uint32_t)file->meshes.size() : 0u;
}

extern "C" int ptc_mesh_file_get_mesh(const ptc_mesh_file *file, uint32_t index, ptc_mesh_desc *out)
{
    if (file == nullptr || out == nullptr || index >= file->meshes.size())
    {
        ptc::set_error("ptc_mesh_file_get_mesh: bad argument");
        return PTC_E_INVALID;
    }
    const ptc_mesh_file::Mesh &m = file->meshes[index];
    std::memset(out, 0, sizeof *out);
    out->positions = m.positions.data();
    out->normals = m.normals.data();
    out->indices = m.indices.data();
    out->n_vertices = (uint32_t)(m.positions.size() / 3);
    out->n_triangles = (uint32_t)(m.indices.size() / 3);
    return PTC_OK;
}

extern "C" void ptc_mesh_file_close(ptc_mesh_file *file)
{
    delete file;
}
