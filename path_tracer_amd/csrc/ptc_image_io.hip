// Radiance .hdr (RGBE) reader: the data format on the input side of the sky.
// The reference loads its environment map with stb_image's stbi_loadf(file,
// &w, &h, &channels, 4) (src/utility.cpp:75-92, called at
// src/renderer.cpp:1842); stb_image itself is a third-party dependency that is
// not part of the reference tree, so this follows the published behaviour of
// its HDR decoder: header "#?RADIANCE" or "#?RGBE", FORMAT=32-bit_rle_rgbe, a
// "-Y h +X w" resolution line, flat or new-style run-length-encoded scanlines,
// texel = mantissa * 2^(e - 136) (no +0.5), alpha = 1, rows top to bottom.
// Host code only: no device work happens here.
#include "ptc_internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace
{

struct Reader
{
    const unsigned char *p;
    const unsigned char *end;
    bool eof() const { return p >= end; }
    int get() { return p < end ? *p++ : -1; }
};

// one header line without the newline; false at end of data
bool read_line(Reader &r, std::string &line)
{
    line.clear();
    if (r.eof())
    {
        return false;
    }
    for (;;)
    {
        const int c = r.get();
        if (c < 0 || c == '\n')
        {
            break;
        }
        line.push_back((char)c);
    }
    return true;
}

void rgbe_to_float(const unsigned char *rgbe, float *out)
{
    if (rgbe[3] != 0)
    {
        const float f = std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
        out[0] = (float)rgbe[0] * f;
        out[1] = (float)rgbe[1] * f;
        out[2] = (float)rgbe[2] * f;
    }
    else
    {
        out[0] = out[1] = out[2] = 0.0f;
    }
    out[3] = 1.0f;
}

int fail(const char *file_name, const char *why)
{
    ptc::set_error(std::string("read_hdr_image \"") + file_name + "\": " + why);
    return PTC_E_IO;
}

} // namespace

extern "C" int ptc_read_hdr(const char *file_name, uint32_t *width, uint32_t *height, float *rgba,
                            size_t capacity_floats)
{
    if (file_name == nullptr || width == nullptr || height == nullptr)
    {
        ptc::set_error("ptc_read_hdr: null argument");
        return PTC_E_INVALID;
    }
    std::FILE *f = std::fopen(file_name, "rb");
    if (f == nullptr)
    {
        return fail(file_name, "can't fopen");
    }
    std::vector<unsigned char> data;
    {
        unsigned char buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0)
        {
            data.insert(data.end(), buf, buf + n);
        }
        std::fclose(f);
    }
    Reader r {data.data(), data.data() + data.size()};
    std::string line;
    if (!read_line(r, line) || (line != "#?RADIANCE" && line != "#?RGBE"))
    {
        return fail(file_name, "not HDR");
    }
    bool valid_format = false;
    for (;;)
    {
        if (!read_line(r, line) || line.empty())
        {
            break;
        }
        if (line == "FORMAT=32-bit_rle_rgbe")
        {
            valid_format = true;
        }
    }
    if (!valid_format)
    {
        return fail(file_name, "unsupported format");
    }
    if (!read_line(r, line) || line.compare(0, 3, "-Y ") != 0)
    {
        return fail(file_name, "unsupported data layout");
    }
    char *rest = nullptr;
    const long h = std::strtol(line.c_str() + 3, &rest, 10);
    while (*rest == ' ')
    {
        ++rest;
    }
    if (std::strncmp(rest, "+X ", 3) != 0)
    {
        return fail(file_name, "unsupported data layout");
    }
    const long w = std::strtol(rest + 3, nullptr, 10);
    if (w <= 0 || h <= 0 || w > (1 << 24) || h > (1 << 24))
    {
        return fail(file_name, "too large");
    }
    *width = (uint32_t)w;
    *height = (uint32_t)h;
    if (rgba == nullptr)
    {
        return PTC_OK; // size query
    }
    const size_t n_pixels = (size_t)w * (size_t)h;
    if (capacity_floats < 4 * n_pixels)
    {
        ptc::set_error("ptc_read_hdr: output buffer too small");
        return PTC_E_INVALID;
    }
    auto read_flat = [&](size_t first_pixel, const unsigned char *first4) -> bool {
        // rest of the file is plain RGBE quadruples; first4 (if given) is a texel already consumed
        size_t i = first_pixel;
        if (first4 != nullptr)
        {
            rgbe_to_float(first4, rgba + 4 * i);
            ++i;
        }
        for (; i < n_pixels; ++i)
        {
            if (r.end - r.p < 4)
            {
                return false;
            }
            rgbe_to_float(r.p, rgba + 4 * i);
            r.p += 4;
        }
        return true;
    };
    if (w < 8 || w >= 32768)
    {
        if (!read_flat(0, nullptr))
        {
            return fail(file_name, "corrupt (truncated)");
        }
        return PTC_OK;
    }
    std::vector<unsigned char> scan((size_t)w * 4);
    for (long j = 0; j < h; ++j)
    {
        if (r.end - r.p < 4)
        {
            return fail(file_name, "corrupt (truncated)");
        }
        const unsigned char c1 = r.p[0], c2 = r.p[1], len_hi = r.p[2], len_lo = r.p[3];
        if (c1 != 2 || c2 != 2 || (len_hi & 0x80))
        {
            // not run-length encoded: these four bytes are the first texel of a flat
            // image (the decoder falls back for the whole file, and only on row 0)
            if (j != 0)
            {
                return fail(file_name, "corrupt (mixed encodings)");
            }
            unsigned char first[4] = {c1, c2, len_hi, len_lo};
            r.p += 4;
            if (!read_flat(0, first))
            {
                return fail(file_name, "corrupt (truncated)");
            }
            return PTC_OK;
        }
        r.p += 4;
        if ((((long)len_hi << 8) | len_lo) != w)
        {
            return fail(file_name, "invalid decoded scanline length");
        }
        for (int k = 0; k < 4; ++k)
        {
            long i = 0;
            while (i < w)
            {
                int count = r.get();
                if (count < 0)
                {
                    return fail(file_name, "corrupt (truncated)");
                }
                if (count > 128)
                {
                    const int value = r.get();
                    count -= 128;
                    if (value < 0 || count == 0 || count > w - i)
                    {
                        return fail(file_name, "corrupt (bad RLE data)");
                    }
                    for (int z = 0; z < count; ++z)
                    {
                        scan[(size_t)(i++) * 4 + k] = (unsigned char)value;
                    }
                }
                else
                {
                    if (count == 0 || count > w - i || r.end - r.p < count)
                    {
                        return fail(file_name, "corrupt (bad RLE data)");
                    }
                    for (int z = 0; z < count; ++z)
                    {
                        scan[(size_t)(i++) * 4 + k] = *r.p++;
                    }
                }
            }
        }
        for (long i = 0; i < w; ++i)
        {
            rgbe_to_float(&scan[(size_t)i * 4], rgba + 4 * ((size_t)j * w + i));
        }
    }
    return PTC_OK;
}
