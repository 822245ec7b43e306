// Device-side float arithmetic for the path-tracing kernels (gfx950).
//
// Every function is a fixed sequence of IEEE-754 binary32 operations; the
// library is compiled with -ffp-contract=off and hipcc's default correctly
// rounded division and square root, so results are reproducible bit for bit
// and can be compared exactly with a scalar CPU evaluation of the same
// sequence. Conventions: dot products, cross products and affine transforms
// use the explicit fma chains below; everything else is plain single-rounded
// arithmetic in the order the reference GLSL writes it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptc
{

#define PTC_DEV __device__ __forceinline__

constexpr float kTwoPi = 6.28318530717958647692f; // 2.0 * PI, shader_common.glsl:2
constexpr float kPi = 3.14159265358979323846f;

struct vec3
{
    float x, y, z;
};

PTC_DEV vec3 v3(float x, float y, float z)
{
    vec3 v;
    v.x = x;
    v.y = y;
    v.z = z;
    return v;
}
PTC_DEV vec3 operator+(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PTC_DEV vec3 operator-(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PTC_DEV vec3 operator*(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PTC_DEV vec3 operator*(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PTC_DEV vec3 operator/(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PTC_DEV vec3 operator-(vec3 a) { return v3(-a.x, -a.y, -a.z); }

PTC_DEV float fmin_(float a, float b) { return a < b ? a : b; }
PTC_DEV float fmax_(float a, float b) { return a > b ? a : b; }

PTC_DEV float dot(vec3 a, vec3 b)
{
    return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}

PTC_DEV vec3 cross(vec3 a, vec3 b)
{
    return v3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)),
              __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
              __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}

// GLSL normalize(v) = v * (1 / sqrt(dot(v, v)))
PTC_DEV vec3 normalize(vec3 v)
{
    const float inv = 1.0f / __builtin_sqrtf(dot(v, v));
    return v * inv;
}

// GLSL reflect(I, N) = I - 2 * dot(N, I) * N
PTC_DEV vec3 reflect(vec3 i, vec3 n)
{
    const float k = 2.0f * dot(n, i);
    return i - n * k;
}

// GLSL faceforward(N, I, Nref) = dot(Nref, I) < 0 ? N : -N
PTC_DEV vec3 faceforward(vec3 n, vec3 i, vec3 nref)
{
    return dot(nref, i) < 0.0f ? n : -n;
}

// 3x4 row-major affine matrix times (p, 1)
PTC_DEV vec3 xform_point(const float *__restrict__ m, vec3 p)
{
    vec3 r;
    r.x = __builtin_fmaf(m[2], p.z, __builtin_fmaf(m[1], p.y, m[0] * p.x)) + m[3];
    r.y = __builtin_fmaf(m[6], p.z, __builtin_fmaf(m[5], p.y, m[4] * p.x)) + m[7];
    r.z = __builtin_fmaf(m[10], p.z, __builtin_fmaf(m[9], p.y, m[8] * p.x)) + m[11];
    return r;
}

// linear part of the 3x4 matrix m times d (a direction: no translation)
PTC_DEV vec3 xform_vector(const float *__restrict__ m, vec3 d)
{
    vec3 r;
    r.x = __builtin_fmaf(m[2], d.z, __builtin_fmaf(m[1], d.y, m[0] * d.x));
    r.y = __builtin_fmaf(m[6], d.z, __builtin_fmaf(m[5], d.y, m[4] * d.x));
    r.z = __builtin_fmaf(m[10], d.z, __builtin_fmaf(m[9], d.y, m[8] * d.x));
    return r;
}

// row vector n times the 3x3 linear part of the 3x4 matrix m
// (GLSL "vec3 * mat4x3 .xyz", closest_hit_common.glsl:82)
PTC_DEV vec3 xform_normal_rowvec(const float *__restrict__ m, vec3 n)
{
    vec3 r;
    r.x = __builtin_fmaf(n.z, m[8], __builtin_fmaf(n.y, m[4], n.x * m[0]));
    r.y = __builtin_fmaf(n.z, m[9], __builtin_fmaf(n.y, m[5], n.x * m[1]));
    r.z = __builtin_fmaf(n.z, m[10], __builtin_fmaf(n.y, m[6], n.x * m[2]));
    return r;
}

// ---- transcendentals ------------------------------------------------------
// GLSL leaves the precision of log/sin/cos/acos/atan to the driver; these are
// Cephes-style reductions with minimax polynomials (about 1-2 ulp).

// sin and cos of theta, |theta| <= ~8
PTC_DEV void sincos_(float theta, float &s, float &c)
{
    const float kf = __builtin_rintf(theta * 0.636619772367581343f);
    const int k = (int)kf;
    float r = __builtin_fmaf(kf, -1.5703125f, theta);
    r = __builtin_fmaf(kf, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(kf, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float sp = __builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    sp = __builtin_fmaf(sp, r2, -1.6666654611e-1f);
    sp = __builtin_fmaf(sp * r2, r, r);
    float cp = __builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    cp = __builtin_fmaf(cp, r2, 4.166664568298827e-2f);
    cp = __builtin_fmaf(cp * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    const int q = k & 3;
    const float a = (q & 1) ? cp : sp;
    const float b = (q & 1) ? sp : cp;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}

// Only the signs of sincos_(theta)'s results, for theta >= 0: what is left of a
// sine and a cosine that are multiplied by zero. The same reduction as above
// decides them: the quadrant, and for the odd polynomial the sign of the reduced
// argument (its value r + r * (r^2 * p) keeps r's sign; the even one is positive;
// a reduced argument of zero is +0 for theta >= 0).
PTC_DEV void sincos_signs(float theta, bool &s_negative, bool &c_negative)
{
    const float kf = __builtin_rintf(theta * 0.636619772367581343f);
    const int k = (int)kf;
    float r = __builtin_fmaf(kf, -1.5703125f, theta);
    r = __builtin_fmaf(kf, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(kf, -7.54978995489188216e-8f, r);
    const bool r_negative = (__float_as_uint(r) >> 31) != 0u;
    const int q = k & 3;
    s_negative = ((q & 1) ? false : r_negative) != ((q & 2) != 0);
    c_negative = ((q & 1) ? r_negative : false) != (((q + 1) & 2) != 0);
}

// natural log of a normal positive float
PTC_DEV float log_(float x)
{
    const uint32_t ix = __float_as_uint(x);
    int e = (int)(ix >> 23) - 126;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f)
    {
        e -= 1;
        m = (m + m) - 1.0f;
    }
    else
    {
        m = m - 1.0f;
    }
    const float z = m * m;
    float p = __builtin_fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = __builtin_fmaf(p, m, 1.1676998740e-1f);
    p = __builtin_fmaf(p, m, -1.2420140846e-1f);
    p = __builtin_fmaf(p, m, 1.4249322787e-1f);
    p = __builtin_fmaf(p, m, -1.6668057665e-1f);
    p = __builtin_fmaf(p, m, 2.0000714765e-1f);
    p = __builtin_fmaf(p, m, -2.4999993993e-1f);
    p = __builtin_fmaf(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y = (p * m) * z;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

PTC_DEV float asin_core(float z)
{
    const float z2 = z * z;
    float p = __builtin_fmaf(4.2163199048e-2f, z2, 2.4181311049e-2f);
    p = __builtin_fmaf(p, z2, 4.5470025998e-2f);
    p = __builtin_fmaf(p, z2, 7.4953002686e-2f);
    p = __builtin_fmaf(p, z2, 1.6666752422e-1f);
    return __builtin_fmaf(p * z2, z, z);
}

// acos with the argument clamped to [-1, 1]:
//   x >  0.5: 2 asin(sqrt((1 - x) / 2));  x < -0.5: pi - 2 asin(sqrt((1 + x) / 2));
//   else pi / 2 - asin(x).
// (One square root and one polynomial serve the three ranges: 1 + x is the same
// single operation as 1 - |x| for a negative x.)
PTC_DEV float acos_(float x)
{
    if (!(x < 1.0f))
    {
        return 0.0f;
    }
    if (x <= -1.0f)
    {
        return kPi;
    }
    const bool outer = x > 0.5f || x < -0.5f;
    const float root = __builtin_sqrtf(0.5f * (1.0f - __builtin_fabsf(x)));
    const float a = asin_core(outer ? root : x);
    if (x > 0.5f)
    {
        return 2.0f * a;
    }
    if (x < -0.5f)
    {
        return kPi - 2.0f * a;
    }
    return 1.57079632679489661923f - a;
}

// atan of t >= 0: t > tan(3 pi / 8): pi / 2 + atan(-1 / t); t > tan(pi / 8):
// pi / 4 + atan((t - 1) / (t + 1)); one division serves both reductions.
PTC_DEV float atan_core(float t)
{
    const bool high = t > 2.414213562373095f, mid = !high && t > 0.4142135623730950f;
    const float y0 = high ? 1.57079632679489661923f : (mid ? 0.785398163397448309616f : 0.0f);
    const float q = (high ? -1.0f : t - 1.0f) / (high ? t : t + 1.0f);
    t = (high || mid) ? q : t;
    const float z = t * t;
    float p = __builtin_fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = __builtin_fmaf(p, z, 1.99777106478e-1f);
    p = __builtin_fmaf(p, z, -3.33329491539e-1f);
    return y0 + __builtin_fmaf(p * z, t, t);
}

// atan2(y, x); (0, 0) -> 0
PTC_DEV float atan2_(float y, float x)
{
    const float ax = __builtin_fabsf(x);
    const float ay = __builtin_fabsf(y);
    if (ax == 0.0f && ay == 0.0f)
    {
        return 0.0f;
    }
    float r = atan_core(ay / ax);
    if (x < 0.0f)
    {
        r = kPi - r;
    }
    if (y < 0.0f)
    {
        r = -r;
    }
    return r;
}

} // namespace ptc
