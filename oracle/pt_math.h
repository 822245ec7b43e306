/*
 * pt_math.h -- scalar float arithmetic used by the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the shipped library includes, links or
 * calls this file; it exists so the tests can check the HIP path against an
 * independent scalar restatement of the reference shaders.
 *
 * Every function below is a fixed sequence of IEEE-754 binary32 operations
 * (+, -, *, /, sqrt, fma, integer bit operations). Compiled with
 * -ffp-contract=off and without fast-math, the sequence has exactly one result
 * on any conforming machine, which is what lets the tests demand bit equality
 * between this oracle and the gfx950 kernels.
 *
 * Two conventions hold everywhere in the oracle:
 *   - dot products, cross products and affine transforms use the explicit
 *     fma chains defined here (pt_dot, pt_cross, pt_xform_point ...);
 *   - all other arithmetic is plain single-rounded mul/add in the order the
 *     reference GLSL writes it.
 *
 * Transcendentals: GLSL leaves log/sin/cos/acos/atan precision to the driver
 * (reference: src/shaders/shader.rgen:45-58, shader.rmiss:20-21,
 * closest_hit_common.glsl:46-52). The oracle therefore has two modes:
 *   PT_MATH_PORTABLE  polynomial implementations below (Cephes-style
 *                     argument reduction and minimax polynomials, ~1-2 ulp);
 *                     these are what the HIP kernels also evaluate, so the
 *                     two agree bit for bit;
 *   PT_MATH_LIBM      the host C library's sinf/cosf/logf/acosf/atan2f; used
 *                     to show that the converged image does not depend on the
 *                     choice (statistical tolerance test).
 */
#ifndef PT_MATH_H
#define PT_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define PT_MATH_PORTABLE 0
#define PT_MATH_LIBM 1

/* 2.0 * PI with PI = 3.1415926535897931 (shader_common.glsl:2), rounded to
 * binary32 the way a GLSL compiler folds the constant expression. */
#define PT_TWO_PI 6.28318530717958647692f
#define PT_PI 3.14159265358979323846f

typedef struct
{
    float x, y, z;
} pt_vec3;

static inline uint32_t pt_f2u(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float pt_u2f(uint32_t u)
{
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline pt_vec3 pt_v3(float x, float y, float z)
{
    pt_vec3 v = {x, y, z};
    return v;
}
static inline pt_vec3 pt_add(pt_vec3 a, pt_vec3 b)
{
    return pt_v3(a.x + b.x, a.y + b.y, a.z + b.z);
}
static inline pt_vec3 pt_sub(pt_vec3 a, pt_vec3 b)
{
    return pt_v3(a.x - b.x, a.y - b.y, a.z - b.z);
}
static inline pt_vec3 pt_mul(pt_vec3 a, pt_vec3 b)
{
    return pt_v3(a.x * b.x, a.y * b.y, a.z * b.z);
}
static inline pt_vec3 pt_scale(pt_vec3 a, float s)
{
    return pt_v3(a.x * s, a.y * s, a.z * s);
}
static inline pt_vec3 pt_divs(pt_vec3 a, float s)
{
    return pt_v3(a.x / s, a.y / s, a.z / s);
}
static inline pt_vec3 pt_neg(pt_vec3 a)
{
    return pt_v3(-a.x, -a.y, -a.z);
}

/* dot(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)) */
static inline float pt_dot(pt_vec3 a, pt_vec3 b)
{
    return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x));
}

/* cross(a,b), each component as fma(p, q, -(r*s)) */
static inline pt_vec3 pt_cross(pt_vec3 a, pt_vec3 b)
{
    return pt_v3(fmaf(a.y, b.z, -(a.z * b.y)),
                 fmaf(a.z, b.x, -(a.x * b.z)),
                 fmaf(a.x, b.y, -(a.y * b.x)));
}

/* GLSL normalize(v): v * (1 / sqrt(dot(v,v))) */
static inline pt_vec3 pt_normalize(pt_vec3 v)
{
    const float inv = 1.0f / sqrtf(pt_dot(v, v));
    return pt_scale(v, inv);
}

/* GLSL reflect(I,N) = I - 2*dot(N,I)*N */
static inline pt_vec3 pt_reflect(pt_vec3 i, pt_vec3 n)
{
    const float k = 2.0f * pt_dot(n, i);
    return pt_sub(i, pt_scale(n, k));
}

/* GLSL faceforward(N,I,Nref) = dot(Nref,I) < 0 ? N : -N */
static inline pt_vec3 pt_faceforward(pt_vec3 n, pt_vec3 i, pt_vec3 nref)
{
    return pt_dot(nref, i) < 0.0f ? n : pt_neg(n);
}

/* 3x4 row-major affine matrix m (VkTransformMatrixKHR layout) times (p,1) */
static inline pt_vec3 pt_xform_point(const float *m, pt_vec3 p)
{
    pt_vec3 r;
    r.x = fmaf(m[2], p.z, fmaf(m[1], p.y, m[0] * p.x)) + m[3];
    r.y = fmaf(m[6], p.z, fmaf(m[5], p.y, m[4] * p.x)) + m[7];
    r.z = fmaf(m[10], p.z, fmaf(m[9], p.y, m[8] * p.x)) + m[11];
    return r;
}

/* linear part of the 3x4 matrix m times d (a direction: no translation) */
static inline pt_vec3 pt_xform_vector(const float *m, pt_vec3 d)
{
    pt_vec3 r;
    r.x = fmaf(m[2], d.z, fmaf(m[1], d.y, m[0] * d.x));
    r.y = fmaf(m[6], d.z, fmaf(m[5], d.y, m[4] * d.x));
    r.z = fmaf(m[10], d.z, fmaf(m[9], d.y, m[8] * d.x));
    return r;
}

/* row vector n times the 3x3 linear part of the 3x4 matrix m:
 * r[c] = n.x*m[0][c] + n.y*m[1][c] + n.z*m[2][c]
 * (GLSL "vec3 * mat4x3 .xyz", closest_hit_common.glsl:82) */
static inline pt_vec3 pt_xform_normal_rowvec(const float *m, pt_vec3 n)
{
    pt_vec3 r;
    r.x = fmaf(n.z, m[8], fmaf(n.y, m[4], n.x * m[0]));
    r.y = fmaf(n.z, m[9], fmaf(n.y, m[5], n.x * m[1]));
    r.z = fmaf(n.z, m[10], fmaf(n.y, m[6], n.x * m[2]));
    return r;
}

static inline float pt_min(float a, float b)
{
    return a < b ? a : b;
}
static inline float pt_max(float a, float b)
{
    return a > b ? a : b;
}

/* ------------------------------------------------------------------------ */
/* Portable transcendentals                                                 */
/* ------------------------------------------------------------------------ */

/* sin and cos of theta for |theta| <= ~8 (callers pass 2*pi*u, u in [0,1]).
 * k = rint(theta * 2/pi); three-term Cody-Waite reduction by pi/2; degree-7 /
 * degree-8 minimax polynomials on [-pi/4, pi/4]; quadrant fix-up. */
static inline void pt_sincos_portable(float theta, float *s, float *c)
{
    const float kf = rintf(theta * 0.636619772367581343f);
    const int k = (int)kf;
    float r = fmaf(kf, -1.5703125f, theta);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float sp = fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    switch (k & 3)
    {
    case 0: *s = sp; *c = cp; break;
    case 1: *s = cp; *c = -sp; break;
    case 2: *s = -sp; *c = -cp; break;
    default: *s = -cp; *c = sp; break;
    }
}

/* natural log for normal positive x (callers pass max(FLT_MIN, u), u <= 1) */
static inline float pt_log_portable(float x)
{
    const uint32_t ix = pt_f2u(x);
    int e = (int)(ix >> 23) - 126;
    float m = pt_u2f((ix & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
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
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    const float fe = (float)e;
    float y = (p * m) * z;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(0.693359375f, fe, r);
    return r;
}

/* asin kernel for |z| <= 0.5 */
static inline float pt_asin_core(float z)
{
    const float z2 = z * z;
    float p = fmaf(4.2163199048e-2f, z2, 2.4181311049e-2f);
    p = fmaf(p, z2, 4.5470025998e-2f);
    p = fmaf(p, z2, 7.4953002686e-2f);
    p = fmaf(p, z2, 1.6666752422e-1f);
    return fmaf(p * z2, z, z);
}

/* acos with the argument clamped to [-1, 1] */
static inline float pt_acos_portable(float x)
{
    if (!(x < 1.0f)) /* also catches NaN */
    {
        return 0.0f;
    }
    if (x <= -1.0f)
    {
        return PT_PI;
    }
    if (x > 0.5f)
    {
        return 2.0f * pt_asin_core(sqrtf(0.5f * (1.0f - x)));
    }
    if (x < -0.5f)
    {
        return PT_PI - 2.0f * pt_asin_core(sqrtf(0.5f * (1.0f + x)));
    }
    return 1.57079632679489661923f - pt_asin_core(x);
}

/* atan for t >= 0 */
static inline float pt_atan_core(float t)
{
    float y0;
    if (t > 2.414213562373095f)
    {
        y0 = 1.57079632679489661923f;
        t = -1.0f / t;
    }
    else if (t > 0.4142135623730950f)
    {
        y0 = 0.785398163397448309616f;
        t = (t - 1.0f) / (t + 1.0f);
    }
    else
    {
        y0 = 0.0f;
    }
    const float z = t * t;
    float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = fmaf(p, z, 1.99777106478e-1f);
    p = fmaf(p, z, -3.33329491539e-1f);
    return y0 + fmaf(p * z, t, t);
}

/* atan2(y, x); (0,0) -> 0 */
static inline float pt_atan2_portable(float y, float x)
{
    const float ax = fabsf(x);
    const float ay = fabsf(y);
    if (ax == 0.0f && ay == 0.0f)
    {
        return 0.0f;
    }
    float r = pt_atan_core(ay / ax);
    if (x < 0.0f)
    {
        r = PT_PI - r;
    }
    if (y < 0.0f)
    {
        r = -r;
    }
    return r;
}

#endif /* PT_MATH_H */
