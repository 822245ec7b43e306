// Wide-BVH builder for gfx950.
//
// Stands in for the driver-side acceleration-structure build that the
// reference requests through vkCmdBuildAccelerationStructuresKHR
// (src/renderer.cpp:1044-1142 BLAS, :1144-1281 TLAS; PreferFastTrace, opaque,
// no culling). Instances are flattened: every (instance, primitive) becomes one
// world-space triangle, so a single tree answers what the reference's
// TLAS + BLAS pair answers.
//
// Pipeline, all on the device:
//   1. flatten      world-space (v0, e1, e2) per triangle, padded AABB, scene bounds
//   2. morton       63-bit Morton code of the AABB centre, radix sort (rocPRIM)
//   3. ploc         binary tree by parallel locally-ordered clustering: every
//                   round each cluster finds, within +-R neighbours along the
//                   Morton curve, the partner whose merged box has the smallest
//                   surface area; mutual choices merge. This is the agglomerative
//                   form of the surface-area heuristic.
//   3b. reinsertion a few rounds in which every subtree looks through the whole tree
//                   for the place where it would add the least surface area, and
//                   the moves that do not touch each other are carried out
//                   (k_opt_*): the clustering only ever saw neighbours along the
//                   Morton curve
//   4. collapse     which binary nodes become wide nodes, and which eight
//                   subtrees become the children of each, is chosen bottom-up by
//                   dynamic programming over the surface-area cost of the wide
//                   tree (k_collapse_cost), then followed top-down, one wide level
//                   per launch; subtrees of <= 3 triangles become leaf slots;
//                   children are placed in slots so that (slot XOR ray octant)
//                   orders them front to back. (collapse = 1 keeps the greedy
//                   rule: open the child of largest area until 8 slots are used.)
//   5. quantise     child boxes to 8 bits on a per-node power-of-two grid,
//                   rounded outward against the float decode the traversal uses
#include <string.h>

#include "ptc_internal.h"

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ptc
{
namespace
{

constexpr int kBlock = 256;
constexpr int kMaxRadius = 32;
constexpr uint32_t kLeafMax = 3;
constexpr float kDefaultTriCost = 0.8f;

__device__ __forceinline__ uint32_t f2ord(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o)
{
    const uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
    return __uint_as_float(u);
}

__device__ __forceinline__ float wave_min(float v)
{
    for (int off = 32; off > 0; off >>= 1)
    {
        v = fmin_(v, __shfl_xor(v, off, 64));
    }
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    for (int off = 32; off > 0; off >>= 1)
    {
        v = fmax_(v, __shfl_xor(v, off, 64));
    }
    return v;
}

// ---- 1. flatten -------------------------------------------------------------

__global__ void __launch_bounds__(kBlock)
k_flatten(const DInstance *__restrict__ inst, uint32_t n_inst,
          const float *__restrict__ positions, const uint32_t *__restrict__ indices,
          uint32_t n_flat, float4 *__restrict__ tri_tmp, float4 *__restrict__ box_lo,
          float4 *__restrict__ box_hi, uint32_t *__restrict__ bounds)
{
    float lx = INFINITY, ly = INFINITY, lz = INFINITY;
    float hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
    for (uint32_t g = blockIdx.x * kBlock + threadIdx.x; g < n_flat; g += gridDim.x * kBlock)
    {
        uint32_t lo = 0, hi = n_inst - 1;
        while (lo < hi)
        {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (inst[mid].first_flat <= g)
            {
                lo = mid;
            }
            else
            {
                hi = mid - 1;
            }
        }
        const DInstance *__restrict__ I = &inst[lo];
        const uint32_t prim = g - I->first_flat;
        const uint32_t *__restrict__ idx = indices + 3u * (size_t)(I->tri_offset + prim);
        const vec3 a = xform_point(I->o2w, load3(positions + 3u * (size_t)(idx[0] + I->vtx_offset)));
        const vec3 b = xform_point(I->o2w, load3(positions + 3u * (size_t)(idx[1] + I->vtx_offset)));
        const vec3 c = xform_point(I->o2w, load3(positions + 3u * (size_t)(idx[2] + I->vtx_offset)));
        const vec3 e1 = b - a;
        const vec3 e2 = c - a;
        tri_tmp[3u * (size_t)g + 0] = make_float4(a.x, a.y, a.z, __uint_as_float(lo));
        tri_tmp[3u * (size_t)g + 1] = make_float4(e1.x, e1.y, e1.z, __uint_as_float(prim));
        tri_tmp[3u * (size_t)g + 2] = make_float4(e2.x, e2.y, e2.z, __uint_as_float(g));
        const float tlx = fmin_(a.x, fmin_(b.x, c.x)), tly = fmin_(a.y, fmin_(b.y, c.y));
        const float tlz = fmin_(a.z, fmin_(b.z, c.z)), thx = fmax_(a.x, fmax_(b.x, c.x));
        const float thy = fmax_(a.y, fmax_(b.y, c.y)), thz = fmax_(a.z, fmax_(b.z, c.z));
        box_lo[g] = make_float4(tlx, tly, tlz, 0.0f);
        box_hi[g] = make_float4(thx, thy, thz, 0.0f);
        lx = fmin_(lx, tlx);
        ly = fmin_(ly, tly);
        lz = fmin_(lz, tlz);
        hx = fmax_(hx, thx);
        hy = fmax_(hy, thy);
        hz = fmax_(hz, thz);
    }
    // scene bounds: wave reduction, then the block's waves through LDS, then six
    // atomics per block (a bounded number of blocks, see the launch)
    __shared__ float s_red[6][kBlock / 64];
    lx = wave_min(lx);
    ly = wave_min(ly);
    lz = wave_min(lz);
    hx = wave_max(hx);
    hy = wave_max(hy);
    hz = wave_max(hz);
    if ((threadIdx.x & 63) == 0)
    {
        const int wv = threadIdx.x >> 6;
        s_red[0][wv] = lx;
        s_red[1][wv] = ly;
        s_red[2][wv] = lz;
        s_red[3][wv] = hx;
        s_red[4][wv] = hy;
        s_red[5][wv] = hz;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        for (int wv = 1; wv < kBlock / 64; ++wv)
        {
            lx = fmin_(lx, s_red[0][wv]);
            ly = fmin_(ly, s_red[1][wv]);
            lz = fmin_(lz, s_red[2][wv]);
            hx = fmax_(hx, s_red[3][wv]);
            hy = fmax_(hy, s_red[4][wv]);
            hz = fmax_(hz, s_red[5][wv]);
        }
    }
    if (threadIdx.x == 0 && lx <= hx)
    {
        atomicMin(&bounds[0], f2ord(lx));
        atomicMin(&bounds[1], f2ord(ly));
        atomicMin(&bounds[2], f2ord(lz));
        atomicMax(&bounds[3], f2ord(hx));
        atomicMax(&bounds[4], f2ord(hy));
        atomicMax(&bounds[5], f2ord(hz));
    }
}

// ---- 2. pad boxes, Morton keys ----------------------------------------------

__device__ __forceinline__ unsigned long long expand21(uint32_t v)
{
    unsigned long long x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x1f00000000ffffULL;
    x = (x | (x << 16)) & 0x1f0000ff0000ffULL;
    x = (x | (x << 8)) & 0x100f00f00f00f00fULL;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ULL;
    x = (x | (x << 2)) & 0x1249249249249249ULL;
    return x;
}

__device__ __forceinline__ uint32_t quant21(float c, float lo, float hi)
{
    const float ext = hi - lo;
    if (!(ext > 0.0f))
    {
        return 0u;
    }
    float t = (c - lo) / ext * 2097152.0f;
    t = fmax_(0.0f, fmin_(t, 2097151.0f));
    return (uint32_t)t;
}

__global__ void __launch_bounds__(kBlock)
k_morton(uint32_t n, float4 *__restrict__ box_lo, float4 *__restrict__ box_hi,
         const uint32_t *__restrict__ bounds, unsigned long long *__restrict__ keys,
         uint32_t *__restrict__ vals)
{
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n)
    {
        return;
    }
    const float slx = ord2f(bounds[0]), sly = ord2f(bounds[1]), slz = ord2f(bounds[2]);
    const float shx = ord2f(bounds[3]), shy = ord2f(bounds[4]), shz = ord2f(bounds[5]);
    float mag = fmax_(__builtin_fabsf(slx), __builtin_fabsf(shx));
    mag = fmax_(mag, fmax_(__builtin_fabsf(sly), __builtin_fabsf(shy)));
    mag = fmax_(mag, fmax_(__builtin_fabsf(slz), __builtin_fabsf(shz)));
    // absolute padding that covers the rounding slop of the triangle test
    const float pad = mag * (1.0f / 262144.0f);
    float4 lo = box_lo[g], hi = box_hi[g];
    const float cx = 0.5f * (lo.x + hi.x), cy = 0.5f * (lo.y + hi.y), cz = 0.5f * (lo.z + hi.z);
    lo.x -= pad;
    lo.y -= pad;
    lo.z -= pad;
    hi.x += pad;
    hi.y += pad;
    hi.z += pad;
    box_lo[g] = lo;
    box_hi[g] = hi;
    const unsigned long long key = (expand21(quant21(cx, slx, shx)) << 2) |
                                   (expand21(quant21(cy, sly, shy)) << 1) |
                                   expand21(quant21(cz, slz, shz));
    keys[g] = key;
    vals[g] = g;
}

// ---- 3. PLOC ------------------------------------------------------------------
// Binary nodes: ids [0, n) are leaves in Morton order, ids >= n are internal.
// node_lo[id] = (box min, subtree triangle count as bits)
// node_hi[id] = (box max, flat triangle index as bits for leaves)
// node_lr[id - n] = (left, right) for internal nodes

__global__ void __launch_bounds__(kBlock)
k_init_clusters(uint32_t n, const uint32_t *__restrict__ sorted_vals,
                const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi,
                float4 *__restrict__ node_lo, float4 *__restrict__ node_hi,
                uint32_t *__restrict__ cl_id, float4 *__restrict__ cl_lo,
                float4 *__restrict__ cl_hi)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const uint32_t p = sorted_vals[i];
    float4 lo = box_lo[p], hi = box_hi[p];
    lo.w = __uint_as_float(1u);
    hi.w = __uint_as_float(p);
    node_lo[i] = lo;
    node_hi[i] = hi;
    cl_id[i] = i;
    cl_lo[i] = lo;
    cl_hi[i] = hi;
}

__device__ __forceinline__ float merged_half_area(float4 alo, float4 ahi, float4 blo, float4 bhi)
{
    const float dx = fmax_(ahi.x, bhi.x) - fmin_(alo.x, blo.x);
    const float dy = fmax_(ahi.y, bhi.y) - fmin_(alo.y, blo.y);
    const float dz = fmax_(ahi.z, bhi.z) - fmin_(alo.z, blo.z);
    return dx * dy + dy * dz + dz * dx;
}

// Order of a cluster's candidate partners: smaller merged area first; among equal areas the nearer index, and of
// the two at one distance d the one whose pair starts at an even multiple of d. The last two keys depend on the
// pair alone, not on which end looks, so clusters with equal boxes (coincident triangles, lattices) still find
// mutual partners in pairs, (0,1), (2,3) ..., instead of all pointing the same way and merging one pair a round.
struct Partner
{
    float area;
    int j;
    uint32_t rank; // (distance << 1) | odd-multiple bit
};
__device__ __forceinline__ void consider_partner(Partner &best, int i, int j, float area)
{
    const uint32_t d = (uint32_t)(i < j ? j - i : i - j), lower = (uint32_t)(i < j ? i : j);
    const uint32_t rank = (d << 1) | ((lower / d) & 1u);
    if (area < best.area || (area == best.area && rank < best.rank))
    {
        best.area = area;
        best.j = j;
        best.rank = rank;
    }
}

__global__ void __launch_bounds__(kBlock)
k_nearest(uint32_t n, int radius, const float4 *__restrict__ cl_lo,
          const float4 *__restrict__ cl_hi, int *__restrict__ nn)
{
    __shared__ float4 s_lo[kBlock + 2 * kMaxRadius];
    __shared__ float4 s_hi[kBlock + 2 * kMaxRadius];
    const long long base = (long long)blockIdx.x * kBlock - radius;
    for (int t = threadIdx.x; t < kBlock + 2 * radius; t += kBlock)
    {
        const long long j = base + t;
        if (j >= 0 && j < (long long)n)
        {
            s_lo[t] = cl_lo[j];
            s_hi[t] = cl_hi[j];
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const int me = threadIdx.x + radius;
    const float4 mlo = s_lo[me], mhi = s_hi[me];
    Partner best = {INFINITY, -1, 0u};
    for (int off = -radius; off <= radius; ++off)
    {
        const long long j = (long long)i + off;
        if (off == 0 || j < 0 || j >= (long long)n)
        {
            continue;
        }
        consider_partner(best, (int)i, (int)j, merged_half_area(mlo, mhi, s_lo[me + off], s_hi[me + off]));
    }
    nn[i] = best.j;
}

__global__ void __launch_bounds__(kBlock)
k_merge_flags(uint32_t n, const int *__restrict__ nn, unsigned long long *__restrict__ flags)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const int j = nn[i];
    const bool mutual = j >= 0 && nn[j] == (int)i;
    const bool merge = mutual && (int)i < j;
    const bool valid = !mutual || merge;
    flags[i] = ((unsigned long long)(merge ? 1u : 0u) << 32) | (valid ? 1u : 0u);
}

__global__ void __launch_bounds__(kBlock)
k_merge_emit(uint32_t n, uint32_t n_leaves, uint32_t next_node, const int *__restrict__ nn,
             const unsigned long long *__restrict__ flags,
             const unsigned long long *__restrict__ scan, const uint32_t *__restrict__ in_id,
             const float4 *__restrict__ in_lo, const float4 *__restrict__ in_hi,
             uint32_t *__restrict__ out_id, float4 *__restrict__ out_lo,
             float4 *__restrict__ out_hi, float4 *__restrict__ node_lo,
             float4 *__restrict__ node_hi, uint2 *__restrict__ node_lr,
             uint32_t *__restrict__ totals)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    const unsigned long long f = flags[i], s = scan[i];
    if (i == n - 1)
    {
        totals[0] = (uint32_t)(s & 0xffffffffu) + (uint32_t)(f & 0xffffffffu); // clusters left
        totals[1] = (uint32_t)(s >> 32) + (uint32_t)(f >> 32);                 // merges done
    }
    if (!(f & 1u))
    {
        return;
    }
    const uint32_t pos = (uint32_t)(s & 0xffffffffu);
    if (f >> 32)
    {
        const int j = nn[i];
        const float4 alo = in_lo[i], ahi = in_hi[i], blo = in_lo[j], bhi = in_hi[j];
        const uint32_t id = next_node + (uint32_t)(s >> 32);
        float4 lo, hi;
        lo.x = fmin_(alo.x, blo.x);
        lo.y = fmin_(alo.y, blo.y);
        lo.z = fmin_(alo.z, blo.z);
        hi.x = fmax_(ahi.x, bhi.x);
        hi.y = fmax_(ahi.y, bhi.y);
        hi.z = fmax_(ahi.z, bhi.z);
        lo.w = __uint_as_float(__float_as_uint(alo.w) + __float_as_uint(blo.w));
        hi.w = 0.0f;
        node_lo[id] = lo;
        node_hi[id] = hi;
        node_lr[id - n_leaves] = make_uint2(in_id[i], in_id[j]);
        out_id[pos] = id;
        out_lo[pos] = lo;
        out_hi[pos] = hi;
    }
    else
    {
        out_id[pos] = in_id[i];
        out_lo[pos] = in_lo[i];
        out_hi[pos] = in_hi[i];
    }
}

// The last clustering rounds in one launch. Once the clusters fit one block
// the per-round work is a few microseconds, and a round of the general path
// (three kernels, a device scan and a host read-back of the new count) costs
// two orders of magnitude more. One block keeps the clusters in LDS and runs
// the same round — nearest neighbour within the radius, mutual pairs merge,
// survivors compact in order — until one cluster is left, producing exactly
// the tree the general path would. totals[1] receives the next free node id.
constexpr int kTailMax = 512;

__global__ void __launch_bounds__(kTailMax)
k_ploc_tail(uint32_t n, uint32_t n_leaves, uint32_t next_node, int radius,
            const uint32_t *__restrict__ in_id, const float4 *__restrict__ in_lo,
            const float4 *__restrict__ in_hi, float4 *__restrict__ node_lo,
            float4 *__restrict__ node_hi, uint2 *__restrict__ node_lr, uint32_t *__restrict__ totals)
{
    __shared__ float4 s_lo[2][kTailMax];
    __shared__ float4 s_hi[2][kTailMax];
    __shared__ uint32_t s_id[2][kTailMax];
    __shared__ int s_nn[kTailMax];
    __shared__ uint32_t s_scan[2][kTailMax]; // (merges << 16) | survivors, inclusive
    const uint32_t i = threadIdx.x;
    if (i < n)
    {
        s_lo[0][i] = in_lo[i];
        s_hi[0][i] = in_hi[i];
        s_id[0][i] = in_id[i];
    }
    int cur = 0;
    __syncthreads();
    while (n > 1u)
    {
        if (i < n)
        {
            const float4 mlo = s_lo[cur][i], mhi = s_hi[cur][i];
            Partner best = {INFINITY, -1, 0u};
            for (int off = -radius; off <= radius; ++off)
            {
                const int j = (int)i + off;
                if (off == 0 || j < 0 || j >= (int)n)
                {
                    continue;
                }
                consider_partner(best, (int)i, j, merged_half_area(mlo, mhi, s_lo[cur][j], s_hi[cur][j]));
            }
            s_nn[i] = best.j;
        }
        __syncthreads();
        bool merge = false, valid = false;
        int j = -1;
        if (i < n)
        {
            j = s_nn[i];
            const bool mutual = j >= 0 && s_nn[j] == (int)i;
            merge = mutual && (int)i < j;
            valid = !mutual || merge;
        }
        const uint32_t mine = ((merge ? 1u : 0u) << 16) | (valid ? 1u : 0u);
        int sb = 0;
        s_scan[0][i] = mine;
        __syncthreads();
        for (uint32_t step = 1; step < (uint32_t)kTailMax; step <<= 1)
        {
            const uint32_t v = s_scan[sb][i] + (i >= step ? s_scan[sb][i - step] : 0u);
            s_scan[sb ^ 1][i] = v;
            sb ^= 1;
            __syncthreads();
        }
        const uint32_t incl = s_scan[sb][i], total = s_scan[sb][kTailMax - 1];
        const uint32_t excl = incl - mine;
        if (valid)
        {
            const uint32_t pos = excl & 0xffffu;
            if (merge)
            {
                const float4 alo = s_lo[cur][i], ahi = s_hi[cur][i], blo = s_lo[cur][j], bhi = s_hi[cur][j];
                const uint32_t id = next_node + (excl >> 16);
                float4 lo, hi;
                lo.x = fmin_(alo.x, blo.x);
                lo.y = fmin_(alo.y, blo.y);
                lo.z = fmin_(alo.z, blo.z);
                hi.x = fmax_(ahi.x, bhi.x);
                hi.y = fmax_(ahi.y, bhi.y);
                hi.z = fmax_(ahi.z, bhi.z);
                lo.w = __uint_as_float(__float_as_uint(alo.w) + __float_as_uint(blo.w));
                hi.w = 0.0f;
                node_lo[id] = lo;
                node_hi[id] = hi;
                node_lr[id - n_leaves] = make_uint2(s_id[cur][i], s_id[cur][j]);
                s_id[cur ^ 1][pos] = id;
                s_lo[cur ^ 1][pos] = lo;
                s_hi[cur ^ 1][pos] = hi;
            }
            else
            {
                s_id[cur ^ 1][pos] = s_id[cur][i];
                s_lo[cur ^ 1][pos] = s_lo[cur][i];
                s_hi[cur ^ 1][pos] = s_hi[cur][i];
            }
        }
        __syncthreads();
        if ((total >> 16) == 0u)
        {
            break; // no mutual pair: cannot happen for n > 1, but never spin
        }
        next_node += total >> 16;
        n = total & 0xffffu;
        cur ^= 1;
    }
    if (i == 0u)
    {
        totals[0] = n;
        totals[1] = next_node;
    }
}

__device__ __forceinline__ float half_area(float4 lo, float4 hi)
{
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}

// ---- 3b. reinsertion -----------------------------------------------------------
// The clustering only ever merges neighbours along the Morton curve, so a subtree
// can end up far (in the tree) from the place where it would add the least
// surface area. Every round, every node `in` looks for the node X next to which
// its subtree would be better placed: taking `in` out removes its parent P (the
// sibling S takes P's place) and shrinks the boxes above; putting it back beside
// X adds a node over (X, in) and grows the boxes above X up to the first common
// ancestor (the pivot), above which nothing changes. The search climbs from P
// to the root and, per ancestor, walks the subtree on the other side with the
// bound "no node below X can gain more than what is left after the growth X's
// own box causes" (after Meister and Bittner, "Parallel Reinsertion for Bounding
// Volume Hierarchy Optimization", 2018). Moves are settled in three steps. A
// move claims, with an atomic maximum of (gain, node), the nodes whose links it
// rewrites or follows: in, P, S and X (a link of an unclaimed node, the slot of
// P's parent or of X's parent, is only ever rewritten by the move that holds
// the node the slot points at). The moves that hold all four then mark X's
// ancestors below the pivot, and a move whose `in` carries such a mark stands
// back: going up from a moved P, the only way to come down again into a subtree
// that was moved as well is through an `in` that is an ancestor of some X, and
// such an `in` at or above that move's pivot has a subtree strictly larger than
// the one it was reached from, so the moves that remain cannot close a cycle:
// they are the same as carried out one after the other. Boxes and triangle
// counts are refitted after every round.

constexpr uint32_t kNoNode = 0xffffffffu;
constexpr int kOptStack = 64;
constexpr uint32_t kDefaultOptimizeRounds = 8u, kMaxOptimizeRounds = 256u;

// up[id] = (parent, sibling) of every node, from the child links; the root's
// parent is kNoNode. Rebuilt before every round, so the moves only rewrite
// child links.
__global__ void __launch_bounds__(kBlock)
k_opt_links(uint32_t n_leaves, uint32_t n_internal, const uint2 *__restrict__ node_lr, uint2 *__restrict__ up)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_internal)
    {
        return;
    }
    const uint2 lr = node_lr[i];
    up[lr.x] = make_uint2(n_leaves + i, lr.y);
    up[lr.y] = make_uint2(n_leaves + i, lr.x);
    if (i == n_internal - 1u)
    {
        up[n_leaves + i] = make_uint2(kNoNode, kNoNode); // the root is the last node made
    }
}

__device__ __forceinline__ unsigned long long opt_key(float gain, uint32_t in)
{
    // gains are positive floats, whose bits order as integers; ties go to the larger id
    return ((unsigned long long)__float_as_uint(gain) << 32) | in;
}

__global__ void __launch_bounds__(kBlock)
k_opt_find(uint32_t n_leaves, uint32_t n_nodes, const float4 *__restrict__ node_lo,
           const float4 *__restrict__ node_hi, const uint2 *__restrict__ node_lr,
           const uint2 *__restrict__ up, uint32_t *__restrict__ cand_x,
           uint32_t *__restrict__ cand_pivot, float *__restrict__ cand_gain,
           unsigned long long *__restrict__ lock, uint32_t *__restrict__ n_cand)
{
    const uint32_t in = blockIdx.x * kBlock + threadIdx.x;
    if (in >= n_nodes)
    {
        return;
    }
    const uint32_t root = n_nodes - 1u;
    cand_gain[in] = 0.0f;
    if (in == root)
    {
        return;
    }
    const uint32_t P = up[in].x, S = up[in].y;
    if (P == root)
    {
        return; // (the root stays: its children are not moved)
    }
    const float4 ilo = node_lo[in], ihi = node_hi[in];
    // Two boxes go up the path with the search: the path node's own (the union
    // of all below it, which is what the refit stored there) and the same
    // without `in`. A level then reads the other child's box and nothing else.
    float4 plo = node_lo[S], phi = node_hi[S];
    float4 olo = plo, ohi = phi;
    auto widen = [](float4 &lo, float4 &hi, float4 blo, float4 bhi) {
        lo.x = fmin_(lo.x, blo.x);
        lo.y = fmin_(lo.y, blo.y);
        lo.z = fmin_(lo.z, blo.z);
        hi.x = fmax_(hi.x, bhi.x);
        hi.y = fmax_(hi.y, bhi.y);
        hi.z = fmax_(hi.z, bhi.z);
    };
    widen(olo, ohi, ilo, ihi);
    // what taking `in` out has gained so far on the way up: P itself, then the
    // shrinking of every ancestor passed
    float D = half_area(olo, ohi);
    float best_gain = 0.0f;
    uint32_t best_x = kNoNode, best_pivot = kNoNode;
    uint32_t st_node[kOptStack];
    float st_ind[kOptStack];
    // candidates in the subtree of `start` (whose box the caller has at hand)
    auto search = [&](uint32_t start, float4 slo, float4 shi, uint32_t pivot) {
        int sp = 0;
        uint32_t X = start;
        float ind = 0.0f; // growth of X's ancestors below the pivot
        float4 xlo = slo, xhi = shi;
        for (;;)
        {
            const float m = merged_half_area(xlo, xhi, ilo, ihi);
            const float gain = D - ind - m;
            if (gain > best_gain)
            {
                best_gain = gain;
                best_x = X;
                best_pivot = pivot;
            }
            if (X >= n_leaves)
            {
                const float ind_below = ind + (m - half_area(xlo, xhi));
                // no box inside X's, merged with in's, is smaller than in's box
                // stretched across the gap between the two
                const float gx = fmax_(fmax_(xlo.x - ihi.x, ilo.x - xhi.x), 0.0f);
                const float gy = fmax_(fmax_(xlo.y - ihi.y, ilo.y - xhi.y), 0.0f);
                const float gz = fmax_(fmax_(xlo.z - ihi.z, ilo.z - xhi.z), 0.0f);
                const float ex = (ihi.x - ilo.x) + gx, ey = (ihi.y - ilo.y) + gy, ez = (ihi.z - ilo.z) + gz;
                const float m_below = ex * ey + ey * ez + ez * ex;
                if (D - ind_below - m_below > best_gain && sp + 2 <= kOptStack)
                {
                    const uint2 lr = node_lr[X - n_leaves];
                    st_node[sp] = lr.x;
                    st_ind[sp] = ind_below;
                    st_node[sp + 1] = lr.y;
                    st_ind[sp + 1] = ind_below;
                    sp += 2;
                }
            }
            if (sp == 0)
            {
                break;
            }
            --sp;
            X = st_node[sp];
            ind = st_ind[sp];
            xlo = node_lo[X];
            xhi = node_hi[X];
        }
    };
    // below the sibling first (S itself gives back the tree as it is: gain 0)
    search(S, plo, phi, P);
    // (the links of the level above are requested before a level's subtree is
    // searched: the climb is a chain of dependent loads, and this takes one of
    // them off every level)
    uint2 link = up[P];
    for (;;)
    {
        const uint32_t A = link.x, C = link.y;
        const float4 clo = node_lo[C], chi = node_hi[C];
        if (A != root)
        {
            link = up[A];
        }
        search(C, clo, chi, A);
        if (A == root)
        {
            break;
        }
        widen(plo, phi, clo, chi);
        widen(olo, ohi, clo, chi);
        D += half_area(olo, ohi) - half_area(plo, phi);
    }
    if (best_x == kNoNode || !(best_gain > 0.0f))
    {
        return;
    }
    if (n_cand != nullptr)
    {
        atomicAdd(n_cand, 1u); // (tracing only)
    }
    cand_gain[in] = best_gain;
    cand_x[in] = best_x;
    cand_pivot[in] = best_pivot;
    const unsigned long long key = opt_key(best_gain, in);
    atomicMax(&lock[in], key);
    atomicMax(&lock[P], key);
    atomicMax(&lock[S], key);
    atomicMax(&lock[best_x], key);
}

// Whether the move of `in` holds all its claims; with them, P and S are in's
// parent and sibling as the search saw them.
__device__ __forceinline__ bool opt_holds(uint32_t in, const uint2 *__restrict__ up, const uint32_t *__restrict__ cand_x,
                                          const float *__restrict__ cand_gain,
                                          const unsigned long long *__restrict__ lock, uint32_t &P, uint32_t &S)
{
    const float gain = cand_gain[in];
    if (!(gain > 0.0f))
    {
        return false;
    }
    const unsigned long long key = opt_key(gain, in);
    P = up[in].x;
    S = up[in].y;
    return lock[in] == key && lock[P] == key && lock[S] == key && lock[cand_x[in]] == key;
}

// The moves that hold their claims mark the ancestors of X below the pivot.
__global__ void __launch_bounds__(kBlock)
k_opt_mark(uint32_t n_nodes, const uint2 *__restrict__ up, const uint32_t *__restrict__ cand_x,
           const uint32_t *__restrict__ cand_pivot, const float *__restrict__ cand_gain,
           const unsigned long long *__restrict__ lock, uint32_t *__restrict__ marked)
{
    const uint32_t in = blockIdx.x * kBlock + threadIdx.x;
    uint32_t P, S;
    if (in >= n_nodes || !opt_holds(in, up, cand_x, cand_gain, lock, P, S))
    {
        return;
    }
    const uint32_t pivot = cand_pivot[in];
    for (uint32_t x = up[cand_x[in]].x; x != pivot; x = up[x].x)
    {
        marked[x] = 1u;
    }
}

// Carries out the moves that hold their claims and whose `in` is not marked, on
// the child links alone (up[] is as the round found it, and is rebuilt after).
__global__ void __launch_bounds__(kBlock)
k_opt_apply(uint32_t n_leaves, uint32_t n_nodes, uint2 *__restrict__ node_lr, const uint2 *__restrict__ up,
            const uint32_t *__restrict__ cand_x, const float *__restrict__ cand_gain,
            const unsigned long long *__restrict__ lock, const uint32_t *__restrict__ marked,
            uint32_t *__restrict__ n_moved, bool count_moves)
{
    const uint32_t in = blockIdx.x * kBlock + threadIdx.x;
    uint32_t P = 0u, S = 0u;
    const bool go = in < n_nodes && opt_holds(in, up, cand_x, cand_gain, lock, P, S) && marked[in] == 0u;
    if (!go)
    {
        return;
    }
    // (a flag when the host only wants to know whether the round moved anything:
    // a count costs an atomic on one address per wave, 2 ms for 10 M triangles)
    if (count_moves)
    {
        atomicAdd(n_moved, 1u);
    }
    else
    {
        *n_moved = 1u;
    }
    const uint32_t X = cand_x[in], G = up[P].x, pX = up[X].x;
    // S takes P's place below G
    uint32_t *g = reinterpret_cast<uint32_t *>(&node_lr[G - n_leaves]);
    if (g[0] == P)
    {
        g[0] = S;
    }
    else
    {
        g[1] = S;
    }
    // P goes between X and X's parent (which may be G, or S: the step above is done by then)
    uint32_t *q = reinterpret_cast<uint32_t *>(&node_lr[pX - n_leaves]);
    if (q[0] == X)
    {
        q[0] = P;
    }
    else
    {
        q[1] = P;
    }
    node_lr[P - n_leaves] = make_uint2(in, X);
}

// Boxes and triangle counts of the internal nodes, bottom-up, one launch per
// height: launch number `pass` (1, 2, ...) does the nodes whose children were
// done by earlier launches and leaves its number in height[], so a node of
// height h is done by launch h, from values that launches before it wrote.
// (One launch in which threads climb from the leaves and meet at the nodes needs
// a release and an acquire across the chip's eight L2s at every node: 3.3 ms
// for 871k leaves, against 0.3 ms for some forty of these launches.)
// The root is the last internal node: once its height is set, all are.
__global__ void __launch_bounds__(kBlock)
k_opt_refit_pass(uint32_t pass, uint32_t n_leaves, uint32_t n_internal, const uint2 *__restrict__ node_lr,
                 float4 *__restrict__ node_lo, float4 *__restrict__ node_hi, uint32_t *__restrict__ height)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_internal || height[i] != 0u)
    {
        return;
    }
    const uint2 lr = node_lr[i];
    const uint32_t hl = lr.x < n_leaves ? 0u : height[lr.x - n_leaves];
    const uint32_t hr = lr.y < n_leaves ? 0u : height[lr.y - n_leaves];
    if (!((lr.x < n_leaves || (hl != 0u && hl < pass)) && (lr.y < n_leaves || (hr != 0u && hr < pass))))
    {
        return;
    }
    const float4 alo = node_lo[lr.x], ahi = node_hi[lr.x], blo = node_lo[lr.y], bhi = node_hi[lr.y];
    float4 lo, hi;
    lo.x = fmin_(alo.x, blo.x);
    lo.y = fmin_(alo.y, blo.y);
    lo.z = fmin_(alo.z, blo.z);
    hi.x = fmax_(ahi.x, bhi.x);
    hi.y = fmax_(ahi.y, bhi.y);
    hi.z = fmax_(ahi.z, bhi.z);
    lo.w = __uint_as_float(__float_as_uint(alo.w) + __float_as_uint(blo.w));
    hi.w = 0.0f;
    node_lo[n_leaves + i] = lo;
    node_hi[n_leaves + i] = hi;
    height[i] = pass;
}

// ---- 4./5. collapse to 8-wide nodes, quantise ---------------------------------


// Cost-optimal collapse. For a binary node n and a budget of i subtree roots
// (1 <= i <= 7), cost[i-1] is the least surface-area cost of covering n's
// triangles with at most i children of one wide node:
//   i == 1: n itself is the child: a leaf slot (<= 3 triangles: area * count *
//           tri_cost), or a wide node of its own (area * 1 + the cheapest way
//           of covering its two halves with 8 roots between them);
//   i >= 2: the cheaper of the budget split k : i - k between n's halves and of
//           cost[i-2] (fewer roots than allowed).
// Children have smaller ids than parents (they were clusters of earlier
// rounds), so one launch per clustering round, in round order, sees finished
// children. dec packs the choices: bits 3(i-1)..3(i-1)+2 = roots given to the
// left half for budget i (0: use budget i - 1; for i == 1: 1 = wide node, 0 =
// leaf slot), bits 21..23 = the left half's share of the 8 roots of a wide node
// at n.
struct CollapseCost
{
    float cost[7];
    uint32_t dec;
};

__device__ __forceinline__ void collapse_cost_node(uint32_t id, uint32_t n_leaves, float tri_cost,
                                                   const float4 *__restrict__ node_lo,
                                                   const float4 *__restrict__ node_hi,
                                                   const uint2 *__restrict__ node_lr,
                                                   const CollapseCost &L, const CollapseCost &R,
                                                   CollapseCost &out)
{
    const float4 lo = node_lo[id], hi = node_hi[id];
    const float area = half_area(lo, hi);
    const uint32_t cnt = __float_as_uint(lo.w);
    uint32_t dec = 0u;
    // the 8-way split of a wide node here
    float best8 = INFINITY;
    uint32_t k8 = 1u;
    for (uint32_t k = 1; k <= 7u; ++k)
    {
        const float c = L.cost[k - 1u] + R.cost[7u - k];
        if (c < best8)
        {
            best8 = c;
            k8 = k;
        }
    }
    dec |= k8 << 21;
    if (cnt <= kLeafMax)
    {
        out.cost[0] = area * (float)cnt * tri_cost;
    }
    else
    {
        out.cost[0] = area + best8;
        dec |= 1u;
    }
    for (uint32_t i = 2; i <= 7u; ++i)
    {
        float best = out.cost[i - 2u];
        uint32_t bk = 0u;
        for (uint32_t k = 1; k < i; ++k)
        {
            const float c = L.cost[k - 1u] + R.cost[i - k - 1u];
            if (c < best)
            {
                best = c;
                bk = k;
            }
        }
        out.cost[i - 1u] = best;
        dec |= bk << (3u * (i - 1u));
    }
    out.dec = dec;
}

// cost table of a child: a single triangle costs the same under every budget
__device__ __forceinline__ CollapseCost collapse_cost_of(uint32_t id, uint32_t n_leaves, float tri_cost,
                                                         const float4 *__restrict__ node_lo,
                                                         const float4 *__restrict__ node_hi,
                                                         const CollapseCost *__restrict__ table)
{
    if (id >= n_leaves)
    {
        return table[id - n_leaves];
    }
    CollapseCost c;
    const float a = half_area(node_lo[id], node_hi[id]) * tri_cost;
    for (int i = 0; i < 7; ++i)
    {
        c.cost[i] = a;
    }
    c.dec = 0u;
    return c;
}

// internal binary nodes [first, last): the nodes of one clustering round
__global__ void __launch_bounds__(kBlock)
k_collapse_cost(uint32_t first, uint32_t last, uint32_t n_leaves, float tri_cost,
                const float4 *__restrict__ node_lo, const float4 *__restrict__ node_hi,
                const uint2 *__restrict__ node_lr, CollapseCost *__restrict__ table)
{
    const uint32_t id = first + blockIdx.x * kBlock + threadIdx.x;
    if (id >= last)
    {
        return;
    }
    const uint2 lr = node_lr[id - n_leaves];
    const CollapseCost L = collapse_cost_of(lr.x, n_leaves, tri_cost, node_lo, node_hi, table);
    const CollapseCost R = collapse_cost_of(lr.y, n_leaves, tri_cost, node_lo, node_hi, table);
    CollapseCost out;
    collapse_cost_node(id, n_leaves, tri_cost, node_lo, node_hi, node_lr, L, R, out);
    table[id - n_leaves] = out;
}

// The nodes [first, last) made by the one-block tail of the clustering
// (k_ploc_tail), whose rounds the host does not see: one block, one thread per
// node; a node is costed in the first pass that finds both children done.
// Every pass completes at least the lowest unfinished node, so the loop ends.
__global__ void __launch_bounds__(kTailMax)
k_collapse_cost_tail(uint32_t first, uint32_t last, uint32_t n_leaves, float tri_cost,
                     const float4 *__restrict__ node_lo, const float4 *__restrict__ node_hi,
                     const uint2 *__restrict__ node_lr, CollapseCost *__restrict__ table)
{
    __shared__ CollapseCost s_table[kTailMax];
    __shared__ uint32_t s_done[kTailMax];
    __shared__ uint32_t s_left;
    const uint32_t id = first + threadIdx.x;
    const bool mine = id < last;
    s_done[threadIdx.x] = 0u;
    if (threadIdx.x == 0u)
    {
        s_left = last - first;
    }
    uint2 lr = make_uint2(0u, 0u);
    if (mine)
    {
        lr = node_lr[id - n_leaves];
    }
    bool done = !mine;
    __syncthreads();
    for (uint32_t pass = 0; pass < (uint32_t)kTailMax; ++pass)
    {
        const bool l_ready = lr.x < first || s_done[lr.x - first] != 0u;
        const bool r_ready = lr.y < first || s_done[lr.y - first] != 0u;
        const bool go = !done && l_ready && r_ready;
        CollapseCost out;
        if (go)
        {
            const CollapseCost L = lr.x >= first ? s_table[lr.x - first]
                                                 : collapse_cost_of(lr.x, n_leaves, tri_cost, node_lo, node_hi, table);
            const CollapseCost R = lr.y >= first ? s_table[lr.y - first]
                                                 : collapse_cost_of(lr.y, n_leaves, tri_cost, node_lo, node_hi, table);
            collapse_cost_node(id, n_leaves, tri_cost, node_lo, node_hi, node_lr, L, R, out);
        }
        __syncthreads();
        if (go)
        {
            s_table[threadIdx.x] = out;
            s_done[threadIdx.x] = 1u;
            table[id - n_leaves] = out;
            done = true;
            atomicSub(&s_left, 1u);
        }
        __syncthreads();
        if (s_left == 0u)
        {
            break;
        }
    }
}

// The same table for a tree whose node ids no longer say which nodes are below
// which (after reinsertion): one launch per height, as the refit left it in
// height[] (k_opt_refit_pass).
__global__ void __launch_bounds__(kBlock)
k_collapse_cost_height(uint32_t pass, uint32_t n_leaves, uint32_t n_internal, float tri_cost,
                       const float4 *__restrict__ node_lo, const float4 *__restrict__ node_hi,
                       const uint2 *__restrict__ node_lr, const uint32_t *__restrict__ height,
                       CollapseCost *__restrict__ table)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_internal || height[i] != pass)
    {
        return;
    }
    const uint2 lr = node_lr[i];
    const CollapseCost L = collapse_cost_of(lr.x, n_leaves, tri_cost, node_lo, node_hi, table);
    const CollapseCost R = collapse_cost_of(lr.y, n_leaves, tri_cost, node_lo, node_hi, table);
    CollapseCost out;
    collapse_cost_node(n_leaves + i, n_leaves, tri_cost, node_lo, node_hi, node_lr, L, R, out);
    table[i] = out;
}

// For every wide node of the current level: pick its (up to 8) children among
// the binary subtree below wide_root[w], count internal children and leaf
// triangles. table != nullptr: the children the cost table chose; nullptr: the
// greedy rule (open the child of largest area until 8 slots are used).
__global__ void __launch_bounds__(kBlock)
k_wide_select(uint32_t level_start, uint32_t level_end, uint32_t n_leaves,
              const uint32_t *__restrict__ wide_root, const float4 *__restrict__ node_lo,
              const float4 *__restrict__ node_hi, const uint2 *__restrict__ node_lr,
              const CollapseCost *__restrict__ table, uint32_t *__restrict__ sel_child,
              uint32_t *__restrict__ sel_n, unsigned long long *__restrict__ counts)
{
    const uint32_t w = level_start + blockIdx.x * kBlock + threadIdx.x;
    if (w >= level_end)
    {
        return;
    }
    uint32_t child[8];
    float area[8];
    uint32_t n;
    const uint32_t b = wide_root[w];
    if (b < n_leaves)
    {
        child[0] = b;
        area[0] = -1.0f;
        n = 1;
    }
    else if (table != nullptr)
    {
        // follow the recorded choices: a work list of (subtree, roots it may become)
        uint32_t todo_id[8], todo_budget[8];
        int top = 0;
        const uint2 lr = node_lr[b - n_leaves];
        const uint32_t k8 = (table[b - n_leaves].dec >> 21) & 7u;
        todo_id[top] = lr.y;
        todo_budget[top++] = 8u - k8;
        todo_id[top] = lr.x;
        todo_budget[top++] = k8;
        n = 0;
        while (top > 0)
        {
            const uint32_t m = todo_id[--top];
            uint32_t j = todo_budget[top];
            uint32_t k = 0u;
            if (m >= n_leaves)
            {
                const uint32_t dec = table[m - n_leaves].dec;
                while (j > 1u && (k = (dec >> (3u * (j - 1u))) & 7u) == 0u)
                {
                    --j;
                }
            }
            if (m < n_leaves || j <= 1u || n + (uint32_t)top >= 7u)
            {
                child[n] = m;
                area[n] = -1.0f;
                ++n; // (the budgets on the list sum to at most 8 - n, each is >= 1: the last test never fires)
            }
            else
            {
                const uint2 lr2 = node_lr[m - n_leaves];
                todo_id[top] = lr2.y;
                todo_budget[top++] = j - k;
                todo_id[top] = lr2.x;
                todo_budget[top++] = k;
            }
        }
    }
    else
    {
        const uint2 lr = node_lr[b - n_leaves];
        child[0] = lr.x;
        child[1] = lr.y;
        n = 2;
        for (uint32_t k = 0; k < 2; ++k)
        {
            area[k] = child[k] < n_leaves ? -1.0f : half_area(node_lo[child[k]], node_hi[child[k]]);
        }
        while (n < 8)
        {
            int pick = -1;
            float best = -1.0f;
            for (uint32_t k = 0; k < n; ++k)
            {
                if (area[k] > best)
                {
                    best = area[k];
                    pick = (int)k;
                }
            }
            if (pick < 0)
            {
                break;
            }
            const uint2 lr2 = node_lr[child[pick] - n_leaves];
            child[pick] = lr2.x;
            child[n] = lr2.y;
            area[pick] = lr2.x < n_leaves ? -1.0f : half_area(node_lo[lr2.x], node_hi[lr2.x]);
            area[n] = lr2.y < n_leaves ? -1.0f : half_area(node_lo[lr2.y], node_hi[lr2.y]);
            ++n;
        }
    }
    uint32_t ni = 0, nt = 0;
    for (uint32_t k = 0; k < 8; ++k)
    {
        uint32_t c = 0xffffffffu;
        if (k < n)
        {
            c = child[k];
            const uint32_t cnt = __float_as_uint(node_lo[c].w);
            if (cnt > kLeafMax)
            {
                ++ni;
            }
            else
            {
                nt += cnt;
            }
        }
        sel_child[8u * (size_t)w + k] = c;
    }
    sel_n[w] = n;
    counts[w - level_start] = ((unsigned long long)ni << 32) | nt;
}

__device__ __forceinline__ uint32_t ceil_pow2_exp(float ext)
{
    // exponent byte e with 2^(e-127) * 255 >= ext (checked again by the caller)
    const float s = ext / 255.0f;
    const uint32_t bits = __float_as_uint(s);
    uint32_t e = (bits >> 23) & 0xffu;
    if (bits & 0x7fffffu)
    {
        e += 1u;
    }
    return e < 1u ? 1u : (e > 254u ? 254u : e);
}

__device__ __forceinline__ int quant_lo(float v, float org, float scale)
{
    int q = (int)__builtin_floorf((v - org) / scale);
    q = q < 0 ? 0 : (q > 255 ? 255 : q);
    while (q > 0 && __builtin_fmaf((float)q, scale, org) > v)
    {
        --q;
    }
    return q;
}
__device__ __forceinline__ int quant_hi(float v, float org, float scale)
{
    int q = (int)__builtin_ceilf((v - org) / scale);
    q = q < 0 ? 0 : (q > 256 ? 256 : q);
    while (q <= 255 && __builtin_fmaf((float)q, scale, org) < v)
    {
        ++q;
    }
    return q; // 256 = does not fit
}

__global__ void __launch_bounds__(kBlock)
k_wide_emit(uint32_t level_start, uint32_t level_end, uint32_t n_leaves, uint32_t child_alloc,
            uint32_t tri_alloc, const unsigned long long *__restrict__ scan,
            const uint32_t *__restrict__ sel_child, const uint32_t *__restrict__ sel_n,
            const float4 *__restrict__ node_lo, const float4 *__restrict__ node_hi,
            const uint2 *__restrict__ node_lr, const float4 *__restrict__ tri_tmp,
            uint32_t *__restrict__ wide_root, uint4 *__restrict__ nodes_out,
            float4 *__restrict__ tris_out, double *__restrict__ cost_sum)
{
    const uint32_t w = level_start + blockIdx.x * kBlock + threadIdx.x;
    if (w >= level_end)
    {
        return;
    }
    const unsigned long long sc = scan[w - level_start];
    const uint32_t child_base = child_alloc + (uint32_t)(sc >> 32);
    const uint32_t tri_base = tri_alloc + (uint32_t)(sc & 0xffffffffu);
    const uint32_t n = sel_n[w];

    uint32_t child[8];
    float4 clo[8], chi[8];
    float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
    float4 hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    for (uint32_t k = 0; k < 8; ++k)
    {
        child[k] = sel_child[8u * (size_t)w + k];
        if (k < n)
        {
            clo[k] = node_lo[child[k]];
            chi[k] = node_hi[child[k]];
            lo.x = fmin_(lo.x, clo[k].x);
            lo.y = fmin_(lo.y, clo[k].y);
            lo.z = fmin_(lo.z, clo[k].z);
            hi.x = fmax_(hi.x, chi[k].x);
            hi.y = fmax_(hi.y, chi[k].y);
            hi.z = fmax_(hi.z, chi[k].z);
        }
    }
    if (n == 0)
    {
        lo = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        hi = lo;
    }

    // slot assignment: slot s should hold the child lying furthest towards
    // +axis where bit(axis) of s is set and towards -axis where it is clear,
    // so that rays of octant o meet slots in decreasing (s ^ (7 - o)) order
    const float ncx = 0.5f * (lo.x + hi.x), ncy = 0.5f * (lo.y + hi.y), ncz = 0.5f * (lo.z + hi.z);
    int slot_of[8];
    int child_in_slot[8];
    for (int k = 0; k < 8; ++k)
    {
        slot_of[k] = -1;
        child_in_slot[k] = -1;
    }
    for (uint32_t round = 0; round < n; ++round)
    {
        float best = -INFINITY;
        int bc = -1, bs = -1;
        for (uint32_t k = 0; k < n; ++k)
        {
            if (slot_of[k] >= 0)
            {
                continue;
            }
            const float dx = 0.5f * (clo[k].x + chi[k].x) - ncx;
            const float dy = 0.5f * (clo[k].y + chi[k].y) - ncy;
            const float dz = 0.5f * (clo[k].z + chi[k].z) - ncz;
            for (int s = 0; s < 8; ++s)
            {
                if (child_in_slot[s] >= 0)
                {
                    continue;
                }
                const float score = ((s & 1) ? dx : -dx) + ((s & 2) ? dy : -dy) + ((s & 4) ? dz : -dz);
                if (score > best)
                {
                    best = score;
                    bc = (int)k;
                    bs = s;
                }
            }
        }
        slot_of[bc] = bs;
        child_in_slot[bs] = bc;
    }

    // per-axis power-of-two grid
    uint32_t ex = ceil_pow2_exp(hi.x - lo.x);
    uint32_t ey = ceil_pow2_exp(hi.y - lo.y);
    uint32_t ez = ceil_pow2_exp(hi.z - lo.z);
    uint32_t qlo[3][8], qhi[3][8];
    for (int attempt = 0; attempt < 4; ++attempt)
    {
        const float sx = __uint_as_float(ex << 23), sy = __uint_as_float(ey << 23),
                    sz = __uint_as_float(ez << 23);
        bool okx = true, oky = true, okz = true;
        for (int s = 0; s < 8; ++s)
        {
            const int k = child_in_slot[s];
            if (k < 0)
            {
                qlo[0][s] = qlo[1][s] = qlo[2][s] = 0u;
                qhi[0][s] = qhi[1][s] = qhi[2][s] = 0u;
                continue;
            }
            const int hx = quant_hi(chi[k].x, lo.x, sx), hy = quant_hi(chi[k].y, lo.y, sy),
                      hz = quant_hi(chi[k].z, lo.z, sz);
            okx = okx && hx <= 255;
            oky = oky && hy <= 255;
            okz = okz && hz <= 255;
            qlo[0][s] = (uint32_t)quant_lo(clo[k].x, lo.x, sx);
            qlo[1][s] = (uint32_t)quant_lo(clo[k].y, lo.y, sy);
            qlo[2][s] = (uint32_t)quant_lo(clo[k].z, lo.z, sz);
            qhi[0][s] = (uint32_t)(hx > 255 ? 255 : hx);
            qhi[1][s] = (uint32_t)(hy > 255 ? 255 : hy);
            qhi[2][s] = (uint32_t)(hz > 255 ? 255 : hz);
        }
        if (okx && oky && okz)
        {
            break;
        }
        ex += okx ? 0u : 1u;
        ey += oky ? 0u : 1u;
        ez += okz ? 0u : 1u;
    }

    // slots in order: meta, internal children ranks, leaf triangles
    uint32_t imask = 0u, meta[8];
    uint32_t rank = 0u, toff = 0u;
    float cost = 0.0f;
    const float sx = __uint_as_float(ex << 23), sy = __uint_as_float(ey << 23),
                sz = __uint_as_float(ez << 23);
    for (int s = 0; s < 8; ++s)
    {
        meta[s] = 0u;
        const int k = child_in_slot[s];
        if (k < 0)
        {
            continue;
        }
        const float bx = (float)(qhi[0][s] - qlo[0][s]) * sx, by = (float)(qhi[1][s] - qlo[1][s]) * sy,
                    bz = (float)(qhi[2][s] - qlo[2][s]) * sz;
        const float a = bx * by + by * bz + bz * bx;
        const uint32_t c = child[k];
        const uint32_t cnt = __float_as_uint(clo[k].w);
        if (cnt > kLeafMax)
        {
            imask |= 1u << s;
            meta[s] = 0x20u | (24u + (uint32_t)s);
            wide_root[child_base + rank] = c;
            ++rank;
            cost += a;
        }
        else
        {
            meta[s] = (((1u << cnt) - 1u) << 5) | toff;
            // gather the <= 3 triangles below c, left to right
            uint32_t stack[4];
            int sp = 0;
            stack[sp++] = c;
            while (sp)
            {
                const uint32_t id = stack[--sp];
                if (id < n_leaves)
                {
                    const uint32_t p = __float_as_uint(node_hi[id].w);
                    float4 *__restrict__ dst = tris_out + 3u * (size_t)(tri_base + toff);
                    dst[0] = tri_tmp[3u * (size_t)p + 0];
                    dst[1] = tri_tmp[3u * (size_t)p + 1];
                    dst[2] = tri_tmp[3u * (size_t)p + 2];
                    ++toff;
                }
                else
                {
                    const uint2 lr = node_lr[id - n_leaves];
                    stack[sp++] = lr.y;
                    stack[sp++] = lr.x;
                }
            }
            cost += a * (float)cnt;
        }
    }

    uint4 w0, w1, w2, w3, w4;
    w0.x = __float_as_uint(lo.x);
    w0.y = __float_as_uint(lo.y);
    w0.z = __float_as_uint(lo.z);
    w0.w = ex | (ey << 8) | (ez << 16) | (imask << 24);
    w1.x = child_base;
    w1.y = tri_base;
    w1.z = meta[0] | (meta[1] << 8) | (meta[2] << 16) | (meta[3] << 24);
    w1.w = meta[4] | (meta[5] << 8) | (meta[6] << 16) | (meta[7] << 24);
#define PTC_PACK4(a, o) ((a)[o] | ((a)[(o) + 1] << 8) | ((a)[(o) + 2] << 16) | ((a)[(o) + 3] << 24))
    w2.x = PTC_PACK4(qlo[0], 0);
    w2.y = PTC_PACK4(qlo[0], 4);
    w2.z = PTC_PACK4(qlo[1], 0);
    w2.w = PTC_PACK4(qlo[1], 4);
    w3.x = PTC_PACK4(qlo[2], 0);
    w3.y = PTC_PACK4(qlo[2], 4);
    w3.z = PTC_PACK4(qhi[0], 0);
    w3.w = PTC_PACK4(qhi[0], 4);
    w4.x = PTC_PACK4(qhi[1], 0);
    w4.y = PTC_PACK4(qhi[1], 4);
    w4.z = PTC_PACK4(qhi[2], 0);
    w4.w = PTC_PACK4(qhi[2], 4);
#undef PTC_PACK4
    uint4 *__restrict__ out = nodes_out + 5u * (size_t)w;
    out[0] = w0;
    out[1] = w1;
    out[2] = w2;
    out[3] = w3;
    out[4] = w4;
    // one atomic per wave: adds of every thread to the one address would
    // serialise in L2 (about 10 ns each) and dominate large builds
    unsigned long long active = __ballot(1);
    const int first = __builtin_ctzll(active);
    double wave_cost = 0.0;
    while (active)
    {
        wave_cost += __shfl((double)cost, __builtin_ctzll(active), 64);
        active &= active - 1ull;
    }
    if ((int)(threadIdx.x & 63u) == first)
    {
        atomicAdd(cost_sum, wave_cost);
    }
}

// Build scratch: one device allocation carved up in order (dozens of separate
// hipMalloc / hipFree calls cost more than the kernels of a small build);
// requests beyond the estimate fall back to allocations of their own.
struct TempPool
{
    char *arena = nullptr;
    size_t capacity = 0, used = 0;
    std::vector<void *> extra;
    ~TempPool()
    {
        (void)hipFree(arena);
        for (void *p : extra)
        {
            (void)hipFree(p);
        }
    }
    hipError_t reserve(size_t bytes)
    {
        const hipError_t e = hipMalloc(reinterpret_cast<void **>(&arena), bytes);
        capacity = e == hipSuccess ? bytes : 0;
        return e;
    }
    template <typename T> hipError_t alloc(T **out, size_t count)
    {
        const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes <= capacity)
        {
            *out = reinterpret_cast<T *>(arena + used);
            used += bytes;
            return hipSuccess;
        }
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess)
        {
            extra.push_back(p);
            *out = static_cast<T *>(p);
        }
        return e;
    }
};

inline uint32_t blocks_for(size_t n)
{
    return (uint32_t)((n + kBlock - 1) / kBlock);
}

} // namespace

namespace
{
int build_bvh_once(ptc_context *ctx, ptc_scene *scene, const std::vector<DInstance> &instances, uint32_t n,
                   const BuildParams &params);
}

int build_bvh(ptc_context *ctx, ptc_scene *scene, const std::vector<DInstance> &instances,
              uint32_t n, const BuildParams &params)
{
    int rc = build_bvh_once(ctx, scene, instances, n, params);
    if (rc == PTC_E_LIMIT && params.optimize_rounds != kNoOptimize)
    {
        // The wide tree came out deeper than the traversal stack holds. The
        // reinsertion rounds lower the tree's cost, not its height: a scene that
        // builds without them must not start failing because of them.
        (void)hipFree(scene->d_tris);
        (void)hipFree(scene->d_nodes);
        scene->d_tris = nullptr;
        scene->d_nodes = nullptr;
        BuildParams plain = params;
        plain.optimize_rounds = kNoOptimize;
        rc = build_bvh_once(ctx, scene, instances, n, plain);
    }
    return rc;
}

namespace
{
int build_bvh_once(ptc_context *ctx, ptc_scene *scene, const std::vector<DInstance> &instances, uint32_t n,
                   const BuildParams &params)
{
    hipStream_t st = ctx->stream;
    // default 8: over the 871k-triangle and 10 M-triangle meshes, the four-instance table and the mesh inside the
    // Cornell box, 8 gives trees as fast as or faster than 12, 16, 24 and 32 to traverse (differences of 1 to 4 %
    // either way, not monotonic in the radius) and the shortest build
    const int radius = (int)std::min<uint32_t>(params.search_radius ? params.search_radius : 8u, kMaxRadius);
    // cost of one triangle test in units of one wide-node visit (a visit is five
    // 16-byte fetches and eight box tests, a triangle test three fetches)
    const float tri_cost = params.tri_cost > 0.0f ? params.tri_cost : kDefaultTriCost;
    const bool greedy = params.collapse == 1u;
    TempPool pool;
    // about 450 bytes of scratch per triangle, plus the sort's and the scan's own
    PTC_HIP_CHECK(pool.reserve((size_t)n * 576 + (1u << 20)));

    // outputs
    const size_t node_cap = std::max<size_t>(n, 1);
    uint4 *nodes_tmp = nullptr;
    PTC_HIP_CHECK(pool.alloc(&nodes_tmp, node_cap * 5));
    PTC_HIP_CHECK(hipMalloc(&scene->d_tris, std::max<size_t>(n, 1) * 48));
    double *d_cost = nullptr;
    PTC_HIP_CHECK(pool.alloc(&d_cost, 1));
    PTC_HIP_CHECK(hipMemsetAsync(d_cost, 0, sizeof(double), st));

    // destroyed on every return path
    struct EventPair
    {
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair()
        {
            if (a)
            {
                (void)hipEventDestroy(a);
            }
            if (b)
            {
                (void)hipEventDestroy(b);
            }
        }
    } ev;
    PTC_HIP_CHECK(hipEventCreate(&ev.a));
    PTC_HIP_CHECK(hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    PTC_HIP_CHECK(hipEventRecord(e0, st));

    uint32_t n_nodes = 1, depth = 1;
    double root_half_area = 0.0;
    scene->d.box_lo = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    scene->d.box_hi = make_float4(-1.0f, -1.0f, -1.0f, 0.0f);
    if (n == 0)
    {
        PTC_HIP_CHECK(hipMemsetAsync(nodes_tmp, 0, 80, st));
    }
    else
    {
        float4 *tri_tmp, *box_lo, *box_hi, *node_lo, *node_hi, *cl_lo[2], *cl_hi[2];
        uint32_t *bounds, *vals_in, *vals_out, *cl_id[2], *wide_root, *sel_child, *sel_n, *totals;
        unsigned long long *keys_in, *keys_out, *flags, *scan;
        uint2 *node_lr;
        int *nn;
        PTC_HIP_CHECK(pool.alloc(&tri_tmp, (size_t)n * 3));
        PTC_HIP_CHECK(pool.alloc(&box_lo, n));
        PTC_HIP_CHECK(pool.alloc(&box_hi, n));
        PTC_HIP_CHECK(pool.alloc(&node_lo, 2 * (size_t)n));
        PTC_HIP_CHECK(pool.alloc(&node_hi, 2 * (size_t)n));
        PTC_HIP_CHECK(pool.alloc(&node_lr, n));
        for (int k = 0; k < 2; ++k)
        {
            PTC_HIP_CHECK(pool.alloc(&cl_lo[k], n));
            PTC_HIP_CHECK(pool.alloc(&cl_hi[k], n));
            PTC_HIP_CHECK(pool.alloc(&cl_id[k], n));
        }
        PTC_HIP_CHECK(pool.alloc(&bounds, 8));
        PTC_HIP_CHECK(pool.alloc(&totals, 2));
        PTC_HIP_CHECK(pool.alloc(&vals_in, n));
        PTC_HIP_CHECK(pool.alloc(&vals_out, n));
        PTC_HIP_CHECK(pool.alloc(&keys_in, n));
        PTC_HIP_CHECK(pool.alloc(&keys_out, n));
        PTC_HIP_CHECK(pool.alloc(&flags, n));
        PTC_HIP_CHECK(pool.alloc(&scan, n));
        PTC_HIP_CHECK(pool.alloc(&nn, n));
        PTC_HIP_CHECK(pool.alloc(&wide_root, node_cap));
        PTC_HIP_CHECK(pool.alloc(&sel_child, node_cap * 8));
        PTC_HIP_CHECK(pool.alloc(&sel_n, node_cap));
        CollapseCost *cost_table = nullptr;
        if (!greedy)
        {
            PTC_HIP_CHECK(pool.alloc(&cost_table, n));
        }
        std::vector<std::pair<uint32_t, uint32_t>> round_nodes; // [first, last) node ids made by each clustering round

        const uint32_t init_bounds[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
        PTC_HIP_CHECK(hipMemcpyAsync(bounds, init_bounds, sizeof init_bounds, hipMemcpyHostToDevice, st));

        k_flatten<<<std::min<uint32_t>(blocks_for(n), 2048u), kBlock, 0, st>>>(scene->d_instances, (uint32_t)instances.size(),
                                                   scene->d_positions, scene->d_indices, n, tri_tmp,
                                                   box_lo, box_hi, bounds);
        k_morton<<<blocks_for(n), kBlock, 0, st>>>(n, box_lo, box_hi, bounds, keys_in, vals_in);

        size_t sort_bytes = 0, scan_bytes = 0;
        PTC_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys_in, keys_out, vals_in,
                                                vals_out, n, 0, 63, st));
        PTC_HIP_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, flags, scan, 0ULL, n,
                                              rocprim::plus<unsigned long long>(), st));
        char *prim_tmp = nullptr;
        const size_t prim_bytes = std::max(sort_bytes, scan_bytes);
        PTC_HIP_CHECK(pool.alloc(&prim_tmp, prim_bytes));
        size_t tmp_bytes = prim_bytes;
        PTC_HIP_CHECK(rocprim::radix_sort_pairs(prim_tmp, tmp_bytes, keys_in, keys_out, vals_in,
                                                vals_out, n, 0, 63, st));

        k_init_clusters<<<blocks_for(n), kBlock, 0, st>>>(n, vals_out, box_lo, box_hi, node_lo, node_hi,
                                                         cl_id[0], cl_lo[0], cl_hi[0]);

        // PLOC rounds
        uint32_t n_cl = n, next_node = n;
        int cur = 0;
        int rounds = 0;
        while (n_cl > (uint32_t)kTailMax)
        {
            k_nearest<<<blocks_for(n_cl), kBlock, 0, st>>>(n_cl, radius, cl_lo[cur], cl_hi[cur], nn);
            k_merge_flags<<<blocks_for(n_cl), kBlock, 0, st>>>(n_cl, nn, flags);
            tmp_bytes = prim_bytes;
            PTC_HIP_CHECK(rocprim::exclusive_scan(prim_tmp, tmp_bytes, flags, scan, 0ULL, n_cl,
                                                  rocprim::plus<unsigned long long>(), st));
            k_merge_emit<<<blocks_for(n_cl), kBlock, 0, st>>>(
                n_cl, n, next_node, nn, flags, scan, cl_id[cur], cl_lo[cur], cl_hi[cur], cl_id[cur ^ 1],
                cl_lo[cur ^ 1], cl_hi[cur ^ 1], node_lo, node_hi, node_lr, totals);
            uint32_t h_tot[2];
            PTC_HIP_CHECK(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, st));
            PTC_HIP_CHECK(hipStreamSynchronize(st));
            if (h_tot[1] == 0 || h_tot[0] >= n_cl)
            {
                set_error("BVH build: clustering round made no progress");
                return PTC_E_DEVICE;
            }
            n_cl = h_tot[0];
            round_nodes.emplace_back(next_node, next_node + h_tot[1]);
            next_node += h_tot[1];
            cur ^= 1;
            if (++rounds > 4096)
            {
                set_error("BVH build: too many clustering rounds");
                return PTC_E_DEVICE;
            }
        }
        if (n_cl > 1)
        {
            k_ploc_tail<<<1, kTailMax, 0, st>>>(n_cl, n, next_node, radius, cl_id[cur], cl_lo[cur], cl_hi[cur],
                                                node_lo, node_hi, node_lr, totals);
            uint32_t h_tot[2];
            PTC_HIP_CHECK(hipMemcpyAsync(h_tot, totals, sizeof h_tot, hipMemcpyDeviceToHost, st));
            PTC_HIP_CHECK(hipStreamSynchronize(st));
            if (h_tot[0] != 1u)
            {
                set_error("BVH build: clustering tail made no progress");
                return PTC_E_DEVICE;
            }
            next_node = h_tot[1];
        }
        const uint32_t binary_root = n == 1 ? 0u : next_node - 1u;
        // reinsertion rounds on the binary tree (see 3b)
        uint32_t opt_rounds = params.optimize_rounds == kNoOptimize
                                  ? 0u
                                  : (params.optimize_rounds ? params.optimize_rounds : kDefaultOptimizeRounds);
        opt_rounds = std::min(opt_rounds, kMaxOptimizeRounds);
        uint2 *up = nullptr;
        uint32_t *height = nullptr;
        uint32_t tree_height = 0u;
        if (n < 4u || next_node != 2u * n - 1u)
        {
            opt_rounds = 0u;
        }
        if (opt_rounds > 0u)
        {
            const uint32_t n_bin = 2u * n - 1u, n_int = n - 1u;
            uint32_t *cand_x, *cand_pivot, *counts, *marked;
            float *cand_gain;
            unsigned long long *lock;
            PTC_HIP_CHECK(pool.alloc(&up, n_bin));
            PTC_HIP_CHECK(pool.alloc(&height, n));
            PTC_HIP_CHECK(pool.alloc(&cand_x, n_bin));
            PTC_HIP_CHECK(pool.alloc(&cand_pivot, n_bin));
            PTC_HIP_CHECK(pool.alloc(&cand_gain, n_bin));
            PTC_HIP_CHECK(pool.alloc(&lock, n_bin));
            PTC_HIP_CHECK(pool.alloc(&counts, 2)); // moves, candidates
            PTC_HIP_CHECK(pool.alloc(&marked, n_bin));
            const bool trace = getenv("PTC_OPT_TRACE") != nullptr;
            uint32_t passes = 24u; // launches of the refit before the host first looks
            for (uint32_t round = 0; round < opt_rounds; ++round)
            {
                PTC_HIP_CHECK(hipMemsetAsync(lock, 0, (size_t)n_bin * 8, st));
                PTC_HIP_CHECK(hipMemsetAsync(counts, 0, 8, st));
                PTC_HIP_CHECK(hipMemsetAsync(marked, 0, (size_t)n_bin * 4, st));
                PTC_HIP_CHECK(hipMemsetAsync(height, 0, (size_t)n * 4, st));
                k_opt_links<<<blocks_for(n_int), kBlock, 0, st>>>(n, n_int, node_lr, up);
                k_opt_find<<<blocks_for(n_bin), kBlock, 0, st>>>(n, n_bin, node_lo, node_hi, node_lr, up, cand_x,
                                                                 cand_pivot, cand_gain, lock, trace ? counts + 1 : nullptr);
                k_opt_mark<<<blocks_for(n_bin), kBlock, 0, st>>>(n_bin, up, cand_x, cand_pivot, cand_gain, lock, marked);
                k_opt_apply<<<blocks_for(n_bin), kBlock, 0, st>>>(n, n_bin, node_lr, up, cand_x, cand_gain, lock,
                                                                  marked, counts, trace);
                // counts: moves, candidates, the root's height (0 until the refit has reached it)
                uint32_t h_counts[3] = {0u, 0u, 0u};
                for (uint32_t pass = 0u; h_counts[2] == 0u;)
                {
                    for (uint32_t k = 0; k < passes; ++k)
                    {
                        k_opt_refit_pass<<<blocks_for(n_int), kBlock, 0, st>>>(++pass, n, n_int, node_lr, node_lo, node_hi,
                                                                               height);
                    }
                    PTC_HIP_CHECK(hipMemcpyAsync(h_counts, counts, 8, hipMemcpyDeviceToHost, st));
                    PTC_HIP_CHECK(hipMemcpyAsync(h_counts + 2, height + (n_int - 1u), 4, hipMemcpyDeviceToHost, st));
                    PTC_HIP_CHECK(hipStreamSynchronize(st));
                    if (h_counts[2] == 0u && pass > 2u * n)
                    {
                        set_error("BVH build: refit after reinsertion does not finish (links of the tree are broken)");
                        return PTC_E_DEVICE;
                    }
                    passes = 8u;
                }
                tree_height = h_counts[2];
                passes = tree_height + 2u; // the next round's tree is about as high
                if (trace)
                {
                    fprintf(stderr, "ptc: reinsertion round %u moved %u of %u candidates (%u nodes, height %u)\n", round,
                            h_counts[0], h_counts[1], n_bin, tree_height);
                }
                if (h_counts[0] == 0u)
                {
                    break;
                }
            }
        }
        if (!greedy && opt_rounds > 0u)
        {
            for (uint32_t pass = 1u; pass <= tree_height; ++pass)
            {
                k_collapse_cost_height<<<blocks_for(n - 1u), kBlock, 0, st>>>(pass, n, n - 1u, tri_cost, node_lo, node_hi,
                                                                            node_lr, height, cost_table);
            }
        }
        else if (!greedy)
        {
            for (const auto &r : round_nodes)
            {
                k_collapse_cost<<<blocks_for(r.second - r.first), kBlock, 0, st>>>(r.first, r.second, n, tri_cost, node_lo,
                                                                               node_hi, node_lr, cost_table);
            }
            const uint32_t tail_first = round_nodes.empty() ? n : round_nodes.back().second;
            if (next_node > tail_first)
            {
                k_collapse_cost_tail<<<1, kTailMax, 0, st>>>(tail_first, next_node, n, tri_cost, node_lo, node_hi, node_lr,
                                                             cost_table);
            }
        }

        // collapse, one wide level per pass
        PTC_HIP_CHECK(hipMemcpyAsync(wide_root, &binary_root, 4, hipMemcpyHostToDevice, st));
        {
            float4 rl, rh;
            PTC_HIP_CHECK(hipMemcpyAsync(&rl, node_lo + binary_root, 16, hipMemcpyDeviceToHost, st));
            PTC_HIP_CHECK(hipMemcpyAsync(&rh, node_hi + binary_root, 16, hipMemcpyDeviceToHost, st));
            PTC_HIP_CHECK(hipStreamSynchronize(st));
            const double dx = (double)rh.x - rl.x, dy = (double)rh.y - rl.y, dz = (double)rh.z - rl.z;
            root_half_area = dx * dy + dy * dz + dz * dx;
            // union of the padded triangle boxes
            scene->d.box_lo = make_float4(rl.x, rl.y, rl.z, 0.0f);
            scene->d.box_hi = make_float4(rh.x, rh.y, rh.z, 0.0f);
        }
        uint32_t level_start = 0, level_end = 1, tri_alloc = 0;
        depth = 0;
        while (level_start < level_end)
        {
            const uint32_t cnt = level_end - level_start;
            k_wide_select<<<blocks_for(cnt), kBlock, 0, st>>>(level_start, level_end, n, wide_root, node_lo,
                                                             node_hi, node_lr, cost_table, sel_child, sel_n, flags);
            tmp_bytes = prim_bytes;
            PTC_HIP_CHECK(rocprim::exclusive_scan(prim_tmp, tmp_bytes, flags, scan, 0ULL, cnt,
                                                  rocprim::plus<unsigned long long>(), st));
            unsigned long long last_f = 0, last_s = 0;
            PTC_HIP_CHECK(hipMemcpyAsync(&last_f, flags + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
            PTC_HIP_CHECK(hipMemcpyAsync(&last_s, scan + (cnt - 1), 8, hipMemcpyDeviceToHost, st));
            PTC_HIP_CHECK(hipStreamSynchronize(st));
            const uint32_t new_nodes = (uint32_t)((last_f + last_s) >> 32);
            const uint32_t new_tris = (uint32_t)((last_f + last_s) & 0xffffffffu);
            if ((size_t)level_end + new_nodes > node_cap || (size_t)tri_alloc + new_tris > n)
            {
                set_error("BVH build: node or triangle allocation overflow");
                return PTC_E_DEVICE;
            }
            k_wide_emit<<<blocks_for(cnt), kBlock, 0, st>>>(
                level_start, level_end, n, level_end, tri_alloc, scan, sel_child, sel_n, node_lo, node_hi,
                node_lr, tri_tmp, wide_root, nodes_tmp, static_cast<float4 *>(scene->d_tris), d_cost);
            tri_alloc += new_tris;
            level_start = level_end;
            level_end += new_nodes;
            ++depth;
        }
        n_nodes = level_end;
        if (tri_alloc != n)
        {
            set_error("BVH build: triangle count mismatch after collapse");
            return PTC_E_DEVICE;
        }
    }

    PTC_HIP_CHECK(hipMalloc(&scene->d_nodes, (size_t)n_nodes * 80));
    PTC_HIP_CHECK(hipMemcpyAsync(scene->d_nodes, nodes_tmp, (size_t)n_nodes * 80, hipMemcpyDeviceToDevice, st));
    double h_cost = 0.0;
    PTC_HIP_CHECK(hipMemcpyAsync(&h_cost, d_cost, sizeof(double), hipMemcpyDeviceToHost, st));
    PTC_HIP_CHECK(hipEventRecord(e1, st));
    PTC_HIP_CHECK(hipStreamSynchronize(st));
    float ms = 0.0f;
    PTC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));

    // a traversal may hold two stack entries per level (siblings and postponed triangles)
    if (2u * depth + 2u > (uint32_t)kStackTotal)
    {
        set_error("BVH build: wide tree deeper than the traversal stack");
        return PTC_E_LIMIT;
    }
    scene->d.nodes = static_cast<const uint4 *>(scene->d_nodes);
    scene->d.tris = static_cast<const float4 *>(scene->d_tris);
    scene->info.n_nodes = n_nodes;
    scene->d.n_nodes = n_nodes;
    scene->info.bvh_depth = depth;
    scene->info.build_ms = ms;
    scene->info.sah_cost = root_half_area > 0.0 ? (float)(1.0 + h_cost / root_half_area) : 1.0f;
    return PTC_OK;
}
} // namespace

namespace
{
inline uint32_t float_bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float bits_float(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// child and triangle indices of a mesh's tree, moved to where the tree sits in
// the concatenated arrays
__global__ void __launch_bounds__(kBlock)
k_rebase_nodes(uint4 *__restrict__ nodes, uint32_t n_nodes, uint32_t node_off, uint32_t tri_off)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_nodes)
    {
        uint4 w1 = nodes[5u * (size_t)i + 1u];
        w1.x += node_off;
        w1.y += tri_off;
        nodes[5u * (size_t)i + 1u] = w1;
    }
}

// One mesh: the positions and normals of each triangle's corners, as the 80-byte
// record the hit shading reads (layout at DScene::shade).
__global__ void __launch_bounds__(kBlock)
k_shade_records(const float *__restrict__ positions, const float *__restrict__ normals,
                const uint32_t *__restrict__ indices, uint32_t n_tris, float4 *__restrict__ out)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_tris)
    {
        return;
    }
    float w[20];
#pragma unroll
    for (int c = 0; c < 3; ++c)
    {
        const size_t v = 3u * (size_t)indices[3u * (size_t)t + c];
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            w[3 * c + k] = positions[v + k];
            w[9 + 3 * c + k] = normals[v + k];
        }
    }
    w[18] = w[19] = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        out[5u * (size_t)t + q] = make_float4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    }
}
} // namespace

int build_shade_records(ptc_context *ctx, ptc_scene *scene, const std::vector<uint32_t> &vtx_off,
                        const std::vector<uint32_t> &tri_off)
{
    const uint32_t n_meshes = (uint32_t)vtx_off.size() - 1u;
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        const uint32_t n = tri_off[m + 1] - tri_off[m];
        if (n == 0u)
        {
            continue;
        }
        k_shade_records<<<blocks_for(n), kBlock, 0, ctx->stream>>>(
            scene->d_positions + 3u * (size_t)vtx_off[m], scene->d_normals + 3u * (size_t)vtx_off[m],
            scene->d_indices + 3u * (size_t)tri_off[m], n, scene->d_shade + 5u * (size_t)tri_off[m]);
    }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess)
    {
        set_error(std::string("shade records: ") + hipGetErrorString(err));
        return PTC_E_DEVICE;
    }
    return PTC_OK;
}

int build_two_level(ptc_context *ctx, ptc_scene *scene, std::vector<DInstance> &instances,
                    const std::vector<uint32_t> &vtx_off, const std::vector<uint32_t> &tri_off,
                    const std::vector<float> &mesh_box, const std::vector<uint32_t> &mesh_of,
                    const BuildParams &params)
{
    hipStream_t st = ctx->stream;
    const uint32_t n_meshes = (uint32_t)vtx_off.size() - 1u;
    // one identity pseudo-instance per mesh drives the ordinary builder
    std::vector<DInstance> pseudo(std::max<uint32_t>(n_meshes, 1u));
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        DInstance &P = pseudo[m];
        std::memset(&P, 0, sizeof P);
        P.o2w[0] = P.o2w[5] = P.o2w[10] = 1.0f;
        P.w2o[0] = P.w2o[5] = P.w2o[10] = 1.0f;
        P.vtx_offset = vtx_off[m];
        P.tri_offset = tri_off[m];
        P.n_tris = tri_off[m + 1] - tri_off[m];
    }
    DInstance *d_pseudo = nullptr;
    PTC_HIP_CHECK(hipMalloc(&d_pseudo, pseudo.size() * sizeof(DInstance)));
    struct Built
    {
        void *nodes = nullptr, *tris = nullptr;
        uint32_t n_nodes = 0, n_tris = 0, depth = 0;
    };
    std::vector<Built> built(n_meshes);
    auto cleanup = [&]() {
        (void)hipFree(d_pseudo);
        for (Built &b : built)
        {
            (void)hipFree(b.nodes);
            (void)hipFree(b.tris);
        }
    };
    hipError_t e = hipMemcpyAsync(d_pseudo, pseudo.data(), pseudo.size() * sizeof(DInstance), hipMemcpyHostToDevice, st);
    if (e != hipSuccess)
    {
        cleanup();
        set_error(std::string("two-level build: ") + hipGetErrorString(e));
        return PTC_E_DEVICE;
    }
    float build_ms = 0.0f;
    double sah_weighted = 0.0;
    uint32_t total_nodes = 0, total_tris = 0, depth = 0;
    for (uint32_t m = 0; m < n_meshes; ++m)
    {
        const uint32_t nt = pseudo[m].n_tris;
        if (nt == 0)
        {
            continue;
        }
        ptc_scene sub;
        sub.ctx = ctx;
        sub.d_positions = scene->d_positions;
        sub.d_indices = scene->d_indices;
        sub.d_instances = d_pseudo + m;
        const std::vector<DInstance> one(1, pseudo[m]);
        const int rc = build_bvh(ctx, &sub, one, nt, params);
        built[m].nodes = sub.d_nodes;
        built[m].tris = sub.d_tris;
        if (rc != PTC_OK)
        {
            cleanup();
            return rc;
        }
        built[m].n_nodes = sub.info.n_nodes;
        built[m].n_tris = nt;
        built[m].depth = sub.info.bvh_depth;
        build_ms += sub.info.build_ms;
        sah_weighted += (double)sub.info.sah_cost * nt;
        depth = std::max(depth, sub.info.bvh_depth);
        total_nodes += sub.info.n_nodes;
        total_tris += nt;
    }
    e = hipMalloc(&scene->d_nodes, std::max<size_t>(total_nodes, 1) * 80);
    if (e == hipSuccess)
    {
        e = hipMalloc(&scene->d_tris, std::max<size_t>(total_tris, 1) * 48);
    }
    std::vector<uint32_t> root_of(n_meshes, kNoBlas);
    uint32_t node_off = 0, tri_at = 0;
    for (uint32_t m = 0; m < n_meshes && e == hipSuccess; ++m)
    {
        const Built &b = built[m];
        if (b.n_tris == 0)
        {
            continue;
        }
        uint4 *dst = static_cast<uint4 *>(scene->d_nodes) + 5u * (size_t)node_off;
        e = hipMemcpyAsync(dst, b.nodes, (size_t)b.n_nodes * 80, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess)
        {
            e = hipMemcpyAsync(static_cast<char *>(scene->d_tris) + (size_t)tri_at * 48, b.tris, (size_t)b.n_tris * 48,
                               hipMemcpyDeviceToDevice, st);
        }
        if (e == hipSuccess)
        {
            k_rebase_nodes<<<blocks_for(b.n_nodes), kBlock, 0, st>>>(dst, b.n_nodes, node_off, tri_at);
            e = hipGetLastError();
        }
        root_of[m] = node_off;
        node_off += b.n_nodes;
        tri_at += b.n_tris;
    }
    // instance records: root of the mesh's tree; world box = the eight corners of
    // the mesh box through o2w, padded (the box only decides whether the
    // instance is entered, so generous padding costs time, never a hit)
    const size_t n_inst = instances.size();
    std::vector<float4> boxes(2 * n_inst);
    for (size_t i = 0; i < n_inst; ++i)
    {
        DInstance &I = instances[i];
        const uint32_t m = mesh_of[i];
        I.blas_root = root_of[m];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (I.blas_root != kNoBlas)
        {
            const float *mb = &mesh_box[6 * (size_t)m];
            float big = 0.0f;
            for (int c = 0; c < 8; ++c)
            {
                const float p[3] = {mb[(c & 1) ? 3 : 0], mb[(c & 2) ? 4 : 1], mb[(c & 4) ? 5 : 2]};
                for (int r = 0; r < 3; ++r)
                {
                    const float w = I.o2w[4 * r] * p[0] + I.o2w[4 * r + 1] * p[1] + I.o2w[4 * r + 2] * p[2] + I.o2w[4 * r + 3];
                    lo[r] = std::min(lo[r], w);
                    hi[r] = std::max(hi[r], w);
                    big = std::max(big, std::fabs(w));
                }
            }
            const float pad = big * (1.0f / 4096.0f) + 1e-30f;
            for (int r = 0; r < 3; ++r)
            {
                lo[r] -= pad;
                hi[r] += pad;
            }
        }
        boxes[2 * i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        boxes[2 * i + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    // The tree over the instances: those with triangles, sorted along a Morton
    // curve through their box centres (ties keep the index order), halved
    // recursively; nodes in depth-first order, each naming the node that follows
    // its subtree, so the walk needs no stack.
    std::vector<uint32_t> order;
    float glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < n_inst; ++i)
    {
        if (instances[i].blas_root != kNoBlas)
        {
            order.push_back((uint32_t)i);
            const float c[3] = {0.5f * (boxes[2 * i].x + boxes[2 * i + 1].x), 0.5f * (boxes[2 * i].y + boxes[2 * i + 1].y),
                                0.5f * (boxes[2 * i].z + boxes[2 * i + 1].z)};
            for (int r = 0; r < 3; ++r)
            {
                glo[r] = std::min(glo[r], c[r]);
                ghi[r] = std::max(ghi[r], c[r]);
            }
        }
    }
    auto spread = [](uint32_t v) { // 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000ffu;
        v = (v | (v << 8)) & 0x0300f00fu;
        v = (v | (v << 4)) & 0x030c30c3u;
        v = (v | (v << 2)) & 0x09249249u;
        return v;
    };
    std::vector<uint32_t> code(n_inst, 0u);
    for (uint32_t i : order)
    {
        uint32_t q[3];
        const float c[3] = {0.5f * (boxes[2 * i].x + boxes[2 * i + 1].x), 0.5f * (boxes[2 * i].y + boxes[2 * i + 1].y),
                            0.5f * (boxes[2 * i].z + boxes[2 * i + 1].z)};
        for (int r = 0; r < 3; ++r)
        {
            const float ext = ghi[r] - glo[r];
            const float f = ext > 0.0f ? (c[r] - glo[r]) / ext : 0.0f;
            q[r] = (uint32_t)std::min(1023.0f, std::max(0.0f, f * 1024.0f));
        }
        code[i] = spread(q[0]) | (spread(q[1]) << 1) | (spread(q[2]) << 2);
    }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });
    std::vector<float4> tree;
    tree.reserve(4 * order.size() + 2);
    struct Range
    {
        uint32_t first, last, node; // instances order[first..last), node already emitted when node != ~0u
    };
    // depth-first emission with an explicit stack: a node is emitted when its
    // range is first met and completed (box, follower) when the range is met again
    std::vector<Range> todo;
    if (!order.empty())
    {
        todo.push_back({0u, (uint32_t)order.size(), ~0u});
    }
    while (!todo.empty())
    {
        const Range r = todo.back();
        todo.pop_back();
        if (r.node != ~0u)
        {
            // both halves are in place behind the node: close it
            const uint32_t left = r.node + 1u, right = float_bits(tree[2 * (size_t)left].w);
            float4 lo = tree[2 * (size_t)left], hi = tree[2 * (size_t)left + 1];
            const float4 rl = tree[2 * (size_t)right], rh = tree[2 * (size_t)right + 1];
            lo.x = std::min(lo.x, rl.x);
            lo.y = std::min(lo.y, rl.y);
            lo.z = std::min(lo.z, rl.z);
            hi.x = std::max(hi.x, rh.x);
            hi.y = std::max(hi.y, rh.y);
            hi.z = std::max(hi.z, rh.z);
            lo.w = bits_float((uint32_t)(tree.size() / 2));
            hi.w = bits_float(kNoInstance);
            tree[2 * (size_t)r.node] = lo;
            tree[2 * (size_t)r.node + 1] = hi;
            continue;
        }
        const uint32_t node = (uint32_t)(tree.size() / 2);
        if (r.last - r.first == 1u)
        {
            const uint32_t i = order[r.first];
            float4 lo = boxes[2 * (size_t)i], hi = boxes[2 * (size_t)i + 1];
            lo.w = bits_float(node + 1u);
            hi.w = bits_float(i);
            tree.push_back(lo);
            tree.push_back(hi);
            continue;
        }
        tree.push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        tree.push_back(make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        const uint32_t mid = r.first + (r.last - r.first) / 2u;
        todo.push_back({r.first, r.last, node}); // closes the node after its halves
        todo.push_back({mid, r.last, ~0u});      // right half (emitted second)
        todo.push_back({r.first, mid, ~0u});     // left half (emitted first, right behind the node)
    }
    scene->d.n_inst_nodes = (uint32_t)(tree.size() / 2);
    if (tree.empty())
    {
        tree.resize(2, make_float4(0.0f, 0.0f, 0.0f, 0.0f)); // never read: n_inst_nodes is 0
    }
    // one allocation: the tree, then the instances' boxes by instance index
    const size_t tree_entries = tree.size();
    tree.insert(tree.end(), boxes.begin(), boxes.end());
    if (e == hipSuccess)
    {
        e = hipMalloc(&scene->d_inst_box, tree.size() * sizeof(float4));
    }
    if (e == hipSuccess)
    {
        e = hipMemcpyAsync(scene->d_inst_box, tree.data(), tree.size() * sizeof(float4), hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess && n_inst)
    {
        e = hipMemcpyAsync(scene->d_instances, instances.data(), n_inst * sizeof(DInstance), hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess)
    {
        e = hipStreamSynchronize(st);
    }
    cleanup();
    if (e != hipSuccess)
    {
        set_error(std::string("two-level build: ") + hipGetErrorString(e));
        return PTC_E_DEVICE;
    }
    scene->d.nodes = static_cast<const uint4 *>(scene->d_nodes);
    scene->d.tris = static_cast<const float4 *>(scene->d_tris);
    scene->d.inst_box = scene->d_inst_box;
    scene->d.inst_bounds = scene->d_inst_box + tree_entries;
    scene->info.n_nodes = total_nodes;
    scene->d.n_nodes = total_nodes;
    scene->info.bvh_depth = depth;
    scene->info.build_ms = build_ms;
    scene->info.sah_cost = total_tris ? (float)(sah_weighted / total_tris) : 1.0f;
    return PTC_OK;
}

} // namespace ptc
