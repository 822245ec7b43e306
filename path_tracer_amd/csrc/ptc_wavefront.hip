// Wavefront integrator: the path loop of src/shaders/shader.rgen:60-100 turned
// inside out. Instead of one lane walking one pixel's path to the end (lanes of
// a wave then idle while the longest path finishes), live paths sit in a
// compacted queue between bounces:
//
//   k_wf_primary  bounce 0, fused: camera ray -> traversal -> material / sky.
//                 Camera rays are coherent, so fusing costs nothing and saves
//                 writing the paths that end at the sky (3/4 of them in an
//                 open scene) to memory. Survivors are appended to the queue.
//   k_wf_drain    one persistent launch that finishes every queued path: each
//                 lane is a small state machine (idle / traversing / waiting to
//                 be shaded) that pulls the next path when its own ends.
//   k_wf_resolve  folds the per-path radiances of each pixel, in dispatch
//                 order, into the film exactly as shader.rgen:202-223 does
//
// and, for comparison (wavefront_bounces > 0), bounces as launch pairs before
// the persistent launch takes over:
//
//   k_wf_trace    traversal + Moller-Trumbore for queue slot i -> hit record
//   k_wf_shade    material / sky, emission, Russian roulette; survivors are
//                 appended to the next queue with one atomic per wave
//
// A path is (pixel, dispatch k); its id is k * n_pixels + pixel. Live-path state
// lives in SoA arrays indexed by queue slot, so queue reads and writes are
// coalesced; only the final radiance is scattered by path id. Each path carries
// its own RNG state, so results do not depend on queue order and equal the
// megakernel's bit for bit.
#include "ptc_internal.h"

#include <algorithm>
#include <cstdlib>
#include <string>

// Workspace of ptc_render_static(): callers that cannot allocate (code under
// HIP graph capture, serving loops, harnesses that forbid hipMalloc) render
// into this array instead of a hipMalloc'ed workspace. 64 MiB holds one
// dispatch of up to 880 k paths with the wavefront integrator (76 bytes per
// path; 156 when bounces run as launch pairs) -- twice the reference's default
// 640x480 film (src/application.cpp:206-207) at one sample per frame -- or 4 M
// paths with the stream integrator (16 bytes per path). Every GPU the
// library runs on holds one copy from module load to process end; all calls on
// a GPU share it, so calls that overlap on one GPU race.
constexpr size_t kStaticScratchBytes = (size_t)64 << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

// Debugging aid: cycles the persistent kernel's waves spent in each part of their loop (PTC_DRAIN_PROFILE=1).
static __device__ unsigned long long g_drain_prof[20];

namespace ptc
{
namespace
{

constexpr int kBlock = 256;
constexpr uint32_t kQueueSegments = 8u; // the chip's XCDs: workgroups are dealt to them round-robin
constexpr uint32_t kFillStride = 64u;   // words between the segments' fill counts: one 256-byte line each
// The primary pass walks the film in square groups of 2x2 tiles of 16x16 pixels, groups row by row, the tiles
// of a group and the pixels of a tile along a Z-order curve. (Groups of 4x4 and 8x8 tiles run at the same
// rate; the smallest leaves the fewest blocks beyond the film's edge: a 1080p film is covered as 1920x1088.)
constexpr uint32_t kGroupBits = 1u, kGroupTiles = 1u << kGroupBits, kGroupPixels = 16u << kGroupBits;

struct WfState
{
    float4 *a; // origin.xyz, rng
    float4 *b; // dir.xyz, path id
    float4 *c; // throughput.xyz, radiance.x
    float2 *d; // radiance.yz
};

struct WfBuffers
{
    WfState q[2];
    float4 *hit;      // t, u, v, flat
    uint2 *hit_ids;   // instance, primitive
    float4 *radiance; // per path id: final radiance (w unused)
    uint32_t *counts; // counts[b] = live paths entering bounce b
    // When the persistent launch follows the primary pass directly (seg_cap != 0),
    // the queue the primary pass fills is cut into kQueueSegments segments of
    // seg_cap slots: block i of the primary pass appends to segment
    // i % kQueueSegments, whose fill count is fill[kFillStride * segment]. (One
    // counter for all blocks bounds a pass in which most paths survive: the L2
    // serialises same-address atomics at about 10 ns each.)
    uint32_t *fill;
    uint32_t seg_cap;
    // RNG state a finished path hands to the next sample of the same pixel and
    // dispatch (samples_per_frame > 1 continue one stream, shader.rgen:175-179);
    // nullptr when samples_per_frame == 1
    uint32_t *carry;
    uint32_t spf;      // samples per dispatch
    uint32_t n_pixels;
    uint32_t per_pixel; // paths per pixel in this batch: dispatches * samples per dispatch
    // Regrouped hand-over (DFrame::regroup): the launch that traces next takes its
    // i-th work item from queue slot order[i] instead of slot i; nullptr = slot order.
    const uint32_t *order;
};

// path id = pixel * per_pixel + dispatch * spf + sample: the paths of a pixel lie
// next to each other, in the order the film folds them. A wave of the primary
// pass is one pixel with 64 dispatches, so the radiances it stores (three
// quarters of all paths end there in an open scene) are 1 KB in a row; with the
// dispatch as the slow index they were 64 stores a whole film apart (an all-sky
// view, which does nothing but generate rays and store radiances, took a third
// longer). The resolve pass reads the rows through LDS (see k_wf_resolve).
PTC_DEV uint32_t path_id(const WfBuffers &B, uint32_t pixel, uint32_t k, uint32_t sample)
{
    return pixel * B.per_pixel + k * B.spf + sample;
}

PTC_DEV void finish_path(const WfBuffers &B, uint32_t pid, vec3 L, uint32_t rng)
{
    B.radiance[pid] = make_float4(L.x, L.y, L.z, 1.0f);
    if (B.carry != nullptr)
    {
        const uint32_t pixel = pid / B.per_pixel, k = (pid - pixel * B.per_pixel) / B.spf;
        B.carry[k * B.n_pixels + pixel] = rng;
    }
}

PTC_DEV uint32_t lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

PTC_DEV void store_state(const WfState &q, uint32_t slot, const PathState &ps, uint32_t pid)
{
    q.a[slot] = make_float4(ps.origin.x, ps.origin.y, ps.origin.z, __uint_as_float(ps.rng));
    q.b[slot] = make_float4(ps.dir.x, ps.dir.y, ps.dir.z, __uint_as_float(pid));
    q.c[slot] = make_float4(ps.throughput.x, ps.throughput.y, ps.throughput.z, ps.radiance.x);
    q.d[slot] = make_float2(ps.radiance.y, ps.radiance.z);
}

PTC_DEV uint32_t load_state(const WfState &q, uint32_t slot, PathState &ps)
{
    const float4 a = q.a[slot], b = q.b[slot], c = q.c[slot];
    const float2 d = q.d[slot];
    ps.origin = v3(a.x, a.y, a.z);
    ps.rng = __float_as_uint(a.w);
    ps.dir = v3(b.x, b.y, b.z);
    ps.throughput = v3(c.x, c.y, c.z);
    ps.radiance = v3(c.w, d.x, d.y);
    return __float_as_uint(b.w); // path id
}

PTC_DEV PathState camera_path(const DFrame &F, uint32_t px, uint32_t py, uint32_t rng)
{
    PathState ps;
    ps.rng = rng;
    camera_ray(F, px, py, ps.rng, ps.origin, ps.dir);
    ps.throughput = v3(1.0f, 1.0f, 1.0f);
    ps.radiance = v3(0.0f, 0.0f, 0.0f);
    return ps;
}

constexpr int kPrimaryLds = 8; // LDS stack entries per lane: 16 KB per block, so registers (not LDS) set the occupancy
// kMapping: 0 = a wave is an 8x8 pixel tile of one dispatch; 1 = a wave is 2 or 4
// pixels with 32 or 16 dispatches each; 2 = a wave is one pixel with 64 dispatches.
// kCount: also count nodes / triangles visited (reproducible visit order, slower).
template <int kMapping, bool kTwoLevel, bool kCount>
__global__ void __launch_bounds__(kBlock)
k_wf_primary(DScene S, DFrame F, WfBuffers B, uint32_t frame_first, uint32_t k_first, uint32_t sample,
             uint32_t n_disp, uint32_t lane_bits, DeviceStats *__restrict__ stats)
{
    TraceCounters ctr = {0u, 0u, 0u, 0u};
    __shared__ uint2 s_stack[kPrimaryLds * kBlock];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tiles_x = (F.width + 15u) >> 4;
    const uint32_t groups_x = (tiles_x + kGroupTiles - 1u) >> kGroupBits;
    uint32_t px, py, k;
    bool inside;
    constexpr bool kPixelMajor = kMapping != 0;
    if (kPixelMajor)
    {
        // A wave is one pixel, its lanes are 64 dispatches of that pixel: the 64
        // camera rays differ by the pixel-filter jitter only, so they visit the
        // same nodes in the same order (node fetches go through the scalar
        // cache, no lane waits for a longer traversal), and the queue receives
        // the survivors of one pixel next to each other. Pixels are numbered
        // along a Z-order curve inside 16x16 tiles, tiles as below.
        // (With fewer than 64 dispatches a wave takes 2 or 4 neighbouring pixels,
        // 32 or 16 dispatches each: lane_bits = log2 of the lanes per pixel.)
        const uint32_t lanes_per_pixel = 1u << lane_bits, pixels_per_block = kBlock >> lane_bits;
        const uint32_t k_groups = (n_disp + lanes_per_pixel - 1u) >> lane_bits;
        uint32_t q = (blockIdx.x / k_groups) * pixels_per_block + (threadIdx.x >> lane_bits); // pixel number
        if (kMapping == 2)
        {
            // one pixel per wave: the pixel's coordinates, its index and the hash of
            // it are the same in every lane and are computed on the scalar unit
            q = __builtin_amdgcn_readfirstlane(q);
        }
        k = (blockIdx.x % k_groups) * lanes_per_pixel + (threadIdx.x & (lanes_per_pixel - 1u));
        const uint32_t tile = q >> 8, l = q & 255u;
        const uint32_t group = tile >> (2u * kGroupBits), m = tile & (kGroupTiles * kGroupTiles - 1u);
        const uint32_t bx = (group % groups_x) * kGroupTiles + ((m & 1u) | ((m >> 1) & 2u) | ((m >> 2) & 4u));
        const uint32_t by = (group / groups_x) * kGroupTiles + (((m >> 1) & 1u) | ((m >> 2) & 2u) | ((m >> 3) & 4u));
        px = bx * 16u + ((l & 1u) | ((l >> 1) & 2u) | ((l >> 2) & 4u) | ((l >> 3) & 8u));
        py = by * 16u + (((l >> 1) & 1u) | ((l >> 2) & 2u) | ((l >> 3) & 4u) | ((l >> 4) & 8u));
        inside = px < F.width && py < F.height && k < n_disp;
    }
    else
    {
        // A wave is an 8x8 pixel tile of one dispatch. Consecutive blocks are the
        // dispatches of one tile, so the blocks resident at any time (and the
        // queue they fill) stay within a few tiles' geometry, and those tiles
        // form a compact patch: tiles are numbered along a Z-order curve inside
        // their groups.
        const uint32_t tile = blockIdx.x / n_disp;
        k = blockIdx.x % n_disp; // dispatch within this batch
        const uint32_t group = tile >> (2u * kGroupBits), m = tile & (kGroupTiles * kGroupTiles - 1u);
        const uint32_t bx = (group % groups_x) * kGroupTiles + ((m & 1u) | ((m >> 1) & 2u) | ((m >> 2) & 4u));
        const uint32_t by = (group / groups_x) * kGroupTiles + (((m >> 1) & 1u) | ((m >> 2) & 2u) | ((m >> 3) & 4u));
        px = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
        py = by * 16u + (wave >> 1) * 8u + (lane >> 3);
        inside = px < F.width && py < F.height; // false for all lanes of tiles past the edge
    }
    bool alive = false;
    PathState ps; // (lanes outside the film never read it)
    ps.origin = ps.dir = ps.throughput = ps.radiance = v3(0.0f, 0.0f, 0.0f);
    ps.rng = 0u;
    uint32_t pid = 0;
    uint32_t ended_in = 0u; // the path loop's counter when this pass ends the path
    if (inside)
    {
        const uint32_t pixel_index = py * F.width + px;
        const uint32_t n_pixels = F.width * F.height;
        pid = path_id(B, pixel_index, k, sample);
        ps = camera_path(F, px, py,
                         sample == 0u ? seed_(pixel_index, frame_first + (k_first + k) * F.frame_stride)
                                      : B.carry[k * n_pixels + pixel_index]);
        alive = F.max_depth > 0u;
        if (alive)
        {
            // Camera rays start outside the scene more often than not: one box
            // test spares those the visit of the root node. (A wave holds samples
            // of one pixel or of one tile, so its lanes mostly agree.)
            Hit h = no_hit(F.tmax);
            if (kTwoLevel || ray_meets_box_fast(S.box_lo, S.box_hi, ps.origin, ps.dir, F.tmax))
            {
                h = trace_scene<kCount, kTwoLevel, kPrimaryLds>(S, ps.origin, ps.dir, F.tmax, s_stack + threadIdx.x,
                                                                kBlock, &ctr);
            }
            const bool goes_on = path_bounce(S, h, ps);
            alive = goes_on && F.max_depth > 1u;
            ended_in = goes_on ? 1u : 0u;
        }
        if (!alive)
        {
            finish_path(B, pid, sample_value(F.debug_view, ps.radiance, ended_in), ps.rng);
        }
    }
    // One atomic per block (the four waves' counts are combined in LDS): with
    // one per wave, a scene where most camera paths survive is bound by the
    // rate at which the L2 can add to that one address.
    __shared__ uint32_t s_alive[kBlock / 64];
    __shared__ uint32_t s_base;
    const unsigned long long alive_mask = __ballot(alive);
    if (lane == 0u)
    {
        s_alive[wave] = (uint32_t)__builtin_popcountll(alive_mask);
    }
    __syncthreads();
    if (threadIdx.x == 0u)
    {
        const uint32_t total = s_alive[0] + s_alive[1] + s_alive[2] + s_alive[3];
        const uint32_t seg = blockIdx.x % kQueueSegments;
        uint32_t *const counter = B.seg_cap != 0u ? B.fill + kFillStride * seg : &B.counts[1];
        s_base = total != 0u ? seg * B.seg_cap + atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (alive)
    {
        uint32_t slot = s_base + (uint32_t)__builtin_popcountll(alive_mask & ((1ull << lane) - 1ull));
        for (uint32_t w = 0; w < wave; ++w)
        {
            slot += s_alive[w];
        }
        store_state(B.q[1], slot, ps, pid);
    }
    if (kCount)
    {
        unsigned long long v[4] = {ctr.nodes, ctr.tris, ctr.uniform_nodes, ctr.uniform_fetches};
        for (int k = 0; k < 4; ++k)
        {
            for (int off = 32; off > 0; off >>= 1)
            {
                v[k] += __shfl_xor(v[k], off, 64);
            }
        }
        if (lane == 0u)
        {
            atomicAdd(&stats->nodes, v[0]);
            atomicAdd(&stats->tris, v[1]);
            atomicAdd(&stats->uniform_nodes, v[2]);
            atomicAdd(&stats->uniform_fetches, v[3]);
        }
    }
}

__global__ void __launch_bounds__(kBlock)
k_wf_trace(DScene S, float tmax, WfState q, float4 *__restrict__ hit, uint2 *__restrict__ hit_ids,
           const uint32_t *__restrict__ count, const uint32_t *__restrict__ order)
{
    __shared__ uint2 s_stack[kLdsStack * kBlock];
    const uint32_t n = *count;
    // (64-bit loop counters in the grid-stride loops of this file: a queue may hold close to 2^32 paths)
    for (uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (uint64_t)gridDim.x * kBlock)
    {
        const uint32_t i = order != nullptr ? order[j] : (uint32_t)j; // the j-th ray traced is the one of slot i
        const float4 a = q.a[i], b = q.b[i];
        const vec3 o = v3(a.x, a.y, a.z), d = v3(b.x, b.y, b.z);
        const Hit h = S.inst_box != nullptr
                          ? trace_scene<false, true>(S, o, d, tmax, s_stack + threadIdx.x, kBlock, nullptr)
                          : trace_scene<false, false>(S, o, d, tmax, s_stack + threadIdx.x, kBlock, nullptr);
        hit[i] = make_float4(h.t, h.u, h.v, __uint_as_float(h.flat));
        hit_ids[i] = make_uint2(h.instance, h.primitive);
    }
}

// Shading of one bounce for every queue slot: material / sky, emission, Russian
// roulette; survivors go to the other queue. 1024 threads per block and one
// append per block and round (the sixteen waves' counts are combined in LDS):
// adds of every wave of the chip to the one counter would serialise in L2 at
// about 10 ns each, 16 ms for the 1.6 M waves of a 100 M-path bounce.
constexpr int kShadeBlock = 1024;
__global__ void __launch_bounds__(kShadeBlock)
k_wf_shade(DScene S, WfBuffers B, uint32_t bounce, uint32_t max_depth, int src, uint32_t debug_view)
{
    __shared__ uint32_t s_alive[kShadeBlock / 64];
    const WfState qi = B.q[src], qo = B.q[src ^ 1];
    const uint32_t n = B.counts[bounce];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t n_round = ((uint64_t)n + kShadeBlock - 1u) / kShadeBlock * kShadeBlock; // whole blocks enter: every thread meets the barriers
    for (uint64_t i64 = (uint64_t)blockIdx.x * kShadeBlock + threadIdx.x; i64 < n_round; i64 += (uint64_t)gridDim.x * kShadeBlock)
    {
        const uint32_t i = (uint32_t)i64; // (only used when i64 < n)
        bool alive = false;
        PathState ps;
        uint32_t pid = 0;
        if (i64 < n)
        {
            pid = load_state(qi, i, ps);
            const float4 hr = B.hit[i];
            const uint2 hid = B.hit_ids[i];
            Hit h;
            h.t = hr.x;
            h.u = hr.y;
            h.v = hr.z;
            h.flat = __float_as_uint(hr.w);
            h.instance = hid.x;
            h.primitive = hid.y;
            const bool goes_on = path_bounce(S, h, ps);
            alive = goes_on && (bounce + 1u) < max_depth;
            if (!alive)
            {
                finish_path(B, pid, sample_value(debug_view, ps.radiance, goes_on ? bounce + 1u : bounce), ps.rng);
            }
        }
        const unsigned long long mask = __ballot(alive);
        if (lane == 0u)
        {
            s_alive[wave] = (uint32_t)__builtin_popcountll(mask);
        }
        __syncthreads();
        if (threadIdx.x < 64u)
        {
            // inclusive prefix sums of the sixteen wave counts, then one atomic
            uint32_t v = threadIdx.x < kShadeBlock / 64 ? s_alive[threadIdx.x] : 0u;
            for (int off = 1; off < kShadeBlock / 64; off <<= 1)
            {
                const uint32_t up = __shfl_up(v, off, 64);
                v += lane >= (uint32_t)off ? up : 0u;
            }
            const uint32_t total = __shfl(v, kShadeBlock / 64 - 1, 64);
            uint32_t base = 0u;
            if (threadIdx.x == 0u && total != 0u)
            {
                base = atomicAdd(&B.counts[bounce + 1u], total);
            }
            base = __shfl(base, 0, 64);
            if (threadIdx.x < kShadeBlock / 64)
            {
                s_alive[threadIdx.x] = base + v - s_alive[threadIdx.x]; // first slot of each wave
            }
        }
        __syncthreads();
        if (alive)
        {
            const uint32_t slot = s_alive[wave] + (uint32_t)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
            store_state(qo, slot, ps, pid);
        }
        __syncthreads(); // s_alive is rewritten in the next round
    }
}

// ---- regrouping the queue between bounces -------------------------------------
// The paths of a queue are handed to the next tracing launch sorted by a 15-bit
// key of their ray: the cell of the origin in a 16 x 16 x 16 grid over the scene's
// box, along a Morton curve (high bits), and the direction's octant (low bits).
// A counting sort in three launches, none of which needs the number of paths on
// the host: keys and a histogram (one atomic per wave and distinct key: the queue
// arrives in pixel order, so the lanes of a wave mostly share a key), an exclusive
// scan of the 32,768 bins, and a scatter of the slot numbers into order[]. Paths
// with equal keys end up in no particular order; results do not depend on it
// (every path carries its own RNG state).
constexpr uint32_t kRegroupBins = 1u << 15;

PTC_DEV uint32_t spread4(uint32_t v) // 4 bits -> every third bit
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

PTC_DEV uint32_t regroup_key(const DScene &S, vec3 o, vec3 d)
{
    float4 lo = S.box_lo, hi = S.box_hi;
    if (S.inst_box != nullptr && S.n_inst_nodes != 0u)
    {
        lo = S.inst_box[0]; // the root of the instance tree: the box of all instances
        hi = S.inst_box[1];
    }
    const float fx = (o.x - lo.x) / (hi.x - lo.x), fy = (o.y - lo.y) / (hi.y - lo.y), fz = (o.z - lo.z) / (hi.z - lo.z);
    // (NaN and out-of-box origins clamp to an edge cell: any key is a valid key)
    const uint32_t cx = (uint32_t)fmin_(fmax_(fx * 16.0f, 0.0f), 15.0f);
    const uint32_t cy = (uint32_t)fmin_(fmax_(fy * 16.0f, 0.0f), 15.0f);
    const uint32_t cz = (uint32_t)fmin_(fmax_(fz * 16.0f, 0.0f), 15.0f);
    const uint32_t cell = spread4(cx) | (spread4(cy) << 1) | (spread4(cz) << 2);
    const uint32_t oct = (__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2);
    return (cell << 3) | oct;
}

// Is slot i of the queue filled? Segmented queue (seg_cap != 0): slot i lies in
// segment i / seg_cap, which holds fill[kFillStride * segment] paths; else the
// queue is one run of *count paths.
PTC_DEV bool slot_filled(uint32_t i, uint32_t seg_cap, const uint32_t *fill, const uint32_t *count)
{
    if (seg_cap != 0u)
    {
        const uint32_t seg = i / seg_cap;
        return seg < kQueueSegments && i - seg * seg_cap < fill[kFillStride * seg];
    }
    return i < *count;
}

__global__ void __launch_bounds__(kBlock)
k_regroup_keys(DScene S, WfState q, uint32_t n_slots, uint32_t seg_cap, const uint32_t *__restrict__ fill,
               const uint32_t *__restrict__ count, uint32_t *__restrict__ keys, uint32_t *__restrict__ bins)
{
    for (uint64_t i64 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i64 < (((uint64_t)n_slots + 63u) & ~(uint64_t)63u);
         i64 += (uint64_t)gridDim.x * kBlock)
    {
        const uint32_t i = (uint32_t)i64;
        const bool valid = i64 < n_slots && slot_filled(i, seg_cap, fill, count);
        uint32_t key = kRegroupBins; // (no key)
        if (valid)
        {
            const float4 a = q.a[i], b = q.b[i];
            key = regroup_key(S, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z));
            keys[i] = key;
        }
        // one atomic per distinct key among the wave's lanes
        unsigned long long todo = __ballot(valid);
        while (todo != 0ull)
        {
            const uint32_t k0 = __shfl(key, (int)__builtin_ctzll(todo), 64);
            const unsigned long long same = __ballot(valid && key == k0);
            if (lane_id() == (uint32_t)__builtin_ctzll(todo))
            {
                atomicAdd(&bins[k0], (uint32_t)__builtin_popcountll(same));
            }
            todo &= ~same;
        }
    }
}

// Exclusive scan of the bins (one block of 1024 threads, 32 bins each); the total
// goes to *total_out, where the next launch reads its number of work items.
__global__ void __launch_bounds__(1024)
k_regroup_scan(uint32_t *__restrict__ bins, uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t s_wave[16];
    constexpr uint32_t kPer = kRegroupBins / 1024u;
    uint32_t local[kPer];
    uint32_t sum = 0u;
    for (uint32_t k = 0; k < kPer; ++k)
    {
        local[k] = sum;
        sum += bins[threadIdx.x * kPer + k];
    }
    // exclusive scan of the threads' sums: within the wave, then over the 16 waves
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t up = __shfl_up(incl, off, 64);
        incl += lane >= (uint32_t)off ? up : 0u;
    }
    if (lane == 63u)
    {
        s_wave[wave] = incl;
    }
    __syncthreads();
    uint32_t before = 0u;
    for (uint32_t w = 0; w < wave; ++w)
    {
        before += s_wave[w];
    }
    const uint32_t base = before + incl - sum;
    for (uint32_t k = 0; k < kPer; ++k)
    {
        bins[threadIdx.x * kPer + k] = base + local[k];
    }
    if (threadIdx.x == 1023u)
    {
        *total_out = base + sum;
    }
}

__global__ void __launch_bounds__(kBlock)
k_regroup_scatter(uint32_t n_slots, uint32_t seg_cap, const uint32_t *__restrict__ fill, const uint32_t *__restrict__ count,
                  const uint32_t *__restrict__ keys, uint32_t *__restrict__ bins, uint32_t *__restrict__ order)
{
    for (uint64_t i64 = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i64 < (((uint64_t)n_slots + 63u) & ~(uint64_t)63u);
         i64 += (uint64_t)gridDim.x * kBlock)
    {
        const uint32_t i = (uint32_t)i64;
        const bool valid = i64 < n_slots && slot_filled(i, seg_cap, fill, count);
        const uint32_t key = valid ? keys[i] : kRegroupBins;
        // the lanes of a wave that share a key take consecutive places, in lane
        // order: neighbouring slots stay neighbours, so the gather through order[]
        // reads runs of consecutive slots
        unsigned long long todo = __ballot(valid);
        while (todo != 0ull)
        {
            const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
            const uint32_t k0 = __shfl(key, (int)leader, 64);
            const unsigned long long same = __ballot(valid && key == k0);
            uint32_t base = 0u;
            if (lane_id() == leader)
            {
                base = atomicAdd(&bins[k0], (uint32_t)__builtin_popcountll(same));
            }
            base = __shfl(base, (int)leader, 64);
            if (valid && key == k0)
            {
                order[base + (uint32_t)__builtin_popcountll(same & ((1ull << lane_id()) - 1ull))] = i;
            }
            todo &= ~same;
        }
    }
}

// One triangle test of the persistent kernel. In a two-level scene the records
// of a mesh's tree carry no instance: a test that improved the hit marks it
// with the instance the lane is in.
template <bool kTwoLevel>
PTC_DEV void drain_tri_eval(const float4 a, const float4 b, const float4 c, const RayCtx &R, uint32_t inst, Hit &best)
{
    if (!kTwoLevel)
    {
        tri_eval(a, b, c, R, best);
        return;
    }
    const float t0 = best.t;
    const uint32_t f0 = best.instance == inst ? best.flat : kNoHit;
    Hit cand = best;
    cand.flat = f0;
    tri_eval(a, b, c, R, cand);
    const bool improved = cand.t != t0 || cand.flat != f0;
    best.t = improved ? cand.t : best.t;
    best.u = improved ? cand.u : best.u;
    best.v = improved ? cand.v : best.v;
    best.flat = improved ? cand.flat : best.flat;
    best.instance = improved ? inst : best.instance;
}

template <bool kTwoLevel>
PTC_DEV void drain_tri_test(const DScene &S, const RayCtx &R, uint32_t tri_index, uint32_t inst, Hit &best)
{
    if (!kTwoLevel)
    {
        tri_test(S, R, tri_index, best);
        return;
    }
    // best.flat counts within its instance's mesh: it decides a tie in t only
    // against a hit of the same instance. Against another instance's hit, a
    // tie that got past the ray's t limit wins (see instance_t_limit()).
    const float t0 = best.t;
    const uint32_t f0 = best.instance == inst ? best.flat : kNoHit;
    Hit cand = best;
    cand.flat = f0;
    tri_test(S, R, tri_index, cand);
    const bool improved = cand.t != t0 || cand.flat != f0;
    best.t = improved ? cand.t : best.t;
    best.u = improved ? cand.u : best.u;
    best.v = improved ? cand.v : best.v;
    best.flat = improved ? cand.flat : best.flat;
    best.instance = improved ? inst : best.instance;
}

// Finishes every path still alive at `bounce` in one launch. Once few paths are
// left, a launch pair per bounce costs more in launch latency than in work, so
// the tail runs persistent: each lane pulls a live path from the queue (one
// atomic per wave), walks it bounce by bounce to its end, then pulls the next,
// which keeps the lanes of a wave busy until the queue is empty. Every wave
// leaves the loop once the queue is exhausted and its own lanes are done.
// Shape of the persistent kernel. 256 threads; 6 LDS stack entries per lane (two
// pushes per level; deeper entries go to scratch: 8 entries are no faster, 4
// cost 6 %); built for 5 resident waves per SIMD (96 registers). The kernel is
// bound by memory latency that its resident waves have to cover: 2 / 3 / 4 / 5
// waves per SIMD finish the 871k-triangle scene's 256-spp launch in 66.6 / 49.3
// / 41.6 / 38.4 ms. Six need 80 registers, which the compiler only reaches by
// spilling loop invariants to scratch (43.9 ms and worse).
// (Camera items keep three more words per lane in LDS, which leaves room for 4.)
constexpr int kDrainBlock = 256, kDrainLds = 6;
#ifndef PTC_INST_MIN
#define PTC_INST_MIN 12
#endif
constexpr int drain_waves(bool camera)
{
    return camera ? 4 : 5;
}

// kCamera = false: work items are the live paths of queue B.q[src] (n read from
// B.counts[bounce0]). kCamera = true: work items are (pixel, dispatch) pairs,
// n_items of them, generated in place as camera rays: item i covers lane
// i % 64 of 8x8 pixel tile (i / 64) / n_disp for dispatch (i / 64) % n_disp, so
// a wave that pulls 64 items at once gets one coherent tile; the
// samples_per_frame samples of an item run back to back in the same lane on
// one RNG stream, as shader.rgen:179 does.
// kProfile: also sums, per wave, the cycles spent in each part of the loop
// (g_drain_prof; a debugging aid that serialises the loop's memory waits).
template <bool kCamera, bool kTwoLevel, bool kProfile>
__global__ void __launch_bounds__(kDrainBlock, drain_waves(kCamera))
k_wf_drain(DScene S, DFrame F, WfBuffers B, uint32_t bounce0, int src, uint32_t n_items, uint32_t n_disp,
           uint32_t k_first, uint32_t *__restrict__ head, DeviceStats *__restrict__ stats)
{
    __shared__ uint2 s_stack[kDrainLds * kDrainBlock];
    // What a path carries but only its shading needs (throughput, radiance, RNG
    // state, path id, bounce number; for camera items also the pixel and the
    // sample number) stays in LDS, one word per lane and field, and is in
    // registers only inside the shading and pull phases. Ten more words per lane
    // take the traversal state of the lanes that are not shaded during a shading
    // phase, whose own register need is the kernel's largest. The traversal
    // loop, whose register count sets the number of resident waves, then carries
    // the ray, the hit and the traversal state only (119 -> 93 registers).
    constexpr int kCold = kCamera ? 12 : 9, kParked = 10;
    __shared__ uint32_t s_words[(kCold + kParked) * kDrainBlock];
#define PTC_WORD(w) s_words[(w) * kDrainBlock + threadIdx.x]
    const WfState qi = B.q[src];
    const uint32_t max_depth = F.max_depth;
    const float tmax = F.tmax;
    // queue-fed: the primary pass filled the queue by segments (B.seg_cap slots
    // each, B.fill[] used), or the queue is one run of B.counts[bounce0] items
    const bool by_fill = !kCamera && B.seg_cap != 0u;
    const uint32_t n = kCamera ? n_items : (by_fill ? kQueueSegments * B.seg_cap : B.counts[bounce0]);
    const uint32_t lane = lane_id();
    // Lane state machine. A lane is idle (kIdle), walking the tree for its
    // path's current ray (kTrav), or has its closest hit and waits to be shaded
    // (kShade). Every loop iteration runs at most one node visit per lane, so
    // lanes with short and long traversals do not wait for each other; the
    // rarer phases (pulling new paths, shading, triangle tests) run when
    // enough lanes of the wave want them, or when a lane has nothing else left.
    // (two-level scenes: kInst, between two instances: the lane's traversal of one
    // instance's tree is complete, or its ray has just started, and it waits to be
    // taken into the next instance on its list)
    constexpr uint32_t kIdle = 0u, kTrav = 1u, kShade = 2u, kInst = 3u;
    // lanes that must want a phase before it runs (swept with the traversal state parked in LDS: 12 to 24 lanes
    // for each of the three are within 1 % of one another on the 871k-triangle scene, 8 and 32 cost 3 to 7 %;
    // scenes whose paths are shaded more often per node visited want the shading quorum higher: 16 / 20 / 24 /
    // 32 lanes give 36.1 / 36.0 / 36.4 / 37.4 ms there, 103.2 / 101.6 / 101.4 / 105.8 with the mesh inside
    // the Cornell box, 27.3 / 25.3 / 24.8 / 24.2 on the Cornell box alone)
    constexpr uint32_t kFetchMin = kCamera ? 32u : 16u, kTriMoreMin = 16u;
    // A tree of a few nodes (the Cornell box has 3) is walked in two or three iterations, so most of such a
    // scene's time goes into shading phases, and they pay for waiting until more than half the wave wants one:
    // 20 / 32 / 40 / 48 lanes give 24.0 / 23.0 / 22.8 / 22.9 ms on the Cornell box at 1024^2; a 54k-triangle mesh
    // is indifferent (20 / 26 / 32: 23.8 ms each), larger ones lose beyond 24 (above).
#ifndef PTC_SHADE_MIN_SMALL
#define PTC_SHADE_MIN_SMALL 40
#endif
#ifndef PTC_SMALL_TREE
#define PTC_SMALL_TREE 64
#endif
    const uint32_t kShadeMin = S.n_nodes <= (uint32_t)PTC_SMALL_TREE ? (uint32_t)PTC_SHADE_MIN_SMALL : 20u;
    // two-level scenes: lanes between two instances that must wait before the wave
    // takes them into their next instances together
    constexpr uint32_t kInstMin = PTC_INST_MIN;
    uint32_t mode = kIdle;
    constexpr uint32_t kChunk = 256u; // work items a wave reserves per atomic (64: Cornell box 17 % slower; 128, 512: within 0.5 %)
    uint32_t chunk_next = 0u, chunk_end = 0u; // this wave's reserved items not handed out yet
    // The queue is cut into kSegments contiguous segments, each with a head of
    // its own; a wave pulls from the segment of its XCD first (workgroups are
    // dealt to the chip's 8 XCDs round-robin) and moves on to the next segment
    // when that one is used up. Eight heads take an eighth of the same-address
    // atomics each (the Cornell box, where every path survives and pulls are
    // frequent, gains 16 %), and one L2 serves paths from one part of the
    // picture (no measurable effect on the 871k-triangle scene).
    constexpr uint32_t kSegments = kQueueSegments;
    const uint32_t seg_len = by_fill ? B.seg_cap : ((n + kSegments * kChunk - 1u) / (kSegments * kChunk)) * kChunk;
    uint32_t seg = blockIdx.x % kSegments, segs_left = kSegments;
    bool queue_done = n == 0u;                // nothing left to reserve beyond chunk_end
    bool exhausted = n == 0u;
    // the lane's current ray in world space, and its per-ray traversal constants
    vec3 org = v3(0.0f, 0.0f, 0.0f), dir = v3(0.0f, 0.0f, 1.0f);
    float idx = 1.0f, idy = 1.0f, idz = 1.0f;
    uint32_t octinv = 7u;
    // Two-level scenes: the ray in the object space of the instance whose tree the
    // lane is in (with its t limit there), that instance, and the lane's place in
    // the instance tree; best.flat is then the primitive index within the mesh of
    // best.instance. (Flattened scenes trace the world-space ray itself.)
    vec3 obj_o = org, obj_d = dir;
    float obj_tmax = tmax;
    // inst_pos == kListed: the instances the ray can meet are listed nearest first in
    // inst_list (see nearest_instances()); otherwise the list did not hold them and
    // inst_pos is the lane's place in the instance tree, walked in its own order
    constexpr uint32_t kListed = 0xffffffffu;
    uint32_t inst = 0u, inst_pos = 0u, inst_list = kNoCandidates;
    auto ray_ctx = [&]() {
        RayCtx R;
        R.o = kTwoLevel ? obj_o : org;
        R.d = kTwoLevel ? obj_d : dir;
        R.idx = idx;
        R.idy = idy;
        R.idz = idz;
        R.octinv = octinv;
        R.tmax = kTwoLevel ? obj_tmax : tmax;
        return R;
    };
    auto set_ray = [&](vec3 o, vec3 d, float limit) {
        const RayCtx R = make_ray_ctx(o, d, limit);
        idx = R.idx;
        idy = R.idy;
        idz = R.idz;
        octinv = R.octinv;
        if (kTwoLevel)
        {
            obj_o = o;
            obj_d = d;
            obj_tmax = limit;
        }
    };
    auto store_cold = [&](const PathState &ps, uint32_t pid, uint32_t bounce) {
        PTC_WORD(0) = __float_as_uint(ps.throughput.x);
        PTC_WORD(1) = __float_as_uint(ps.throughput.y);
        PTC_WORD(2) = __float_as_uint(ps.throughput.z);
        PTC_WORD(3) = __float_as_uint(ps.radiance.x);
        PTC_WORD(4) = __float_as_uint(ps.radiance.y);
        PTC_WORD(5) = __float_as_uint(ps.radiance.z);
        PTC_WORD(6) = ps.rng;
        PTC_WORD(7) = pid;
        PTC_WORD(8) = bounce;
    };
    uint32_t wave_rays = 0u; // rays started by this wave (wave-uniform)
    Hit best = no_hit(tmax);
    uint2 ng = make_uint2(0u, 0u), tg = make_uint2(0u, 0u);
    uint2 *const lds = s_stack + threadIdx.x;
    uint2 spill[kStackTotal - kDrainLds];
    int sp = 0;
    // kProfile: cycles per loop part (0 loop top, 1 shading, 2 pull, 3 ray start
    // and stack pop, 4 node visit, 5 triangle tests, 6 further triangle rounds),
    // then 7 further rounds run, 8 shading phases, 9 iterations
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = kProfile ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_begin = t_prev;
    unsigned long long t_exhausted = 0ull; // kProfile: when the wave found the queue empty
    uint32_t ray_iters = 0u, ray_iters_max = 0u; // kProfile: loop iterations of the lane's current ray, and the most any of its rays took
    // kProfile: stack entries this lane pushed, how many of them beyond the LDS part (to scratch memory), entries it
    // popped from there, and its deepest stack
    uint32_t n_push = 0u, n_push_spill = 0u, n_pop_spill = 0u, sp_max = 0u;
#define PTC_PROF(k)                                                     \
    if (kProfile)                                                       \
    {                                                                   \
        __builtin_amdgcn_s_waitcnt(0);                                  \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();  \
        prof[k] += t_now - t_prev;                                      \
        t_prev = t_now;                                                 \
    }
    for (;;)
    {
        const unsigned long long trav = __ballot(mode == kTrav);
        bool start = false; // this lane starts tracing a new ray in this iteration
        // ---- shade lanes whose traversal is complete
        const unsigned long long shade = __ballot(mode == kShade);
        const uint32_t n_shade = (uint32_t)__builtin_popcountll(shade);
        const bool run_shade = n_shade != 0u && (n_shade >= kShadeMin || trav == 0ull);
        PTC_PROF(0)
        if (run_shade)
        {
            // park what the lanes that are not shaded still need (the compiler
            // barriers keep it from holding the values in registers all the same)
            PTC_WORD(kCold + 0) = __float_as_uint(idx);
            PTC_WORD(kCold + 1) = __float_as_uint(idy);
            PTC_WORD(kCold + 2) = __float_as_uint(idz);
            PTC_WORD(kCold + 3) = octinv;
            PTC_WORD(kCold + 4) = ng.x;
            PTC_WORD(kCold + 5) = ng.y;
            PTC_WORD(kCold + 6) = tg.x;
            PTC_WORD(kCold + 7) = tg.y;
            PTC_WORD(kCold + 8) = (uint32_t)sp;
            PTC_WORD(kCold + 9) = __float_as_uint(best.t);
            __asm__ volatile("" ::: "memory");
            if (mode == kShade)
            {
                if (kTwoLevel && best.flat != kNoHit)
                {
                    best.primitive = best.flat;
                    best.flat += S.instances[best.instance].first_flat;
                }
                PathState ps;
                ps.origin = org;
                ps.dir = dir;
                ps.throughput = v3(__uint_as_float(PTC_WORD(0)), __uint_as_float(PTC_WORD(1)), __uint_as_float(PTC_WORD(2)));
                ps.radiance = v3(__uint_as_float(PTC_WORD(3)), __uint_as_float(PTC_WORD(4)), __uint_as_float(PTC_WORD(5)));
                ps.rng = PTC_WORD(6);
                uint32_t pid = PTC_WORD(7);
                uint32_t bounce = PTC_WORD(8);
                const bool goes_on = path_bounce(S, best, ps);
                const bool alive = goes_on && (bounce + 1u) < max_depth;
                // the path loop's counter if the path ends here (shader.rgen:65, :193)
                const uint32_t ended_in = goes_on ? bounce + 1u : bounce;
                ++bounce;
                if (alive)
                {
                    start = true;
                }
                else if (kCamera)
                {
                    const vec3 value = sample_value(F.debug_view, ps.radiance, ended_in);
                    B.radiance[pid] = make_float4(value.x, value.y, value.z, 1.0f);
                    const uint32_t sample = PTC_WORD(11) + 1u;
                    if (sample < B.spf)
                    {
                        // next sample of the same dispatch continues the RNG stream
                        pid += 1u;
                        ps = camera_path(F, PTC_WORD(9), PTC_WORD(10), ps.rng);
                        bounce = 0u;
                        start = true;
                        PTC_WORD(11) = sample;
                    }
                    else
                    {
                        mode = kIdle;
                    }
                }
                else
                {
                    finish_path(B, pid, sample_value(F.debug_view, ps.radiance, ended_in), ps.rng);
                    mode = kIdle;
                }
                org = ps.origin;
                dir = ps.dir;
                store_cold(ps, pid, bounce);
            }
            __asm__ volatile("" ::: "memory");
            idx = __uint_as_float(PTC_WORD(kCold + 0));
            idy = __uint_as_float(PTC_WORD(kCold + 1));
            idz = __uint_as_float(PTC_WORD(kCold + 2));
            octinv = PTC_WORD(kCold + 3);
            ng.x = PTC_WORD(kCold + 4);
            ng.y = PTC_WORD(kCold + 5);
            tg.x = PTC_WORD(kCold + 6);
            tg.y = PTC_WORD(kCold + 7);
            sp = (int)PTC_WORD(kCold + 8);
            best.t = __uint_as_float(PTC_WORD(kCold + 9));
            if (kProfile)
            {
                prof[8] += 1;
            }
        }
        PTC_PROF(1)
        // ---- pull new paths for idle lanes: when enough are idle, and right
        // after a shading phase (the lanes whose paths just ended restart together
        // with the lanes whose paths continue, instead of waiting for company)
        const unsigned long long need = __ballot(mode == kIdle);
        const uint32_t n_need = (uint32_t)__builtin_popcountll(need);
        if (!exhausted && n_need != 0u && (n_need >= kFetchMin || run_shade || trav == 0ull))
        {
            // Work items come out of a chunk the wave reserved with one atomic;
            // the next chunk is reserved when this one is used up. (Every wave of
            // the chip adding to one address serialises in L2 at several ns per
            // atomic: one atomic per pull would bound the whole kernel.) A pull
            // that the rest of the chunk cannot serve fully leaves the other
            // lanes idle until the next pull.
            while (chunk_next == chunk_end && !queue_done)
            {
                const uint32_t leader = (uint32_t)__builtin_ctzll(need);
                uint32_t fresh = 0;
                if (lane == leader)
                {
                    fresh = atomicAdd(head + seg, kChunk);
                }
                fresh = __shfl(fresh, (int)leader, 64);
                const uint32_t lo = seg * seg_len;
                const uint32_t hi = by_fill ? lo + B.fill[kFillStride * seg] : (lo + seg_len < n ? lo + seg_len : n);
                if (lo < hi && fresh < hi - lo)
                {
                    chunk_next = lo + fresh;
                    chunk_end = chunk_next + kChunk < hi ? chunk_next + kChunk : hi;
                }
                else
                {
                    seg = (seg + 1u) % kSegments;
                    queue_done = --segs_left == 0u;
                }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t rank = (uint32_t)__builtin_popcountll(need & ((1ull << lane) - 1ull));
            const uint32_t i = chunk_next + rank;
            bool got = mode == kIdle && rank < avail;
            chunk_next += n_need < avail ? n_need : avail;
            if (got && kCamera)
            {
                const uint32_t tile_k = i >> 6, l = i & 63u;
                const uint32_t tile = tile_k / n_disp, k = tile_k % n_disp;
                const uint32_t tiles_x = (F.width + 7u) >> 3;
                const uint32_t px = (tile % tiles_x) * 8u + (l & 7u);
                const uint32_t py = (tile / tiles_x) * 8u + (l >> 3);
                got = px < F.width && py < F.height;
                if (got)
                {
                    const uint32_t pixel_index = py * F.width + px;
                    const PathState ps = camera_path(F, px, py, seed_(pixel_index, F.global_frame_count +
                                                                                       (k_first + k) * F.frame_stride));
                    org = ps.origin;
                    dir = ps.dir;
                    store_cold(ps, path_id(B, pixel_index, k, 0u), 0u);
                    PTC_WORD(9) = px;
                    PTC_WORD(10) = py;
                    PTC_WORD(11) = 0u;
                }
            }
            else if (got)
            {
                PathState ps;
                const uint32_t pid = load_state(qi, B.order != nullptr ? B.order[i] : i, ps);
                org = ps.origin;
                dir = ps.dir;
                store_cold(ps, pid, bounce0);
            }
            start = start || got;
            exhausted = queue_done && chunk_next == chunk_end; // wave-uniform
            if (kProfile && exhausted && t_exhausted == 0ull)
            {
                t_exhausted = __builtin_amdgcn_s_memtime();
            }
        }
        PTC_PROF(2)
        wave_rays += (uint32_t)__builtin_popcountll(__ballot(start));
        if (kProfile)
        {
            ray_iters = start ? 0u : ray_iters + 1u;
            ray_iters_max = ray_iters > ray_iters_max ? ray_iters : ray_iters_max;
        }
        if (start)
        {
            best = no_hit(tmax);
            const bool finite = ray_is_finite(org, dir); // a non-finite ray hits nothing and must not enter the tree
            if (kTwoLevel)
            {
                // the instances the ray can meet, nearest first; the first is entered below
                inst_pos = kListed;
                inst_list = kNoCandidates; // (a non-finite ray meets none)
                if (finite && !nearest_instances(S, org, dir, tmax, inst_list))
                {
                    inst_pos = 0u;
                }
                ng = make_uint2(0u, 0u);
            }
            else
            {
                set_ray(org, dir, tmax);
                ng = make_uint2(0u, S.n_tris != 0u && finite ? 0x80000000u : 0u);
            }
            tg = make_uint2(0u, 0u);
            sp = 0;
            mode = kTwoLevel ? kInst : kTrav;
        }
        if (exhausted && __ballot(mode != kIdle) == 0ull)
        {
            break;
        }
        // ---- two-level scenes: lanes between two instances enter their next one.
        // Finding it and taking the ray to its object space (a box test or a few,
        // the instance's matrix, three divisions) is the heavy step of such a
        // scene; run whenever one lane wanted it, it kept the other 63 waiting
        // once or twice per loop iteration. So the lanes wait for company, as they
        // do for shading: the step runs when enough of them wait, right after a
        // shading or pull phase (the rays started there all need their first
        // instance), or when no lane of the wave is traversing.
        if (kTwoLevel)
        {
            const unsigned long long waiting = __ballot(mode == kInst);
            const uint32_t n_waiting = (uint32_t)__builtin_popcountll(waiting);
            if (n_waiting != 0u && (n_waiting >= kInstMin || __ballot(start) != 0ull || __ballot(mode == kTrav) == 0ull) &&
                mode == kInst)
            {
                uint32_t next = kNoInstance;
                if (inst_pos == kListed)
                {
                    while (inst_list != kNoCandidates)
                    {
                        const uint32_t id = inst_list & 0xffu;
                        inst_list = (inst_list >> 8) | 0xff000000u;
                        // once there is a hit, the box is tested again: the hit may rule it out by now
                        if (best.flat == kNoHit ||
                            ray_meets_box_fast(S.inst_bounds[2u * id], S.inst_bounds[2u * id + 1u], org, dir, best.t))
                        {
                            next = id;
                            break;
                        }
                    }
                }
                else
                {
                    next = next_instance(S, inst_pos, org, dir, best.t);
                }
                mode = kShade;
                if (next != kNoInstance)
                {
                    inst = next;
                    const DInstance *__restrict__ I = &S.instances[inst];
                    set_ray(xform_point(I->w2o, org), xform_vector(I->w2o, dir), instance_t_limit(best, inst));
                    ng = make_uint2(I->blas_root, 0x80000000u);
                    mode = kTrav;
                }
            }
        }
        // ---- out of local work: pop the stack, or the traversal is complete
        if (mode == kTrav && !(ng.y & kNodeBits) && tg.y == 0u)
        {
            if (sp == 0)
            {
                mode = kTwoLevel ? kInst : kShade; // (two-level: on to the next instance, above, in a later iteration)
            }
            else
            {
                if (kProfile && sp > kDrainLds)
                {
                    ++n_pop_spill;
                }
                const uint2 e = stack_pop<kDrainLds>(lds, kDrainBlock, spill, sp);
                if (e.y & kNodeBits)
                {
                    ng = e;
                }
                else
                {
                    tg = e;
                }
            }
        }
        PTC_PROF(3)
        // ---- one node visit for every lane that has one pending
        if (mode == kTrav && (ng.y & kNodeBits))
        {
            const uint32_t node_idx = next_child(ng, octinv);
            if (kProfile)
            {
                const uint32_t pushes = ((ng.y & kNodeBits) ? 1u : 0u) + (tg.y ? 1u : 0u);
                const uint32_t in_lds = (uint32_t)sp < (uint32_t)kDrainLds ? (uint32_t)kDrainLds - (uint32_t)sp : 0u;
                n_push += pushes;
                n_push_spill += pushes > in_lds ? pushes - in_lds : 0u;
                sp_max = (uint32_t)sp + pushes > sp_max ? (uint32_t)sp + pushes : sp_max;
            }
            if (ng.y & kNodeBits)
            {
                stack_push<kDrainLds>(lds, kDrainBlock, spill, sp, ng);
            }
            if (tg.y)
            {
                stack_push<kDrainLds>(lds, kDrainBlock, spill, sp, tg); // triangles found earlier and not tested yet
            }
            node_test<kCamera>(S, ray_ctx(), best.t, node_idx, ng, tg);
        }
        PTC_PROF(4)
        // ---- one triangle test per lane and iteration, so that lanes with many
        // pending triangles do not hold the wave; two when enough lanes of the wave
        // have a second one pending (leaf slots hold up to three), with both
        // records requested before either is tested: one memory round trip.
        {
            const bool t1 = mode == kTrav && tg.y != 0u;
            const uint32_t i1 = (uint32_t)__builtin_ctz(tg.y | 0x80000000u);
            const uint32_t rest = tg.y & (tg.y - 1u);
            const bool t2 = t1 && rest != 0u &&
                            (uint32_t)__builtin_popcountll(__ballot(t1 && rest != 0u)) >= kTriMoreMin;
            const uint32_t i2 = (uint32_t)__builtin_ctz(rest | 0x80000000u);
            float4 a1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b1 = a1, c1 = a1, a2 = a1, b2 = a1, c2 = a1;
            if (t1)
            {
                const float4 *__restrict__ T = S.tris + 3u * (size_t)(tg.x + i1);
                a1 = T[0];
                b1 = T[1];
                c1 = T[2];
            }
            if (t2)
            {
                const float4 *__restrict__ T = S.tris + 3u * (size_t)(tg.x + i2);
                a2 = T[0];
                b2 = T[1];
                c2 = T[2];
            }
            if (t1)
            {
                drain_tri_eval<kTwoLevel>(a1, b1, c1, ray_ctx(), inst, best);
                tg.y = rest;
            }
            if (t2)
            {
                drain_tri_eval<kTwoLevel>(a2, b2, c2, ray_ctx(), inst, best);
                tg.y = rest & (rest - 1u);
            }
        }
        PTC_PROF(5)
        // While many lanes of the wave still have triangles pending (scenes with
        // few, large leaves are bound by this), they test up to two more right away.
        for (int extra = 0; extra < 2; ++extra)
        {
            const bool more = mode == kTrav && tg.y != 0u;
            if ((uint32_t)__builtin_popcountll(__ballot(more)) < kTriMoreMin)
            {
                break;
            }
            if (more)
            {
                const uint32_t i = (uint32_t)__builtin_ctz(tg.y);
                tg.y &= tg.y - 1u;
                drain_tri_test<kTwoLevel>(S, ray_ctx(), tg.x + i, inst, best);
            }
            if (kProfile)
            {
                prof[7] += 1;
            }
        }
        PTC_PROF(6)
        if (kProfile)
        {
            prof[9] += 1;
        }
    }
#undef PTC_PROF
#undef PTC_WORD
    if (kProfile)
    {
        // 10: most iterations of one wave; 11: most cycles of one wave; 12: most iterations of one ray;
        atomicMax(&g_drain_prof[12], (unsigned long long)ray_iters_max);
        // 16: stack pushes; 17: of them to scratch memory; 18: pops from scratch memory; 19: deepest stack of any lane
        unsigned long long sums[3] = {n_push, n_push_spill, n_pop_spill};
        for (int k = 0; k < 3; ++k)
        {
            for (int off = 32; off > 0; off >>= 1)
            {
                sums[k] += __shfl_xor(sums[k], off, 64);
            }
        }
        atomicMax(&g_drain_prof[19], (unsigned long long)sp_max);
        if (lane == 0u)
        {
            atomicAdd(&g_drain_prof[16], sums[0]);
            atomicAdd(&g_drain_prof[17], sums[1]);
            atomicAdd(&g_drain_prof[18], sums[2]);
        }
        if (lane == 0u)
        {
            for (int k = 0; k < 10; ++k)
            {
                atomicAdd(&g_drain_prof[k], prof[k]);
            }
            const unsigned long long t_end = __builtin_amdgcn_s_memtime();
            atomicMax(&g_drain_prof[10], prof[9]);
            atomicMax(&g_drain_prof[11], t_end - t_begin);
            // 13: cycles all waves lived; 14: waves; 15: cycles they lived after the queue had nothing left for them
            atomicAdd(&g_drain_prof[13], t_end - t_begin);
            atomicAdd(&g_drain_prof[14], 1ull);
            atomicAdd(&g_drain_prof[15], t_end - (t_exhausted != 0ull ? t_exhausted : t_end));
        }
    }
    if (lane == 0u && wave_rays != 0u)
    {
        atomicAdd(&stats->rays, (unsigned long long)wave_rays);
    }
}

// Per pixel: fold the batch's per-path radiances into the film in dispatch
// order (shader.rgen:196, :202-223), then tone-map. A lane owns a pixel (the
// fold is sequential per pixel), but a pixel's radiances lie in a row, so lanes
// reading their own rows would touch 64 lines per load: the block stages
// kResolveTile entries of each of its 256 rows in LDS (8 lanes read the 128
// contiguous bytes of one row) and every lane then takes its row's entries from
// there. Rows are padded by one entry in LDS against bank conflicts.
constexpr uint32_t kResolveTile = 8u; // (4: 64-byte runs, 16: half the resident blocks: the pass takes 2.4 and 2.8 ms against 1.67)
__global__ void __launch_bounds__(kBlock)
k_wf_resolve(DFrame F, const float4 *__restrict__ radiance, uint32_t n_disp, uint32_t spf,
             uint32_t sample_count0, float4 *__restrict__ accum, uint32_t *__restrict__ rgba8)
{
    __shared__ float4 s_tile[kBlock * (kResolveTile + 1u)];
    const uint32_t first_pixel = blockIdx.x * kBlock, i = first_pixel + threadIdx.x;
    const uint32_t n_pixels = F.width * F.height;
    const bool inside = i < n_pixels;
    float4 avg = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t sample_count = sample_count0;
    if (inside && sample_count > 0u)
    {
        avg = accum[i];
    }
    // n_disp consecutive paths of this pixel; every spf of them form one dispatch
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t in_dispatch = 0u;
    for (uint32_t j0 = 0; j0 < n_disp; j0 += kResolveTile)
    {
        const uint32_t n_tile = n_disp - j0 < kResolveTile ? n_disp - j0 : kResolveTile;
        __syncthreads(); // the previous tile has been read
        for (uint32_t r = 0; r < kResolveTile; ++r)
        {
            const uint32_t l = r * kBlock + threadIdx.x, row = l / kResolveTile, e = l % kResolveTile;
            if (first_pixel + row < n_pixels && e < n_tile)
            {
                s_tile[row * (kResolveTile + 1u) + e] = radiance[(size_t)(first_pixel + row) * n_disp + j0 + e];
            }
        }
        __syncthreads();
        if (inside)
        {
            for (uint32_t e = 0; e < n_tile; ++e)
            {
                const float4 r = s_tile[threadIdx.x * (kResolveTile + 1u) + e];
                acc.x += r.x;
                acc.y += r.y;
                acc.z += r.z;
                acc.w += 1.0f;
                if (++in_dispatch == spf)
                {
                    avg = accumulate(avg, acc, sample_count, spf, F.accumulate_sum != 0u);
                    sample_count += spf;
                    acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    in_dispatch = 0u;
                }
            }
        }
    }
    if (!inside)
    {
        return;
    }
    accum[i] = avg;
    vec3 lin = v3(avg.x, avg.y, avg.z);
    if (F.accumulate_sum != 0u)
    {
        lin = lin / avg.w;
    }
    rgba8[i] = pack_rgba8(lin, F.tonemap);
}

// rays traced by the primary pass and by the trace launches of bounces [1, end_b)
__global__ void k_wf_stats(const uint32_t *__restrict__ counts, uint32_t n_paths, uint32_t max_depth,
                           uint32_t end_b, DeviceStats *__restrict__ stats)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        unsigned long long rays = max_depth > 0u ? n_paths : 0u;
        for (uint32_t b = 1; b < end_b; ++b)
        {
            rays += counts[b];
        }
        stats->rays += rays;
        stats->paths += n_paths;
    }
}

size_t align256(size_t x)
{
    return (x + 255) & ~(size_t)255;
}

// Launches the persistent kernel over at most max_items work items: as many
// blocks as the chip holds at the kernel's waves per SIMD.
template <bool kCamera>
void launch_drain(ptc_context *ctx, const ptc_scene *scene, const DFrame &F0, const WfBuffers &B, uint32_t bounce0,
                  int src, uint32_t n_items, uint32_t n_disp, uint32_t k_first, uint32_t max_items)
{
    const bool two_level = scene->d.inst_box != nullptr;
    uint32_t *heads = B.counts + (F0.max_depth + 1u);
    const uint32_t resident = (uint32_t)ctx->n_cus * (4u * (uint32_t)drain_waves(kCamera) * 64u / (uint32_t)kDrainBlock);
    const uint32_t blocks = std::max(1u, std::min(resident, (max_items + kDrainBlock - 1u) / kDrainBlock));
    auto go = [&](auto kernel) {
        kernel<<<blocks, kDrainBlock, 0, ctx->stream>>>(scene->d, F0, B, bounce0, src, n_items, n_disp, k_first, heads,
                                                        ctx->d_stats);
    };
    // PTC_DRAIN_PROFILE=1 (debugging): the variant that times its loop parts, read by ptc_debug_drain_profile()
    static const bool profile = getenv("PTC_DRAIN_PROFILE") != nullptr && atoi(getenv("PTC_DRAIN_PROFILE")) != 0;
    if (profile)
    {
        two_level ? go(k_wf_drain<kCamera, true, true>) : go(k_wf_drain<kCamera, false, true>);
    }
    else
    {
        two_level ? go(k_wf_drain<kCamera, true, false>) : go(k_wf_drain<kCamera, false, false>);
    }
}

// Blocks of the primary pass over nb dispatches of a width x height film, and the
// lanes per pixel (as a power of two) its pixel-major form uses: 64, 32 or 16,
// whichever leaves the fewest lanes without a dispatch (ties go to the wider
// one); below 16 dispatches a wave is an 8x8 pixel tile of one dispatch.
uint64_t primary_grid(uint32_t width, uint32_t height, uint32_t nb, uint32_t *lane_bits_out)
{
    // whole groups of 16x16-pixel tiles
    const uint64_t tiles = (uint64_t)((width + kGroupPixels - 1u) / kGroupPixels) * ((height + kGroupPixels - 1u) / kGroupPixels) *
                           (kGroupTiles * kGroupTiles);
    if (nb < 16u)
    {
        *lane_bits_out = 0u;
        return tiles * nb;
    }
    uint32_t lane_bits = 0, best_waste = ~0u;
    for (uint32_t bits = 6; bits >= 4; --bits)
    {
        const uint32_t groups = (nb + (1u << bits) - 1u) >> bits;
        const uint32_t waste = (groups << bits) - nb;
        if (waste < best_waste)
        {
            best_waste = waste;
            lane_bits = bits;
        }
    }
    *lane_bits_out = lane_bits;
    const uint64_t pixel_blocks = (tiles * 256u) / ((uint32_t)kBlock >> lane_bits);
    const uint32_t k_groups = (nb + (1u << lane_bits) - 1u) >> lane_bits;
    return pixel_blocks * k_groups;
}

// Slots per queue segment for a primary pass of `grid` blocks: every block may
// append all of its kBlock paths, and segment s takes the blocks i with
// i % kQueueSegments == s.
uint64_t segment_slots(uint64_t grid)
{
    return ((grid + kQueueSegments - 1u) / kQueueSegments) * (uint64_t)kBlock;
}

// Words of the counter block: live paths per bounce, the persistent launch's
// queue heads, then (from a multiple of kFillStride on) the segments' fill counts.
size_t fill_offset(uint32_t max_depth)
{
    return (((size_t)max_depth + 1 + kQueueSegments + kFillStride - 1) / kFillStride) * kFillStride;
}
size_t counter_words(uint32_t max_depth)
{
    return fill_offset(max_depth) + (size_t)kQueueSegments * kFillStride;
}

// Slots of the queue the primary pass fills, for batches of max_batch dispatches:
// none with the stream integrator (live paths stay in registers), one per path
// when bounces run as launch pairs (one run of items behind one counter), else
// the segments' capacities (the film rounded up to the primary pass's blocks).
size_t queue_slots(uint32_t width, uint32_t height, uint32_t max_batch, bool stream, bool pairs)
{
    if (stream)
    {
        return 1;
    }
    if (pairs)
    {
        return (size_t)width * height * max_batch;
    }
    uint32_t bits;
    return (size_t)kQueueSegments * segment_slots(primary_grid(width, height, max_batch, &bits));
}

// Workspace bytes of a batch of max_batch dispatches (the carving order of
// launch_render_wavefront below).
// (pairs: bounces run as traversal / shading launch pairs, which take the second
// queue and the hit records; the default, the persistent launch right after the
// first bounce, needs one queue: 76 bytes per path instead of 156, a little more
// where the film is not whole 32x32-pixel groups)
// (regroup: the sorted hand-over takes a key and a place in the order per queue
// slot, 8 bytes, and the bins of the counting sort)
size_t workspace_bytes(uint32_t width, uint32_t height, uint32_t spf, uint32_t max_batch, uint32_t max_depth,
                       bool stream, bool pairs, bool regroup)
{
    const size_t cap = (size_t)width * height * max_batch;
    const size_t qcap = queue_slots(width, height, max_batch, stream, pairs), pcap = pairs && !stream ? cap : 1;
    const size_t rcap = regroup && !stream ? qcap : 0;
    return (align256(qcap * 16) * 3 + align256(qcap * 8)) + (align256(pcap * 16) * 3 + align256(pcap * 8)) +
           align256(pcap * 16) + align256(pcap * 8) + align256(cap * spf * 16) +
           align256(counter_words(max_depth) * 4) + align256(qcap * 4) +
           (rcap ? 2 * align256(rcap * 4) + align256((size_t)kRegroupBins * 4) : 0);
}

} // namespace

int plan_wavefront(const DFrame &F0, bool stream, bool static_scratch, WavefrontPlan *plan)
{
    const uint32_t n_pixels = F0.width * F0.height;
    const uint32_t spf = F0.samples_per_frame;
    const bool pairs = F0.wavefront_bounces > 0u && F0.max_depth > 1u; // some bounce runs as a traversal / shading launch pair
    // dispatches per batch: bounded so the workspace stays bounded (by default 2^29 paths, about 40 GB)
    uint32_t max_batch = std::max<uint32_t>(
        1u, std::min<uint64_t>(F0.n_frames,
                               F0.max_paths_in_flight / std::max<uint64_t>((uint64_t)n_pixels * spf, 1u)));
    // Path ids and work-item numbers are 32-bit: a batch, with the film padded
    // to whole 8x8 tiles (the stream integrator's items), must stay below 2^32.
    {
        const uint64_t padded = (uint64_t)((F0.width + 7u) >> 3) * ((F0.height + 7u) >> 3) * 64u;
        const uint64_t fit = 0xffffffffull / (padded * spf);
        if (fit == 0u)
        {
            set_error("ptc_render: width * height * samples_per_frame exceeds 2^32 paths per dispatch");
            return PTC_E_LIMIT;
        }
        max_batch = (uint32_t)std::min<uint64_t>(max_batch, fit);
    }
    // A call that takes several batches runs them in whole groups of 64 (or 16)
    // dispatches: the primary pass deals the dispatches of a pixel to the 64, 32
    // or 16 lanes of a group, so a batch of 129 would leave 15 of 144 lanes idle
    // and size the queue for them.
    auto whole_groups = [&](uint32_t batch) {
        return batch >= F0.n_frames || batch < 16u ? batch : (batch >= 64u ? batch & ~63u : batch & ~15u);
    };
    max_batch = whole_groups(max_batch);
    // The queue's segments are sized by the blocks of the primary pass, which cover
    // the film rounded up to whole 32x32-pixel groups: for a film much smaller
    // than its groups the batch shrinks so that the queue has at most twice the
    // slots max_paths_in_flight asks for (and fewer than 2^32: slots are 32-bit).
    if (!stream && !pairs)
    {
        const uint64_t bound = std::min<uint64_t>(
            0xffffffffull, 2u * std::max<uint64_t>(F0.max_paths_in_flight, (uint64_t)n_pixels * spf));
        while (max_batch > 1u && queue_slots(F0.width, F0.height, max_batch, stream, pairs) > bound)
        {
            max_batch = whole_groups(max_batch / 2u);
        }
        if (queue_slots(F0.width, F0.height, max_batch, stream, pairs) > 0xffffffffull)
        {
            set_error("ptc_render: the film rounded up to 32x32-pixel groups exceeds 2^32 paths per dispatch");
            return PTC_E_LIMIT;
        }
    }
    auto bytes_for = [&](uint32_t batch) {
        return workspace_bytes(F0.width, F0.height, spf, batch, F0.max_depth, stream, pairs, F0.regroup != 0u);
    };
    if (static_scratch)
    {
        // as many dispatches per batch as the static array holds; one at least
        while (max_batch > 1u && bytes_for(max_batch) > kStaticScratchBytes)
        {
            max_batch = whole_groups(max_batch / 2u);
        }
        const size_t need = bytes_for(max_batch);
        if (need > kStaticScratchBytes)
        {
            set_error("ptc_render_static: one dispatch of " + std::to_string(n_pixels) + " pixels x " +
                      std::to_string(spf) + " samples needs " + std::to_string(need) +
                      " bytes of workspace, the static array holds " + std::to_string(kStaticScratchBytes) +
                      " (use ptc_render, a smaller film or fewer samples per frame)");
            return PTC_E_LIMIT;
        }
    }
    plan->dispatches_per_batch = max_batch;
    plan->bytes = bytes_for(max_batch);
    plan->pairs = pairs;
    return PTC_OK;
}

int wavefront_static_init(ptc_context *ctx)
{
    // `scratch` is a host-side stand-in for the device array, not an address a
    // kernel can use: the device address on the current device comes from here
    // and is what every later use (kernel arguments, memsets) takes.
    void *ptr = nullptr;
    PTC_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (ptr == nullptr)
    {
        set_error("hipGetSymbolAddress returned no address for the static workspace");
        return PTC_E_DEVICE;
    }
    ctx->static_scratch = ptr;
    ctx->static_scratch_bytes = kStaticScratchBytes;
    return PTC_OK;
}

// Renders F.n_frames dispatches of F.samples_per_frame samples with ray queues.
// Dispatches are independent (each re-seeds its RNG), so a batch of them is in
// flight at once. The samples of one dispatch share an RNG stream, so they run
// as successive passes, each pass starting from the state the previous sample
// of the same pixel ended with.
int launch_render_wavefront(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const DFrame &F0,
                            bool stream, bool static_scratch, bool count)
{
    hipStream_t st = ctx->stream;
    const uint32_t n_pixels = F0.width * F0.height;
    const uint32_t spf = F0.samples_per_frame;
    const bool two_level = scene->d.inst_box != nullptr;
    const uint32_t wf_bounces = F0.wavefront_bounces;
    WavefrontPlan plan;
    if (const int rc = plan_wavefront(F0, stream, static_scratch, &plan); rc != PTC_OK)
    {
        return rc;
    }
    const bool pairs = plan.pairs;
    const uint32_t max_batch = plan.dispatches_per_batch;
    const size_t n_counts = counter_words(F0.max_depth);
    char *workspace = nullptr;
    if (static_scratch)
    {
        if (ctx->static_scratch == nullptr || ctx->static_scratch_bytes < plan.bytes)
        {
            set_error("ptc_render_static: the static workspace was not set up on this device");
            return PTC_E_DEVICE;
        }
        workspace = static_cast<char *>(ctx->static_scratch);
    }
    const size_t cap = (size_t)n_pixels * max_batch; // live paths of one pass
    // stream mode keeps live paths in registers: only the radiance array is needed
    const size_t qcap = queue_slots(F0.width, F0.height, max_batch, stream, pairs);
    const size_t pcap = pairs && !stream ? cap : 1;
    const size_t bytes = plan.bytes;
    if (!static_scratch)
    {
        if (bytes > ctx->wf_bytes)
        {
            PTC_HIP_CHECK(hipStreamSynchronize(st));
            (void)hipFree(ctx->wf_buf);
            ctx->wf_buf = nullptr;
            ctx->wf_bytes = 0;
            if (hipMalloc(&ctx->wf_buf, bytes) != hipSuccess)
            {
                (void)hipGetLastError();
                ctx->wf_buf = nullptr;
                set_error("ptc_render: out of device memory for " + std::to_string(bytes) +
                          " bytes of workspace (lower max_paths_in_flight)");
                return PTC_E_NOMEM; // nothing was launched
            }
            ctx->wf_bytes = bytes;
        }
        workspace = static_cast<char *>(ctx->wf_buf);
    }
    char *p = workspace;
    auto take = [&](size_t n) {
        char *r = p;
        p += align256(n);
        return r;
    };
    WfBuffers B;
    // queue 1 is the one the primary pass fills; queue 0 and the hit records serve the launch pairs only
    for (int k = 1; k >= 0; --k)
    {
        const size_t c = k == 1 ? qcap : pcap;
        B.q[k].a = reinterpret_cast<float4 *>(take(c * 16));
        B.q[k].b = reinterpret_cast<float4 *>(take(c * 16));
        B.q[k].c = reinterpret_cast<float4 *>(take(c * 16));
        B.q[k].d = reinterpret_cast<float2 *>(take(c * 8));
    }
    B.hit = reinterpret_cast<float4 *>(take(pcap * 16));
    B.hit_ids = reinterpret_cast<uint2 *>(take(pcap * 8));
    B.radiance = reinterpret_cast<float4 *>(take(cap * spf * 16));
    B.counts = reinterpret_cast<uint32_t *>(take(n_counts * 4));
    B.fill = B.counts + fill_offset(F0.max_depth);
    B.seg_cap = 0u;
    uint32_t *carry = reinterpret_cast<uint32_t *>(take(qcap * 4));
    B.carry = spf > 1u && !stream ? carry : nullptr;
    B.spf = spf;
    B.n_pixels = n_pixels;
    B.per_pixel = spf; // (set per batch below)
    B.order = nullptr;
    const bool regroup = F0.regroup != 0u && !stream;
    uint32_t *rg_keys = nullptr, *rg_order = nullptr, *rg_bins = nullptr;
    if (regroup)
    {
        rg_keys = reinterpret_cast<uint32_t *>(take(qcap * 4));
        rg_order = reinterpret_cast<uint32_t *>(take(qcap * 4));
        rg_bins = reinterpret_cast<uint32_t *>(take((size_t)kRegroupBins * 4));
    }
    // Sorts the filled slots of queue `which` (its first n_slots slots; segmented
    // when seg_cap != 0, else one run of *count paths) into rg_order and leaves
    // their number in *total_out.
    auto regroup_queue = [&](int which, size_t n_slots, uint32_t seg_cap, const uint32_t *count, uint32_t *total_out) -> int {
        PTC_HIP_CHECK(hipMemsetAsync(rg_bins, 0, (size_t)kRegroupBins * 4, st));
        const uint32_t blocks = (uint32_t)std::min<size_t>((n_slots + kBlock - 1) / kBlock, (size_t)ctx->n_cus * 32u);
        k_regroup_keys<<<std::max(blocks, 1u), kBlock, 0, st>>>(scene->d, B.q[which], (uint32_t)n_slots, seg_cap, B.fill, count,
                                                               rg_keys, rg_bins);
        k_regroup_scan<<<1, 1024, 0, st>>>(rg_bins, total_out);
        k_regroup_scatter<<<std::max(blocks, 1u), kBlock, 0, st>>>((uint32_t)n_slots, seg_cap, B.fill, count, rg_keys, rg_bins,
                                                                  rg_order);
        return PTC_OK;
    };

    const uint32_t persistent_blocks = (uint32_t)ctx->n_cus * 6u;
    uint32_t sample_count = F0.sample_count;
    for (uint32_t k0 = 0; k0 < F0.n_frames; k0 += max_batch)
    {
        const uint32_t nb = std::min(max_batch, F0.n_frames - k0);
        const uint32_t n_paths = n_pixels * nb;
        B.per_pixel = nb * spf;
        if (stream)
        {
            // one persistent launch: camera rays are generated as lanes fall idle
            const uint32_t tiles8 = ((F0.width + 7u) >> 3) * ((F0.height + 7u) >> 3);
            const uint32_t n_items = tiles8 * 64u * nb;
            PTC_HIP_CHECK(hipMemsetAsync(B.counts, 0, n_counts * 4, st));
            {
                PhaseTimer timer(ctx, PTC_PHASE_DRAIN);
                launch_drain<true>(ctx, scene, F0, B, 0u, 0, n_items, nb, k0, n_items);
            }
            k_wf_stats<<<1, 64, 0, st>>>(B.counts, n_paths * spf, 0u, 1u, ctx->d_stats);
        }
        for (uint32_t s = 0; s < (stream ? 0u : spf); ++s)
        {
            PTC_HIP_CHECK(hipMemsetAsync(B.counts, 0, n_counts * 4, st));
            {
                PhaseTimer timer(ctx, PTC_PHASE_PRIMARY);
                uint32_t lane_bits = 0;
                const uint32_t grid = (uint32_t)primary_grid(F0.width, F0.height, nb, &lane_bits);
                B.seg_cap = pairs ? 0u : (uint32_t)segment_slots(grid);
                // kernel variant by (pixel-major mapping, two-level scene, counting)
                auto launch = [&](auto kernel) {
                    kernel<<<grid, kBlock, 0, st>>>(scene->d, F0, B, F0.global_frame_count, k0, s, nb, lane_bits, ctx->d_stats);
                };
                if (nb >= 16u)
                {
                    if (count)
                    {
                        two_level ? launch(k_wf_primary<1, true, true>) : launch(k_wf_primary<1, false, true>);
                    }
                    else if (lane_bits == 6u)
                    {
                        two_level ? launch(k_wf_primary<2, true, false>) : launch(k_wf_primary<2, false, false>);
                    }
                    else
                    {
                        two_level ? launch(k_wf_primary<1, true, false>) : launch(k_wf_primary<1, false, false>);
                    }
                }
                else if (count)
                {
                    two_level ? launch(k_wf_primary<0, true, true>) : launch(k_wf_primary<0, false, true>);
                }
                else
                {
                    two_level ? launch(k_wf_primary<0, true, false>) : launch(k_wf_primary<0, false, false>);
                }
            }
            const uint32_t blocks = std::min<uint32_t>(persistent_blocks, (n_paths + kBlock - 1) / kBlock);
            // bounces [1, drain_b) as trace / shade launch pairs over compacted queues
            const uint32_t drain_b = std::min(F0.max_depth, 1u + wf_bounces);
            for (uint32_t b = 1; b < drain_b; ++b)
            {
                PhaseTimer timer(ctx, PTC_PHASE_BOUNCE);
                const int src = (int)(b & 1u); // the primary pass writes set 1
                if (regroup)
                {
                    if (const int rc = regroup_queue(src, n_paths, 0u, B.counts + b, B.counts + b); rc != PTC_OK)
                    {
                        return rc;
                    }
                }
                k_wf_trace<<<blocks, kBlock, 0, st>>>(scene->d, F0.tmax, B.q[src], B.hit, B.hit_ids, B.counts + b,
                                                      regroup ? rg_order : nullptr);
                k_wf_shade<<<std::max(1u, blocks / (kShadeBlock / kBlock)), kShadeBlock, 0, st>>>(scene->d, B, b, F0.max_depth, src, F0.debug_view);
            }
            // whatever is still alive runs to its end in one persistent launch
            if (drain_b < F0.max_depth)
            {
                PhaseTimer timer(ctx, PTC_PHASE_DRAIN);
                WfBuffers D = B;
                if (regroup)
                {
                    // (timed with the launch it serves) the sorted order turns the queue into one run of
                    // counts[drain_b] work items, whatever shape it was filled in
                    const size_t n_slots = B.seg_cap != 0u ? (size_t)kQueueSegments * B.seg_cap : n_paths;
                    if (const int rc = regroup_queue((int)(drain_b & 1u), n_slots, B.seg_cap, B.counts + drain_b,
                                                     B.counts + drain_b);
                        rc != PTC_OK)
                    {
                        return rc;
                    }
                    D.seg_cap = 0u;
                    D.order = rg_order;
                }
                launch_drain<false>(ctx, scene, F0, D, drain_b, (int)(drain_b & 1u), 0u, 0u, 0u, n_paths);
            }
            k_wf_stats<<<1, 64, 0, st>>>(B.counts, n_paths, F0.max_depth, drain_b, ctx->d_stats);
        }
        {
            PhaseTimer timer(ctx, PTC_PHASE_RESOLVE);
            k_wf_resolve<<<(n_pixels + kBlock - 1) / kBlock, kBlock, 0, st>>>(F0, B.radiance, nb * spf, spf, sample_count,
                                                                             film->d_accum, film->d_rgba8);
        }
        sample_count += nb * spf;
    }
    PTC_HIP_CHECK(hipGetLastError());
    return PTC_OK;
}

} // namespace ptc

// Debugging aid (not part of the public header): reads and clears the cycle
// counters of the profiled persistent kernel (tools/drain_prof.py).
extern "C" int ptc_debug_drain_profile(unsigned long long *out20)
{
    if (hipMemcpyFromSymbol(out20, HIP_SYMBOL(g_drain_prof), sizeof(unsigned long long) * 20) != hipSuccess)
    {
        return -1;
    }
    unsigned long long zero[20] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_drain_prof), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
