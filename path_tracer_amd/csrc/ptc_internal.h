// Host-side objects behind the opaque handles of include/ptc.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ptc.h"
#include "ptc_device.h"

namespace ptc
{

void set_error(const std::string &msg);

#define PTC_HIP_CHECK(expr)                                                              \
    do                                                                                   \
    {                                                                                    \
        const hipError_t err_ = (expr);                                                  \
        if (err_ != hipSuccess)                                                          \
        {                                                                                \
            ::ptc::set_error(std::string(#expr) + ": " + hipGetErrorString(err_));       \
            return PTC_E_DEVICE;                                                         \
        }                                                                                \
    } while (0)

struct DeviceStats
{
    unsigned long long rays;
    unsigned long long paths;
    unsigned long long nodes;
    unsigned long long tris;
    unsigned long long uniform_nodes;
    unsigned long long uniform_fetches;
};

} // namespace ptc

struct ptc_film;

struct ptc_context
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    ptc::DeviceStats *d_stats = nullptr; // device counters
    hipEvent_t ev_start = nullptr, ev_stop = nullptr; // most recent ptc_render
    double last_render_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; // one pair per recorded render
    size_t ev_used = 0;
    // event pairs around the kernel phases of the renders, tagged PTC_PHASE_*
    struct PhaseEvent
    {
        hipEvent_t start, stop;
        uint32_t tag;
    };
    std::vector<PhaseEvent> phase_pool;
    size_t phase_used = 0;
    int n_cus = 256;
    // wavefront integrator workspace, grown on demand
    void *wf_buf = nullptr;
    size_t wf_bytes = 0;
    // the library's static device array on this context's device (ptc_render_static)
    void *static_scratch = nullptr;
    size_t static_scratch_bytes = 0;
    // staging memory of the query calls that take host pointers (ptc_primary_hits,
    // ptc_trace_rays, ptc_math_eval): kept between calls, grown when a call needs more
    void *stage_buf = nullptr;
    size_t stage_bytes = 0;
    bool no_events = false; // set for the duration of a call that must not create events
    std::vector<ptc_film *> films; // films created on this context (they may hold recorded dispatches)
};

struct ptc_scene
{
    ptc_context *ctx = nullptr;
    ptc::DScene d{};          // device pointers, passed to kernels by value
    void *d_nodes = nullptr;  // n_nodes * 80 bytes
    void *d_tris = nullptr;   // n_tris * 48 bytes
    float *d_positions = nullptr;
    float *d_normals = nullptr;
    uint32_t *d_indices = nullptr;
    float4 *d_shade = nullptr; // 5 x float4 per mesh triangle: its corners' positions and normals
    ptc::DInstance *d_instances = nullptr;
    float4 *d_inst_box = nullptr; // two-level scenes: (lo, hi) world box per instance
    float4 *d_env = nullptr;
    std::vector<uint32_t> first_flat; // per instance: flattened index of its primitive 0 (ascending)
    ptc_scene_info info{};
};

struct ptc_film
{
    ptc_context *ctx = nullptr;
    uint32_t width = 0, height = 0;
    uint32_t tonemap = 0; // PTC_TONEMAP_* of the last render call
    float4 *d_accum = nullptr;
    uint32_t *d_rgba8 = nullptr;
    bool own_accum = false, own_rgba8 = false;
    uint32_t sample_count = 0;
    // Dispatches recorded by ptc_render_deferred() and not launched yet
    // (params.n_frames of them; 0 = none): consecutive calls with one scene, one
    // camera and one parameter set collapse into one ptc_render().
    struct Pending
    {
        const ptc_scene *scene = nullptr;
        ptc_camera camera{};
        ptc_render_params params{};
    } pending;
    // ptc_film_reset() came while dispatches were recorded. The reference's
    // reset_render only zeroes the counter (src/renderer.cpp:2236-2239) and every
    // earlier draw_frame has been submitted, so the recorded run stays launchable:
    // whatever reads the film before the next dispatch launches it first (the film
    // then holds every submitted frame), and the next dispatch, which overwrites
    // the film, drops it. sample_count keeps the run's base count until then.
    bool reset_pending = false;
};

namespace ptc
{

// Brackets one kernel phase of a render with events (at most 1024 are kept
// between two ptc_phase_times() calls; later ones go untimed).
struct PhaseTimer
{
    ptc_context *ctx;
    hipEvent_t stop = nullptr;
    PhaseTimer(ptc_context *c, uint32_t tag) : ctx(c)
    {
        if (ctx->no_events)
        {
            return;
        }
        if (ctx->phase_used == ctx->phase_pool.size() && ctx->phase_pool.size() < 1024)
        {
            ptc_context::PhaseEvent e{};
            if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess)
            {
                return;
            }
            ctx->phase_pool.push_back(e);
        }
        if (ctx->phase_used < ctx->phase_pool.size())
        {
            ptc_context::PhaseEvent &e = ctx->phase_pool[ctx->phase_used++];
            e.tag = tag;
            (void)hipEventRecord(e.start, ctx->stream);
            stop = e.stop;
        }
    }
    ~PhaseTimer()
    {
        if (stop)
        {
            (void)hipEventRecord(stop, ctx->stream);
        }
    }
};

// ptc_render_params.max_paths_in_flight = 0: a whole step of 256 dispatches of a 1920x1080 film (40 GB of
// workspace at 76 bytes per path, grown on demand; the part has 288 GB)
constexpr uint32_t kDefaultPathsInFlight = 1u << 29;

// What ptc_build_options says about the tree (0 = default everywhere).
struct BuildParams
{
    uint32_t search_radius = 0;
    uint32_t collapse = 0; // 0: cost-optimal choice of the wide nodes; 1: greedy by area
    float tri_cost = 0.0f; // cost of a triangle test relative to a node visit
    // rounds of subtree reinsertion run on the binary tree before it is cut into
    // wide nodes (0 = default; kNoOptimize = none)
    uint32_t optimize_rounds = 0;
};
constexpr uint32_t kNoOptimize = 0xffffffffu;

// BVH build on the device (ptc_bvh_build.hip). Fills scene->d_nodes, d_tris,
// d.nodes, d.tris and the BVH fields of scene->info.
int build_bvh(ptc_context *ctx, ptc_scene *scene, const std::vector<DInstance> &instances,
              uint32_t n_flat_tris, const BuildParams &params);
// Writes scene->d_shade from d_positions, d_normals and d_indices (all uploaded
// already): per mesh triangle the object-space positions and normals of its three
// corners, so that shading a hit takes one gather instead of the indices first and
// the six vertex attributes after them.
int build_shade_records(ptc_context *ctx, ptc_scene *scene, const std::vector<uint32_t> &vtx_off,
                        const std::vector<uint32_t> &tri_off);
// Two-level variant: one tree per mesh in object space (concatenated into
// d_nodes / d_tris), instances[i].blas_root and the world boxes in d_inst_box.
// mesh_box holds (lo.xyz, hi.xyz) of every mesh in object space.
int build_two_level(ptc_context *ctx, ptc_scene *scene, std::vector<DInstance> &instances,
                    const std::vector<uint32_t> &vtx_off, const std::vector<uint32_t> &tri_off,
                    const std::vector<float> &mesh_box, const std::vector<uint32_t> &mesh_of,
                    const BuildParams &params);

// Render kernels (ptc_render.hip, ptc_wavefront.hip).
int launch_render_mega(ptc_context *ctx, const ptc_scene *scene, ptc_film *film, const DFrame &frame,
                       bool count);
// static_scratch: the workspace is the library's static device array instead of
// ctx->wf_buf; nothing is allocated, freed or synchronised, and a call whose
// single dispatch does not fit returns PTC_E_LIMIT.
// count: the fused first-bounce kernel also counts nodes / triangles visited.
int launch_render_wavefront(ptc_context *ctx, const ptc_scene *scene, ptc_film *film,
                            const DFrame &frame, bool stream, bool static_scratch, bool count);
// How launch_render_wavefront would run a call (host arithmetic only): the
// dispatches it holds in flight per batch and the workspace they take; with
// static_scratch, as bounded by the static array. PTC_E_LIMIT (with the error
// text set) for a call that cannot run.
struct WavefrontPlan
{
    uint32_t dispatches_per_batch = 0;
    size_t bytes = 0;
    bool pairs = false; // bounces run as traversal / shading launch pairs
};
int plan_wavefront(const DFrame &frame, bool stream, bool static_scratch, WavefrontPlan *plan);
// Looks up the device address of the static array on ctx's (current) device.
int wavefront_static_init(ptc_context *ctx);
int launch_primary_hits(ptc_context *ctx, const ptc_scene *scene, const DFrame &frame,
                        uint32_t *d_ids);
int launch_trace(ptc_context *ctx, const ptc_scene *scene, const float *d_rays, size_t n,
                 float tmax, ptc_hit *d_hits, bool count);
int launch_resolve(ptc_context *ctx, ptc_film *film, int sum_mode);
int launch_math_eval(ptc_context *ctx, int fn, const float *d_a, const float *d_b, float *d_out,
                     float *d_out2, size_t n);

} // namespace ptc
