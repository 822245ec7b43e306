/*
 * ptc.h -- C ABI of libptcore, the MI355X (gfx950) path-tracing core.
 *
 * The library replaces one hot path of mschertenleib/path_tracer: the Vulkan
 * KHR ray-tracing pipeline behind the renderer interface of
 * src/renderer.hpp:256-278 (create_render_resources / draw_frame /
 * reset_render / write_to_png) together with the device programs in
 * src/shaders. Each entry point below names the reference interface it stands
 * in for. Plain pointers and sizes only; no C++ or framework types cross this
 * boundary. All functions return PTC_OK (0) or a negative PTC_E_* code; the
 * message for the calling thread's last failure is ptc_last_error().
 *
 * Threading: a context owns one HIP device and one stream. Calls on one
 * context are ordered on that stream and must not be issued concurrently
 * from several host threads. The read-back calls synchronise the stream.
 */
#ifndef PTC_H
#define PTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTC_OK 0
#define PTC_E_INVALID (-1)   /* bad argument */
#define PTC_E_DEVICE (-2)    /* HIP runtime failure */
#define PTC_E_NOMEM (-3)
#define PTC_E_IO (-4)
#define PTC_E_LIMIT (-5)     /* size beyond a documented maximum */

/* Material kinds: the reference's shader-binding-table record offset
 * (src/renderer.cpp:1170, hit groups in the order of :1548-1567). */
#define PTC_KIND_DIFFUSE 0u    /* src/shaders/diffuse.rchit */
#define PTC_KIND_SPECULAR 1u   /* src/shaders/specular.rchit */
#define PTC_KIND_EMISSIVE 2u   /* src/shaders/emissive.rchit */
#define PTC_KIND_DIELECTRIC 3u /* src/shaders/dielectric.rchit */

#define PTC_NO_HIT 0xffffffffu

typedef struct ptc_context ptc_context;
typedef struct ptc_scene ptc_scene;
typedef struct ptc_film ptc_film;

typedef struct ptc_vec3
{
    float x, y, z;
} ptc_vec3; /* src/vec3.hpp:6-16 */

/* Byte-identical to the reference's Camera (src/camera.hpp:6-19), so the
 * output of its create_camera() can be passed straight through. */
typedef struct ptc_camera
{
    float distance;
    ptc_vec3 position;
    ptc_vec3 target;
    ptc_vec3 direction_x; /* right */
    ptc_vec3 direction_y; /* down */
    ptc_vec3 direction_z; /* forward */
    float sensor_distance;
    float sensor_half_width;
    float sensor_half_height;
    float focus_distance;
    float aperture_radius;
} ptc_camera;

/* One triangle mesh: the three arrays create_render_resources() uploads from
 * aiScene::mMeshes[i] (src/renderer.cpp:1818-1838). Host pointers; copied.
 * Coordinates must be finite and within +-1e9 (the reference's importer
 * normalises scenes to [-1, 1], src/application.cpp:161); ptc_scene_create
 * returns PTC_E_INVALID otherwise, and likewise for instance transforms. */
typedef struct ptc_mesh_desc
{
    const float *positions;  /* 3 * n_vertices  (aiMesh::mVertices) */
    const float *normals;    /* 3 * n_vertices  (aiMesh::mNormals) */
    const uint32_t *indices; /* 3 * n_triangles (flattened aiMesh::mFaces) */
    uint32_t n_vertices;
    uint32_t n_triangles;
} ptc_mesh_desc;

/* One TLAS instance (src/renderer.cpp:1147-1174): 3x4 row-major
 * object-to-world transform (VkTransformMatrixKHR layout), the mesh it
 * instantiates and its material kind. */
typedef struct ptc_instance_desc
{
    float transform[12];
    uint32_t mesh;
    uint32_t kind;
} ptc_instance_desc;

/* Equirectangular RGBA32F environment map (src/renderer.cpp:1840-1938,
 * sampled by src/shaders/shader.rmiss:20-24). rgba == NULL selects the
 * two-tone sky of shader.rmiss:13-18. */
typedef struct ptc_envmap_desc
{
    const float *rgba; /* 4 * width * height, row-major, row 0 = v 0 */
    uint32_t width;
    uint32_t height;
} ptc_envmap_desc;

typedef struct ptc_build_options
{
    /* Neighbourhood (in Morton order) each cluster searches for its merge
     * partner during the bottom-up build; 0 selects the default of 8,
     * values above 32 are clamped. Larger = slower build and a slightly lower
     * surface-area cost, which traversal times follow only loosely. */
    uint32_t search_radius;
    /* 0: every instance's triangles are transformed to world space and go
     * into one tree (fastest traversal; memory grows with the instance count).
     * 1: one tree per mesh in object space plus a tree over the world-space
     * boxes of the instances, the structure of the reference's TLAS over one
     * BLAS (src/renderer.cpp:1044-1281): memory does not grow with the
     * triangle count of repeated meshes; a ray enters an instance through its
     * world-to-object matrix, so hits are computed in object space (they agree
     * with mode 0 up to the rounding of that transform). */
    uint32_t two_level;
    /* How the binary tree is cut into 8-wide nodes. 0: the cut that minimises
     * the surface-area cost of the wide tree (which subtrees become wide nodes,
     * and which eight subtrees their children, by dynamic programming over the
     * binary tree). 1: greedy, open the child of largest area until eight
     * slots are used (faster build by a few percent, about a third more nodes
     * visited per ray). */
    uint32_t collapse;
    /* Cost of one triangle test in units of one wide-node visit, for collapse 0;
     * 0 selects the default of 0.8 (a node visit fetches 80 bytes and tests
     * eight boxes, a triangle test fetches 48 bytes and costs about a third of
     * the arithmetic; measured render rates are flat from 0.5 to 1.1 and half
     * a percent to one percent lower at 0.45 and at 1.5). */
    float tri_cost;
    /* Rounds of subtree reinsertion run on the binary tree before it is cut
     * into wide nodes: every round, every subtree looks for the place in the
     * tree where it would add the least surface area, and the moves that do
     * not touch each other are carried out. 0 selects the default of 8 rounds
     * (871k triangles: 8 ms on top of a 3.6 ms build, 7 % fewer node visits
     * per incoherent ray); PTC_OPTIMIZE_NONE skips the step; values above 256
     * are clamped; a round that moves nothing ends it early. The driver-side
     * build this stands in for is asked to prefer trace speed over build time
     * (src/renderer.cpp:1080, :1212). */
    uint32_t optimize_rounds;
    uint32_t reserved[3];
} ptc_build_options;
#define PTC_OPTIMIZE_NONE 0xffffffffu

typedef struct ptc_scene_info
{
    uint32_t n_triangles;  /* flattened over instances */
    uint32_t n_nodes;      /* 80-byte wide nodes */
    uint32_t n_instances;
    uint32_t bvh_depth;    /* levels of the wide tree */
    float build_ms;        /* device time of the BVH build */
    float sah_cost;        /* SAH cost of the wide tree (Ct = Cn = 1) */
    uint64_t device_bytes; /* HBM held by the scene */
    /* 48-byte triangle records stored beside the nodes: n_triangles for a
     * flattened scene; for a two_level scene the sum over the meshes (a mesh
     * that several instances share is stored once, one no instance uses is
     * stored all the same). This is the count ptc_scene_read_bvh() copies. */
    uint32_t n_bvh_triangles;
    uint32_t reserved;
} ptc_scene_info;

/* 80-byte wide-BVH node and 48-byte triangle as they lie in HBM; returned by
 * ptc_scene_read_bvh() so a checker can validate and traverse the tree. */
typedef struct ptc_bvh8_node
{
    float origin[3];      /* lower corner of the node's quantisation grid */
    uint8_t exp[3];       /* per-axis grid step 2^(exp-127), as exponent bits */
    uint8_t imask;        /* bit s: slot s holds an internal node */
    uint32_t child_base;  /* first internal child; the rest follow in slot order */
    uint32_t tri_base;    /* first triangle of this node's leaf slots */
    uint8_t meta[8];      /* 0 empty; internal 0x20|(24+s); leaf (unary n<<5)|offset */
    uint8_t qlo[3][8];    /* qlo[axis][slot] */
    uint8_t qhi[3][8];
} ptc_bvh8_node;

typedef struct ptc_bvh_triangle
{
    float v0[3];
    uint32_t instance;
    float e1[3];
    uint32_t primitive;
    float e2[3];
    uint32_t flat; /* index in instance-major order; the tie-break key */
} ptc_bvh_triangle;

typedef struct ptc_hit
{
    float t, u, v;
    uint32_t flat; /* PTC_NO_HIT on a miss */
} ptc_hit;

#define PTC_INTEGRATOR_MEGAKERNEL 0u /* one lane walks one pixel's paths */
#define PTC_INTEGRATOR_WAVEFRONT 1u  /* ray queues, compaction between bounces */
#define PTC_INTEGRATOR_STREAM 2u     /* one persistent launch; idle lanes start new camera paths */
/* The faster of the two above for the call's shape (all integrators give the
 * same film): the wavefront integrator runs the samples_per_frame samples of a
 * dispatch as successive passes, so it needs many dispatches in flight (16 or
 * more, or one sample per dispatch); the stream integrator has no pass
 * boundaries and wins for the reference's own pattern, one dispatch per call
 * with several samples (1920x1080, 871k triangles, 64 samples: 3.0 against
 * 1.2 Grays/s). */
#define PTC_INTEGRATOR_AUTO 3u

#define PTC_ACCUM_RUNNING_MEAN 0u /* reference semantics (shader.rgen:202-216) */
#define PTC_ACCUM_SUM 1u          /* un-normalised sum, for sharding samples */

/* One or more dispatches of the ray-generation program. Dispatch k of
 * n_frames uses global_frame_count + k and continues the film's sample
 * count, exactly as n_frames consecutive draw_frame() calls would
 * (src/renderer.cpp:2064-2110, :2224). */
typedef struct ptc_render_params
{
    uint32_t global_frame_count; /* Push_constants::global_frame_count */
    uint32_t samples_per_frame;  /* Push_constants::samples_per_frame */
    uint32_t n_frames;           /* consecutive dispatches in this call (>= 1) */
    uint32_t frame_stride;       /* frame count step between dispatches (0 -> 1) */
    uint32_t max_depth;          /* the literal 32 of shader.rgen:65 (0 -> 32) */
    float tmax;                  /* the literal 10000.0 of shader.rgen:77 (0 -> 10000) */
    uint32_t integrator;         /* PTC_INTEGRATOR_* */
    uint32_t accumulate;         /* PTC_ACCUM_* */
    /* 1: also count BVH nodes / triangles visited. Megakernel: all bounces.
     * Wavefront: the fused first bounce only (the persistent launch that
     * follows counts rays, nothing else). Stream: ignored. Counting runs use
     * the reproducible visit order, so they are slower than plain ones. */
    uint32_t instrument;
    /* Wavefront integrator: bounces after the first that run as separate
     * traversal / shading launches over compacted queues before one persistent
     * launch finishes the remaining paths. 0 (default, fastest measured): the
     * persistent launch starts right after the first bounce. Does not change
     * results. */
    uint32_t wavefront_bounces;
    /* Tone map applied when the RGBA8 target is written: PBR neutral is what the
     * reference compiles in (shader.rgen:218-219), ACES is its alternative under
     * "#else" (shader.rgen:107-135, :220-221). Remembered by the film for
     * ptc_film_resolve(). */
    uint32_t tonemap;
    /* Wavefront / stream integrators: upper bound on the paths in flight, which
     * sizes the per-context workspace (76 bytes per path with the wavefront
     * integrator, 60 of them per pixel of the film rounded up to whole
     * 32x32-pixel groups; 156 when wavefront_bounces > 0, 16 with the stream
     * one). A call
     * with more paths runs in several batches of whole dispatches. 0 = 2^29
     * (about 40 GB of the part's 288, allocated when a call needs them: 256
     * dispatches of a 1920x1080 film); at least one dispatch is always in
     * flight. Does not change results. */
    uint32_t max_paths_in_flight;
    /* 1: the debug view the reference keeps under "#else" at shader.rgen:195-199:
     * every sample adds bounces / 16 (the path loop's counter when the path
     * ended, shader.rgen:65: the iteration it ended in, or max_depth) to its
     * pixel instead of its radiance. 0: radiance. */
    uint32_t debug_view;
    /* Wavefront integrator: 1 hands the paths that survive the first bounce to
     * the persistent launch sorted by the cell of their ray's origin (Morton
     * order in the scene's box) and its direction octant, instead of in the
     * order the first bounce appended them (pixel order). Does not change
     * results. 0 (default): queue order. */
    uint32_t regroup;
} ptc_render_params;
#define PTC_TONEMAP_PBR_NEUTRAL 0u
#define PTC_TONEMAP_ACES 1u

typedef struct ptc_stats
{
    uint64_t rays;   /* closest-hit queries since the last reset */
    uint64_t paths;  /* camera samples */
    uint64_t nodes_visited; /* counted only by instrumented calls */
    uint64_t tris_tested;
    double render_ms; /* device time of the last ptc_render */
    /* Of nodes_visited: visits whose 80-byte node was fetched once for all lanes
     * of the wave that were at that node together (through the scalar cache),
     * and the number of such shared fetches. Bytes actually requested for nodes
     * = 80 * (nodes_visited - nodes_wave_uniform + uniform_fetches). */
    uint64_t nodes_wave_uniform;
    uint64_t uniform_fetches;
} ptc_stats;

/* ---- context: stands in for create_context() (src/renderer.hpp:256) ------ */
int ptc_context_create(int device_ordinal, ptc_context **out);
void ptc_context_destroy(ptc_context *ctx);
/* Use an existing HIP stream (hipStream_t) for all work of this context. */
int ptc_context_set_stream(ptc_context *ctx, void *hip_stream);
const char *ptc_last_error(void);

/* ---- scene: the geometry upload + BLAS/TLAS build + environment-map upload
 * of create_render_resources() (src/renderer.cpp:1818-1941). Builds the wide
 * BVH on the device. opts may be NULL. */
int ptc_scene_create(ptc_context *ctx, const ptc_mesh_desc *meshes,
                     uint32_t n_meshes, const ptc_instance_desc *instances,
                     uint32_t n_instances, const ptc_envmap_desc *envmap,
                     const ptc_build_options *opts, ptc_scene **out);
void ptc_scene_destroy(ptc_scene *scene);
int ptc_scene_get_info(const ptc_scene *scene, ptc_scene_info *out);
/* Copy the device BVH to host buffers: n_nodes and n_triangles must equal
 * ptc_scene_info::n_nodes and ::n_bvh_triangles. In a two_level scene the
 * arrays hold one tree per mesh, back to back, in object space; a triangle
 * record's instance word is then 0 and its flat word the primitive index. */
int ptc_scene_read_bvh(const ptc_scene *scene, ptc_bvh8_node *nodes,
                       size_t n_nodes, ptc_bvh_triangle *triangles,
                       size_t n_triangles);

/* ---- film: the RGBA32F storage image + RGBA8 render target + sample
 * counters of Vulkan_render_resources (src/renderer.hpp:225-228, :251-253).
 * accum_dev / rgba8_dev may point at caller-owned device memory
 * (16 * w * h and 4 * w * h bytes) or be NULL to let the library allocate. */
int ptc_film_create(ptc_context *ctx, uint32_t width, uint32_t height,
                    void *accum_dev, void *rgba8_dev, ptc_film **out);
void ptc_film_destroy(ptc_film *film);
/* reset_render() (src/renderer.cpp:2236-2239): sample_count = 0. The film's
 * contents stay until the next dispatch overwrites them; dispatches recorded by
 * ptc_render_deferred() still show in a read-back that comes before it. */
int ptc_film_reset(ptc_film *film);
uint32_t ptc_film_sample_count(const ptc_film *film);
/* Continue from an externally supplied sample count (sharded rendering). */
int ptc_film_set_sample_count(ptc_film *film, uint32_t sample_count);
int ptc_film_read_accum(ptc_film *film, float *rgba_out);    /* 4*w*h floats */
int ptc_film_read_rgba8(ptc_film *film, uint8_t *rgba_out);  /* 4*w*h bytes */
/* Device address of the RGBA32F accumulator. Launches the film's recorded
 * dispatches first (asynchronously, on the context's stream); NULL, with the
 * reason in ptc_last_error(), if that launch fails. */
void *ptc_film_accum_device_ptr(ptc_film *film);
/* Tone-map the accumulator into the RGBA8 target again (after an external
 * reduction wrote the accumulator). sum_mode: divide rgb by alpha first. */
int ptc_film_resolve(ptc_film *film, int sum_mode);

/* ---- render: the traced part of draw_frame() (src/renderer.cpp:2064-2110),
 * i.e. vkCmdTraceRaysKHR(W, H, 1) with the push constants built from camera. */
int ptc_render(ptc_context *ctx, const ptc_scene *scene, ptc_film *film,
               const ptc_camera *camera, const ptc_render_params *params);

/* ptc_render() for callers that cannot allocate device memory: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc. It has
 * no counterpart in the reference. Same arguments, same film bit for bit; the
 * queue-based integrators take their workspace from a 64 MiB array that the
 * library holds statically on every GPU it is loaded on, instead of the
 * context's hipMalloc'ed one, and the call creates no events (ptc_render_times
 * and ptc_phase_times do not see it). It calls no hipMalloc / hipFree and never
 * synchronises. Limit: one dispatch (width * height * samples_per_frame paths)
 * must fit the array — 76 bytes per path with the wavefront integrator, 60 of
 * them per pixel of the film rounded up to whole 32x32-pixel groups (880 k
 * paths: the reference's default 640x480 film at one sample per frame fits
 * twice over; 156 bytes per path when wavefront_bounces > 0), 16 with the
 * stream integrator (4 M paths); the megakernel uses no workspace. Beyond it:
 * PTC_E_LIMIT with the sizes in ptc_last_error(), nothing rendered. n_frames
 * beyond what fits run in several batches, as in ptc_render().
 * All calls on one GPU share the array (also from different contexts): calls
 * that overlap in time on one GPU race; calls on one context's stream are
 * ordered and safe. */
int ptc_render_static(ptc_context *ctx, const ptc_scene *scene, ptc_film *film,
                      const ptc_camera *camera, const ptc_render_params *params);

/* What ptc_render() (static_workspace = 0) or ptc_render_static() (1) would do
 * with a call of these parameters on a width x height film, without doing it
 * and without a device: the bytes of workspace it takes (0 for the megakernel)
 * and the dispatches it holds in flight per batch. PTC_E_LIMIT, with the
 * reason in ptc_last_error(), for a call that would be refused. No counterpart
 * in the reference, whose allocations are the Vulkan driver's. Either output
 * pointer may be NULL. */
int ptc_render_workspace(uint32_t width, uint32_t height, const ptc_render_params *params,
                         int static_workspace, uint64_t *bytes_out, uint32_t *dispatches_per_batch_out);

/* draw_frame() as the reference issues it: the call records its commands and
 * returns while earlier frames still run (frames_in_flight, src/renderer.hpp:208;
 * the queue submit of src/renderer.cpp:2170-2172 does not wait). Same arguments
 * and same film, bit for bit, as ptc_render(); the dispatch is recorded on the
 * film, not launched. A call that continues the recorded run (same scene,
 * camera bytes and parameters, global_frame_count = the run's next frame count)
 * extends it; anything else launches the run first, as one ptc_render() of all
 * its dispatches, so a loop of one-dispatch calls reaches the rate of a batched
 * call. The run is also launched once it holds max_paths_in_flight paths, by
 * ptc_film_flush(), ptc_synchronize() and ptc_get_stats() / ptc_render_times() /
 * ptc_phase_times(), before the scene is destroyed, and by everything that
 * reads or writes the film (read-backs, ptc_write_png, ptc_film_resolve,
 * ptc_render* -- ptc_render_static launches it through its own static
 * workspace, so it still allocates nothing --, ptc_film_set_sample_count to
 * another count, ptc_film_accum_device_ptr); ptc_film_destroy() drops it.
 * ptc_film_reset() keeps it: in the reference every earlier draw_frame has been
 * submitted when reset_render zeroes the counter (src/renderer.cpp:2236-2239),
 * so whatever reads the film before the next dispatch launches the run first
 * and sees every submitted frame; the next dispatch, which overwrites the film,
 * drops it unlaunched. ptc_film_sample_count() includes the recorded samples
 * (and is 0 after a reset either way). A film over caller-owned memory
 * (ptc_film_create with accum_dev or rgba8_dev) never records: its dispatches
 * are launched at once, as by ptc_render(), because such a caller orders its
 * own work behind them by the stream alone. A launch refused before it started
 * (PTC_E_LIMIT from ptc_render_static's workspace bound, PTC_E_NOMEM) leaves
 * the run recorded; any other error of a deferred launch is returned by the
 * call that launches it and the run is gone. The camera and the parameters are
 * copied; the scene must stay alive (ptc_scene_destroy() launches what uses
 * it). */
int ptc_render_deferred(ptc_context *ctx, const ptc_scene *scene, ptc_film *film,
                        const ptc_camera *camera, const ptc_render_params *params);
/* Launches the dispatches recorded on the film, if any (asynchronous). */
int ptc_film_flush(ptc_film *film);

/* Flattened triangle index hit by the first sample's camera ray of every
 * pixel for one dispatch (the hit the reference would report through
 * gl_InstanceID / gl_PrimitiveID at bounce 0). ids_out: w*h host uint32. */
int ptc_primary_hits(ptc_context *ctx, const ptc_scene *scene,
                     const ptc_camera *camera, uint32_t width, uint32_t height,
                     uint32_t global_frame_count, float tmax,
                     uint32_t *ids_out);

/* Flattened triangle indices (ptc_primary_hits, ptc_hit::flat) as the pair a
 * closest-hit shader of the reference reads: gl_InstanceID (index into the
 * instance list, src/renderer.cpp:1147-1174) and gl_PrimitiveID (triangle
 * within the instance's mesh, closest_hit_common.glsl:61-64). PTC_NO_HIT and
 * indices beyond the scene give PTC_NO_HIT for both. Host-side; no GPU work. */
int ptc_scene_triangle_ids(const ptc_scene *scene, const uint32_t *flat, size_t n,
                           uint32_t *instance_out, uint32_t *primitive_out);

/* Closest-hit queries for caller-supplied rays: one traceRayEXT each
 * (shader.rgen:68-78: opaque, no culling, tmin 0). rays = 6 floats per ray
 * (origin, direction). Host pointers. counters may be NULL; when given the
 * call runs the instrumented kernel and adds nodes/triangles visited. */
int ptc_trace_rays(ptc_context *ctx, const ptc_scene *scene, const float *rays,
                   size_t n_rays, float tmax, ptc_hit *hits_out,
                   ptc_stats *counters);
/* Same with device pointers; nothing is copied or synchronised. */
int ptc_trace_rays_device(ptc_context *ctx, const ptc_scene *scene,
                          const void *rays_dev, size_t n_rays, float tmax,
                          void *hits_dev);

int ptc_get_stats(ptc_context *ctx, ptc_stats *out);
/* Device durations (HIP events on the context's stream) of the ptc_render
 * calls issued since the last call of this function, oldest first, at most
 * 256 kept. Synchronises the stream. *n_out receives the number written. */
int ptc_render_times(ptc_context *ctx, double *ms_out, size_t capacity, size_t *n_out);
/* Device time per kernel phase, summed over the ptc_render calls since the
 * last call of this function (HIP events on the context's stream; synchronises).
 * ms_out / launches_out: PTC_N_PHASES entries each. */
#define PTC_PHASE_MEGAKERNEL 0u /* k_render_mega */
#define PTC_PHASE_PRIMARY 1u    /* fused first bounce of the wavefront integrator */
#define PTC_PHASE_BOUNCE 2u     /* traversal + shading launch pairs */
#define PTC_PHASE_DRAIN 3u      /* persistent launch finishing all live paths */
#define PTC_PHASE_RESOLVE 4u    /* accumulate + tone map */
#define PTC_N_PHASES 5u
int ptc_phase_times(ptc_context *ctx, double *ms_out, uint32_t *launches_out);
int ptc_reset_stats(ptc_context *ctx);
int ptc_synchronize(ptc_context *ctx);

/* write_to_png() (src/renderer.cpp:2241-2327, src/utility.cpp:94-111):
 * read the RGBA8 target back and write a 4-channel PNG with stride 4*w.
 * Returns PTC_OK, or PTC_E_IO with the reference's message text in
 * ptc_last_error(). */
int ptc_write_png(ptc_film *film, const char *file_name);

/* read_hdr_image() (src/utility.cpp:75-92; the reference calls stb_image's
 * stbi_loadf(file, &w, &h, &channels, 4) and uploads the result as the
 * RGBA32F environment map, src/renderer.cpp:1842-1860): decode a Radiance
 * .hdr file (RGBE, flat or run-length-encoded scanlines) to RGBA float, rows
 * top to bottom, alpha 1. With rgba == NULL only the size is returned.
 * capacity_floats must be >= 4 * width * height. Host-side; no GPU needed.
 * Returns PTC_OK, PTC_E_IO (message in ptc_last_error()) or PTC_E_INVALID. */
int ptc_read_hdr(const char *file_name, uint32_t *width, uint32_t *height, float *rgba,
                 size_t capacity_floats);

/* Scene import, the part of open_scene() that runs before
 * create_render_resources() (src/application.cpp:150-200): the reference hands
 * the file to Assimp (third-party, v5.3.1) with PreTransformVertices +
 * normalisation to [-1, 1], Triangulate, FindDegenerates, SortByPType (lines
 * and points removed), FixInfacingNormals, GenSmoothNormals with a maximum
 * smoothing angle of 80 degrees and JoinIdenticalVertices, then reads
 * mVertices / mNormals / mFaces of the meshes. This importer reads Wavefront
 * OBJ and Stanford PLY (ascii and binary) files and applies the same steps
 * with Assimp's tolerances (quads cut from their concave corner, larger
 * polygons ear-clipped; smoothing over corners within 1e-4 of the bounding-box
 * diagonal; joining within 1e-5);
 * mesh i is returned as the ptc_mesh_desc that ptc_scene_create() takes.
 * Pointers in the descriptor stay valid until ptc_mesh_file_close().
 * Failure: PTC_E_IO with the importer's message ("Scene has no meshes", ...)
 * in ptc_last_error(). Host-side; no GPU needed. */
typedef struct ptc_mesh_file ptc_mesh_file;
#define PTC_IMPORT_NO_NORMALIZE 1u /* keep file coordinates (no AI_CONFIG_PP_PTV_NORMALIZE) */
#define PTC_IMPORT_NO_JOIN 2u      /* keep one vertex per triangle corner */
#define PTC_IMPORT_EXACT_MATCH 4u  /* smoothing and joining match positions and normals bit for bit */
#define PTC_IMPORT_FAN 8u          /* every polygon is split as a fan from its first corner */
typedef struct ptc_import_options
{
    float max_smoothing_angle_deg; /* AI_CONFIG_PP_GSN_MAX_SMOOTHING_ANGLE; 0 = 80 */
    uint32_t flags;                /* PTC_IMPORT_* */
    /* GenSmoothNormals treats corners as one position when they lie within this
     * fraction of the diagonal of the mesh's bounding box; 0 = Assimp's 1e-4. */
    float position_epsilon_rel;
    /* JoinIdenticalVertices joins a corner with an earlier vertex whose position
     * and normal are both within this distance; 0 = Assimp's 1e-5. */
    float join_epsilon;
    uint32_t reserved[4];
} ptc_import_options;
int ptc_mesh_file_open(const char *file_name, const ptc_import_options *options, ptc_mesh_file **out);
uint32_t ptc_mesh_file_mesh_count(const ptc_mesh_file *file);
int ptc_mesh_file_get_mesh(const ptc_mesh_file *file, uint32_t index, ptc_mesh_desc *out);
void ptc_mesh_file_close(ptc_mesh_file *file);

/* Element-wise evaluation of the device math primitives the kernels use, so
 * a checker can compare them with its own: fn 0 sincos(a) -> out, out2;
 * 1 log(a); 2 acos(a); 3 atan2(a, b); 4 a / b; 5 sqrt(a);
 * 6 (float)(uint32 bits of a) * 2^-32; 7 zeros with the signs of sin(a) and
 * cos(a) as the zero-aperture camera path derives them without evaluating
 * either (a >= 0) -> out, out2. Host pointers, n floats each. */
int ptc_math_eval(ptc_context *ctx, int fn, const float *a, const float *b,
                  float *out, float *out2, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* PTC_H */
