"""Host-side mirror of the reference renderer interface over libptcore's C ABI.

The free functions keep the names, argument meaning and error behaviour of
the reference's src/renderer.hpp:256-278 (create_context,
create_render_resources, draw_frame, reset_render, write_to_png) and
src/camera.cpp:65-96 (create_camera, camera_set_distance). Everything that
computes runs in the HIP library; there is no CPU fallback: if libptcore.so is
missing or no GPU is present these calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PTC_LIBRARY names another build of the library: kernel experiments are measured side by side that way)
LIB_PATH = os.environ.get("PTC_LIBRARY") or os.path.join(_HERE, "libptcore.so")

KIND_DIFFUSE, KIND_SPECULAR, KIND_EMISSIVE, KIND_DIELECTRIC = 0, 1, 2, 3
TONEMAP_PBR_NEUTRAL, TONEMAP_ACES = 0, 1
INTEGRATOR_MEGAKERNEL, INTEGRATOR_WAVEFRONT, INTEGRATOR_STREAM, INTEGRATOR_AUTO = 0, 1, 2, 3
ACCUM_RUNNING_MEAN, ACCUM_SUM = 0, 1
NO_HIT = 0xFFFFFFFF


class PtcError(RuntimeError):
    pass


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    """Byte-identical to the reference's Camera (src/camera.hpp:6-19)."""

    _fields_ = [
        ("distance", C.c_float),
        ("position", Vec3),
        ("target", Vec3),
        ("direction_x", Vec3),
        ("direction_y", Vec3),
        ("direction_z", Vec3),
        ("sensor_distance", C.c_float),
        ("sensor_half_width", C.c_float),
        ("sensor_half_height", C.c_float),
        ("focus_distance", C.c_float),
        ("aperture_radius", C.c_float),
    ]

    def as_dict(self):
        v = lambda a: (a.x, a.y, a.z)
        return dict(position=v(self.position), direction_x=v(self.direction_x),
                    direction_y=v(self.direction_y), direction_z=v(self.direction_z),
                    sensor_distance=self.sensor_distance,
                    sensor_half_width=self.sensor_half_width,
                    sensor_half_height=self.sensor_half_height,
                    focus_distance=self.focus_distance,
                    aperture_radius=self.aperture_radius)

    def as_array(self) -> np.ndarray:
        return np.frombuffer(bytes(self), dtype=np.float32).copy()


assert C.sizeof(Camera) == 84


class MeshDesc(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("indices", C.c_void_p),
                ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32)]


class InstanceDesc(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("mesh", C.c_uint32), ("kind", C.c_uint32)]


class EnvmapDesc(C.Structure):
    _fields_ = [("rgba", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class ImportOptions(C.Structure):
    _fields_ = [("max_smoothing_angle_deg", C.c_float), ("flags", C.c_uint32), ("position_epsilon_rel", C.c_float),
                ("join_epsilon", C.c_float), ("reserved", C.c_uint32 * 4)]


IMPORT_NO_NORMALIZE, IMPORT_NO_JOIN, IMPORT_EXACT_MATCH, IMPORT_FAN = 1, 2, 4, 8


OPTIMIZE_NONE = 0xffffffff


class BuildOptions(C.Structure):
    _fields_ = [("search_radius", C.c_uint32), ("two_level", C.c_uint32), ("collapse", C.c_uint32),
                ("tri_cost", C.c_float), ("optimize_rounds", C.c_uint32), ("reserved", C.c_uint32 * 3)]


class SceneInfo(C.Structure):
    _fields_ = [("n_triangles", C.c_uint32), ("n_nodes", C.c_uint32),
                ("n_instances", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("build_ms", C.c_float), ("sah_cost", C.c_float),
                ("device_bytes", C.c_uint64), ("n_bvh_triangles", C.c_uint32), ("reserved", C.c_uint32)]


class RenderParams(C.Structure):
    _fields_ = [("global_frame_count", C.c_uint32), ("samples_per_frame", C.c_uint32),
                ("n_frames", C.c_uint32), ("frame_stride", C.c_uint32),
                ("max_depth", C.c_uint32), ("tmax", C.c_float),
                ("integrator", C.c_uint32), ("accumulate", C.c_uint32),
                ("instrument", C.c_uint32), ("wavefront_bounces", C.c_uint32),
                ("tonemap", C.c_uint32), ("max_paths_in_flight", C.c_uint32),
                ("debug_view", C.c_uint32), ("regroup", C.c_uint32)]


def render_workspace(width: int, height: int, n_frames: int = 1, samples_per_frame: int = 1,
                     integrator: int = 3, max_depth: int = 32, wavefront_bounces: int = 0,
                     max_paths_in_flight: int = 0, static: bool = False):
    """ptc_render_workspace: (bytes of workspace, dispatches in flight per batch)
    of such a ptc_render / ptc_render_static call; host arithmetic, no device.
    (integrator 3 = PTC_INTEGRATOR_AUTO.)"""
    p = RenderParams()
    p.n_frames, p.samples_per_frame, p.integrator, p.max_depth = n_frames, samples_per_frame, integrator, max_depth
    p.wavefront_bounces, p.max_paths_in_flight = wavefront_bounces, max_paths_in_flight
    nbytes, batch = C.c_uint64(0), C.c_uint32(0)
    _check(lib().ptc_render_workspace(width, height, C.byref(p), int(static), C.byref(nbytes), C.byref(batch)))
    return nbytes.value, batch.value


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("paths", C.c_uint64), ("nodes_visited", C.c_uint64),
                ("tris_tested", C.c_uint64), ("render_ms", C.c_double),
                ("nodes_wave_uniform", C.c_uint64), ("uniform_fetches", C.c_uint64)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("flat", "<u4")])

# every symbol include/ptc.h declares
EXPORTS = [
    "ptc_context_create", "ptc_context_destroy", "ptc_context_set_stream", "ptc_last_error",
    "ptc_scene_create", "ptc_scene_destroy", "ptc_scene_get_info", "ptc_scene_read_bvh",
    "ptc_film_create", "ptc_film_destroy", "ptc_film_reset", "ptc_film_sample_count",
    "ptc_film_set_sample_count", "ptc_film_read_accum", "ptc_film_read_rgba8",
    "ptc_film_accum_device_ptr", "ptc_film_resolve", "ptc_render", "ptc_render_static", "ptc_render_deferred", "ptc_film_flush", "ptc_render_workspace",
    "ptc_primary_hits", "ptc_scene_triangle_ids",
    "ptc_trace_rays", "ptc_trace_rays_device", "ptc_get_stats", "ptc_render_times",
    "ptc_phase_times",
    "ptc_reset_stats",
    "ptc_synchronize", "ptc_write_png", "ptc_math_eval", "ptc_read_hdr",
    "ptc_mesh_file_open", "ptc_mesh_file_mesh_count", "ptc_mesh_file_get_mesh", "ptc_mesh_file_close",
]

_lib = None


def lib():
    """Load libptcore.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtcError(f"{LIB_PATH} not found: build it with __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        vp, u32, sz, f32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_float
        L.ptc_last_error.restype = C.c_char_p
        L.ptc_context_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.ptc_context_destroy.argtypes = [vp]
        L.ptc_context_destroy.restype = None
        L.ptc_context_set_stream.argtypes = [vp, vp]
        L.ptc_scene_create.argtypes = [vp, vp, u32, vp, u32, vp, vp, C.POINTER(vp)]
        L.ptc_scene_destroy.argtypes = [vp]
        L.ptc_scene_destroy.restype = None
        L.ptc_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
        L.ptc_scene_read_bvh.argtypes = [vp, vp, sz, vp, sz]
        L.ptc_film_create.argtypes = [vp, u32, u32, vp, vp, C.POINTER(vp)]
        L.ptc_film_destroy.argtypes = [vp]
        L.ptc_film_destroy.restype = None
        L.ptc_film_reset.argtypes = [vp]
        L.ptc_film_sample_count.argtypes = [vp]
        L.ptc_film_sample_count.restype = u32
        L.ptc_film_set_sample_count.argtypes = [vp, u32]
        L.ptc_film_read_accum.argtypes = [vp, vp]
        L.ptc_film_read_rgba8.argtypes = [vp, vp]
        L.ptc_film_accum_device_ptr.argtypes = [vp]
        L.ptc_film_accum_device_ptr.restype = vp
        L.ptc_film_resolve.argtypes = [vp, C.c_int]
        L.ptc_render.argtypes = [vp, vp, vp, C.POINTER(Camera), C.POINTER(RenderParams)]
        L.ptc_render_static.argtypes = [vp, vp, vp, C.POINTER(Camera), C.POINTER(RenderParams)]
        L.ptc_render_deferred.argtypes = [vp, vp, vp, C.POINTER(Camera), C.POINTER(RenderParams)]
        L.ptc_render_workspace.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(RenderParams), C.c_int,
                                           C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.ptc_film_flush.argtypes = [vp]
        L.ptc_primary_hits.argtypes = [vp, vp, C.POINTER(Camera), u32, u32, u32, f32, vp]
        L.ptc_scene_triangle_ids.argtypes = [vp, vp, sz, vp, vp]
        L.ptc_trace_rays.argtypes = [vp, vp, vp, sz, f32, vp, C.POINTER(Stats)]
        L.ptc_trace_rays_device.argtypes = [vp, vp, vp, sz, f32, vp]
        L.ptc_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.ptc_render_times.argtypes = [vp, vp, sz, C.POINTER(sz)]
        L.ptc_phase_times.argtypes = [vp, vp, vp]
        L.ptc_reset_stats.argtypes = [vp]
        L.ptc_synchronize.argtypes = [vp]
        L.ptc_write_png.argtypes = [vp, C.c_char_p]
        L.ptc_math_eval.argtypes = [vp, C.c_int, vp, vp, vp, vp, sz]
        L.ptc_read_hdr.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp, sz]
        L.ptc_mesh_file_open.argtypes = [C.c_char_p, C.POINTER(ImportOptions), C.POINTER(vp)]
        L.ptc_mesh_file_mesh_count.argtypes = [vp]
        L.ptc_mesh_file_mesh_count.restype = u32
        L.ptc_mesh_file_get_mesh.argtypes = [vp, u32, C.POINTER(MeshDesc)]
        L.ptc_mesh_file_close.argtypes = [vp]
        L.ptc_mesh_file_close.restype = None
        _lib = L
    return _lib


def _check(rc: int):
    if rc != 0:
        raise PtcError(f"libptcore error {rc}: {lib().ptc_last_error().decode()}")


# ---- camera: src/camera.cpp:65-96, src/vec3.hpp (float32 arithmetic) ---------

_f = np.float32


def _dot(u, v):
    return _f(_f(_f(u[0] * v[0]) + _f(u[1] * v[1])) + _f(u[2] * v[2]))


def _cross(u, v):
    return np.array([_f(_f(u[1] * v[2]) - _f(u[2] * v[1])),
                     _f(_f(u[2] * v[0]) - _f(u[0] * v[2])),
                     _f(_f(u[0] * v[1]) - _f(u[1] * v[0]))], dtype=np.float32)


def _norm(v):
    return _f(np.sqrt(_dot(v, v)))


def _normalize(v):
    return (v * _f(_f(1.0) / _norm(v))).astype(np.float32)


def _set3(dst: Vec3, v):
    dst.x, dst.y, dst.z = float(v[0]), float(v[1]), float(v[2])


def create_camera(position, target, sensor_distance, sensor_half_width,
                  sensor_half_height) -> Camera:
    """create_camera() of src/camera.cpp:65-90. (target - position) and the
    world up vector (0,1,0) must not be colinear."""
    position = np.asarray(position, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    world_up = np.array([0.0, 1.0, 0.0], dtype=np.float32)
    cam = Camera()
    d = (target - position).astype(np.float32)
    cam.distance = float(_norm(d))
    _set3(cam.position, position)
    _set3(cam.target, target)
    dz = _normalize(d)
    dx = _normalize(_cross(dz, world_up))
    dy = _normalize(_cross(dz, dx))
    _set3(cam.direction_z, dz)
    _set3(cam.direction_x, dx)
    _set3(cam.direction_y, dy)
    cam.sensor_distance = sensor_distance
    cam.sensor_half_width = sensor_half_width
    cam.sensor_half_height = sensor_half_height
    cam.focus_distance = cam.distance
    cam.aperture_radius = 0.0
    return cam


def camera_set_distance(camera: Camera, distance: float) -> None:
    """camera_set_distance() of src/camera.cpp:92-96."""
    dz = np.array([camera.direction_z.x, camera.direction_z.y, camera.direction_z.z],
                  dtype=np.float32)
    tgt = np.array([camera.target.x, camera.target.y, camera.target.z], dtype=np.float32)
    camera.distance = distance
    _set3(camera.position, (tgt - _f(distance) * _normalize(dz)).astype(np.float32))


def default_camera(width: int, height: int) -> Camera:
    """The camera open_scene() sets up (src/application.cpp:209-222): at
    (0,0,3.5) looking at the origin, vertical field of view 45 degrees."""
    vfov = _f(45.0 / 180.0 * 3.1415927)
    aspect = _f(width) / _f(height)
    shh = _f(np.tan(_f(vfov * _f(0.5))))
    return create_camera((0.0, 0.0, 3.5), (0.0, 0.0, 0.0), 1.0, float(aspect * shh), float(shh))


# ---- scene description (what the reference reads out of aiScene) -------------


@dataclass
class Mesh:
    positions: np.ndarray  # float32 [V, 3]  (aiMesh::mVertices)
    normals: np.ndarray    # float32 [V, 3]  (aiMesh::mNormals)
    indices: np.ndarray    # uint32  [F, 3]  (aiMesh::mFaces)


@dataclass
class Instance:
    transform: np.ndarray  # float32 [3, 4] object-to-world
    mesh: int
    kind: int


@dataclass
class Scene:
    meshes: list = field(default_factory=list)
    instances: list = field(default_factory=list)
    envmap: np.ndarray | None = None  # float32 [H, W, 4]

    def oracle_args(self):
        return ([(m.positions, m.normals, m.indices) for m in self.meshes],
                [(i.transform, i.mesh, i.kind) for i in self.instances], self.envmap)

    @property
    def n_triangles(self) -> int:
        return sum(self.meshes[i.mesh].indices.shape[0] for i in self.instances)


IDENTITY = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=np.float32)


# ---- context / resources -------------------------------------------------------


class RenderContext:
    """create_context() (src/renderer.hpp:256): device + stream."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib().ptc_context_create(device, C.byref(h)))
        self.handle = h
        self.device = device
        self.global_frame_count = 0  # Vulkan_context::global_frame_count
        self._children = []  # weak references to scenes / films created on this context

    def set_stream(self, hip_stream: int):
        _check(lib().ptc_context_set_stream(self.handle, C.c_void_p(hip_stream)))

    def synchronize(self):
        _check(lib().ptc_synchronize(self.handle))

    def stats(self) -> Stats:
        s = Stats()
        _check(lib().ptc_get_stats(self.handle, C.byref(s)))
        return s

    def reset_stats(self):
        _check(lib().ptc_reset_stats(self.handle))

    def render_times(self) -> np.ndarray:
        """Device milliseconds of every ptc_render since the last call."""
        out = np.empty(256, np.float64)
        n = C.c_size_t(0)
        _check(lib().ptc_render_times(self.handle, out.ctypes.data, out.size, C.byref(n)))
        return out[: n.value].copy()

    PHASES = ("megakernel", "primary", "bounce", "drain", "resolve")

    def phase_times(self) -> dict:
        """{phase: (device ms, launches)} summed over the renders since the last call."""
        ms = np.zeros(len(self.PHASES), np.float64)
        n = np.zeros(len(self.PHASES), np.uint32)
        _check(lib().ptc_phase_times(self.handle, ms.ctypes.data, n.ctypes.data))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(self.PHASES)}

    def math_eval(self, fn: int, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32) if b is not None else np.ones_like(a)
        out, out2 = np.empty_like(a), np.empty_like(a)
        _check(lib().ptc_math_eval(self.handle, fn, a.ctypes.data, b.ctypes.data,
                                   out.ctypes.data, out2.ctypes.data, a.size))
        return (out, out2) if fn == 0 else out

    def _adopt(self, child):
        import weakref
        self._children.append(weakref.ref(child))

    def close(self):
        """Destroys the context after every scene and film created on it."""
        if self.handle:
            for ref in reversed(self._children):
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            lib().ptc_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_context(device: int = 0) -> RenderContext:
    return RenderContext(device)


class DeviceScene:
    def __init__(self, ctx: RenderContext, scene: Scene, search_radius: int = 0, two_level: bool = False,
                 collapse: int = 0, tri_cost: float = 0.0, optimize_rounds: int = 0):
        # optimize_rounds: 0 = the library's default, OPTIMIZE_NONE = no reinsertion rounds
        self.ctx = ctx
        self.two_level = two_level
        keep = []
        md = (MeshDesc * max(1, len(scene.meshes)))()
        for i, m in enumerate(scene.meshes):
            p = np.ascontiguousarray(m.positions, dtype=np.float32).reshape(-1, 3)
            n = np.ascontiguousarray(m.normals, dtype=np.float32).reshape(-1, 3)
            idx = np.ascontiguousarray(m.indices, dtype=np.uint32).reshape(-1, 3)
            if n.shape != p.shape:
                raise PtcError("mesh normals and positions differ in shape")
            keep += [p, n, idx]
            md[i].positions, md[i].normals, md[i].indices = p.ctypes.data, n.ctypes.data, idx.ctypes.data
            md[i].n_vertices, md[i].n_triangles = p.shape[0], idx.shape[0]
        idesc = (InstanceDesc * max(1, len(scene.instances)))()
        for i, inst in enumerate(scene.instances):
            idesc[i].transform[:] = [float(v) for v in np.asarray(inst.transform, np.float32).reshape(12)]
            idesc[i].mesh, idesc[i].kind = inst.mesh, inst.kind
        env = None
        if scene.envmap is not None:
            e = np.ascontiguousarray(scene.envmap, dtype=np.float32)
            if e.ndim != 3 or e.shape[2] != 4:
                raise PtcError("environment map must be [H, W, 4] float32")
            keep.append(e)
            env = EnvmapDesc(e.ctypes.data, e.shape[1], e.shape[0])
        opts = BuildOptions()
        opts.search_radius = search_radius
        opts.two_level = int(two_level)
        opts.collapse = collapse
        opts.tri_cost = tri_cost
        opts.optimize_rounds = optimize_rounds
        h = C.c_void_p()
        _check(lib().ptc_scene_create(ctx.handle, md, len(scene.meshes), idesc,
                                      len(scene.instances), C.byref(env) if env else None,
                                      C.byref(opts), C.byref(h)))
        self.handle = h
        ctx._adopt(self)
        self.info = SceneInfo()
        _check(lib().ptc_scene_get_info(h, C.byref(self.info)))

    def read_bvh(self):
        nodes = np.empty((self.info.n_nodes, 80), np.uint8)
        tris = np.empty((self.info.n_bvh_triangles, 48), np.uint8)
        _check(lib().ptc_scene_read_bvh(self.handle, nodes.ctypes.data, nodes.shape[0],
                                        tris.ctypes.data, tris.shape[0]))
        return nodes, tris

    def trace(self, rays, tmax: float = 10000.0, count: bool = False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        hits = np.empty(rays.shape[0], HIT_DTYPE)
        st = Stats()
        _check(lib().ptc_trace_rays(self.ctx.handle, self.handle, rays.ctypes.data,
                                    rays.shape[0], tmax, hits.ctypes.data,
                                    C.byref(st) if count else None))
        return (hits, st) if count else hits

    def primary_hits(self, camera: Camera, width: int, height: int, frame: int = 0,
                     tmax: float = 10000.0):
        ids = np.empty((height, width), np.uint32)
        _check(lib().ptc_primary_hits(self.ctx.handle, self.handle, C.byref(camera), width,
                                      height, frame, tmax, ids.ctypes.data))
        return ids

    def triangle_ids(self, flat):
        """(instance, primitive) of flattened triangle indices: what gl_InstanceID /
        gl_PrimitiveID would read in the reference's closest-hit shaders."""
        flat = np.ascontiguousarray(flat, dtype=np.uint32)
        inst, prim = np.empty_like(flat), np.empty_like(flat)
        _check(lib().ptc_scene_triangle_ids(self.handle, flat.ctypes.data, flat.size, inst.ctypes.data, prim.ctypes.data))
        return inst, prim

    def close(self):
        if self.handle:
            lib().ptc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RenderResources:
    """Vulkan_render_resources (src/renderer.hpp:223-254): scene + storage image
    + render target + the three sample counters the UI edits."""

    def __init__(self, ctx: RenderContext, width: int, height: int, scene,
                 accum_dev: int = 0, rgba8_dev: int = 0, search_radius: int = 0, two_level: bool = False,
                 optimize_rounds: int = 0):
        self.ctx = ctx
        self.width, self.height = width, height
        self.scene = scene if isinstance(scene, DeviceScene) else DeviceScene(ctx, scene, search_radius, two_level,
                                                                                optimize_rounds=optimize_rounds)
        h = C.c_void_p()
        _check(lib().ptc_film_create(ctx.handle, width, height, C.c_void_p(accum_dev or None),
                                     C.c_void_p(rgba8_dev or None), C.byref(h)))
        self.film = h
        ctx._adopt(self)
        # src/renderer.cpp:1950-1952
        self.samples_to_render = 1000
        self.samples_per_frame = 1
        # knobs the reference hard-codes in shader.rgen
        self.max_depth = 32
        self.tmax = 10000.0
        self.integrator = INTEGRATOR_AUTO  # like the C++ mirror; all integrators give the same film
        self.accumulate = ACCUM_RUNNING_MEAN
        self.instrument = False
        self.wavefront_bounces = 0  # 0 = library default
        self.max_paths_in_flight = 0  # 0 = library default (2^29)
        self.tonemap = TONEMAP_PBR_NEUTRAL  # the one the reference compiles in (shader.rgen:218-219)
        self.debug_view = 0  # 1 = bounce counts / 16 instead of radiance (shader.rgen:195-199, the "#else" branch)
        self.regroup = 0  # 1 = the wavefront integrator sorts the queue by ray origin cell and octant
        # draw_frame() records its dispatch and returns, as the reference's does with its frames in flight
        # (src/renderer.hpp:208); False = every draw_frame() launches its own dispatch at once
        self.frames_in_flight = True

    @property
    def sample_count(self) -> int:
        return lib().ptc_film_sample_count(self.film)

    @sample_count.setter
    def sample_count(self, v: int):
        _check(lib().ptc_film_set_sample_count(self.film, v))

    def render(self, camera: Camera, frame: int, samples_per_frame: int = 1, n_frames: int = 1,
               frame_stride: int = 1, static: bool = False, deferred: bool = False):
        """ptc_render; with static=True ptc_render_static (no allocation; the
        library's static workspace bounds the paths of one dispatch); with
        deferred=True ptc_render_deferred (recorded on the film, launched together
        with the calls that continue it)."""
        p = RenderParams()
        p.global_frame_count = frame
        p.samples_per_frame = samples_per_frame
        p.n_frames = n_frames
        p.frame_stride = frame_stride
        p.max_depth = self.max_depth
        p.tmax = self.tmax
        p.integrator = self.integrator
        p.accumulate = self.accumulate
        p.instrument = int(self.instrument)
        p.wavefront_bounces = self.wavefront_bounces
        p.max_paths_in_flight = self.max_paths_in_flight
        p.tonemap = self.tonemap
        p.debug_view = self.debug_view
        p.regroup = self.regroup
        call = lib().ptc_render_static if static else lib().ptc_render_deferred if deferred else lib().ptc_render
        _check(call(self.ctx.handle, self.scene.handle, self.film, C.byref(camera), C.byref(p)))

    def flush(self):
        """Launches the dispatches draw_frame() has recorded and not launched yet."""
        _check(lib().ptc_film_flush(self.film))

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.float32)
        _check(lib().ptc_film_read_accum(self.film, out.ctypes.data))
        return out

    def read_rgba8(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), np.uint8)
        _check(lib().ptc_film_read_rgba8(self.film, out.ctypes.data))
        return out

    def accum_device_ptr(self) -> int:
        return lib().ptc_film_accum_device_ptr(self.film) or 0

    def resolve(self, sum_mode: bool):
        _check(lib().ptc_film_resolve(self.film, int(sum_mode)))

    def close(self):
        if self.film:
            lib().ptc_film_destroy(self.film)
            self.film = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_render_resources(context: RenderContext, render_width: int, render_height: int,
                            scene, **kw) -> RenderResources:
    """create_render_resources() (src/renderer.cpp:1738-1955)."""
    return RenderResources(context, render_width, render_height, scene, **kw)


def draw_frame(context: RenderContext, render_resources: RenderResources, camera: Camera) -> None:
    """The traced part of draw_frame() (src/renderer.cpp:2015-2130, :2224):
    renders min(remaining, samples_per_frame) samples when the target has not
    been reached, and advances global_frame_count on every call. Like the
    reference's, the call returns once its dispatch is recorded
    (src/renderer.cpp:2170-2172): consecutive calls with one camera are launched
    together, at the latest when something reads the film."""
    rr = render_resources
    sc = rr.sample_count
    if sc < rr.samples_to_render:
        samples_this_frame = min(rr.samples_to_render - sc, rr.samples_per_frame)
        rr.render(camera, context.global_frame_count, samples_this_frame, deferred=rr.frames_in_flight)
    context.global_frame_count = (context.global_frame_count + 1) & 0xFFFFFFFF


def draw_frames(context: RenderContext, render_resources: RenderResources, camera: Camera, n: int) -> None:
    """n consecutive draw_frame() calls with one camera as one call into the
    library: the same film and counters afterwards, but the dispatches that
    render a full samples_per_frame are in flight together (what the wavefront
    integrator needs to reach its rate). The partial last dispatch and calls
    beyond samples_to_render go through draw_frame()."""
    rr = render_resources
    spf = max(rr.samples_per_frame, 1)
    remaining = max(rr.samples_to_render - rr.sample_count, 0)
    n_full = min(n, remaining // spf)
    if n_full > 1:
        rr.render(camera, context.global_frame_count, spf, n_frames=n_full)
        context.global_frame_count = (context.global_frame_count + n_full) & 0xFFFFFFFF
        n -= n_full
    for _ in range(n):
        draw_frame(context, rr, camera)


def reset_render(render_resources: RenderResources) -> None:
    """reset_render() (src/renderer.cpp:2236-2239)."""
    _check(lib().ptc_film_reset(render_resources.film))


def import_meshes(file_name: str, max_smoothing_angle: float = 80.0, normalize: bool = True,
                  join_identical_vertices: bool = True, exact_match: bool = False, fan: bool = False,
                  position_epsilon_rel: float = 0.0, join_epsilon: float = 0.0) -> list:
    """The import half of open_scene() (src/application.cpp:150-200): reads a
    Wavefront OBJ or Stanford PLY file and post-processes it the way the
    reference configures Assimp (normalise to [-1, 1], triangulate, drop
    degenerate faces, smooth normals up to 80 degrees where the file has none,
    join identical vertices). Returns the meshes; raises PtcError with the
    importer's message otherwise."""
    opts = ImportOptions()
    opts.max_smoothing_angle_deg = max_smoothing_angle
    opts.flags = ((0 if normalize else IMPORT_NO_NORMALIZE) | (0 if join_identical_vertices else IMPORT_NO_JOIN) |
                  (IMPORT_EXACT_MATCH if exact_match else 0) | (IMPORT_FAN if fan else 0))
    opts.position_epsilon_rel = position_epsilon_rel
    opts.join_epsilon = join_epsilon
    h = C.c_void_p()
    _check(lib().ptc_mesh_file_open(os.fsencode(file_name), C.byref(opts), C.byref(h)))
    try:
        meshes = []
        for i in range(lib().ptc_mesh_file_mesh_count(h)):
            d = MeshDesc()
            _check(lib().ptc_mesh_file_get_mesh(h, i, C.byref(d)))
            nv, nt = d.n_vertices, d.n_triangles
            as_f32 = lambda ptr: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), (nv, 3)).copy()
            idx = np.ctypeslib.as_array(C.cast(d.indices, C.POINTER(C.c_uint32)), (nt, 3)).copy()
            meshes.append(Mesh(as_f32(d.positions), as_f32(d.normals), idx))
        return meshes
    finally:
        lib().ptc_mesh_file_close(h)


def read_hdr_image(file_name: str) -> np.ndarray:
    """read_hdr_image() (src/utility.cpp:75-92): a Radiance .hdr file as an
    (height, width, 4) float32 array, rows top to bottom, alpha 1. Raises
    PtcError with the decoder's reason on failure, as the reference throws."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    name = os.fsencode(file_name)
    _check(lib().ptc_read_hdr(name, C.byref(w), C.byref(h), None, 0))
    out = np.empty((h.value, w.value, 4), np.float32)
    _check(lib().ptc_read_hdr(name, C.byref(w), C.byref(h), out.ctypes.data, out.size))
    return out


def write_to_png(context: RenderContext, render_resources: RenderResources, file_name: str) -> str:
    """write_to_png() (src/renderer.cpp:2241-2327): returns "" on success and
    the error message on failure."""
    rc = lib().ptc_write_png(render_resources.film, os.fsencode(file_name))
    return "" if rc == 0 else lib().ptc_last_error().decode()
