"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The shipped package never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

MATH_PORTABLE = 0
MATH_LIBM = 1
NO_HIT = 0xFFFFFFFF


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when the reference tree exists)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "all"])


class MeshDesc(C.Structure):
    _fields_ = [
        ("positions", C.c_void_p),
        ("normals", C.c_void_p),
        ("indices", C.c_void_p),
        ("n_vertices", C.c_uint32),
        ("n_triangles", C.c_uint32),
    ]


class InstanceDesc(C.Structure):
    _fields_ = [
        ("transform", C.c_float * 12),
        ("mesh", C.c_uint32),
        ("kind", C.c_uint32),
    ]


class PushConstants(C.Structure):
    _fields_ = [
        ("global_frame_count", C.c_uint32),
        ("sample_count", C.c_uint32),
        ("samples_per_frame", C.c_uint32),
        ("camera_position", C.c_float * 3),
        ("camera_dir_x", C.c_float * 3),
        ("camera_dir_y", C.c_float * 3),
        ("camera_dir_z", C.c_float * 3),
        ("sensor_distance", C.c_float),
        ("sensor_half_width", C.c_float),
        ("sensor_half_height", C.c_float),
        ("focus_distance", C.c_float),
        ("aperture_radius", C.c_float),
    ]


assert C.sizeof(PushConstants) == 80

HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("id", "<u4")])


class Stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("paths", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("tmax", C.c_float),
        ("n_frames", C.c_uint32),
        ("accumulate_sum", C.c_uint32),
        ("use_bvh", C.c_int),
        ("n_threads", C.c_int),
        ("tonemap", C.c_int),
        ("debug_view", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.pt_hash.restype = C.c_uint32
        L.pt_hash.argtypes = [C.c_uint32]
        L.pt_random.restype = C.c_float
        L.pt_random.argtypes = [C.POINTER(C.c_uint32)]
        L.pt_seed.restype = C.c_uint32
        L.pt_seed.argtypes = [C.c_uint32, C.c_uint32]
        L.pt_unorm8.restype = C.c_uint8
        L.pt_unorm8.argtypes = [C.c_float]
        L.pt_scene_create.restype = C.c_void_p
        L.pt_scene_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                      C.c_void_p, C.c_uint32, C.c_uint32]
        L.pt_scene_destroy.argtypes = [C.c_void_p]
        L.pt_scene_set_two_level.argtypes = [C.c_void_p, C.c_int]
        L.pt_scene_set_two_level.restype = None
        L.pt_scene_triangle_count.restype = C.c_uint32
        L.pt_scene_triangle_count.argtypes = [C.c_void_p]
        L.pt_scene_triangle_ids.argtypes = [C.c_void_p, C.c_uint32,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_scene_world_triangles.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_scene_bvh_sah_cost.restype = C.c_double
        L.pt_scene_bvh_sah_cost.argtypes = [C.c_void_p]
        L.pt_scene_wide_sah_cost.restype = C.c_double
        L.pt_scene_wide_sah_cost.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.pt_miss.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                   C.c_int, C.c_void_p, C.c_void_p]
        L.pt_bvh8_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_size_t, C.c_float, C.c_void_p,
                                        C.c_void_p]
        L.pt_bvh8_validate.restype = C.c_int
        L.pt_bvh8_validate.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.POINTER(C.c_double)]
        L.pt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_void_p]
        L.pt_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_render_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_math_eval.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t]
        L.pt_camera_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p,
                                    C.c_void_p]
        L.pt_offset_position_along_normal.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.pt_material_scatter.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.pt_tonemap_pbr_neutral.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_tonemap_aces.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_inverse_affine.argtypes = [C.c_void_p, C.c_void_p]
        L.pt_set_math_mode.argtypes = [C.c_int]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def set_math_mode(mode: int) -> None:
    lib().pt_set_math_mode(mode)


def hash32(x: int) -> int:
    return lib().pt_hash(x & 0xFFFFFFFF)


def seed(pixel_index: int, frame: int) -> int:
    return lib().pt_seed(pixel_index, frame)


def random_stream(state: int, n: int):
    """n draws of random(); returns (values f32[n], states u32[n])."""
    s = C.c_uint32(state)
    vals = np.empty(n, np.float32)
    states = np.empty(n, np.uint32)
    L = lib()
    for i in range(n):
        vals[i] = L.pt_random(C.byref(s))
        states[i] = s.value
    return vals, states


def math_eval(fn: int, mode: int, a, b=None):
    a = _f32(a)
    b = _f32(b) if b is not None else np.zeros_like(a)
    out = np.empty_like(a)
    out2 = np.empty_like(a)
    lib().pt_math_eval(fn, mode, _ptr(a), _ptr(b), _ptr(out), _ptr(out2), a.size)
    return (out, out2) if fn == 0 else out


def make_push(camera, frame=0, sample_count=0, samples_per_frame=1) -> PushConstants:
    """camera: mapping with position, direction_x/y/z, sensor_distance,
    sensor_half_width, sensor_half_height, focus_distance, aperture_radius
    (the fields src/renderer.cpp:2082-2094 copies out of Camera)."""
    p = PushConstants()
    p.global_frame_count = frame
    p.sample_count = sample_count
    p.samples_per_frame = samples_per_frame
    p.camera_position[:] = [float(v) for v in camera["position"]]
    p.camera_dir_x[:] = [float(v) for v in camera["direction_x"]]
    p.camera_dir_y[:] = [float(v) for v in camera["direction_y"]]
    p.camera_dir_z[:] = [float(v) for v in camera["direction_z"]]
    p.sensor_distance = camera["sensor_distance"]
    p.sensor_half_width = camera["sensor_half_width"]
    p.sensor_half_height = camera["sensor_half_height"]
    p.focus_distance = camera["focus_distance"]
    p.aperture_radius = camera["aperture_radius"]
    return p


def camera_ray(push: PushConstants, width, height, px, py, rng_state):
    s = C.c_uint32(rng_state)
    o = np.empty(3, np.float32)
    d = np.empty(3, np.float32)
    lib().pt_camera_ray(C.byref(push), width, height, px, py, C.byref(s), _ptr(o), _ptr(d))
    return o, d, s.value


def offset_position_along_normal(p, n):
    p = _f32(p)
    n = _f32(n)
    out = np.empty(3, np.float32)
    lib().pt_offset_position_along_normal(_ptr(p), _ptr(n), _ptr(out))
    return out


def material_scatter(kind, ray_dir, pos, nrm, rng_state):
    ray_dir, pos, nrm = _f32(ray_dir), _f32(pos), _f32(nrm)
    s = C.c_uint32(rng_state)
    o = np.empty(3, np.float32)
    d = np.empty(3, np.float32)
    col = np.empty(3, np.float32)
    em = np.empty(3, np.float32)
    att = C.c_float(0)
    lib().pt_material_scatter(kind, _ptr(ray_dir), _ptr(pos), _ptr(nrm), C.byref(s),
                              _ptr(o), _ptr(d), _ptr(col), _ptr(em), C.byref(att))
    return dict(origin=o, dir=d, color=col, emissivity=em, attenuation=att.value,
                rng=s.value)


TONEMAP_PBR_NEUTRAL, TONEMAP_ACES = 0, 1


def tonemap(rgb, kind=TONEMAP_PBR_NEUTRAL):
    rgb = _f32(rgb)
    out = np.empty(3, np.float32)
    (lib().pt_tonemap_aces if kind == TONEMAP_ACES else lib().pt_tonemap_pbr_neutral)(_ptr(rgb), _ptr(out))
    return out


def unorm8(x: float) -> int:
    return lib().pt_unorm8(x)


def inverse_affine(m):
    m = _f32(m).reshape(12)
    out = np.empty(12, np.float32)
    lib().pt_inverse_affine(_ptr(m), _ptr(out))
    return out


class Scene:
    """meshes: list of (positions f32[V,3], normals f32[V,3], indices u32[F,3]);
    instances: list of (transform 3x4, mesh index, kind); env: f32[H,W,4] or None."""

    def __init__(self, meshes, instances, env=None, two_level=False):
        L = lib()
        self._two_level = two_level
        self._keep = []
        md = (MeshDesc * max(1, len(meshes)))()
        for i, (p, n, idx) in enumerate(meshes):
            p = _f32(p).reshape(-1, 3)
            n = _f32(n).reshape(-1, 3)
            idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)
            self._keep += [p, n, idx]
            md[i].positions = p.ctypes.data
            md[i].normals = n.ctypes.data
            md[i].indices = idx.ctypes.data
            md[i].n_vertices = p.shape[0]
            md[i].n_triangles = idx.shape[0]
        idesc = (InstanceDesc * max(1, len(instances)))()
        for i, (xf, mesh, kind) in enumerate(instances):
            idesc[i].transform[:] = [float(v) for v in _f32(xf).reshape(12)]
            idesc[i].mesh = mesh
            idesc[i].kind = kind
        ew = eh = 0
        eptr = None
        if env is not None:
            env = _f32(env)
            eh, ew = env.shape[0], env.shape[1]
            assert env.shape[2] == 4
            self._keep.append(env)
            eptr = env.ctypes.data
        self.handle = L.pt_scene_create(md, len(meshes), idesc, len(instances), eptr, ew, eh)
        self.n_triangles = L.pt_scene_triangle_count(self.handle)
        if two_level:
            L.pt_scene_set_two_level(self.handle, 1)

    def __del__(self):
        if getattr(self, "handle", None) and lib is not None:  # (module globals are gone at interpreter exit)
            lib().pt_scene_destroy(self.handle)
            self.handle = None

    def triangle_ids(self, flat: int):
        a, b = C.c_uint32(), C.c_uint32()
        lib().pt_scene_triangle_ids(self.handle, flat, C.byref(a), C.byref(b))
        return a.value, b.value

    def world_triangles(self):
        out = np.empty((self.n_triangles, 9), np.float32)
        lib().pt_scene_world_triangles(self.handle, _ptr(out))
        return out

    def sah_cost(self) -> float:
        return lib().pt_scene_bvh_sah_cost(self.handle)

    def wide_sah_cost(self):
        """(cost, wide nodes) of the 8-wide collapse of a top-down binned-SAH tree of
        this scene: the yardstick for the shipped library's ptc_scene_info.sah_cost."""
        n = C.c_uint64(0)
        cost = lib().pt_scene_wide_sah_cost(self.handle, C.byref(n))
        return cost, n.value

    def miss(self, d):
        d = _f32(d)
        out = np.empty(3, np.float32)
        lib().pt_miss(self.handle, _ptr(d), _ptr(out))
        return out

    def intersect(self, rays, tmax=10000.0, use_bvh=True):
        rays = _f32(rays).reshape(-1, 6)
        out = np.empty(rays.shape[0], HIT_DTYPE)
        st = Stats()
        lib().pt_intersect(self.handle, _ptr(rays), rays.shape[0], tmax, int(use_bvh),
                           _ptr(out), C.byref(st))
        return out, st

    def render(self, push: PushConstants, width, height, accum=None, n_frames=1,
               max_depth=32, tmax=10000.0, accumulate_sum=False, use_bvh=True,
               n_threads=0, want_rgba8=True, tonemap=TONEMAP_PBR_NEUTRAL, debug_view=0):
        """debug_view=1: every sample adds its bounce count / 16 instead of its
        radiance (the branch under "#else" at shader.rgen:197-198)."""
        rp = RenderParams(width, height, max_depth, tmax, n_frames,
                          int(accumulate_sum), int(use_bvh), n_threads, int(tonemap), int(debug_view))
        if accum is None:
            accum = np.zeros((height, width, 4), np.float32)
        else:
            accum = np.ascontiguousarray(accum, np.float32).copy()
        rgba8 = np.zeros((height, width, 4), np.uint8) if want_rgba8 else None
        st = Stats()
        lib().pt_render(self.handle, C.byref(push), C.byref(rp), _ptr(accum),
                        _ptr(rgba8) if want_rgba8 else None, C.byref(st))
        return accum, rgba8, st

    def render_pixels(self, push: PushConstants, width, height, pixel_indices, n_frames=1, max_depth=32,
                      tmax=10000.0, accumulate_sum=False, use_bvh=True, tonemap=TONEMAP_PBR_NEUTRAL, debug_view=0):
        """The film pt_render() would leave in the chosen pixels (index = py * width + px)
        of a fresh width x height film: (accum f32[n, 4], rgba8 u8[n, 4], stats)."""
        rp = RenderParams(width, height, max_depth, tmax, n_frames, int(accumulate_sum), int(use_bvh), 0,
                          int(tonemap), int(debug_view))
        idx = np.ascontiguousarray(pixel_indices, dtype=np.uint32).ravel()
        accum = np.zeros((idx.size, 4), np.float32)
        rgba8 = np.zeros((idx.size, 4), np.uint8)
        st = Stats()
        lib().pt_render_pixels(self.handle, C.byref(push), C.byref(rp), _ptr(idx), idx.size, _ptr(accum),
                               _ptr(rgba8), C.byref(st))
        return accum, rgba8, st

    def primary_hits(self, push: PushConstants, width, height, tmax=10000.0,
                     use_bvh=True):
        rp = RenderParams(width, height, 32, tmax, 1, 0, int(use_bvh), 0, 0, 0)
        ids = np.empty((height, width), np.uint32)
        lib().pt_primary_hits(self.handle, C.byref(push), C.byref(rp), _ptr(ids))
        return ids


def bvh8_intersect(nodes: np.ndarray, tris: np.ndarray, rays, tmax=10000.0):
    """nodes: uint8[n_nodes, 80]; tris: uint8[n_tris, 48] in the shipped format."""
    rays = _f32(rays).reshape(-1, 6)
    nodes = np.ascontiguousarray(nodes)
    tris = np.ascontiguousarray(tris)
    out = np.empty(rays.shape[0], HIT_DTYPE)
    st = Stats()
    lib().pt_bvh8_intersect(_ptr(nodes), nodes.shape[0], _ptr(tris), tris.shape[0],
                            _ptr(rays), rays.shape[0], tmax, _ptr(out), C.byref(st))
    return out, st


def bvh8_validate(nodes: np.ndarray, tris: np.ndarray, world_tris=None):
    nodes = np.ascontiguousarray(nodes)
    tris = np.ascontiguousarray(tris)
    cost = C.c_double(0)
    wt = None
    if world_tris is not None:
        world_tris = _f32(world_tris)
        wt = _ptr(world_tris)
    rc = lib().pt_bvh8_validate(_ptr(nodes), nodes.shape[0], _ptr(tris), tris.shape[0],
                                wt, C.byref(cost))
    return rc, cost.value


# --- reference camera compiled from /root/reference/src/camera.cpp ----------

_REF_CAM = os.path.join(_HERE, "_ref", "libref_camera.so")


def ref_camera_available() -> bool:
    return os.path.exists(_REF_CAM)


def ref_create_camera(position, target, sensor_distance, shw, shh):
    L = C.CDLL(_REF_CAM)
    L.ref_create_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p]
    p, t = _f32(position), _f32(target)
    out = np.empty(21, np.float32)
    L.ref_create_camera(_ptr(p), _ptr(t), sensor_distance, shw, shh, _ptr(out))
    return out
