"""Procedural scenes standing in for the assets the reference loads via Assimp.

Neither the Cornell box, the Stanford dragon nor powerplant.hdr ship with the
reference (src/renderer.cpp:1842 reads the sky from a hard-coded path outside
the repository), so every scene here is generated: same array shapes the
reference uploads (float32 positions / normals per vertex, uint32 indices,
src/renderer.cpp:1818-1838), geometry normalised to [-1, 1] the way its import
flag AI_CONFIG_PP_PTV_NORMALIZE does (src/application.cpp:161).
"""
from __future__ import annotations

import numpy as np

from .renderer import (IDENTITY, KIND_DIELECTRIC, KIND_DIFFUSE, KIND_EMISSIVE, KIND_SPECULAR,
                       Instance, Mesh, Scene)


def _quad_mesh(quads) -> Mesh:
    """quads: list of 4 corner points; two triangles and a flat normal each."""
    pos, nrm, idx = [], [], []
    for q in quads:
        q = np.asarray(q, dtype=np.float64)
        n = np.cross(q[1] - q[0], q[2] - q[0])
        n = n / np.linalg.norm(n)
        b = len(pos)
        pos += [q[0], q[1], q[2], q[3]]
        nrm += [n, n, n, n]
        idx += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    return Mesh(np.asarray(pos, np.float32), np.asarray(nrm, np.float32), np.asarray(idx, np.uint32))


def _cornell_xform(p):
    """Published Cornell-box millimetres -> [-1, 1], opening towards +z so the
    reference's default camera at (0, 0, 3.5) looks into the box."""
    p = np.asarray(p, dtype=np.float64)
    return np.stack([(p[..., 0] - 278.0) / 280.0, (p[..., 1] - 274.4) / 280.0,
                     -(p[..., 2] - 279.6) / 280.0], axis=-1)


def cornell_box(tall_kind: int = KIND_DIFFUSE, short_kind: int = KIND_DIFFUSE,
                envmap: np.ndarray | None = None) -> Scene:
    """The classic Cornell box (32 triangles) from the published measurement
    data: five walls, an emissive quad just under the ceiling, a short and a
    tall block. One mesh per part so each part carries its own material kind."""
    X = _cornell_xform
    walls = _quad_mesh([
        X([[552.8, 0, 0], [0, 0, 0], [0, 0, 559.2], [549.6, 0, 559.2]]),          # floor
        X([[556, 548.8, 0], [556, 548.8, 559.2], [0, 548.8, 559.2], [0, 548.8, 0]]),  # ceiling
        X([[549.6, 0, 559.2], [0, 0, 559.2], [0, 548.8, 559.2], [556, 548.8, 559.2]]),  # back
        X([[0, 0, 559.2], [0, 0, 0], [0, 548.8, 0], [0, 548.8, 559.2]]),          # right
        X([[552.8, 0, 0], [549.6, 0, 559.2], [556, 548.8, 559.2], [556, 548.8, 0]]),  # left
    ])
    light = _quad_mesh([X([[343, 548.0, 227], [343, 548.0, 332], [213, 548.0, 332], [213, 548.0, 227]])])
    short = _quad_mesh([
        X([[130, 165, 65], [82, 165, 225], [240, 165, 272], [290, 165, 114]]),
        X([[290, 0, 114], [290, 165, 114], [240, 165, 272], [240, 0, 272]]),
        X([[130, 0, 65], [130, 165, 65], [290, 165, 114], [290, 0, 114]]),
        X([[82, 0, 225], [82, 165, 225], [130, 165, 65], [130, 0, 65]]),
        X([[240, 0, 272], [240, 165, 272], [82, 165, 225], [82, 0, 225]]),
    ])
    tall = _quad_mesh([
        X([[423, 330, 247], [265, 330, 296], [314, 330, 456], [472, 330, 406]]),
        X([[423, 0, 247], [423, 330, 247], [472, 330, 406], [472, 0, 406]]),
        X([[472, 0, 406], [472, 330, 406], [314, 330, 456], [314, 0, 456]]),
        X([[314, 0, 456], [314, 330, 456], [265, 330, 296], [265, 0, 296]]),
        X([[265, 0, 296], [265, 330, 296], [423, 330, 247], [423, 0, 247]]),
    ])
    return Scene(
        meshes=[walls, light, short, tall],
        instances=[Instance(IDENTITY, 0, KIND_DIFFUSE), Instance(IDENTITY, 1, KIND_EMISSIVE),
                   Instance(IDENTITY, 2, short_kind), Instance(IDENTITY, 3, tall_kind)],
        envmap=envmap)


def _vertex_normals(pos: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """Area-weighted smooth vertex normals (what aiProcess_GenSmoothNormals
    yields on a smooth closed surface)."""
    p = pos.astype(np.float64)
    fn = np.cross(p[idx[:, 1]] - p[idx[:, 0]], p[idx[:, 2]] - p[idx[:, 0]])
    n = np.zeros_like(p)
    for k in range(3):
        np.add.at(n, idx[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    ln[ln == 0] = 1.0
    return (n / ln).astype(np.float32)


def _normalise_unit(pos: np.ndarray) -> np.ndarray:
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    c = 0.5 * (lo + hi)
    s = 0.5 * float((hi - lo).max())
    return ((pos - c) / s).astype(np.float32)


def _grid_indices(nu: int, nv: int) -> np.ndarray:
    """Triangles of a (nu x nv) grid periodic in both directions."""
    u = np.arange(nu, dtype=np.int64)[:, None]
    v = np.arange(nv, dtype=np.int64)[None, :]
    a = u * nv + v
    b = ((u + 1) % nu) * nv + v
    c = ((u + 1) % nu) * nv + (v + 1) % nv
    d = u * nv + (v + 1) % nv
    t1 = np.stack([a, b, c], axis=-1).reshape(-1, 3)
    t2 = np.stack([a, c, d], axis=-1).reshape(-1, 3)
    out = np.empty((t1.shape[0] * 2, 3), np.uint32)
    out[0::2] = t1
    out[1::2] = t2
    return out


def _bump_field(u, v, seed: int, octaves: int, base: float):
    """Deterministic multi-octave sinusoid field on the (u, v) torus."""
    rng = np.random.default_rng(seed)
    f = np.zeros(np.broadcast(u, v).shape, dtype=np.float64)
    amp = 1.0
    for o in range(octaves):
        ku = int(base * 2 ** o) + int(rng.integers(0, 3))
        kv = int(base * 2 ** o / 2) + int(rng.integers(1, 3))
        ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
        f += amp * np.sin(ku * u + ph1) * np.cos(kv * v + ph2)
        amp *= 0.55
    return f


def knot_mesh(nu: int = 929, nv: int = 469, seed: int = 7) -> Mesh:
    """Stand-in for the Stanford dragon (871,414 triangles): a (2,3) torus knot
    tube with scale-like bumps, 2 * nu * nv triangles (871,402 by default),
    smooth normals, normalised to [-1, 1]."""
    u = np.linspace(0.0, 2 * np.pi, nu, endpoint=False)[:, None]
    v = np.linspace(0.0, 2 * np.pi, nv, endpoint=False)[None, :]
    p, q = 2.0, 3.0
    r = 1.0 + 0.45 * np.cos(q * u)
    cx, cy, cz = r * np.cos(p * u), r * np.sin(p * u), 0.45 * np.sin(q * u)
    # Frenet-like frame from finite differences of the centre line
    du = 1e-4
    r2 = 1.0 + 0.45 * np.cos(q * (u + du))
    tx, ty, tz = (r2 * np.cos(p * (u + du)) - cx), (r2 * np.sin(p * (u + du)) - cy), \
        (0.45 * np.sin(q * (u + du)) - cz)
    t = np.concatenate([tx, ty, tz], axis=1)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    n1 = np.cross(t, up)
    n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
    n2 = np.cross(t, n1)
    rad = 0.22 * (1.0 + 0.18 * _bump_field(u, v, seed, 5, 24.0))
    c = np.concatenate([cx, cy, cz], axis=1)[:, None, :]
    pos = c + rad[..., None] * (np.cos(v)[..., None] * n1[:, None, :] + np.sin(v)[..., None] * n2[:, None, :])
    pos = _normalise_unit(pos.reshape(-1, 3))
    idx = _grid_indices(nu, nv)
    return Mesh(pos, _vertex_normals(pos, idx), idx)


def torus_mesh(nu: int, nv: int, seed: int = 11, roughness: float = 0.25) -> Mesh:
    """Closed fractal-displaced torus, 2 * nu * nv triangles; the 10 M-triangle
    configuration uses nu = 3162, nv = 1581."""
    u = np.linspace(0.0, 2 * np.pi, nu, endpoint=False, dtype=np.float64)[:, None]
    v = np.linspace(0.0, 2 * np.pi, nv, endpoint=False, dtype=np.float64)[None, :]
    rad = 0.35 * (1.0 + roughness * _bump_field(u, v, seed, 7, 6.0))
    x = (1.0 + rad * np.cos(v)) * np.cos(u)
    y = rad * np.sin(v) + 0.0 * u
    z = (1.0 + rad * np.cos(v)) * np.sin(u)
    pos = _normalise_unit(np.stack([x, y, z], axis=-1).reshape(-1, 3))
    idx = _grid_indices(nu, nv)
    return Mesh(pos, _vertex_normals(pos, idx), idx)


def procedural_sky(width: int = 512, height: int = 256) -> np.ndarray:
    """Equirectangular RGBA32F sky: horizon-to-zenith gradient, a bright sun
    disc and a dim ground, values well above 1 near the sun like an HDR map."""
    v = (np.arange(height, dtype=np.float64) + 0.5) / height  # 0 = +y (up)
    u = (np.arange(width, dtype=np.float64) + 0.5) / width
    inc = v[:, None] * np.pi
    az = u[None, :] * 2 * np.pi - np.pi
    d = np.stack([np.sin(inc) * np.cos(az), np.cos(inc) + 0 * az, np.sin(inc) * np.sin(az)], axis=-1)
    up = np.clip(d[..., 1], 0.0, 1.0)
    sky = np.stack([0.35 + 0.25 * (1 - up), 0.5 + 0.2 * (1 - up), 0.9 - 0.1 * up], axis=-1)
    ground = np.array([0.12, 0.10, 0.08])
    img = np.where(d[..., 1:2] >= 0.0, sky, ground)
    sun_dir = np.array([0.5, 0.7, 0.5])
    sun_dir /= np.linalg.norm(sun_dir)
    cs = np.clip((d * sun_dir).sum(-1), 0.0, 1.0)
    img = img + 40.0 * np.power(cs, 800.0)[..., None] * np.array([1.0, 0.9, 0.7])
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = img.astype(np.float32)
    return out


def dragon_scene(kind: int = KIND_SPECULAR, envmap: np.ndarray | None = None,
                 nu: int = 929, nv: int = 469) -> Scene:
    """One metallic instance of the dragon stand-in under a sky."""
    return Scene(meshes=[knot_mesh(nu, nv)], instances=[Instance(IDENTITY, 0, kind)], envmap=envmap)


def ref4_scene(mesh: Mesh, envmap: np.ndarray | None = None) -> Scene:
    """The reference's own hard-coded scene: four instances of mesh 0, one per
    material kind (src/renderer.cpp:1147-1174)."""
    xf = [
        [[1, 0, 0, -1], [0, 1, 0, 0], [0, 0, 1, 0]],
        [[0.7, 0, 0, 1], [0, 0.7, 0, 0], [0, 0, 0.7, 0]],
        [[0.2, 0, 0, 0], [0, 0.2, 0, 0], [0, 0, 0.2, 0]],
        [[0.8, 0, 0, 0], [0, 0.8, 0, 0], [0, 0, 0.8, 0.8]],
    ]
    return Scene(meshes=[mesh],
                 instances=[Instance(np.asarray(m, np.float32), 0, i % 4) for i, m in enumerate(xf)],
                 envmap=envmap)


def write_hdr(file_name: str, rgb: np.ndarray) -> None:
    """Writes float32 [H, W, 3+] as a Radiance .hdr file (RGBE, flat scanlines):
    the file format the reference reads its sky from (src/utility.cpp:75-92)."""
    rgb = np.asarray(rgb, np.float32)[..., :3]
    h, w = rgb.shape[:2]
    m = rgb.max(axis=-1)
    mant, expo = np.frexp(np.where(m > 1e-32, m, 1.0))  # m = mant * 2^expo, mant in [0.5, 1)
    scale = np.where(m > 1e-32, mant * 256.0 / np.where(m > 1e-32, m, 1.0), 0.0)
    out = np.zeros((h, w, 4), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0, 255).astype(np.uint8)
    out[..., 3] = np.where(m > 1e-32, expo + 128, 0).astype(np.uint8)
    with open(file_name, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(out.tobytes())


def c5_sky_from_hdr(width: int = 2048, height: int = 1024) -> np.ndarray:
    """The 4K configuration's environment map the way the reference gets its
    own: from a Radiance .hdr file on disk through the library's decoder
    (read_hdr_image, src/utility.cpp:75-92). The file is the procedural sky,
    written to a temporary directory first; RGBE keeps 8 bits of mantissa, so
    the array returned is the quantised sky, [H, W, 4] float32."""
    import os
    import tempfile
    from .renderer import read_hdr_image
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sky.hdr")
        write_hdr(path, procedural_sky(width, height))
        return read_hdr_image(path)


def dragon_in_cornell(envmap: np.ndarray | None = None, nu: int = 929, nv: int = 469) -> Scene:
    """Dragon + Cornell box under an environment map: the box's five walls and
    its light, with the dragon stand-in (specular, 2 * nu * nv triangles) in the
    place of the two blocks. The box opens towards the camera, so mirror
    reflections leave through the opening and end on the sky."""
    box = cornell_box()
    s = 0.7
    xf = np.array([[s, 0, 0, 0.0], [0, s, 0, -0.25], [0, 0, s, 0.1]], np.float32)
    return Scene(meshes=[box.meshes[0], box.meshes[1], knot_mesh(nu, nv)],
                 instances=[Instance(IDENTITY, 0, KIND_DIFFUSE), Instance(IDENTITY, 1, KIND_EMISSIVE),
                            Instance(xf, 2, KIND_SPECULAR)],
                 envmap=envmap)


def random_rays(n: int, seed: int = 1, radius: float = 3.0, target_radius: float = 1.0) -> np.ndarray:
    """Incoherent rays: origins on a sphere, aimed at random points near the
    origin; float32 [n, 6]."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = radius * o / np.linalg.norm(o, axis=1, keepdims=True)
    t = rng.uniform(-target_radius, target_radius, size=(n, 3))
    d = t - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


__all__ = ["cornell_box", "knot_mesh", "torus_mesh", "procedural_sky", "dragon_scene",
           "dragon_in_cornell", "c5_sky_from_hdr", "write_hdr", "ref4_scene", "random_rays", "KIND_DIFFUSE", "KIND_SPECULAR", "KIND_EMISSIVE",
           "KIND_DIELECTRIC"]
