"""TEST INFRASTRUCTURE ONLY. A numpy restatement of the scene-import steps the
reference asks Assimp for (src/application.cpp:150-189), used to check the
product's importer (ptc_mesh_file_open) on small meshes.

Assimp (v5.3.1, CMakeLists.txt:89-92) is a third-party dependency absent from
the reference tree; the steps below follow its published post-processing:
normalise the scene bounding box to [-1, 1] (PreTransformVertices with
AI_CONFIG_PP_PTV_NORMALIZE), fan-triangulate, drop triangles with coincident
corners (FindDegenerates + SortByPType), per-corner smooth normals = normalised
sum of the unit face normals of the corners at the same position within the
maximum smoothing angle (GenSmoothNormals), join identical (position, normal)
vertices. PARITY UNPINNED: no output of Assimp itself is available here.

Only tests/ import this module."""
import numpy as np


def normalize_positions(p: np.ndarray) -> np.ndarray:
    p = np.asarray(p, np.float32)
    lo, hi = p.min(axis=0), p.max(axis=0)
    d = (hi - lo).astype(np.float32)
    div = np.float32(d.max() * np.float32(0.5))
    c = (lo + d * np.float32(0.5)).astype(np.float32)
    return ((p - c) / div).astype(np.float32)


def import_polygons(positions, faces, max_angle_deg=80.0, normalize=True):
    """positions [V,3]; faces: list of index lists. Returns (positions, normals,
    indices) after the steps above, vertices in first-use order."""
    p = np.asarray(positions, np.float32)
    if normalize:
        p = normalize_positions(p)
    tris = []
    for f in faces:
        for k in range(2, len(f)):
            t = (f[0], f[k - 1], f[k])
            a, b, c = p[t[0]], p[t[1]], p[t[2]]
            if (a == b).all() or (b == c).all() or (a == c).all():
                continue
            tris.append(t)
    tris = np.asarray(tris, np.int64).reshape(-1, 3)
    corners = p[tris.reshape(-1)]                                  # verbose vertices
    a, b, c = corners[0::3], corners[1::3], corners[2::3]
    fn = np.cross((b - a).astype(np.float32), (c - a).astype(np.float32)).astype(np.float32)
    ln = np.sqrt((fn * fn).sum(axis=1, dtype=np.float32), dtype=np.float32)
    fn = np.where(ln[:, None] > 0, fn / np.where(ln > 0, ln, 1)[:, None], fn).astype(np.float32)
    face = np.repeat(fn, 3, axis=0)
    limit = np.float32(np.cos(np.float32(max_angle_deg) * np.float32(np.pi) / np.float32(180.0)))
    keys = {}
    for i, q in enumerate(corners):
        keys.setdefault((q + np.float32(0.0)).tobytes(), []).append(i)
    normals = np.zeros_like(corners)
    for i in range(len(corners)):
        s = np.zeros(3, np.float32)
        # same visiting order as a last-in-first-out chain per position
        for j in reversed(keys[(corners[i] + np.float32(0.0)).tobytes()]):
            if j == i or np.float32(np.dot(face[j], face[i])) >= limit or max_angle_deg >= 175.0:
                s = (s + face[j]).astype(np.float32)
        l = np.sqrt(np.float32((s * s).sum(dtype=np.float32)))
        normals[i] = s / l if l > 0 else s
    seen, out_p, out_n, idx = {}, [], [], []
    for q, n in zip(corners, normals):
        k = (q + np.float32(0.0)).tobytes() + (n + np.float32(0.0)).tobytes()
        if k not in seen:
            seen[k] = len(out_p)
            out_p.append(q)
            out_n.append(n)
        idx.append(seen[k])
    return (np.asarray(out_p, np.float32), np.asarray(out_n, np.float32),
            np.asarray(idx, np.uint32).reshape(-1, 3))
