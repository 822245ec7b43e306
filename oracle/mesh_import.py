"""TEST INFRASTRUCTURE ONLY. A numpy restatement of the scene-import steps the
reference asks Assimp for (src/application.cpp:150-189), used to check the
product's importer (ptc_mesh_file_open) on small meshes.

Assimp (v5.3.1, CMakeLists.txt:89-92) is a third-party dependency absent from
the reference tree; the steps below follow its published post-processing:
normalise the scene bounding box to [-1, 1] (PreTransformVertices with
AI_CONFIG_PP_PTV_NORMALIZE); triangulate (a quad from its concave corner, if it
has one, larger polygons by ear clipping in the plane of their Newell normal);
drop triangles with coincident corners (FindDegenerates + SortByPType);
per-corner smooth normals = normalised sum of the unit face normals, within the
maximum smoothing angle, of the corners closer than 1e-4 of the mesh's
bounding-box diagonal (GenSmoothNormals, ComputePositionEpsilon); join a corner
with the earliest vertex whose position and normal are both within 1e-5
(JoinIdenticalVertices); between the last two, turn normals and winding over when
the normals point into the mesh (FixInfacingNormals). Where Assimp's own result depends on the order its
containers hand out candidates, sums run in ascending corner order and joins
take the earliest vertex. PARITY UNPINNED: no output of Assimp itself is
available here.

Only tests/ import this module. Neighbour searches are quadratic: small meshes
only."""
import numpy as np

F = np.float32


def normalize_positions(p: np.ndarray) -> np.ndarray:
    p = np.asarray(p, np.float32)
    lo, hi = p.min(axis=0), p.max(axis=0)
    d = (hi - lo).astype(np.float32)
    div = np.float32(d.max() * np.float32(0.5))
    c = (lo + d * np.float32(0.5)).astype(np.float32)
    return ((p - c) / div).astype(np.float32)


def _dot(a, b):
    return F(F(F(a[0] * b[0]) + F(a[1] * b[1])) + F(a[2] * b[2]))


def _normalize_safe(a):
    a = np.asarray(a, np.float32)
    ln = F(np.sqrt(_dot(a, a)))
    return (a / ln).astype(np.float32) if ln > 0 else a


def _area2(a, b, c):
    return (b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1])


def split_polygon(pts, fan=False):
    """Triangles of one polygon as local corner numbers."""
    n = len(pts)
    if n < 3:
        return []

    def fan_from(s):
        return [(s, (s + k - 1) % n, (s + k) % n) for k in range(2, n)]

    if n == 3 or fan:
        return fan_from(0)
    pts = [np.asarray(q, np.float32) for q in pts]
    if n == 4:
        start = 0
        for i in range(4):
            v = pts[i]
            left, diag, right = (_normalize_safe((pts[(i + k) % 4] - v).astype(np.float32)) for k in (3, 2, 1))
            cl = lambda x: F(min(max(x, F(-1.0)), F(1.0)))
            angle = F(F(np.arccos(cl(_dot(left, diag)))) + F(np.arccos(cl(_dot(right, diag)))))
            if angle > F(np.pi):
                start = i
                break
        return fan_from(start)
    nrm = np.zeros(3, np.float64)
    for i in range(n):
        a, b = pts[i].astype(np.float64), pts[(i + 1) % n].astype(np.float64)
        nrm += [(a[1] - b[1]) * (a[2] + b[2]), (a[2] - b[2]) * (a[0] + b[0]), (a[0] - b[0]) * (a[1] + b[1])]
    ax = np.abs(nrm)
    drop = 2 if ax[2] >= ax[0] and ax[2] >= ax[1] else (1 if ax[1] >= ax[0] else 0)
    flat = []
    for q in pts:
        u, v = float(q[(drop + 1) % 3]), float(q[(drop + 2) % 3])
        flat.append((v, u) if nrm[drop] < 0 else (u, v))
    done = [False] * n

    def nxt(i):
        i = (i + 1) % n
        while done[i]:
            i = (i + 1) % n
        return i

    def prv(i):
        i = (i - 1) % n
        while done[i]:
            i = (i - 1) % n
        return i

    out, left, ear = [], n, 0
    while left > 3:
        e, found = ear, False
        for _ in range(left):
            a, b = prv(e), nxt(e)
            if _area2(flat[a], flat[e], flat[b]) > 0:
                inside = False
                for t in range(n):
                    if flat[t] in (flat[a], flat[e], flat[b]):
                        continue
                    if (_area2(flat[a], flat[e], flat[t]) >= 0 and _area2(flat[e], flat[b], flat[t]) >= 0 and
                            _area2(flat[b], flat[a], flat[t]) >= 0):
                        inside = True
                        break
                if not inside:
                    out.append((a, e, b))
                    done[e] = True
                    left -= 1
                    ear = b
                    found = True
                    break
            e = nxt(e)
        if not found:
            return fan_from(0)
    a = nxt(n - 1)
    b = nxt(a)
    out.append((a, b, nxt(b)))
    return out


def fix_infacing_normals(corners, normals):
    """FixInfacingNormals: when the bounding box of (position + normal) is smaller
    than that of the positions the normals point inwards: normals and winding
    are turned over. Planar meshes and inconsistent boxes are left alone."""
    q = (corners + normals).astype(np.float32)
    d0 = (q.max(axis=0) - q.min(axis=0)).astype(np.float32)
    d1 = (corners.max(axis=0) - corners.min(axis=0)).astype(np.float32)
    if any((d0[k] > 0) != (d1[k] > 0) for k in range(3)):
        return corners, normals
    sq = lambda x: F(np.sqrt(F(x)))
    if (d1[0] < F(0.05) * sq(d1[1] * d1[2]) or d1[1] < F(0.05) * sq(d1[2] * d1[0]) or
            d1[2] < F(0.05) * sq(d1[1] * d1[0])):
        return corners, normals
    if abs(F(F(d0[0] * d0[1]) * d0[2])) < abs(F(F(d1[0] * d1[1]) * d1[2])):
        normals = (-normals).astype(np.float32)
        corners, normals = corners.copy(), normals.copy()
        for t in range(0, len(corners), 3):
            corners[[t, t + 2]] = corners[[t + 2, t]]
            normals[[t, t + 2]] = normals[[t + 2, t]]
    return corners, normals


def import_polygons(positions, faces, max_angle_deg=80.0, normalize=True, exact=False, fan=False,
                    position_epsilon_rel=1e-4, join_epsilon=1e-5):
    """positions [V,3]; faces: list of index lists. Returns (positions, normals,
    indices) after the steps above, vertices in first-use order."""
    p = np.asarray(positions, np.float32)
    if normalize:
        p = normalize_positions(p)
    tris = []
    for f in faces:
        for t in split_polygon([p[i] for i in f], fan):
            t = tuple(f[k] for k in t)
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
    d = (corners.max(axis=0) - corners.min(axis=0)).astype(np.float32)
    radius = F(0.0) if exact else F(F(np.sqrt(_dot(d, d))) * F(position_epsilon_rel))
    r2 = F(radius * radius)

    def close(q, r, lim2):
        if (q + F(0.0)).tobytes() == (r + F(0.0)).tobytes():
            return True
        dd = (r - q).astype(np.float32)
        return lim2 > 0 and _dot(dd, dd) < lim2

    normals = np.zeros_like(corners)
    for i in range(len(corners)):
        s = np.zeros(3, np.float32)
        for j in range(len(corners)):
            if not close(corners[i], corners[j], r2):
                continue
            if j == i or np.float32(_dot(face[j], face[i])) >= limit or max_angle_deg >= 175.0:
                s = (s + face[j]).astype(np.float32)
        normals[i] = _normalize_safe(s)
    corners, normals = fix_infacing_normals(corners, normals)
    out_p, out_n, idx = [], [], []
    je = F(0.0) if exact else F(join_epsilon)
    e2 = F(je * je)
    for q, n in zip(corners, normals):
        hit = None
        for v in range(len(out_p)):
            if not close(q, out_p[v], e2):
                continue
            dn = (out_n[v] - n).astype(np.float32)
            same_n = (n + F(0.0)).tobytes() == (out_n[v] + F(0.0)).tobytes() if exact else not (_dot(dn, dn) > e2)
            if same_n:
                hit = v
                break
        if hit is None:
            hit = len(out_p)
            out_p.append(q)
            out_n.append(n)
        idx.append(hit)
    return (np.asarray(out_p, np.float32), np.asarray(out_n, np.float32),
            np.asarray(idx, np.uint32).reshape(-1, 3))
