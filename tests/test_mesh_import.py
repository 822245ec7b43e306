"""Scene import of the C ABI (ptc_mesh_file_open: OBJ / PLY + the
post-processing the reference configures Assimp with,
src/application.cpp:150-189) against the numpy restatement in
oracle/mesh_import.py. Host code only: runs without a GPU."""
import struct

import numpy as np
import pytest

from oracle import mesh_import as M
from path_tracer_amd import renderer as R
from path_tracer_amd import scenes


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def cube():
    p = np.array([[x, y, z] for x in (0, 2) for y in (0, 4) for z in (0, 1)], np.float32) + np.float32(3.0)
    quads = [[0, 1, 3, 2], [4, 6, 7, 5], [0, 4, 5, 1], [2, 3, 7, 6], [0, 2, 6, 4], [1, 5, 7, 3]]
    return p, quads


def write_obj(path, p, faces, normals=None):
    with open(path, "w") as f:
        f.write("# fixture\no thing\n")
        for v in p:
            f.write(f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n")
        if normals is not None:
            for n in normals:
                f.write(f"vn {float(n[0])!r} {float(n[1])!r} {float(n[2])!r}\n")
        for face in faces:
            if normals is None:
                f.write("f " + " ".join(str(i + 1) for i in face) + "\n")
            else:
                f.write("f " + " ".join(f"{i + 1}//{i + 1}" for i in face) + "\n")


def write_ply(path, p, faces, fmt):
    head = (f"ply\nformat {fmt} 1.0\ncomment fixture\nelement vertex {len(p)}\nproperty float x\n"
            f"property float y\nproperty float z\nproperty uchar quality\nelement face {len(faces)}\n"
            "property list uchar int vertex_indices\nend_header\n").encode()
    with open(path, "wb") as f:
        f.write(head)
        if fmt == "ascii":
            for v in p:
                f.write(f"{float(v[0])!r} {float(v[1])!r} {float(v[2])!r} 7\n".encode())
            for face in faces:
                f.write((str(len(face)) + " " + " ".join(map(str, face)) + "\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            for v in p:
                f.write(struct.pack(e + "fffB", *map(float, v), 7))
            for face in faces:
                f.write(struct.pack(e + "B" + "i" * len(face), len(face), *face))


def check_against_restatement(meshes, p, faces, angle=80.0, normalize=True):
    assert len(meshes) == 1
    rp, rn, ri = M.import_polygons(p, faces, angle, normalize)
    m = meshes[0]
    assert np.array_equal(m.indices, ri)
    assert np.array_equal(_bits(m.positions), _bits(rp))
    assert np.array_equal(_bits(m.normals), _bits(rn))


@pytest.mark.parametrize("fmt", ["obj", "ascii", "binary_little_endian", "binary_big_endian"])
def test_cube_keeps_hard_edges(tmp_path, fmt):
    p, quads = cube()
    path = tmp_path / ("cube.obj" if fmt == "obj" else "cube.ply")
    write_obj(path, p, quads) if fmt == "obj" else write_ply(path, p, quads, fmt)
    meshes = R.import_meshes(str(path))
    check_against_restatement(meshes, p, quads)
    m = meshes[0]
    # 90-degree creases exceed the 80-degree limit: 6 faces x 4 corners stay distinct
    assert m.positions.shape == (24, 3) and m.indices.shape == (12, 3)
    # normalised about the centre along the longest axis (y: extent 4 -> [-1, 1])
    assert m.positions.min(axis=0).tolist() == [-0.5, -1.0, -0.25]
    assert m.positions.max(axis=0).tolist() == [0.5, 1.0, 0.25]
    # every normal is an axis direction pointing away from the centre
    assert np.all(np.abs(m.normals).sum(axis=1) == 1.0)
    assert np.all((m.normals * m.positions).sum(axis=1) > 0)


def sphere(n_lon=24, n_lat=12):
    """Latitude-longitude grid with a full ring of coincident vertices at each
    pole, as exporters write it: the quads touching a pole have two equal corners."""
    lat = np.linspace(0.0, np.pi, n_lat + 1)
    lon = np.arange(n_lon) * (2 * np.pi / n_lon)
    p = np.stack([np.outer(np.sin(lat), np.cos(lon)), np.outer(np.cos(lat), np.ones(n_lon)),
                  np.outer(np.sin(lat), np.sin(lon))], axis=-1).astype(np.float32)
    p[0], p[-1] = (0, 1, 0), (0, -1, 0)
    quads = [[i * n_lon + j, i * n_lon + (j + 1) % n_lon, (i + 1) * n_lon + (j + 1) % n_lon, (i + 1) * n_lon + j]
             for i in range(n_lat) for j in range(n_lon)]
    return p.reshape(-1, 3), quads


def test_smooth_surface_joins_vertices(tmp_path):
    p, quads = sphere()
    path = tmp_path / "sphere.ply"
    write_ply(path, p, quads, "binary_little_endian")
    meshes = R.import_meshes(str(path))
    check_against_restatement(meshes, p, quads)
    m = meshes[0]
    n_lon, n_lat = 24, 12
    # one of the two triangles of every pole quad is degenerate and dropped
    assert m.indices.shape[0] == 2 * n_lon * n_lat - 2 * n_lon
    # smooth everywhere: one vertex per distinct position (the pole rings collapse)
    assert m.positions.shape[0] == n_lon * (n_lat - 1) + 2
    assert np.allclose(np.linalg.norm(m.normals, axis=1), 1.0, atol=1e-6)
    assert np.all((m.normals * m.positions).sum(axis=1) > 0.9)  # outward, close to the radius vector
    # a wider limit changes nothing on a smooth surface; a tiny one splits every corner
    assert R.import_meshes(str(path), max_smoothing_angle=175.0)[0].positions.shape[0] == m.positions.shape[0]
    assert R.import_meshes(str(path), max_smoothing_angle=0.01)[0].positions.shape[0] > 3 * m.positions.shape[0]


def test_inward_winding_is_fixed(tmp_path):
    p, quads = sphere()
    path = tmp_path / "inside_out.obj"
    write_obj(path, p, [q[::-1] for q in quads])
    m = R.import_meshes(str(path))[0]
    assert np.all((m.normals * m.positions).sum(axis=1) > 0.9)  # FixInfacingNormals flipped them outwards
    a, b, c = (m.positions[m.indices[:, k]] for k in range(3))
    assert np.all((np.cross(b - a, c - a) * (a + b + c)).sum(axis=1) > 0)  # and the winding with them


def test_file_normals_are_kept_and_options(tmp_path):
    p, quads = cube()
    n = (p - p.mean(axis=0)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    path = tmp_path / "cube_n.obj"
    write_obj(path, p, quads, normals=n)
    m = R.import_meshes(str(path), normalize=False)[0]
    assert m.positions.shape == (8, 3)  # corner normals given per vertex: corners join
    order = [int(np.flatnonzero((p == q).all(axis=1))[0]) for q in m.positions]
    assert np.array_equal(_bits(m.normals), _bits(n[order]))
    assert np.array_equal(_bits(m.positions), _bits(p[order]))  # file coordinates kept
    assert R.import_meshes(str(path), join_identical_vertices=False)[0].positions.shape == (36, 3)


def test_degenerate_faces_polygons_and_groups(tmp_path):
    path = tmp_path / "multi.obj"
    path.write_text(
        "v 0 0 0\nv 1 0 0\nv 1.5 1 0\nv 0.5 2 0\nv -0.5 1 0\nv 0 0 0\n"
        "o first\nusemtl a\nf 1 2 3 4 5\nf 1 6 2\nf -6 -5 -4\n"       # pentagon, degenerate, negative indices
        "usemtl b\nf 1/1 2/2 3/3\nl 1 2\np 3\n")
    meshes = R.import_meshes(str(path), normalize=False)
    assert [m.indices.shape[0] for m in meshes] == [4, 1]
    assert all(np.array_equal(m.normals, np.tile([0, 0, 1], (len(m.normals), 1))) for m in meshes)


@pytest.mark.parametrize("content,name,reason", [
    (b"v 0 0 0\n", "empty.obj", "Scene has no meshes"),
    (b"f 1 2 3\n", "bad.obj", "out of range"),
    (b"solid x\n", "x.stl", "No suitable reader"),
    (b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\n", "cut.ply", "header not terminated"),
    # a face count no file could hold: refused before anything is reserved (no exception leaves the C ABI)
    (b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
     b"element face 99999999999999999999\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n",
     "huge.ply", "element count exceeds the file size"),
    (b"ply\nformat binary_little_endian 1.0\nelement vertex 4000000000\nproperty float x\nend_header\n\0\0\0\0",
     "huge_bin.ply", "element count exceeds the file size"),
    # an element without properties stores nothing, whatever its count claims
    (b"ply\nformat ascii 1.0\nelement pad 18446744073709551615\nelement vertex 1\nproperty float x\nend_header\n0\n",
     "pad.ply", "Scene has no meshes"),
])
def test_import_errors(tmp_path, content, name, reason):
    path = tmp_path / name
    path.write_bytes(content)
    with pytest.raises(R.PtcError, match=reason):
        R.import_meshes(str(path))
    with pytest.raises(R.PtcError, match="Unable to open file"):
        R.import_meshes(str(tmp_path / "missing.obj"))


def test_seam_within_the_position_tolerance_is_smoothed_and_joined(tmp_path):
    """Two halves of a curved sheet whose seam vertices are written twice, a few
    1e-6 apart: within 1e-4 of the bounding-box diagonal the seam's corners
    share their smoothed normals, and within 1e-5 they become one vertex.
    Bit-exact matching leaves the seam split and creased."""
    nu, nv = 9, 7
    u, v = np.meshgrid(np.linspace(-1, 1, nu), np.linspace(-1, 1, nv), indexing="ij")
    grid = np.stack([u, v, 0.3 * np.cos(2 * u) * np.cos(v)], axis=-1).astype(np.float32)
    left = grid[: nu // 2 + 1].reshape(-1, 3)
    right = grid[nu // 2:].reshape(-1, 3).copy()
    right[:nv] += np.array([2e-6, -3e-6, 1e-6], np.float32)  # the seam column again, slightly off
    p = np.concatenate([left, right])
    quad = lambda base, i, j: [base + i * nv + j, base + (i + 1) * nv + j, base + (i + 1) * nv + j + 1, base + i * nv + j + 1]
    faces = [quad(0, i, j) for i in range(nu // 2) for j in range(nv - 1)]
    faces += [quad(len(left), i, j) for i in range(nu - nu // 2 - 1) for j in range(nv - 1)]
    path = tmp_path / "seam.ply"
    write_ply(path, p, faces, "binary_little_endian")
    meshes = R.import_meshes(str(path), normalize=False)
    rp, rn, ri = M.import_polygons(p, faces, 80.0, normalize=False)
    m = meshes[0]
    assert np.array_equal(m.indices, ri)
    assert np.array_equal(_bits(m.positions), _bits(rp)) and np.array_equal(_bits(m.normals), _bits(rn))
    assert m.positions.shape[0] == nu * nv  # the doubled seam column collapsed
    exact = R.import_meshes(str(path), normalize=False, exact_match=True)[0]
    ep, en, ei = M.import_polygons(p, faces, 80.0, normalize=False, exact=True)
    assert np.array_equal(exact.indices, ei) and np.array_equal(_bits(exact.normals), _bits(en))
    assert exact.positions.shape[0] == nu * nv + nv
    # wider tolerances through the options: everything within 0.6 of a corner smooths and joins with it
    wide = R.import_meshes(str(path), normalize=False, position_epsilon_rel=0.2, join_epsilon=0.3)[0]
    wp, wn, wi = M.import_polygons(p, faces, 80.0, normalize=False, position_epsilon_rel=0.2, join_epsilon=0.3)
    assert np.array_equal(wide.indices, wi) and np.array_equal(_bits(wide.normals), _bits(wn))
    assert wide.positions.shape[0] < m.positions.shape[0]


def _polygon_area(p, face):
    q = p[face]
    return 0.5 * np.linalg.norm(sum(np.cross(q[i], q[(i + 1) % len(face)]) for i in range(len(face))))


def test_concave_polygons_are_cut_inside_their_outline(tmp_path):
    """An arrowhead quad, an L-shaped hexagon, a star and a spiral-like comb in
    three different planes: the triangles tile the polygon (areas add up, all
    with one orientation), where a fan from corner 0 would not."""
    arrow = [[0, 0, 0], [2, 1, 0], [0, 2, 0], [0.5, 1, 0]]                      # concave at corner 3
    ell = [[0, 0, 1], [3, 0, 1], [3, 1, 1], [1, 1, 1], [1, 3, 1], [0, 3, 1]]
    ang = np.arange(10) * (np.pi / 5)
    rad = np.where(np.arange(10) % 2 == 0, 2.0, 0.7)
    star = np.stack([np.zeros(10) + 5, rad * np.cos(ang), rad * np.sin(ang)], axis=1)   # in a plane x = 5
    comb = [[0, 9, 0], [7, 9, 0], [7, 9, 3], [6, 9, 3], [6, 9, 1], [4, 9, 1], [4, 9, 3], [3, 9, 3], [3, 9, 1], [1, 9, 1],
            [1, 9, 3], [0, 9, 3]]                                                  # in a plane y = 9, clockwise seen from +y
    p = np.concatenate([np.asarray(a, np.float32) for a in (arrow, ell, star, comb)])
    faces, base = [], 0
    for poly in (arrow, ell, star, comb):
        faces.append(list(range(base, base + len(poly))))
        base += len(poly)
    path = tmp_path / "concave.obj"
    write_obj(path, p, faces)
    meshes = R.import_meshes(str(path), normalize=False, join_identical_vertices=False)
    rp, rn, ri = M.import_polygons(p, faces, 80.0, normalize=False, join_epsilon=0.0, exact=False)
    m = meshes[0]
    assert m.indices.shape[0] == sum(len(f) - 2 for f in faces)
    tri = m.positions[m.indices.reshape(-1)].reshape(-1, 3, 3).astype(np.float64)
    cr = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    first = 0
    for f in faces:
        k = len(f) - 2
        nrm = sum(np.cross(p[f[i]].astype(np.float64), p[f[(i + 1) % len(f)]].astype(np.float64)) for i in range(len(f)))
        c = cr[first:first + k]
        assert np.all(c @ nrm > 0) or np.all(c @ nrm < 0)  # one orientation (FixInfacingNormals may turn the mesh over)
        assert np.isclose(0.5 * np.linalg.norm(c, axis=1).sum(), _polygon_area(p.astype(np.float64), f), rtol=1e-6)
        first += k
    # the same triangles as the restatement (one vertex per corner, in order)
    got = m.positions[m.indices.reshape(-1)]
    want = rp[ri.reshape(-1)]
    assert np.array_equal(_bits(got), _bits(want))
    # a fan from corner 0 leaves the arrowhead's outline: the option brings it back for comparison
    fan = R.import_meshes(str(path), normalize=False, fan=True)[0]
    ftri = fan.positions[fan.indices.reshape(-1)].reshape(-1, 3, 3).astype(np.float64)
    fcr = np.cross(ftri[:, 1] - ftri[:, 0], ftri[:, 2] - ftri[:, 0])
    assert 0.5 * np.linalg.norm(fcr, axis=1).sum() > sum(_polygon_area(p.astype(np.float64), f) for f in faces) * 1.05


def test_binary_ply_of_871k_triangles_round_trips(tmp_path):
    """The headline mesh size through the importer: a binary PLY without normals
    comes back with the same triangles and unit normals close to the
    area-weighted ones, in a time that is printed; without an angle limit there
    is one vertex per position (the limit of 80 degrees splits the mesh's scale
    edges), and bit-exact matching finds the same vertices (no two of its
    positions are closer than the tolerances)."""
    import time
    mesh = scenes.knot_mesh()
    path = tmp_path / "knot.ply"
    nv, nt = mesh.positions.shape[0], mesh.indices.shape[0]
    assert nt == 871402
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {nv}\nproperty float x\nproperty float y\n"
                 f"property float z\nelement face {nt}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(np.ascontiguousarray(mesh.positions, "<f4").tobytes())
        rec = np.empty(nt, np.dtype([("n", "u1"), ("i", "<i4", 3)]))
        rec["n"], rec["i"] = 3, mesh.indices
        f.write(rec.tobytes())
    t0 = time.perf_counter()
    m = R.import_meshes(str(path), normalize=False)[0]
    dt = time.perf_counter() - t0
    print(f"import of {nt:,} triangles / {nv:,} vertices from binary PLY: {dt:.2f} s")
    assert m.indices.shape == (nt, 3)
    assert np.array_equal(_bits(m.positions[m.indices.reshape(-1)]), _bits(mesh.positions[mesh.indices.reshape(-1)]))
    n_positions = np.unique(mesh.positions, axis=0).shape[0]
    assert m.positions.shape[0] >= n_positions
    assert R.import_meshes(str(path), normalize=False, max_smoothing_angle=175.0)[0].positions.shape[0] == n_positions
    exact = R.import_meshes(str(path), normalize=False, exact_match=True)[0]
    assert np.array_equal(exact.indices, m.indices) and np.array_equal(_bits(exact.normals), _bits(m.normals))
    assert np.allclose(np.linalg.norm(m.normals, axis=1), 1.0, atol=1e-5)
    # every corner's normal lies within the 80-degree limit of its own face normal
    tri = mesh.positions[mesh.indices.reshape(-1)].reshape(-1, 3, 3)
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    got = m.normals[m.indices.reshape(-1)].reshape(-1, 3, 3)
    assert np.all((got * fn[:, None, :]).sum(axis=2) > np.cos(np.radians(80.0)) - 1e-4)
    assert dt < 60.0
