"""Randomised differential test: random triangle soups (with degenerate,
duplicated and needle triangles), random instance transforms (rotations,
non-uniform and mirroring scales, shears), all four materials and random
cameras, rendered by every integrator and compared bit for bit with the
oracle, whose closest hit is the brute-force definition here (the world-space
one for flattened scenes, the object-space one for two-level scenes)."""
import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _random_mesh(rng, n_tris):
    n_verts = max(3, n_tris)  # shared vertices: triangles meet edge to edge and at points
    pos = rng.uniform(-1.0, 1.0, (n_verts, 3)).astype(np.float32)
    idx = rng.integers(0, n_verts, (n_tris, 3)).astype(np.uint32)
    # local clusters, so that there is something to hit twice
    for t in range(0, n_tris, 3):
        c = rng.uniform(-0.8, 0.8, 3)
        for k in idx[t]:
            pos[k] = (c + rng.normal(0.0, 0.25, 3)).astype(np.float32)
    if n_tris >= 8:
        idx[1] = idx[0]                      # the same triangle twice: a tie in t
        idx[2] = [idx[2][0], idx[2][0], idx[2][1]]  # zero area
        pos[idx[3][1]] = pos[idx[3][0]] + np.float32(1e-6)  # needle
    nrm = rng.normal(0.0, 1.0, (n_verts, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True).astype(np.float32)
    return R.Mesh(pos, nrm.astype(np.float32), idx)


def _random_transform(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    scale = np.diag(rng.uniform(0.3, 1.2, 3) * rng.choice([-1.0, 1.0], 3))  # mirrors included
    shear = np.eye(3)
    shear[0, 1] = rng.uniform(-0.3, 0.3)
    m = q @ scale @ shear
    t = rng.uniform(-0.6, 0.6, 3)
    return np.concatenate([m, t[:, None]], axis=1).astype(np.float32)


def _random_scene(seed):
    rng = np.random.default_rng(seed)
    meshes = [_random_mesh(rng, int(rng.integers(1, 60))) for _ in range(int(rng.integers(1, 4)))]
    instances = [R.Instance(_random_transform(rng), int(rng.integers(0, len(meshes))), int(rng.integers(0, 4)))
                 for _ in range(int(rng.integers(1, 6)))]
    sky = scenes.procedural_sky(32, 16) if seed % 2 else None
    eye = rng.uniform(-1.0, 1.0, 3)
    eye = eye / np.linalg.norm(eye) * rng.uniform(0.2, 4.0)  # inside the geometry as well as outside
    eye[1] *= 0.5  # keep the view direction off the world up vector
    cam = R.create_camera(eye, rng.uniform(-0.2, 0.2, 3), 1.0, 0.7, 0.45)
    return R.Scene(meshes, instances, sky), cam


@pytest.mark.parametrize("two_level", [False, True], ids=["flattened", "two_level"])
@pytest.mark.parametrize("seed", range(12))
def test_random_scene_bit_exact(ctx, oracle, seed, two_level):
    sc, cam = _random_scene(seed)
    w, h = 48, 32
    orc = oracle.Scene(*sc.oracle_args(), two_level=two_level)
    push = oracle.make_push(cam.as_dict(), frame=seed, sample_count=0, samples_per_frame=2)
    ref_acc, ref_img, _ = orc.render(push, w, h, n_frames=3, max_depth=6, use_bvh=False)
    ref_ids = orc.primary_hits(oracle.make_push(cam.as_dict(), frame=seed), w, h, use_bvh=False)
    dev = R.DeviceScene(ctx, sc, two_level=two_level)
    assert np.array_equal(dev.primary_hits(cam, w, h, seed), ref_ids)
    rays = scenes.random_rays(4000, seed=seed + 100)
    hits = dev.trace(rays)
    ref, _ = orc.intersect(rays, use_bvh=False)
    assert np.array_equal(hits["flat"], ref["id"])
    hit = ref["id"] != oracle.NO_HIT
    for k in ("t", "u", "v"):
        assert np.array_equal(_bits(hits[k][hit]), _bits(ref[k][hit]))
    for integrator in (R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM):
        rr = R.RenderResources(ctx, w, h, dev)
        rr.max_depth = 6
        rr.integrator = integrator
        rr.render(cam, frame=seed, samples_per_frame=2, n_frames=3)
        assert np.array_equal(_bits(rr.read_accum()), _bits(ref_acc)), f"seed {seed}, integrator {integrator}"
        assert np.array_equal(rr.read_rgba8(), ref_img)
        rr.close()
    dev.close()
