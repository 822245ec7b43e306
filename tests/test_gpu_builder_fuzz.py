"""The device BVH builder on triangle distributions chosen to be hard for it
(everything in one place, on a line, on a lattice with ties everywhere, a few
huge triangles among many tiny ones, a small object far from the origin,
needles, one Morton cell): the tree must validate against the oracle's
structural check (every triangle present once, child boxes containing their
subtrees after the exact dequantisation the traversal uses) and random rays
must hit exactly what the oracle's brute force finds, with both scene kinds."""
import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu


def _soup(p):
    """p: (n, 3, 3) corner positions -> a mesh with flat normals."""
    p = np.ascontiguousarray(p, np.float32)
    n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(ln > 0, n / np.maximum(ln, 1e-30), np.array([[0, 0, 1]], np.float32)).astype(np.float32)
    return R.Mesh(p.reshape(-1, 3), np.repeat(n, 3, axis=0), np.arange(3 * len(p), dtype=np.uint32).reshape(-1, 3))


def _shapes(rng, n):
    tiny = lambda c, s: c[:, None, :] + rng.normal(0, s, (len(c), 3, 3))
    yield "identical", np.tile(np.array([[[0, 0, 0], [0.3, 0, 0], [0, 0.3, 0]]], np.float32), (n, 1, 1))
    yield "one point", tiny(np.zeros((n, 3)), 1e-7)
    t = rng.uniform(-1, 1, n)
    yield "on a line", tiny(np.stack([t, 0.5 * t, -0.25 * t], 1), 1e-3)
    g = np.stack(np.meshgrid(*[np.arange(int(round(n ** (1 / 3))) + 1)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
    quad = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    yield "lattice", (g[:, None, :] + quad[None]) * 0.05 - 0.4
    big = rng.uniform(-1, 1, (8, 3, 3)) * 3.0
    yield "huge among tiny", np.concatenate([big, tiny(rng.uniform(-0.5, 0.5, (n - 8, 3)), 2e-3)])
    yield "far from the origin", tiny(rng.uniform(-0.01, 0.01, (n, 3)), 1e-3) + np.array([900.0, -700.0, 500.0])
    a = rng.uniform(-1, 1, (n, 3))
    d = rng.normal(0, 1, (n, 3))
    yield "needles", np.stack([a, a + d, a + d * (1 + 1e-4) + rng.normal(0, 1e-5, (n, 3))], 1)
    yield "two clusters", np.concatenate([tiny(np.full((n // 2, 3), -0.9), 1e-4), tiny(np.full((n - n // 2, 3), 0.9), 1e-4)])
    yield "uniform", tiny(rng.uniform(-1, 1, (n, 3)), 0.02)


@pytest.mark.parametrize("n", [9, 130, 1500, 20000])
def test_builder_on_adversarial_distributions(ctx, oracle, n):
    rng = np.random.default_rng(n)
    for name, p in _shapes(rng, n):
        mesh = _soup(p)
        inst = [R.Instance(R.IDENTITY, 0, 1)]
        if n <= 1500:  # a second, transformed instance of the same mesh
            inst.append(R.Instance(np.array([[0.5, 0.1, 0, 0.3], [0, -0.7, 0, -0.2], [0.2, 0, 0.9, 0.1]], np.float32), 0, 0))
        sc = R.Scene([mesh], inst, None)
        # (a two-level scene intersects in the object space of each instance: its own hit definition)
        orcs = {False: oracle.Scene(*sc.oracle_args()), True: oracle.Scene(*sc.oracle_args(), two_level=True)}
        lo, hi = p.reshape(-1, 3).min(0), p.reshape(-1, 3).max(0)
        centre, radius = (lo + hi) / 2, float(np.linalg.norm(hi - lo)) / 2 + 1e-3
        rays = scenes.random_rays(3000, seed=n)
        rays[:, :3] = centre + rays[:, :3] * radius            # origins around and inside the geometry
        for two_level in (False, True):
            dev = R.DeviceScene(ctx, sc, two_level=two_level)
            if not two_level:
                nodes, tris = dev.read_bvh()
                rc, _ = oracle.bvh8_validate(nodes, tris, orcs[False].world_triangles())
                assert rc == 0, (name, n, rc)
            hits = dev.trace(rays)
            ref, _ = orcs[two_level].intersect(rays, use_bvh=n > 1500)  # brute force is the definition; the oracle's own tree for size
            assert np.array_equal(hits["flat"], ref["id"]), (name, n, two_level)
            same = ref["id"] != oracle.NO_HIT
            for f in ("t", "u", "v"):
                assert np.array_equal(hits[f][same].view(np.uint32), ref[f][same].view(np.uint32)), (name, n, two_level, f)
            dev.close()


def test_reinsertion_rounds_keep_the_tree_valid_and_lower_its_cost(ctx, oracle):
    """The builder's reinsertion step (subtrees moved to where they add the least
    surface area, several rounds): whatever the number of rounds, every triangle
    stays in the tree once, boxes contain their subtrees, rays hit what brute
    force finds; the cost falls with the rounds; two builds give the same bytes."""
    rng = np.random.default_rng(77)
    shapes = dict(_shapes(rng, 20000))
    cases = [("uniform", R.Scene([_soup(shapes["uniform"])], [R.Instance(R.IDENTITY, 0, 0)], None)),
             ("huge among tiny", R.Scene([_soup(shapes["huge among tiny"])], [R.Instance(R.IDENTITY, 0, 1)], None)),
             ("mesh in a box", scenes.dragon_in_cornell(nu=116, nv=59))]
    for name, sc in cases:
        orc = oracle.Scene(*sc.oracle_args())
        world = orc.world_triangles()
        rays = scenes.random_rays(20000, seed=5)
        ref, _ = orc.intersect(rays, use_bvh=True)
        hit = ref["id"] != oracle.NO_HIT
        costs = {}
        for rounds in (R.OPTIMIZE_NONE, 1, 0, 40):
            dev = R.DeviceScene(ctx, sc, optimize_rounds=rounds)
            nodes, tris = dev.read_bvh()
            rc, _ = oracle.bvh8_validate(nodes, tris, world)
            assert rc == 0, (name, rounds, rc)
            hits = dev.trace(rays)
            assert np.array_equal(hits["flat"], ref["id"]), (name, rounds)
            for f in "tuv":
                assert np.array_equal(hits[f][hit].view(np.uint32), ref[f][hit].view(np.uint32)), (name, rounds, f)
            costs[rounds] = dev.info.sah_cost
            again = R.DeviceScene(ctx, sc, optimize_rounds=rounds)
            nodes2, tris2 = again.read_bvh()
            assert np.array_equal(nodes, nodes2) and np.array_equal(tris, tris2), (name, rounds)
            again.close()
            dev.close()
        print(f"{name}: SAH cost without / 1 / 8 / 40 rounds: {costs[R.OPTIMIZE_NONE]:.2f} {costs[1]:.2f} {costs[0]:.2f} {costs[40]:.2f}")
        assert costs[1] < costs[R.OPTIMIZE_NONE] and costs[0] < costs[1] and costs[40] <= costs[0] * 1.001, (name, costs)


def test_two_level_trees_are_reinserted_too(ctx, oracle):
    sc = scenes.ref4_scene(scenes.knot_mesh(nu=116, nv=59))
    a = R.DeviceScene(ctx, sc, two_level=True, optimize_rounds=R.OPTIMIZE_NONE)
    b = R.DeviceScene(ctx, sc, two_level=True)
    assert b.info.sah_cost < a.info.sah_cost
    rays = scenes.random_rays(20000, seed=6)
    ref, _ = oracle.Scene(*sc.oracle_args(), two_level=True).intersect(rays, use_bvh=True)
    assert np.array_equal(b.trace(rays)["flat"], ref["id"])
    a.close()
    b.close()
