"""CPU tests of the oracle: golden vectors, known answers, and the quirks of
the reference that the restatement must keep."""
import json
import os

import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def unhx(lst):
    return np.array([int(x, 16) for x in lst], dtype=np.uint32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).ravel().view(np.uint32)


def test_xorshift32_known_answer(oracle):
    # Marsaglia's xorshift32 (13, 17, 5) from seed 2463534242: first output 723471715
    _, states = oracle.random_stream(2463534242, 3)
    assert int(states[0]) == 723471715
    # zero is a fixed point: random() stays 0 (a quirk the restatement keeps)
    v, st = oracle.random_stream(0, 4)
    assert not st.any() and not v.any()


def _xorshift_prev(s):
    def inv_left(x, k):
        r = x
        for _ in range(32 // k + 1):
            r = x ^ ((r << k) & 0xFFFFFFFF)
        return r

    def inv_right(x, k):
        r = x
        for _ in range(32 // k + 1):
            r = x ^ (r >> k)
        return r
    return inv_left(inv_right(inv_left(s, 5), 17), 13)


def test_random_can_return_one(oracle):
    # uint -> float rounds to nearest even: states >= 0xffffff80 convert to 2^32
    s0 = _xorshift_prev(0xFFFFFFFF)
    v, st = oracle.random_stream(s0, 1)
    assert int(st[0]) == 0xFFFFFFFF and v[0] == np.float32(1.0)
    s1 = _xorshift_prev(0xFFFFFF7F)
    v, _ = oracle.random_stream(s1, 1)
    assert v[0] < np.float32(1.0)


def test_rng_golden(oracle):
    g = gold("rng.json")
    for k, v in g["hash"].items():
        assert oracle.hash32(int(k)) == v
    for p, f, s in g["seed"]:
        assert oracle.seed(p, f) == s
    for seed, d in g["streams"].items():
        v, st = oracle.random_stream(int(seed), 8)
        assert np.array_equal(bits(v), unhx(d["values"])) and [int(x) for x in st] == d["states"]


def test_camera_matches_reference_source(oracle):
    g = gold("camera.json")
    for c in g["cases"]:
        cam = R.create_camera(c["position"], c["target"], c["sensor_distance"], c["shw"], c["shh"])
        assert np.array_equal(bits(cam.as_array()), unhx(c["restated"]))
        if "reference" in c:  # produced by the reference's own camera.cpp
            assert c["reference"] == c["restated"]
        if oracle.ref_camera_available():
            ref = oracle.ref_create_camera(c["position"], c["target"], c["sensor_distance"], c["shw"], c["shh"])
            assert np.array_equal(bits(ref), bits(cam.as_array()))
    # y points down, z forward, x right (src/camera.hpp:11-13)
    cam = R.default_camera(640, 480)
    assert (cam.direction_x.x, cam.direction_y.y, cam.direction_z.z) == (1.0, -1.0, -1.0)
    assert cam.aperture_radius == 0.0 and cam.focus_distance == cam.distance == 3.5


def test_primary_rays_golden(oracle):
    g = gold("primary_rays.json")
    cam = R.Camera.from_buffer_copy(unhx(g["camera"]).tobytes())
    for r in g["rays"]:
        push = oracle.make_push(cam.as_dict(), frame=r["frame"])
        o, d, s = oracle.camera_ray(push, g["width"], g["height"], r["px"], r["py"],
                                    oracle.seed(r["py"] * g["width"] + r["px"], r["frame"]))
        assert np.array_equal(bits(o), unhx(r["origin"])) and np.array_equal(bits(d), unhx(r["dir"]))
        assert s == r["rng_after"]


def test_lens_samples_drawn_even_without_aperture(oracle):
    # four randoms per camera sample even when aperture_radius == 0 (shader.rgen:184)
    cam = R.default_camera(64, 64)
    push = oracle.make_push(cam.as_dict())
    seed = oracle.seed(5, 0)
    _, _, after = oracle.camera_ray(push, 64, 64, 5, 0, seed)
    _, states = oracle.random_stream(seed, 4)
    assert after == int(states[3])


def test_offset_golden_and_branches(oracle):
    for c in gold("offset.json"):
        p = unhx(c["p"]).view(np.float32)
        n = unhx(c["n"]).view(np.float32)
        assert np.array_equal(bits(oracle.offset_position_along_normal(p, n)), unhx(c["out"]))
    # integer path: |p| >= 1/32 moves by int(256 * n) ulps, sign aware
    out = oracle.offset_position_along_normal((1.0, -1.0, 2.0), (1.0, 1.0, -1.0))
    b = bits(out).astype(np.int64) - bits(np.array([1.0, -1.0, 2.0], np.float32)).astype(np.int64)
    assert list(b) == [256, -256, -256]
    # float path near the origin
    out = oracle.offset_position_along_normal((0.01, 0.0, -0.01), (1.0, 0.0, 0.0))
    assert out[0] == np.float32(0.01) + np.float32(1.0 / 65536.0)


def test_materials_golden(oracle):
    for c in gold("materials.json"):
        r = oracle.material_scatter(c["kind"], unhx(c["dir"]).view(np.float32), unhx(c["pos"]).view(np.float32),
                                    unhx(c["nrm"]).view(np.float32), c["seed"])
        assert np.array_equal(bits(r["origin"]), unhx(c["origin"]))
        assert np.array_equal(bits(r["dir"]), unhx(c["out_dir"]))
        assert np.array_equal(bits(r["color"]), unhx(c["color"]))
        assert np.array_equal(bits(r["emissivity"]), unhx(c["emissivity"]))
        assert bits([r["attenuation"]])[0] == int(c["attenuation"], 16) and r["rng"] == c["rng"]


def test_material_semantics(oracle):
    d = np.array([0.0, -1.0, 0.0], np.float32)
    n = np.array([0.0, 1.0, 0.0], np.float32)
    # diffuse: albedo visualises the normal, no emission, two randoms drawn
    r = oracle.material_scatter(0, d, (0, 0, 0), n, 42)
    assert np.allclose(r["color"], (0.5, 1.0, 0.5)) and not r["emissivity"].any()
    _, st = oracle.random_stream(42, 2)
    assert r["rng"] == int(st[1])
    # specular: mirror direction, no randoms
    r = oracle.material_scatter(1, np.array([0.6, -0.8, 0.0], np.float32), (0, 0, 0), n, 42)
    assert np.allclose(r["dir"], (0.6, 0.8, 0.0)) and r["rng"] == 42 and np.allclose(r["color"], 0.75)
    # emissive: (5,5,5) and the path continues
    r = oracle.material_scatter(2, d, (0, 0, 0), n, 42)
    assert np.allclose(r["emissivity"], 5.0) and np.allclose(r["color"], 0.75)
    # dielectric: one random; attenuation is Re/P or Tr/(1-P); normal incidence Re = 0.04
    r = oracle.material_scatter(3, d, (0, 0, 0), n, 42)
    _, st = oracle.random_stream(42, 1)
    assert r["rng"] == int(st[0]) and np.allclose(r["color"], 1.0)
    P = 0.25 + 0.5 * 0.04
    assert np.isclose(r["attenuation"], 0.04 / P, rtol=1e-5) or np.isclose(r["attenuation"], 0.96 / (1 - P), rtol=1e-5)


def test_tonemap_golden_and_unorm8(oracle):
    g = gold("tonemap.json")
    for c in g["tonemap"]:
        t = oracle.tonemap(unhx(c["rgb"]).view(np.float32))
        assert np.array_equal(bits(t), unhx(c["out"]))
        assert [oracle.unorm8(float(x)) for x in t] == c["unorm8"]
        a = oracle.tonemap(unhx(c["rgb"]).view(np.float32), oracle.TONEMAP_ACES)
        assert np.array_equal(a.view(np.uint32), unhx(c["aces"]))
        assert [oracle.unorm8(float(x)) for x in a] == c["aces_unorm8"]
    for hexbits, expect in g["unorm8"]:
        x = np.array([int(hexbits, 16)], np.uint32).view(np.float32)[0]
        assert oracle.unorm8(float(x)) == expect
    # round half to even, clamp, NaN -> 0
    assert oracle.unorm8(0.5 / 255) == 0 and oracle.unorm8(1.5 / 255) == 2 and oracle.unorm8(2.5 / 255) == 2
    assert oracle.unorm8(-3.0) == 0 and oracle.unorm8(7.0) == 255 and oracle.unorm8(float("nan")) == 0
    # below the compression knee the map is x - offset(x)
    assert np.allclose(oracle.tonemap((0.5, 0.5, 0.5)), 0.46)
    # ACES (shader.rgen:107-135) against a float64 evaluation of the same formula
    inp = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
    outp = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    for rgb in [(0.0, 0.0, 0.0), (0.18, 0.18, 0.18), (1.0, 0.5, 0.25), (4.0, 9.0, 0.01), (1e4, 1e4, 1e4)]:
        v = inp @ np.array(rgb)
        want = np.clip(outp @ ((v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.4329510) + 0.238081)), 0, 1)
        assert np.allclose(oracle.tonemap(rgb, oracle.TONEMAP_ACES), want, atol=2e-6)


def test_portable_math_close_to_libm(oracle):
    rng = np.random.default_rng(1)
    u = rng.random(100000, dtype=np.float32)
    theta = (np.float32(6.2831855) * u).astype(np.float32)
    s, c = oracle.math_eval(0, oracle.MATH_PORTABLE, theta)
    assert np.abs(s - np.sin(theta.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(theta.astype(np.float64))).max() < 3e-7
    x = np.maximum(u, np.float32(1.175494351e-38))
    lg = oracle.math_eval(1, oracle.MATH_PORTABLE, x)
    ref = np.log(x.astype(np.float64))
    assert (np.abs(lg - ref) <= 4e-7 * np.maximum(1.0, np.abs(ref))).all()
    assert oracle.math_eval(1, oracle.MATH_PORTABLE, np.array([1.175494351e-38], np.float32))[0] == pytest.approx(-87.33654, rel=1e-6)
    y = (2 * u - 1).astype(np.float32)
    ac = oracle.math_eval(2, oracle.MATH_PORTABLE, y)
    assert np.abs(ac - np.arccos(y.astype(np.float64))).max() < 1e-6
    edge = oracle.math_eval(2, oracle.MATH_PORTABLE, np.array([1.0, -1.0, 1.5, -1.5, np.nan], np.float32))
    assert list(edge[:4]) == [0.0, np.float32(np.pi), 0.0, np.float32(np.pi)] and edge[4] == 0.0
    a = rng.standard_normal(100000).astype(np.float32)
    b = rng.standard_normal(100000).astype(np.float32)
    at = oracle.math_eval(3, oracle.MATH_PORTABLE, a, b)
    assert np.abs(at - np.arctan2(a.astype(np.float64), b.astype(np.float64))).max() < 1e-6
    assert oracle.math_eval(3, oracle.MATH_PORTABLE, np.zeros(1, np.float32), np.zeros(1, np.float32))[0] == 0.0


def test_brute_force_hits_golden(oracle):
    g = gold("cornell_hits.json")
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    orc = oracle.Scene(*sc.oracle_args())
    rays = unhx(g["rays"]).view(np.float32).reshape(-1, 6)
    for use_bvh in (False, True):
        h, _ = orc.intersect(rays, use_bvh=use_bvh)
        assert [int(x) for x in h["id"]] == g["id"]
        hit = h["id"] != oracle.NO_HIT
        for k in "tuv":
            assert np.array_equal(bits(h[k])[hit], unhx(g[k])[hit])


@pytest.mark.parametrize("mk", ["cornell", "ref4"])
def test_oracle_bvh_equals_brute_force(oracle, mk):
    if mk == "cornell":
        sc = scenes.cornell_box()
    else:
        sc = scenes.ref4_scene(scenes.knot_mesh(nu=64, nv=16), None)
    orc = oracle.Scene(*sc.oracle_args())
    rays = np.concatenate([scenes.random_rays(6000, seed=5), scenes.random_rays(2000, seed=6, radius=0.2)])
    a, _ = orc.intersect(rays, use_bvh=False)
    b, _ = orc.intersect(rays, use_bvh=True)
    assert np.array_equal(a, b)
    # flattened ids are instance-major
    total = 0
    for i, inst in enumerate(sc.instances):
        assert orc.triangle_ids(total) == (i, 0)
        total += sc.meshes[inst.mesh].indices.shape[0]
    assert total == orc.n_triangles


def test_render_golden_images(oracle):
    g = np.load(os.path.join(GOLD, "cornell_32x32_4spp.npz"))
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    orc = oracle.Scene(*sc.oracle_args())
    cam = R.default_camera(32, 32)
    for use_bvh in (False, True):
        acc, img, st = orc.render(oracle.make_push(cam.as_dict(), frame=3), 32, 32, n_frames=4, max_depth=8,
                                  use_bvh=use_bvh)
        assert np.array_equal(acc.view(np.uint32), g["accum_bits"]) and np.array_equal(img, g["rgba8"])
        assert st.rays == int(g["rays"])
    # the same dispatches in the bounce-count debug view: another film, the same rays
    gd = np.load(os.path.join(GOLD, "cornell_debug_view_32x32_4spp.npz"))
    acc, img, st = orc.render(oracle.make_push(cam.as_dict(), frame=3), 32, 32, n_frames=4, max_depth=8, debug_view=1)
    assert np.array_equal(acc.view(np.uint32), gd["accum_bits"]) and np.array_equal(img, gd["rgba8"])
    assert st.rays == int(gd["rays"]) == int(g["rays"])
    g = np.load(os.path.join(GOLD, "ref4_32x32_4spp.npz"))
    sc = scenes.ref4_scene(scenes.knot_mesh(nu=48, nv=12), scenes.procedural_sky(32, 16))
    orc = oracle.Scene(*sc.oracle_args())
    acc, img, st = orc.render(oracle.make_push(cam.as_dict(), frame=0, samples_per_frame=2), 32, 32, n_frames=2)
    assert np.array_equal(acc.view(np.uint32), g["accum_bits"]) and np.array_equal(img, g["rgba8"])


def test_c5_scene_golden_image(oracle):
    """Object in the Cornell box under a sky that went through a Radiance file:
    the numpy decoder and the library's give the same texels, and the oracle
    (BVH and brute force) renders the committed film from them."""
    import tempfile
    from oracle import hdr_image as H
    g = np.load(os.path.join(GOLD, "c5_64x36_4spp.npz"))
    sky = scenes.c5_sky_from_hdr(64, 32)
    with tempfile.TemporaryDirectory() as d:
        scenes.write_hdr(os.path.join(d, "sky.hdr"), scenes.procedural_sky(64, 32))
        restated = H.decode(open(os.path.join(d, "sky.hdr"), "rb").read())
    assert np.array_equal(sky.view(np.uint32), np.ascontiguousarray(restated, np.float32).view(np.uint32))
    sc = scenes.dragon_in_cornell(sky, nu=64, nv=16)
    orc = oracle.Scene(*sc.oracle_args())
    cam = R.default_camera(64, 36)
    push = oracle.make_push(cam.as_dict(), frame=1)
    for use_bvh in (False, True):
        acc, img, st = orc.render(push, 64, 36, n_frames=4, use_bvh=use_bvh)
        assert np.array_equal(acc.view(np.uint32), g["accum_bits"]) and np.array_equal(img, g["rgba8"])
        assert st.rays == int(g["rays"])
        assert np.array_equal(orc.primary_hits(push, 64, 36, use_bvh=use_bvh), g["primary_ids"])


def test_running_mean_equals_batched_dispatches(oracle):
    """n_frames dispatches in one call == the same dispatches one call at a
    time feeding the accumulator back (shader.rgen:202-216); alpha stays 1."""
    sc = scenes.cornell_box()
    orc = oracle.Scene(*sc.oracle_args())
    cam = R.default_camera(24, 16)
    acc_all, _, _ = orc.render(oracle.make_push(cam.as_dict(), frame=10), 24, 16, n_frames=3, use_bvh=False)
    acc = None
    for k in range(3):
        acc, _, _ = orc.render(oracle.make_push(cam.as_dict(), frame=10 + k, sample_count=k), 24, 16, accum=acc,
                               n_frames=1, use_bvh=False)
    assert np.array_equal(acc.view(np.uint32), acc_all.view(np.uint32))
    assert (acc[..., 3] == 1.0).all()


def test_debug_view_counts_the_path_loop(oracle):
    """shader.rgen:195-199, the branch under "#else": a sample is worth
    bounces / 16, where bounces is the loop counter of radiance() when it
    returned (shader.rgen:65): 0 for a camera ray that reaches the sky, the
    iteration of the miss or of the failed roulette otherwise, max_depth for a
    path that survives every iteration."""
    cam = R.default_camera(24, 16)
    push = oracle.make_push(cam.as_dict(), frame=2)
    # no triangles: every camera ray reaches the sky in iteration 0
    empty = oracle.Scene([], [], None)
    acc, img, st = empty.render(push, 24, 16, n_frames=2, debug_view=1)
    assert (acc[..., :3] == 0.0).all() and (acc[..., 3] == 1.0).all() and st.rays == 2 * 24 * 16
    # closed box, one iteration allowed: a sample is 1/16 if the path survived its roulette, else 0
    sc = scenes.cornell_box()
    orc = oracle.Scene(*sc.oracle_args())
    acc, _, _ = orc.render(push, 24, 16, n_frames=1, max_depth=1, use_bvh=False, debug_view=1)
    assert set(np.unique(acc[..., 0])) <= {0.0, 1.0 / 16.0} and acc[..., 0].max() == 1.0 / 16.0
    # the mean over dispatches of integers / 16 never exceeds max_depth / 16, and relates to the ray count:
    # a path that ended in iteration b traced b + 1 rays, one that ran through all of them max_depth
    acc, _, st = orc.render(push, 24, 16, n_frames=1, max_depth=5, use_bvh=False, debug_view=1)
    b = np.rint(acc[..., 0] * 16.0).astype(int)
    assert np.array_equal(acc[..., 0], b / np.float32(16.0)) and b.max() <= 5
    assert b.sum() <= st.rays <= b.sum() + 24 * 16
    plain, _, st_plain = orc.render(push, 24, 16, n_frames=1, max_depth=5, use_bvh=False)
    assert st_plain.rays == st.rays  # the view changes what is accumulated, not the paths


def test_sum_mode_matches_mean(oracle):
    sc = scenes.cornell_box()
    orc = oracle.Scene(*sc.oracle_args())
    cam = R.default_camera(24, 16)
    mean, _, _ = orc.render(oracle.make_push(cam.as_dict()), 24, 16, n_frames=4, use_bvh=False)
    s, _, _ = orc.render(oracle.make_push(cam.as_dict()), 24, 16, n_frames=4, use_bvh=False, accumulate_sum=True)
    assert (s[..., 3] == 4.0).all()
    assert np.allclose(s[..., :3] / 4.0, mean[..., :3], rtol=1e-5, atol=1e-7)


def test_libm_mode_converges_to_the_same_image(oracle):
    """The choice of transcendental implementation does not change the
    converged image: mean relative luminance within 1 %, per-pixel RMSE small
    against the Monte-Carlo noise level."""
    sc = scenes.cornell_box()
    orc = oracle.Scene(*sc.oracle_args())
    cam = R.default_camera(24, 24)
    a, _, _ = orc.render(oracle.make_push(cam.as_dict()), 24, 24, n_frames=256, max_depth=8, use_bvh=False)
    oracle.set_math_mode(oracle.MATH_LIBM)
    try:
        b, _, _ = orc.render(oracle.make_push(cam.as_dict()), 24, 24, n_frames=256, max_depth=8, use_bvh=False)
    finally:
        oracle.set_math_mode(oracle.MATH_PORTABLE)
    lum = lambda x: 0.2126 * x[..., 0] + 0.7152 * x[..., 1] + 0.0722 * x[..., 2]
    assert abs(lum(a).mean() - lum(b).mean()) <= 0.02 * lum(a).mean()
    rmse = np.sqrt(((a[..., :3] - b[..., :3]) ** 2).mean())
    assert rmse < 0.5 * a[..., :3].mean() + 0.05


def test_empty_and_degenerate_scenes(oracle):
    cam = R.default_camera(8, 8)
    empty = oracle.Scene([], [], None)
    acc, img, st = empty.render(oracle.make_push(cam.as_dict()), 8, 8, use_bvh=True)
    assert st.rays == 64 and set(np.unique(acc[..., 0])) <= {np.float32(0.75), np.float32(0.1)}
    # one zero-area triangle never reports a hit
    m = (np.zeros((3, 3), np.float32), np.tile(np.array([[0, 0, 1]], np.float32), (3, 1)),
         np.array([[0, 1, 2]], np.uint32))
    deg = oracle.Scene([m], [(R.IDENTITY, 0, 0)], None)
    h, _ = deg.intersect(scenes.random_rays(100), use_bvh=True)
    assert (h["id"] == oracle.NO_HIT).all()


def test_inverse_affine(oracle):
    m = np.array([[0.7, 0, 0, 1], [0, 0.7, 0, 0], [0, 0, 0.7, 0]], np.float32)
    inv = oracle.inverse_affine(m).reshape(3, 4)
    assert np.allclose(inv[:, :3], np.eye(3) / 0.7, rtol=1e-6) and np.isclose(inv[0, 3], -1 / 0.7, rtol=1e-6)
