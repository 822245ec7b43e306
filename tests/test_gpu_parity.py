"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle.

Bar: bit-exact. Hit ids, (t, u, v), the RGBA32F accumulator and the RGBA8
target must all be identical to the oracle's, because both sides evaluate the
same sequence of IEEE-754 binary32 operations."""
import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_device_math_bit_exact(ctx, oracle):
    rng = np.random.default_rng(0)
    u = rng.random(200000, dtype=np.float32)
    theta = (np.float32(6.2831855) * u).astype(np.float32)
    s, c = ctx.math_eval(0, theta)
    so, co = oracle.math_eval(0, oracle.MATH_PORTABLE, theta)
    assert np.array_equal(_bits(s), _bits(so)) and np.array_equal(_bits(c), _bits(co))
    x = np.maximum(u, np.float32(1.175494351e-38))
    assert np.array_equal(_bits(ctx.math_eval(1, x)), _bits(oracle.math_eval(1, oracle.MATH_PORTABLE, x)))
    y = (2 * u - 1).astype(np.float32)
    assert np.array_equal(_bits(ctx.math_eval(2, y)), _bits(oracle.math_eval(2, oracle.MATH_PORTABLE, y)))
    z = rng.standard_normal(200000).astype(np.float32)
    w = rng.standard_normal(200000).astype(np.float32)
    assert np.array_equal(_bits(ctx.math_eval(3, z, w)),
                          _bits(oracle.math_eval(3, oracle.MATH_PORTABLE, z, w)))
    # IEEE division, square root and uint -> float conversion
    big = (z * np.float32(1e-3)).astype(np.float32)
    assert np.array_equal(_bits(ctx.math_eval(4, z, w)), _bits(z / w))
    assert np.array_equal(_bits(ctx.math_eval(4, big, w)), _bits(big / w))
    assert np.array_equal(_bits(ctx.math_eval(5, np.abs(z))), _bits(np.sqrt(np.abs(z))))
    ints = rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32)
    ref = (ints.astype(np.float32) * np.float32(1.0 / 4294967296.0)).astype(np.float32)
    assert np.array_equal(_bits(ctx.math_eval(6, ints.view(np.float32))), _bits(ref))
    # subnormals are kept, not flushed
    tiny = np.full(16, 1e-40, np.float32)
    assert np.array_equal(_bits(ctx.math_eval(4, tiny, np.full(16, 2.0, np.float32))), _bits(tiny / np.float32(2)))


def test_zero_aperture_lens_sample_keeps_only_the_signs(ctx, oracle):
    """With aperture_radius == 0 (the reference's default, src/camera.cpp:87) the lens
    sample of shader.rgen:184 is multiplied by zero; the kernels then skip the square
    root and the polynomials and derive the signs of the two zeros from the angle's
    quadrant. Those signs must be the signs of the values sincos_ computes, for every
    angle 2 pi u the RNG can produce: checked on random draws, on the quadrant
    boundaries and their float neighbours, on 0 and on u = 1.0."""
    rng = np.random.default_rng(11)
    u = (rng.integers(0, 2 ** 32, 400000, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32)).astype(np.float32)
    edges = np.array([0.0, 0.25, 0.5, 0.75, 1.0, 0.125, 0.375, 0.625, 0.875], np.float32)
    near = np.concatenate([np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-1)), edges,
                           np.arange(1, 4097, dtype=np.float32) * np.float32(2.0 ** -32)])
    u = np.concatenate([u, near[(near >= 0) & (near <= 1)]]).astype(np.float32)
    theta = (np.float32(6.28318530717958647692) * u).astype(np.float32)
    s, c = ctx.math_eval(0, theta)
    out = np.empty_like(theta), np.empty_like(theta)
    R._check(R.lib().ptc_math_eval(ctx.handle, 7, theta.ctypes.data, theta.ctypes.data, out[0].ctypes.data,
                                   out[1].ctypes.data, theta.size))
    assert np.array_equal(np.signbit(out[0]), np.signbit(s))
    assert np.array_equal(np.signbit(out[1]), np.signbit(c))
    assert (out[0] == 0).all() and (out[1] == 0).all()
    # and a render whose camera has zeros of both signs in its position and an aperture of -0: the film equals
    # the oracle's, which evaluates the lens sample in full
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    w, h = 40, 24
    for aperture, pos_x in ((0.0, -0.0), (-0.0, 0.0), (0.05, -0.0)):
        cam = R.default_camera(w, h)
        cam.position.x, cam.position.y = pos_x, -0.0
        cam.aperture_radius = aperture
        for integrator in (R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM):
            rr = R.RenderResources(ctx, w, h, dev)
            rr.max_depth, rr.integrator = 5, integrator
            rr.render(cam, frame=1, n_frames=3)
            ref_acc, ref_img, _ = orc.render(oracle.make_push(cam.as_dict(), frame=1), w, h, n_frames=3, max_depth=5,
                                             use_bvh=False)
            assert np.array_equal(_bits(rr.read_accum()), _bits(ref_acc)), (aperture, integrator)
            assert np.array_equal(rr.read_rgba8(), ref_img)
            rr.close()
    dev.close()


def _small_knot():
    return scenes.knot_mesh(nu=96, nv=24)


SCENES = {
    "cornell": lambda: scenes.cornell_box(),
    "cornell_glass_mirror": lambda: scenes.cornell_box(tall_kind=1, short_kind=3),
    "ref4_knot_sky": lambda: scenes.ref4_scene(_small_knot(), scenes.procedural_sky(64, 32)),
    "knot_specular_sky": lambda: scenes.dragon_scene(envmap=scenes.procedural_sky(64, 32), nu=160, nv=40),
}


@pytest.fixture(scope="module", params=sorted(SCENES))
def scene_pair(request, ctx, oracle):
    sc = SCENES[request.param]()
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    yield request.param, sc, dev, orc
    dev.close()


def test_bvh_structure_and_triangles(scene_pair, oracle):
    name, sc, dev, orc = scene_pair
    nodes, tris = dev.read_bvh()
    assert dev.info.n_triangles == orc.n_triangles == sc.n_triangles
    rc, cost = oracle.bvh8_validate(nodes, tris, orc.world_triangles())
    assert rc == 0, f"{name}: BVH validation code {rc}"
    assert abs(cost - dev.info.sah_cost) <= 0.02 * cost + 1e-3


def test_trace_matches_brute_force(scene_pair, oracle):
    name, sc, dev, orc = scene_pair
    rays = np.concatenate([scenes.random_rays(30000, seed=3), scenes.random_rays(10000, seed=4, radius=0.3)])
    hits, st = dev.trace(rays, count=True)
    use_bvh = orc.n_triangles > 2000  # brute force is the definition; BVH only to keep the test short
    ref, _ = orc.intersect(rays, use_bvh=use_bvh)
    assert np.array_equal(hits["flat"], ref["id"]), f"{name}: hit ids differ"
    hit = ref["id"] != oracle.NO_HIT
    for k in ("t", "u", "v"):
        assert np.array_equal(_bits(hits[k][hit]), _bits(ref[k][hit])), f"{name}: {k} bits differ"
    # the oracle walking the device-built tree visits exactly the same nodes
    nodes, tris = dev.read_bvh()
    ref8, st8 = oracle.bvh8_intersect(nodes, tris, rays)
    assert np.array_equal(ref8["id"], ref["id"])
    assert st.nodes_visited == st8.nodes_visited and st.tris_tested == st8.tris_tested


def test_primary_hit_ids(scene_pair, oracle):
    name, sc, dev, orc = scene_pair
    w, h = 96, 64
    cam = R.default_camera(w, h)
    for frame in (0, 17):
        ids = dev.primary_hits(cam, w, h, frame)
        ref = orc.primary_hits(oracle.make_push(cam.as_dict(), frame=frame), w, h,
                               use_bvh=orc.n_triangles > 2000)
        assert np.array_equal(ids, ref), f"{name}: {np.count_nonzero(ids != ref)} primary ids differ"
    # the same hits as (gl_InstanceID, gl_PrimitiveID) pairs
    inst, prim = dev.triangle_ids(ids)
    miss = ids == R.NO_HIT
    assert np.all(inst[miss] == R.NO_HIT) and np.all(prim[miss] == R.NO_HIT)
    counts = np.array([sc.meshes[i.mesh].indices.shape[0] for i in sc.instances], np.int64)
    first = np.concatenate([[0], np.cumsum(counts)])
    assert np.all(prim[~miss] < counts[inst[~miss]])
    assert np.array_equal(first[inst[~miss]] + prim[~miss], ids[~miss])


@pytest.mark.parametrize("integrator", [R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM],
                         ids=["megakernel", "wavefront", "stream"])
@pytest.mark.parametrize("spf,n_frames", [(1, 3), (2, 2)])
def test_render_bit_exact(scene_pair, ctx, oracle, spf, n_frames, integrator):
    name, sc, dev, orc = scene_pair
    w, h = 80, 48
    cam = R.default_camera(w, h)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth = 8
    rr.integrator = integrator
    # two calls: a fresh film, then a continuation with sample_count > 0
    rr.render(cam, frame=5, samples_per_frame=spf, n_frames=n_frames)
    rr.render(cam, frame=5 + n_frames, samples_per_frame=spf, n_frames=1)
    acc, img = rr.read_accum(), rr.read_rgba8()
    push = oracle.make_push(cam.as_dict(), frame=5, sample_count=0, samples_per_frame=spf)
    ref_acc, ref_img, st = orc.render(push, w, h, n_frames=n_frames + 1, max_depth=8,
                                      use_bvh=orc.n_triangles > 2000)
    assert rr.sample_count == spf * (n_frames + 1)
    assert np.array_equal(_bits(acc), _bits(ref_acc)), \
        f"{name}: {np.count_nonzero(_bits(acc) != _bits(ref_acc))} accumulator words differ"
    assert np.array_equal(img, ref_img)
    rr.close()


@pytest.mark.parametrize("integrator", [R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM],
                         ids=["megakernel", "wavefront", "stream"])
@pytest.mark.parametrize("max_depth,wavefront_bounces", [(8, 0), (3, 0), (1, 0), (6, 2)])
def test_bounce_count_debug_view_bit_exact(scene_pair, ctx, oracle, integrator, max_depth, wavefront_bounces):
    """The debug view the reference keeps under "#else" (shader.rgen:195-199): each
    sample adds bounces / 16, the path loop's counter when the path ended. Paths
    that run through all max_depth iterations, that end at the sky and that end
    by Russian roulette all occur in these scenes; every integrator ends paths
    in different kernels (first-bounce pass, launch pairs, persistent launch)."""
    name, sc, dev, orc = scene_pair
    w, h = 64, 40
    cam = R.default_camera(w, h)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth, rr.integrator, rr.debug_view, rr.wavefront_bounces = max_depth, integrator, 1, wavefront_bounces
    rr.render(cam, frame=9, samples_per_frame=2, n_frames=2)
    rr.render(cam, frame=11, samples_per_frame=1, n_frames=17)  # the pixel-major first-bounce pass
    acc, img = rr.read_accum(), rr.read_rgba8()
    use_bvh = orc.n_triangles > 2000
    ref_acc, _, _ = orc.render(oracle.make_push(cam.as_dict(), frame=9, samples_per_frame=2), w, h, n_frames=2,
                               max_depth=max_depth, use_bvh=use_bvh, debug_view=1)
    ref_acc, ref_img, _ = orc.render(oracle.make_push(cam.as_dict(), frame=11, sample_count=4), w, h, accum=ref_acc,
                                     n_frames=17, max_depth=max_depth, use_bvh=use_bvh, debug_view=1)
    assert np.array_equal(_bits(acc), _bits(ref_acc))
    assert np.array_equal(img, ref_img)
    assert acc[..., :3].max() <= max_depth / 16.0
    if max_depth == 1 and name.startswith("cornell"):
        assert acc[..., :3].max() > 0.0  # camera paths that survive their only iteration count 1
    rr.close()


@pytest.mark.parametrize("two_level", [False, True], ids=["flattened", "two_level"])
@pytest.mark.parametrize("spf,n_frames,wavefront_bounces,static", [(1, 3, 0, False), (2, 2, 0, False), (1, 20, 0, False),
                                                                  (1, 5, 2, False), (2, 17, 3, False), (1, 4, 0, True)])
def test_regrouped_hand_over_bit_exact(ctx, oracle, two_level, spf, n_frames, wavefront_bounces, static):
    """ptc_render_params.regroup: the paths that survive a bounce are handed to
    the next tracing launch sorted by the cell of their ray's origin and its
    octant. Every path carries its RNG state, so the film does not change: equal
    to the oracle's bit for bit, with the persistent launch right after the
    first bounce, with bounces as launch pairs before it, and through the static
    workspace."""
    sc = scenes.ref4_scene(scenes.knot_mesh(nu=48, nv=12), scenes.procedural_sky(64, 32))
    dev = R.DeviceScene(ctx, sc, two_level=two_level)
    orc = oracle.Scene(*sc.oracle_args(), two_level=two_level)
    w, h = 72, 40
    cam = R.default_camera(w, h)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth, rr.integrator, rr.regroup, rr.wavefront_bounces = 7, R.INTEGRATOR_WAVEFRONT, 1, wavefront_bounces
    ctx.reset_stats()
    rr.render(cam, frame=2, samples_per_frame=spf, n_frames=n_frames, static=static)
    rr.render(cam, frame=2 + n_frames, samples_per_frame=spf, n_frames=1, static=static)
    acc, img, rays = rr.read_accum(), rr.read_rgba8(), ctx.stats().rays
    push = oracle.make_push(cam.as_dict(), frame=2, samples_per_frame=spf)
    ref_acc, ref_img, st = orc.render(push, w, h, n_frames=n_frames + 1, max_depth=7, use_bvh=not two_level)
    assert np.array_equal(_bits(acc), _bits(ref_acc))
    assert np.array_equal(img, ref_img)
    assert rays == st.rays
    rr.close()
    dev.close()


@pytest.mark.parametrize("spf,n_frames", [(1, 40), (1, 70), (2, 33)])
def test_large_batches_bit_exact(ctx, oracle, spf, n_frames):
    """Batches of 32 and more dispatches take the primary pass's other mapping
    (one wave per pixel, dispatches as lanes, partial last group); image sizes
    that are not multiples of the tile size; all integrators agree with the oracle."""
    sc = scenes.ref4_scene(_small_knot(), scenes.procedural_sky(32, 16))
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    w, h = 37, 21
    cam = R.default_camera(w, h)
    push = oracle.make_push(cam.as_dict(), frame=2, sample_count=0, samples_per_frame=spf)
    ref_acc, ref_img, _ = orc.render(push, w, h, n_frames=n_frames, max_depth=6)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth = 6
    for integrator in (R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM, R.INTEGRATOR_MEGAKERNEL):
        rr.integrator = integrator
        R.reset_render(rr)
        rr.render(cam, frame=2, samples_per_frame=spf, n_frames=n_frames)
        assert np.array_equal(_bits(rr.read_accum()), _bits(ref_acc)), integrator
        assert np.array_equal(rr.read_rgba8(), ref_img), integrator
    # launch-pair bounces before the persistent kernel give the same film
    rr.integrator, rr.wavefront_bounces = R.INTEGRATOR_WAVEFRONT, 2
    R.reset_render(rr)
    rr.render(cam, frame=2, samples_per_frame=spf, n_frames=n_frames)
    assert np.array_equal(_bits(rr.read_accum()), _bits(ref_acc))
    rr.close()
    dev.close()


def test_aces_tone_map_bit_exact(ctx, oracle):
    """The alternative tone map of shader.rgen:107-135: same accumulator, RGBA8
    target identical to the oracle's, also through ptc_film_resolve."""
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    w, h = 72, 40
    cam = R.default_camera(w, h)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth = 6
    ref = {t: orc.render(oracle.make_push(cam.as_dict(), frame=3), w, h, n_frames=3, max_depth=6,
                         use_bvh=False, tonemap=t) for t in (oracle.TONEMAP_PBR_NEUTRAL, oracle.TONEMAP_ACES)}
    assert not np.array_equal(ref[0][1], ref[1][1])
    for integrator in (R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT):
        rr.integrator, rr.tonemap = integrator, R.TONEMAP_ACES
        R.reset_render(rr)
        rr.render(cam, frame=3, n_frames=3)
        assert np.array_equal(_bits(rr.read_accum()), _bits(ref[1][0]))
        assert np.array_equal(rr.read_rgba8(), ref[1][1])
        rr.resolve(sum_mode=False)
        assert np.array_equal(rr.read_rgba8(), ref[1][1])
    rr.tonemap = 7
    with pytest.raises(R.PtcError, match="unknown tonemap"):
        rr.render(cam, frame=0)
    rr.close()
    dev.close()


def test_ray_and_path_counters(ctx, oracle):
    sc = scenes.cornell_box()
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    w, h = 64, 64
    cam = R.default_camera(w, h)
    rr = R.RenderResources(ctx, w, h, dev)
    _, _, ref = orc.render(oracle.make_push(cam.as_dict()), w, h, n_frames=2, use_bvh=False)
    for integrator in (R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM):
        rr.integrator = integrator
        R.reset_render(rr)
        ctx.reset_stats()
        rr.render(cam, frame=0, n_frames=2)
        st = ctx.stats()
        assert (st.rays, st.paths) == (ref.rays, ref.paths)
    rr.close()
    dev.close()
