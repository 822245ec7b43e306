"""Edge cases of the HIP path through the C ABI, each against the CPU oracle:
empty and degenerate scenes, depth 0 and 1, tiny films, a workspace bound that
splits one call into several batches, continued films, caller-owned buffers,
device-resident ray queries, and argument errors."""
import ctypes as C

import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu

INTEGRATORS = [R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM, R.INTEGRATOR_AUTO]


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _render_all(ctx, oracle, sc, w, h, n_frames, max_depth, spf=1, **kw):
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    cam = R.default_camera(w, h)
    push = oracle.make_push(cam.as_dict(), frame=1, sample_count=0, samples_per_frame=spf)
    ref_acc, ref_img, ref_st = orc.render(push, w, h, n_frames=n_frames, max_depth=max_depth, use_bvh=False)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth = max_depth
    for k, v in kw.items():
        setattr(rr, k, v)
    for integrator in INTEGRATORS:
        rr.integrator = integrator
        R.reset_render(rr)
        ctx.reset_stats()
        rr.render(cam, frame=1, samples_per_frame=spf, n_frames=n_frames)
        assert np.array_equal(_bits(rr.read_accum()), _bits(ref_acc)), integrator
        assert np.array_equal(rr.read_rgba8(), ref_img), integrator
        st = ctx.stats()
        assert (st.rays, st.paths) == (ref_st.rays, ref_st.paths), integrator
    rr.close()
    dev.close()


def test_scene_without_triangles(ctx, oracle):
    """No geometry at all: every path ends at the sky after one ray."""
    sky = scenes.procedural_sky(32, 16)
    empty = R.Mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    for sc in (R.Scene([], [], sky), R.Scene([empty], [R.Instance(R.IDENTITY, 0, 0)], None)):
        _render_all(ctx, oracle, sc, 24, 10, n_frames=3, max_depth=4)
        dev = R.DeviceScene(ctx, sc)
        assert dev.info.n_triangles == 0
        hits, _ = dev.trace(scenes.random_rays(100), count=True)
        assert np.all(hits["flat"] == 0xFFFFFFFF)
        dev.close()


def test_degenerate_and_single_triangles(ctx, oracle):
    """One triangle; zero-area and collinear triangles among good ones (they can
    never be hit: det == 0) must not disturb the build or the traversal."""
    p = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0],      # a real triangle
                  [0.5, 0.5, -0.5], [0.5, 0.5, -0.5], [0.5, 0.5, -0.5],  # a point
                  [-1, 0, -1], [0, 0, -1], [1, 0, -1]], np.float32)      # collinear
    n = np.tile(np.array([0, 0, 1], np.float32), (9, 1))
    one = R.Mesh(p[:3], n[:3], np.array([[0, 1, 2]], np.uint32))
    mixed = R.Mesh(p, n, np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.uint32))
    for mesh in (one, mixed):
        sc = R.Scene([mesh], [R.Instance(R.IDENTITY, 0, 0), R.Instance(np.array(
            [[0.5, 0, 0, 0], [0, 0.5, 0, 0.2], [0, 0, 0.5, 1.0]], np.float32), 0, 1)], None)
        _render_all(ctx, oracle, sc, 33, 17, n_frames=4, max_depth=5)


@pytest.mark.parametrize("max_depth", [1, 2])
def test_shallow_depths(ctx, oracle, max_depth):
    _render_all(ctx, oracle, scenes.cornell_box(tall_kind=1, short_kind=3), 40, 24, n_frames=3, max_depth=max_depth)


@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (17, 1), (1, 19)])
def test_tiny_films(ctx, oracle, w, h):
    _render_all(ctx, oracle, scenes.cornell_box(), w, h, n_frames=20, max_depth=4)


@pytest.mark.parametrize("spf", [1, 3])
def test_workspace_bound_splits_a_call_into_batches(ctx, oracle, spf):
    """max_paths_in_flight below one call's paths: several batches, the last one
    partial, same film."""
    w, h, n_frames = 20, 12, 37
    _render_all(ctx, oracle, scenes.cornell_box(tall_kind=1), w, h, n_frames=n_frames, max_depth=5, spf=spf,
                max_paths_in_flight=w * h * spf * 16 + 7)   # 16 dispatches per batch at most (fewer where the queue,
                                                            # which covers whole 32x32-pixel groups, would exceed
                                                            # twice the bound): several batches, the last one partial
    _render_all(ctx, oracle, scenes.cornell_box(tall_kind=1), w, h, n_frames=5, max_depth=5, spf=spf,
                max_paths_in_flight=1)                      # one dispatch at a time


def test_continued_film_and_external_sample_count(ctx, oracle):
    """draw_frame's bookkeeping (src/renderer.cpp:2018-2110): a film rendered in
    pieces equals one rendered in a single call; ptc_film_set_sample_count moves
    the running-mean weight as the host counter would."""
    sc = scenes.cornell_box()
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    w, h = 28, 20
    cam = R.default_camera(w, h)
    ref_acc, ref_img, _ = orc.render(oracle.make_push(cam.as_dict(), frame=0), w, h, n_frames=7, max_depth=4,
                                     use_bvh=False)
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth = 4
    for first, n in ((0, 1), (1, 3), (4, 2), (6, 1)):
        rr.render(cam, frame=first, n_frames=n)
    assert rr.sample_count == 7
    assert np.array_equal(_bits(rr.read_accum()), _bits(ref_acc))
    assert np.array_equal(rr.read_rgba8(), ref_img)
    # the same last dispatch again, from the state before it
    acc6, _, _ = orc.render(oracle.make_push(cam.as_dict(), frame=0), w, h, n_frames=6, max_depth=4, use_bvh=False)
    rr2 = R.RenderResources(ctx, w, h, dev)
    rr2.max_depth = 4
    rr2.render(cam, frame=0, n_frames=6)
    assert np.array_equal(_bits(rr2.read_accum()), _bits(acc6))
    R._check(R.lib().ptc_film_set_sample_count(rr2.film, 6))
    rr2.render(cam, frame=6, n_frames=1)
    assert np.array_equal(_bits(rr2.read_accum()), _bits(ref_acc))
    rr.close()
    rr2.close()
    dev.close()


def test_caller_owned_film_buffers_and_device_rays(ctx, oracle):
    """The film can live in caller-owned device memory (what an interop display
    or a collective needs), and rays / hits can stay on the device."""
    torch = pytest.importorskip("torch")
    sc = scenes.cornell_box(short_kind=1)
    dev = R.DeviceScene(ctx, sc)
    orc = oracle.Scene(*sc.oracle_args())
    w, h = 32, 16
    cam = R.default_camera(w, h)
    accum = torch.full((h, w, 4), 7.0, dtype=torch.float32, device="cuda:0")
    rgba8 = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    rr = R.RenderResources(ctx, w, h, dev, accum_dev=accum.data_ptr(), rgba8_dev=rgba8.data_ptr())
    rr.max_depth = 4
    rr.render(cam, frame=0, n_frames=2)
    ctx.synchronize()
    ref_acc, ref_img, _ = orc.render(oracle.make_push(cam.as_dict(), frame=0), w, h, n_frames=2, max_depth=4,
                                     use_bvh=False)
    assert np.array_equal(_bits(accum.cpu().numpy()), _bits(ref_acc))
    assert np.array_equal(rgba8.cpu().numpy().view(np.uint8).reshape(h, w, 4), ref_img)
    rays = scenes.random_rays(5000, seed=9)
    d_rays = torch.from_numpy(rays).to("cuda:0")
    d_hits = torch.zeros((rays.shape[0], 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    R._check(R.lib().ptc_trace_rays_device(ctx.handle, dev.handle, d_rays.data_ptr(), rays.shape[0], 0.0,
                                           d_hits.data_ptr()))
    ctx.synchronize()
    got = d_hits.cpu().numpy().view(R.HIT_DTYPE).reshape(-1)
    ref, _ = orc.intersect(rays, use_bvh=False)
    assert np.array_equal(got["flat"], ref["id"])
    hit = ref["id"] != oracle.NO_HIT
    assert np.array_equal(_bits(got["t"][hit]), _bits(ref["t"][hit]))
    rr.close()
    dev.close()


def test_argument_errors(ctx):
    L = R.lib()
    sc = scenes.cornell_box()
    dev = R.DeviceScene(ctx, sc)
    rr = R.RenderResources(ctx, 8, 8, dev)
    cam = R.default_camera(8, 8)
    p = R.RenderParams()
    p.n_frames = 1
    p.samples_per_frame = 1
    assert L.ptc_render(ctx.handle, dev.handle, rr.film, None, C.byref(p)) == -1
    assert L.ptc_render(ctx.handle, None, rr.film, C.byref(cam), C.byref(p)) == -1
    assert b"NULL" in L.ptc_last_error()
    p.integrator = 9
    assert L.ptc_render(ctx.handle, dev.handle, rr.film, C.byref(cam), C.byref(p)) == -1
    other = R.create_context(0)
    p.integrator = 0
    assert L.ptc_render(other.handle, dev.handle, rr.film, C.byref(cam), C.byref(p)) == -1
    assert b"another context" in L.ptc_last_error()
    other.close()
    h = C.c_void_p()
    assert L.ptc_film_create(ctx.handle, 0, 4, None, None, C.byref(h)) == -1
    with pytest.raises(R.PtcError):
        R.DeviceScene(ctx, R.Scene([sc.meshes[0]], [R.Instance(R.IDENTITY, 3, 0)], None))  # mesh index out of range
    with pytest.raises(R.PtcError):
        R.DeviceScene(ctx, R.Scene([sc.meshes[0]], [R.Instance(R.IDENTITY, 0, 7)], None))  # unknown material kind
    bad = R.Mesh(sc.meshes[0].positions, sc.meshes[0].normals, sc.meshes[0].indices + 10 ** 6)
    with pytest.raises(R.PtcError):
        R.DeviceScene(ctx, R.Scene([bad], [R.Instance(R.IDENTITY, 0, 0)], None))           # vertex index out of range
    for bad_value in (np.nan, np.inf, 3e38):
        pos = sc.meshes[0].positions.copy()
        pos[1, 2] = bad_value
        with pytest.raises(R.PtcError, match="not finite or beyond"):
            R.DeviceScene(ctx, R.Scene([R.Mesh(pos, sc.meshes[0].normals, sc.meshes[0].indices)],
                                       [R.Instance(R.IDENTITY, 0, 0)], None))
        xf = R.IDENTITY.copy()
        xf[0, 3] = bad_value
        with pytest.raises(R.PtcError, match="not finite or beyond"):
            R.DeviceScene(ctx, R.Scene([sc.meshes[0]], [R.Instance(xf, 0, 0)], None))
    # a singular transform (zero scale on one axis) has no inverse to take rays to object space with
    flat_axis = R.IDENTITY.copy()
    flat_axis[1, 1] = 0.0
    for two_level in (False, True):
        with pytest.raises(R.PtcError, match="singular"):
            R.DeviceScene(ctx, R.Scene([sc.meshes[0]], [R.Instance(flat_axis, 0, 0)], None), two_level=two_level)
    rr.close()
    dev.close()


def test_path_ids_beyond_32_bits_are_refused(ctx):
    """Path ids are 32-bit: a dispatch of an 8K film with 130 samples per frame
    has more paths than that. The queue-based integrators refuse it (before
    allocating anything); the megakernel, which numbers nothing, is not asked
    to render it here (it would take minutes)."""
    sc = scenes.cornell_box()
    dev = R.DeviceScene(ctx, sc)
    w, h = 7680, 4320
    rr = R.RenderResources(ctx, w, h, dev)
    cam = R.default_camera(w, h)
    for integrator in (R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM, R.INTEGRATOR_AUTO):
        rr.integrator = integrator
        with pytest.raises(R.PtcError, match="exceeds 2\\^32 paths"):
            rr.render(cam, frame=0, samples_per_frame=130)
        assert rr.sample_count == 0
    rr.close()
    dev.close()


def test_draw_frames_equals_consecutive_draw_frame_calls(ctx):
    """draw_frames(n) = n draw_frame() calls: same film, same counters, including
    a partial last dispatch and calls beyond samples_to_render."""
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    dev = R.DeviceScene(ctx, sc)
    w, h = 48, 32
    cam = R.default_camera(w, h)
    films = []
    for batched in (False, True):
        rr = R.RenderResources(ctx, w, h, dev)
        rr.max_depth = 6
        rr.samples_to_render, rr.samples_per_frame = 23, 3   # 7 full dispatches, one of 2 samples, then idle calls
        ctx.global_frame_count = 40
        R.reset_render(rr)
        if batched:
            R.draw_frames(ctx, rr, cam, 5)
            R.draw_frames(ctx, rr, cam, 6)
        else:
            for _ in range(11):
                R.draw_frame(ctx, rr, cam)
        assert rr.sample_count == 23 and ctx.global_frame_count == 51
        films.append((rr.read_accum(), rr.read_rgba8()))
        rr.close()
    assert np.array_equal(_bits(films[0][0]), _bits(films[1][0]))
    assert np.array_equal(films[0][1], films[1][1])
    dev.close()


def test_draw_frame_with_frames_in_flight_equals_immediate_dispatches(ctx):
    """draw_frame() records its dispatch and returns; the recorded run is launched
    as one batch when the camera changes, when the workspace bound is reached,
    or when something reads the film. Film and counters equal those of a loop
    that launches every dispatch at once, through a camera change, a partial
    last dispatch, idle calls and a reset in the middle of a run."""
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    dev = R.DeviceScene(ctx, sc)
    w, h = 48, 32
    cam_a = R.default_camera(w, h)
    cam_b = R.default_camera(w, h)
    R.camera_set_distance(cam_b, 3.0)
    out = []
    for in_flight in (False, True):
        rr = R.RenderResources(ctx, w, h, dev)
        rr.frames_in_flight = in_flight
        rr.max_depth = 6
        rr.samples_to_render, rr.samples_per_frame = 31, 2
        rr.max_paths_in_flight = 5 * w * h * 2  # a recorded run is launched after five dispatches at the latest
        ctx.global_frame_count = 7
        R.reset_render(rr)
        for _ in range(3):
            R.draw_frame(ctx, rr, cam_a)  # reset below; launched all the same by the statistics reset, which flushes
        R.reset_render(rr)
        assert rr.sample_count == 0
        ctx.reset_stats()
        for k in range(7):
            R.draw_frame(ctx, rr, cam_a)
            assert rr.sample_count == 2 * (k + 1)  # recorded samples count
        for _ in range(4):
            R.draw_frame(ctx, rr, cam_b)  # another camera: the run so far is launched first
        st = ctx.stats()  # launches what is recorded
        assert st.paths == 11 * 2 * w * h
        for _ in range(8):
            R.draw_frame(ctx, rr, cam_b)  # 4 full dispatches, one of a single sample, three idle calls
        assert rr.sample_count == 31 and ctx.global_frame_count == 7 + 3 + 19
        out.append((rr.read_accum(), rr.read_rgba8(), ctx.stats().rays))
        rr.close()
    assert np.array_equal(_bits(out[0][0]), _bits(out[1][0]))
    assert np.array_equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2]
    # a recorded run is launched before its scene goes away, and dropped with its film
    rr = R.RenderResources(ctx, w, h, R.DeviceScene(ctx, sc))
    R.draw_frame(ctx, rr, cam_a)
    rr.scene.close()
    assert rr.sample_count == 1
    rr.flush()
    rr.close()
    rr = R.RenderResources(ctx, w, h, dev)
    R.draw_frame(ctx, rr, cam_a)
    rr.close()
    dev.close()


def test_reset_keeps_a_recorded_run_until_the_next_dispatch(ctx):
    """reset_render() only zeroes the counter (src/renderer.cpp:2236-2239) and every
    earlier draw_frame has been submitted: a read-back between the reset and the
    next dispatch shows all of them, recorded ones included; the next dispatch
    overwrites the film, so a run still recorded then is dropped unlaunched."""
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    dev = R.DeviceScene(ctx, sc)
    w, h = 40, 24
    cam = R.default_camera(w, h)
    ref = R.RenderResources(ctx, w, h, dev)
    ref.max_depth = 5
    ref.render(cam, frame=3, n_frames=3)
    three = (ref.read_accum(), ref.read_rgba8())
    R.reset_render(ref)
    ref.render(cam, frame=6, n_frames=1)
    one = (ref.read_accum(), ref.read_rgba8())
    rr = R.RenderResources(ctx, w, h, dev)
    rr.max_depth = 5
    ctx.global_frame_count = 3
    for _ in range(3):
        R.draw_frame(ctx, rr, cam)
    R.reset_render(rr)
    assert rr.sample_count == 0
    assert np.array_equal(_bits(rr.read_accum()), _bits(three[0]))  # the read launches the recorded run
    assert np.array_equal(rr.read_rgba8(), three[1])
    assert rr.sample_count == 0
    R.draw_frame(ctx, rr, cam)  # frame 6 on a reset film
    assert rr.sample_count == 1
    assert np.array_equal(_bits(rr.read_accum()), _bits(one[0]))
    assert np.array_equal(rr.read_rgba8(), one[1])
    # reset, then a dispatch straight away: the recorded run is never launched
    ctx.global_frame_count = 3
    R.reset_render(rr)
    ctx.reset_stats()
    for _ in range(3):
        R.draw_frame(ctx, rr, cam)
    R.reset_render(rr)
    R.reset_render(rr)
    ctx.global_frame_count = 6
    R.draw_frame(ctx, rr, cam)
    assert rr.sample_count == 1
    assert ctx.stats().paths == w * h  # one dispatch ran, not four
    assert np.array_equal(_bits(rr.read_accum()), _bits(one[0]))
    # a sample count supplied from outside after the reset continues from what the film then holds
    ctx.global_frame_count = 3
    R.reset_render(rr)
    for _ in range(3):
        R.draw_frame(ctx, rr, cam)
    R.reset_render(rr)
    rr.sample_count = 3  # "the film holds three samples": it does once the run is launched
    assert np.array_equal(_bits(rr.read_accum()), _bits(three[0]))
    assert rr.sample_count == 3
    rr.close()
    ref.close()
    dev.close()


def test_film_over_caller_memory_is_ordered_by_the_stream_alone(ctx):
    """A film over caller-owned device memory never records: draw_frame() launches
    its dispatch at once, so a caller that orders by the stream only (a
    collective or a copy enqueued behind it, a stream synchronisation) sees every
    dispatch. The accumulator's device pointer of a library-owned film launches
    what is recorded."""
    torch = pytest.importorskip("torch")
    sc = scenes.cornell_box(short_kind=1)
    dev = R.DeviceScene(ctx, sc)
    w, h = 48, 24
    cam = R.default_camera(w, h)
    ref = R.RenderResources(ctx, w, h, dev)
    ref.max_depth = 4
    ref.render(cam, frame=11, n_frames=5)
    want = ref.read_accum()
    own = R.create_context(0)
    stream = torch.cuda.Stream(device="cuda:0")
    own.set_stream(stream.cuda_stream)
    dev2 = R.DeviceScene(own, sc)
    accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    rr = R.RenderResources(own, w, h, dev2, accum_dev=accum.data_ptr())
    rr.max_depth = 4
    own.global_frame_count = 11
    for _ in range(5):
        R.draw_frame(own, rr, cam)  # frames_in_flight is on by default
    with torch.cuda.stream(stream):
        copy = accum.clone()  # enqueued behind the dispatches on the same stream
    stream.synchronize()      # no library call in between
    assert np.array_equal(_bits(copy.cpu().numpy()), _bits(want))
    rr.close()
    # library-owned film: the pointer accessor launches the recorded run
    rr = R.RenderResources(own, w, h, dev2)
    rr.max_depth = 4
    own.global_frame_count = 11
    for _ in range(5):
        R.draw_frame(own, rr, cam)
    ptr = rr.accum_device_ptr()
    stream.synchronize()
    class DeviceArray:  # the film's accumulator as torch sees foreign device memory
        __cuda_array_interface__ = dict(shape=(h, w, 4), typestr="<f4", data=(ptr, False), version=2)

    view = torch.as_tensor(DeviceArray(), device="cuda:0")
    assert np.array_equal(_bits(view.cpu().numpy()), _bits(want))
    rr.close()
    dev2.close()
    own.close()
    ref.close()
    dev.close()


def test_non_finite_rays_hit_nothing_and_visit_nothing(ctx, oracle):
    """A ray with a NaN or an infinity in its origin or direction (paths produce
    them once in 10^8 samples) hits nothing by the definition of the triangle
    test, in both scene kinds, and does not walk the tree: the box test's
    hardware minima / maxima drop NaNs, so without a guard such a ray passes
    every box and visits every node and triangle of the scene."""
    sc = scenes.dragon_scene(nu=232, nv=117)
    good = scenes.random_rays(64, seed=3)
    bad = np.repeat(good[:1], 12, axis=0)
    for k in range(6):
        bad[k, k] = np.nan
        bad[6 + k, k] = np.inf if k % 2 else -np.inf
    orc = oracle.Scene(*sc.oracle_args())
    for two_level in (False, True):
        dev = R.DeviceScene(ctx, sc, two_level=two_level)
        hits, st = dev.trace(bad, count=True)
        assert np.all(hits["flat"] == R.NO_HIT)
        assert st.nodes_visited == 0 and st.tris_tested == 0
        ref, _ = orc.intersect(bad, use_bvh=False)
        assert np.all(ref["id"] == oracle.NO_HIT)
        # finite rays beside them are traced as ever
        mixed = np.concatenate([bad[:3], good, bad[3:]])
        h = dev.trace(mixed)
        assert np.array_equal(h["flat"][3:3 + 64], dev.trace(good)["flat"])
        assert (h["flat"][3:3 + 64] != R.NO_HIT).any()
        dev.close()
