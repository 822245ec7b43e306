"""The HIP path against the committed fixtures of tests/golden/ (not against an
oracle built on the spot): the same workloads tests/golden/make_golden.py
rendered and traced when the files were written, repeated on the device through
the C ABI and compared bit for bit. If the oracle and the device code ever
drift together, the oracle's own CPU tests and these catch it from both sides."""
import json
import os

import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = [R.INTEGRATOR_MEGAKERNEL, R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM]


def _unhex(words):
    return np.array([int(w, 16) for w in words], dtype=np.uint32)


def _check_film(ctx, sc, fixture, frame, spf, n_frames, max_depth, debug_view=0):
    gold = np.load(os.path.join(GOLDEN, fixture))
    dev = R.DeviceScene(ctx, sc)
    cam = R.default_camera(32, 32)
    rr = R.RenderResources(ctx, 32, 32, dev)
    rr.max_depth, rr.debug_view = max_depth, debug_view
    for integrator in ALL:
        rr.integrator = integrator
        R.reset_render(rr)
        ctx.reset_stats()
        rr.render(cam, frame=frame, samples_per_frame=spf, n_frames=n_frames)
        assert np.array_equal(rr.read_accum().view(np.uint32), gold["accum_bits"]), (fixture, integrator)
        assert np.array_equal(rr.read_rgba8(), gold["rgba8"]), (fixture, integrator)
        assert ctx.stats().rays == int(gold["rays"]), (fixture, integrator)
    rr.close()
    dev.close()


def test_cornell_film_equals_the_committed_bits(ctx):
    _check_film(ctx, scenes.cornell_box(tall_kind=1, short_kind=3), "cornell_32x32_4spp.npz",
                frame=3, spf=1, n_frames=4, max_depth=8)


def test_cornell_debug_view_equals_the_committed_bits(ctx):
    _check_film(ctx, scenes.cornell_box(tall_kind=1, short_kind=3), "cornell_debug_view_32x32_4spp.npz",
                frame=3, spf=1, n_frames=4, max_depth=8, debug_view=1)


def test_ref4_film_equals_the_committed_bits(ctx):
    sc = scenes.ref4_scene(scenes.knot_mesh(nu=48, nv=12), scenes.procedural_sky(32, 16))
    _check_film(ctx, sc, "ref4_32x32_4spp.npz", frame=0, spf=2, n_frames=2, max_depth=32)


def test_c5_film_equals_the_committed_bits(ctx):
    """The 4K configuration's scene (object inside the Cornell box under an
    equirectangular sky read from a Radiance file) at 64x36."""
    gold = np.load(os.path.join(GOLDEN, "c5_64x36_4spp.npz"))
    sc = scenes.dragon_in_cornell(scenes.c5_sky_from_hdr(64, 32), nu=64, nv=16)
    dev = R.DeviceScene(ctx, sc)
    cam = R.default_camera(64, 36)
    rr = R.RenderResources(ctx, 64, 36, dev)
    for integrator in ALL:
        rr.integrator = integrator
        R.reset_render(rr)
        rr.render(cam, frame=1, n_frames=4)
        assert np.array_equal(rr.read_accum().view(np.uint32), gold["accum_bits"]), integrator
        assert np.array_equal(rr.read_rgba8(), gold["rgba8"]), integrator
    assert np.array_equal(dev.primary_hits(cam, 64, 36, frame=1), gold["primary_ids"])
    rr.close()
    dev.close()


def test_cornell_hits_equal_the_committed_bits(ctx):
    gold = json.load(open(os.path.join(GOLDEN, "cornell_hits.json")))
    rays = _unhex(gold["rays"]).view(np.float32).reshape(-1, 6)
    for two_level in (False, True):
        dev = R.DeviceScene(ctx, scenes.cornell_box(tall_kind=1, short_kind=3), two_level=two_level)
        hits = dev.trace(rays)
        ids = np.array(gold["id"], dtype=np.uint32)
        assert np.array_equal(hits["flat"], ids)
        hit = ids != R.NO_HIT
        assert hit.any()
        if not two_level:  # identity transforms: object space is world space, but only the flattened mode promises these bits
            for k in "tuv":
                assert np.array_equal(hits[k][hit].view(np.uint32), _unhex(gold[k])[hit]), k
        dev.close()


def test_device_math_on_the_committed_tone_map_inputs(ctx):
    """A film whose accumulator holds the fixture's radiances resolves to the
    fixture's bytes: tone map + UNORM8 on the device against tonemap.json."""
    import torch
    gold = json.load(open(os.path.join(GOLDEN, "tonemap.json")))["tonemap"]
    n = len(gold)
    acc = np.ones((1, n, 4), np.float32)
    for i, e in enumerate(gold):
        acc[0, i, :3] = _unhex(e["rgb"]).view(np.float32)
    d_acc = torch.from_numpy(acc).cuda()
    rr = R.RenderResources(ctx, n, 1, scenes.cornell_box(), accum_dev=d_acc.data_ptr())
    for tonemap, key in ((R.TONEMAP_PBR_NEUTRAL, "unorm8"), (R.TONEMAP_ACES, "aces_unorm8")):
        # the film remembers the tone map of its last render call: render nothing visible first
        rr.tonemap = tonemap
        rr.max_depth = 1
        rr.render(R.default_camera(n, 1), frame=0)
        ctx.synchronize()  # the context's stream is not torch's: the render must be over before torch writes the film
        d_acc.copy_(torch.from_numpy(acc))
        torch.cuda.synchronize()
        rr.resolve(sum_mode=False)
        out = rr.read_rgba8()[0]
        for i, e in enumerate(gold):
            assert list(out[i, :3]) == e[key], (key, i)
            assert out[i, 3] == 255
        R.reset_render(rr)
    rr.close()
