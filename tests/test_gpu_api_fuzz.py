"""Random sequences of C-ABI calls on one film (immediate, recorded and static
dispatches with every queue-based integrator, camera changes, resets, flushes,
read-backs in between) against the same dispatches launched one by one with
the megakernel on a second film: accumulator, RGBA8 target and sample count
must agree bit for bit wherever the sequence looks. Covers what single-purpose
tests do not reach: recorded runs cut by other calls, odd film sizes against the
segmented path queue, workspace bounds that change between calls."""
import numpy as np
import pytest

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu

QUEUE_INTEGRATORS = [R.INTEGRATOR_WAVEFRONT, R.INTEGRATOR_STREAM, R.INTEGRATOR_AUTO]


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check(a, b, what):
    # (also right after a reset: the reference's reset_render only zeroes the counter, src/renderer.cpp:2236-2239,
    # so until the next dispatch the film shows every frame submitted before it, recorded ones included)
    assert a.sample_count == b.sample_count, what
    assert np.array_equal(_bits(a.read_accum()), _bits(b.read_accum())), what
    assert np.array_equal(a.read_rgba8(), b.read_rgba8()), what


@pytest.mark.parametrize("seed", range(20))
def test_random_call_sequences_equal_one_by_one_megakernel_dispatches(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    sc = [scenes.cornell_box(tall_kind=1, short_kind=3),
          scenes.ref4_scene(scenes.knot_mesh(nu=48, nv=12), scenes.procedural_sky(32, 16))][seed % 2]
    dev = R.DeviceScene(ctx, sc, two_level=bool(seed % 3 == 2))
    for _ in range(4):
        w, h = int(rng.integers(1, 130)), int(rng.integers(1, 70))
        cams = [R.default_camera(w, h), R.default_camera(w, h)]
        R.camera_set_distance(cams[1], 2.5)
        got = R.RenderResources(ctx, w, h, dev)
        ref = R.RenderResources(ctx, w, h, dev)
        ref.integrator = R.INTEGRATOR_MEGAKERNEL
        got.max_depth = ref.max_depth = int(rng.integers(1, 7))
        cam, frame, log = cams[0], int(rng.integers(0, 1000)), []
        for step in range(int(rng.integers(8, 20))):
            op = int(rng.integers(0, 10))
            if op <= 5:
                # (16 dispatches and more take the pixel-major primary pass: 16, 32 or 64 lanes per pixel)
                n_frames = int(rng.choice([1, 2, 3, 4, 16, 17, 33, 64, 70], p=[.2, .2, .15, .15, .06, .06, .06, .06, .06]))
                spf = int(rng.integers(1, 3))
                mode = ["immediate", "deferred", "static"][int(rng.integers(0, 3))]
                got.integrator = QUEUE_INTEGRATORS[int(rng.integers(0, 3))]
                # sometimes a bound that cuts the call (or the recorded run) into several batches
                got.max_paths_in_flight = int(rng.choice([0, w * h * spf, 3 * w * h * spf + 1]))
                got.regroup = int(rng.integers(0, 2))  # the queue handed over sorted or as filled: the same film
                got.render(cam, frame=frame, samples_per_frame=spf, n_frames=n_frames,
                           static=mode == "static", deferred=mode == "deferred")
                for k in range(n_frames):
                    ref.render(cam, frame=frame + k, samples_per_frame=spf)
                frame += n_frames
                log.append((mode, got.integrator, n_frames, spf, got.max_paths_in_flight))
            elif op == 6:
                cam = cams[int(rng.integers(0, 2))]
                log.append("camera")
            elif op == 7:
                R.reset_render(got)
                R.reset_render(ref)
                log.append("reset")
            elif op == 8:
                got.flush()
                log.append("flush")
            else:
                _check(got, ref, (seed, w, h, log))
                log.append("read")
            assert got.sample_count == ref.sample_count, (seed, w, h, log)
        _check(got, ref, (seed, w, h, log))
        got.close()
        ref.close()
    dev.close()
