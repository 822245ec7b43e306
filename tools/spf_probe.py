#!/usr/bin/env python3
"""The reference's own call pattern and its neighbourhood: n_frames dispatches
per call, each with samples_per_frame samples on one RNG stream per pixel (the
reference: one dispatch per call). Device time per call and Mrays/s for the
three integrators."""
import sys
import torch
torch.cuda.init()
sys.path.insert(0, '.')
from path_tracer_amd import renderer as R, scenes
ctx = R.create_context(0)
sc = scenes.dragon_scene(envmap=scenes.procedural_sky(2048, 1024))
W, H = 1920, 1080
cam = R.default_camera(W, H)
dev = R.DeviceScene(ctx, sc)
import itertools
for nf, spf in itertools.product((1, 2, 4, 8, 16), (1, 2, 4, 16, 64)):
    if nf * spf > 256:
        continue
    line = f"n_frames {nf:2d} spf {spf:3d}:"
    for name, integ in (("megakernel", R.INTEGRATOR_MEGAKERNEL), ("wavefront", R.INTEGRATOR_WAVEFRONT),
                        ("stream", R.INTEGRATOR_STREAM)):
        rr = R.RenderResources(ctx, W, H, dev)
        rr.max_depth = 32
        rr.integrator = integ
        reps = max(2, 64 // (spf * nf))
        rr.render(cam, frame=0, samples_per_frame=spf, n_frames=nf)
        ctx.synchronize(); ctx.reset_stats(); ctx.render_times()
        for k in range(reps):
            rr.render(cam, frame=(1 + k) * nf, samples_per_frame=spf, n_frames=nf)
        ctx.synchronize()
        ms = sum(ctx.render_times())
        line += f"  {name} {ms / reps:8.3f} ms/call {ctx.stats().rays / ms / 1e3:7.0f} Mrays/s"
        rr.close()
    print(line)
