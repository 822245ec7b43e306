#!/usr/bin/env python3
"""How much of the primary pass is ray generation and sky shading: renders the
headline configuration once with the reference camera and once with the camera
turned away from the scene (every camera ray misses), and prints the device
time per phase of a 64-spp step."""
import os, sys, torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import renderer as R, scenes
ctx = R.create_context(0)
sky = scenes.procedural_sky(2048, 1024)
sc = scenes.dragon_scene(envmap=sky)
W, H = 1920, 1080
for name, cam in (("default", R.default_camera(W, H)),
                  ("looking away", R.create_camera((0.0, 0.0, 3.5), (0.0, 0.0, 7.0), 1.0, float(W / H * np.tan(np.pi / 8)), float(np.tan(np.pi / 8))))):
    rr = R.create_render_resources(ctx, W, H, sc)
    rr.max_depth = 32
    for k in range(3):
        rr.render(cam, frame=k * 64, samples_per_frame=1, n_frames=64)
    ctx.synchronize(); ctx.phase_times()
    for k in range(4):
        rr.render(cam, frame=k * 64, samples_per_frame=1, n_frames=64)
    ctx.synchronize()
    print(name, {k: round(v[0] / 4, 3) for k, v in ctx.phase_times().items()})
    rr.close()
