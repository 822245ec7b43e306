#!/usr/bin/env python3
"""Throughput of the headline configuration against the number of dispatches
in flight per launch (the workspace bound max_paths_in_flight permitting)."""
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
for spp in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    rr = R.RenderResources(ctx, W, H, dev)
    rr.max_depth = 32
    rr.max_paths_in_flight = min(spp * W * H, 2 ** 32 - 1)
    reps = max(1, 256 // spp)
    rr.render(cam, frame=0, samples_per_frame=1, n_frames=spp)
    ctx.synchronize(); ctx.reset_stats(); ctx.render_times()
    for k in range(reps):
        rr.render(cam, frame=k * spp, samples_per_frame=1, n_frames=spp)
    ctx.synchronize()
    ms = sum(ctx.render_times())
    st = ctx.stats()
    print(f"{spp:4d} spp per launch: {ms / reps:8.3f} ms per call, {st.rays / ms / 1e3:8.1f} Mrays/s (device time)")
    rr.close()
