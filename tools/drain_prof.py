#!/usr/bin/env python3
"""where the persistent kernel's waves spend their cycles (PTC_DRAIN_PROFILE=1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTC_DRAIN_PROFILE"] = "1"
import numpy as np, torch
torch.cuda.is_available()
from path_tracer_amd import renderer as R, scenes
sc = scenes.dragon_scene(envmap=scenes.procedural_sky(2048, 1024))
ctx = R.create_context(0)
w, h = 1920, 1080
cam = R.default_camera(w, h)
rr = R.create_render_resources(ctx, w, h, sc)
rr.integrator, rr.accumulate = R.INTEGRATOR_WAVEFRONT, R.ACCUM_SUM
rr.max_paths_in_flight = 256 * w * h
rr.render(cam, frame=0, n_frames=256); ctx.synchronize()
out = (C.c_ulonglong * 16)()
R.lib().ptc_debug_drain_profile(out)
ctx.phase_times()
rr.render(cam, frame=256, n_frames=256); ctx.synchronize()
R.lib().ptc_debug_drain_profile(out)
p = np.array(out[:10], dtype=np.float64)
names = ["loop top / ballots", "shading (incl. park)", "pull", "start + pop", "node visit", "triangle tests (1-2)", "further triangle rounds"]
tot = p[:7].sum()
print("drain ms", ctx.phase_times()["drain"])
for n, v in zip(names, p[:7]):
    print(f"{n:28s} {v / tot:6.1%}   {v / p[9]:8.0f} cycles per iteration")
print(f"iterations {p[9]:.3g}, shading phases {p[8]:.3g} ({p[8] / p[9]:.3f} per iteration), extra tri rounds {p[7]:.3g} ({p[7] / p[9]:.3f} per iteration)")
print(f"cycles per iteration {tot / p[9]:.0f}; per shading phase {p[1] / max(p[8], 1):.0f}; per extra round {p[6] / max(p[7], 1):.0f}")
