#!/usr/bin/env python3
"""Where the persistent kernel's waves spend their cycles (PTC_DRAIN_PROFILE=1).
Usage: drain_prof.py [dragon | c5 | ref4 | ref4tl | cornell | torus10m] [dispatches in flight] [width height max_depth]
(ref4tl: the four-instance table as a two-level scene)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTC_DRAIN_PROFILE"] = "1"
import numpy as np, torch
torch.cuda.is_available()
from path_tracer_amd import renderer as R, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
sky = scenes.procedural_sky(2048, 1024)
sc = {"dragon": lambda: scenes.dragon_scene(envmap=sky), "c5": lambda: scenes.dragon_in_cornell(sky),
      "ref4": lambda: scenes.ref4_scene(scenes.knot_mesh(), sky), "ref4tl": lambda: scenes.ref4_scene(scenes.knot_mesh(), sky),
      "cornell": scenes.cornell_box,
      "torus10m": lambda: R.Scene([scenes.torus_mesh(3162, 1581)], [R.Instance(R.IDENTITY, 0, 0)], sky)}[name]()
ctx = R.create_context(0)
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
cam = R.default_camera(w, h)
rr = R.create_render_resources(ctx, w, h, sc, two_level=name == "ref4tl")
rr.integrator, rr.accumulate = R.INTEGRATOR_WAVEFRONT, R.ACCUM_SUM
if len(sys.argv) > 5:
    rr.max_depth = int(sys.argv[5])
rr.max_paths_in_flight = spp * w * h
print(name, "nodes", rr.scene.info.n_nodes, "depth", rr.scene.info.bvh_depth, "SAH", round(rr.scene.info.sah_cost, 2))
rr.render(cam, frame=0, n_frames=spp); ctx.synchronize()
out = (C.c_ulonglong * 20)()
R.lib().ptc_debug_drain_profile(out)
ctx.phase_times()
ctx.reset_stats()
rr.render(cam, frame=spp, n_frames=spp); ctx.synchronize()
st = ctx.stats()
print("rays", st.rays, "paths", st.paths)
R.lib().ptc_debug_drain_profile(out)
p = np.array(out[:10], dtype=np.float64)
print(f"most iterations of one wave {out[10]}, most cycles of one wave {out[11]:.3g}, most iterations of one ray {out[12]}")
if out[14]:
    print(f"waves {out[14]}: mean life {out[13] / out[14] / out[11]:.3f} of the longest; after the queue ran dry for them "
          f"{out[15] / out[14] / out[11]:.3f} of the longest life on average")
names = ["loop top / ballots", "shading (incl. park)", "pull", "start + pop", "node visit", "triangle tests (1-2)", "further triangle rounds"]
tot = p[:7].sum()
print("drain ms", ctx.phase_times()["drain"])
for n, v in zip(names, p[:7]):
    print(f"{n:28s} {v / tot:6.1%}   {v / p[9]:8.0f} cycles per iteration")
print(f"iterations {p[9]:.3g}, shading phases {p[8]:.3g} ({p[8] / p[9]:.3f} per iteration), extra tri rounds {p[7]:.3g} ({p[7] / p[9]:.3f} per iteration)")
print(f"cycles per iteration {tot / p[9]:.0f}; per shading phase {p[1] / max(p[8], 1):.0f}; per extra round {p[6] / max(p[7], 1):.0f}")
drain_rays = st.rays - st.paths  # the first bounce runs in the primary pass
print(f"stack: {out[16] / drain_rays:.2f} pushes per ray, {out[17] / drain_rays:.3f} of them to scratch memory "
      f"({out[17] / max(out[16], 1):.2%} of the pushes), {out[18] / drain_rays:.3f} pops from there per ray; deepest stack {out[19]}")
