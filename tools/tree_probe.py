#!/usr/bin/env python3
"""Tree quality of the device builder against the clustering radius and the
collapse rule (greedy by area, or cost-optimal with several triangle costs):
build time, SAH cost of the wide tree, and nodes visited / triangles tested per
ray for a set of incoherent rays (reproducible visit order).
Usage: tree_probe.py [scene]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (loads the HIP runtime first)

torch.cuda.is_available()
from path_tracer_amd import renderer as R, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = {"dragon": lambda: scenes.dragon_scene(), "c5": lambda: scenes.dragon_in_cornell(),
      "cornell": lambda: scenes.cornell_box(),
      "torus10m": lambda: R.Scene([scenes.torus_mesh(3162, 1581)], [R.Instance(R.IDENTITY, 0, 0)], None)}[name]()
ctx = R.create_context(0)
rays = np.concatenate([scenes.random_rays(400000, seed=31), scenes.random_rays(100000, seed=32, radius=0.6)])
for radius, collapse, tri_cost in ([(r, 0, 0.0) for r in (4, 8, 16, 24, 32)] + [(16, 1, 0.0)] +
                                   [(16, 0, c) for c in (0.2, 0.3, 0.45, 0.6, 0.8, 1.0, 1.5)]):
    t0 = time.time()
    dev = R.DeviceScene(ctx, sc, search_radius=radius, collapse=collapse, tri_cost=tri_cost)
    wall = time.time() - t0
    hits, st = dev.trace(rays, count=True)
    hit = (hits["flat"] != R.NO_HIT).mean()
    print(f"{name} radius {radius:2d} {'greedy' if collapse else f'optimal, triangle cost {tri_cost or 0.8:.2f}'}: build {dev.info.build_ms:7.2f} ms (wall {wall * 1e3:7.1f}), nodes {dev.info.n_nodes}, "
          f"depth {dev.info.bvh_depth}, SAH {dev.info.sah_cost:.2f}, nodes/ray {st.nodes_visited / len(rays):.2f}, "
          f"tris/ray {st.tris_tested / len(rays):.2f}, hit rate {hit:.3f}", flush=True)
    dev.close()
ctx.close()
