#!/usr/bin/env python3
"""Slot usage of the wide tree of a bench scene: children per node, triangles per leaf slot.
Usage (on the GPU box): node_fill.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.is_available()
import bench
from path_tracer_amd import renderer as R
scene, desc = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "dragon", 256)
ctx = R.create_context(0)
dev = R.DeviceScene(ctx, scene)
nodes, tris = dev.read_bvh()
meta = nodes[:, 24:32].astype(np.uint32)          # w1.z, w1.w: one byte per slot
count = meta >> 5
inner = ((meta >> 3) & 3) == 3
used = count > 0
print(f"{len(nodes)} nodes, {len(tris)} triangles, SAH {dev.info.sah_cost:.2f}")
print("slots used per node:", np.bincount(used.sum(1), minlength=9), "mean", used.sum(1).mean())
print("internal children per node: mean", (inner & used).sum(1).mean())
leaf = used & ~inner
print("leaf slots per node: mean", leaf.sum(1).mean(), "; triangles per leaf slot:", np.bincount(count[leaf], minlength=4)[1:4])
