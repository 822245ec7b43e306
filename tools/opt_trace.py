#!/usr/bin/env python3
"""Builds a bench scene with PTC_OPT_ROUNDS / PTC_OPT_TRACE from the environment and prints the tree's figures.
Usage: opt_trace.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.is_available()
import bench
from path_tracer_amd import renderer as R
scene, desc = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "dragon", 256)
ctx = R.create_context(0)
dev = R.DeviceScene(ctx, scene)
i = dev.info
print(f"nodes {i.n_nodes} depth {i.bvh_depth} SAH {i.sah_cost:.3f} build {i.build_ms:.2f} ms")
