#!/usr/bin/env python3
"""Builds the adversarial triangle distributions of tests/test_gpu_builder_fuzz.py with both collapse
rules and prints nodes, depth and SAH cost, or the error. Usage: builder_probe.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.is_available()
from path_tracer_amd import renderer as R
from tests.test_gpu_builder_fuzz import _shapes, _soup
ctx = R.create_context(0)
for n in [int(a) for a in sys.argv[1:]] or [1500]:
    for name, p in _shapes(np.random.default_rng(n), n):
        sc = R.Scene([_soup(p)], [R.Instance(R.IDENTITY, 0, 1)], None)
        for collapse in (0, 1):
            try:
                dev = R.DeviceScene(ctx, sc, collapse=collapse)
                i = dev.info
                print(f"n {n:6d} {name:20s} collapse {collapse}: nodes {i.n_nodes:6d} depth {i.bvh_depth:3d} SAH {i.sah_cost:10.2f}")
                dev.close()
            except R.PtcError as e:
                print(f"n {n:6d} {name:20s} collapse {collapse}: {e}")
