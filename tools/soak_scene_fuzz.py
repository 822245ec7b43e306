#!/usr/bin/env python3
"""Runs the random-scene differential test of tests/test_gpu_fuzz.py (random
triangle soups, mirrored / sheared instances, both scene kinds, all integrators
against the oracle's brute force) over many more seeds than the test suite does.
Usage (on the GPU box): soak_scene_fuzz.py [first seed] [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.is_available()
from path_tracer_amd import renderer as R
from oracle import pt_oracle as oracle
from tests.test_gpu_fuzz import test_random_scene_bit_exact as run

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = R.create_context(0)
t0 = time.time()
for seed in range(first, first + count):
    for two_level in (False, True):
        run(ctx, oracle, seed, two_level)
    if (seed - first) % 10 == 9:
        print(f"seeds {first}..{seed} ok, {time.time() - t0:.0f} s", flush=True)
print("all", count, "seeds ok")
