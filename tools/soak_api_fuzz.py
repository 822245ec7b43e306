#!/usr/bin/env python3
"""Runs the random call-sequence test of tests/test_gpu_api_fuzz.py over many more
seeds than the test suite does. Usage (on the GPU box): soak_api_fuzz.py [first seed] [count]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.is_available()
from path_tracer_amd import renderer as R
from tests.test_gpu_api_fuzz import test_random_call_sequences_equal_one_by_one_megakernel_dispatches as run

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = R.create_context(0)
t0 = time.time()
for seed in range(first, first + count):
    run.__wrapped__(ctx, seed) if hasattr(run, "__wrapped__") else run(ctx, seed)
    if (seed - first) % 20 == 19:
        print(f"seeds {first}..{seed} ok, {time.time() - t0:.0f} s", flush=True)
print("all", count, "seeds ok")
