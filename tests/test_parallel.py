"""spp sharding: schedule arithmetic, and a world_size-2 gloo run on the CPU in
which each rank renders its share with the oracle (standing in for the GPU),
the sums are reduced, and the result is compared with the single-process
render of the same frame schedule."""
import os
import subprocess
import sys

import numpy as np

from path_tracer_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_schedule_partitions_the_single_gpu_schedule():
    for world in (1, 2, 4, 8):
        frames = sorted(f for r in range(world) for f in parallel.frames_of(r, world, 100, 16))
        assert frames == list(range(100, 100 + 16 * world))
    assert parallel.split_spp(10, 4) == [3, 3, 2, 2] and sum(parallel.split_spp(4096, 8)) == 4096


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["PT_ROOT"])
import numpy as np, torch, torch.distributed as dist
from oracle import pt_oracle as O
from path_tracer_amd import parallel, renderer as R, scenes
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sc = scenes.cornell_box()
orc = O.Scene(*sc.oracle_args())
w, h, spp = 24, 16, 3
cam = R.default_camera(w, h)
acc = None
for k, f in enumerate(parallel.frames_of(rank, world, 7, spp)):
    acc, _, _ = orc.render(O.make_push(cam.as_dict(), frame=f, sample_count=k), w, h, accum=acc,
                           accumulate_sum=True, use_bvh=False, want_rgba8=False)
t = torch.from_numpy(acc)
parallel.reduce_sum(t, dst=0)
if rank == 0:
    ref, _, _ = orc.render(O.make_push(cam.as_dict(), frame=7), w, h, n_frames=spp * world,
                           accumulate_sum=True, use_bvh=False, want_rgba8=False)
    got = t.numpy()
    assert (got[..., 3] == spp * world).all()
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6), np.abs(got - ref).max()
    mean = parallel.normalise(t).numpy()
    assert np.allclose(mean[..., :3], ref[..., :3] / (spp * world), rtol=1e-5, atol=1e-6)
    print("PARALLEL_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_reduce_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PT_ROOT=ROOT, OMP_NUM_THREADS="2")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "PARALLEL_OK" in out.stdout


def test_bench_gpus_n_starts_its_own_ranks_and_fails_cleanly_without_the_gpus():
    """`python bench.py --gpus 2` typed as is (no launcher, no WORLD_SIZE) starts two
    ranks itself; on a machine with fewer than 2 GPUs every rank says so and the
    parent returns the launcher's non-zero code without a JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("machine has 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "needs 2 GPUs" in out.stderr, out.stderr[-2000:]
    assert "launch with torch.distributed.run" not in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_strong_scaling_split_covers_the_step_once():
    """--total-spp: rank g of G renders frames g, g+G, ... of the step's T frame
    indices, split_spp(T, G)[g] of them; the union is 0..T-1 exactly once."""
    for total, world in ((4096, 8), (10, 4), (7, 2), (5, 5)):
        counts = parallel.split_spp(total, world)
        frames = sorted(r + j * world for r in range(world) for j in range(counts[r]))
        assert frames == list(range(total))


def test_bench_rank_code_path_without_a_gpu():
    """bench.py --gpus 2 --dry-ranks: the launcher starts two ranks, which
    rendezvous (gloo), work out their shares of every step with the function
    the real run uses, reduce a stand-in film and exchange their timings; rank 0
    alone prints one line. The union of the ranks' frames is the single-GPU
    schedule, each frame once, for a step that does not divide evenly."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-ranks", "--total-spp", "5",
                          "--steps", "3"], capture_output=True, text=True, timeout=300, cwd=ROOT,
                         env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["step_total"] == 5
    assert j["every_frame_rendered_once"] and j["frames_total"] == 15
    assert "cpu_baseline" in j and j["cpu_baseline"] is None  # reported by the single-GPU run only, and said so
    frames = {r["rank"]: r["frames"] for r in j["ranks"]}
    assert frames[0] == [0, 2, 4, 5, 7, 9, 10, 12, 14] and frames[1] == [1, 3, 6, 8, 11, 13]
    assert [r["spp_per_step"] for r in j["ranks"]] == [3, 2]
    # weak scaling: every rank renders --spp-per-step dispatches per step
    from bench import rank_frames
    assert rank_frames(2, 8, 1, 4, 2) == (17, 4, 2)
