"""spp sharding with the device path under torch.distributed: two processes on
the one GPU each render their share of the frame schedule through libptcore
into a caller-owned torch accumulator (un-normalised sums), the sums are
reduced (gloo here; bench.py uses nccl = RCCL with one GPU per rank), rank 0
tone-maps, and the result equals the single-process render of the same schedule
up to the order of the fp32 additions."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["PT_ROOT"])
import numpy as np, torch, torch.distributed as dist
torch.cuda.init()
from path_tracer_amd import parallel, renderer as R, scenes
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = R.create_context(0)
sc = scenes.ref4_scene(scenes.knot_mesh(nu=96, nv=24), scenes.procedural_sky(64, 32))
w, h, spp = 160, 96, 6
cam = R.default_camera(w, h)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
rr = R.create_render_resources(ctx, w, h, sc, accum_dev=accum.data_ptr())
rr.accumulate = R.ACCUM_SUM
rr.max_depth = 8
f0, stride, n = parallel.frame_schedule(rank, world, 11, spp)
rr.render(cam, frame=f0, samples_per_frame=1, n_frames=n, frame_stride=stride)
ctx.synchronize()
host = accum.cpu()
parallel.reduce_sum(host, dst=0)
if rank == 0:
    accum.copy_(host)
    torch.cuda.synchronize()
    rr.resolve(sum_mode=True)
    got_img = rr.read_rgba8()
    single = R.create_render_resources(ctx, w, h, sc)
    single.accumulate = R.ACCUM_SUM
    single.max_depth = 8
    single.render(cam, frame=11, samples_per_frame=1, n_frames=spp * world)
    ref = single.read_accum()
    got = host.numpy()
    assert (got[..., 3] == spp * world).all()
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6), np.abs(got - ref).max()
    single.resolve(sum_mode=True)
    assert np.abs(got_img.astype(int) - single.read_rgba8().astype(int)).max() <= 1
    print("GPU_PARALLEL_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PT_ROOT=ROOT)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "GPU_PARALLEL_OK" in out.stdout


NCCL_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["PT_ROOT"])
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from path_tracer_amd import parallel, renderer as R, scenes
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
ctx = R.create_context(0)
sc = scenes.cornell_box(tall_kind=1, short_kind=3)
w, h, spp = 128, 72, 5
cam = R.default_camera(w, h)
accum = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
rr = R.create_render_resources(ctx, w, h, sc, accum_dev=accum.data_ptr())
rr.accumulate = R.ACCUM_SUM
rr.max_depth = 6
f0, stride, n = parallel.frame_schedule(0, 1, 3, spp)
rr.render(cam, frame=f0, n_frames=n, frame_stride=stride)
ctx.synchronize()
before = accum.clone()
# the collectives bench.py issues, on the film's own device memory, through RCCL
dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(accum, before) and t.tolist() == [1.5, 2.0]   # one rank: the sum is the rank's own film
rr.resolve(sum_mode=True)
img = rr.read_rgba8()
single = R.create_render_resources(ctx, w, h, sc)
single.max_depth = 6
single.render(cam, frame=3, n_frames=spp)
assert np.abs(img.astype(int) - single.read_rgba8().astype(int)).max() <= 1
assert (accum[..., 3] == spp).all()
print("NCCL_WS1_OK")
dist.destroy_process_group()
"""


def test_rccl_backend_reduces_the_film_in_place_with_one_rank(tmp_path):
    """The nccl (= RCCL) code path of bench.py on the one GPU a test box has:
    process group of one rank, reduce / all_reduce / barrier on device tensors,
    the film's accumulator among them, then the resolve."""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, PT_ROOT=ROOT)
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
         "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "NCCL_WS1_OK" in out.stdout


def test_bench_strong_scaling_line_on_one_gpu():
    """--total-spp on one GPU: the line says "strong" and the step renders exactly that many dispatches."""
    import json
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", "dragon_small", "--width", "160", "--height", "90",
         "--total-spp", "6", "--steps", "2", "--warmup", "0", "--no-cpu-baseline"],
        capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    j = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert j["scaling"] == "strong" and j["n_gpus"] == 1
    assert abs(j["ms_per_spp"] * 12 - j["ms_per_step"] * 2) < 1e-2 * j["ms_per_step"] + 1e-3


def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """Every line of bench.py's rank logic with two ranks (launcher, rendezvous,
    shares of a step, warm-up and final reduce onto rank 0, max / sum / spread
    exchanges, rank 0 alone printing), on the one GPU a test box has: with
    PTC_BENCH_REHEARSE=1 the ranks share the device and take their collectives
    through gloo and host memory. The film the two ranks sum is the one a single
    rank renders of the same schedule, up to fp32 summation order."""
    import json
    from PIL import Image
    import numpy as np
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        lines, images = [], []
        for gpus in (2, 1):
            png = os.path.join(d, f"ranks{gpus}.png")
            out = subprocess.run(
                [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--scene", "dragon_small", "--width", "160",
                 "--height", "90", "--total-spp", "6", "--steps", "2", "--warmup", "1", "--save-png", png],
                capture_output=True, text=True, timeout=600, cwd=ROOT,
                env=dict(os.environ, PTC_BENCH_REHEARSE="1", OMP_NUM_THREADS="4"))
            assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
            found = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
            assert len(found) == 1
            lines.append(json.loads(found[0]))
            images.append(np.asarray(Image.open(png)).astype(np.int32))
    two, one = lines
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["cpu_baseline"] is None
    assert two["reduce_ms"] is not None and two["rank_elapsed_ms"]["spread"] >= 0
    assert "REHEARSAL" in two["config"]["workload"] and "REHEARSAL" not in one["config"]["workload"]
    # the same 12 dispatches either way: the same rays, and pictures that differ by summation order at most
    assert abs(two["rays_per_path"] - one["rays_per_path"]) < 1e-9
    assert np.abs(images[0] - images[1]).max() <= 1
