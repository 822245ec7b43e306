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
