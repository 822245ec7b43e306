#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (whole job) and ms/spp on the dragon-sized scene
at 1920x1080, samples-per-pixel sharded over the GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed, backend nccl = RCCL). A step is one
pass of the render hot path over the whole 1080p frame: --spp-per-step
consecutive dispatches (frame indices) per GPU. GPU g of G renders frame
indices g, g+G, g+2G, ... into an un-normalised sum; after the last step the
sums are reduced to rank 0 over RCCL and tone-mapped once. The scene and its
BVH are resident in HBM before the timed region starts; nothing crosses PCIe
inside it. Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic bytes: 80 B per wide node visited, 48 B per triangle tested, 84 B
# per shaded hit (3 indices + 3 positions + 3 normals), 64 B per escaped path
# (4 RGBA32F texels), 36 B per pixel per dispatch (accumulator read + write,
# RGBA8 write).
B_NODE, B_TRI, B_HIT, B_MISS, B_PIXEL = 80, 48, 84, 64, 36
HBM_PEAK_GBS = 8000.0


def build_scene(name: str, env_res: int):
    from path_tracer_amd import scenes
    sky = scenes.procedural_sky(env_res, env_res // 2)
    if name == "dragon":
        return scenes.dragon_scene(envmap=sky), "dragon stand-in: (2,3) torus knot, 871,402 triangles, specular, procedural sky"
    if name == "dragon_small":
        return scenes.dragon_scene(envmap=sky, nu=232, nv=117), "small knot, 54,288 triangles"
    if name == "ref4":
        # the reference's own scene: four instances of mesh 0, one per material (src/renderer.cpp:1147-1174)
        return (scenes.ref4_scene(scenes.knot_mesh(), sky),
                "reference instance table: 4 x 871,402-triangle knot (diffuse, specular, emissive, dielectric), procedural sky")
    if name == "cornell":
        return scenes.cornell_box(), "Cornell box, 32 triangles, diffuse + emissive"
    if name == "torus10m":
        from path_tracer_amd.renderer import IDENTITY, Instance, Scene
        return (Scene([scenes.torus_mesh(3162, 1581)], [Instance(IDENTITY, 0, 0)], sky),
                "fractal torus, 9,998,244 triangles, diffuse")
    if os.path.isfile(name):
        # a mesh file (OBJ / PLY, e.g. the real dragon_vrip.ply where it is at hand): mesh 0, imported with
        # the reference's post-processing steps, as the one specular instance of the headline configuration
        from path_tracer_amd.renderer import IDENTITY, KIND_SPECULAR, Instance, Scene, import_meshes
        mesh = import_meshes(name)[0]
        return (Scene([mesh], [Instance(IDENTITY, 0, KIND_SPECULAR)], sky),
                f"{os.path.basename(name)}: {mesh.indices.shape[0]:,} triangles, specular, procedural sky")
    raise SystemExit(f"unknown scene {name}")


def cpu_baseline(scene, cam, width, height, max_depth, seconds_target: float):
    """Oracle (scalar C port of the reference shaders, its own binned-SAH BVH,
    OpenMP over pixels) on this host's cores, on a bounded sample of the same
    workload: the same scene and camera at reduced resolution."""
    from oracle import pt_oracle as O
    t0 = time.time()
    orc = O.Scene(*scene.oracle_args())
    build_s = time.time() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = min(cores, int(os.environ.get("OMP_NUM_THREADS", cores)))
    w, h = width // 4, height // 4
    push = O.make_push(cam.as_dict(), frame=0)
    # calibrate with one dispatch, then size the sample for ~seconds_target
    t0 = time.time()
    _, _, st = orc.render(push, w, h, n_frames=1, max_depth=max_depth, n_threads=cores, want_rgba8=False)
    dt = max(time.time() - t0, 1e-3)
    n_frames = int(max(1, min(4096, seconds_target / dt)))
    t0 = time.time()
    _, _, st = orc.render(push, w, h, n_frames=n_frames, max_depth=max_depth, n_threads=cores,
                          want_rgba8=False)
    dt = time.time() - t0
    return {
        "value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{w}x{h}, {n_frames} spp, depth {max_depth}, same scene and camera; "
                  f"{st.rays} rays in {dt:.2f} s (oracle BVH build {build_s:.1f} s not counted)",
        "ms_per_spp_at_full_res": round(dt / n_frames * (width * height) / (w * h) * 1e3, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=256,
                    help="dispatches per GPU per step, all in flight at once (the persistent kernel's tail of long "
                         "paths is paid once per launch: 64 -> 9.4, 128 -> 9.9, 256 -> 10.2 Grays/s)")
    ap.add_argument("--max-paths", type=int, default=0,
                    help="bound of the wavefront workspace in paths (about 160 B each); 0 = one step's worth")
    ap.add_argument("--scene", default="dragon",
                    help="dragon | ref4 | cornell | torus10m | dragon_small | path of an OBJ / PLY file")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-depth", type=int, default=32)
    ap.add_argument("--integrator", default="wavefront", choices=["megakernel", "wavefront", "stream"])
    ap.add_argument("--env-res", type=int, default=2048)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--wavefront-bounces", type=int, default=0)
    ap.add_argument("--search-radius", type=int, default=0)
    ap.add_argument("--two-level", action="store_true",
                    help="one tree per mesh + instance list instead of one tree over all instanced triangles")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from path_tracer_amd import renderer as R

    scene, scene_desc = build_scene(args.scene, args.env_res)
    ctx = R.create_context(local_rank)
    W, H = args.width, args.height
    cam = R.default_camera(W, H)
    # the film's accumulator lives in a torch tensor so RCCL can reduce it in place
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    rr = R.create_render_resources(ctx, W, H, scene, accum_dev=accum.data_ptr(),
                                   search_radius=args.search_radius, two_level=args.two_level)
    rr.max_depth = args.max_depth
    rr.accumulate = R.ACCUM_SUM
    rr.integrator = R.INTEGRATOR_WAVEFRONT if args.integrator == "wavefront" else R.INTEGRATOR_MEGAKERNEL
    rr.wavefront_bounces = args.wavefront_bounces
    # workspace for a whole step in flight: 256 spp at 1080p are 2^29 paths, 85 GB of the 288 GB
    # (bounded by half of the free device memory at about 170 B per path: larger steps run in several batches)
    fits = int(torch.cuda.mem_get_info(local_rank)[0] * 0.5 / 170)
    rr.max_paths_in_flight = min(args.max_paths or args.spp_per_step * W * H, fits, 2 ** 32 - 1)
    info = rr.scene.info
    spp = args.spp_per_step

    def step(k: int):
        # dispatch j of step k on this rank renders frame index (k*spp + j)*world + rank
        rr.render(cam, frame=k * spp * world + rank, samples_per_frame=1, n_frames=spp,
                  frame_stride=world)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Instrumented passes (untimed, megakernel with counters: same rays, same
    # visit order as the oracle's traversal of this BVH): BVH nodes visited and
    # triangles tested for all rays and for the first bounce alone; the
    # difference is what the persistent drain launch has to fetch.
    def instrumented(depth):
        rr.integrator, rr.instrument, rr.max_depth = R.INTEGRATOR_MEGAKERNEL, True, depth
        R.reset_render(rr)
        ctx.reset_stats()
        step(0)
        ctx.synchronize()
        st_ = ctx.stats()
        return st_.rays, st_.paths, st_.nodes_visited, st_.tris_tested

    inst_all = instrumented(args.max_depth)
    inst_first = instrumented(1)
    rr.instrument, rr.max_depth = False, args.max_depth
    rr.integrator = {"megakernel": R.INTEGRATOR_MEGAKERNEL, "wavefront": R.INTEGRATOR_WAVEFRONT,
                     "stream": R.INTEGRATOR_STREAM}[args.integrator]
    R.reset_render(rr)

    for k in range(args.warmup):
        step(k)
    if world > 1 and args.warmup > 0:
        # the collective's own warm-up (channel set-up for a buffer of this size), on a scratch tensor
        dist.reduce(torch.zeros_like(accum), dst=0, op=dist.ReduceOp.SUM)
    R.reset_render(rr)
    ctx.reset_stats()
    ctx.render_times()
    ctx.phase_times()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    ctx.synchronize()
    reduce_ms = None
    if world > 1:
        # the one collective of the run: per-rank sums -> rank 0 (timed on its own as well)
        t_red = time.perf_counter()
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - t_red) * 1e3
    sync_all()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    phases = ctx.phase_times()  # HIP events recorded inside the timed region
    t = torch.tensor([elapsed, float(st.rays), float(st.paths)], dtype=torch.float64,
                     device=f"cuda:{local_rank}")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_rays, total_paths = float(t[1]), float(t[2])

    if rank == 0:
        rr.resolve(sum_mode=True)
        if args.save_png:
            msg = R.write_to_png(ctx, rr, args.save_png)
            if msg:
                print(msg, file=sys.stderr)
        total_spp = args.steps * spp * world
        rays_a, paths_a, nodes_a, tris_a = inst_all
        rays_f, _, nodes_f, tris_f = inst_first
        # dominant kernel: the one with the largest share of device time
        dom = max(phases, key=lambda k: phases[k][0])
        dom_ms, dom_n = phases[dom]
        if dom == "drain":
            # bounces >= 1 only: the rays, nodes and triangles beyond the first bounce
            rays_d, nodes_d, tris_d = rays_a - rays_f, nodes_a - nodes_f, tris_a - tris_f
            # every such ray ends in a shaded hit (84 B gather) or a sky lookup (64 B)
            alg_bytes = nodes_d * B_NODE + tris_d * B_TRI + rays_d * B_HIT
        else:
            rays_d, nodes_d, tris_d = rays_a, nodes_a, tris_a
            alg_bytes = (nodes_d * B_NODE + tris_d * B_TRI + (rays_a - paths_a) * B_HIT + paths_a * B_MISS +
                         W * H * spp * B_PIXEL)
        ws_mb = (info.n_nodes * B_NODE + info.n_triangles * B_TRI) / 1e6
        launches_per_step = max(dom_n // max(args.steps, 1), 1)
        avg_ms = dom_ms / max(dom_n, 1)
        achieved = alg_bytes / launches_per_step / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else float("nan")
        traffic = l2_hit = None
        tpath = os.path.join(ROOT, "profiles", "r01_dominant_kernel_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("kernel") == dom and tj.get("spp_per_step") == spp and tj.get("scene") == args.scene:
                traffic = tj.get("hbm_bytes_per_launch")
                l2_hit = round(tj["l2_hit_rate"], 3) if "l2_hit_rate" in tj else None
        out = {
            "metric": "Mrays/s (whole node), dragon 1080p", "value": round(total_rays / elapsed / 1e6, 2),
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ms_per_spp": round(elapsed / total_spp * 1e3, 4),
            "reduce_ms": None if reduce_ms is None else round(reduce_ms, 3),
            "rays_per_path": round(total_rays / max(total_paths, 1.0), 3),
            "config": {"workload": f"{scene_desc}; {W}x{H}, {spp} spp per GPU per step, "
                                   f"max depth {args.max_depth}, {args.integrator}"
                                   f"{', two-level scene' if args.two_level else ''}, spp sharded "
                                   f"over {world} GPU(s), RCCL sum at the end",
                       "triangles": info.n_triangles, "bvh_nodes": info.n_nodes,
                       "bvh_build_ms": round(info.build_ms, 2), "bvh_sah_cost": round(info.sah_cost, 2)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "l2_hit_rate": l2_hit,
                         "kernel": dom, "avg_launch_ms": round(avg_ms, 3),
                         "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": int(alg_bytes / launches_per_step),
                         "rays_per_launch": int(rays_d / launches_per_step),
                         "nodes_per_ray": round(nodes_d / max(rays_d, 1), 2),
                         "tris_per_ray": round(tris_d / max(rays_d, 1), 2),
                         "phase_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in phases.items() if v[1]},
                         "note": f"node + triangle working set {ws_mb:.0f} MB "
                                 + ("fits the 256 MB Infinity Cache; HBM peak is the stated bound, not the "
                                    "level that serves the fetches" if ws_mb < 256 else
                                    "exceeds the 256 MB Infinity Cache: HBM serves the misses")},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, cam, W, H, args.max_depth, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    rr.close()
    ctx.close()


if __name__ == "__main__":
    main()
