#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (whole job) and ms/spp on the dragon-sized scene
at 1920x1080, samples-per-pixel sharded over the GPUs of one node.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed, backend nccl = RCCL). With --gpus N > 1
and no WORLD_SIZE in the environment this process only starts the N ranks
(python -m torch.distributed.run ... bench.py <same arguments>), relays rank
0's JSON line and exits with the launcher's code; it never touches the GPU
itself. A step is one pass of the render hot path over the whole frame:
--spp-per-step consecutive dispatches (frame indices) per GPU (weak scaling), or
--total-spp dispatches split over the GPUs (strong scaling). In both cases GPU
g of G renders the step's frame indices g, g+G, g+2G, ... into an un-normalised
sum; after the last step the sums are reduced to rank 0 over RCCL and tone-mapped
once. The scene and its BVH are resident in HBM before the timed region starts;
nothing crosses PCIe inside it. Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic bytes: 80 B per wide node fetched (a fetch shared by the lanes of a
# wave counts once), 48 B per triangle tested, 80 B per shaded hit (the triangle's
# shading record), 64 B per escaped path (4 RGBA32F texels), 36 B per
# pixel per dispatch (accumulator read + write, RGBA8 write).
B_NODE, B_TRI, B_HIT, B_MISS, B_PIXEL = 80, 48, 80, 64, 36
B_QUEUE, B_RADIANCE = 56, 16  # a live path's queue entry, a finished path's radiance
HBM_PEAK_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0  # what a streaming copy reaches of it
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth of the 8 XCDs
MALL_GATHER_GBS = 8600.0  # random-row gather out of the Infinity Cache
# Where the node + triangle working set of a scene lives. Every XCD reads the whole
# tree, so the L2 that serves a fetch holds one XCD's 4 MiB of it, not the chip's 32.
L1_BYTES, L2_XCD_BYTES, MALL_BYTES = 32 << 10, 4 << 20, 256 << 20
# sources whose text decides what the render kernels and the tree builder do: a counter
# profile belongs to the build these hash to
KERNEL_SOURCES = ("ptc_wavefront.hip", "ptc_render.hip", "ptc_device.h", "ptc_device_math.h", "ptc_bvh_build.hip")


def kernel_source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "path_tracer_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def memory_level(working_set_bytes: float):
    """(level, that level's peak in GB/s) for a node + triangle working set."""
    if working_set_bytes <= L1_BYTES:
        return "l1", None  # re-fetches never leave the CU; no chip-wide byte rate bounds them
    if working_set_bytes <= L2_XCD_BYTES:
        return "l2", L2_PEAK_GBS
    if working_set_bytes <= MALL_BYTES:
        return "mall", MALL_GATHER_GBS
    return "hbm", HBM_ACHIEVABLE_GBS


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=256,
                    help="dispatches per GPU per step, all in flight at once (the persistent kernel's tail of long "
                         "paths is paid once per launch); weak scaling")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="dispatches per step over ALL GPUs (strong scaling: 4096 with --gpus 8 --steps 1 is the "
                         "headline configuration as written, 512 per GPU); overrides --spp-per-step")
    ap.add_argument("--max-paths", type=int, default=0,
                    help="bound of the wavefront workspace in paths (76 B each, 156 with --wavefront-bounces); 0 = one step's worth")
    ap.add_argument("--scene", default="dragon",
                    help="dragon | ref4 | cornell | torus10m | c5 | dragon_small | path of an OBJ / PLY file")
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
    ap.add_argument("--optimize-rounds", type=int, default=0,
                    help="reinsertion rounds of the tree builder (0 = the library's default, -1 = none)")
    ap.add_argument("--two-level", action="store_true",
                    help="one tree per mesh + instance list instead of one tree over all instanced triangles")
    ap.add_argument("--regroup", action="store_true",
                    help="hand the paths that survive the first bounce to the persistent launch sorted by ray origin "
                         "cell and octant")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo, work out the frames each would render, reduce a "
                         "stand-in for the film (one count per frame index) onto rank 0, which checks that the "
                         "union is the single-GPU schedule and prints the plan")
    return ap.parse_args()


def self_launch(n_gpus: int) -> int:
    """Starts the ranks as children of this (GPU-free) process and relays rank
    0's JSON line. Nothing here imports torch or touches HIP."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "16")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT)
    for line in proc.stdout:
        if line.startswith("{"):
            sys.stdout.write(line)  # rank 0's result line
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


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
    if name == "c5":
        return (scenes.dragon_in_cornell(scenes.c5_sky_from_hdr(env_res, env_res // 2)),
                "dragon stand-in (871,402 triangles, specular) inside the Cornell box (walls + light), "
                "equirectangular sky read from a Radiance .hdr file")
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
        # the reference itself on Mesa lavapipe would be the preferred baseline: neither a Vulkan loader /
        # ICD, glslc nor the reference's fetched dependencies exist on this machine, so it cannot run here
        "reference_lavapipe": "unavailable",
        # yardstick for config.bvh_sah_cost: the same cost for the 8-wide collapse of a top-down binned-SAH tree
        "oracle_wide_sah_cost": round(orc.wide_sah_cost()[0], 2),
    }


def pmc_entry(scene: str, width: int, height: int, spp: int, max_depth: int, kernel: str, two_level: bool = False,
              regroup: bool = False):
    """Counter-derived figures of the dominant kernel for this configuration,
    from the newest profiles/r*_pmc_index.json (written by
    tools/update_profiles.py from a tools/profile_pmc.sh run); None if that
    configuration was not profiled. An entry taken with other kernel sources
    than the ones in this tree comes back as {"stale": ...}: its counters say
    nothing certain about this build."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_index.json")), reverse=True):
        for e in json.load(open(path)):
            if (e.get("scene"), e.get("width"), e.get("height"), e.get("spp_per_step"), e.get("max_depth"),
                    e.get("kernel"), bool(e.get("two_level", False)), bool(e.get("regroup", False))) == (
                        scene, width, height, spp, max_depth, kernel, two_level, regroup):
                if e.get("kernel_source_hash") != kernel_source_hash():
                    return {"stale": f"{e.get('source')} was taken with kernel sources {e.get('kernel_source_hash')}, "
                                     f"this tree has {kernel_source_hash()}"}
                return e
    return None


def rank_frames(step: int, step_total: int, rank: int, world: int, spp: int):
    """(first frame index, stride, count) of the dispatches rank renders in a
    step: dispatch j is frame index step * step_total + rank + j * world."""
    from path_tracer_amd import parallel
    return parallel.frame_schedule(rank, world, step * step_total, spp)


def dry_ranks(args) -> None:
    """The rank code path without a GPU: rendezvous (gloo), this rank's share of
    every step, one reduce of a stand-in film (a count per frame index) onto rank
    0, the max-over-ranks timing exchange; rank 0 alone prints."""
    import torch
    import torch.distributed as dist
    from path_tracer_amd import parallel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    if world > 1:
        dist.init_process_group("gloo")
    step_total = args.total_spp if args.total_spp > 0 else args.spp_per_step * world
    if step_total < world:
        raise SystemExit("bench.py: fewer dispatches per step than GPUs")
    spp = parallel.split_spp(step_total, world)[rank]
    film = torch.zeros(args.steps * step_total, dtype=torch.float32)
    t0 = time.perf_counter()
    mine = []
    for k in range(args.steps):
        first, stride, count = rank_frames(k, step_total, rank, world, spp)
        frames = [first + j * stride for j in range(count)]
        mine += frames
        film[frames] += 1.0
    if world > 1:
        dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, -elapsed], dtype=torch.float64)
    plans = [None] * world
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_gather_object(plans, {"rank": rank, "spp_per_step": spp, "frames": mine})
    else:
        plans = [{"rank": rank, "spp_per_step": spp, "frames": mine}]
    if rank == 0:
        print(json.dumps({"dry_ranks": True, "n_gpus": world, "steps": args.steps, "step_total": step_total,
                          "scaling": "strong" if args.total_spp > 0 else "weak",
                          "every_frame_rendered_once": bool((film == 1.0).all()),
                          "frames_total": int(film.numel()), "ranks": plans,
                          "elapsed_spread_ms": round((float(t[0]) + float(t[1])) * 1e3, 3),
                          "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.dry_ranks:
        return dry_ranks(args)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}")
    n_dev = torch.cuda.device_count()  # counting does not initialise the GPU
    # Rehearsal of the multi-rank path on a machine with fewer GPUs than ranks (a test hook, PTC_BENCH_REHEARSE=1):
    # the ranks share the GPUs there are, rendezvous over gloo and take their collectives through host memory.
    # Every line of the rank logic runs; the timings mean nothing.
    rehearse = os.environ.get("PTC_BENCH_REHEARSE") == "1" and world > 1
    if n_dev < world and not rehearse:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} GPUs, this machine has {n_dev}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    local_rank = local_rank % n_dev if rehearse else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def reduce_to_rank0(tensor):
        if rehearse:
            host = tensor.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            if rank == 0:
                tensor.copy_(host)
        else:
            dist.reduce(tensor, dst=0, op=dist.ReduceOp.SUM)

    def all_reduce(tensor, op):
        if rehearse:
            host = tensor.cpu()
            dist.all_reduce(host, op=op)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, op=op)

    from path_tracer_amd import parallel
    from path_tracer_amd import renderer as R

    scene, scene_desc = build_scene(args.scene, args.env_res)
    ctx = R.create_context(local_rank)
    W, H = args.width, args.height
    cam = R.default_camera(W, H)
    # the film's accumulator lives in a torch tensor so RCCL can reduce it in place
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    rr = R.create_render_resources(ctx, W, H, scene, accum_dev=accum.data_ptr(),
                                   search_radius=args.search_radius, two_level=args.two_level,
                                   optimize_rounds=R.OPTIMIZE_NONE if args.optimize_rounds < 0 else args.optimize_rounds)
    rr.max_depth = args.max_depth
    rr.accumulate = R.ACCUM_SUM
    rr.wavefront_bounces = args.wavefront_bounces
    rr.regroup = int(args.regroup)
    # dispatches per step: over all ranks, and this rank's share (rank g takes the step's frame indices g, g+G, ...)
    step_total = args.total_spp if args.total_spp > 0 else args.spp_per_step * world
    spp = parallel.split_spp(step_total, world)[rank]
    if step_total < world:
        raise SystemExit("bench.py: fewer dispatches per step than GPUs")
    # workspace for a whole step in flight: 256 spp at 1080p are 2^29 paths, 40 GB of the 288 GB at 76 B per path
    # (156 B when bounces run as launch pairs; bounded by half of the free device memory: larger steps run in
    # several batches)
    per_path = 160 if args.wavefront_bounces > 0 else 80
    fits = int(torch.cuda.mem_get_info(local_rank)[0] * 0.5 / per_path)
    rr.max_paths_in_flight = min(args.max_paths or spp * W * H, fits, 2 ** 32 - 1)
    info = rr.scene.info
    integrator = {"megakernel": R.INTEGRATOR_MEGAKERNEL, "wavefront": R.INTEGRATOR_WAVEFRONT,
                  "stream": R.INTEGRATOR_STREAM}[args.integrator]

    def step(k: int):
        # dispatch j of step k on this rank renders frame index k * step_total + rank + j * world
        first, stride, count = rank_frames(k, step_total, rank, world, spp)
        rr.render(cam, frame=first, samples_per_frame=1, n_frames=count, frame_stride=stride)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Instrumented passes (untimed; counting kernels walk the tree in the same order as the oracle's traversal
    # of this BVH): nodes visited and triangles tested for all rays and for the first bounce alone with the
    # megakernel (the difference is what the persistent drain launch has to fetch), and the first bounce again
    # with the wavefront integrator's own primary kernel, whose waves share most node fetches.
    def instrumented(integ, depth):
        rr.integrator, rr.instrument, rr.max_depth = integ, True, depth
        R.reset_render(rr)
        ctx.reset_stats()
        step(0)
        ctx.synchronize()
        s = ctx.stats()
        return dict(rays=s.rays, paths=s.paths, nodes=s.nodes_visited, tris=s.tris_tested,
                    uniform_nodes=s.nodes_wave_uniform, uniform_fetches=s.uniform_fetches)

    inst_all = instrumented(R.INTEGRATOR_MEGAKERNEL, args.max_depth)
    inst_first = instrumented(R.INTEGRATOR_MEGAKERNEL, 1)
    # rays of the second bounce = camera paths that survive the first = entries of the persistent launch's queue
    inst_second = instrumented(R.INTEGRATOR_MEGAKERNEL, 2) if args.max_depth > 1 else inst_first
    inst_primary = instrumented(R.INTEGRATOR_WAVEFRONT, 1) if args.integrator == "wavefront" else inst_first
    rr.instrument, rr.max_depth, rr.integrator = False, args.max_depth, integrator
    R.reset_render(rr)

    for k in range(args.warmup):
        step(k)
    if world > 1 and args.warmup > 0:
        # the collective's own warm-up (channel set-up for a buffer of this size), on a scratch tensor
        reduce_to_rank0(torch.zeros_like(accum))
    R.reset_render(rr)
    ctx.reset_stats()
    ctx.render_times()
    ctx.phase_times()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    ctx.synchronize()
    t_steps = time.perf_counter() - t0
    reduce_ms = None
    if world > 1:
        # the one collective of the run: per-rank sums -> rank 0 (timed on its own as well)
        t_red = time.perf_counter()
        reduce_to_rank0(accum)
        torch.cuda.synchronize()
        reduce_ms = (time.perf_counter() - t_red) * 1e3
    sync_all()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    phases = ctx.phase_times()  # HIP events recorded inside the timed region, on the library's stream
    t = torch.tensor([elapsed, float(st.rays), float(st.paths)], dtype=torch.float64,
                     device=f"cuda:{local_rank}")
    # each rank's own time to finish its steps (before the reduce and the closing barrier): the spread over the
    # ranks is the tail imbalance of the sharding
    own = torch.tensor([t_steps, -t_steps], dtype=torch.float64, device=f"cuda:{local_rank}")
    if world > 1:
        tmax = t.clone()
        all_reduce(tmax, dist.ReduceOp.MAX)
        all_reduce(t, dist.ReduceOp.SUM)
        all_reduce(own, dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
    total_rays, total_paths = float(t[1]), float(t[2])
    rank_elapsed_max_ms, rank_elapsed_min_ms = float(own[0]) * 1e3, -float(own[1]) * 1e3

    if rank == 0:
        rr.resolve(sum_mode=True)
        if args.save_png:
            msg = R.write_to_png(ctx, rr, args.save_png)
            if msg:
                print(msg, file=sys.stderr)
        total_spp = args.steps * step_total
        A, F1, F2, P = inst_all, inst_first, inst_second, inst_primary
        # dominant kernel: the one with the largest share of device time
        dom = max(phases, key=lambda k: phases[k][0])
        dom_ms, dom_n = phases[dom]
        ws_bytes = info.n_nodes * B_NODE + info.n_bvh_triangles * B_TRI
        level, level_peak = memory_level(ws_bytes)
        # Bytes a launch cannot avoid moving to and from memory whatever the caches hold: a queue entry in per
        # path that enters the persistent launch, a radiance out per path that ends in the kernel, four sky texels
        # per escaped path, the film per pixel and dispatch.
        survivors = (F2["rays"] - F1["rays"]) if args.max_depth > 1 else 0
        # node bytes: a fetch shared by a wave is billed once (the drain's waves hold unrelated rays and never
        # share; the primary kernel's and the megakernel's do)
        node_fetches = lambda c: c["nodes"] - c["uniform_nodes"] + c["uniform_fetches"]
        if dom == "drain":
            # bounces >= 1 only: the rays, nodes and triangles beyond the first bounce; every such ray ends in
            # a sky lookup (64 B) or a shaded hit (80 B gather, billed as 64)
            rays_d, nodes_d, tris_d = A["rays"] - F1["rays"], A["nodes"] - F1["nodes"], A["tris"] - F1["tris"]
            fetches_d = nodes_d
            alg_bytes = fetches_d * B_NODE + tris_d * B_TRI + rays_d * B_MISS
            compulsory = survivors * (B_QUEUE + B_RADIANCE + B_MISS)
        elif dom == "primary":
            rays_d, nodes_d, tris_d, fetches_d = P["rays"], P["nodes"], P["tris"], node_fetches(P)
            # every camera path ends its first bounce in a sky lookup (64 B) or a hit gather (80 B, billed as 64)
            # and stores its radiance (16 B) or its queue entry (56 B, billed as 16)
            alg_bytes = fetches_d * B_NODE + tris_d * B_TRI + P["paths"] * (B_MISS + 16)
            compulsory = (P["paths"] - survivors) * (B_MISS + B_RADIANCE) + survivors * B_QUEUE
        else:
            rays_d, nodes_d, tris_d, fetches_d = A["rays"], A["nodes"], A["tris"], node_fetches(A)
            alg_bytes = (fetches_d * B_NODE + tris_d * B_TRI + (A["rays"] - A["paths"]) * B_HIT +
                         A["paths"] * B_MISS + W * H * spp * B_PIXEL)
            compulsory = A["paths"] * B_MISS + W * H * spp * B_PIXEL
        launches_per_step = max(dom_n // max(args.steps, 1), 1)
        avg_ms = dom_ms / max(dom_n, 1)
        per_s = lambda nbytes: nbytes / launches_per_step / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else float("nan")
        # A tree that fits a CU's L1 or an XCD's L2 is fetched from memory once per cache, not once per visit:
        # billing 80 B per node visit there counts bytes that never leave the CU / XCD against HBM, which is how a
        # 32-triangle scene once showed more than the peak. Such lines bill the compulsory bytes.
        billed = compulsory if level in ("l1", "l2") else alg_bytes
        achieved = per_s(billed)
        pmc = pmc_entry(args.scene, W, H, spp, args.max_depth, dom, args.two_level, args.regroup) or {}
        pmc_note = pmc.get("stale")
        if pmc_note:
            pmc = {}
        traffic = pmc.get("hbm_bytes_per_launch")
        traffic_gbs = traffic / (avg_ms * 1e-3) / 1e9 if traffic and avg_ms > 0 else None
        # how busy the level that holds the working set is: bytes that reach it (counter traffic beyond L2 for
        # the Infinity Cache and HBM; the per-visit bytes, an upper bound, for an XCD's L2) against its peak
        if level in ("mall", "hbm"):
            frac_at_level = round(traffic_gbs / level_peak, 4) if traffic_gbs else None
        elif level == "l2":
            frac_at_level = round(per_s(alg_bytes) / level_peak, 4)
        else:
            frac_at_level = None
        # rates of the two kinds of rays: coherent camera rays (primary pass) and everything after (drain)
        rate = lambda n_rays, ph: (round(n_rays * args.steps / (phases[ph][0] * 1e-3) / 1e6, 1)
                                   if phases.get(ph, (0, 0))[0] > 0 else None)
        scene_label = "dragon" if args.scene == "dragon" else os.path.basename(args.scene)
        size_label = "1080p" if (W, H) == (1920, 1080) else f"{W}x{H}"
        out = {
            "metric": f"Mrays/s (whole node), {scene_label} {size_label}", "value": round(total_rays / elapsed / 1e6, 2),
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.total_spp > 0 else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "ms_per_spp": round(elapsed / total_spp * 1e3, 4),
            "reduce_ms": None if reduce_ms is None else round(reduce_ms, 3),
            # slowest and fastest rank's own time over the steps (before the reduce): the tail imbalance
            "rank_elapsed_ms": {"max": round(rank_elapsed_max_ms, 3), "min": round(rank_elapsed_min_ms, 3),
                                "spread": round(rank_elapsed_max_ms - rank_elapsed_min_ms, 3)},
            "rays_per_path": round(total_rays / max(total_paths, 1.0), 3),
            "primary_mrays_s": rate(P["rays"], "primary"),
            "incoherent_mrays_s": rate(A["rays"] - F1["rays"], "drain"),
            "config": {"workload": f"{scene_desc}; {W}x{H}, {step_total} spp per step over {world} GPU(s) "
                                   f"({spp} per GPU, all in flight: workspace of {rr.max_paths_in_flight:,} paths = "
                                   f"{rr.max_paths_in_flight * (per_path - 4) / 1e9:.0f} GB; the library's default bound is "
                                   f"2^29 paths), "
                                   f"max depth {args.max_depth}, {args.integrator}"
                                   f"{', two-level scene' if args.two_level else ''}"
                                   f"{', queue regrouped by ray origin and octant' if args.regroup else ''}, spp sharded "
                                   f"over {world} GPU(s), RCCL sum at the end"
                                   f"{' (REHEARSAL: ranks share a GPU, gloo through host memory; timings mean nothing)' if rehearse else ''}",
                       "triangles": info.n_triangles, "bvh_nodes": info.n_nodes,
                       "bvh_build_ms": round(info.build_ms, 2), "bvh_sah_cost": round(info.sah_cost, 2),
                       "bvh_optimize_rounds": "default (8)" if args.optimize_rounds == 0 else args.optimize_rounds},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         # which bytes `achieved` bills: per-visit ("algorithmic": 80 B per node visit, 48 B per
                         # triangle test, 64 B per ray's end) or, for a tree that stays in L1 / L2, "compulsory"
                         "billed": "compulsory" if level in ("l1", "l2") else "algorithmic",
                         # the cache level the node + triangle working set fits, that level's peak, and how much
                         # of it the launch uses
                         "level": level, "level_peak": level_peak, "frac_at_level": frac_at_level,
                         "working_set_bytes": int(ws_bytes),
                         # what the memory system beyond L2 actually moved (counters), against the HBM peak and
                         # against what a streaming copy reaches of it; an upper bound: Infinity-Cache hits count
                         "hbm_traffic_frac": round(traffic_gbs / HBM_PEAK_GBS, 4) if traffic_gbs else None,
                         "hbm_achievable_frac": round(traffic_gbs / HBM_ACHIEVABLE_GBS, 4) if traffic_gbs else None,
                         "algorithmic_gbs": round(per_s(alg_bytes), 1),
                         "l2_frac": round(per_s(alg_bytes) / L2_PEAK_GBS, 4),
                         "l2_hit_rate": pmc.get("l2_hit_rate"), "l1_hit_rate": pmc.get("l1_hit_rate"),
                         "td_busy": pmc.get("td_busy"), "ta_busy": pmc.get("ta_busy"),
                         "lane_util": pmc.get("lane_util"), "valu_busy": pmc.get("valu_busy"),
                         "pmc_source": pmc.get("source") or pmc_note,
                         "kernel": dom, "avg_launch_ms": round(avg_ms, 3),
                         "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": int(alg_bytes / launches_per_step),
                         "compulsory_bytes_per_launch": int(compulsory / launches_per_step),
                         "rays_per_launch": int(rays_d / launches_per_step),
                         "nodes_per_ray": round(nodes_d / max(rays_d, 1), 2),
                         "node_fetches_per_ray": round(fetches_d / max(rays_d, 1), 2),
                         "tris_per_ray": round(tris_d / max(rays_d, 1), 2),
                         # the coherent first bounce, counted in the primary kernel itself
                         "primary_nodes_per_ray": round(P["nodes"] / max(P["rays"], 1), 2),
                         "primary_node_fetches_per_ray": round(node_fetches(P) / max(P["rays"], 1), 3),
                         "primary_tris_per_ray": round(P["tris"] / max(P["rays"], 1), 2),
                         "phase_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in phases.items() if v[1]},
                         "note": f"node + triangle working set {ws_bytes / 1e6:.1f} MB "
                                 + {"l1": "fits a CU's 32 KB L1: no chip-wide byte rate bounds the kernel; frac bills "
                                          "the bytes that must move (queue entries, radiances, sky texels)",
                                    "l2": "fits one XCD's 4 MiB L2: frac bills the bytes that must move; frac_at_level "
                                          "is the per-visit byte rate against the L2s' aggregate peak",
                                    "mall": "fits the 256 MiB Infinity Cache: frac is the per-visit byte rate against "
                                            "the HBM peak the contract names, not a statement about HBM, which "
                                            "the fetches do not reach; frac_at_level is the counter traffic beyond "
                                            "L2 against the Infinity Cache's gather rate, hbm_achievable_frac the "
                                            "same traffic against what HBM streams",
                                    "hbm": "exceeds the 256 MiB Infinity Cache: HBM serves the misses; "
                                           "frac_at_level is the counter traffic beyond L2 against what HBM "
                                           "streams"}[level]},
        }
        out["cpu_baseline"] = None  # reported by the single-GPU run only
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
