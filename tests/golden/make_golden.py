#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle (oracle/libpt_oracle.so)
and, for the camera, from the reference's own src/camera.cpp compiled into
oracle/_ref/libref_camera.so (only possible where /root/reference exists).

The reference holds no tests or golden vectors of its own, so these files pin
the oracle against regressions; only camera.json is pinned by reference code.
Floats are stored as hexadecimal bit patterns.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pt_oracle as O  # noqa: E402
from path_tracer_amd import renderer as R  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402


def hx(a):
    return [f"{int(v):08x}" for v in np.ascontiguousarray(a, np.float32).ravel().view(np.uint32)]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")
    print("wrote", name)


def hdr_fixture():
    # small .hdr files (both scanline encodings) with the texels the format defines
    import base64
    from oracle import hdr_image as H
    rng = np.random.default_rng(11)
    img = (rng.random((6, 20, 3), dtype=np.float32) * np.float32(8.0)).astype(np.float32)
    img[0] = np.float32(0.75)
    img[5, :4] = 0.0
    img[2, 3] = np.float32(40000.0)
    rgbe = H.float_to_rgbe(img)
    cases = []
    for name, rle in (("sky_rle.hdr", True), ("sky_flat.hdr", False)):
        data = H.encode(rgbe, rle=rle)
        cases.append({"name": name, "width": 20, "height": 6, "file_b64": base64.b64encode(data).decode(),
                      "rgba_bits": hx(H.rgbe_to_float(rgbe))})
    dump("hdr_image.json", {"cases": cases})


def main():
    O.build()
    hdr_fixture()
    # (i) hash / random streams, incl. the 1.0 return and the zero fixed point
    seeds = [1, 2463534242, 0xFFFFFFFF, 0, O.seed(0, 0), O.seed(12345, 77)]
    rng = {"hash": {str(x): O.hash32(x) for x in [0, 1, 2, 255, 65536, 0xDEADBEEF, 0xFFFFFFFF]},
           "seed": [[p, f, O.seed(p, f)] for p, f in [(0, 0), (1, 0), (0, 1), (307199, 999), (2073599, 4095)]],
           "streams": {}}
    for s in seeds:
        v, st = O.random_stream(s, 8)
        rng["streams"][str(s)] = {"values": hx(v), "states": [int(x) for x in st]}
    dump("rng.json", rng)

    # (ii) camera: reference camera.cpp vs restatement
    cams = []
    cases = [((0, 0, 3.5), (0, 0, 0), 1.0, 16 / 9 * 0.41421357, 0.41421357),
             ((1.5, 0.7, -2.0), (0.1, -0.2, 0.3), 1.3, 0.6, 0.45),
             ((-3, 4, 5), (0, 1, 0), 0.8, 0.5, 0.5)]
    for pos, tgt, sd, shw, shh in cases:
        entry = {"position": pos, "target": tgt, "sensor_distance": sd, "shw": shw, "shh": shh,
                 "restated": hx(R.create_camera(pos, tgt, sd, shw, shh).as_array())}
        if O.ref_camera_available():
            entry["reference"] = hx(O.ref_create_camera(pos, tgt, sd, shw, shh))
        cams.append(entry)
    dump("camera.json", {"source": "reference src/camera.cpp via oracle/_ref" if O.ref_camera_available()
                         else "restatement only", "cases": cams})

    # (iii) primary rays
    cam = R.default_camera(640, 480)
    cam.aperture_radius = 0.05
    rays = []
    for px, py, frame in [(0, 0, 0), (639, 479, 0), (320, 240, 5), (17, 401, 999)]:
        push = O.make_push(cam.as_dict(), frame=frame)
        o, d, s = O.camera_ray(push, 640, 480, px, py, O.seed(py * 640 + px, frame))
        rays.append({"px": px, "py": py, "frame": frame, "origin": hx(o), "dir": hx(d), "rng_after": s})
    dump("primary_rays.json", {"camera": hx(cam.as_array()), "width": 640, "height": 480, "rays": rays})

    # (iv) self-intersection offset across its branches
    offs = []
    for p, n in [((0.5, -0.5, 2.0), (0.0, 1.0, 0.0)), ((0.01, -0.02, 0.03), (0.577, 0.577, -0.577)),
                 ((-100.0, 1e-3, 0.03125), (-1.0, 0.0, 0.0)), ((0.0, -0.0, 1.0), (0.3, -0.9, 0.3)),
                 ((3.0, 3.0, 3.0), (0.001, -0.001, 0.9999))]:
        offs.append({"p": hx(p), "n": hx(n), "out": hx(O.offset_position_along_normal(p, n))})
    dump("offset.json", offs)

    # (v) material scatter
    mats = []
    d = np.array([0.3, -0.8, 0.52], np.float32)
    d /= np.linalg.norm(d)
    for kind in range(4):
        for nrm in [(0.0, 1.0, 0.0), (0.6, 0.8, 0.0)]:
            for seed in [1, 987654321]:
                r = O.material_scatter(kind, d, (0.2, -0.4, 0.7), nrm, seed)
                mats.append({"kind": kind, "dir": hx(d), "pos": hx((0.2, -0.4, 0.7)), "nrm": hx(nrm),
                             "seed": seed, "origin": hx(r["origin"]), "out_dir": hx(r["dir"]),
                             "color": hx(r["color"]), "emissivity": hx(r["emissivity"]),
                             "attenuation": hx([r["attenuation"]])[0], "rng": r["rng"]})
    dump("materials.json", mats)

    # (vi) tone map + UNORM8
    tm = []
    for rgb in [(0, 0, 0), (0.05, 0.05, 0.05), (0.5, 0.4, 0.3), (0.9, 0.8, 0.1), (5, 5, 5), (40, 1, 0.2)]:
        t = O.tonemap(rgb)
        a = O.tonemap(rgb, O.TONEMAP_ACES)
        tm.append({"rgb": hx(rgb), "out": hx(t), "unorm8": [O.unorm8(float(x)) for x in t],
                   "aces": hx(a), "aces_unorm8": [O.unorm8(float(x)) for x in a]})
    dump("tonemap.json", {"tonemap": tm,
                          "unorm8": [[f"{np.float32(x).view(np.uint32):08x}", O.unorm8(float(np.float32(x)))]
                                     for x in [-1.0, 0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 0.5, 1.0, 2.0, float("nan")]]})

    # (vii) brute-force hits + (viii) small accumulators
    sc = scenes.cornell_box(tall_kind=1, short_kind=3)
    orc = O.Scene(*sc.oracle_args())
    rays_in = scenes.random_rays(64, seed=9)
    hits, _ = orc.intersect(rays_in, use_bvh=False)
    dump("cornell_hits.json", {"rays": hx(rays_in), "t": hx(hits["t"]), "u": hx(hits["u"]), "v": hx(hits["v"]),
                               "id": [int(x) for x in hits["id"]]})
    cam = R.default_camera(32, 32)
    acc, img, st = orc.render(O.make_push(cam.as_dict(), frame=3), 32, 32, n_frames=4, max_depth=8, use_bvh=False)
    np.savez_compressed(os.path.join(HERE, "cornell_32x32_4spp.npz"), accum_bits=acc.view(np.uint32), rgba8=img,
                        rays=np.uint64(st.rays))
    # the same dispatches in the bounce-count debug view (shader.rgen:195-199, the branch under "#else")
    accd, imgd, std = orc.render(O.make_push(cam.as_dict(), frame=3), 32, 32, n_frames=4, max_depth=8, use_bvh=False,
                                 debug_view=1)
    np.savez_compressed(os.path.join(HERE, "cornell_debug_view_32x32_4spp.npz"), accum_bits=accd.view(np.uint32),
                        rgba8=imgd, rays=np.uint64(std.rays))
    sc2 = scenes.ref4_scene(scenes.knot_mesh(nu=48, nv=12), scenes.procedural_sky(32, 16))
    orc2 = O.Scene(*sc2.oracle_args())
    acc2, img2, st2 = orc2.render(O.make_push(cam.as_dict(), frame=0, samples_per_frame=2), 32, 32, n_frames=2,
                                  use_bvh=False)
    np.savez_compressed(os.path.join(HERE, "ref4_32x32_4spp.npz"), accum_bits=acc2.view(np.uint32), rgba8=img2,
                        rays=np.uint64(st2.rays))
    # (ix) the 4K configuration's scene in small: object in the Cornell box under a sky that went through a
    # Radiance file. The library's decoder produced the sky; the numpy restatement must decode the same file alike.
    import tempfile
    from oracle import hdr_image as H
    sky = scenes.c5_sky_from_hdr(64, 32)
    with tempfile.TemporaryDirectory() as d:
        scenes.write_hdr(os.path.join(d, "sky.hdr"), scenes.procedural_sky(64, 32))
        restated = H.decode(open(os.path.join(d, "sky.hdr"), "rb").read())
    assert np.array_equal(sky.view(np.uint32), np.ascontiguousarray(restated, np.float32).view(np.uint32))
    sc5 = scenes.dragon_in_cornell(sky, nu=64, nv=16)
    orc5 = O.Scene(*sc5.oracle_args())
    cam5 = R.default_camera(64, 36)
    push5 = O.make_push(cam5.as_dict(), frame=1)
    acc5, img5, st5 = orc5.render(push5, 64, 36, n_frames=4, use_bvh=False)
    ids5 = orc5.primary_hits(push5, 64, 36, use_bvh=False)
    np.savez_compressed(os.path.join(HERE, "c5_64x36_4spp.npz"), accum_bits=acc5.view(np.uint32), rgba8=img5,
                        rays=np.uint64(st5.rays), primary_ids=ids5)
    print("wrote npz fixtures")


if __name__ == "__main__":
    main()
