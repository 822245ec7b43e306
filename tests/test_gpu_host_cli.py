"""The C++ host mirror (path_tracer_amd/host) end to end on the GPU: the
headless CLI renders, writes a PNG that decodes, and the two integrators give
the same file (they are bit-identical by construction)."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from path_tracer_amd import renderer as R
from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "path_tracer_amd", "host", "ptc_headless")


def run_cli(*args):
    out = subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_builtin_scene_both_integrators(tmp_path):
    a, b = str(tmp_path / "wf.png"), str(tmp_path / "mk.png")
    log = run_cli("--width", "96", "--height", "64", "--spp", "6", "--spf", "2", "--depth", "8", "--out", a)
    assert "samples 6" in log
    run_cli("--width", "96", "--height", "64", "--spp", "6", "--spf", "2", "--depth", "8", "--megakernel", "--out", b)
    ia, ib = np.asarray(Image.open(a)), np.asarray(Image.open(b))
    assert ia.shape == (64, 96, 4) and (ia[..., 3] == 255).all()
    assert np.array_equal(ia, ib)
    assert ia[..., :3].std() > 5  # an actual picture, not a constant


def test_progressive_preview_does_not_change_the_picture(tmp_path):
    a, b = str(tmp_path / "plain.png"), str(tmp_path / "preview.png")
    run_cli("--width", "64", "--height", "48", "--spp", "8", "--spf", "2", "--depth", "4", "--out", a)
    log = run_cli("--width", "64", "--height", "48", "--spp", "8", "--spf", "2", "--depth", "4", "--preview-every", "2",
                  "--out", b)
    assert log.count("preview at") == 3 and "preview at 6 samples" in log  # after 2, 4 and 6 of the 8 samples
    assert np.array_equal(np.asarray(Image.open(a)), np.asarray(Image.open(b)))
    c = str(tmp_path / "one_by_one.png")
    run_cli("--width", "64", "--height", "48", "--spp", "8", "--spf", "2", "--depth", "4", "--batch", "1", "--out", c)
    assert np.array_equal(np.asarray(Image.open(a)), np.asarray(Image.open(c)))  # batched calls = single calls


def test_obj_file_as_the_reference_scene(tmp_path, ctx):
    m = scenes.knot_mesh(nu=40, nv=10)
    obj = tmp_path / "knot.obj"
    with open(obj, "w") as f:
        for p in m.positions:
            f.write(f"v {p[0]:.9g} {p[1]:.9g} {p[2]:.9g}\n")
        for t in m.indices:
            f.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
    png = str(tmp_path / "knot.png")
    log = run_cli(str(obj), "--width", "80", "--height", "60", "--spp", "4", "--out", png)
    assert f"triangles {4 * m.indices.shape[0]}" in log  # four instances of mesh 0
    img = np.asarray(Image.open(png))
    assert img.shape == (60, 80, 4)
    # the Python mirror importing the same file and rendering the same schedule gives the same picture
    rr = R.create_render_resources(ctx, 80, 60, scenes.ref4_scene(R.import_meshes(str(obj))[0]))
    cam = R.default_camera(80, 60)
    for frame in range(4):
        rr.render(cam, frame=frame)
    assert np.array_equal(rr.read_rgba8(), img)
    rr.close()
    # --two-level: one tree for the mesh + four instances; again the mirror's picture
    png2 = str(tmp_path / "knot2.png")
    run_cli(str(obj), "--width", "80", "--height", "60", "--spp", "4", "--two-level", "--out", png2)
    rr = R.create_render_resources(ctx, 80, 60, scenes.ref4_scene(R.import_meshes(str(obj))[0]), two_level=True)
    for frame in range(4):
        rr.render(cam, frame=frame)
    img2 = np.asarray(Image.open(png2))
    assert np.array_equal(rr.read_rgba8(), img2)
    rr.close()
    assert np.mean(img2 != img) < 0.01  # the two hit definitions differ in rounding only
    # --bounce-view: the reference's debug view (shader.rgen:195-199), through the C++ mirror's debug_view
    png3 = str(tmp_path / "knot3.png")
    run_cli(str(obj), "--width", "80", "--height", "60", "--spp", "4", "--bounce-view", "--out", png3)
    rr = R.create_render_resources(ctx, 80, 60, scenes.ref4_scene(R.import_meshes(str(obj))[0]))
    rr.debug_view = 1
    for frame in range(4):
        rr.render(cam, frame=frame)
    img3 = np.asarray(Image.open(png3))
    assert np.array_equal(rr.read_rgba8(), img3) and not np.array_equal(img3, img)
    rr.close()


def test_env_map_from_hdr_file(tmp_path, ctx):
    """--env sky.hdr: the CLI's picture equals the Python mirror's render of the
    same scene with the same decoded map (both go through ptc_read_hdr)."""
    from oracle import hdr_image as H
    sky = scenes.procedural_sky(64, 32)
    hdr = tmp_path / "sky.hdr"
    hdr.write_bytes(H.encode(H.float_to_rgbe(sky[..., :3]), rle=True))
    decoded = R.read_hdr_image(str(hdr))
    assert decoded.shape == (32, 64, 4)
    # shared exponent, 8-bit mantissas: each channel is within 1/128 of the texel's largest channel
    assert np.all(np.abs(decoded[..., :3] - sky[..., :3]) <= sky[..., :3].max(axis=-1, keepdims=True) / 128)
    a, b = str(tmp_path / "sky.png"), str(tmp_path / "nosky.png")
    run_cli("--width", "64", "--height", "48", "--spp", "4", "--depth", "4", "--env", str(hdr), "--out", a)
    run_cli("--width", "64", "--height", "48", "--spp", "4", "--depth", "4", "--out", b)
    assert not np.array_equal(np.asarray(Image.open(a)), np.asarray(Image.open(b)))
    out = subprocess.run([CLI, "--env", str(tmp_path / "missing.hdr")], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "can't fopen" in out.stderr


def test_write_to_png_error_message(ctx, tmp_path):
    rr = R.create_render_resources(ctx, 16, 16, scenes.cornell_box())
    rr.render(R.default_camera(16, 16), frame=0)
    assert R.write_to_png(ctx, rr, str(tmp_path / "ok.png")) == ""
    msg = R.write_to_png(ctx, rr, str(tmp_path / "no_such_dir" / "x.png"))
    assert msg.startswith('Failed to write PNG image to "')
    img = np.asarray(Image.open(tmp_path / "ok.png"))
    assert np.array_equal(img, rr.read_rgba8())
    rr.close()
