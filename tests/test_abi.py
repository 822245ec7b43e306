"""The C-ABI library loads and exports every symbol include/ptc.h declares; the
structs that cross the boundary have the sizes the header implies. No compute
call is made here (no GPU needed)."""
import ctypes as C
import os
import re

from path_tracer_amd import renderer as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = C.CDLL(R.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"libptcore.so does not export {n}"
    assert sorted(R.EXPORTS) == names


def test_struct_layouts():
    assert C.sizeof(R.Camera) == 84          # 21 floats, src/camera.hpp:6-19
    assert C.sizeof(R.MeshDesc) == 32
    assert C.sizeof(R.InstanceDesc) == 56
    assert C.sizeof(R.EnvmapDesc) == 16
    assert C.sizeof(R.BuildOptions) == 32 and R.BuildOptions.optimize_rounds.offset == 16  # took a reserved word
    assert C.sizeof(R.SceneInfo) == 40     # + n_bvh_triangles, reserved
    assert C.sizeof(R.RenderParams) == 56  # + debug_view, regroup
    assert C.sizeof(R.Stats) == 56         # + nodes_wave_uniform, uniform_fetches
    assert R.HIT_DTYPE.itemsize == 16


def test_errors_are_reported_without_a_device_call():
    L = R.lib()
    assert L.ptc_render(None, None, None, None, None) == -1
    assert b"NULL" in L.ptc_last_error()
    assert L.ptc_film_reset(None) == -1
    assert L.ptc_film_sample_count(None) == 0


def test_product_does_not_touch_the_oracle():
    """Nothing under path_tracer_amd/ may import, include or link oracle/."""
    pkg = os.path.join(ROOT, "path_tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pt_oracle" not in text and "oracle/" not in text, f"{f} references the oracle"


def test_render_workspace_matches_the_documented_sizes():
    """ptc_render_workspace is host arithmetic: the figures include/ptc.h and the
    design notes quote can be checked without a GPU."""
    # 1080p, the default bound of 2^29 paths: 256 dispatches (258 fit; batches are whole groups of 64) of 2,073,600 pixels in flight, about 76 bytes per path
    nbytes, batch = R.render_workspace(1920, 1080, n_frames=4096, integrator=R.INTEGRATOR_WAVEFRONT)
    assert batch == 256
    per_path = nbytes / (batch * 1920 * 1080)
    assert 76.0 <= per_path <= 78.5, per_path
    # launch pairs take the second queue and the hit records
    nbytes2, batch2 = R.render_workspace(1920, 1080, n_frames=4096, integrator=R.INTEGRATOR_WAVEFRONT,
                                         wavefront_bounces=2)
    assert batch2 == 256 and 155.0 <= nbytes2 / (batch2 * 1920 * 1080) <= 157.0
    # the stream integrator keeps live paths in registers: the radiance array only
    nbytes3, _ = R.render_workspace(1920, 1080, n_frames=64, samples_per_frame=2, integrator=R.INTEGRATOR_STREAM)
    assert 32.0 <= nbytes3 / (64 * 1920 * 1080) <= 32.1
    # the megakernel takes none
    assert R.render_workspace(1920, 1080, n_frames=7, integrator=R.INTEGRATOR_MEGAKERNEL) == (0, 7)
    # static workspace (64 MiB): the reference's default film fits twice, 1024x768 once, 1080p not at all
    assert R.render_workspace(640, 480, n_frames=3, integrator=R.INTEGRATOR_WAVEFRONT, static=True)[1] in (1, 2)
    nbytes4, batch4 = R.render_workspace(640, 480, n_frames=2, integrator=R.INTEGRATOR_WAVEFRONT, static=True)
    assert batch4 == 2 and nbytes4 <= 64 << 20
    assert R.render_workspace(1024, 768, integrator=R.INTEGRATOR_WAVEFRONT, static=True)[1] == 1
    import pytest
    with pytest.raises(R.PtcError, match="static array holds 67108864"):
        R.render_workspace(1920, 1080, integrator=R.INTEGRATOR_WAVEFRONT, static=True)
    assert R.render_workspace(1920, 1080, integrator=R.INTEGRATOR_STREAM, static=True)[0] <= 64 << 20
    # a film much smaller than the 32x32-pixel groups the queue covers: smaller batches, never more than twice
    # the slots the bound asks for
    nbytes5, batch5 = R.render_workspace(3, 2, n_frames=100000, integrator=R.INTEGRATOR_WAVEFRONT,
                                         max_paths_in_flight=6 * 1000)
    assert 1 <= batch5 < 1000 and nbytes5 <= 2 * 6000 * 64 + 6000 * 16 + 65536
    # AUTO: one sample per dispatch or 16+ dispatches in flight -> wavefront sizes, otherwise stream sizes
    assert R.render_workspace(640, 480, n_frames=4, samples_per_frame=3) == \
        R.render_workspace(640, 480, n_frames=4, samples_per_frame=3, integrator=R.INTEGRATOR_STREAM)
    assert R.render_workspace(640, 480, n_frames=4) == \
        R.render_workspace(640, 480, n_frames=4, integrator=R.INTEGRATOR_WAVEFRONT)
    with pytest.raises(R.PtcError):
        R.render_workspace(0, 480)
