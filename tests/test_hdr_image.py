"""Radiance .hdr reader of the C ABI (ptc_read_hdr, the stand-in for the
reference's read_hdr_image, src/utility.cpp:75-92) against the numpy
restatement of the format in oracle/hdr_image.py. Host code only: runs
without a GPU. Bar: bit-exact float32 texels."""
import base64
import json
import os

import numpy as np
import pytest

from oracle import hdr_image as H
from path_tracer_amd import renderer as R

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "hdr_image.json")


def _sky(h, w, seed):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3), dtype=np.float32) * np.float32(4.0)
    img[: h // 3] = np.float32(0.75)            # constant band -> long runs
    img[h // 3, : w // 2] = np.float32(1500.0)  # a sun: large exponent
    img[-1, -3:] = 0.0                          # e == 0 texels
    img[-1, 0] = np.float32(3e-6)               # small exponent
    return img


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("rle", [True, False], ids=["rle", "flat"])
@pytest.mark.parametrize("shape", [(9, 33), (4, 7), (16, 300), (1, 8)])
def test_reader_matches_restatement(tmp_path, rle, shape):
    rgbe = H.float_to_rgbe(_sky(*shape, seed=shape[1]))
    data = H.encode(rgbe, rle=rle, magic="#?RGBE" if shape[0] == 4 else "#?RADIANCE")
    path = tmp_path / "sky.hdr"
    path.write_bytes(data)
    got = R.read_hdr_image(str(path))
    ref = H.decode(data)
    assert got.shape == (shape[0], shape[1], 4)
    assert np.array_equal(_bits(got), _bits(ref))
    assert np.array_equal(_bits(got), _bits(H.rgbe_to_float(rgbe)))  # both encodings decode to the same texels
    assert np.all(got[..., 3] == 1.0)


def test_hand_computed_texels(tmp_path):
    # mantissa * 2^(e - 136), no rounding offset; e == 0 is black
    rgbe = np.array([[[128, 64, 32, 129], [255, 0, 1, 136], [200, 100, 50, 0], [1, 1, 1, 1]] * 2], np.uint8)
    path = tmp_path / "t.hdr"
    path.write_bytes(H.encode(rgbe, rle=True))
    got = R.read_hdr_image(str(path))[0]
    assert got[0].tolist() == [1.0, 0.5, 0.25, 1.0]
    assert got[1].tolist() == [255.0, 0.0, 1.0, 1.0]
    assert got[2].tolist() == [0.0, 0.0, 0.0, 1.0]
    assert got[3, 0] == np.float32(2.0) ** -135


def test_golden_file(tmp_path):
    g = json.load(open(GOLDEN))
    for case in g["cases"]:
        path = tmp_path / case["name"]
        path.write_bytes(base64.b64decode(case["file_b64"]))
        got = R.read_hdr_image(str(path))
        want = np.array([int(x, 16) for x in case["rgba_bits"]], np.uint32).reshape(case["height"], case["width"], 4)
        assert np.array_equal(_bits(got), want), case["name"]
        assert np.array_equal(_bits(H.decode(base64.b64decode(case["file_b64"]))), want), case["name"]


@pytest.mark.parametrize("content,reason", [
    (b"P6\n1 1\n255\n\0\0\0", "not HDR"),
    (b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 1 +X 8\n", "unsupported format"),
    (b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+X 8 -Y 1\n", "unsupported data layout"),
    (b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 8\n" + bytes([2, 2, 0, 8, 136, 1]), "corrupt"),
    (b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 9\n" + bytes([2, 2, 0, 8]), "invalid decoded scanline length"),
])
def test_errors_are_reported(tmp_path, content, reason):
    path = tmp_path / "bad.hdr"
    path.write_bytes(content)
    with pytest.raises(R.PtcError, match=reason):
        R.read_hdr_image(str(path))
    with pytest.raises(R.PtcError, match="can't fopen"):
        R.read_hdr_image(str(tmp_path / "missing.hdr"))
