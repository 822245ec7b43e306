"""TEST INFRASTRUCTURE ONLY. A numpy restatement of the Radiance .hdr (RGBE)
format as stb_image's stbi_loadf decodes it, plus an encoder for fixtures.

The reference reads its environment map with stbi_loadf(file, &w, &h, &c, 4)
(src/utility.cpp:75-92). stb_image is a third-party dependency pinned in the
reference's CMakeLists.txt:75-79 (nothings/stb @ f4a71b1) and absent from the
reference tree, so this restates its published HDR decoder: header line
"#?RADIANCE" or "#?RGBE", "FORMAT=32-bit_rle_rgbe", blank line, "-Y h +X w";
scanlines either flat RGBE quadruples or new-style RLE (2, 2, w_hi, w_lo, then
the four channels separately as runs: count > 128 = repeat next byte count-128
times, else count literal bytes); texel = mantissa * 2^(e - 136), zero when
e == 0; alpha 1; rows top to bottom. PARITY UNPINNED: no output of stb_image
itself is available here; the product's reader is checked against this file and
against hand-computed texels only.

Only tests/ and tests/golden/make_golden.py import this module."""
import numpy as np


def float_to_rgbe(rgb: np.ndarray) -> np.ndarray:
    """(…, 3) float32 >= 0 -> (…, 4) uint8 RGBE, the classic Radiance rule:
    v = max(r, g, b); if v < 1e-32: 0; else m, e = frexp(v); scale = m * 256 / v."""
    rgb = np.asarray(rgb, np.float64)
    v = rgb.max(axis=-1)
    m, e = np.frexp(v)
    scale = np.where(v < 1e-32, 0.0, m * 256.0 / np.where(v < 1e-32, 1.0, v))
    out = np.zeros(rgb.shape[:-1] + (4,), np.uint8)
    out[..., :3] = np.clip(np.floor(rgb * scale[..., None]), 0, 255).astype(np.uint8)
    out[..., 3] = np.where(v < 1e-32, 0, e + 128).astype(np.uint8)
    out[v < 1e-32] = 0
    return out


def rgbe_to_float(rgbe: np.ndarray) -> np.ndarray:
    """(…, 4) uint8 -> (…, 4) float32 RGBA exactly as the decoder computes it."""
    rgbe = np.asarray(rgbe, np.uint8)
    e = rgbe[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = np.empty(rgbe.shape[:-1] + (4,), np.float32)
    out[..., :3] = rgbe[..., :3].astype(np.float32) * f[..., None]
    out[..., :3][e == 0] = 0.0
    out[..., 3] = 1.0
    return out


def _rle_channel(row: np.ndarray) -> bytes:
    out = bytearray()
    i, n = 0, len(row)
    while i < n:
        run = 1
        while i + run < n and run < 127 and row[i + run] == row[i]:
            run += 1
        if run >= 4:
            out += bytes([128 + run, int(row[i])])
            i += run
            continue
        j = i
        while j < n and j - i < 128:
            r = 1
            while j + r < n and r < 4 and row[j + r] == row[j]:
                r += 1
            if r >= 4:
                break
            j += 1
        out += bytes([j - i]) + bytes(row[i:j].tolist())
        i = j
    return bytes(out)


def encode(rgbe: np.ndarray, rle: bool, magic: str = "#?RADIANCE") -> bytes:
    """(h, w, 4) uint8 RGBE -> file bytes."""
    h, w = rgbe.shape[:2]
    head = f"{magic}\n# fixture\nFORMAT=32-bit_rle_rgbe\n\n-Y {h} +X {w}\n".encode()
    if not rle or w < 8 or w >= 32768:
        return head + rgbe.tobytes()
    body = bytearray()
    for j in range(h):
        body += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            body += _rle_channel(rgbe[j, :, k])
    return head + bytes(body)


def decode(data: bytes) -> np.ndarray:
    """File bytes -> (h, w, 4) float32, following the decoder described above."""
    pos = 0

    def line():
        nonlocal pos
        end = data.index(b"\n", pos)
        s = data[pos:end].decode("latin-1")
        pos = end + 1
        return s

    if line() not in ("#?RADIANCE", "#?RGBE"):
        raise ValueError("not HDR")
    ok = False
    while True:
        s = line()
        if s == "":
            break
        ok |= s == "FORMAT=32-bit_rle_rgbe"
    if not ok:
        raise ValueError("unsupported format")
    parts = line().split()
    if len(parts) != 4 or parts[0] != "-Y" or parts[2] != "+X":
        raise ValueError("unsupported data layout")
    h, w = int(parts[1]), int(parts[3])
    body = np.frombuffer(data, np.uint8, offset=pos)
    flat = w < 8 or w >= 32768 or not (body[0] == 2 and body[1] == 2 and not body[2] & 0x80)
    if flat:
        return rgbe_to_float(body[:4 * w * h].reshape(h, w, 4))
    rgbe = np.zeros((h, w, 4), np.uint8)
    p = 0
    for j in range(h):
        assert body[p] == 2 and body[p + 1] == 2 and (int(body[p + 2]) << 8 | int(body[p + 3])) == w
        p += 4
        for k in range(4):
            i = 0
            while i < w:
                c = int(body[p])
                p += 1
                if c > 128:
                    rgbe[j, i:i + c - 128, k] = body[p]
                    p += 1
                    i += c - 128
                else:
                    rgbe[j, i:i + c, k] = body[p:p + c]
                    p += c
                    i += c
    return rgbe_to_float(rgbe)
