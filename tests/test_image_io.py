"""PNG / JPG routes of read_img / save_img (reference utils/tool.py:85-91, 100-101: cv2.imread(path, -1) / cv2.imwrite):
the own zlib codec against Pillow's, every row filter of the PNG spec, cv2's channel order and bit depths."""
import os
import struct
import warnings
import zlib

import numpy as np
import pytest

from brief_pytorch_amd.tool import get_dimension, read_img, read_png, save_img, write_png

Image = pytest.importorskip("PIL.Image")


def _img(seed=0, h=37, w=53):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    return (np.sin(yy / 7.0) * np.cos(xx / 5.0) * 100 + 128 + rng.integers(0, 8, (h, w))).astype(np.uint8)


def _pil(arr, mode):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        return Image.fromarray(arr, mode)


def test_png_read_matches_pillow_with_cv2_channel_order(tmp_path):
    g = _img()
    rgb = np.stack([g, g[::-1], np.roll(g, 5, 1)], -1)
    rgba = np.concatenate([rgb, np.roll(g, 9, 0)[..., None]], -1)
    g16 = (g.astype(np.uint16) * 257 + np.random.default_rng(1).integers(0, 200, g.shape)).astype(np.uint16)
    cases = [("L", g, g), ("RGB", rgb, rgb[..., ::-1]), ("RGBA", rgba, rgba[..., [2, 1, 0, 3]]),
             ("LA", rgba[..., :2], np.stack([g, g, g, rgba[..., 1]], -1)), ("I;16", g16, g16)]
    for mode, arr, want in cases:
        p = str(tmp_path / ("%s.png" % mode.replace(";", "")))
        _pil(arr, mode).save(p)                                   # Pillow picks Sub / Up / Paeth rows adaptively
        got = read_png(p)
        assert got.dtype == want.dtype and np.array_equal(got, want), mode
    pal = _pil(rgb, "RGB").quantize(16)                           # 4-bit palette image
    p = str(tmp_path / "pal.png")
    pal.save(p)
    assert np.array_equal(read_png(p), np.asarray(pal.convert("RGB"))[..., ::-1])


def _filtered(rows, bpp, ft):
    """PNG spec section 9 filters, encoder side, one type for every row"""
    h, n = rows.shape
    out = np.zeros((h, n + 1), np.uint8)
    out[:, 0] = ft
    prev = np.zeros(n, np.int64)
    for y in range(h):
        cur = rows[y].astype(np.int64)
        left = np.concatenate([np.zeros(bpp, np.int64), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        out[y, 1:] = (cur - pred) & 255
        prev = cur
    return out


@pytest.mark.parametrize("ft", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("depth,ch", [(8, 1), (8, 3), (16, 1), (16, 3)])
def test_png_every_row_filter(tmp_path, ft, depth, ch):
    g = _img(3, 19, 23)
    a = np.stack([np.roll(g, 3 * k, 1) for k in range(ch)], -1).astype(np.uint16 if depth == 16 else np.uint8)
    if depth == 16:
        a = a * 251 + 7
    rows = a.astype(">u2").copy().view(np.uint8).reshape(19, -1) if depth == 16 else a.reshape(19, -1)
    body = zlib.compress(_filtered(rows, ch * depth // 8, ft).tobytes())

    def chunk(kind, b):
        return struct.pack(">I", len(b)) + kind + b + struct.pack(">I", zlib.crc32(kind + b) & 0xFFFFFFFF)
    p = str(tmp_path / "f.png")
    with open(p, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 23, 19, depth, {1: 0, 3: 2}[ch], 0, 0, 0)) + chunk(b"IDAT", body) + chunk(b"IEND", b""))
    want = a[..., 0] if ch == 1 else a[..., ::-1]
    assert np.array_equal(read_png(p), want)
    if depth == 8:                                                # Pillow decodes the same file to the same pixels
        assert np.array_equal(np.asarray(Image.open(p)), a[..., 0] if ch == 1 else a)


def test_png_write_round_trip_and_pillow_reads_it(tmp_path):
    g = _img(5)
    g16 = g.astype(np.uint16) * 255 + 3
    bgr = np.stack([g, g[::-1], np.roll(g, 5, 1)], -1)
    for arr in (g, g16, bgr, np.stack([g16, g16[::-1], np.roll(g16, 5, 1)], -1), np.concatenate([bgr, g[..., None]], -1)):
        p = str(tmp_path / "w.png")
        write_png(p, arr)
        assert np.array_equal(read_png(p), arr)
        if arr.dtype == np.uint8:
            im = np.asarray(Image.open(p))
            assert np.array_equal(im if arr.ndim == 2 else im[..., [2, 1, 0] + ([3] if arr.shape[-1] == 4 else [])], arr)
        elif arr.ndim == 2:
            assert np.array_equal(np.asarray(Image.open(p)).astype(np.uint16), arr)
    with pytest.raises(NotImplementedError):
        write_png(str(tmp_path / "f.png"), g.astype(np.float32))
    p = str(tmp_path / "bad.png")
    write_png(p, g)
    raw = bytearray(open(p, "rb").read())
    raw[40] ^= 1
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        read_png(p)                                               # a flipped bit fails the chunk CRC instead of decoding garbage


def test_read_img_save_img_layout_contract(tmp_path):
    g = _img(7)
    p = str(tmp_path / "x.png")
    save_img(p, g[..., None])
    assert read_img(p).shape == g.shape + (1,) and get_dimension(p) == 2       # utils/tool.py:87-89: 2-D gray gains a channel axis
    bgr = np.stack([g, g[::-1], np.roll(g, 5, 1)], -1)
    save_img(p, bgr)
    assert np.array_equal(read_img(p), bgr)
    j = str(tmp_path / "x.jpg")
    save_img(j, bgr)
    back = read_img(j)
    assert back.shape == bgr.shape and np.abs(back.astype(int) - bgr.astype(int)).mean() < 6        # lossy, channel order kept
    with pytest.raises(NotImplementedError):
        read_img(str(tmp_path / "x.mp4"))
