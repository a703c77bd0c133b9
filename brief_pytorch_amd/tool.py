"""Volume / image file I/O for the CLI (reference utils/tool.py:71-110 uses tifffile + cv2, neither
of which exists in this image).  Supported: .npy (any dtype) and baseline TIFF — uncompressed,
8/16-bit grayscale, single or multi page, little or big endian — which is what tifffile.imsave
writes for the reference's (d,h,w) stacks.  Layout contract kept: 3-D data is (d,h,w,c), 2-D (h,w,c).
"""
import os
import struct

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 16: ("Q", 8)}


def _read_ifd(buf, off, bo):
    n = struct.unpack_from(bo + "H", buf, off)[0]
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from(bo + "HHI4s", buf, off + 2 + 12 * i)
        if typ not in _TYPES or typ == 5:
            continue
        code, size = _TYPES[typ]
        if cnt * size <= 4:
            raw = val[:cnt * size]
        else:
            p = struct.unpack(bo + "I", val)[0]
            raw = buf[p:p + cnt * size]
        if typ == 2:
            tags[tag] = raw
        else:
            tags[tag] = list(struct.unpack(bo + code * cnt, raw))
    nxt = struct.unpack_from(bo + "I", buf, off + 2 + 12 * n)[0]
    return tags, nxt


def read_tiff(path):
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}[buf[:2]]
    if struct.unpack_from(bo + "H", buf, 2)[0] != 42:
        raise NotImplementedError("BigTIFF is not supported")
    off = struct.unpack_from(bo + "I", buf, 4)[0]
    pages = []
    while off:
        t, off = _read_ifd(buf, off, bo)
        if t.get(259, [1])[0] != 1:
            raise NotImplementedError("compressed TIFF is not supported")
        w, h = t[256][0], t[257][0]
        bits = t.get(258, [1])[0]
        spp = t.get(277, [1])[0]
        if bits not in (8, 16):
            raise NotImplementedError("TIFF bit depth %d" % bits)
        dt = np.dtype(("<" if bo == "<" else ">") + ("u1" if bits == 8 else "u2"))
        offs, cnts = t[273], t.get(279, [w * h * spp * bits // 8])
        raw = b"".join(buf[o:o + c] for o, c in zip(offs, cnts))
        img = np.frombuffer(raw, dtype=dt, count=w * h * spp).astype(dt.newbyteorder("="))
        pages.append(img.reshape(h, w, spp) if spp > 1 else img.reshape(h, w))
    return np.stack(pages) if len(pages) > 1 else pages[0]


def write_tiff(path, arr):
    """uncompressed little-endian baseline TIFF, one strip per page; arr (pages,h,w) or (h,w)"""
    a = np.ascontiguousarray(arr)
    if a.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError("TIFF dtype %s" % a.dtype)
    if a.ndim == 2:
        a = a[None]
    n, h, w = a.shape
    bits = a.dtype.itemsize * 8
    page_bytes = h * w * a.dtype.itemsize
    ntags = 9
    ifd_size = 2 + 12 * ntags + 4
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, 8))
        data0 = 8 + n * ifd_size
        for i in range(n):
            tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1),
                    (273, 4, 1, data0 + i * page_bytes), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, page_bytes)]
            f.write(struct.pack("<H", ntags))
            for tag, typ, cnt, val in tags:
                f.write(struct.pack("<HHII", tag, typ, cnt, val))
            f.write(struct.pack("<I", 8 + (i + 1) * ifd_size if i + 1 < n else 0))
        f.write(a.astype(a.dtype.newbyteorder("<")).tobytes())


def get_dimension(path):
    ext = os.path.splitext(path)[-1]
    if ext in (".tif", ".tiff", ".npy"):
        return 3
    if ext in (".png", ".jpg"):
        return 2
    raise NotImplementedError(ext)


def read_img(path):
    """-> (d,h,w,c) for stacks, (h,w,c) for images (utils/tool.py:73-92)"""
    ext = os.path.splitext(path)[-1]
    if ext in (".tif", ".tiff"):
        img = read_tiff(path)
        if img.ndim == 2:
            img = img[None]
        if img.ndim == 3:
            img = img[..., None]
        assert img.ndim == 4
        return img
    if ext == ".npy":
        img = np.load(path)
        return img[..., None] if img.ndim in (2, 3) and img.shape[-1] not in (1, 3) or img.ndim == 2 else img
    raise NotImplementedError("only .tif/.tiff/.npy inputs are supported in this build (no cv2 for %s)" % ext)


def save_img(path, img):
    ext = os.path.splitext(path)[-1]
    if ext in (".tif", ".tiff"):
        a = np.asarray(img)
        if a.ndim == 4 and a.shape[-1] == 1:
            a = a[..., 0]
        elif a.ndim == 3 and a.shape[-1] == 1:
            a = a[..., 0]
        return write_tiff(path, a)
    if ext == ".npy":
        return np.save(path, np.asarray(img))
    raise NotImplementedError("only .tif/.tiff/.npy outputs are supported in this build (no cv2 for %s)" % ext)
