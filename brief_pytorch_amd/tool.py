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


def _tiff_pages(buf):
    """[(h, w, spp, dtype, strip offsets, strip byte counts)] of a classic TIFF held in a bytes-like object"""
    bo = {b"II": "<", b"MM": ">"}[bytes(buf[:2])]
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
        pages.append((h, w, spp, dt, t[273], t.get(279, [w * h * spp * bits // 8])))
    return pages


def read_tiff(path, mmap=False):
    """mmap=True: when the file is an equally spaced stack of single-strip native-endian pages (what write_tiff and
    tifffile.imsave produce) the result is a read-only np.memmap over the pixel data: nothing is read until sliced"""
    if mmap:
        mm = np.memmap(path, dtype=np.uint8, mode="r")
        pages = _tiff_pages(mm)
        h, w, spp, dt, offs, cnts = pages[0]
        page_bytes = h * w * spp * dt.itemsize
        uniform = all(p[:4] == (h, w, spp, dt) and len(p[4]) == 1 and p[5][0] == page_bytes for p in pages)
        stride = pages[1][4][0] - offs[0] if len(pages) > 1 else page_bytes
        if uniform and dt.isnative and stride == page_bytes and all(p[4][0] == offs[0] + i * stride for i, p in enumerate(pages)):
            shape = (len(pages), h, w) + ((spp,) if spp > 1 else ())
            arr = np.memmap(path, dtype=dt, mode="r", offset=offs[0], shape=shape)
            return arr if len(pages) > 1 else arr[0]
        del mm
    with open(path, "rb") as f:
        buf = f.read()
    out = []
    for h, w, spp, dt, offs, cnts in _tiff_pages(buf):
        raw = b"".join(buf[o:o + c] for o, c in zip(offs, cnts))
        img = np.frombuffer(raw, dtype=dt, count=w * h * spp).astype(dt.newbyteorder("="))
        out.append(img.reshape(h, w, spp) if spp > 1 else img.reshape(h, w))
    return np.stack(out) if len(out) > 1 else out[0]


def _tiff_header(f, n, h, w, dtype):
    """header + one IFD per page (one strip each) of an uncompressed little-endian stack; returns the pixel data offset"""
    bits = np.dtype(dtype).itemsize * 8
    page_bytes = h * w * np.dtype(dtype).itemsize
    ntags = 9
    ifd_size = 2 + 12 * ntags + 4
    data0 = 8 + n * ifd_size
    if data0 + n * page_bytes >= 2 ** 32:
        raise NotImplementedError("classic TIFF holds less than 4 GiB (this stack needs %.2f GiB): save it as .npy" % ((data0 + n * page_bytes) / 2.0 ** 30))
    f.write(struct.pack("<2sHI", b"II", 42, 8))
    for i in range(n):
        tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1),
                (273, 4, 1, data0 + i * page_bytes), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, page_bytes)]
        f.write(struct.pack("<H", ntags))
        for tag, typ, cnt, val in tags:
            f.write(struct.pack("<HHII", tag, typ, cnt, val))
        f.write(struct.pack("<I", 8 + (i + 1) * ifd_size if i + 1 < n else 0))
    return data0


def write_tiff(path, arr):
    """uncompressed little-endian baseline TIFF, one strip per page; arr (pages,h,w) or (h,w)"""
    a = np.ascontiguousarray(arr)
    if a.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError("TIFF dtype %s" % a.dtype)
    if a.ndim == 2:
        a = a[None]
    n, h, w = a.shape
    with open(path, "wb") as f:
        _tiff_header(f, n, h, w, a.dtype)
        f.write(a.astype(a.dtype.newbyteorder("<")).tobytes())


def create_stack(path, shape, dtype):
    """an empty (d,h,w[,1]) stack on disk (.tif: header + zero pages; .npy) that ranks then fill slab by slab with
    write_slab: the merged DivideTask volume is never assembled in one process"""
    d, h, w = (int(v) for v in shape[:3])
    ext = os.path.splitext(path)[-1]
    if ext in (".tif", ".tiff"):
        if np.dtype(dtype) not in (np.uint8, np.uint16) or (len(shape) == 4 and shape[3] != 1):
            raise NotImplementedError("TIFF stacks are single-channel uint8 / uint16")
        with open(path, "wb") as f:
            data0 = _tiff_header(f, d, h, w, dtype)
            f.truncate(data0 + d * h * w * np.dtype(dtype).itemsize)
    elif ext == ".npy":
        np.lib.format.open_memmap(path, mode="w+", dtype=np.dtype(dtype), shape=tuple(int(v) for v in shape)).flush()
    else:
        raise NotImplementedError(ext)


def write_slab(path, z0, slab):
    """slices z0 .. z0 + len(slab) of a stack made by create_stack"""
    ext = os.path.splitext(path)[-1]
    if ext == ".npy":
        mm = np.load(path, mmap_mode="r+")
        mm[z0:z0 + slab.shape[0]] = slab.reshape((slab.shape[0],) + mm.shape[1:])
        mm.flush()
        return
    a = np.ascontiguousarray(slab)
    a = a.reshape(a.shape[0], a.shape[1], a.shape[2])
    with open(path, "r+b") as f:
        head = f.read(8 + 2 + 12 * 9)
        data0 = struct.unpack_from("<I", head, 8 + 2 + 12 * 5 + 8)[0]      # StripOffsets of page 0
        f.seek(data0 + z0 * a.shape[1] * a.shape[2] * a.dtype.itemsize)
        f.write(a.astype(a.dtype.newbyteorder("<")).tobytes())


def get_dimension(path):
    ext = os.path.splitext(path)[-1]
    if ext in (".tif", ".tiff", ".npy"):
        return 3
    if ext in (".png", ".jpg"):
        return 2
    raise NotImplementedError(ext)


def read_img(path, mmap=False):
    """-> (d,h,w,c) for stacks, (h,w,c) for images (utils/tool.py:73-92).  mmap=True returns a read-only memory map
    where the file allows it (.npy, plain uncompressed .tif stacks): DivideTask ranks then touch only their own blocks."""
    ext = os.path.splitext(path)[-1]
    if ext in (".tif", ".tiff"):
        img = read_tiff(path, mmap=mmap)
        if img.ndim == 2:
            img = img[None]
        if img.ndim == 3:
            img = img[..., None]
        assert img.ndim == 4
        return img
    if ext == ".npy":
        img = np.load(path, mmap_mode="r" if mmap else None)
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
