"""Volume / image file I/O for the CLI (reference utils/tool.py:71-110 uses tifffile + cv2, neither
of which exists in this image).  Supported: .npy (any dtype); baseline TIFF — uncompressed,
8/16-bit grayscale, single or multi page, little or big endian — which is what tifffile.imsave
writes for the reference's (d,h,w) stacks; PNG — 8/16-bit gray, gray+alpha, RGB, RGBA and palette,
non-interlaced, through an own zlib codec with cv2's conventions (colour channels in B,G,R(,A) order,
bit depth kept: what `cv2.imread(path, -1)` / `cv2.imwrite` do at utils/tool.py:85-91, 100-101);
.jpg through Pillow when it is importable.  .mp4 needs a video codec this image does not have and raises.
Layout contract kept: 3-D data is (d,h,w,c), 2-D (h,w,c).
"""
import os
import struct
import zlib

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


_PNG_SIG = b"\x89PNG\r\n\x1a\n"
_PNG_CHANNELS = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _png_unfilter(raw, h, stride, bpp):
    """undo the per-row filters (PNG spec section 9): None/Sub/Up are whole-row numpy operations, Average/Paeth walk the row"""
    out = np.zeros((h + 1, stride), np.uint8)                      # row 0 = the all-zero row above the image
    rows = np.frombuffer(raw, np.uint8).reshape(h, stride + 1)
    for y in range(h):
        ft, line, up = int(rows[y, 0]), rows[y, 1:], out[y]
        cur = out[y + 1]
        if ft == 0:
            cur[:] = line
        elif ft == 1:                                                # Sub: running sum per byte lane, modulo 256
            pad = (-stride) % bpp
            lanes = np.concatenate([line, np.zeros(pad, np.uint8)]).reshape(-1, bpp).astype(np.uint32)
            cur[:] = (np.cumsum(lanes, axis=0) & 255).astype(np.uint8).reshape(-1)[:stride]
        elif ft == 2:
            cur[:] = line + up
        elif ft in (3, 4):
            a = bytearray(stride)
            ln, u = line.tobytes(), up.tobytes()
            if ft == 3:
                for i in range(stride):
                    left = a[i - bpp] if i >= bpp else 0
                    a[i] = (ln[i] + ((left + u[i]) >> 1)) & 255
            else:
                for i in range(stride):
                    left = a[i - bpp] if i >= bpp else 0
                    ul = u[i - bpp] if i >= bpp else 0
                    pa, pb, pc = abs(u[i] - ul), abs(left - ul), abs(left + u[i] - 2 * ul)
                    pred = left if (pa <= pb and pa <= pc) else (u[i] if pb <= pc else ul)
                    a[i] = (ln[i] + pred) & 255
            cur[:] = np.frombuffer(bytes(a), np.uint8)
        else:
            raise ValueError("PNG filter type %d" % ft)
    return out[1:]


def read_png(path):
    """-> (h,w) or (h,w,c) uint8 / uint16, colour in B,G,R(,A) order like cv2.imread(path, -1)"""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != _PNG_SIG:
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, plte, trns, hdr = 8, [], None, None, None
    while pos < len(buf):
        n, kind = struct.unpack(">I4s", buf[pos:pos + 8])
        body = buf[pos + 8:pos + 8 + n]
        if zlib.crc32(kind + body) & 0xFFFFFFFF != struct.unpack(">I", buf[pos + 8 + n:pos + 12 + n])[0]:
            raise ValueError("%s: CRC mismatch in %s chunk" % (path, kind.decode("latin1")))
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b"tRNS":
            trns = body
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if interlace:
        raise NotImplementedError("interlaced PNG is not supported")
    if ctype not in _PNG_CHANNELS or depth not in (1, 2, 4, 8, 16) or (depth == 16 and ctype == 3) or (depth < 8 and ctype not in (0, 3)):
        raise ValueError("%s: colour type %d with bit depth %d" % (path, ctype, depth))
    ch = _PNG_CHANNELS[ctype]
    stride = (w * ch * depth + 7) // 8
    data = _png_unfilter(zlib.decompress(b"".join(idat)), h, stride, max(1, ch * depth // 8))
    if depth == 16:
        img = data.reshape(h, w * ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8 | img[..., 1]).reshape(h, w, ch)
    elif depth == 8:
        img = data.reshape(h, w, ch)
    else:                                                           # packed samples: gray is scaled to 8 bits like cv2, palette indices are not
        bits = np.unpackbits(data, axis=1)[:, :w * depth].reshape(h, w, depth)
        val = np.zeros((h, w), np.uint16)
        for b in range(depth):
            val = val << 1 | bits[..., b]
        img = (val.astype(np.uint8) if ctype == 3 else (val * (255 // ((1 << depth) - 1))).astype(np.uint8))[..., None]
    if ctype == 3:
        img = plte[img[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255, np.uint8)
            alpha[:len(trns)] = np.frombuffer(trns, np.uint8)
            img = np.concatenate([img, alpha[data.reshape(h, w) if depth == 8 else val][..., None]], axis=-1)
    if img.shape[-1] == 1:
        return np.ascontiguousarray(img[..., 0])
    if img.shape[-1] == 2:                                          # gray + alpha: cv2 hands back B,G,R,A with the gray replicated
        img = np.concatenate([img[..., :1]] * 3 + [img[..., 1:]], axis=-1)
    order = [2, 1, 0] + ([3] if img.shape[-1] == 4 else [])
    return np.ascontiguousarray(img[..., order])


def write_png(path, img, level=6):
    """(h,w) / (h,w,1|3|4) uint8 or uint16, colour in B,G,R(,A) order like cv2.imwrite; rows are Up-filtered (vectorised)"""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[..., None]
    if a.ndim != 3 or a.shape[-1] not in (1, 3, 4) or a.dtype not in (np.uint8, np.uint16):
        raise NotImplementedError("PNG holds (h,w[,1|3|4]) uint8/uint16 images, got %s %s" % (a.shape, a.dtype))
    h, w, ch = a.shape
    if ch >= 3:
        a = a[..., [2, 1, 0] + ([3] if ch == 4 else [])]
    depth = 8 * a.dtype.itemsize
    rows = np.ascontiguousarray(a, dtype=">u2").view(np.uint8).reshape(h, -1) if depth == 16 else np.ascontiguousarray(a).reshape(h, -1)
    filt = np.empty((h, rows.shape[1] + 1), np.uint8)
    filt[:, 0] = 2
    filt[0, 1:] = rows[0]
    filt[1:, 1:] = rows[1:] - rows[:-1]

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, {1: 0, 3: 2, 4: 6}[ch], 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(filt.tobytes(), level)) + chunk(b"IEND", b""))


def _pillow():
    try:
        from PIL import Image
        return Image
    except ImportError:
        raise NotImplementedError(".jpg needs Pillow (cv2 is not in this image); use .png, .tif or .npy")


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
    if ext == ".png":
        img = read_png(path)
        return img[..., None] if img.ndim == 2 else img
    if ext == ".jpg":
        img = np.asarray(_pillow().open(path))
        return img[..., None] if img.ndim == 2 else np.ascontiguousarray(img[..., ::-1])      # cv2 order: B,G,R
    raise NotImplementedError("inputs: .tif/.tiff/.npy/.png/.jpg (%s needs a video codec this image does not have)" % ext)


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
    if ext == ".png":
        return write_png(path, np.asarray(img))
    if ext == ".jpg":
        a = np.asarray(img)
        a = a[..., 0] if a.ndim == 3 and a.shape[-1] == 1 else (a[..., ::-1] if a.ndim == 3 else a)
        return _pillow().fromarray(np.ascontiguousarray(a)).save(path, quality=95)          # cv2.imwrite's default quality
    raise NotImplementedError("outputs: .tif/.tiff/.npy/.png/.jpg (%s needs a video codec this image does not have)" % ext)
