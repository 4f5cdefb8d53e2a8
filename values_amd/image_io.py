"""Minimal PNG (8-bit RGB / grey) and TIFF (32-bit float, single channel) codecs for the 2D results directory.

The reference writes the 2D outputs with OpenCV (test_2D.py:116-159): `cv2.imwrite(<id>_NN.png, BGR colour image)` for
the arg-max masks and `cv2.imwrite(<id>.tif, float32 map)` for the uncertainty maps, and the evaluation side reads them
back with cv2.imread.  OpenCV is not a dependency here; these writers produce standard files any reader (OpenCV,
PIL, tifffile) decodes to the same arrays.  Byte streams are not the same as OpenCV's (different deflate / no LZW):
parity is on decoded content.  CPU I/O code -- not on the GPU path.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np


# ----------------------------------------------------------------------------------------------- PNG
def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, img: np.ndarray, level: int = 3) -> None:
    """img: (H, W, 3) RGB or (H, W) grey, uint8."""
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
        raise ValueError("write_png: uint8 (H, W) or (H, W, 3) expected")
    h, w = a.shape[:2]
    color_type = 2 if a.ndim == 3 else 0
    rows = a.reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()   # filter type 0 on every scanline
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(raw, level)))
        f.write(_chunk(b"IEND", b""))


def read_png(path) -> np.ndarray:
    """8-bit grey / RGB / RGBA, non-interlaced (all five scanline filters)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(buf):
        n, tag = struct.unpack(">I4s", buf[pos:pos + 8])
        data = buf[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 6):
        raise ValueError("read_png: only 8-bit non-interlaced grey/RGB/RGBA")
    bpp = {0: 1, 2: 3, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * bpp)
    out = np.zeros((h, w * bpp), dtype=np.uint8)
    prev = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 0xFF
        else:  # 1 (sub), 3 (average), 4 (Paeth): sequential along the scanline
            cur = np.zeros_like(line)
            for i in range(w * bpp):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pred) & 0xFF
        out[y] = cur
        prev = cur
    return out.reshape(h, w) if bpp == 1 else out.reshape(h, w, bpp)


# ----------------------------------------------------------------------------------------------- TIFF
def write_tiff_f32(path, img: np.ndarray) -> None:
    """img (H, W) float32 -> baseline little-endian TIFF: one strip, uncompressed, SampleFormat = IEEE float."""
    a = np.ascontiguousarray(img, dtype="<f4")
    if a.ndim != 2:
        raise ValueError("write_tiff_f32: (H, W) expected")
    h, w = a.shape
    data = a.tobytes()
    tags = [  # (tag, type, count, value)   type 3 = SHORT, 4 = LONG
        (256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 32), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8),
        (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, len(data)), (284, 3, 1, 1), (339, 3, 1, 3),
    ]
    ifd_off = 8 + len(data)
    if ifd_off % 2:
        data += b"\x00"
        ifd_off += 1
    ifd = struct.pack("<H", len(tags))
    for tag, typ, cnt, val in tags:
        ifd += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    ifd += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II*\x00" + struct.pack("<I", ifd_off))
        f.write(data)
        f.write(ifd)


def read_tiff_f32(path) -> np.ndarray:
    """Reads what write_tiff_f32 writes (and any uncompressed single-channel float32 strip TIFF, either byte order)."""
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}[buf[:2]]
    if struct.unpack(bo + "H", buf[2:4])[0] != 42:
        raise ValueError("not a TIFF file")
    off = struct.unpack(bo + "I", buf[4:8])[0]
    n = struct.unpack(bo + "H", buf[off:off + 2])[0]
    tags = {}
    for i in range(n):
        e = buf[off + 2 + 12 * i: off + 14 + 12 * i]
        tag, typ, cnt = struct.unpack(bo + "HHI", e[:8])
        size = {1: 1, 3: 2, 4: 4}.get(typ)
        if size is None:
            continue
        fmt = {1: "B", 3: "H", 4: "I"}[typ]
        if size * cnt <= 4:
            vals = struct.unpack(bo + fmt * cnt, e[8:8 + size * cnt])
        else:
            p = struct.unpack(bo + "I", e[8:12])[0]
            vals = struct.unpack(bo + fmt * cnt, buf[p:p + size * cnt])
        tags[tag] = vals
    w, h = tags[256][0], tags[257][0]
    if tags.get(259, (1,))[0] != 1 or tags.get(258, (0,))[0] != 32 or tags.get(339, (1,))[0] != 3 or tags.get(277, (1,))[0] != 1:
        raise ValueError("read_tiff_f32: only uncompressed single-channel 32-bit float")
    chunks = [buf[o:o + c] for o, c in zip(tags[273], tags[279])]
    return np.frombuffer(b"".join(chunks), dtype=bo + "f4").reshape(h, w).astype(np.float32)
