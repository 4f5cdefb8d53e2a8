"""Minimal NIfTI-1 single-file codec (.nii / .nii.gz) for the results directory.

The reference writes and reads its result volumes through MedPy (`medpy.io.save / load`, i.e. SimpleITK;
data_carrier_3D.py:233-371, experiment_dataloader.py:44-169).  MedPy is a third-party dependency that is absent
here, so byte-level parity of the files is UNPINNED; this module follows the published NIfTI-1 layout (348-byte
header, vox_offset 352, little endian) and MedPy's axis convention: a numpy array indexed [x, y, z] is stored with x
varying fastest (dim[1..3] = X, Y, Z), which is what SimpleITK produces for `medpy.io.save(arr, ...)`.
Files written by the reference load here with the same [x, y, z] indexing.
"""
from __future__ import annotations

import gzip
import struct

import numpy as np

_DT = {np.dtype("uint8"): (2, 8), np.dtype("int16"): (4, 16), np.dtype("int32"): (8, 32), np.dtype("float32"): (16, 32),
       np.dtype("float64"): (64, 64), np.dtype("int8"): (256, 8), np.dtype("uint16"): (512, 16),
       np.dtype("uint32"): (768, 32), np.dtype("int64"): (1024, 64), np.dtype("uint64"): (1280, 64)}
_CODE = {v[0]: k for k, v in _DT.items()}


def save(arr, path: str, header=None) -> None:
    """arr: numpy array (or anything np.asarray accepts) indexed [x, y(, z, ...)].  `header` (a dict with optional
    "pixdim" and "affine") plays the role of medpy's header argument; False / None = identity geometry."""
    a = np.asarray(arr)
    if a.dtype == np.bool_:
        a = a.astype(np.uint8)
    if a.dtype.newbyteorder("<") not in _DT and a.dtype not in _DT:
        a = a.astype(np.float64)
    a = a.astype(a.dtype.newbyteorder("<"), copy=False)
    code, bitpix = _DT[np.dtype(a.dtype.name)]
    nd = a.ndim
    if not 1 <= nd <= 7:
        raise ValueError("NIfTI-1 stores 1..7 dimensions")
    dim = [nd] + list(a.shape) + [1] * (7 - nd)
    pixdim = [1.0] * 8
    affine = np.eye(4)
    if isinstance(header, dict):
        for i, p in enumerate(header.get("pixdim", [])[:nd]):
            pixdim[1 + i] = float(p)
        affine = np.asarray(header.get("affine", np.diag(pixdim[1:4] + [1.0])), dtype=np.float64)
    else:
        affine = np.diag(pixdim[1:4] + [1.0])
    h = bytearray(348)
    struct.pack_into("<i", h, 0, 348)
    struct.pack_into("<8h", h, 40, *dim)
    struct.pack_into("<h", h, 70, code)
    struct.pack_into("<h", h, 72, bitpix)
    struct.pack_into("<8f", h, 76, *pixdim)
    struct.pack_into("<f", h, 108, 352.0)       # vox_offset
    struct.pack_into("<f", h, 112, 1.0)         # scl_slope
    struct.pack_into("<B", h, 123, 2)           # xyzt_units: mm
    struct.pack_into("<h", h, 252, 0)           # qform_code
    struct.pack_into("<h", h, 254, 2)           # sform_code: aligned
    struct.pack_into("<4f", h, 280, *affine[0])
    struct.pack_into("<4f", h, 296, *affine[1])
    struct.pack_into("<4f", h, 312, *affine[2])
    h[344:348] = b"n+1\0"
    payload = bytes(h) + b"\0\0\0\0" + np.asfortranarray(a).tobytes(order="F")
    if str(path).endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


def load(path: str):
    """-> (array indexed [x, y, z], header dict) like medpy.io.load."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    if struct.unpack_from("<i", raw, 0)[0] == 348:
        e = "<"
    elif struct.unpack_from(">i", raw, 0)[0] == 348:
        e = ">"
    else:
        raise ValueError(f"{path}: not a NIfTI-1 file")
    dim = struct.unpack_from(e + "8h", raw, 40)
    code = struct.unpack_from(e + "h", raw, 70)[0]
    pixdim = struct.unpack_from(e + "8f", raw, 76)
    vox_offset = int(struct.unpack_from(e + "f", raw, 108)[0])
    slope, inter = struct.unpack_from(e + "2f", raw, 112)
    if code not in _CODE:
        raise ValueError(f"{path}: unsupported NIfTI datatype {code}")
    shape = tuple(int(d) for d in dim[1:1 + dim[0]])
    dt = _CODE[code].newbyteorder(e)
    n = int(np.prod(shape))
    a = np.frombuffer(raw, dtype=dt, count=n, offset=max(vox_offset, 352)).reshape(shape, order="F")
    a = np.ascontiguousarray(a.astype(dt.newbyteorder("=")))
    if slope not in (0.0, 1.0) or inter != 0.0:
        a = a * slope + inter
    affine = np.eye(4)
    affine[0] = struct.unpack_from(e + "4f", raw, 280)
    affine[1] = struct.unpack_from(e + "4f", raw, 296)
    affine[2] = struct.unpack_from(e + "4f", raw, 312)
    return a, {"pixdim": list(pixdim[1:1 + dim[0]]), "affine": affine, "datatype": code}
