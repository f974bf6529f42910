"""Minimal zarr-v2 + Blosc(lz4, byte shuffle) array reader (TEST INFRASTRUCTURE, used by make_v3_golden.py only).

zarr / numcodecs are not installed in this image.  The reference's v3-JIT golden trajectories
(tests/test_data/test_interpolation_jit_*.zarr, compared by tests/test_interpolation.py:362-378) are zarr-v2 stores whose
chunks are Blosc-1 frames: 16-byte header (version, versionlz, flags, typesize, nbytes, blocksize, cbytes), a table of
block offsets, and per block either one LZ4 stream or `typesize` split streams (each prefixed by its int32 compressed
length; a length equal to the raw length means "stored").  LZ4 block decoding comes from pyarrow's `lz4_raw` codec.
"""

from __future__ import annotations

import json
import os
import struct

import numpy as np
import pyarrow as pa

_BLOSC_DOSHUFFLE = 0x1
_BLOSC_MEMCPYED = 0x2
_BLOSC_DOBITSHUFFLE = 0x4
_BLOSC_DONT_SPLIT = 0x10


def _lz4(buf: bytes, n: int) -> bytes:
    if len(buf) == n:
        return bytes(buf)
    return pa.Codec("lz4_raw").decompress(buf, decompressed_size=n).to_pybytes()


def blosc_decompress(frame: bytes) -> bytes:
    _ver, _verlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes, blocksize, cbytes = struct.unpack("<III", frame[4:16])
    if cbytes != len(frame):
        raise ValueError("truncated blosc frame")
    if flags & _BLOSC_MEMCPYED:
        return bytes(frame[16 : 16 + nbytes])
    if flags & _BLOSC_DOBITSHUFFLE:
        raise NotImplementedError("bit shuffle")
    if (flags >> 5) != 1:
        raise NotImplementedError("only the lz4 codec")
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack(f"<{nblocks}i", frame[16 : 16 + 4 * nblocks])
    out = bytearray()
    for k in range(nblocks):
        bsize = min(blocksize, nbytes - k * blocksize)
        leftover = bsize != blocksize
        # blosc splits a block into `typesize` streams unless told not to (never for the leftover block)
        split = not (flags & _BLOSC_DONT_SPLIT) and not leftover and 1 < typesize <= 16 and blocksize // typesize >= 128
        nsplits = typesize if split else 1
        p = bstarts[k]
        block = bytearray()
        for _ in range(nsplits):
            n = bsize // nsplits
            (c,) = struct.unpack("<i", frame[p : p + 4])
            p += 4
            block += _lz4(frame[p : p + c], n)
            p += c
        if flags & _BLOSC_DOSHUFFLE and typesize > 1:
            nel = bsize // typesize
            body = np.frombuffer(bytes(block[: nel * typesize]), np.uint8).reshape(typesize, nel).T.tobytes()
            block = bytearray(body) + block[nel * typesize :]
        out += block
    return bytes(out)


def read_array(path) -> np.ndarray:
    meta = json.load(open(os.path.join(path, ".zarray")))
    if meta["zarr_format"] != 2 or meta.get("filters") or meta["order"] != "C":
        raise NotImplementedError("zarr v2, C order, no filters only")
    dtype = np.dtype(meta["dtype"])
    shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
    fill = meta.get("fill_value")
    out = np.empty(shape, dtype)
    out[...] = np.nan if fill == "NaN" else (0 if fill is None else fill)
    grid = [(s + c - 1) // c for s, c in zip(shape, chunks)]
    for idx in np.ndindex(*grid):
        f = os.path.join(path, ".".join(str(i) for i in idx))
        if not os.path.exists(f):
            continue
        raw = open(f, "rb").read()
        comp = meta.get("compressor")
        if comp is not None:
            if comp["id"] != "blosc":
                raise NotImplementedError(comp["id"])
            raw = blosc_decompress(raw)
        chunk = np.frombuffer(raw, dtype=dtype).reshape(chunks)
        sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
        out[sel] = chunk[tuple(slice(0, s.stop - s.start) for s in sel)]
    return out


def read_group(path) -> dict:
    return {name: read_array(os.path.join(path, name)) for name in sorted(os.listdir(path))
            if os.path.exists(os.path.join(path, name, ".zarray"))}
