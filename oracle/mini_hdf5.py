"""Minimal reader for small, uncompressed NetCDF-4/HDF5 files (TEST INFRASTRUCTURE, used by make_v3_golden.py only).

h5py / netCDF4 are not installed in this image.  The reference's v3 regression inputs
(tests/test_data/test_interpolation_data_random_*.nc, read by tests/test_interpolation.py:307) are HDF5 files with a
version-2 superblock, version-2 object headers, compact link messages and contiguous (or compact) dataset layouts --
the subset of the HDF5 file format specification this module understands.  Anything else raises NotImplementedError.
"""

from __future__ import annotations

import struct

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


class MiniHDF5:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.b = f.read()
        b = self.b
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise ValueError("not an HDF5 file")
        if b[8] not in (2, 3) or b[9] != 8 or b[10] != 8:
            raise NotImplementedError("only superblock v2/v3 with 8-byte offsets and lengths")
        self.base, _ext, _eof, self.root = struct.unpack("<QQQQ", b[12:44])
        self.datasets: dict[str, np.ndarray] = {}
        for name, addr in self._links(self.root).items():
            arr = self._dataset(addr)
            if arr is not None:
                self.datasets[name] = arr

    # -- object headers (version 2) ---------------------------------------------------------------------------
    def _messages(self, addr):
        b = self.b
        p = self.base + addr
        if b[p : p + 4] != b"OHDR" or b[p + 4] != 2:
            raise NotImplementedError("only version-2 object headers")
        flags = b[p + 5]
        p += 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        chunk0 = int.from_bytes(b[p : p + nsz], "little")
        p += nsz
        track = bool(flags & 0x04)
        out = []
        todo = [(p, p + chunk0)]
        while todo:
            q, end = todo.pop(0)
            while q + 4 <= end:
                mtype = b[q]
                msize = struct.unpack("<H", b[q + 1 : q + 3])[0]
                q += 4 + (2 if track else 0)
                body = b[q : q + msize]
                q += msize
                if mtype == 0x10:  # continuation -> OCHK block (signature, messages, checksum)
                    off, length = struct.unpack("<QQ", body[:16])
                    s = self.base + off
                    if b[s : s + 4] != b"OCHK":
                        raise ValueError("bad continuation block")
                    todo.append((s + 4, s + length - 4))
                elif mtype != 0:
                    out.append((mtype, body))
        return out

    def _links(self, addr):
        links = {}
        for mtype, m in self._messages(addr):
            if mtype != 0x06:
                if mtype == 0x02:  # link info: dense storage would need fractal heaps
                    heap = struct.unpack("<Q", m[2 + (8 if m[1] & 1 else 0) :][:8])[0]
                    if heap != _UNDEF:
                        raise NotImplementedError("dense link storage")
                continue
            flags = m[1]
            p = 2
            ltype = 0
            if flags & 0x08:
                ltype = m[p]
                p += 1
            if flags & 0x04:
                p += 8
            if flags & 0x10:
                p += 1
            ln = 1 << (flags & 3)
            n = int.from_bytes(m[p : p + ln], "little")
            p += ln
            name = m[p : p + n].decode()
            p += n
            if ltype == 0:
                links[name] = struct.unpack("<Q", m[p : p + 8])[0]
        return links

    def _dataset(self, addr):
        shape = dtype = None
        data = None
        for mtype, m in self._messages(addr):
            if mtype == 0x01:  # dataspace
                ver, rank, flags = m[0], m[1], m[2]
                p = 8 if ver == 1 else 4
                shape = struct.unpack(f"<{rank}Q", m[p : p + 8 * rank])
            elif mtype == 0x03:  # datatype
                cls = m[0] & 0x0F
                bits0 = m[1]
                size = struct.unpack("<I", m[4:8])[0]
                order = ">" if bits0 & 1 else "<"
                if cls == 0:
                    signed = bool(bits0 & 0x08)
                    dtype = np.dtype(f"{order}{'i' if signed else 'u'}{size}")
                elif cls == 1:
                    dtype = np.dtype(f"{order}f{size}")
                else:
                    dtype = None  # strings, references (dimension-scale bookkeeping): not needed
            elif mtype == 0x08:  # data layout
                ver, lclass = m[0], m[1]
                if ver not in (3, 4):
                    raise NotImplementedError("layout message version")
                if lclass == 1:
                    a, n = struct.unpack("<QQ", m[2:18])
                    data = None if a == _UNDEF else self.b[self.base + a : self.base + a + n]
                elif lclass == 0:
                    n = struct.unpack("<H", m[2:4])[0]
                    data = m[4 : 4 + n]
                else:
                    raise NotImplementedError("chunked dataset layout")
        if shape is None or dtype is None or data is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        return np.frombuffer(data, dtype=dtype, count=n).reshape(shape).astype(dtype.newbyteorder("="))


def read(path) -> dict:
    return MiniHDF5(path).datasets
