"""Host-side build of the Morton spatial hash used for curvilinear cell search.

Restates (NumPy, host) what the reference builds in ``SpatialHash.__init__`` / ``_initialize_hash_table``
(src/parcels/_core/spatialhash.py:45-387): face bounding boxes in unit-sphere Cartesian (spherical mesh) or lon/lat
(flat mesh) -> 10-bit quantisation -> 30-bit Morton codes -> CSR table ``keys/starts/counts/faces`` sorted by
(code, face).  The table is uploaded once per grid; the *query* (spatialhash.py:389-535) runs on the GPU inside the
advection kernel (csrc/pk_device.h: hash_query).

The GPU build of this table is SURVEY.md section 8(f) item 3 ("next"); it only runs once per grid.
"""

from __future__ import annotations

import numpy as np

HASH_ENTRIES_PER_FACE = 16  # spatialhash.py:24
HASH_ENTRY_BUDGET_MIN = 2**22  # spatialhash.py:25
HASH_MAX_BITWIDTH = 1023  # spatialhash.py:26


def latlon_rad_to_xyz(lat, lon):
    """index_search.py:439-450"""
    return np.cos(lon) * np.cos(lat), np.sin(lon) * np.cos(lat), np.sin(lat)


def quantize_coordinates(x, y, z, bbox, bitwidth):
    """spatialhash.py:647-695"""
    xmin, xmax, ymin, ymax, zmin, zmax = bbox
    dx, dy, dz = xmax - xmin, ymax - ymin, zmax - zmin
    with np.errstate(invalid="ignore", divide="ignore"):
        xn = np.where(dx != 0, (np.asarray(x) - xmin) / dx, 0.0) if dx != 0 else np.zeros_like(np.asarray(x, dtype=float))
        yn = np.where(dy != 0, (np.asarray(y) - ymin) / dy, 0.0) if dy != 0 else np.zeros_like(np.asarray(y, dtype=float))
        zn = np.where(dz != 0, (np.asarray(z) - zmin) / dz, 0.0) if dz != 0 else np.zeros_like(np.asarray(z, dtype=float))
        xq = np.clip(xn * bitwidth, 0, bitwidth).astype(np.uint32)
        yq = np.clip(yn * bitwidth, 0, bitwidth).astype(np.uint32)
        zq = np.clip(zn * bitwidth, 0, bitwidth).astype(np.uint32)
    return xq, yq, zq


def dilate_bits(n):
    """spatialhash.py:554-597"""
    n = np.asarray(n, dtype=np.uint32) & np.uint32(0x000003FF)
    n = (n | (n << np.uint32(16))) & np.uint32(0xFF0000FF)
    n = (n | (n << np.uint32(8))) & np.uint32(0x0300F00F)
    n = (n | (n << np.uint32(4))) & np.uint32(0x030C30C3)
    n = (n | (n << np.uint32(2))) & np.uint32(0x09249249)
    return n


def encode_morton3d(xq, yq, zq):
    return ((dilate_bits(zq) << np.uint32(2)) | (dilate_bits(yq) << np.uint32(1)) | dilate_bits(xq)).astype(np.uint32)


class SpatialHash:
    """CSR Morton hash over the faces of a curvilinear XGrid (lon/lat 2-D, degrees)."""

    def __init__(self, lon: np.ndarray, lat: np.ndarray, spherical: bool):
        lon = np.asarray(lon)
        lat = np.asarray(lat)
        if spherical:  # spatialhash.py:60-108
            x, y, z = latlon_rad_to_xyz(np.deg2rad(lat), np.deg2rad(lon))
            self.bbox = (np.nanmin(x), np.nanmax(x), np.nanmin(y), np.nanmax(y), np.nanmin(z), np.nanmax(z))
        else:  # spatialhash.py:127-165
            x, y, z = lon, lat, None
            self.bbox = (np.nanmin(x), np.nanmax(x), np.nanmin(y), np.nanmax(y), 0.0, 0.0)

        def lowhigh(a):
            b = np.stack((a[:-1, :-1], a[:-1, 1:], a[1:, 1:], a[1:, :-1]), axis=-1)
            return np.min(b, axis=-1), np.max(b, axis=-1)

        self.xlow, self.xhigh = lowhigh(x)
        self.ylow, self.yhigh = lowhigh(y)
        if z is not None:
            self.zlow, self.zhigh = lowhigh(z)
        else:
            self.zlow = np.zeros_like(self.xlow)
            self.zhigh = np.zeros_like(self.xlow)
        self.face_shape = self.xlow.shape
        self.valid = ~(
            np.isnan(self.xlow) | np.isnan(self.xhigh) | np.isnan(self.ylow) | np.isnan(self.yhigh)
            | np.isnan(self.zlow) | np.isnan(self.zhigh)
        )
        self.bitwidth = HASH_MAX_BITWIDTH
        budget = max(HASH_ENTRIES_PER_FACE * self.xlow.size, HASH_ENTRY_BUDGET_MIN)  # spatialhash.py:214-228
        if self._total_entries(self.bitwidth) > budget:
            lo, hi = 1, self.bitwidth
            while lo < hi:
                mid = (lo + hi + 1) // 2
                if self._total_entries(mid) <= budget:
                    lo = mid
                else:
                    hi = mid - 1
            self.bitwidth = lo
        self.keys, self.starts, self.counts, self.faces = self._build()

    def _quant_boxes(self, bitwidth):
        lo = quantize_coordinates(self.xlow, self.ylow, self.zlow, self.bbox, bitwidth)
        hi = quantize_coordinates(self.xhigh, self.yhigh, self.zhigh, self.bbox, bitwidth)
        return lo, hi

    def _total_entries(self, bitwidth) -> int:
        (xl, yl, zl), (xh, yh, zh) = self._quant_boxes(bitwidth)
        nx = xh.astype(np.int64) - xl + 1
        ny = yh.astype(np.int64) - yl + 1
        nz = zh.astype(np.int64) - zl + 1
        return int(np.where(self.valid, nx * ny * nz, 0).sum())

    def _build(self):
        """spatialhash.py:269-387 (face-major entry generation, one fused (code<<32 | face) sort, CSR)."""
        (xl, yl, zl), (xh, yh, zh) = self._quant_boxes(self.bitwidth)
        xl, yl, zl = (a.ravel().astype(np.int32) for a in (xl, yl, zl))
        xh, yh, zh = (a.ravel().astype(np.int32) for a in (xh, yh, zh))
        nx, ny, nz = xh - xl + 1, yh - yl + 1, zh - zl + 1
        per_face = np.where(self.valid.ravel(), nx * ny * nz, 0).astype(np.int64)
        total = int(per_face.sum())
        nface = per_face.size
        face_ids = np.repeat(np.arange(nface, dtype=np.uint32), per_face)
        face_starts = np.concatenate(([0], np.cumsum(per_face)))[:-1]
        intra = np.arange(total, dtype=np.int64) - np.repeat(face_starts, per_face)
        ny_nz = np.repeat((ny * nz).astype(np.int64), per_face)
        nz_rep = np.repeat(nz.astype(np.int64), per_face)
        xi = intra // ny_nz
        rem = intra % ny_nz
        yi = rem // nz_rep
        zi = rem % nz_rep
        xq = np.repeat(xl, per_face) + xi
        yq = np.repeat(yl, per_face) + yi
        zq = np.repeat(zl, per_face) + zi
        codes = encode_morton3d(xq.astype(np.uint32), yq.astype(np.uint32), zq.astype(np.uint32))
        packed = (codes.astype(np.uint64) << np.uint64(32)) | face_ids.astype(np.uint64)
        packed.sort()
        faces = packed.astype(np.uint32)
        codes_sorted = (packed >> np.uint64(32)).astype(np.uint32)
        if codes_sorted.size == 0:
            z0 = np.zeros(0, dtype=np.int64)
            return np.zeros(0, np.uint32), z0, z0, faces
        starts = np.concatenate(([0], np.flatnonzero(codes_sorted[1:] != codes_sorted[:-1]) + 1)).astype(np.int64)
        keys = codes_sorted[starts]
        counts = np.diff(np.concatenate((starts, [codes_sorted.size]))).astype(np.int64)
        return keys, starts, counts, faces

    def table(self) -> dict:
        return dict(keys=self.keys, starts=self.starts, counts=self.counts, faces=self.faces, bitwidth=int(self.bitwidth),
                    bbox=np.asarray(self.bbox, dtype=np.float64))

    def checksum(self) -> dict:
        return table_checksum(self.table())


def table_checksum(t: dict) -> dict:
    import hashlib

    out = {}
    for k, dt in (("keys", np.uint32), ("starts", np.int64), ("counts", np.int64), ("faces", np.uint32)):
        a = np.ascontiguousarray(np.asarray(t[k]).astype(dt))
        out[k] = hashlib.sha256(a.tobytes()).hexdigest()
        out["n_" + k] = int(a.size)
    out["bitwidth"] = int(t["bitwidth"])
    out["bbox"] = [float(v) for v in np.asarray(t["bbox"], dtype=np.float64)]
    return out
