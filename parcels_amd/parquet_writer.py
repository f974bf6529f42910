"""A small multi-threaded Parquet writer for the trajectory tables of ParticleFile (particlefile.py:142-221 of the reference writes
them with pyarrow's ParquetWriter).

pyarrow encodes one table on ONE thread -- 0.5 s per table of 1e7 particles on the GPU box's host, 0.4 s of it without any compression
(profiles/r02_h_writeout.json) -- so with output switched on the writer, not the GPU, bounds a run.  The tables written here are
the simplest Parquet there is: flat numeric columns without nulls.  This writer therefore emits PLAIN-encoded data pages straight
from the NumPy buffers (no intermediate Arrow table), one page per column and row group, compresses the pages of all columns and
row groups concurrently on a thread pool (zstd / lz4 / snappy / gzip through pyarrow's codecs, which release the GIL), and writes
them in order -- the bytes of a file do not depend on the number of threads.  Pages are DataPageV2 (levels outside the
compressed block), which every Parquet reader of the last decade reads.  The footer is the Thrift compact encoding of
parquet.thrift's FileMetaData, including pyarrow's "ARROW:schema" entry, so that pyarrow / pandas read the file back with the
schema (field and file metadata) ``get_schema`` declared, exactly as they read the reference's files.

Anything but bool / int / float columns, or a null in the data, is not for this writer: ParticleFile falls back to pyarrow's.
"""

from __future__ import annotations

import base64
import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np

__all__ = ["FastParquetWriter", "supports_schema"]

# parquet.thrift enums
_TYPE = {"bool": 0, "int32": 1, "int64": 2, "float32": 4, "float64": 5}
_CODEC = {None: 0, "none": 0, "snappy": 1, "gzip": 2, "lz4": 7, "zstd": 6}
_ARROW_CODEC = {"snappy": "snappy", "gzip": "gzip", "lz4": "lz4_raw", "zstd": "zstd"}
# how narrower / unsigned NumPy dtypes are stored (Parquet has INT32 / INT64 physical types + a converted type)
_STORE = {"int8": ("int32", 15), "int16": ("int32", 16), "int32": ("int32", None), "int64": ("int64", None), "uint8": ("int32", 11), "uint16": ("int32", 12),
          "uint32": ("int32", 13), "uint64": ("int64", 14), "float32": ("float32", None), "float64": ("float64", None), "bool": ("bool", None)}


# ---- Thrift compact protocol (the subset parquet.thrift needs) ----------------------------------------------------------------------
def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _zigzag(n: int) -> bytes:
    return _varint((n << 1) ^ (n >> 63))


class _Struct:
    """Fields must be added in increasing id order."""

    def __init__(self):
        self.b = bytearray()
        self.last = 0

    def _head(self, fid, ftype):
        d = fid - self.last
        if 0 < d <= 15:
            self.b.append((d << 4) | ftype)
        else:
            self.b.append(ftype)
            self.b += _zigzag(fid)
        self.last = fid

    def i32(self, fid, v):
        self._head(fid, 5)
        self.b += _zigzag(int(v))
        return self

    def i64(self, fid, v):
        self._head(fid, 6)
        self.b += _zigzag(int(v))
        return self

    def binary(self, fid, v):
        v = v.encode() if isinstance(v, str) else bytes(v)
        self._head(fid, 8)
        self.b += _varint(len(v)) + v
        return self

    def struct(self, fid, s: "_Struct"):
        self._head(fid, 12)
        self.b += s.done()
        return self

    def list_(self, fid, etype, items):
        """items: already-encoded elements (structs: their bytes incl. stop; i32: zigzag varints; binaries: length-prefixed)."""
        self._head(fid, 9)
        n = len(items)
        self.b += bytes([(n << 4) | etype]) if n < 15 else bytes([0xF0 | etype]) + _varint(n)
        for it in items:
            self.b += it
        return self

    def done(self) -> bytes:
        return bytes(self.b) + b"\x00"


def _bin(v) -> bytes:
    v = v.encode() if isinstance(v, str) else bytes(v)
    return _varint(len(v)) + v


def supports_schema(schema) -> bool:
    """Can FastParquetWriter write tables of this pyarrow schema?"""
    import pyarrow as pa

    for f in schema:
        try:
            if str(np.dtype(f.type.to_pandas_dtype())) not in _STORE or pa.types.is_temporal(f.type) or pa.types.is_nested(f.type):
                return False
        except Exception:
            return False
    return True


import threading

_tls = threading.local()


def _pwrite_all(fd, parts, pos):
    for b in parts:
        mv = memoryview(b).cast("B")
        done = 0
        while done < len(mv):
            done += os.pwrite(fd, mv[done:], pos + done)
        pos += len(mv)


class FastParquetWriter:
    def __init__(self, path, schema, compression="zstd", row_group_rows=1 << 20, threads=None, page_rows=1 << 19):
        """schema: the pyarrow schema of the tables (field + file metadata included), as pyarrow.parquet.ParquetWriter takes it.
        page_rows: rows per data page.  A column chunk of a row group is a run of pages, each compressed on its own -- round 5 wrote ONE page
        per chunk: a table of 4e6 rows was 20 jobs of 8 MB, so 32 threads were no faster than 8 (31 ms per 160 MB table, 5 GB/s;
        profiles/r05_m_writeout_4e6.json).  Smaller pages keep the pool busy to the end, but every page is also a Python job (header, codec
        call, future): on the 128-thread host of the MI355X box 1 MB pages (160 jobs per 4e6-row table) lost more to that than they won
        (29 vs 23 ms; profiles/r06c_writeout_*.json), so the default is 4 MB pages of 512K rows (two per column chunk).  The pages are
        written with positional writes from the pool as well (their offsets are known once compressed)."""
        import pyarrow as pa

        if compression not in _CODEC:
            raise ValueError(f"unsupported compression {compression!r}")
        self.schema = schema
        self.compression = None if compression in (None, "none") else compression
        self.codec = _ARROW_CODEC[self.compression] if self.compression else None  # (one pyarrow.Codec object must not be shared by threads)
        self.row_group_rows = int(row_group_rows)
        self.page_rows = max(int(page_rows), 1024)
        self.names = [f.name for f in schema]
        self.np_dtypes = [np.dtype(f.type.to_pandas_dtype()) for f in schema]
        self.store = [_STORE[str(dt)] for dt in self.np_dtypes]
        self.f = open(path, "wb", buffering=0)
        self.f.write(b"PAR1")
        self.pos = 4
        self.seconds = {"compress_wait": 0.0, "file_write": 0.0}  # where write_columns spent its wall time (bench_writeout.py prints it)
        self.row_groups = []  # encoded RowGroup structs
        self.num_rows = 0
        nthreads = threads or min(64, os.cpu_count() or 1)
        self.pool = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="parquet-encode")

    # One data page (DataPageV2) of one column chunk: the definition levels (all 1: one RLE run) stay uncompressed in front, the PLAIN
    # values are compressed straight out of the NumPy buffer (no intermediate copy of the column).
    def _page(self, values: np.ndarray, phys: str):
        n = values.shape[0]
        if phys == "bool":
            body = np.packbits(values.astype(np.uint8), bitorder="little")
        else:
            body = np.ascontiguousarray(values, dtype={"int32": "<i4", "int64": "<i8", "float32": "<f4", "float64": "<f8"}[phys])
        view = memoryview(body).cast("B")
        levels = _varint(n << 1) + b"\x01"  # one run of n definition levels of value 1 (bit width 1)
        if self.codec:
            codec = getattr(_tls, "codec", None)
            if codec is None or _tls.codec_name != self.codec:  # one codec per pool thread: its compression context is not thread-safe
                import pyarrow as pa

                codec = _tls.codec = pa.Codec(self.codec)
                _tls.codec_name = self.codec
            comp = codec.compress(view, asbytes=True)
        else:
            comp = view
        v2 = (_Struct().i32(1, n).i32(2, 0).i32(3, n).i32(4, 0).i32(5, len(levels)).i32(6, 0))  # values, nulls, rows, PLAIN, def / rep level bytes
        head = _Struct().i32(1, 3).i32(2, len(levels) + len(view)).i32(3, len(levels) + len(comp)).struct(8, v2).done()  # DATA_PAGE_V2
        return head + levels, comp, len(levels) + len(view) + len(head), n

    def write_columns(self, columns: dict):
        """Append one table given as {name: 1-D NumPy array} (all of one length, no nulls)."""
        self.commit(self.prepare(columns))

    def prepare(self, columns: dict):
        """First half of write_columns: check the columns and start compressing their pages on the pool.  Touches nothing of the file, so
        the table BEHIND the one being committed may prepare meanwhile (two writer threads of ParticleFile's asynchronous writer: the page
        compression of table k+1 overlaps the file writes of table k).  -> ticket for commit(); tickets must be committed in table order."""
        cols = []
        n = None
        for name, dt in zip(self.names, self.np_dtypes):
            a = np.asarray(columns[name])
            if a.ndim != 1:
                raise ValueError(f"column {name!r} must be one-dimensional")
            if a.dtype != dt:
                a = a.astype(dt)
            n = a.shape[0] if n is None else n
            if a.shape[0] != n:
                raise ValueError("columns of one table must have one length")
            cols.append(a)
        if not n:
            return None
        starts = list(range(0, n, self.row_group_rows))
        jobs = {}
        for g, lo in enumerate(starts):
            hi = min(lo + self.row_group_rows, n)
            for c, a in enumerate(cols):
                for p, plo in enumerate(range(lo, hi, self.page_rows)):
                    jobs[(g, c, p)] = self.pool.submit(self._page, a[plo:min(plo + self.page_rows, hi)], self.store[c][0])
        return {"n": n, "starts": starts, "jobs": jobs, "cols": cols}  # (cols: the buffers the jobs read stay referenced until the commit)

    def commit(self, ticket):
        """Second half: the pages in file order -- offsets, positional writes from the pool, row-group metadata."""
        if ticket is None:
            return
        import time as _time

        n, starts, jobs = ticket["n"], ticket["starts"], ticket["jobs"]
        fd = self.f.fileno()
        writes = []
        for g, lo in enumerate(starts):
            hi = min(lo + self.row_group_rows, n)
            chunks = []
            total_unc = total_comp = 0
            rg_start = self.pos
            for c, name in enumerate(self.names):
                off = self.pos
                unc_size = size = nv = 0
                for p in range(len(range(lo, hi, self.page_rows))):
                    t0 = _time.perf_counter()
                    head, comp, unc_p, nv_p = jobs.pop((g, c, p)).result()
                    self.seconds["compress_wait"] += _time.perf_counter() - t0
                    # positional writes from the pool: the page cache copy of a 100 MB table does not serialise behind one thread
                    writes.append(self.pool.submit(_pwrite_all, fd, (head, comp), self.pos))
                    psize = len(head) + len(comp)
                    self.pos += psize
                    size += psize
                    unc_size += unc_p
                    nv += nv_p
                md = (_Struct().i32(1, _TYPE[self.store[c][0]]).list_(2, 5, [_zigzag(0), _zigzag(3)]).list_(3, 8, [_bin(name)])
                      .i32(4, _CODEC[self.compression]).i64(5, nv).i64(6, unc_size).i64(7, size).i64(9, off))
                chunks.append(_Struct().i64(2, off + size).struct(3, md).done())
                total_unc += unc_size
                total_comp += size
            rg = _Struct().list_(1, 12, chunks).i64(2, total_unc).i64(3, hi - lo).i64(5, rg_start).i64(6, total_comp)
            self.row_groups.append(rg.done())
        t0 = _time.perf_counter()
        for w in writes:
            w.result()
        self.seconds["file_write"] += _time.perf_counter() - t0
        self.num_rows += n

    def close(self):
        if self.f is None:
            return
        # schema: root + one OPTIONAL leaf per column (pyarrow's fields are nullable; the data has no nulls)
        elems = [_Struct().binary(4, "schema").i32(5, len(self.names)).done()]
        for name, (phys, conv) in zip(self.names, self.store):
            e = _Struct().i32(1, _TYPE[phys]).i32(3, 1).binary(4, name)
            if conv is not None:
                e.i32(6, conv)
            elems.append(e.done())
        kv = [_Struct().binary(1, "ARROW:schema").binary(2, base64.b64encode(self.schema.serialize().to_pybytes())).done()]
        for k, v in (self.schema.metadata or {}).items():
            kv.append(_Struct().binary(1, k).binary(2, v).done())
        meta = (_Struct().i32(1, 2).list_(2, 12, elems).i64(3, self.num_rows).list_(4, 12, self.row_groups).list_(5, 12, kv)
                .binary(6, "parcels_amd FastParquetWriter").done())
        _pwrite_all(self.f.fileno(), (meta, struct.pack("<I", len(meta)), b"PAR1"), self.pos)
        self.f.close()
        self.f = None
        self.pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
