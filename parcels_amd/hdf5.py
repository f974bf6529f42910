"""Read-only HDF5 for NetCDF-4 model output, written against the HDF5 File Format Specification (no h5py / netCDF4 needed):
what `NetCDFLevels` (parcels_amd/sources.py) needs to hand ONE time level of a variable to the device ring at a time.

The reference reads such files through xarray + dask (src/parcels/_xarray.py:13-35) and lets WindowedArray pull one level per
request (_windowed_array.py:56-97); here the chunk index of a dataset is read once and a level request touches only the chunks
that intersect it (os.pread: the file is never loaded as a whole).

Supported (everything netCDF-C writes by default, and h5repack / PyTables variants of it):
  superblock v0 - v3; object headers v1 and v2 (continuations, creation-order tracking); groups as symbol tables (B-tree v1 + local
  heap), as compact link messages, or as dense link storage (fractal heap; direct root block or one indirect level); datatypes
  fixed-point and IEEE float of either byte order; dataspaces v1 / v2; data layouts compact, contiguous, chunked (layout message
  v1 - v3 with the B-tree v1 chunk index; v4 with the single-chunk, implicit and fixed-array indices); filters deflate, shuffle,
  fletcher32; fill values; numeric attributes (scale_factor, add_offset, _FillValue, missing_value ...).
Anything else (compound / variable-length types, szip, external or virtual storage, extensible-array / B-tree v2 chunk indices)
raises NotImplementedError with the feature's name.
"""

from __future__ import annotations

import os
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"


class HDF5File:
    def __init__(self, path):
        self.path = str(path)
        self.fd = os.open(self.path, os.O_RDONLY)
        self.size = os.fstat(self.fd).st_size
        self._super()
        self._dsets: dict[str, "Dataset"] = {}

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- raw access --------------------------------------------------------------------------------------------------
    def rd(self, off: int, n: int) -> bytes:
        b = os.pread(self.fd, n, off)
        if len(b) != n:
            raise ValueError(f"{self.path}: short read at {off} (+{n}): truncated file?")
        return b

    def _o(self, b, p):  # an "offset"-sized little-endian integer
        return int.from_bytes(b[p:p + self.so], "little")

    def _l(self, b, p):
        return int.from_bytes(b[p:p + self.sl], "little")

    def _undef(self, a):
        return a == (1 << (8 * self.so)) - 1

    # ---- superblock ----------------------------------------------------------------------------------------------------
    def _super(self):
        base = 0
        while True:  # the signature sits at 0, 512, 1024, ... (a user block may precede it)
            if base + 8 > self.size:
                raise ValueError(f"{self.path}: not an HDF5 file")
            if self.rd(base, 8) == _SIG:
                break
            base = 512 if base == 0 else base * 2
        b = self.rd(base, 128 if base + 128 <= self.size else self.size - base)
        ver = b[8]
        self.sb_version = ver
        if ver in (0, 1):
            self.so, self.sl = b[13], b[14]
            p = 24 if ver == 0 else 28
            self.base = self._o(b, p)
            p += 4 * self.so  # base, free-space info, end of file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            self.root = self._o(b, p + self.so)
        elif ver in (2, 3):
            self.so, self.sl = b[9], b[10]
            self.base = self._o(b, 12)
            self.root = self._o(b, 12 + 3 * self.so)
        else:
            raise NotImplementedError(f"HDF5 superblock version {ver}")
        if self.base in (0, (1 << (8 * self.so)) - 1):
            self.base = base  # addresses are relative to the superblock
        self.base = base if self.base == 0 else self.base

    # ---- object headers ------------------------------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, flags, body bytes)] of the object header at `addr` (continuation blocks followed)."""
        a = self.base + addr
        head = self.rd(a, 16)
        out = []
        if head[:4] == b"OHDR":
            if head[4] != 2:
                raise NotImplementedError(f"object header version {head[4]}")
            flags = head[5]
            p = 6 + (16 if flags & 0x20 else 0) + (4 if flags & 0x10 else 0)
            nsz = 1 << (flags & 3)
            hb = self.rd(a, p + nsz)
            chunk0 = int.from_bytes(hb[p:p + nsz], "little")
            track = bool(flags & 0x04)
            todo = [(a + p + nsz, chunk0)]
            while todo:
                q0, ln = todo.pop(0)
                blk = self.rd(q0, ln)
                q = 0
                while q + 4 <= ln:
                    mtype = blk[q]
                    msize = struct.unpack_from("<H", blk, q + 1)[0]
                    mflags = blk[q + 3]
                    q += 4 + (2 if track else 0)
                    body = blk[q:q + msize]
                    q += msize
                    if mtype == 0x10:
                        off, length = self._o(body, 0), self._l(body, self.so)
                        s = self.base + off
                        if self.rd(s, 4) != b"OCHK":
                            raise ValueError("bad object header continuation block")
                        todo.append((s + 4, length - 8))  # signature and checksum excluded
                    elif mtype != 0:
                        out.append((mtype, mflags, body))
            return out
        if head[0] != 1:
            raise NotImplementedError(f"object header version {head[0]}")
        nmsg = struct.unpack_from("<H", head, 2)[0]
        hsize = struct.unpack_from("<I", head, 8)[0]
        todo = [(a + 16, hsize)]
        while todo and len(out) < nmsg + 64:
            q0, ln = todo.pop(0)
            blk = self.rd(q0, ln)
            q = 0
            while q + 8 <= ln:
                mtype, msize, mflags = struct.unpack_from("<HHB", blk, q)
                q += 8
                body = blk[q:q + msize]
                q += msize
                if mtype == 0x10:
                    todo.append((self.base + self._o(body, 0), self._l(body, self.so)))
                elif mtype != 0:
                    out.append((mtype, mflags, body))
        return out

    # ---- groups -----------------------------------------------------------------------------------------------------------
    def links(self, addr) -> dict:
        """name -> object header address of the hard links of the group at `addr`."""
        out = {}
        for mtype, _f, m in self.messages(addr):
            if mtype == 0x11:  # symbol table: B-tree v1 of symbol-table nodes + local heap of names
                self._symtab(self._o(m, 0), self._o(m, self.so), out)
            elif mtype == 0x06:
                nm, a = self._link_msg(m)
                if a is not None:
                    out[nm] = a
            elif mtype == 0x02:  # link info: dense storage = link messages inside a fractal heap
                p = 2 + (8 if m[1] & 1 else 0)
                heap = self._o(m, p)
                if not self._undef(heap):
                    for blob in self._fractal_objects(heap):
                        q = 0
                        while q < len(blob) and blob[q] == 1:  # link message version 1
                            nm, a, used = self._link_msg(blob[q:], want_len=True)
                            if a is not None:
                                out[nm] = a
                            q += used
        return out

    def _link_msg(self, m, want_len=False):
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
        n = int.from_bytes(m[p:p + ln], "little")
        p += ln
        name = m[p:p + n].decode("utf-8", "replace")
        p += n
        addr = None
        if ltype == 0:
            addr = self._o(m, p)
            p += self.so
        elif ltype == 1:  # soft link: length + path
            p += 2 + struct.unpack_from("<H", m, p)[0]
        else:  # external / user-defined
            p += 2 + struct.unpack_from("<H", m, p)[0]
        return (name, addr, p) if want_len else (name, addr)

    def _symtab(self, btree, heap, out):
        hb = self.rd(self.base + heap, 8 + 2 * self.sl + self.so)
        if hb[:4] != b"HEAP":
            raise ValueError("bad local heap")
        dsize = self._l(hb, 8)
        daddr = self._o(hb, 8 + 2 * self.sl)
        names = self.rd(self.base + daddr, dsize)

        def walk(node):
            nb = self.rd(self.base + node, 8 + 2 * self.so)
            if nb[:4] != b"TREE" or nb[4] != 0:
                raise ValueError("bad group B-tree node")
            level, used = nb[5], struct.unpack_from("<H", nb, 6)[0]
            body = self.rd(self.base + node + 8 + 2 * self.so, (2 * used + 1) * max(self.sl, self.so) + used * self.so)
            p = self.sl  # key 0
            for _ in range(used):
                child = self._o(body, p)
                p += self.so + self.sl
                if level > 0:
                    walk(child)
                else:
                    sn = self.rd(self.base + child, 8)
                    if sn[:4] != b"SNOD":
                        raise ValueError("bad symbol table node")
                    nsym = struct.unpack_from("<H", sn, 6)[0]
                    esz = 2 * self.so + 24
                    eb = self.rd(self.base + child + 8, nsym * esz)
                    for k in range(nsym):
                        noff = self._o(eb, k * esz)
                        oaddr = self._o(eb, k * esz + self.so)
                        end = names.index(b"\0", noff)
                        out[names[noff:end].decode("utf-8", "replace")] = oaddr

        walk(btree)

    def _fractal_objects(self, addr):
        """The managed space of a fractal heap as the byte strings of its direct blocks (objects lie back to back from the start)."""
        h = self.rd(self.base + addr, 512 if self.base + addr + 512 <= self.size else self.size - self.base - addr)
        if h[:4] != b"FRHP":
            raise ValueError("bad fractal heap header")
        p = 5
        p += 2  # heap ID length
        filt_len = struct.unpack_from("<H", h, p)[0]
        p += 2
        hflags = h[p]
        p += 1 + 4  # flags, max size of managed objects
        p += self.sl + self.so + self.sl + self.so  # next huge id, huge b-tree, free space, free-space manager
        p += 4 * self.sl  # managed space, allocated managed space, iterator offset, number of managed objects
        p += 4 * self.sl  # huge size / count, tiny size / count
        width = struct.unpack_from("<H", h, p)[0]
        p += 2
        start = self._l(h, p)
        p += self.sl
        max_direct = self._l(h, p)
        p += self.sl
        max_heap_bits = struct.unpack_from("<H", h, p)[0]
        p += 2 + 2  # + starting rows of the root indirect block
        root = self._o(h, p)
        p += self.so
        cur_rows = struct.unpack_from("<H", h, p)[0]
        if filt_len:
            raise NotImplementedError("filtered fractal heap (compressed link names)")
        if self._undef(root):
            return []
        off_sz = (max_heap_bits + 7) // 8
        dhead = 5 + self.so + off_sz + (4 if hflags & 0x02 else 0)

        def direct(a, size):
            b = self.rd(self.base + a, size)
            if b[:4] != b"FHDB":
                raise ValueError("bad fractal heap direct block")
            return b[dhead:]

        if cur_rows == 0:
            return [direct(root, start)]
        ib = self.rd(self.base + root, 5 + self.so + off_sz + cur_rows * width * self.so)
        if ib[:4] != b"FHIB":
            raise ValueError("bad fractal heap indirect block")
        blobs = []
        q = 5 + self.so + off_sz
        for r in range(cur_rows):
            size = start * (1 << max(0, r - 1))
            for _ in range(width):
                a = self._o(ib, q)
                q += self.so
                if self._undef(a):
                    continue
                if size > max_direct:
                    raise NotImplementedError("fractal heap with nested indirect blocks (a group with thousands of links)")
                blobs.append(direct(a, size))
        return blobs

    # ---- datasets ---------------------------------------------------------------------------------------------------------
    def walk(self, addr=None, prefix=""):
        """{"path/name": object header address} of every object reachable from the root group."""
        out = {}
        seen = set()

        def rec(a, pre):
            if a in seen:
                return
            seen.add(a)
            for nm, ca in self.links(a).items():
                out[pre + nm] = ca
                if any(t in (0x11, 0x02, 0x06) for t, _f, _m in self.messages(ca)) and not any(t == 0x08 for t, _f, _m in self.messages(ca)):
                    rec(ca, pre + nm + "/")

        rec(self.root if addr is None else addr, prefix)
        return out

    def dataset(self, name) -> "Dataset":
        if name not in self._dsets:
            objs = self.walk()
            if name not in objs:
                raise KeyError(f"{self.path}: no object {name!r} (have {sorted(objs)[:20]})")
            self._dsets[name] = Dataset(self, name, objs[name])
        return self._dsets[name]

    def datasets(self) -> dict:
        """name -> Dataset for every object that has a data layout message."""
        out = {}
        for nm, a in self.walk().items():
            if any(t == 0x08 for t, _f, _m in self.messages(a)):
                out[nm] = self.dataset(nm)
        return out


def _parse_dtype(m):
    cls, ver = m[0] & 0x0F, m[0] >> 4
    bits0 = m[1]
    size = struct.unpack_from("<I", m, 4)[0]
    order = ">" if bits0 & 1 else "<"
    if cls == 0:
        return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
    if cls == 1:
        if size not in (2, 4, 8):
            raise NotImplementedError(f"{size}-byte floating point type")
        return np.dtype(f"{order}f{size}")
    names = {2: "time", 3: "string", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum", 9: "variable-length", 10: "array"}
    raise NotImplementedError(f"HDF5 datatype class {names.get(cls, cls)} (version {ver})")


def _parse_space(f, m):
    ver, rank, flags = m[0], m[1], m[2]
    p = 8 if ver == 1 else 4
    if ver == 2 and m[3] == 2:
        return None  # null dataspace
    return tuple(f._l(m, p + f.sl * k) for k in range(rank))


class Dataset:
    """One HDF5 dataset: shape, dtype, attributes, and hyperslab reads along the first axis."""

    def __init__(self, f: HDF5File, name, addr):
        self.f, self.name = f, name
        self.shape = self.dtype = None
        self.layout = None
        self.filters = []
        self.fill = None
        self.attrs = {}
        self._chunks = None
        for mtype, _fl, m in f.messages(addr):
            if mtype == 0x01:
                self.shape = _parse_space(f, m)
            elif mtype == 0x03:
                try:
                    self.dtype = _parse_dtype(m)
                except NotImplementedError as e:
                    self.dtype_error = str(e)
            elif mtype == 0x05 and len(m) >= 4:
                ver = m[0]
                if ver in (1, 2):
                    if ver == 1 or m[3]:
                        n = struct.unpack_from("<I", m, 4)[0] if len(m) >= 8 else 0
                        self.fill = m[8:8 + n] if n else None
                elif ver == 3 and m[1] & 0x20:
                    n = struct.unpack_from("<I", m, 2)[0]
                    self.fill = m[6:6 + n] if n else None
            elif mtype == 0x04 and self.fill is None and len(m) >= 4:
                n = struct.unpack_from("<I", m, 0)[0]
                self.fill = m[4:4 + n] if n else None
            elif mtype == 0x08:
                self.layout = self._parse_layout(m)
            elif mtype == 0x0B:
                self.filters = self._parse_filters(m)
            elif mtype == 0x0C:
                self._parse_attr(m)
        if self.shape is None:
            self.shape = ()

    # ---- messages ------------------------------------------------------------------------------------------------------
    def _parse_layout(self, m):
        f = self.f
        ver = m[0]
        if ver in (1, 2):
            nd, cls = m[1], m[2]
            p = 8
            addr = None
            if cls != 0:
                addr = f._o(m, p)
                p += f.so
            dims = struct.unpack_from(f"<{nd}I", m, p)
            p += 4 * nd
            if cls == 0:
                n = struct.unpack_from("<I", m, p)[0]
                return {"class": "compact", "data": m[p + 4:p + 4 + n]}
            if cls == 1:
                return {"class": "contiguous", "addr": addr, "size": int(np.prod(dims))}
            return {"class": "chunked", "index": "btree1", "addr": addr, "chunk": tuple(dims[:-1]), "elem": dims[-1]}
        if ver == 3:
            cls = m[1]
            if cls == 0:
                n = struct.unpack_from("<H", m, 2)[0]
                return {"class": "compact", "data": m[4:4 + n]}
            if cls == 1:
                return {"class": "contiguous", "addr": f._o(m, 2), "size": f._l(m, 2 + f.so)}
            if cls == 2:
                nd = m[2]
                addr = f._o(m, 3)
                dims = struct.unpack_from(f"<{nd}I", m, 3 + f.so)
                return {"class": "chunked", "index": "btree1", "addr": addr, "chunk": tuple(dims[:-1]), "elem": dims[-1]}
            raise NotImplementedError(f"data layout class {cls}")
        if ver == 4:
            cls = m[1]
            if cls == 0:
                n = struct.unpack_from("<H", m, 2)[0]
                return {"class": "compact", "data": m[4:4 + n]}
            if cls == 1:
                return {"class": "contiguous", "addr": f._o(m, 2), "size": f._l(m, 2 + f.so)}
            if cls == 2:
                flags, nd, enc = m[2], m[3], m[4]
                dims = tuple(int.from_bytes(m[5 + enc * k:5 + enc * (k + 1)], "little") for k in range(nd))
                p = 5 + enc * nd
                itype = m[p]
                p += 1
                lay = {"class": "chunked", "chunk": dims[:-1], "elem": dims[-1]}
                if itype == 1:  # single chunk
                    if flags & 0x02:
                        lay["single_size"] = f._l(m, p)
                        lay["single_mask"] = struct.unpack_from("<I", m, p + f.sl)[0]
                        p += f.sl + 4
                    lay.update(index="single", addr=f._o(m, p))
                elif itype == 2:
                    lay.update(index="implicit", addr=f._o(m, p))
                elif itype == 3:
                    lay.update(index="fixed_array", addr=f._o(m, p + 1))
                else:
                    raise NotImplementedError({4: "extensible-array", 5: "B-tree v2"}.get(itype, str(itype)) + " chunk index (write the file with the default library version bounds)")
                return lay
            raise NotImplementedError("virtual dataset layout")
        raise NotImplementedError(f"data layout message version {ver}")

    def _parse_filters(self, m):
        ver, n = m[0], m[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(n):
            fid = struct.unpack_from("<H", m, p)[0]
            p += 2
            nlen = 0
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", m, p)[0]
                p += 2
            _flags, ncl = struct.unpack_from("<HH", m, p)
            p += 4
            if nlen:
                p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            cd = struct.unpack_from(f"<{ncl}I", m, p)
            p += 4 * ncl
            if ver == 1 and ncl % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _parse_attr(self, m):
        f = self.f
        ver = m[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", m, 2)
        p = 8 + (1 if ver == 3 else 0)
        pad = (lambda n: (n + 7) // 8 * 8) if ver == 1 else (lambda n: n)
        name = m[p:p + nsz].split(b"\0")[0].decode("utf-8", "replace")
        p += pad(nsz)
        tm = m[p:p + tsz]
        p += pad(tsz)
        sm = m[p:p + ssz]
        p += pad(ssz)
        try:
            dt = _parse_dtype(tm)
            shp = _parse_space(f, sm) or ()
        except NotImplementedError:
            return  # strings and the dimension-scale bookkeeping: not needed for numeric reads
        n = int(np.prod(shp)) if shp else 1
        if len(m) - p >= n * dt.itemsize:
            v = np.frombuffer(m, dtype=dt, count=n, offset=p).astype(dt.newbyteorder("="))
            self.attrs[name] = v.reshape(shp) if shp else v[0]

    # ---- data ------------------------------------------------------------------------------------------------------------
    def _unfilter(self, raw, mask):
        for k in range(len(self.filters) - 1, -1, -1):
            if mask & (1 << k):
                continue
            fid, cd = self.filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:  # byte shuffle
                esz = cd[0] if cd else self.dtype.itemsize
                n = len(raw) // esz
                if esz > 1 and n:
                    a = np.frombuffer(raw, np.uint8, count=n * esz).reshape(esz, n).T
                    raw = np.ascontiguousarray(a).tobytes() + raw[n * esz:]
            elif fid == 3:  # fletcher32: trailing checksum
                raw = raw[:-4]
            else:
                names = {4: "szip", 5: "nbit", 6: "scaleoffset", 32001: "blosc", 32004: "lz4", 32015: "zstd"}
                raise NotImplementedError(f"HDF5 filter {names.get(fid, fid)} (re-pack with deflate: nccopy -d4 / h5repack -f GZIP=4)")
        return raw

    def chunk_index(self):
        """{chunk offset tuple: (file address, stored size, filter mask)}, read once."""
        if self._chunks is not None:
            return self._chunks
        f, lay = self.f, self.layout
        idx = {}
        rank = len(lay["chunk"])
        if lay["index"] == "btree1":
            def walk(node):
                hb = f.rd(f.base + node, 8 + 2 * f.so)
                if hb[:4] != b"TREE" or hb[4] != 1:
                    raise ValueError("bad chunk B-tree node")
                level, used = hb[5], struct.unpack_from("<H", hb, 6)[0]
                ksz = 8 + 8 * (rank + 1)
                body = f.rd(f.base + node + 8 + 2 * f.so, used * (ksz + f.so) + ksz)
                p = 0
                for _ in range(used):
                    size, mask = struct.unpack_from("<II", body, p)
                    offs = struct.unpack_from(f"<{rank}Q", body, p + 8)
                    child = f._o(body, p + ksz)
                    p += ksz + f.so
                    if level > 0:
                        walk(child)
                    else:
                        idx[tuple(offs)] = (child, size, mask)

            if not f._undef(lay["addr"]):
                walk(lay["addr"])
        elif lay["index"] == "single":
            if not f._undef(lay["addr"]):
                size = lay.get("single_size", int(np.prod(lay["chunk"])) * lay["elem"])
                idx[(0,) * rank] = (lay["addr"], size, lay.get("single_mask", 0))
        elif lay["index"] in ("implicit", "fixed_array"):
            nch = [-(-s // c) for s, c in zip(self.shape, lay["chunk"])]
            csz = int(np.prod(lay["chunk"])) * lay["elem"]
            grid = np.stack(np.unravel_index(np.arange(int(np.prod(nch))), nch), axis=1) * np.array(lay["chunk"])
            if lay["index"] == "implicit":
                if not f._undef(lay["addr"]):
                    for k, offs in enumerate(grid):
                        idx[tuple(int(v) for v in offs)] = (lay["addr"] + k * csz, csz, 0)
            else:
                hb = f.rd(f.base + lay["addr"], 8 + f.sl + f.so)  # FAHD: version, client id, entry size, page bits, entries, data block
                if hb[:4] != b"FAHD":
                    raise ValueError("bad fixed array header")
                client, esz, page_bits = hb[5], hb[6], hb[7]
                nent = f._l(hb, 8)
                dblk = f._o(hb, 8 + f.sl)
                if not f._undef(dblk):
                    if nent > (1 << page_bits):
                        raise NotImplementedError("paged fixed-array chunk index (more than 2^page_bits chunks)")
                    dh = 6 + f.so  # FADB: signature, version, client id, header address
                    db = f.rd(f.base + dblk, dh + nent * esz)
                    if db[:4] != b"FADB":
                        raise ValueError("bad fixed array data block")
                    for k in range(min(nent, len(grid))):
                        e = db[dh + k * esz:dh + (k + 1) * esz]
                        a = f._o(e, 0)
                        if f._undef(a):
                            continue
                        if client == 1:  # filtered chunks: address, stored size, filter mask
                            ssz = esz - f.so - 4
                            idx[tuple(int(v) for v in grid[k])] = (a, int.from_bytes(e[f.so:f.so + ssz], "little"), struct.unpack_from("<I", e, f.so + ssz)[0])
                        else:
                            idx[tuple(int(v) for v in grid[k])] = (a, csz, 0)
        self._chunks = idx
        return idx

    def _fill_value(self):
        if self.fill is not None and len(self.fill) >= self.dtype.itemsize:
            return np.frombuffer(self.fill, dtype=self.dtype, count=1)[0]
        return 0

    def read(self, first=None) -> np.ndarray:
        """The whole dataset, or -- `first` = k -- the hyperslab [k] along the first axis (shape[1:])."""
        if self.dtype is None:
            raise NotImplementedError(getattr(self, "dtype_error", "dataset without a numeric datatype"))
        f, lay = self.f, self.layout
        shape = tuple(self.shape)
        native = self.dtype.newbyteorder("=")
        if first is not None and (not shape or not 0 <= first < shape[0]):
            raise IndexError(f"{self.name}: index {first} out of range for axis 0 of {shape}")
        out_shape = shape if first is None else shape[1:]
        n_out = int(np.prod(out_shape)) if out_shape else 1
        if lay["class"] in ("compact", "contiguous"):
            if lay["class"] == "compact":
                buf = lay["data"]
                a = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(shape)) if shape else 1).reshape(shape)
                return (a if first is None else a[first]).astype(native)
            if f._undef(lay["addr"]):  # never written: all fill value
                return np.full(out_shape, self._fill_value(), dtype=native)
            off = 0 if first is None else first * n_out * self.dtype.itemsize
            raw = f.rd(f.base + lay["addr"] + off, n_out * self.dtype.itemsize)
            return np.frombuffer(raw, dtype=self.dtype, count=n_out).reshape(out_shape).astype(native)
        # chunked
        chunk = lay["chunk"]
        idx = self.chunk_index()
        out = np.full(shape if first is None else (1,) + shape[1:], self._fill_value(), dtype=native)
        lo0 = 0 if first is None else first
        hi0 = shape[0] if first is None else first + 1
        c0_lo = lo0 // chunk[0] * chunk[0]
        ranges = [range(c0_lo, hi0, chunk[0])] + [range(0, s, c) for s, c in zip(shape[1:], chunk[1:])]
        for offs in np.ndindex(*[len(r) for r in ranges]):
            o = tuple(r[i] for r, i in zip(ranges, offs))
            ent = idx.get(o)
            if ent is None:
                continue
            raw = f.rd(f.base + ent[0], ent[1])
            raw = self._unfilter(raw, ent[2])
            blk = np.frombuffer(raw, dtype=self.dtype, count=int(np.prod(chunk))).reshape(chunk)
            src = [slice(max(lo0 - o[0], 0), min(hi0 - o[0], chunk[0]))] + [slice(0, min(c, s - oo)) for c, s, oo in zip(chunk[1:], shape[1:], o[1:])]
            dst = [slice(o[0] + src[0].start - lo0, o[0] + src[0].stop - lo0)] + [slice(oo, oo + sl.stop) for oo, sl in zip(o[1:], src[1:])]
            out[tuple(dst)] = blk[tuple(src)]
        return out if first is None else out[0]


# ---- NetCDF classic (CDF-1 / CDF-2 / CDF-5): the format of NetCDF-3 files ---------------------------------------------------------
class NetCDF3File:
    """The classic NetCDF format (magic "CDF\\x01" / "\\x02" / "\\x05"): a header followed by the variables' big-endian arrays; the
    variables along the record (unlimited) dimension are interleaved record by record."""

    _TYPES = {1: ">i1", 2: "S1", 3: ">i2", 4: ">i4", 5: ">f4", 6: ">f8", 7: ">u1", 8: ">u2", 9: ">u4", 10: ">i8", 11: ">u8"}

    def __init__(self, path):
        self.path = str(path)
        self.fd = os.open(self.path, os.O_RDONLY)
        size = os.fstat(self.fd).st_size
        want = 1 << 22
        while True:  # the header has no length field: parse a prefix of the file, a longer one when it does not fit
            head = os.pread(self.fd, min(size, want), 0)
            if head[:3] != b"CDF" or head[3] not in (1, 2, 5):
                raise ValueError(f"{self.path}: not a classic NetCDF file")
            try:
                self._parse_header(head)
                # byte slices (names, attribute values) never raise when the prefix ends inside them: an item that straddles the end of
                # the prefix shows as a parse offset beyond it
                if self._p > len(head):
                    raise IndexError("header item beyond the prefix")
                break
            except (struct.error, IndexError, KeyError, UnicodeDecodeError):
                if want >= size:
                    raise ValueError(f"{self.path}: truncated or malformed classic NetCDF header") from None
                want *= 8
        if self.numrecs in (0xFFFFFFFF, -1) or (self.ver == 5 and self.numrecs == 0xFFFFFFFFFFFFFFFF):
            # the "streaming" sentinel (the writer did not go back to fill in the record count): what the file size holds
            first = min((v["begin"] for v in self.vars.values() if v["dimids"] and v["dimids"][0] == self.recdim), default=None)
            self.numrecs = 0 if first is None or not self.recsize else max((size - first) // self.recsize, 0)

    def _parse_header(self, head):
        self.ver = head[3]
        self._b, self._p = head, 4
        nn = self._int if self.ver < 5 else self._int64
        self.numrecs = nn()
        self.dims = []
        tag, n = self._int(), nn()
        if tag == 0x0A:
            for _ in range(n):
                nm = self._name()
                self.dims.append((nm, nn()))
        self.gattrs = self._atts()
        self.vars = {}
        tag, n = self._int(), nn()
        if tag == 0x0B:
            for _ in range(n):
                nm = self._name()
                nd = nn()
                dimids = [nn() for _ in range(nd)]
                at = self._atts()
                typ = self._int()
                vsize = nn()
                begin = self._int() if self.ver == 1 else self._int64()
                self.vars[nm] = {"dimids": dimids, "attrs": at, "dtype": np.dtype(self._TYPES[typ]), "vsize": vsize, "begin": begin}
        recdim = [k for k, (_nm, ln) in enumerate(self.dims) if ln == 0]
        self.recdim = recdim[0] if recdim else -1
        self.recsize = sum(v["vsize"] for v in self.vars.values() if v["dimids"] and v["dimids"][0] == self.recdim)
        nrec_vars = sum(1 for v in self.vars.values() if v["dimids"] and v["dimids"][0] == self.recdim)
        if nrec_vars == 1:  # a single record variable is not padded to 4 bytes
            v = next(v for v in self.vars.values() if v["dimids"] and v["dimids"][0] == self.recdim)
            self.recsize = int(np.prod([self.dims[d][1] for d in v["dimids"][1:]])) * v["dtype"].itemsize if len(v["dimids"]) > 1 else v["dtype"].itemsize
        del self._b

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _int(self):
        v = struct.unpack_from(">i", self._b, self._p)[0]
        self._p += 4
        return v

    def _int64(self):
        v = struct.unpack_from(">q", self._b, self._p)[0]
        self._p += 8
        return v

    def _name(self):
        n = self._int() if self.ver < 5 else self._int64()
        s = self._b[self._p:self._p + n].decode("utf-8", "replace")
        self._p += (n + 3) // 4 * 4
        return s

    def _atts(self):
        nn = self._int if self.ver < 5 else self._int64
        tag, n = self._int(), nn()
        out = {}
        if tag != 0x0C:
            return out
        for _ in range(n):
            nm = self._name()
            typ = self._int()
            cnt = nn()
            dt = np.dtype(self._TYPES[typ])
            raw = self._b[self._p:self._p + cnt * dt.itemsize]
            self._p += (cnt * dt.itemsize + 3) // 4 * 4
            if dt.kind == "S":
                out[nm] = raw.decode("utf-8", "replace")
            else:
                v = np.frombuffer(raw, dtype=dt, count=cnt).astype(dt.newbyteorder("="))
                out[nm] = v[0] if cnt == 1 else v
        return out

    def shape(self, name):
        v = self.vars[name]
        return tuple(self.numrecs if d == self.recdim else self.dims[d][1] for d in v["dimids"])

    def read(self, name, first=None):
        v = self.vars[name]
        shp = self.shape(name)
        dt = v["dtype"]
        rec = bool(v["dimids"]) and v["dimids"][0] == self.recdim
        inner = shp[1:] if shp else ()
        n_in = int(np.prod(inner)) if inner else 1
        if first is not None:
            if not shp or not 0 <= first < shp[0]:
                raise IndexError(f"{name}: index {first} out of range for axis 0 of {shp}")
            off = v["begin"] + first * (self.recsize if rec else n_in * dt.itemsize)
            raw = os.pread(self.fd, n_in * dt.itemsize, off)
            return np.frombuffer(raw, dtype=dt, count=n_in).reshape(inner).astype(dt.newbyteorder("="))
        if not rec:
            n = int(np.prod(shp)) if shp else 1
            raw = os.pread(self.fd, n * dt.itemsize, v["begin"])
            return np.frombuffer(raw, dtype=dt, count=n).reshape(shp).astype(dt.newbyteorder("="))
        return np.stack([self.read(name, k) for k in range(shp[0])]) if shp[0] else np.zeros(shp, dt.newbyteorder("="))
