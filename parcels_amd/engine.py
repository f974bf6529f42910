"""DeviceEngine: the device-resident copy of a FieldSet and the driver of ``pk_execute``.

Replaces the body of the reference's ``Kernel.execute`` (src/parcels/_core/kernel.py:174-247) and its field backend
(``WindowedArray``, src/parcels/_core/_windowed_array.py:25-113):

* grids (coordinates, ravel dims, C-grid offsets, Morton hash table) and field time levels live in HBM;
* when all time levels of a field fit the memory budget they are uploaded once; otherwise each field keeps a ring
  of ``nslots`` levels and the next level is copied on a second HIP stream while the RK sub-steps of the current
  levels run (particles whose next step would leave the resident window pause and are resumed by the next launch,
  so the trajectory is independent of the window size);
* particle columns are bound from the ParticleSet's NumPy SoA dict, copied once per ``execute`` and advanced on
  the device until ``endtime``.
"""

from __future__ import annotations

import ctypes as C
import os
import warnings

import numpy as np

from . import _hip
from .columns import LazyColumns
from .field import Field, VectorField
from .interpolators import CGrid_Velocity, XConstantField
from .statuscodes import StatusCode


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _RawView:
    """dict-like access to the arrays of a LazyColumns set as they are (no download, no dirty mark)."""

    def __init__(self, lc):
        self.lc = lc

    def __getitem__(self, k):
        return self.lc.raw(k)

    def __setitem__(self, k, v):
        self.lc.set_raw(k, v)

    def get(self, k, default=None):
        return self.lc.raw(k) if k in self.lc else default


class _LazyLevels:
    """A level source (parcels_amd.sources) behind the `host[level]` indexing the engine uses for NumPy arrays."""

    def __init__(self, src, dtype, fill_nan=True):
        self.src, self.dtype = src, np.dtype(dtype)
        self.shape = tuple(src.shape)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.fill_nan = bool(fill_nan)

    def __getitem__(self, level):
        """One level ready for the upload, whatever the source (a LevelSource or any object with read_level / shape / dtype): shape
        checked, NaN -> 0 per the model's setting (model.py:135-143), C-contiguous, target dtype."""
        a = np.asarray(self.src.read_level(int(level)))
        if tuple(a.shape) != self.shape[1:]:
            raise ValueError(f"level {level} of {type(self.src).__name__} has shape {a.shape}, expected {self.shape[1:]}")
        if self.fill_nan and np.issubdtype(a.dtype, np.floating) and np.isnan(a).any():
            a = np.nan_to_num(a, nan=0.0)
        return np.ascontiguousarray(a, dtype=self.dtype)


class DeviceEngine:
    def __init__(self, fieldset, device: int = 0, nslots: int | None = None, memory_fraction: float = 0.6,
                 hash_build: str | None = None, neighbour_probe: int = 0):
        # hash_build: "device" (default) builds the Morton table of a curvilinear grid on the GPU (csrc/pk_hashbuild.hip);
        # "host" uploads parcels_amd.spatialhash.SpatialHash's table.  A grid whose host table already exists uploads it.
        self.nslots_request = nslots  # (FieldSet.to_windowed_arrays compares it with a later request)
        self.hash_build = hash_build or os.environ.get("PARCELS_AMD_HASH_BUILD", "device")
        if self.hash_build not in ("device", "host"):
            raise ValueError("hash_build must be 'device' or 'host'")
        # neighbour_probe: pk_grid_desc.neighbour_probe (0 automatic, 1 always, -1 never; include/parcels_hip.h)
        self.neighbour_probe = int(neighbour_probe)
        self.fieldset = fieldset
        self.device = int(device)
        self.ctx = _hip.Context(self.device)
        self.lib = self.ctx.lib
        self._keep = []  # host buffers referenced by the library during calls
        self.grids = list(fieldset.gridset)
        self.grid_ids = []
        for g in self.grids:
            self.grid_ids.append(self._create_grid(g))
        # scalar fields in fieldset order
        self.scalar_fields = [f for f in fieldset.fields.values() if isinstance(f, Field)]
        if len(self.scalar_fields) > _hip.PK_MAX_FIELDS:
            raise ValueError(f"at most {_hip.PK_MAX_FIELDS} fields are supported on the device")
        self.field_ids: dict[str, int] = {}
        self.field_host: dict[str, np.ndarray] = {}
        self.field_nslots: dict[str, int] = {}
        self._plan_and_create_fields(nslots, memory_fraction)
        self._bound_sig = None
        self._bound = None
        self._sorted_t = None  # model time at which the device rows were last cell-sorted (None: host order / unknown)
        # what crossed PCIe for the particle columns: calls and column counts (tests and bench.py's `repeat_execute` leg read it)
        self.transfers = {"h2d_full": 0, "h2d_columns": 0, "d2h_full": 0, "d2h_columns": 0, "columns_up": 0, "columns_down": 0}
        self.comm = None  # (rank, world) once the context has joined an RCCL communicator (comm_init)
        self.comm_stats = {"gathers": 0, "allreduces": 0}
        self.comm_last_counts = None
        self._next_dt_f32 = None
        self.device_variables: list[str] = []  # user Variables bound as extra device columns (set by Kernel: SampleField targets)
        # a sharded ParticleSet is one batch: hooks of parcels_amd.distributed.batch_agreement (set by ParticleSet.execute for a collective run)
        self.agree_min = None
        self.agree_codes = None
        self.last_stats: dict | None = None

    # ---- grids -----------------------------------------------------------------------------------------------
    def _create_grid(self, g) -> int:
        d = _hip.GridDesc()
        axes = g.axes
        lon, lat = np.asarray(g.lon), np.asarray(g.lat)
        depth = np.asarray(g.depth) if "Z" in axes else None
        d.kind = 1 if g.is_curvilinear else 0
        d.spherical = int(g._mesh.is_spherical())
        d.has_x, d.has_y, d.has_z = int("X" in axes), int("Y" in axes), int("Z" in axes)
        d.nx = lon.shape[-1]
        d.ny = lat.shape[0]
        d.nz = depth.shape[0] if depth is not None else 0
        d.xdim = g.xdim if "X" in axes else 0
        d.ydim = g.ydim if "Y" in axes else 0
        d.zdim = g.zdim if "Z" in axes else 0
        off = g.offsets()
        d.off_x, d.off_y, d.off_z = off["X"], off["Y"], off["Z"]
        d.lon_f32 = int(lon.dtype == np.float32)
        d.lat_f32 = int(lat.dtype == np.float32)
        d.depth_f32 = int(depth is not None and depth.dtype == np.float32)
        d.deg2m = float(g.deg2m)
        lon64 = np.ascontiguousarray(lon, dtype=np.float64)
        lat64 = np.ascontiguousarray(lat, dtype=np.float64)
        dep64 = np.ascontiguousarray(depth, dtype=np.float64) if depth is not None else None
        d.lon, d.lat, d.depth = _ptr(lon64), _ptr(lat64), _ptr(dep64)
        keep = [lon64, lat64, dep64]
        if d.kind == 1 and d.spherical:
            # the reference's own expression (index_search.py:439-450 on deg2rad of the corner lon/lat)
            lonr, latr = np.deg2rad(lon64), np.deg2rad(lat64)
            xyz = np.ascontiguousarray(np.stack((np.cos(lonr) * np.cos(latr), np.sin(lonr) * np.cos(latr), np.sin(latr))))
            keep.append(xyz)
            d.node_xyz = _ptr(xyz)
        if d.kind == 1 and (self.hash_build == "host" or g._spatialhash is not None):
            t = g.get_spatial_hash().table()
            keys = np.ascontiguousarray(t["keys"], dtype=np.uint32)
            starts = np.ascontiguousarray(t["starts"], dtype=np.int64)
            counts = np.ascontiguousarray(t["counts"], dtype=np.int64)
            faces = np.ascontiguousarray(t["faces"], dtype=np.uint32)
            keep += [keys, starts, counts, faces]
            d.h_keys, d.h_starts, d.h_counts, d.h_faces = _ptr(keys), _ptr(starts), _ptr(counts), _ptr(faces)
            d.h_nkeys = keys.size
            d.h_nentries = faces.size
            d.h_bitwidth = int(t["bitwidth"])
            for i, v in enumerate(np.asarray(t["bbox"], dtype=np.float64)):
                d.h_bbox[i] = float(v)
        d.neighbour_probe = self.neighbour_probe
        gid = C.c_int32(-1)
        self.ctx.check(self.lib.pk_grid_create(self.ctx.handle, C.byref(d), C.byref(gid)), "pk_grid_create")
        del keep  # copied on call
        return gid.value

    def hash_table(self, igrid: int = 0) -> dict:
        """The spatial-hash table resident on the device for grid ``igrid`` (same dict layout as SpatialHash.table())."""
        info = _hip.HashInfo()
        gid = self.grid_ids[igrid]
        self.ctx.check(self.lib.pk_grid_hash_info(self.ctx.handle, gid, C.byref(info)), "pk_grid_hash_info")
        keys = np.empty(info.nkeys, np.uint32)
        starts = np.empty(info.nkeys, np.int64)
        counts = np.empty(info.nkeys, np.int64)
        faces = np.empty(info.nentries, np.uint32)
        self.ctx.check(self.lib.pk_grid_hash_download(self.ctx.handle, gid, _ptr(keys), _ptr(starts), _ptr(counts), _ptr(faces)),
                       "pk_grid_hash_download")
        return dict(keys=keys, starts=starts, counts=counts, faces=faces, bitwidth=int(info.bitwidth),
                    bbox=np.array(list(info.bbox), dtype=np.float64), neighbour_probe=int(info.neighbour_probe))

    # ---- fields ----------------------------------------------------------------------------------------------
    def _plan_and_create_fields(self, nslots, memory_fraction):
        fs = self.fieldset
        # U, V, W of a vector field must share one dtype on the device; widening f32 -> f64 is exact
        share = {}
        for f in fs.fields.values():
            if isinstance(f, VectorField):
                comps = [c for c in (f.U, f.V, f.W) if c is not None]
                dts = {np.dtype(c.data.data.dtype) for c in comps}
                tgt = np.float32 if dts == {np.dtype(np.float32)} else np.float64
                for c in comps:
                    share[c.name] = np.float64 if share.get(c.name) is np.float64 else tgt
        hosts = {}
        from .sources import is_level_source

        for f in self.scalar_fields:
            a = f.data.data
            tgt = share.get(f.name, np.float32 if np.dtype(a.dtype) == np.float32 else np.float64)
            if is_level_source(a):
                fill = getattr(f.model, "level_fill_nan", {}).get(f.name, getattr(a, "fill_nan", True))
                hosts[f.name] = _LazyLevels(a, tgt, fill_nan=fill)  # levels are read (and converted) when `_upload` asks for them
            else:  # no copy for a C-contiguous array / np.memmap of the target dtype
                hosts[f.name] = np.ascontiguousarray(np.asarray(a), dtype=tgt)
        # residency plan: keep all levels if they fit the budget, else a ring
        info = self.ctx.device_info()
        budget = memory_fraction * info["free_mem"]
        total_all = sum(h.nbytes for h in hosts.values())
        for f in self.scalar_fields:
            h = hosts[f.name]
            nt = h.shape[0]
            if nslots is not None:
                ns = nt if nslots >= nt else max(int(nslots), 2)
            elif total_all <= budget:
                ns = nt
            else:
                level_bytes = sum(x.nbytes // x.shape[0] for x in hosts.values() if x.shape[0] > 1)
                ns = max(4, min(nt, int(budget // max(level_bytes, 1))))
                ns = nt if h.shape[0] == 1 else min(ns, nt)
            self.field_nslots[f.name] = ns
        # C-grid velocity components are stored interleaved ({U,V,W} per cell): the staggered corner values of one
        # evaluation then share cache lines, which is what bounds the sparse NEMO-size configuration
        pack: dict[str, tuple[str, int]] = {}  # field name -> (leader name, group size)
        self.pack_groups: dict[str, list[str]] = {}  # leader name -> member names in component order
        self.pack_leader_of: dict[str, str] = {}
        cgrid_vectors = [vf for vf in fs.fields.values() if isinstance(vf, VectorField) and isinstance(vf.interp_method, CGrid_Velocity)]
        cgrid_vectors.sort(key=lambda vf: -(3 if vf.W is not None else 2))  # UVW before UV: the larger group wins
        for vf in cgrid_vectors:
            comps = [c for c in (vf.U, vf.V, vf.W) if c is not None]
            if any(c.name in pack for c in comps):
                continue  # already part of a (larger) group
            ref = hosts[comps[0].name]
            same = all(hosts[c.name].shape == ref.shape and hosts[c.name].dtype == ref.dtype
                       and self.field_nslots[c.name] == self.field_nslots[comps[0].name] for c in comps)
            if same:
                order = [f.name for f in self.scalar_fields if f.name in {c.name for c in comps}]  # creation order
                for nme in order:
                    pack[nme] = (order[0], len(order))
                self.pack_groups[order[0]] = order
        for f in self.scalar_fields:
            h = hosts[f.name]
            dims = f.data.dims
            d = _hip.FieldDesc()
            d.pack_leader = -1
            if f.name in pack:
                leader, cnt = pack[f.name]
                if leader == f.name:
                    d.pack_count = cnt
                else:
                    d.pack_leader = self.field_ids[leader]
            d.grid = self.grid_ids[self.grids.index(f.grid)]
            d.dtype = _hip.PK_F64 if h.dtype == np.float64 else _hip.PK_F32
            d.nt, d.nz, d.ny, d.nx = h.shape
            d2a = f.grid.sgrid_metadata.dim_to_axis()
            d.has_t = int(dims[0] == "time")
            d.has_z = int(d2a.get(dims[1]) == "Z")
            d.has_y = int(d2a.get(dims[2]) == "Y")
            d.has_x = int(d2a.get(dims[3]) == "X")
            tflt = f.model.time_flt
            has_ti = f.time_interval is not None and tflt is not None
            d.has_time_interval = int(has_ti)
            is_const = False
            d.is_const = 0
            try:
                is_const = isinstance(f.interp_method, XConstantField)
                d.is_const = int(f.interp_method.kind)  # scalar interpolator code (include/parcels_hip.h)
            except AttributeError:
                pass
            if is_const:
                d.has_y = d.has_x = 1
            d.nslots = self.field_nslots[f.name]
            tarr = np.ascontiguousarray(tflt, dtype=np.float64) if has_ti else np.zeros(h.shape[0])
            d.time = _ptr(tarr)
            fid = C.c_int32(-1)
            self.ctx.check(self.lib.pk_field_create(self.ctx.handle, C.byref(d), C.byref(fid)), f"pk_field_create({f.name})")
            self.field_ids[f.name] = fid.value
            self.field_host[f.name] = h
        for leader, members in self.pack_groups.items():
            for m in members:
                self.pack_leader_of[m] = leader
        for f in self.scalar_fields:  # resident fields: all levels now (packed groups need every member created first)
            if self.field_nslots[f.name] >= self.field_host[f.name].shape[0]:
                for lv in range(self.field_host[f.name].shape[0]):
                    self._upload(f.name, lv, asynchronous=False)
        self.windowed = any(self.field_nslots[f.name] < self.field_host[f.name].shape[0] for f in self.scalar_fields)
        self.exact_error_stop = True  # kernel.py:236-245: after an error every particle stops where the reference's batch loop did

    def _upload(self, name, level, asynchronous):
        leader = self.pack_leader_of.get(name)
        if leader is not None:
            if leader != name:
                return  # travels with its group leader
            # packed {U,V,W} group: one call, interleaved by the host threads that stage the level for the DMA anyway
            members = self.pack_groups[leader]
            lvls = [self.field_host[m][level] for m in members]
            ptrs = (C.c_void_p * len(members))(*[l.ctypes.data for l in lvls])
            self.ctx.check(
                self.lib.pk_field_upload_group_level(self.ctx.handle, self.field_ids[leader], int(level), ptrs, len(members), int(asynchronous)),
                f"pk_field_upload_group_level({'+'.join(members)}, {level})",
            )
            return
        h = self.field_host[name]
        lvl = h[level]
        self.ctx.check(
            self.lib.pk_field_upload_level(self.ctx.handle, self.field_ids[name], int(level), _ptr(lvl), int(asynchronous)),
            f"pk_field_upload_level({name}, {level})",
        )

    def _slots(self, name):
        ns = self.field_nslots[name]
        arr = (C.c_int32 * ns)()
        n = C.c_int32()
        self.ctx.check(self.lib.pk_field_slots(self.ctx.handle, self.field_ids[name], arr, C.byref(n)), "pk_field_slots")
        return list(arr)

    # field-slab streaming --------------------------------------------------------------------------------------
    def _windowed_fields(self):
        return [f for f in self.scalar_fields if self.field_nslots[f.name] < self.field_host[f.name].shape[0]]

    def _plan_window(self, f, t_live: float, sign: int):
        """Levels of field ``f`` that must be committed for the next launch and the level to prefetch behind them.
        Every windowed field is planned on ITS OWN time axis (fields of one FieldSet may come from models with different
        level times); the library pauses a particle where the intersection of the fields' resident windows ends."""
        time = f.model.time_flt
        nt = len(time)
        ns = self.field_nslots[f.name]
        ncommit = ns - 1 if ns > 2 else ns
        if sign > 0:
            k0 = int(np.clip(np.searchsorted(time, t_live, side="right") - 1, 0, nt - 1))
            want = list(range(k0, min(k0 + ncommit, nt)))
            nxt = k0 + ncommit if k0 + ncommit < nt else None
        else:
            k1 = int(np.clip(np.searchsorted(time, t_live, side="left"), 0, nt - 1))
            want = list(range(max(k1 - ncommit + 1, 0), k1 + 1))
            nxt = k1 - ncommit if k1 - ncommit >= 0 else None
        if ncommit >= ns:
            nxt = None  # no spare slot to prefetch into
        return want, nxt

    def _commit_window(self, t_live: float, sign: int):
        """Make the wanted levels resident (normally they were prefetched during the previous launch) and drop every other
        level: what an earlier run / a far-away particle left in the ring would make the resident set non-contiguous."""
        self.ctx.check(self.lib.pk_field_sync(self.ctx.handle), "pk_field_sync")  # commit earlier prefetches
        plan = {}
        for f in self._windowed_fields():
            want, nxt = self._plan_window(f, t_live, sign)
            # keep the wanted levels and the prefetch target (it usually arrived during the previous launch and is what lets a
            # step that straddles the last wanted level proceed); everything else in the ring is stale
            keep = want + ([nxt] if nxt is not None else [])
            self.ctx.check(self.lib.pk_field_evict_outside(self.ctx.handle, self.field_ids[f.name], int(min(keep)), int(max(keep))),
                           "pk_field_evict_outside")
            have = set(self._slots(f.name))
            for lv in want:
                if lv not in have:
                    self._upload(f.name, lv, asynchronous=False)
            plan[f.name] = nxt
        return plan

    def _prefetch(self, plan):
        """Stage and enqueue the next level of every windowed field on the copy stream WHILE the advection kernel runs (the
        host-side staging memcpy -- or the page-ins of a memory-mapped file -- and the DMA both overlap the RK sub-steps)."""
        issued = False
        for f in self._windowed_fields() if plan else ():
            nxt = plan.get(f.name)
            if nxt is not None and nxt not in set(self._slots(f.name)):
                self._upload(f.name, nxt, asynchronous=True)
                issued = True
        return issued

    # ---- particles -------------------------------------------------------------------------------------------
    def _particles_desc(self, data: dict):
        """Validate the SoA dict (particle.py:182-222) and describe it for the library (pk_particles_desc)."""
        src = None
        if isinstance(data, LazyColumns):  # (the arrays as they are: no download, no dirty mark -- bind_particles released the set)
            src = data
            src._extra_refs.pop("next_dt", None)
            data = _RawView(src)
        n = data["x"].shape[0]
        for k in ("t", "z", "y", "x", "dz", "dy", "dx", "dt", "state", "ei", "particle_id"):
            a = data[k]
            if not a.flags["C_CONTIGUOUS"]:
                data[k] = np.ascontiguousarray(a)
        sdt = data["x"].dtype
        if any(data[k].dtype != sdt for k in ("z", "y", "dz", "dy", "dx")) or sdt not in (np.float32, np.float64):
            raise TypeError("z, y, x, dz, dy, dx must share one dtype (float32 or float64)")
        if data["t"].dtype != np.float64 or data["dt"].dtype != np.float64:
            raise TypeError("t and dt must be float64")
        if data["state"].dtype != np.int32 or data["ei"].dtype != np.int32 or data["particle_id"].dtype != np.int64:
            raise TypeError("state/ei must be int32 and particle_id int64")
        d = _hip.ParticlesDesc()
        d.n = n
        d.ngrids = data["ei"].shape[1]
        d.spatial_dtype = _hip.PK_F32 if sdt == np.float32 else _hip.PK_F64
        d.t = _ptr(data["t"])
        d.z, d.y, d.x = _ptr(data["z"]), _ptr(data["y"]), _ptr(data["x"])
        d.dz, d.dy, d.dx = _ptr(data["dz"]), _ptr(data["dy"]), _ptr(data["dx"])
        d.dt = _ptr(data["dt"])
        nd = data.get("next_dt")
        self._next_dt_f32 = None
        if nd is not None and nd.dtype == np.float32:
            # Variable("next_dt") defaults to float32 (particle.py:36-60).  The device column is f64; the kernel rounds every
            # store to f32 (pk_exec_params.next_dt_f32), so this f64 shadow converts back exactly.
            self._next_dt_f32 = (nd, nd.astype(np.float64))
            nd = self._next_dt_f32[1]
            if src is not None:
                src._extra_refs["next_dt"] = 1  # (this tuple: not a holder outside the set, columns.py)
        elif nd is not None and nd.dtype != np.float64:
            raise TypeError("next_dt must be float32 or float64")
        d.next_dt = _ptr(nd) if nd is not None else None
        d.state, d.ei, d.particle_id = _ptr(data["state"]), _ptr(data["ei"]), _ptr(data["particle_id"])
        d.n_extra = len(self.device_variables)
        for k, name in enumerate(self.device_variables):
            a = data[name]
            if not a.flags["C_CONTIGUOUS"]:
                a = data[name] = np.ascontiguousarray(a)
            if a.dtype not in (np.float32, np.float64, np.int32, np.int64) or a.shape != (n,):
                raise TypeError(f"device Variable '{name}' must be a float32 / float64 / int32 / int64 column of length n")
            # the library moves these columns as opaque 4- or 8-byte elements (only PK_KERNEL_SAMPLE_FIELD interprets its targets, and
            # Kernel restricts those to float Variables): integer Variables of compiled user kernels travel under the float code of their size
            d.extra_dtype[k] = _hip.PK_F32 if a.dtype.itemsize == 4 else _hip.PK_F64
            d.extra[k] = a.ctypes.data
        return d

    def _column_bit(self, name):
        if name in self.device_variables:
            return _hip.PK_COL_EXTRA0 << self.device_variables.index(name)
        return _hip.COLUMN_BITS.get(name, 0)  # other user Variables live on the host only (no kernel writes them)

    def bind_particles(self, data: dict):
        """Describe the host columns to the library (device columns sized, nothing copied).  A LazyColumns set that was resident -- this
        one or another ParticleSet's -- first gets its stale columns back: from here on the HOST arrays are what counts, until h2d()."""
        prev = getattr(self, "_bound", None)
        for lc in {id(prev): prev, id(data): data}.values():
            if isinstance(lc, LazyColumns):
                lc.release()
        d = self._particles_desc(data)
        self.ctx.check(self.lib.pk_particles_bind(self.ctx.handle, C.byref(d)), "pk_particles_bind")
        self._bound = data
        self._bound_devvars = tuple(self.device_variables)

    def attach(self, data) -> bool:
        """Make the device rows current for `data` with as little PCIe traffic as the host's accesses since the last launch allow
        (parcels_amd/columns.py).  True: the set was still resident -- only the columns the host touched were uploaded."""
        if (isinstance(data, LazyColumns) and data.resident() and data._engine is self and tuple(self.device_variables) == getattr(self, "_bound_devvars", None)
                and not os.environ.get("PARCELS_AMD_NO_RESIDENT")):
            dirty = sorted(k for k in data._dirty if self._column_bit(k))
            if dirty:
                self.h2d_columns(dirty)
            data._dirty.clear()
            return True
        self.bind_particles(data)
        self.h2d()
        return False

    def h2d_columns(self, columns):
        """Upload the named host columns into the device rows they belong to (the row order of the cell sort is kept)."""
        mask = 0
        for name in columns:
            mask |= self._column_bit(name)
        if self._next_dt_f32 is not None and "next_dt" in columns:
            self._next_dt_f32[1][:] = self._next_dt_f32[0]
        if mask:
            self.ctx.check(self.lib.pk_particles_h2d_columns(self.ctx.handle, mask), "pk_particles_h2d_columns")
            self.transfers["h2d_columns"] += 1
            self.transfers["columns_up"] += bin(mask).count("1")
        if any(k in ("x", "y", "z") for k in columns):
            self._sorted_t = None  # the host moved particles: the next launch sorts again

    def fill_column(self, name, value):
        """`particles.<name> = value` on the device (float64 columns t / dt / next_dt); the host array becomes stale."""
        self.ctx.check(self.lib.pk_particles_fill_f64(self.ctx.handle, self._column_bit(name), float(value)), "pk_particles_fill_f64")
        if isinstance(self._bound, LazyColumns):
            self._bound.mark_launched([name])

    def t_stats(self):
        """(smallest, largest non-NaN t, number of NaNs) of the device column."""
        lo, hi, nn = C.c_double(), C.c_double(), C.c_int64()
        self.ctx.check(self.lib.pk_particles_t_stats(self.ctx.handle, C.byref(lo), C.byref(hi), C.byref(nn)), "pk_particles_t_stats")
        return lo.value, hi.value, int(nn.value)

    def device_column_names(self, data):
        """The columns of `data` that live on the device and that a launch may write."""
        return [k for k in data.keys() if k != "particle_id" and self._column_bit(k)]

    def mark_launched(self, data=None):
        """After launches: the host mirror of the written columns is stale (nothing is copied; LazyColumns downloads on access)."""
        data = self._bound if data is None else data
        if isinstance(data, LazyColumns) and data._engine is self:
            data.mark_launched(self.device_column_names(data))
            return True
        return False

    def compact_deleted(self, data: dict) -> dict:
        """Kernel.remove_deleted (kernel.py:98-106) without moving the columns: the rows in state Delete are removed from
        the device-resident columns (pk_particles_compact); only the `state` column comes back, to tell the host which rows
        survive.  Returns the new SoA dict: `state` and host-only user Variables are compacted here, the device-bound columns
        are fresh arrays of the surviving length that the next d2h() fills."""
        if isinstance(data, LazyColumns):
            # `state` decides which rows survive: always fetched (a resident set has it marked stale; an eager one -- small sets, no marks --
            # would otherwise read the host copy of before the launch)
            if data._engine is self:
                self.d2h(["state"])
            state = data.raw("state")
            pairs = [(k, data.raw(k)) for k in data.raw_keys()]  # (arrays as they are: the device columns are not downloaded)
        else:
            self.d2h(["state"])
            state = data["state"]
            pairs = list(data.items())
        keep = state != StatusCode.Delete
        n_new = int(np.count_nonzero(keep))
        new = {}
        for name, arr in pairs:
            if name != "state" and (name in _hip.COLUMN_BITS or name in self.device_variables):
                new[name] = np.empty((n_new,) + arr.shape[1:], dtype=arr.dtype)
            else:
                new[name] = np.ascontiguousarray(arr[keep])
        lazy = isinstance(data, LazyColumns)
        if lazy:
            new = LazyColumns(new)
        d = self._particles_desc(new)
        got = C.c_int64(-1)
        self.ctx.check(self.lib.pk_particles_compact(self.ctx.handle, C.byref(d), C.byref(got)), "pk_particles_compact")
        assert got.value == n_new
        self._bound = new
        if lazy:  # the surviving rows live on the device; every device column but `state` is a fresh host array the next access fills
            data._engine = None
            data._stale.clear()
            data._dirty.clear()
            new._engine = self
            new._stale = {k for k in new.raw_keys() if k != "state" and (k in _hip.COLUMN_BITS or k in self.device_variables)}
        return new

    def h2d(self):
        if self._next_dt_f32 is not None:
            self._next_dt_f32[1][:] = self._next_dt_f32[0]
        self.ctx.check(self.lib.pk_particles_h2d(self.ctx.handle), "pk_particles_h2d")
        self.transfers["h2d_full"] += 1
        self._sorted_t = None  # host row order again
        b = self._bound
        if isinstance(b, LazyColumns):  # host and device agree: the set is resident from here on
            b._engine = self
            b._stale.clear()
            b._dirty.clear()

    def d2h(self, columns=None):
        """Copy the particle columns back to the bound NumPy arrays (all, or only the named ones)."""
        if columns is None:
            self.ctx.check(self.lib.pk_particles_d2h(self.ctx.handle), "pk_particles_d2h")
            self.transfers["d2h_full"] += 1
        else:
            mask = 0
            for name in columns:
                mask |= self._column_bit(name)
            self.ctx.check(self.lib.pk_particles_d2h_columns(self.ctx.handle, mask), "pk_particles_d2h_columns")
            self.transfers["d2h_columns"] += 1
            self.transfers["columns_down"] += bin(mask).count("1")
        if self._next_dt_f32 is not None and (columns is None or "next_dt" in columns):
            self._next_dt_f32[0][:] = self._next_dt_f32[1]
        b = self._bound
        if isinstance(b, LazyColumns):
            if columns is None:
                b._stale.clear()
            else:
                b._stale -= set(columns)

    # ---- the multi-GPU exchange through the C ABI (csrc/pk_comm.inc: RCCL opened by the library) -------------------
    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """pk_comm_init: this engine's context joins a communicator of `world` ranks (the 128-byte id: comm_unique_id() of one rank)."""
        buf = (C.c_uint8 * _hip.PK_COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self.ctx.check(self.lib.pk_comm_init(self.ctx.handle, int(rank), int(world), buf), "pk_comm_init")
        self.comm = (int(rank), int(world))

    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * _hip.PK_COMM_ID_BYTES)()
        rc = self.lib.pk_comm_unique_id(buf)
        if rc != 0:
            msg = self.lib.pk_last_error(None)
            raise _hip.HipLibraryError(f"pk_comm_unique_id failed: {msg.decode() if msg else rc}")
        return bytes(buf)

    def comm_destroy(self):
        if getattr(self, "comm", None) is not None:
            self.lib.pk_comm_destroy(self.ctx.handle)
            self.comm = None

    def comm_allreduce(self, values, op: str):
        """Element-wise min / max / sum of a few int64 over the ranks (pk_comm_allreduce_i64)."""
        a = np.ascontiguousarray(values, dtype=np.int64).copy()
        code = {"min": _hip.PK_OP_MIN, "max": _hip.PK_OP_MAX, "sum": _hip.PK_OP_SUM}[op]
        self.ctx.check(self.lib.pk_comm_allreduce_i64(self.ctx.handle, _ptr(a), len(a), code), "pk_comm_allreduce_i64")
        self.comm_stats["allreduces"] += 1
        return a

    def comm_allgather(self, values):
        """Every rank's int64 values to every rank: a (world, n) array in rank order (pk_comm_allgather_i64)."""
        a = np.ascontiguousarray(values, dtype=np.int64)
        out = np.zeros((self.comm[1], len(a)), np.int64)
        self.ctx.check(self.lib.pk_comm_allgather_i64(self.ctx.handle, _ptr(a), len(a), _ptr(out)), "pk_comm_allgather_i64")
        self.comm_stats["allreduces"] += 1
        return out

    def gather_rows(self, names, t, apply_filter=True, to_all=False, fetch=True):
        """The write-out exchange (include/parcels_hip.h: pk_gather_rows_to_root / pk_allgather_output): the rows of the device-resident
        columns `names` that pass the reference's write filter at output time `t`, of ALL ranks in rank order -- NumPy arrays on the
        receiving rank(s), None on the others.  (counts per rank in `self.comm_last_counts`.)"""
        mask = 0
        for n in names:
            bit = self._column_bit(n)
            if not bit:
                raise KeyError(f"'{n}' is not a device-resident particle column")
            mask |= bit
        world = self.comm[1]
        counts = np.zeros(world, np.int64)
        fn = self.lib.pk_allgather_output if to_all else self.lib.pk_gather_rows_to_root
        self.ctx.check(fn(self.ctx.handle, float(t), int(bool(apply_filter)), mask, _ptr(counts)), "pk_allgather_output" if to_all else "pk_gather_rows_to_root")
        self.comm_last_counts = counts
        self.comm_stats["gathers"] += 1
        if not fetch or (not to_all and self.comm[0] != 0):
            return None  # (fetch=False: the gathered rows stay in the library's device staging; bench.py times the exchange alone)
        total = int(counts.sum())
        b = self._bound
        raw = b.raw if isinstance(b, LazyColumns) else b.__getitem__  # (dtypes / widths only)
        out = {}
        d = _hip.ParticlesDesc()
        d.n = total
        for n in names:
            src = raw(n)
            out[n] = np.empty((total,) + src.shape[1:], dtype=np.float64 if (n == "next_dt" and src.dtype == np.float32) else src.dtype)
            if n in _hip.COLUMN_BITS:
                setattr(d, n, _ptr(out[n]))
            else:
                k = self.device_variables.index(n)
                d.extra[k] = _ptr(out[n])
        self.ctx.check(self.lib.pk_gathered_fetch(self.ctx.handle, C.byref(d), total), "pk_gathered_fetch")
        if "next_dt" in out and raw("next_dt").dtype == np.float32:
            out["next_dt"] = out["next_dt"].astype(np.float32)
        return out

    # ---- asynchronous write-out snapshots ------------------------------------------------------------------------
    _SNAP_COLS = ("t", "z", "y", "x", "dz", "dy", "dx", "dt", "next_dt", "state", "ei", "particle_id")

    def snapshot_begin(self, columns, slot: int, filter_t=None):
        """Enqueue the copy of the named device columns (host row order) into the pinned host set ``slot``; returns at once.
        filter_t: only the rows that pass ParticleFile's write filter at that output time (pk_particles_snapshot_filtered: selected and
        packed on the device -- the table's rows are all that crosses PCIe)."""
        mask = 0
        for name in columns:
            mask |= self._column_bit(name)
        self._snap_extra = list(self.device_variables)
        if filter_t is None:
            self.ctx.check(self.lib.pk_particles_snapshot_begin(self.ctx.handle, mask, int(slot)), "pk_particles_snapshot_begin")
        else:
            self.ctx.check(self.lib.pk_particles_snapshot_filtered(self.ctx.handle, mask, int(slot), float(filter_t)), "pk_particles_snapshot_filtered")
        b = self._bound
        raw = b.raw if isinstance(b, LazyColumns) else b.__getitem__  # (dtypes only: no download, no dirty mark)
        self._snap_dtypes = {k: raw(k).dtype for k in b.keys() if k in _hip.COLUMN_BITS or k in self.device_variables}

    def snapshot_wait(self, slot: int) -> dict:
        """Block until snapshot ``slot`` has landed; NumPy views of its pinned columns (valid until the slot is reused).  Callable
        from a worker thread while the main thread drives the next launch."""
        d = _hip.ParticlesDesc()
        rc = self.lib.pk_particles_snapshot_wait(self.ctx.handle, int(slot), C.byref(d))
        if rc != 0:
            raise _hip.HipLibraryError(f"pk_particles_snapshot_wait(slot={slot}) failed ({rc})")
        n = int(d.n)
        sp = np.float32 if d.spatial_dtype == _hip.PK_F32 else np.float64
        spec = {"t": np.float64, "z": sp, "y": sp, "x": sp, "dz": sp, "dy": sp, "dx": sp, "dt": np.float64, "next_dt": np.float64,
                "state": np.int32, "ei": np.int32, "particle_id": np.int64}
        out = {}
        for name in self._SNAP_COLS:
            ptr = getattr(d, name)
            if not ptr:
                continue
            width = int(d.ngrids) if name == "ei" else 1
            dt = np.dtype(spec[name])
            buf = (C.c_char * (n * width * dt.itemsize)).from_address(ptr)
            a = np.frombuffer(buf, dtype=dt, count=n * width)
            if name == "ei":
                a = a.reshape(n, width)
            want = self._snap_dtypes.get(name)
            if name == "next_dt" and want is not None and want != dt:  # float32 Variable shadowed by a float64 device column
                a = a.astype(want)
            out[name] = a
        for k, name in enumerate(self._snap_extra):
            ptr = d.extra[k]
            if not ptr:
                continue
            dt = np.dtype(np.float32 if d.extra_dtype[k] == _hip.PK_F32 else np.float64)
            want = self._snap_dtypes.get(name)
            if want is not None and want.itemsize == dt.itemsize:
                dt = want  # (an integer Variable: same bytes)
            out[name] = np.frombuffer((C.c_char * (n * dt.itemsize)).from_address(ptr), dtype=dt, count=n)
        return out

    # ---- execution -------------------------------------------------------------------------------------------
    def make_params(self, kernel_ids, *, endtime, dt0, context=None, seed=0, reset_state=1, have_guess0=0, sort_by_cell=0, samples=None,
                    horizon=None, max_iters=0, twe_keys=()):
        fs = self.fieldset
        context = context or {}
        p = _hip.ExecParams()
        if not (1 <= len(kernel_ids) <= _hip.PK_MAX_KERNELS):
            raise ValueError(f"between 1 and {_hip.PK_MAX_KERNELS} kernels are supported")
        p.nk = len(kernel_ids)
        for i, k in enumerate(kernel_ids):
            p.kernels[i] = int(k)
        uv = fs.fields.get("UV")
        uvw = fs.fields.get("UVW")
        vec = uvw if uvw is not None else uv
        p.interp_uv = int(vec.interp_method.kind) if vec is not None else 0
        fid = self.field_ids
        p.fU = fid.get(vec.U.name, -1) if vec is not None else -1
        p.fV = fid.get(vec.V.name, -1) if vec is not None else -1
        p.fW = fid.get(uvw.W.name, -1) if uvw is not None else -1
        p.fKh_zonal = fid.get("Kh_zonal", -1)
        p.fKh_meridional = fid.get("Kh_meridional", -1)
        p.rk45_mode = int("RK45_tol" in context)
        p.next_dt_f32 = int(getattr(self, "_next_dt_f32", None) is not None)
        p.reset_state = int(reset_state)
        p.have_guess0 = int(have_guess0)
        p.sort_by_cell = int(sort_by_cell)
        p.endtime = float(endtime)
        p.dt0 = float(dt0)
        p.rk45_tol = float(context.get("RK45_tol", 0.0))
        p.rk45_min_dt = float(context.get("RK45_min_dt", 0.0))
        p.rk45_max_dt = float(context.get("RK45_max_dt", 0.0))
        p.dres = float(context.get("dres", 0.0))
        p.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        p.horizon_lo, p.horizon_hi = (-np.inf, np.inf) if horizon is None else (float(horizon[0]), float(horizon[1]))
        p.max_iters = int(max_iters)
        p.twe_n = len(twe_keys)
        if twe_keys:  # ascending, alive as long as the params object
            p._twe_keys = (C.c_int64 * len(twe_keys))(*sorted(int(k) for k in twe_keys))
            p.twe_key = C.cast(p._twe_keys, C.POINTER(C.c_int64))
        for slot in range(_hip.PK_MAX_KERNELS):
            p.sample_field[slot] = p.sample_var[slot] = -1
        for slot, (fname, var) in (samples or {}).items():
            p.sample_field[slot] = {"UV": -2, "UVW": -3}[fname] if fname in ("UV", "UVW") else self.field_ids[fname]  # PK_SAMPLE_UV / _UVW
            p.sample_var[slot] = int(var)
        return p

    @staticmethod
    def _repeat_decision(pass_err: int, pass_twk: int, cap: int):
        """What a pass over a Kernel.execute call found -> "key" (a new sample fails call-wide), "cap" (a lower iteration limit) or None.
        A sample that fails call-wide changes everything from that sample on -- the error iterations and later keys of the pass were
        computed without it -- unless the batch stops before its iteration (kernel.py:236-245)."""
        it_k = pass_twk >> 32
        if pass_twk and (pass_err == 0 or it_k <= pass_err) and (cap == 0 or it_k <= cap):
            return "key"
        if pass_err > 0 and (cap == 0 or pass_err < cap):
            return "cap"
        return None

    TWE_SPECULATIVE_PASSES = 64  # after that many repeats of one call: one key per pass (the round-4 scheme), which provably ends

    @staticmethod
    def _twe_step(keys, cap, pass_err, found, hits, speculative=True):
        """One pass over a Kernel.execute call is over; `keys` (ascending) were listed as failing call-wide, `cap` was the iteration limit.
        The pass reports: pass_err (first erring iteration, 0 = none), found (ascending: unlisted samples at which a particle left a time
        interval) and hits (per listed key: some particle really was outside there).  -> (decision, keys, cap) with decision "key" (the
        list changed: repeat without iteration limit), "cap" (a lower iteration limit: repeat) or None (the call stands).

        Everything before the first PROBLEM -- the smallest new or unjustified key -- is exact: the trajectories up to there were
        computed with exactly the samples failing that the reference fails (induction over the key order).  The problem itself is fixed (a
        new key is listed: it was found on valid trajectories; an unjustified one is dropped: nobody leaves the interval there).  What lies
        BEHIND it was computed on trajectories that change now; `speculative` keeps the later keys of this pass anyway -- a particle past
        the last time level fails every later sample too, whatever the others do, so they usually all stand -- and the next pass validates
        them the same way.  Keys beyond the iteration at which the batch stops (error stop, iteration limit) are never reached and do not
        count.  Ends: the first problem moves up with every pass."""
        limit = min(cap or (1 << 62), pass_err or (1 << 62))
        it = lambda k: int(k) >> 32  # noqa: E731
        new = [k for k in found if it(k) <= limit]
        unjust = [k for k, h in zip(keys, hits) if not h and it(k) <= limit]
        if new or unjust:
            p = min(new + unjust)
            out = [k for k in keys if k < p]
            if p in new:
                out.append(p)
            if len(out) > _hip.PK_MAX_TWE:
                raise RuntimeError(f"more than {_hip.PK_MAX_TWE} call-wide time errors in one Kernel.execute")
            if speculative:
                tail = sorted({k for k, h in zip(keys, hits) if k > p and h} | {k for k in found if k > p})
                out += tail[: _hip.PK_MAX_TWE - len(out)]
            return "key", out, 0
        if pass_err > 0 and (cap == 0 or pass_err < cap):
            return "cap", list(keys), pass_err
        return None, list(keys), cap

    def _twe_report(self, keys, pass_twk):
        """(found, hits) of the launch that just ended: every unlisted failing sample and the justification flags of the listed ones
        (pk_execute_twe_report) and True; a library without the entry point (scripted stand-ins of the CPU suite) or an overflowing device
        set: the smallest key only, every listed key taken as justified, and False -- nothing validates a listing then, so the caller must not
        keep keys speculatively (the one-key-per-pass scheme of round 4)."""
        fn = getattr(self.lib, "pk_execute_twe_report", None)
        if fn is not None:
            cap = 4096
            found = (C.c_int64 * cap)()
            nf = C.c_int32(0)
            hit = (C.c_uint8 * max(len(keys), 1))()
            self.ctx.check(fn(self.ctx.handle, found, cap, C.byref(nf), hit, len(keys)), "pk_execute_twe_report")
            if nf.value >= 0:
                fl = [int(found[k]) for k in range(min(nf.value, cap))]
                if pass_twk and pass_twk not in fl:  # (reported by a dedicated kernel: it names the smallest key only)
                    fl = sorted(fl + [int(pass_twk)])
                return fl, [bool(hit[k]) for k in range(len(keys))], True
        return ([int(pass_twk)] if pass_twk else []), [True] * len(keys), False

    def set_user_program(self, program):
        """Register (or, with None, unregister) the run-time compiled module that carries a kernel list's user kernels (jit.py)."""
        cur = getattr(self, "_user_program", None)
        if program is cur:
            return
        fids = (C.c_int32 * 4)(*(list(program.sample_fids) + [0] * 4)[:4]) if program is not None else (C.c_int32 * 4)()
        self.ctx.check(self.lib.pk_set_user_program(self.ctx.handle, C.c_void_p(program.launcher() if program is not None else None),
                                                    int(program.flags) if program is not None else 0,
                                                    len(program.sample_fids) if program is not None else 0, fids), "pk_set_user_program")
        self._user_program = program

    def execute(self, kernel_ids, *, endtime, dt0, context=None, seed=0, have_guess0=0, sort_by_cell=0, t_start=None, samples=None,
                resort_every=None, in_place_variables=False) -> dict:
        """One Kernel.execute(pset, endtime, dt) on the bound (device-resident) particle columns.

        ``resort_every`` (seconds of model time, with ``sort_by_cell``): the fused launch is cut at a soft horizon every that many
        seconds -- particles pause untouched before a step that would cross it -- and the device rows are re-sorted by cell before the
        next piece, so that the gather locality of a long run does not decay.  Trajectories do not depend on it.

        Two things are properties of the reference's BATCH and need a second look at the whole call (``self.exact_error_stop = False``
        skips both and leaves every particle with its own, per-particle, outcome):

        * when a particle enters an error state the reference raises after THAT iteration of its batch loop (kernel.py:236-245), with
          every other particle stopped there too: the call is repeated from the state before it with that iteration limit;
        * a field sample fails as a whole when ANY particle of the view lies outside the field's time interval
          (index_search.py:85-86) and ``Field.__getitem__`` then writes ErrorOutsideTimeInterval into EVERY particle of the view and
          returns 0 (field.py:31-44, 187-195, 297-304): a pass reports the first sample at which somebody left a time interval
          (``first_time_error_key``), the call is repeated with that sample failing for everybody, reports the next one, ... until
          a pass reports nothing new (include/parcels_hip.h: pk_exec_params.twe_key).

        A single launch is repeated from the second column set (``pk_execute_rerun_keys``: the launch wrote there, the state before it
        is intact); a call of several launches (streamed levels, re-sort horizons, user kernels that update Variables in place) from
        the device checkpoint taken before its first launch.  ``self.agree_min`` (set for a sharded ParticleSet, distributed.py):
        the batch is ALL shards, so the ranks agree on the smallest error iteration / sample key after every pass."""
        sign = 1 if dt0 > 0 else -1
        import time as _time

        # host-side split of a streamed run: commit = synchronous level uploads before a launch, prefetch = staging + enqueueing
        # the next level while the kernel runs, wait = pk_execute_end after the prefetch was enqueued
        total = {"steps": 0, "attempts": 0, "kernel_ms": 0.0, "sort_ms": 0.0, "launches": 0, "commit_s": 0.0, "prefetch_s": 0.0, "wait_s": 0.0,
                 "first_error_iter": 0, "reran": 0, "time_error_keys": [], "pack_ms": 0.0, "packs": 0}
        span0 = None
        if resort_every and sort_by_cell:
            ctx_ = context or {}
            span0 = max(float(resort_every), abs(float(dt0)), abs(float(ctx_.get("RK45_max_dt", 0.0))))
        # A run that may take several launches keeps the state before the first one on the device: an error in a LATER launch stops the
        # whole batch at an iteration some particles passed launches ago (kernel.py:236-245), so the run starts over with that limit.
        checkpointed = False
        # (a re-sort horizon only cuts runs longer than itself: the usual output interval is one launch and needs no checkpoint)
        several = self.windowed or (span0 is not None and not (t_start is not None and np.isfinite(t_start) and abs(float(endtime) - float(t_start)) <= span0))
        # user kernels update their Variables in place (they are not part of the column set a launch writes): repeating a launch from the
        # columns it read would apply them twice, so a repeat of such a list always restarts from the checkpoint
        several = several or in_place_variables
        if several and self.exact_error_stop:
            self.ctx.check(self.lib.pk_particles_checkpoint(self.ctx.handle), "pk_particles_checkpoint")
            checkpointed = True
        # Rows that are still in the cell order of an earlier call (device-resident columns, parcels_amd/columns.py) are not sorted again
        # before `resort_every` of model time has passed: the order of ONE sort holds for weeks of a smooth flow, and a
        # sort costs a third of a 24-step launch.  Locality only -- trajectories do not depend on the row order.
        sorted_t = getattr(self, "_sorted_t", None)
        skip_first_sort = bool(sort_by_cell and sorted_t is not None and t_start is not None and np.isfinite(t_start)
                               and abs(float(t_start) - sorted_t) < (float(resort_every) if resort_every else np.inf))
        cap = 0  # iteration limit of the batch loop (0: none)
        keys: list[int] = []  # samples that fail call-wide with OutsideTimeInterval (pk_exec_params.twe_key)
        rerun = False  # the next pass is a single launch repeated from the second column set
        agree = getattr(self, "agree_min", None)
        while True:  # passes over the whole call
            reset = 1
            t_live = t_start
            last_live = None
            span = span0
            pass_err, pass_twk = 0, 0
            pass_found: set = set()          # unlisted failing samples of this pass (all launches)
            pass_hits = [False] * len(keys)  # listed samples some particle justified in this pass
            validated = True                 # every launch of the pass reported real justification flags
            total["steps"] = total["attempts"] = 0
            try:
                while True:  # launches of one pass
                    st = _hip.ExecStats()
                    prefetched = False
                    if rerun:
                        karr = (C.c_int64 * max(len(keys), 1))(*keys)
                        self.ctx.check(self.lib.pk_execute_rerun_keys(self.ctx.handle, int(cap), len(keys), karr, C.byref(st)), "pk_execute_rerun_keys")
                        rerun = False
                    else:
                        nxt = None
                        if self.windowed:
                            if t_live is None or not np.isfinite(t_live):
                                t_live = 0.0 if sign > 0 else max(float(f.model.time_flt[-1]) for f in self._windowed_fields())
                            _t = _time.perf_counter()
                            nxt = self._commit_window(float(t_live), sign)
                            total["commit_s"] += _time.perf_counter() - _t
                        horizon = None
                        if span is not None and t_live is not None and np.isfinite(t_live):
                            horizon = (-np.inf, float(t_live) + span) if sign > 0 else (float(t_live) - span, np.inf)
                        sort_now = 0 if (reset and skip_first_sort) else sort_by_cell
                        prm = self.make_params(kernel_ids, endtime=endtime, dt0=dt0, context=context, seed=seed, reset_state=reset,
                                               have_guess0=(have_guess0 if reset else 1), sort_by_cell=sort_now, samples=samples, horizon=horizon,
                                               max_iters=cap, twe_keys=keys)

                        if sort_now and t_live is not None and np.isfinite(t_live):
                            self._sorted_t = float(t_live)
                        elif sort_now:
                            self._sorted_t = None
                        self.ctx.check(self.lib.pk_execute_begin(self.ctx.handle, C.byref(prm)), "pk_execute_begin")
                        try:
                            _t = _time.perf_counter()
                            prefetched = self._prefetch(nxt)  # overlaps the kernel that was just launched
                            total["prefetch_s"] += _time.perf_counter() - _t
                        finally:
                            _t = _time.perf_counter()
                            self.ctx.check(self.lib.pk_execute_end(self.ctx.handle, C.byref(st)), "pk_execute_end")
                            total["wait_s"] += _time.perf_counter() - _t
                    reset = 0
                    total["steps"] += st.steps
                    total["attempts"] += st.attempts
                    total["kernel_ms"] += st.kernel_ms
                    total["sort_ms"] += st.sort_ms
                    total["pack_ms"] += getattr(st, "pack_ms", 0.0)
                    total["packs"] += getattr(st, "packs", 0)
                    total["launches"] += st.launches
                    total["sclk_mhz"] = float(getattr(st, "sclk_mhz", 0.0))  # shader clock of the (last) launch's kernel
                    total["program"] = int(st.program)  # which device program the (last) launch ran: include/parcels_hip.h, pk_exec_stats
                    counts = {code: int(st.state_counts[code]) for code in range(_hip.PK_NUM_STATE_CODES) if st.state_counts[code]}
                    if st.first_error_iter > 0:
                        pass_err = int(st.first_error_iter) if pass_err == 0 else min(pass_err, int(st.first_error_iter))
                    if st.first_time_error_key > 0:
                        pass_twk = int(st.first_time_error_key) if pass_twk == 0 else min(pass_twk, int(st.first_time_error_key))
                    if self.exact_error_stop and (st.first_time_error_key > 0 or keys):
                        fl, hl, real = self._twe_report(keys, int(st.first_time_error_key))
                        pass_found.update(fl)
                        pass_hits = [a or b for a, b in zip(pass_hits, hl)]
                        validated = validated and real
                    if st.paused == 0:
                        break
                    if not self.windowed and span is None:
                        raise _hip.HipLibraryError("particles paused although all time levels are resident (internal error)")
                    # (what this launch found already decides that the pass will be repeated: stop it here.  The launches not made would have
                    # reached listed samples too: a pass that was cut short says nothing against a listed key)
                    if self.exact_error_stop and agree is None and (pass_found or self._repeat_decision(pass_err, pass_twk, cap) is not None):
                        pass_hits = [True] * len(keys)
                        break
                    t_live = st.t_min_live if sign > 0 else st.t_max_live
                    # no particle moved AND no new level is on its way (a small ring needs one launch per cycle just to bring in the
                    # level behind the window; the next commit makes it resident): the step itself does not fit
                    if last_live is not None and t_live == last_live and not prefetched:
                        if span is not None:
                            span *= 2.0  # the soft horizon, not the ring, stopped every particle: widen it
                        else:
                            raise RuntimeError(
                                "field window too small: a single step does not fit into the resident time levels; "
                                "increase nslots (FieldSet.to_device(nslots=...))"
                            )
                    last_live = t_live
            except Exception:
                # a collective run: the other ranks are on their way to the agreement of this pass -- meet them there (they raise
                # CollectiveAbort), then let this rank's own exception through
                if agree is not None and self.exact_error_stop:
                    try:
                        agree(0, 0, failed=True)
                    except Exception as e2:  # noqa: BLE001 -- the original exception matters; the swallowed one is logged
                        import logging

                        logging.getLogger("parcels_amd").warning("the failure agreement of a collective run raised as well: %r", e2)
                raise
            if not self.exact_error_stop:
                break
            found = sorted(pass_found)
            if agree is not None:  # the batch of the reference is every shard's particles
                pass_err, pass_twk = agree(pass_err, pass_twk)
                agree_keys = getattr(agree, "keys", None)
                if not pass_twk and not keys:
                    pass  # nobody found a failing sample and none is listed (both known to every rank): the common pass costs ONE all-reduce
                else:
                    got = agree_keys(found if validated else None, pass_hits) if agree_keys is not None else None
                    if got is not None and got[0] is not None:
                        found, pass_hits = got
                    else:  # (an agreement of the round-4 form, or a rank without validation: the smallest key only, one key per pass)
                        validated = False
            if not validated:
                found, pass_hits = ([pass_twk] if pass_twk else []), [True] * len(keys)
            # kernel.py:236-245 / field.py:31-44: what the pass found decides whether the call is repeated -- with the failing samples listed
            # (all that were found, validated by the next pass) or with the iteration limit of the first error
            decision, keys, cap = self._twe_step(keys, cap, pass_err, found, pass_hits,
                                                 speculative=validated and total["reran"] < self.TWE_SPECULATIVE_PASSES)
            repeat = decision is not None
            if not repeat:
                break
            total["reran"] += 1
            if checkpointed:
                self.ctx.check(self.lib.pk_particles_restore(self.ctx.handle), "pk_particles_restore")
            elif several:
                raise _hip.HipLibraryError("a call of several launches has to be repeated but has no checkpoint (internal error)")
            else:
                rerun = True  # the state before the (only) launch still sits in the second column set
        total["first_error_iter"] = cap
        total["time_error_keys"] = list(keys)
        total["state_counts"] = counts
        if self.agree_codes is not None and self.exact_error_stop:
            total["codes_any_shard"] = self._agree_on_codes(counts)
        self.last_stats = total
        return total

    _RAISING_CODES = (70, 60, 61, 51, 52, 50)  # statuscodes.ErrorsToThrow, in its order (kernel.py:31-38)

    def _agree_on_codes(self, counts):
        present = self.agree_codes([1 if counts.get(code) else 0 for code in self._RAISING_CODES])
        return [code for code, p in zip(self._RAISING_CODES, present) if p]

    def execute_idle(self) -> dict:
        """The part an EMPTY shard takes in a collective Kernel.execute: the same sequence of agreements as `execute` (one per pass over
        the call, one on the final error codes), with nothing of its own to report."""
        total = {"steps": 0, "attempts": 0, "kernel_ms": 0.0, "sort_ms": 0.0, "launches": 0, "first_error_iter": 0, "reran": 0,
                 "time_error_keys": [], "state_counts": {}}
        if not self.exact_error_stop or self.agree_min is None:
            return total
        cap, keys = 0, []
        while True:
            pass_err, pass_twk = self.agree_min(0, 0)
            agree_keys = getattr(self.agree_min, "keys", None)
            if not pass_twk and not keys:  # (the same shortcut, from the same agreed values, as `execute`)
                validated, found, hits = True, [], []
            else:
                got = agree_keys([], [False] * len(keys)) if agree_keys is not None else None
                validated = got is not None and got[0] is not None
                found, hits = got if validated else (([pass_twk] if pass_twk else []), [True] * len(keys))
            # (the same decision, from the same agreed values, as the shards that hold particles -- also its RuntimeError at PK_MAX_TWE)
            decision, keys, cap = self._twe_step(keys, cap, pass_err, found, hits, speculative=validated and total["reran"] < self.TWE_SPECULATIVE_PASSES)
            if decision is None:
                break
            total["reran"] += 1
        total["first_error_iter"], total["time_error_keys"] = cap, keys
        if self.agree_codes is not None:
            total["codes_any_shard"] = self._agree_on_codes({})
        return total

    # ---- sampling (Field.eval / VectorField.eval) ---------------------------------------------------------------
    def sample(self, name, t, z, y, x):
        fs = self.fieldset
        f = fs.fields[name]
        t, z, y, x = np.broadcast_arrays(*(np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (t, z, y, x)))
        t, z, y, x = (np.ascontiguousarray(v) for v in (t, z, y, x))
        m = x.shape[0]
        if self.windowed and m > 0:
            return self._sample_streamed(name, t, z, y, x)
        u, v, w = np.zeros(m), np.zeros(m), np.zeros(m)
        st = np.zeros(m, np.int32)
        prm = self.make_params([4], endtime=0.0, dt0=1.0)
        if isinstance(f, VectorField):
            prm.interp_uv = int(f.interp_method.kind)
            prm.fU, prm.fV = self.field_ids[f.U.name], self.field_ids[f.V.name]
            prm.fW = self.field_ids[f.W.name] if f.W is not None else -1
            what = -2 if f.W is not None else -1
        else:
            what = self.field_ids[name]
            prm.fU = prm.fV = what
            prm.fW = -1
        self.ctx.check(
            self.lib.pk_eval(self.ctx.handle, C.byref(prm), what, m, _ptr(t), _ptr(z), _ptr(y), _ptr(x), _ptr(u), _ptr(v), _ptr(w), _ptr(st)),
            "pk_eval",
        )
        self.last_sample_state = self._finish_sample_state(st)
        return u, v, w

    @staticmethod
    def _finish_sample_state(st):
        """Strip PK_EVAL_MASKED from the state codes of pk_eval and give the reference's warning for masked values (field.py:359-370)."""
        masked = (st & _hip.PK_EVAL_MASKED) != 0
        if masked.any():
            from .field import FieldEvalWarning

            warnings.warn("Some interpolated values are out-of-bounds. These values are set to 0. Treat carefully.", FieldEvalWarning, stacklevel=4)
            st = st & ~np.int32(_hip.PK_EVAL_MASKED)
        return st


def _sample_streamed(self, name, t, z, y, x):
    """Field.eval at explicit points when the field levels stream through a ring: the points are visited in time order, one
    resident window at a time (what WindowedArray does for the reference's batches, _windowed_array.py:56-97)."""
    m = x.shape[0]
    order = np.argsort(t, kind="stable")  # NaN last
    ts = t[order]
    u, v, w = np.zeros(m), np.zeros(m), np.zeros(m)
    st = np.zeros(m, np.int32)
    wf = self._windowed_fields()
    was = self.windowed
    i = 0
    try:
        while i < m:
            t0 = float(ts[i])
            j = m
            if np.isfinite(t0):
                self._commit_window(t0, 1)
                hi = np.inf
                for f in wf:
                    tf = np.asarray(f.model.time_flt, dtype=np.float64)
                    lv = [l for l in self._slots(f.name) if l >= 0]
                    if max(lv) < len(tf) - 1:
                        hi = min(hi, float(tf[max(lv)]))
                j = max(int(np.searchsorted(ts, hi, side="right")), i + 1)
            sel = order[i:j]
            self.windowed = False  # the chunk's levels are resident: evaluate it like a resident field
            cu, cv, cw = self.sample(name, t[sel], z[sel], y[sel], x[sel])
            self.windowed = was
            u[sel], v[sel], w[sel] = cu, cv, cw
            st[sel] = self.last_sample_state
            i = j
    finally:
        self.windowed = was
    self.last_sample_state = st
    return u, v, w


DeviceEngine._sample_streamed = _sample_streamed


def _engine_search(self, igrid, z, y, x):
    z, y, x = (np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (z, y, x))
    ei = np.zeros(x.shape[0], np.int32)
    self.ctx.check(self.lib.pk_search(self.ctx.handle, int(self.grid_ids[igrid]), x.shape[0], _ptr(z), _ptr(y), _ptr(x), _ptr(ei)), "pk_search")
    return ei


DeviceEngine.search = _engine_search


def raise_particle_errors(data: dict, first_code=None):
    """kernel.py:236-245: raise for the first error code present, in ErrorsToThrow order.  ``first_code`` (a sharded ParticleSet): the
    first code of the WHOLE batch -- every rank raises that one, with the particles of its own shard that carry it (possibly none)."""
    from .statuscodes import ErrorsToThrow

    state = data["state"]
    for code, fn in ErrorsToThrow.items():
        inds = state == code
        if first_code is not None and code != int(first_code):
            continue
        if np.any(inds) or first_code is not None:
            if code == StatusCode.ErrorOutsideTimeInterval:
                fn(data["t"][inds])
            else:
                fn(data["z"][inds], data["y"][inds], data["x"][inds])
