"""ctypes binding of libparcels_hip.so (the C ABI declared in include/parcels_hip.h).

There is no CPU fallback: if the shared library cannot be loaded this module raises, and every caller in the
package fails loudly with it.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PARCELS_HIP_LIB", os.path.join(_HERE, "libparcels_hip.so"))  # override: A/B builds

PK_ABI_VERSION = 9
PK_F32, PK_F64 = 0, 1
PK_MAX_GRIDS, PK_MAX_FIELDS, PK_MAX_KERNELS, PK_NUM_STATE_CODES = 4, 64, 8, 80
PK_MAX_EXTRA = 8
PK_MAX_TWE = 1024
PK_KERNEL_SAMPLE_FIELD = 10
PK_EVAL_MASKED = 0x10000  # pk_eval: or'ed into out_state where the value was zeroed for an out-of-bounds index
PK_COL_EXTRA0 = 0x1000
PK_COL_T, PK_COL_Z, PK_COL_Y, PK_COL_X, PK_COL_DT, PK_COL_STATE, PK_COL_PARTICLE_ID = 0x001, 0x002, 0x004, 0x008, 0x080, 0x200, 0x800
PK_COMM_ID_BYTES = 128
PK_OP_MIN, PK_OP_MAX, PK_OP_SUM = 0, 1, 2
COLUMN_BITS = {n: 1 << i for i, n in enumerate(
    ["t", "z", "y", "x", "dz", "dy", "dx", "dt", "next_dt", "state", "ei", "particle_id"])}


class HipLibraryError(RuntimeError):
    """libparcels_hip.so is missing, failed to load, or returned a library error."""


class DeviceInfo(C.Structure):
    _fields_ = [
        ("name", C.c_char * 128),
        ("arch", C.c_char * 64),
        ("compute_units", C.c_int32),
        ("wavefront_size", C.c_int32),
        ("lds_bytes_per_block", C.c_int32),
        ("clock_khz", C.c_int32),
        ("total_mem", C.c_int64),
        ("free_mem", C.c_int64),
    ]


class GridDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("spherical", C.c_int32),
        ("has_x", C.c_int32),
        ("has_y", C.c_int32),
        ("has_z", C.c_int32),
        ("nx", C.c_int32),
        ("ny", C.c_int32),
        ("nz", C.c_int32),
        ("xdim", C.c_int32),
        ("ydim", C.c_int32),
        ("zdim", C.c_int32),
        ("off_x", C.c_int32),
        ("off_y", C.c_int32),
        ("off_z", C.c_int32),
        ("lon_f32", C.c_int32),
        ("lat_f32", C.c_int32),
        ("depth_f32", C.c_int32),
        ("reserved0", C.c_int32),
        ("deg2m", C.c_double),
        ("lon", C.c_void_p),
        ("lat", C.c_void_p),
        ("depth", C.c_void_p),
        ("node_xyz", C.c_void_p),
        ("h_keys", C.c_void_p),
        ("h_starts", C.c_void_p),
        ("h_counts", C.c_void_p),
        ("h_faces", C.c_void_p),
        ("h_nkeys", C.c_int64),
        ("h_nentries", C.c_int64),
        ("h_bitwidth", C.c_int32),
        ("neighbour_probe", C.c_int32),
        ("h_bbox", C.c_double * 6),
    ]


class FieldDesc(C.Structure):
    _fields_ = [
        ("grid", C.c_int32),
        ("dtype", C.c_int32),
        ("nt", C.c_int32),
        ("nz", C.c_int32),
        ("ny", C.c_int32),
        ("nx", C.c_int32),
        ("has_t", C.c_int32),
        ("has_z", C.c_int32),
        ("has_y", C.c_int32),
        ("has_x", C.c_int32),
        ("has_time_interval", C.c_int32),
        ("is_const", C.c_int32),
        ("nslots", C.c_int32),
        ("pack_count", C.c_int32),
        ("pack_leader", C.c_int32),
        ("reserved0", C.c_int32),
        ("time", C.c_void_p),
    ]


class ParticlesDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int64),
        ("ngrids", C.c_int32),
        ("spatial_dtype", C.c_int32),
        ("t", C.c_void_p),
        ("z", C.c_void_p),
        ("y", C.c_void_p),
        ("x", C.c_void_p),
        ("dz", C.c_void_p),
        ("dy", C.c_void_p),
        ("dx", C.c_void_p),
        ("dt", C.c_void_p),
        ("next_dt", C.c_void_p),
        ("state", C.c_void_p),
        ("ei", C.c_void_p),
        ("particle_id", C.c_void_p),
        ("n_extra", C.c_int32),
        ("extra_dtype", C.c_int32 * PK_MAX_EXTRA),
        ("reserved1", C.c_int32),
        ("extra", C.c_void_p * PK_MAX_EXTRA),
    ]


class ExecParams(C.Structure):
    _fields_ = [
        ("nk", C.c_int32),
        ("kernels", C.c_int32 * PK_MAX_KERNELS),
        ("interp_uv", C.c_int32),
        ("rk45_mode", C.c_int32),
        ("reset_state", C.c_int32),
        ("have_guess0", C.c_int32),
        ("fU", C.c_int32),
        ("fV", C.c_int32),
        ("fW", C.c_int32),
        ("fKh_zonal", C.c_int32),
        ("fKh_meridional", C.c_int32),
        ("sort_by_cell", C.c_int32),
        ("force_lent", C.c_int32),
        ("force_lenz", C.c_int32),
        ("sample_field", C.c_int32 * PK_MAX_KERNELS),
        ("sample_var", C.c_int32 * PK_MAX_KERNELS),
        ("next_dt_f32", C.c_int32),
        ("endtime", C.c_double),
        ("dt0", C.c_double),
        ("rk45_tol", C.c_double),
        ("rk45_min_dt", C.c_double),
        ("rk45_max_dt", C.c_double),
        ("dres", C.c_double),
        ("seed", C.c_uint64),
        ("horizon_lo", C.c_double),
        ("horizon_hi", C.c_double),
        ("max_iters", C.c_int32),
        ("body_only", C.c_int32),
        ("twe_n", C.c_int32),
        ("reserved1", C.c_int32),
        ("twe_key", C.POINTER(C.c_int64)),
    ]


class ExecStats(C.Structure):
    _fields_ = [
        ("steps", C.c_int64),
        ("attempts", C.c_int64),
        ("paused", C.c_int64),
        ("state_counts", C.c_int64 * PK_NUM_STATE_CODES),
        ("t_min_live", C.c_double),
        ("t_max_live", C.c_double),
        ("kernel_ms", C.c_double),
        ("sort_ms", C.c_double),
        ("launches", C.c_int32),
        ("program", C.c_int32),
        ("first_error_iter", C.c_int64),
        ("first_time_error_key", C.c_int64),
        ("pack_ms", C.c_double),
        ("packs", C.c_int32),
        ("pad0", C.c_int32),
        ("sclk_mhz", C.c_double),
    ]


# every symbol include/parcels_hip.h declares (tests/test_abi.py checks the library exports all of them)
class HashInfo(C.Structure):
    _fields_ = [("nkeys", C.c_int64), ("nentries", C.c_int64), ("bitwidth", C.c_int32), ("neighbour_probe", C.c_int32), ("bbox", C.c_double * 6)]


ABI_SYMBOLS = [
    "pk_abi_version",
    "pk_init",
    "pk_destroy",
    "pk_last_error",
    "pk_get_device_info",
    "pk_grid_create",
    "pk_grid_hash_info",
    "pk_grid_hash_download",
    "pk_field_create",
    "pk_field_upload_level",
    "pk_field_upload_group_level",
    "pk_field_sync",
    "pk_field_slots",
    "pk_field_evict_outside",
    "pk_particles_bind",
    "pk_particles_h2d",
    "pk_particles_d2h",
    "pk_particles_d2h_columns",
    "pk_particles_h2d_columns",
    "pk_particles_fill_f64",
    "pk_particles_t_stats",
    "pk_particles_set_mask",
    "pk_particles_checkpoint",
    "pk_particles_restore",
    "pk_generic_variant",
    "pk_set_user_program",
    "pk_particles_snapshot_begin",
    "pk_particles_snapshot_wait",
    "pk_particles_device",
    "pk_particles_compact",
    "pk_execute",
    "pk_execute_begin",
    "pk_execute_end",
    "pk_execute_rerun",
    "pk_execute_rerun_keys",
    "pk_execute_twe_report",
    "pk_eval",
    "pk_search",
    "pk_measure_copy_bandwidth",
    "pk_set_option",
    "pk_upload_stats",
    "pk_host_stage_selftest",
    "pk_particles_snapshot_filtered",
    "pk_comm_unique_id",
    "pk_comm_init",
    "pk_comm_destroy",
    "pk_comm_info",
    "pk_comm_allreduce_i64",
    "pk_comm_allgather_i64",
    "pk_gather_rows_to_root",
    "pk_allgather_output",
    "pk_gathered_fetch",
]

_lib = None


def build_library(force: bool = False, jobs: int = 8) -> str:
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s", f"-j{jobs}"])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C parcels_amd/csrc`). parcels_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing ROCm runtime etc.
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    lib.pk_abi_version.restype = C.c_int32
    lib.pk_last_error.restype = C.c_char_p
    lib.pk_last_error.argtypes = [C.c_void_p]
    lib.pk_init.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    lib.pk_destroy.argtypes = [C.c_void_p]
    lib.pk_get_device_info.argtypes = [C.c_void_p, C.POINTER(DeviceInfo)]
    lib.pk_grid_create.argtypes = [C.c_void_p, C.POINTER(GridDesc), C.POINTER(C.c_int32)]
    lib.pk_grid_hash_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(HashInfo)]
    lib.pk_grid_hash_download.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 4
    lib.pk_field_create.argtypes = [C.c_void_p, C.POINTER(FieldDesc), C.POINTER(C.c_int32)]
    lib.pk_field_upload_level.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    lib.pk_field_upload_group_level.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_int32]
    lib.pk_field_sync.argtypes = [C.c_void_p]
    lib.pk_field_evict_outside.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.pk_field_slots.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pk_particles_bind.argtypes = [C.c_void_p, C.POINTER(ParticlesDesc)]
    lib.pk_particles_h2d.argtypes = [C.c_void_p]
    lib.pk_particles_d2h.argtypes = [C.c_void_p]
    lib.pk_particles_d2h_columns.argtypes = [C.c_void_p, C.c_uint32]
    lib.pk_particles_h2d_columns.argtypes = [C.c_void_p, C.c_uint32]
    lib.pk_particles_fill_f64.argtypes = [C.c_void_p, C.c_uint32, C.c_double]
    lib.pk_particles_t_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.pk_particles_set_mask.argtypes = [C.c_void_p, C.c_void_p]
    lib.pk_particles_checkpoint.argtypes = [C.c_void_p]
    lib.pk_generic_variant.argtypes = [C.c_void_p, C.POINTER(ExecParams), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pk_set_user_program.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.pk_particles_restore.argtypes = [C.c_void_p]
    lib.pk_particles_snapshot_begin.argtypes = [C.c_void_p, C.c_uint32, C.c_int32]
    lib.pk_particles_snapshot_wait.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ParticlesDesc)]
    lib.pk_particles_device.argtypes = [C.c_void_p, C.POINTER(ParticlesDesc), C.POINTER(C.c_void_p)]
    lib.pk_particles_compact.argtypes = [C.c_void_p, C.POINTER(ParticlesDesc), C.POINTER(C.c_int64)]
    lib.pk_execute.argtypes = [C.c_void_p, C.POINTER(ExecParams), C.POINTER(ExecStats)]
    lib.pk_execute_begin.argtypes = [C.c_void_p, C.POINTER(ExecParams)]
    lib.pk_execute_end.argtypes = [C.c_void_p, C.POINTER(ExecStats)]
    lib.pk_execute_rerun.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ExecStats)]
    lib.pk_execute_rerun_keys.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(ExecStats)]
    lib.pk_execute_twe_report.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.c_int32]
    lib.pk_eval.argtypes = [C.c_void_p, C.POINTER(ExecParams), C.c_int32, C.c_int64] + [C.c_void_p] * 8
    lib.pk_search.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pk_measure_copy_bandwidth.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    lib.pk_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
    lib.pk_upload_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    if os.environ.get("PARCELS_HIP_ALLOW_ABI") and not hasattr(lib, "pk_comm_init"):
        # A/B measurements against a library of an OLDER round (tools/ab_*.sh: PARCELS_HIP_LIB + PARCELS_HIP_ALLOW_ABI=<its version>): the
        # entry points it lacks are simply absent; nothing else is tolerated
        if lib.pk_abi_version() != int(os.environ["PARCELS_HIP_ALLOW_ABI"]):
            raise HipLibraryError(f"ABI version mismatch: library {lib.pk_abi_version()}, allowed {os.environ['PARCELS_HIP_ALLOW_ABI']}")
        _lib = lib
        return lib
    lib.pk_particles_snapshot_filtered.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_double]
    lib.pk_comm_unique_id.argtypes = [C.c_void_p]
    lib.pk_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.pk_comm_destroy.argtypes = [C.c_void_p]
    lib.pk_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pk_comm_allreduce_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.pk_comm_allgather_i64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.pk_gather_rows_to_root.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_uint32, C.c_void_p]
    lib.pk_allgather_output.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_uint32, C.c_void_p]
    lib.pk_gathered_fetch.argtypes = [C.c_void_p, C.POINTER(ParticlesDesc), C.c_int64]
    if lib.pk_abi_version() != PK_ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {lib.pk_abi_version()}, binding {PK_ABI_VERSION}")
    _lib = lib
    return lib


class Context:
    """One pk_ctx (one device)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        self.handle = C.c_void_p()
        rc = self.lib.pk_init(int(device), C.byref(self.handle))
        if rc != 0:
            msg = self.lib.pk_last_error(self.handle if self.handle else None)
            raise HipLibraryError(f"pk_init(device={device}) failed: {msg.decode() if msg else rc}")
        self.device = int(device)

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.pk_last_error(self.handle)
            raise HipLibraryError(f"{what} failed: {msg.decode() if msg else rc}")

    def device_info(self) -> dict:
        info = DeviceInfo()
        self.check(self.lib.pk_get_device_info(self.handle, C.byref(info)), "pk_get_device_info")
        return {
            "name": info.name.decode(),
            "arch": info.arch.decode(),
            "compute_units": info.compute_units,
            "wavefront_size": info.wavefront_size,
            "lds_bytes_per_block": info.lds_bytes_per_block,
            "clock_khz": info.clock_khz,
            "total_mem": info.total_mem,
            "free_mem": info.free_mem,
        }

    def set_option(self, name: str, value: int):
        """Tuning / A-B switches (include/parcels_hip.h: pk_set_option), e.g. set_option("fast_path", 0)."""
        self.check(self.lib.pk_set_option(self.handle, name.encode(), int(value)), f"pk_set_option({name})")

    def upload_stats(self) -> dict:
        """Host-side accounting of the level stream since pk_init (include/parcels_hip.h: pk_upload_stats)."""
        out = (C.c_double * 4)()
        self.check(self.lib.pk_upload_stats(self.handle, out), "pk_upload_stats")
        return {"stage_fill_s": out[0], "stage_wait_s": out[1], "stage_bytes": out[2], "stage_threads": int(out[3])}

    def copy_bandwidth(self, nbytes: int = 1 << 30, iters: int = 10) -> float:
        g = C.c_double()
        self.check(self.lib.pk_measure_copy_bandwidth(self.handle, int(nbytes), int(iters), C.byref(g)), "pk_measure_copy_bandwidth")
        return g.value

    def close(self):
        if getattr(self, "handle", None) and self.handle:
            self.lib.pk_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
