/*
 * parcels_hip.h -- C ABI of libparcels_hip.so, the MI355X (gfx950) engine behind the Parcels
 * ParticleSet.execute() hot path.
 *
 * The reference (Parcels v4-alpha, pure Python/NumPy) has no FFI for this path; its plug-in points are
 * Python protocols.  Each entry point below names the reference interface it stands in for
 * (paths relative to /root/reference/src/parcels).  INTEGRATION.md shows the ctypes stub a Parcels
 * maintainer would add to route `Kernel.execute` through this library.
 *
 * Conventions
 *   - every function returns int32: 0 = ok, <0 = library error (text via pk_last_error);
 *     no C++ exception crosses the ABI;
 *   - per-particle failures are NOT library errors: they are the reference's StatusCode integers in the
 *     particle `state` column (statuscodes.py:19-34), and pk_exec_stats.state_counts summarises them;
 *   - host buffers are owned by the caller (C-contiguous, dtype as stated) and must outlive the call
 *     (for async uploads: until pk_field_sync); the library owns all device memory and pinned staging;
 *   - one pk_ctx per device and per host thread; not thread-safe.
 */
#ifndef PARCELS_HIP_H
#define PARCELS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PK_ABI_VERSION 9 /* 9: pk_particles_snapshot_filtered (device-side write filter for the asynchronous write-out); the multi-GPU exchange in the ABI (pk_comm_unique_id / _init / _destroy / _info / _allreduce_i64 / _allgather_i64, pk_gather_rows_to_root, pk_allgather_output, pk_gathered_fetch: RCCL, opened by the library); 8: pk_execute_twe_report (all failing samples of a pass at once), pk_particles_h2d_columns / _fill_f64 / _t_stats (device-resident columns across execute calls), pk_exec_stats.pack_ms / packs (the pair-copy packing ahead of a launch, timed) / sclk_mhz, "velocity_pairs" opt-in; 7: the call-wide OutsideTimeInterval of the reference on the device (pk_exec_params.twe_n / twe_key, pk_exec_stats.first_time_error_key, pk_execute_rerun_keys); 6: PK_MAX_FIELDS 64 (descriptors in device memory), PK_MAX_EXTRA 8, pk_particles_checkpoint / _restore, user kernels (PK_KERNEL_USER0 .., pk_generic_variant, pk_set_user_program), "eval_points_f32" option; 5: pk_exec_params.horizon_lo / horizon_hi / max_iters, pk_exec_stats.first_error_iter / program, pk_execute_rerun, "fast_cgrid" option; 4: pk_upload_stats, pk_host_stage_selftest, PK_KERNEL_DO_NOTHING / _MOVE_EAST / _MOVE_NORTH, vector PK_KERNEL_SAMPLE_FIELD, "cell_table" option */
#define PK_MAX_GRIDS 4
#define PK_MAX_FIELDS 64
#define PK_MAX_KERNELS 8
#define PK_MAX_EXTRA 8 /* user Variables that device kernels write (PK_KERNEL_SAMPLE_FIELD, compiled user kernels) */
#define PK_NUM_STATE_CODES 80
#define PK_MAX_TWE 1024 /* samples of one Kernel.execute call that fail call-wide with OutsideTimeInterval (pk_exec_params.twe_key) */

typedef struct pk_ctx pk_ctx;

/* dtype tags */
#define PK_F32 0
#define PK_F64 1

/* StatusCode (statuscodes.py:19-34) */
#define PK_SUCCESS 0
#define PK_ENDOFLOOP 1
#define PK_EVALUATE 10
#define PK_REPEAT 20
#define PK_DELETE 30
#define PK_STOPEXECUTION 40
#define PK_STOPALLEXECUTION 41
#define PK_ERROR 50
#define PK_ERRORINTERPOLATION 51
#define PK_ERRORGRIDSEARCHING 52
#define PK_ERROROUTOFBOUNDS 60
#define PK_ERRORTHROUGHSURFACE 61
#define PK_ERROROUTSIDETIMEINTERVAL 70

/* Built-in kernels, recognised by identity on the Python side like kernel.py:129-134 does.
 * 1-9: kernels/_advection.py:21-155, kernels/_advectiondiffusion.py:21-153.
 * 20-25: native forms of the recovery / no-op kernels the reference's tests interleave with them
 *        (tests/common_kernels.py:12-13, tests/test_advection.py:157-174). */
#define PK_KERNEL_ADVECTION_EE 1
#define PK_KERNEL_ADVECTION_RK2 2
#define PK_KERNEL_ADVECTION_RK2_3D 3
#define PK_KERNEL_ADVECTION_RK4 4
#define PK_KERNEL_ADVECTION_RK4_3D 5
#define PK_KERNEL_ADVECTION_RK45 6
#define PK_KERNEL_ADVECTIONDIFFUSION_M1 7
#define PK_KERNEL_ADVECTIONDIFFUSION_EM 8
#define PK_KERNEL_DIFFUSION_UNIFORM_KH 9
#define PK_KERNEL_SAMPLE_FIELD 10 /* `particles.<var> = fieldset.<F>[particles]`: the user kernel every tutorial writes (kernel.py:206-216 runs
                                   it as Python; tests/test_particleset_execute.py:182-205 SampleU / SampleUV).  Field and target
                                   column per kernel-list slot: pk_exec_params.sample_field / sample_var.  Vector form
                                   `particles.<a>, particles.<b>[, particles.<c>] = fieldset.UV[W][particles]` (VectorField.__getitem__,
                                   field.py:250-304: the converted velocity components): sample_field = PK_SAMPLE_UV / PK_SAMPLE_UVW,
                                   sample_var = a | b << 8 | c << 16 with PK_SAMPLE_DISCARD for a component assigned to `_` */
#define PK_SAMPLE_UV (-2)
#define PK_SAMPLE_UVW (-3)
#define PK_SAMPLE_DISCARD 0xFF
#define PK_KERNEL_DELETE_ON_ERROR 20
#define PK_KERNEL_DELETE_OUT_OF_BOUNDS 21
#define PK_KERNEL_SUBMERGE_THROUGH_SURFACE 22
#define PK_KERNEL_DO_NOTHING 23 /* tests/common_kernels.py:8-9: the kernel of the reference's loop / output tests (time passes, nothing moves) */
#define PK_KERNEL_MOVE_EAST 24  /* tests/common_kernels.py:16-17: particles.dx += 0.1 */
#define PK_KERNEL_MOVE_NORTH 25 /* tests/common_kernels.py:20-21: particles.dy += 0.1 */
/* 40 .. 47: user-written kernels (the reference's plug-in point #1, `def kernel(particles, fieldset)`, kernel.py:67-70) compiled at run
 * time into the kernel-list interpreter: parcels_amd/jit.py translates the Python function into a stage of the fused step loop, builds
 * a module for the program variant pk_generic_variant names and registers its launcher with pk_set_user_program. */
#define PK_KERNEL_USER0 40
#define PK_MAX_USER_KERNELS 8

/* ---- context ------------------------------------------------------------------------------------ */
int32_t pk_abi_version(void);
int32_t pk_init(int32_t device, pk_ctx** out);
int32_t pk_destroy(pk_ctx* ctx);
const char* pk_last_error(const pk_ctx* ctx); /* ctx may be NULL: error of the last failed pk_init */

/* Tuning / A-B switches of the library (the reference has no counterpart; its behaviour is the same for every setting):
 *   "fast_path"        1 (default) AdvectionRK4 / AdvectionRK4_3D with XLinear_Velocity on a rectilinear grid with float64
 *                      coordinates run the dedicated kernels of csrc/pk_fast_agrid.h; 0 = the general program.  (This and
 *                      "fast_cgrid": same discrete results -- state, ei, t, deleted set -- and positions within 1e-12 of the
 *                      coordinate scale; since ABI 9's second build the dedicated kernels no longer give the general program's
 *                      bits: quotients by reciprocals, sines / cosines near known ones)
 *   "fast_cgrid"       1 (default) AdvectionRK4 / AdvectionRK4_3D with CGrid_Velocity on a spherical curvilinear grid with float64
 *                      node coordinates run the dedicated kernels of csrc/pk_fast_cgrid.h (needs "cell_table"; 256 B more per
 *                      cell); 0 = the general program
 *   "velocity_pairs"   0 (default since ABI 8; 1 = on) the 2-D kernels of "fast_cgrid" read cell-packed copies of the staggered velocity, one
 *                      8-value group per cell and pair of adjacent resident time levels (32 B per cell and ring slot more for float32
 *                      fields; made on the device ahead of a launch and timed: pk_exec_stats.pack_ms -- 14 ms per level pair at BASELINE
 *                      config 5 for a kernel that gets 1.6 ms faster, hence off; without the memory for them the kernels read the rings)
 *   "clock_probe"      0 (default) / 1 / microseconds: measure the shader clock beside every advection kernel (pk_exec_stats.sclk_mhz)
 *   "special_programs" 1 (default) single-kernel programs for AdvectionRK45 / AdvectionDiffusionM1; 0 = kernel-list interpreter
 *   "cell_cache"       1 (default) per-lane LDS cache of the curvilinear cell;  "hash_directory" 1 (default) key directory;
 *                      "cell_table" 1 (default) per-cell table of the query-independent part of the point-in-cell test (192 B per
 *                      cell of a curvilinear grid); they take effect for grids created / launches made afterwards
 *   "sort_horizontal"  -1 (default) automatic, 0 depth-major, 1 horizontal-major cell sort of curvilinear grids
 *   "eval_points_f32"  0 (default); 1 = the y / x / z handed to pk_eval are float32 particle columns widened to double: the reference then
 *                      forms np.cos(np.deg2rad(y)) -- and, with float32 coordinate arrays, the barycentric coordinates -- in float32
 *                      (what a fused launch does for the default float32 Particle); set around the pk_eval calls it applies to
 * Environment variables PK_NO_FAST, PK_NO_FAST_CGRID, PK_NO_VELOCITY_PAIRS, PK_NO_SPECIAL, PK_NO_CELL_CACHE, PK_NO_HASH_DIR, PK_NO_CELL_TABLE, PK_SORT_HORIZONTAL give the initial
 * values. */
int32_t pk_set_option(pk_ctx* ctx, const char* name, int32_t value);

/* Host-side accounting of the level stream (pk_field_upload_level / _group_level with async=1), cumulative since pk_init:
 * out4[0] seconds the host threads spent filling pinned staging chunks (memcpy / {U,V,W} interleave), out4[1] seconds blocked on
 * the DMA that still read the chunk about to be refilled (= the PCIe link was the limit), out4[2] bytes handed to the DMA engine,
 * out4[3] the number of staging threads.  The reference's counterpart is the time WindowedArray spends materialising levels
 * (src/parcels/_core/_windowed_array.py:56-97); it keeps no such counter. */
int32_t pk_upload_stats(pk_ctx* ctx, double* out4);

/* Host-only self-test of the staging fills behind pk_field_upload_group_level (float / double, 2 / 3 planes, AVX2 body and scalar
 * tail, one thread and the pool) against the plain loop: 0 = all equal.  Needs no device. */
int32_t pk_host_stage_selftest(void);

typedef struct pk_device_info {
    char name[128];
    char arch[64];
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t lds_bytes_per_block;
    int32_t clock_khz;
    int64_t total_mem;
    int64_t free_mem;
} pk_device_info;
int32_t pk_get_device_info(pk_ctx* ctx, pk_device_info* out);

/* ---- grids: XGrid (xgrid.py:108-356) + its SpatialHash table (spatialhash.py:269-387) ------------ */
typedef struct pk_grid_desc {
    int32_t kind;      /* 0 rectilinear (1-D lon/lat), 1 curvilinear (2-D lon/lat)                  */
    int32_t spherical; /* mesh.py:23-47                                                            */
    int32_t has_x, has_y, has_z; /* XGrid.axes                                                     */
    int32_t nx, ny, nz;          /* node counts (nz = 0 without a vertical axis)                   */
    int32_t xdim, ydim, zdim;    /* ravel dims of `ei` (xgrid.py:21-24,208-231; basegrid.py:83-152) */
    int32_t off_x, off_y, off_z; /* C-grid index offsets from SGRID padding (_xinterpolators.py:99-109) */
    int32_t lon_f32, lat_f32, depth_f32; /* the dataset stores that coordinate as float32 (NumPy keeps
                                            f32-f32 arithmetic in f32; the kernels reproduce it)    */
    int32_t reserved0;
    double deg2m;      /* XGrid.deg2m (xgrid.py:201-206): radius*pi/180, or 1.0 on a flat mesh      */
    const double* lon; /* nx, or ny*nx row-major; exact widening of the dataset's values           */
    const double* lat; /* ny, or ny*nx                                                             */
    const double* depth; /* nz (may be NULL when nz == 0)                                          */
    const double* node_xyz; /* spherical curvilinear grids: unit-sphere coordinates of every node, three (ny, nx)
                               planes X = cos(lon)cos(lat), Y = sin(lon)cos(lat), Z = sin(lat) (index_search.py:439-450);
                               NULL otherwise.  Computed once on the host instead of 8 sin/cos pairs per evaluation. */
    /* CSR Morton hash, curvilinear only (all NULL/0 otherwise).  Pass the four arrays to reuse a table built by the
       caller (the reference's SpatialHash._hash_table, spatialhash.py:269-387); leave h_keys NULL and the library
       builds the same table on the device from lon/lat/node_xyz (bit-identical; pk_grid_hash_download reads it back). */
    const uint32_t* h_keys;
    const int64_t* h_starts;
    const int64_t* h_counts;
    const uint32_t* h_faces;
    int64_t h_nkeys;
    int64_t h_nentries;
    int32_t h_bitwidth;
    int32_t neighbour_probe; /* curvilinear search when a particle has left its guessed cell: 0 = automatic (default): on
                                a mesh WITHOUT coincident nodes (no cyclic halo / fold rows, so cells cannot overlap) the
                                neighbour cell the barycentric coordinates point at is tested before the hash-cell faces
                                -- same answer as the reference's table-order walk (spatialhash.py:389-535), found in one
                                probe instead of ~10; on a mesh WITH coincident nodes the table order is kept, and so it is on
                                a mesh with a cell whose point-in-cell test is numerically unreliable (a parallelogram to
                                rounding that still takes the quadratic branch of index_search.py:122-177 -- flat meshes in
                                metres), because such a cell "contains" points of its neighbours.
                                1 = always probe the neighbour first, -1 = never.                                        */
    double h_bbox[6]; /* xmin,xmax,ymin,ymax,zmin,zmax of the hash grid                             */
} pk_grid_desc;
int32_t pk_grid_create(pk_ctx* ctx, const pk_grid_desc* desc, int32_t* grid_id);

/* The spatial-hash table of a curvilinear grid as resident on the device (SpatialHash._hash_table, _bitwidth and the
   hash-grid bounds, spatialhash.py:60-108,214-228,269-387). */
typedef struct pk_hash_info {
    int64_t nkeys;
    int64_t nentries;
    int32_t bitwidth;
    int32_t neighbour_probe; /* 1: the search probes the neighbour cell first on this grid (pk_grid_desc.neighbour_probe) */
    double bbox[6];
} pk_hash_info;
int32_t pk_grid_hash_info(pk_ctx* ctx, int32_t grid, pk_hash_info* out);
/* keys[nkeys], starts[nkeys], counts[nkeys], faces[nentries]; any pointer may be NULL to skip that array. */
int32_t pk_grid_hash_download(pk_ctx* ctx, int32_t grid, uint32_t* keys, int64_t* starts, int64_t* counts, uint32_t* faces);

/* ---- fields: Field data backend (model.py:67-113, _windowed_array.py:25-113) --------------------- */
typedef struct pk_field_desc {
    int32_t grid;  /* grid id == `igrid` (column of the particle `ei` array)                       */
    int32_t dtype; /* PK_F32 | PK_F64 of the TZYX data                                              */
    int32_t nt, nz, ny, nx; /* TZYX extents; size 1 for axes the field does not have               */
    int32_t has_t, has_z, has_y, has_x; /* the field has a dimension on that axis                   */
    int32_t has_time_interval; /* Field.time_interval is not None (field.py:111-116)                */
    int32_t is_const;   /* scalar interpolator of this field: 0 XLinear, 1 XConstantField (value = data[0,0,0,0]),
                           2 XNearest, 3 CGrid_Tracer, 4 XLinearInvdistLandTracer
                           (_xinterpolators.py:112-166, 335-383, 505-613)                            */
    int32_t nslots;     /* device-resident time levels: >= nt keeps all, else a ring (>= 2)         */
    int32_t pack_count; /* > 1: this field leads a group of pack_count same-shaped fields (U,V,W of a C-grid)
                           stored interleaved, one {U,V,W} struct per cell, so that the staggered corner values
                           of one evaluation share cache lines; 0/1: plain array                            */
    int32_t pack_leader; /* field id of the group leader this field joins, or -1                            */
    int32_t reserved0;
    const double* time; /* nt level times, seconds since time_interval.left (index_search.py:88)    */
} pk_field_desc;
int32_t pk_field_create(pk_ctx* ctx, const pk_field_desc* desc, int32_t* field_id);
/* Copy time level `level` (nz*ny*nx values, host) into ring slot level % nslots on the copy stream.
 *   async == 0: blocking; the level is usable when the call returns.
 *   async != 0: enqueued through a pinned staging ring and returns at once; the level whose slot is being
 *               overwritten is evicted immediately, the new level becomes usable at the next pk_field_sync().
 * pk_execute only reads committed levels and never waits on the copy stream, so an async upload of level k+1
 * overlaps the RK sub-steps running on level k.  Stands in for WindowedArray._ensure (_windowed_array.py:56-72). */
int32_t pk_field_upload_level(pk_ctx* ctx, int32_t field_id, int32_t level, const void* host_data, int32_t async);
/* The same for a whole packed group (pack_count > 1): host_data[k] is the level of the k-th component in creation order (the
 * leader first).  The components are interleaved into {U,V,W} structs by the host threads that fill the pinned staging chunks
 * anyway, so the DMA lands the level in its final layout and no device-side repack runs (per-field uploads of a packed field
 * go through a device staging buffer + an interleave kernel instead).  Slot bookkeeping as above, for every field of the group. */
int32_t pk_field_upload_group_level(pk_ctx* ctx, int32_t leader_field_id, int32_t level, const void* const* host_data, int32_t ncomp,
                                    int32_t async);
int32_t pk_field_sync(pk_ctx* ctx); /* wait for the copy stream and commit every pending level */
/* Drop every committed level of the field's ring outside [lo_level, hi_level] (WindowedArray's eviction behind the clock,
 * _windowed_array.py:64-72, for either time direction or a jump): pk_execute requires the resident levels of a ring to be
 * contiguous, and the particles pause at the edge of the intersection of all rings' windows. */
int32_t pk_field_evict_outside(pk_ctx* ctx, int32_t field_id, int32_t lo_level, int32_t hi_level);
/* which committed level each ring slot currently holds (-1 = empty/pending); `levels` has room for nslots ints */
int32_t pk_field_slots(pk_ctx* ctx, int32_t field_id, int32_t* levels, int32_t* nslots);

/* ---- particles: the SoA dict of ParticleSet (particle.py:182-222) -------------------------------- */
typedef struct pk_particles_desc {
    int64_t n;
    int32_t ngrids;        /* columns of ei                                                        */
    int32_t spatial_dtype; /* PK_F32 (default Particle) | PK_F64 for z,y,x,dz,dy,dx                 */
    double* t;
    void *z, *y, *x, *dz, *dy, *dx;
    double* dt;
    double* next_dt; /* NULL unless the particle class has it (AdvectionRK45)                      */
    int32_t* state;
    int32_t* ei; /* n * ngrids                                                                     */
    int64_t* particle_id;
    /* user Variables that live on the device because a device kernel writes them (Particle.add_variable, particle.py:79-113):
       n_extra columns of dtype PK_F32 / PK_F64; all other user Variables stay host-only */
    int32_t n_extra;
    int32_t extra_dtype[PK_MAX_EXTRA];
    int32_t reserved1;
    void* extra[PK_MAX_EXTRA];
} pk_particles_desc;
int32_t pk_particles_bind(pk_ctx* ctx, const pk_particles_desc* host); /* remember host columns, size device columns */
int32_t pk_particles_h2d(pk_ctx* ctx);
int32_t pk_particles_d2h(pk_ctx* ctx);
/* copy back only the selected columns (bit k = k-th column in the order t,z,y,x,dz,dy,dx,dt,next_dt,state,ei,particle_id):
 * the periodic write-out needs the to_write columns only (particlefile.py:142-180), the rest stays device-resident */
#define PK_COL_T 0x001u
#define PK_COL_Z 0x002u
#define PK_COL_Y 0x004u
#define PK_COL_X 0x008u
#define PK_COL_DZ 0x010u
#define PK_COL_DY 0x020u
#define PK_COL_DX 0x040u
#define PK_COL_DT 0x080u
#define PK_COL_NEXT_DT 0x100u
#define PK_COL_STATE 0x200u
#define PK_COL_EI 0x400u
#define PK_COL_PARTICLE_ID 0x800u
#define PK_COL_EXTRA0 0x1000u /* extra column k: PK_COL_EXTRA0 << k */
int32_t pk_particles_d2h_columns(pk_ctx* ctx, uint32_t column_mask);
/* The particle columns stay on the device from one ParticleSet.execute to the next (particleset.py:355-470 re-reads every column from the
 * host arrays on every call; here only what the host touched in between crosses PCIe -- parcels_amd/columns.py):
 * _h2d_columns uploads the selected host columns into the device rows they belong to (through the row order of the cell sort, which is
 * kept -- unlike pk_particles_h2d, which uploads everything in host order); _fill_f64 sets a float64 column (t, dt, next_dt) to one
 * value on the device (`particles.dt = dt` at the start of execute, particleset.py:381); _t_stats reduces the `t` column: smallest and
 * largest finite value (NaN when there is none) and the number of NaNs (unset release times, particleset.py:523-585). */
int32_t pk_particles_h2d_columns(pk_ctx* ctx, uint32_t column_mask);
int32_t pk_particles_fill_f64(pk_ctx* ctx, uint32_t column_mask, double value);
int32_t pk_particles_t_stats(pk_ctx* ctx, double* t_min, double* t_max, int64_t* n_nan);
/* Selection of a pk_exec_params.body_only launch: mask[i] != 0 <=> host row i takes part (n int32 values in host row order). */
int32_t pk_particles_set_mask(pk_ctx* ctx, const int32_t* mask);
/* Device-side checkpoint of every particle column (and of the row order of the cell sort): _checkpoint keeps a copy, _restore puts
 * it back.  For a Kernel.execute that spans several launches (streamed time levels, re-sort horizons): when a particle errs in a later
 * launch the reference has stopped EVERY particle after that iteration of its batch loop (kernel.py:236-245), which only a run from
 * the state before the first launch with pk_exec_params.max_iters can reproduce (single launches: pk_execute_rerun).  Binding other
 * particles discards the checkpoint; ~88 B of HBM per particle. */
int32_t pk_particles_checkpoint(pk_ctx* ctx);
int32_t pk_particles_restore(pk_ctx* ctx);
/* Asynchronous write-out (ParticleSet.execute's output step, particleset.py:452-459, overlapped with the next interval):
 * _begin snapshots the selected columns -- un-sorted into host row order -- into one of two device staging sets on the compute
 * stream and enqueues their copy into pinned host columns on the copy stream; it returns at once and the next pk_execute may
 * start.  _wait blocks until the copy of that slot has landed and returns the pinned columns (NULL for columns outside the mask;
 * valid until the next _begin on the slot).  _wait only waits on an event and may be called from a second host thread (the
 * Parquet encoder) while the first thread drives the next launch. */
int32_t pk_particles_snapshot_begin(pk_ctx* ctx, uint32_t column_mask, int32_t slot);
int32_t pk_particles_snapshot_wait(pk_ctx* ctx, int32_t slot, pk_particles_desc* out_host_columns);
/* The same snapshot, but only of the rows that pass ParticleFile's write filter `|t_p - t| <= |dt|/2` (particlefile.py:198-221) at output
 * time t -- selected and packed ON THE DEVICE in host row order (csrc/pk_select.inc): only the rows a table holds cross PCIe, and the
 * writer thread has nothing left to filter.  pk_particles_snapshot_wait reports their number in pk_particles_desc.n. */
int32_t pk_particles_snapshot_filtered(pk_ctx* ctx, uint32_t column_mask, int32_t slot, double t);
/* device pointers of the bound columns in CURRENT device order (for RCCL all-gather of the output
 * columns at write-out; see parcels_amd/distributed.py).  perm (int64*, may be NULL when the particles
 * have not been cell-sorted) maps device row -> original row. */
int32_t pk_particles_device(pk_ctx* ctx, pk_particles_desc* dev, int64_t** perm);
/* Kernel.remove_deleted (kernel.py:98-106) on the device-resident columns: rows whose state is Delete are removed (the
 * survivors keep their relative order, in device order and in host order), no column crosses PCIe.  `new_host` describes
 * the caller's host arrays of the surviving length (same schema as the bound ones); later pk_particles_d2h* calls fill
 * THOSE.  The caller learns the survivors from the `state` column (pk_particles_d2h_columns(PK_COL_STATE)) beforehand.
 * *n_new = number of surviving particles (must equal new_host->n). */
int32_t pk_particles_compact(pk_ctx* ctx, const pk_particles_desc* new_host, int64_t* n_new);

/* ---- execution: Kernel.execute (kernel.py:174-247) ----------------------------------------------- */
typedef struct pk_exec_params {
    int32_t nk;
    int32_t kernels[PK_MAX_KERNELS]; /* PK_KERNEL_*, applied in order to every evaluated particle   */
    int32_t interp_uv;  /* 0 XLinear_Velocity (A-grid), 1 CGrid_Velocity, 2 XFreeslip, 3 XPartialslip
                           (_xinterpolators.py:169-190, 193-332, 386-502)                           */
    int32_t rk45_mode;  /* hasattr(fieldset, "RK45_tol") (kernel.py:118,225)                        */
    int32_t reset_state; /* 1: state[:] = Evaluate first (kernel.py:188); 0: continue a paused call */
    int32_t have_guess0; /* a particle had a non-zero xi guess at entry (index_search.py:269)       */
    int32_t fU, fV, fW, fKh_zonal, fKh_meridional; /* field ids, -1 if absent                       */
    int32_t sort_by_cell; /* 1: reorder device rows by cell key before stepping (row order on the
                              host is unaffected)                                                  */
    int32_t force_lent, force_lenz; /* lenT / lenZ of XLinearInvdistLandTracer and the slip interpolators, which the reference takes over
                            the WHOLE batch (`2 if np.any(tau > 0) else 1`, _xinterpolators.py:130-131,401-402,575-576) and which
                            change values there: 0 = per particle (pk_execute: a fused multi-step launch has no batch), 1 / 2 =
                            that value for every particle.  pk_eval (one call == one batch) fills them in itself.            */
    int32_t sample_field[PK_MAX_KERNELS]; /* PK_KERNEL_SAMPLE_FIELD in kernel-list slot k: id of the scalar field to sample ...   */
    int32_t sample_var[PK_MAX_KERNELS];   /* ... and the extra particle column (0 .. n_extra-1) that receives the value           */
    int32_t next_dt_f32; /* the particle class declares next_dt as float32 (the default dtype of Variable, particle.py:36-60;
                            tests/utils.py:24-25): AdvectionRK45's store into it rounds to f32, and `dt = next_dt`
                            (kernel.py:118-120) then carries the rounded value.  The bound column itself stays f64.   */
    double endtime; /* seconds; every live particle is advanced from its own t to endtime           */
    double dt0;     /* execute()'s dt (sign gives the time direction)                               */
    double rk45_tol, rk45_min_dt, rk45_max_dt; /* fieldset.context (kernel.py:134-159)              */
    double dres;    /* fieldset.dres (_advectiondiffusion.py:40-58)                                 */
    uint64_t seed;  /* counter-based RNG seed of the stochastic kernels                             */
    double horizon_lo, horizon_hi; /* soft time horizon (seconds; used when lo < hi, so a zeroed struct means none): a particle whose next step [t, t+dt] would leave it
                            pauses untouched (state Evaluate, counted in pk_exec_stats.paused) exactly like at the edge of a level
                            ring -- the host re-sorts by cell and relaunches with reset_state = 0.  No step is clipped or altered.  */
    int32_t max_iters; /* 0 = no limit; otherwise a particle makes at most this many iterations of the loop of kernel.py:190 counted
                            from the start of the Kernel.execute call (reset_state = 1 zeroes the per-particle count) and then stays
                            in Evaluate: how pk_execute_rerun reproduces the reference's stop after the first erroring iteration     */
    int32_t body_only; /* 1: run the kernel list ONCE on every particle selected by pk_particles_set_mask (the `evaluate_particles` of
                            kernel.py:193-195, fixed for one iteration whatever a kernel does to the states) and return: no dt clipping,
                            no position update, no EndofLoop, states neither reset nor interpreted -- the caller owns the loop of
                            kernel.py:190-245.  This is how arbitrary Python kernels run next to device
                            kernels (parcels_amd/hostkernels.py: the loop on the host columns, the built-in kernels' bodies here).   */
    int32_t twe_n;     /* number of entries of twe_key (0: none known)                                                                */
    int32_t reserved1;
    const int64_t* twe_key; /* twe_n <= PK_MAX_TWE keys in ASCENDING order (host memory, read by pk_execute_begin).  The call-wide OutsideTimeInterval (index_search.py:85-86, field.py:31-44,187-195,297-304): in the
                            reference a field sample fails as a WHOLE when any particle of the view it was called with lies outside the
                            field's time interval -- Field.__getitem__ then writes ErrorOutsideTimeInterval into the state of EVERY
                            particle of that view and returns 0 for all of them; no `ei`, no other state of that call is written.  A
                            sample of a Kernel.execute call is named by key = (iteration << 32) | (kernel slot * 1000 + number of the
                            sample within that kernel's call(s) of the iteration) (iteration: 1-based index of the loop of kernel.py:190;
                            the samples of a Repeat re-run of the kernel, kernel.py:211-216, count on).  A launch reports the smallest
                            key at which some particle left a time interval (pk_exec_stats.first_time_error_key, only that particle got
                            the code); the caller runs the call again from the state before it with that key listed here, and every
                            particle that reaches a listed sample takes code 70 and the value 0 there -- and so on until a run reports
                            no new key (parcels_amd/engine.py: DeviceEngine.execute; pk_execute_rerun_keys).  A launch with listed
                            samples runs the general programs (same time keys as the dedicated kernels, which only report).             */
} pk_exec_params;

typedef struct pk_exec_stats {
    int64_t steps;    /* accepted position updates (kernel.py:219-222)                              */
    int64_t attempts; /* kernel evaluations including RK45 repeats                                  */
    int64_t paused;   /* particles that stopped because their next step needs a non-resident level  */
    int64_t state_counts[PK_NUM_STATE_CODES]; /* histogram of `state` after the call                 */
    double t_min_live, t_max_live; /* over particles still in Evaluate (NaN if none)                */
    double kernel_ms; /* HIP-event time of the advection kernel(s) on the compute stream            */
    double sort_ms;   /* HIP-event time of the cell sort (0 when not sorting)                        */
    int32_t launches;
    int32_t program; /* which device program ran (diagnostic; same results whichever): 0-5 the general programs RK4, RK4_3D, kernel-list
                        interpreter, RK45, M1, dtype-emulating interpreter; 100 the dedicated A-grid kernels (csrc/pk_fast_agrid.h), 101 the
                        dedicated curvilinear C-grid kernels (csrc/pk_fast_cgrid.h)                                     */
    int64_t first_error_iter; /* 0 = no particle entered an error state (or StopAllExecution); else the smallest 1-based index of the
                        iteration of the loop of kernel.py:190 in which one did (kernel.py:236-245 raises after THAT iteration)          */
    int64_t first_time_error_key; /* 0 = no sample of this launch left a field's time interval; else the smallest key (see
                        pk_exec_params.twe_key) of a sample, not listed in twe_key, at which a particle did                                */
    double pack_ms;  /* HIP-event time of the cell-packed pair copies of the staggered velocity made ahead of this launch (option
                        "velocity_pairs", off by default; csrc/pk_api.hip: ensure_velocity_pairs) -- NOT part of kernel_ms                  */
    int32_t packs;   /* level pairs packed for it */
    int32_t pad0;
    double sclk_mhz; /* shader clock while the advection kernel of the (last) launch ran (option "clock_probe": sixteen wavefronts on a second
                        stream spin beside it for 1 ms of the 100 MHz counter and count shader-clock cycles; the median); 0 = not measured  */
} pk_exec_stats;
int32_t pk_execute(pk_ctx* ctx, const pk_exec_params* params, pk_exec_stats* stats);
/* kernel.py:236-245: the reference checks the error codes after every iteration of its batch loop, so when it raises, EVERY particle
 * has made exactly as many iterations as the first erroring one.  A fused launch runs every particle to `endtime`; when its
 * pk_exec_stats.first_error_iter is non-zero, this call restores the particle columns to their state before that launch (the launch
 * wrote into the second column set: nothing was copied) and runs it again with max_iters = that index.  Must directly follow the
 * pk_execute / pk_execute_end of the launch (before any pk_particles_* call); stats replace those of the launch. */
/* The call-wide OutsideTimeInterval in few passes (field.py:31-44, index_search.py:85-86).  After pk_execute / pk_execute_end / pk_execute_rerun*:
 * found[0 .. *n_found) = every sample key (ascending; see pk_exec_params.twe_key) NOT in the launch's list at which a particle left a field's time
 * interval, as far as the general programs reported them (the dedicated kernels report only the smallest: pk_exec_stats.first_time_error_key);
 * *n_found = -1 when there were more distinct keys than the device-side set holds.  listed_hit[k] = 1 when at LISTED sample k some particle really
 * was outside the interval -- a listed sample nobody justifies was listed on a trajectory that no longer exists and has to go.  The host lists
 * all found keys at once and validates the listing with the next pass, instead of finding one key per pass (parcels_amd/engine.py). */
int32_t pk_execute_twe_report(pk_ctx* ctx, int64_t* found, int32_t cap, int32_t* n_found, uint8_t* listed_hit, int32_t n_listed);
/* User kernels (PK_KERNEL_USER0 ..), compiled at run time (parcels_amd/jit.py).
 * pk_generic_variant: what a launch with `prm` (the real kernel list, user ids included) runs on this context, given what the user kernels
 * sample (sample_flags: PK_USER_SAMPLES_UV / _UVW; sample_fids: nsample <= 4 scalar field ids) -- key = (float32 fields ? 6 : 0) +
 * (curvilinear main grid ? 3 : 0) + min(interp_uv, 2) and lds (the 1-D coordinate vectors are staged in LDS) name the instantiation of the
 * kernel-list interpreter, typed = NumPy float32 dtype propagation (float32 coordinate arrays: no user kernels), fast = which dedicated
 * kernel would take the list with the user kernels riding along: 1 / 2 the A-grid kernel 2-D / 3-D (csrc/pk_fast_agrid.h: list of the shape
 * [.., AdvectionRK4 / AdvectionRK4_3D, ..] around sampling-free recovery kernels and user kernels; sampled scalar fields laid out exactly
 * like U), 3 / 4 the curvilinear C-grid kernel (csrc/pk_fast_cgrid.h: user kernels that sample nothing), 0 none.
 * pk_set_user_program: the launcher of a module built for exactly that --
 *   void launcher(const void* kargs, int32_t prog, int32_t key, int32_t lds, uint64_t lds_bytes, void* hip_stream)
 * prog 0: the interpreter variant (key, lds); prog 1 / 2 (3 / 4): the dedicated A-grid (C-grid) kernel 2-D / 3-D with key = float32 fields
 * * 2 + float32 particles; it must refuse (abort) what it was not built for.  flags: PK_USER_RIDE = the module carries the dedicated kernel
 * pk_generic_variant named, PK_USER_SAMPLES_* and sample_fids as above.  NULL unregisters.  A list with a user id and no launcher fails. */
#define PK_USER_RIDE 1
#define PK_USER_SAMPLES_UV 2
#define PK_USER_SAMPLES_UVW 4
int32_t pk_generic_variant(pk_ctx* ctx, const pk_exec_params* prm, int32_t sample_flags, int32_t nsample, const int32_t* sample_fids, int32_t* key,
                           int32_t* lds, int32_t* typed, int32_t* fast);
int32_t pk_set_user_program(pk_ctx* ctx, void* launcher, int32_t flags, int32_t nsample, const int32_t* sample_fids);
int32_t pk_execute_rerun(pk_ctx* ctx, int32_t max_iters, pk_exec_stats* stats);
/* The same with the call-wide time errors known so far (pk_exec_params.twe_key): the launch is repeated from the state before it with
 * max_iters (0 = no limit) and the n_keys <= PK_MAX_TWE listed samples failing for every particle that reaches them
 * (field.py:31-44: _deal_with_errors writes the code into the whole view). */
int32_t pk_execute_rerun_keys(pk_ctx* ctx, int32_t max_iters, int32_t n_keys, const int64_t* keys, pk_exec_stats* stats);
/* The same in two halves: _begin enqueues the sort + advection kernel + statistics on the compute stream and returns;
 * the host can then stage and enqueue the NEXT field level (pk_field_upload_level async) while the RK sub-steps run;
 * _end waits and fills the statistics.  pk_execute == begin + end. */
int32_t pk_execute_begin(pk_ctx* ctx, const pk_exec_params* params);
int32_t pk_execute_end(pk_ctx* ctx, pk_exec_stats* stats);

/* ---- sampling: Field.eval / VectorField.eval at explicit points (field.py:145-195, 250-304) ------ */
/* what = field id, or -1 for UV, -2 for UVW (fields taken from params).  All pointers are host
 * arrays of length m; out_v/out_w/out_state may be NULL.  out_state[i] = the StatusCode the sample leaves (PK_EVALUATE = none),
 * with PK_EVAL_MASKED or'ed in where the value was set to 0 because an index was out of bounds (_mask_outofbounds_values,
 * field.py:359-370 -- also for the left / bottom exits in X / Y that set no error code, field.py:327-356). */
#define PK_EVAL_MASKED 0x10000
int32_t pk_eval(pk_ctx* ctx, const pk_exec_params* params, int32_t what, int64_t m, const double* t, const double* z,
                const double* y, const double* x, double* out_u, double* out_v, double* out_w, int32_t* out_state);

/* XGrid.search + ravel_index with no guess: ei_out[i] = ravel(search(z, y, x)) on grid `grid_id`
 * (ParticleSet.populate_indices, particleset.py:252-262).  Host arrays of length m. */
int32_t pk_search(pk_ctx* ctx, int32_t grid_id, int64_t m, const double* z, const double* y, const double* x, int32_t* ei_out);

/* achieved copy bandwidth probe (device-to-device float4 copy), GB/s; used as a measured roofline denominator */
int32_t pk_measure_copy_bandwidth(pk_ctx* ctx, int64_t bytes, int32_t iters, double* gbps);

/* ---- multi-GPU: the exchanges of a run whose particles are sharded by id over one process per GPU ----------------
 * The reference has no multi-process mode; what these stand in for is what a single process does with ALL particles in one place:
 *   - ParticleFile.write (particlefile.py:142-221) filters `|t_p - t| <= |dt|/2` over every particle and appends one table:
 *     pk_gather_rows_to_root applies that filter to this rank's DEVICE rows (in host row order, whatever the cell sort did to the device
 *     order), all-gathers the row counts and sends the surviving rows of the masked columns to rank 0 (grouped ncclSend / ncclRecv over
 *     xGMI; ranks may hold different, also zero, counts).  pk_allgather_output is the all-gather variant `north_star` names: every rank
 *     receives the rows of all ranks, in rank order (= id order for contiguous shards).  pk_gathered_fetch copies the gathered columns to
 *     the caller's arrays (rank 0, or every rank after pk_allgather_output).
 *   - Kernel.execute's batch-wide rules (kernel.py:236-245 stops EVERY particle at the iteration of the first error;
 *     index_search.py:85-86 fails a sample for every particle of the call): the shards agree on them with pk_comm_allreduce_i64
 *     (element-wise MIN / MAX / SUM of a few int64 over the ranks).
 * The communicator belongs to the pk_ctx: one per process.  The 128-byte id comes from pk_comm_unique_id on ONE rank and reaches the
 * others over the host's own channel (a file, MPI, a socket, torch.distributed's store ...).  librccl.so is opened at the first of these
 * calls -- a single-GPU host never loads it.  Collective semantics: every rank of the communicator must make the same sequence of
 * pk_gather_rows_to_root / pk_allgather_output / pk_comm_allreduce_i64 calls. */
#define PK_COMM_ID_BYTES 128
#define PK_OP_MIN 0
#define PK_OP_MAX 1
#define PK_OP_SUM 2
int32_t pk_comm_unique_id(uint8_t* id128);
int32_t pk_comm_init(pk_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id128);
int32_t pk_comm_destroy(pk_ctx* ctx);
int32_t pk_comm_info(pk_ctx* ctx, int32_t* rank, int32_t* world, int32_t* rccl_version);
int32_t pk_comm_allreduce_i64(pk_ctx* ctx, int64_t* values, int32_t n, int32_t op); /* in place */
int32_t pk_comm_allgather_i64(pk_ctx* ctx, const int64_t* send, int32_t n, int64_t* recv); /* recv[world * n], rank order: the failing samples every shard found */
/* t: the output time; apply_filter 0 = every row; mask: PK_COL_* of the columns to exchange; counts[world]: rows of every rank (out) */
int32_t pk_gather_rows_to_root(pk_ctx* ctx, double t, int32_t apply_filter, uint32_t mask, int64_t* counts);
int32_t pk_allgather_output(pk_ctx* ctx, double t, int32_t apply_filter, uint32_t mask, int64_t* counts);
/* out: arrays of >= `capacity` rows for the columns of the last exchange's mask (the other pointers are ignored) */
int32_t pk_gathered_fetch(pk_ctx* ctx, const pk_particles_desc* out, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* PARCELS_HIP_H */
