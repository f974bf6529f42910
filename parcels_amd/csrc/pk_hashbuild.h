// Device build of the Morton spatial-hash table of a curvilinear grid.
// Reference: SpatialHash.__init__ / _initialize_hash_table, src/parcels/_core/spatialhash.py:45-387.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace pk {

struct HashBuildResult {
    uint32_t* keys = nullptr;   // nkeys unique Morton codes, ascending
    int64_t* starts = nullptr;  // nkeys
    int64_t* counts = nullptr;  // nkeys
    uint32_t* faces = nullptr;  // nentries flat face ids ordered by (code, face)
    int64_t nkeys = 0, nentries = 0;
    int32_t bitwidth = 0;
    double bbox[6] = {0, 0, 0, 0, 0, 0};
};

// node_tab: the grid's device node table {lon, lat, X, Y, Z} per node (pk_device.h: DGrid::node_tab).
// The four output arrays are hipMalloc'ed; ownership passes to the caller.
hipError_t build_spatial_hash(hipStream_t stream, const double* node_tab, int ny, int nx, int spherical, HashBuildResult* out,
                              std::string* err);

// Directory over the sorted key array: dir[b] = index of the first key whose top `bits` bits (of the 30-bit Morton code)
// are >= b, for b in [0, 2^bits].  A query then binary-searches keys[dir[b], dir[b+1]) -- a handful of keys -- instead of
// all of them (23 dependent loads for the 5e6 keys of a 1/12-degree grid).  Same result as the full search.
hipError_t build_hash_directory(hipStream_t stream, const uint32_t* keys, int64_t nkeys, int32_t** dir, int32_t* shift, std::string* err);

// Does the mesh have coincident nodes (cyclic halo columns, a north-fold row, a degenerate pole row)?  Then two distinct
// cells can hold the same point and only the reference's table order decides between them.  Nodes count as coincident when
// they fall into the same cell of a 2^-20-spaced lattice of the unit sphere (~6 m on Earth; 2^-30 of the bbox on a flat
// mesh) in either of two lattices offset by half a cell.  NaN (masked) nodes never coincide.
hipError_t mesh_has_coincident_nodes(hipStream_t stream, const double* node_tab, int ny, int nx, int spherical, const double bbox[6],
                                     bool* coincident, std::string* err);

}  // namespace pk
