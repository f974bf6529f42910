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

}  // namespace pk
