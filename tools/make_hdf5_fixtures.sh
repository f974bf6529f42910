#!/bin/bash
# TEST INFRASTRUCTURE (build container only): regenerate tests/golden/hdf5/*.{h5,nc} with the libhdf5 of /opt/conda.
set -e
cd "$(dirname "$0")/.."
gcc -O1 -I/opt/conda/include tools/make_hdf5_fixtures.c -L/opt/conda/lib -Wl,-rpath,/opt/conda/lib -lhdf5 -o /tmp/make_hdf5_fixtures
/tmp/make_hdf5_fixtures tests/golden/hdf5
# the reference's own NetCDF-4 sample, re-packed by the HDF5 tools into chunked / compressed variants of the same values
SRC=/root/reference/tests/test_data/test_interpolation_data_random_linear.nc
if [ -f "$SRC" ]; then
  /opt/conda/bin/h5repack -l U,V,W:CHUNK=1x5x10x10 -f GZIP=5 "$SRC" tests/golden/hdf5/reference_linear_chunked_gzip.nc
fi
ls -la tests/golden/hdf5
