#!/bin/bash
# A/B builds of the fast kernels: tools/build_variant.sh NAME -DPK_MIN_WAVES_FAST=5 ... -> parcels_amd/libparcels_hip_NAME.so
# (select it at run time with PARCELS_HIP_LIB=...; the other objects are those of the last `make`)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../parcels_amd/csrc"
mkdir -p /tmp/pkv_$NAME
FL="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function"
for tu in ${PK_VARIANT_TUS:-pk_prog_rk4_fast pk_prog_rk4_3d_fast}; do
  /opt/rocm/bin/hipcc $FL "$@" -c $tu.hip -o /tmp/pkv_$NAME/$tu.o &
done
wait
OBJS=""
for o in pk_api pk_host_stage pk_hashbuild pk_prog_rk4 pk_prog_rk4_3d pk_prog_rk45 pk_prog_m1 pk_prog_generic pk_prog_typed pk_prog_rk4_fast pk_prog_rk4_3d_fast pk_prog_cgrid_fast; do
  if [ -f /tmp/pkv_$NAME/$o.o ]; then OBJS="$OBJS /tmp/pkv_$NAME/$o.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../libparcels_hip_$NAME.so $OBJS
echo built ../libparcels_hip_$NAME.so
