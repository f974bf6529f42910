#!/bin/bash
# A/B of the occupancy target / cell-cache placement of the three fast C-grid kernels (pk_kernels.h: PK_MIN_WAVES_CGRID*, PK_CG_CACHE*):
# build the variants with tools/build_variant.sh (PK_VARIANT_TUS="pk_api pk_prog_cgrid_fast"), then on the GPU box:
#   bash tools/ab_cgrid_occupancy.sh OUTDIR "c3 variants" "c5 variants"       (variant "base" = the library of `make`)
out=${1:-gpurun_out/ab}; mkdir -p $out
for cfg in c3 c5; do
  if [ $cfg = c3 ]; then vs=$2; else vs=$3; fi
  for v in $vs; do
    if [ $v = base ]; then unset PARCELS_HIP_LIB; else export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so; fi
    PK_PRINT_OCCUPANCY=1 timeout 600 python tools/bench_configs.py --config $cfg > $out/${cfg}_$v.json 2> $out/${cfg}_$v.err
  done
done
grep -H "workgroups" $out/*.err | sed 's/.*\///' | sort | uniq > $out/occ.txt
