#!/bin/bash
# round 6, first GPU call: (1) parity of the LDS-DMA cell fetch (exec-masked global_load_lds: the lanes that stayed in their cell must keep
# their slot) and of the refactored default kernels; (2) C5 / C3 A/B of the cell-cache placements: base (round 5) | new (same placement,
# this tree) | dma | pxg (corner coordinates from the table) | pxgdma; (3) C2 A/B base vs new (flag word instead of saved lane masks) with
# the VALU instruction counters of both
out=gpurun_out/r06a; mkdir -p $out; OUT=$PWD/$out
export TMPDIR=/tmp
for v in new dma pxgdma; do
  if [ $v = new ]; then unset PARCELS_HIP_LIB; else export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so; fi
  timeout 900 python -m pytest tests/test_gpu_fast_cgrid.py tests/test_gpu_fast_path.py -m gpu -q -x > $out/pytest_$v.log 2>&1; echo "pytest $v rc $?" | tee -a $out/summary.txt; tail -3 $out/pytest_$v.log | tee -a $out/summary.txt
done
unset PARCELS_HIP_LIB
bash tools/ab_c5_variants.sh $out/ab_c5 "base new dma pxg pxgdma" 2 "--reps 3 --pairs-leg 0 --check 1e5" c5 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c3 "base new dma" 2 "--reps 3 --check 1e5" c3 | tee -a $out/summary.txt
bash tools/ab_c2_variants.sh $out/ab_c2 "base new" 3 | tee -a $out/summary.txt
B="python $PWD/bench.py --no-cpu-baseline --steps 24 --warmup 2 --secondary 0 --user-kernels 0"
for v in base new; do
  if [ $v = new ]; then unset PARCELS_HIP_LIB; else export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_$v.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_c2_$v -o p --output-format csv -- $B > /dev/null 2> /dev/null)
done
