#!/bin/bash
# round 6: corner coordinates requested before the search (PK_CG_PXY_EARLY) in the RK45 / M1 kernels: A/B on config 5
out=gpurun_out/r06t; mkdir -p $out
export TMPDIR=/tmp
bash tools/ab_c5_variants.sh $out/ab_c5 "new early" 2 "--reps 3 --pairs-leg 0" c5 | tee -a $out/summary.txt
echo finished | tee -a $out/summary.txt
