#!/bin/bash
# round 6: sines / cosines near the particle's own position (PK_CG_NEAR) in the dedicated C-grid kernels: base (HEAD's kernels) vs new on configs 5
# and 3 with the oracle re-run of 1e5 ids at 1e-12, and what the fast-vs-general tests say
out=gpurun_out/${OUT:-r06o}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fast_cgrid.py -q > $out/pytest_fast_cgrid.log 2>&1; echo "pytest fast_cgrid rc $?" | tee -a $out/summary.txt; grep -E "^FAILED|passed|failed" $out/pytest_fast_cgrid.log | cut -c1-200 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c5 "base new e2" 2 "--reps 3 --pairs-leg 0 --check 100000" c5 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c3 "base new e2" 2 "--reps 3 --check 100000" c3 | tee -a $out/summary.txt
grep -h "check" $out/ab_c5/*.err $out/ab_c3/*.err | head -20 | tee -a $out/summary.txt
echo finished | tee -a $out/summary.txt
