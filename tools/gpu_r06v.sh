#!/bin/bash
# round 6: contracted (fused multiply-add) interpolation arithmetic in the dedicated kernels: base (= the library of commit 80954bf) vs new on the
# headline and configs 3 / 5 with the oracle re-runs, the tests of both dedicated paths + parity, 3000 fuzz seeds
out=gpurun_out/${OUT:-r06v}; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_cgrid.py tests/test_gpu_jit_kernels.py tests/test_gpu_parity.py tests/test_gpu_lean_math.py -q -n 4 > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/summary.txt; grep -E "^FAILED" $out/pytest.log | cut -c1-200 | tee -a $out/summary.txt; tail -1 $out/pytest.log | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=600000 PARCELS_FUZZ_SEEDS=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration -n 4 > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt
bash tools/ab_c2_variants.sh $out/ab_c2 "base new" 3 | tee -a $out/summary.txt
python - $out/ab_c2/c2_new_1.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print("check", d.get("check"))
PY
bash tools/ab_c5_variants.sh $out/ab_c5 "base new" 2 "--reps 3 --pairs-leg 0 --check 100000" c5 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c3 "base new" 2 "--reps 3 --check 100000" c3 | tee -a $out/summary.txt
echo finished | tee -a $out/summary.txt
