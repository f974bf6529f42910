#!/bin/bash
# round 6: PK_FAST_LEAN in the dedicated A-grid kernel (barycentric coordinate by the tabulated reciprocal, conversions by reciprocals): the tests of
# both dedicated paths, the JIT tests (riding kernels keep the exact variants), parity, 3000 fuzz seeds, headline A/B against the previous kernels
out=gpurun_out/${OUT:-r06r}; mkdir -p $out
export TMPDIR=/tmp
timeout 1700 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_fast_cgrid.py tests/test_gpu_jit_kernels.py tests/test_gpu_parity.py tests/test_gpu_semantics.py -q -n 4 > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/summary.txt; grep -E "^FAILED" $out/pytest.log | cut -c1-200 | tee -a $out/summary.txt; tail -1 $out/pytest.log | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=400000 PARCELS_FUZZ_SEEDS=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration -n 4 > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt
bash tools/ab_c2_variants.sh $out/ab_c2 "base new" 3 | tee -a $out/summary.txt
python - $out/ab_c2/c2_new_1.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print("check", d.get("check"))
PY
echo finished | tee -a $out/summary.txt
