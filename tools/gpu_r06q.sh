#!/bin/bash
# round 6: sincos_near + lean sqrt / division / reciprocal in the dedicated C-grid kernels: accuracy of the routines on the device, the tests
# that exercise them, 3000 fuzz seeds, configs 3 / 5 with the oracle re-run of 1e5 ids
out=gpurun_out/${OUT:-r06q}; mkdir -p $out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I parcels_amd/csrc tools/lean_math_check.hip -o /tmp/lean_math_check 2> /dev/null && /tmp/lean_math_check | tee -a $out/summary.txt; echo "lean_math_check rc $?" | tee -a $out/summary.txt
timeout 1500 python -m pytest tests/test_gpu_fast_cgrid.py tests/test_gpu_jit_kernels.py tests/test_gpu_parity.py -q -n 4 > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/summary.txt; grep -E "^FAILED" $out/pytest.log | cut -c1-200 | tee -a $out/summary.txt; tail -1 $out/pytest.log | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=300000 PARCELS_FUZZ_SEEDS=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration -n 4 > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c5 "new" 2 "--reps 3 --pairs-leg 0 --check 100000" c5 | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_c3 "new" 2 "--reps 3 --check 100000" c3 | tee -a $out/summary.txt
echo finished | tee -a $out/summary.txt
