#!/bin/bash
# round 6: the branch-free three-candidate lat / lon search (PK_FAST_SEARCH2 = 2) -- parity (fast-path tests, fixtures, 3000 fuzz seeds), then A/B
out=gpurun_out/r06h; mkdir -p $out
export TMPDIR=/tmp
export PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_s3.so
timeout 900 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_parity.py tests/test_gpu_semantics.py tests/test_gpu_bench_size.py -m gpu -q > $out/pytest_s3.log 2>&1; echo "pytest s3 rc $?" | tee -a $out/summary.txt; tail -3 $out/pytest_s3.log | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=60000 PARCELS_FUZZ_SEEDS=3000 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration > $out/fuzz_s3.log 2>&1; echo "fuzz s3 rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz_s3.log | tee -a $out/summary.txt
unset PARCELS_HIP_LIB
bash tools/ab_c2_variants.sh $out/ab_c2 "new s3" 4 | tee -a $out/summary.txt
