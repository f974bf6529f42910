#!/bin/bash
out=gpurun_out/r05j; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/pytest.log | tee -a $out/summary.txt; grep -E "^FAILED|^ERROR" $out/pytest.log | head -20 | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=52000 PARCELS_FUZZ_SEEDS=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
