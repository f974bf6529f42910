#!/bin/bash
# round 6: 20000 more fuzz seeds (HIP vs oracle) on the final binary
out=gpurun_out/r06y; mkdir -p $out
PARCELS_FUZZ_SEED0=800000 PARCELS_FUZZ_SEEDS=20000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration -n 4 > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
