#!/bin/bash
# round 6: GPU fuzz of the round-6 kernels (flag word + fused search in the A-grid kernel; LDS-DMA fetch, table coordinates, no time memo in the
# RK45 kernel): 6000 random configurations HIP vs oracle on the final binary
out=gpurun_out/r06i; mkdir -p $out
PARCELS_FUZZ_SEED0=70000 PARCELS_FUZZ_SEEDS=6000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
