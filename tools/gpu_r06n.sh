#!/bin/bash
# round 6: (1) one shard of BASELINE config 4 at full size on one GPU -- 1e7 particles on the 4322 x 3059 x 75 grid, AdvectionRK4_3D, a table every
# 6 steps through the C-ABI exchange (world 1) -- with the single-process file compared byte for byte; (2) 12000 more fuzz seeds on the final binary
out=gpurun_out/r06n; mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python tools/bench_configs.py --config c4 --particles 1e7 --verify-single > $out/c4_one_shard.json 2> $out/c4_one_shard.err; echo "c4 rc $?" | tee -a $out/summary.txt
tail -c 1500 $out/c4_one_shard.json | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=100000 PARCELS_FUZZ_SEEDS=12000 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration -n 4 > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
echo finished | tee -a $out/summary.txt
