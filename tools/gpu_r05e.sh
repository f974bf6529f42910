#!/bin/bash
# round 5, fifth GPU call: RK45 workgroup-miss-queue kernel -- parity first (under a timeout: barriers), then A/B against the product kernel;
# C2 load-batching A/B; GPU fuzz sweep of the round-5 kernels
out=gpurun_out/r05e; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_fast_cgrid.py -m gpu -q -x -k "miss_queue" > $out/pytest_queue.log 2>&1; rc=$?; echo "pytest queue rc $rc" | tee -a $out/summary.txt; tail -15 $out/pytest_queue.log | tee -a $out/summary.txt
if [ $rc -eq 0 ]; then
  for r in 1 2; do
    for v in one queue queue2; do
      unset PARCELS_HIP_LIB PK_RK45_QUEUE
      if [ $v = queue ]; then export PK_RK45_QUEUE=1; fi
      if [ $v = queue2 ]; then export PK_RK45_QUEUE=1 PARCELS_HIP_LIB=$PWD/parcels_amd/libparcels_hip_q2.so; fi
      timeout 300 python tools/bench_configs.py --config c5 --reps 3 --pairs-leg 0 --only rk45 --check 1e5 > $out/c5_${v}_$r.json 2> $out/c5_${v}_$r.err
      python - $out/c5_${v}_$r.json $v $r <<'PY' | tee -a $out/summary.txt
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"{sys.argv[2]:8s} rep {sys.argv[3]} {d['kernels']:16s} {d['kernel_ms']:8.3f} ms  cold {d['kernel_ms_stats']['cold']:.3f} steps {d['particle_steps']} attempts {d['attempts']} check {'ok' if d.get('check') else None}", flush=True)
PY
    done
  done
  unset PARCELS_HIP_LIB PK_RK45_QUEUE
fi
bash tools/ab_c2_variants.sh $out/ab_c2 "new b4" 2 | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=50000 PARCELS_FUZZ_SEEDS=2000 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
