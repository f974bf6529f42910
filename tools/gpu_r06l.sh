#!/bin/bash
# round 6: the persistent refilling-wavefront AdvectionRK45 (option "rk45_refill" / PK_RK45_REFILL): parity, then config 5 by wavefronts per CU
# (needs tools/patches/rk45_refilling_wavefronts.patch applied and the library rebuilt)
out=gpurun_out/${OUT:-r06l}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fast_cgrid.py -q -k "rk45" > $out/pytest_rk45.log 2>&1; echo "pytest rk45 rc $?" | tee -a $out/summary.txt; tail -3 $out/pytest_rk45.log | tee -a $out/summary.txt
for r in 1 2; do
  for k in ${KS:-0 12 9 6 4 3 24}; do
    PK_RK45_REFILL=$k timeout 600 python tools/bench_configs.py --config c5 --reps 3 --only rk45 --pairs-leg 0 --check 20000 > $out/c5_refill${k}_$r.json 2> $out/c5_refill${k}_$r.err
    python - $out/c5_refill${k}_$r.json $k $r <<'PY' | tee -a $out/summary.txt
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f"refill {sys.argv[2]:>3s} rep {sys.argv[3]} {d['kernels']:22s} {d['kernel_ms']:8.3f} ms  steps {d['particle_steps']} attempts {d['attempts']} check {(d.get('check') or {}).get('passed')}", flush=True)
PY
  done
done
echo finished | tee -a $out/summary.txt
