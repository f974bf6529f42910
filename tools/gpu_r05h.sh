#!/bin/bash
# round 5: the clock probe after its rewrite (same-wavefront spin of 20 us behind the kernel): three bench runs, seven timed launches each -- do the readings agree?
out=gpurun_out/r05h; mkdir -p $out
for r in 1 2 3; do
  timeout 300 python bench.py --secondary 0 --no-cpu-baseline --user-kernels 0 --long-run 0 --repeat-execute 0 > $out/bench_$r.json 2> $out/bench_$r.err
  python - $out/bench_$r.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d['roofline']
print('kernel ms',d['timed_reps']['kernel_ms'],'sclk median',r.get('sclk_mhz'),'reps',[round(v,1) for v in (r.get('sclk_mhz_reps') or [])])
PY
done
timeout 300 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_resident_columns.py -m gpu -q 2>&1 | tail -2 | tee -a $out/summary.txt
