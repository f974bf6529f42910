#!/bin/bash
# round 5, third GPU call: resident-column tests again, RK45 cache-warming prefetch A/B (new | pf1 | pf2 | w2 | pf1w2), the repeat_execute leg
out=gpurun_out/r05c; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_resident_columns.py -m gpu -q > $out/pytest_res.log 2>&1; echo "pytest resident rc $?" | tee -a $out/summary.txt; tail -4 $out/pytest_res.log | tee -a $out/summary.txt
bash tools/ab_c5_variants.sh $out/ab_rk45 "new pf1 pf2 w2 pf1w2" 2 "--reps 3 --pairs-leg 0 --only rk45 --check 1e5" c5 | tee -a $out/summary.txt
grep -l "Traceback\|AssertionError" $out/ab_rk45/*.err | tee -a $out/summary.txt
timeout 300 python bench.py --secondary 0 --no-cpu-baseline --user-kernels 0 > $out/bench_c2.json 2> $out/bench_c2.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_c2.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print('value',d['value'],'kernel',d['timed_reps']['kernel_ms'],'long_run',(d.get('long_run') or {}).get('value'), (d.get('long_run') or {}).get('kernel_ms'))
print('repeat_execute',json.dumps(d.get('repeat_execute'))[:2500])
PY
