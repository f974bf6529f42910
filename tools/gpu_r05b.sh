#!/bin/bash
# round 5, second GPU call: the new GPU tests first (fail fast), the whole GPU suite, the default bench line (repeat_execute, sclk, fp64-peak roofline), the TWE pass timing
out=gpurun_out/r05b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_resident_columns.py tests/test_gpu_twe_passes.py tests/test_gpu_fast_cgrid.py -m gpu -q -x > $out/pytest_new.log 2>&1; echo "pytest new rc $?" | tee -a $out/summary.txt; tail -25 $out/pytest_new.log | tee -a $out/summary.txt
timeout 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/pytest.log | tee -a $out/summary.txt; grep -E "^FAILED|^ERROR" $out/pytest.log | head -20 | tee -a $out/summary.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d['roofline']
print('value',d['value'],'ms_per_step',d['ms_per_step'],'kernel',d['timed_reps']['kernel_ms'],'long_run',(d.get('long_run') or {}).get('value'))
print('roofline frac',r['frac'],'achieved',r['achieved'],r['unit'],'sclk',r.get('sclk_mhz'),'valu_busy',r.get('valu_busy_frac'))
print('repeat_execute',json.dumps(d.get('repeat_execute'))[:1500])
for s in d.get('secondary',[]): print(s.get('kernels'),s.get('kernel_ms'),(s.get('roofline') or {}).get('frac'),(s.get('check') or {}).get('passed'), json.dumps(s.get('velocity_pairs'))[:400])
print(d.get('legs_wall_s'))
PY
timeout 600 python tools/bench_twe_passes.py 1e6 > $out/twe_passes.json 2> $out/twe_passes.err; echo "twe rc $?" | tee -a $out/summary.txt; cat $out/twe_passes.json | tee -a $out/summary.txt
