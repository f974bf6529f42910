#!/bin/bash
# round 5, last call: the whole GPU suite on the final tree, smoke(), a 6000-seed GPU fuzz sweep (HIP vs oracle) of the final binary, the default bench line
out=gpurun_out/r05i; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/pytest.log | tee -a $out/summary.txt; grep -E "^FAILED|^ERROR" $out/pytest.log | head -20 | tee -a $out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" | tee -a $out/summary.txt; tail -1 $out/smoke.log | tee -a $out/summary.txt
PARCELS_FUZZ_SEED0=52000 PARCELS_FUZZ_SEEDS=6000 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_configuration > $out/fuzz.log 2>&1; echo "fuzz rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/fuzz.log | tee -a $out/summary.txt; grep -E "^FAILED" $out/fuzz.log | head | tee -a $out/summary.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d['roofline']
print('value',d['value'],'ms_per_step',d['ms_per_step'],'kernel',d['timed_reps']['kernel_ms'],'long_run',(d.get('long_run') or {}).get('value'))
print('roofline frac',r['frac'],'achieved',r['achieved'],r['unit'],'sclk',r.get('sclk_mhz'),[round(v) for v in r.get('sclk_mhz_reps') or []],'valu_busy',r.get('valu_busy_frac'),'stale',r.get('counters_stale'))
re_=d.get('repeat_execute') or {}
print('repeat_execute e2e later',re_.get('value_end_to_end_later_calls'),'wall',re_.get('wall_ms_later_calls'),'err',re_.get('error'))
for s in d.get('secondary',[]): print(s.get('kernels'),s.get('kernel_ms'),(s.get('roofline') or {}).get('frac'),'stale',(s.get('roofline') or {}).get('counters_stale'),(s.get('check') or {}).get('passed'), (s.get('velocity_pairs') or {}).get('frac_incl_pack'))
print(d.get('legs_wall_s'))
PY
