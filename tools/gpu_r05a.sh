#!/bin/bash
# round 5, first GPU call: full GPU suite, C2 block-cache A/B (new vs base), C5 with the timed pair-copy leg, the default bench line
out=gpurun_out/r05a; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/summary.txt; tail -3 $out/pytest.log | tee -a $out/summary.txt
bash tools/ab_c2_variants.sh $out/ab_c2 "new base" 2 | tee -a $out/summary.txt
PK_NO_BLOCK_CACHE=1 bash tools/ab_c2_variants.sh $out/ab_c2_env "new" 1 | sed 's/^new/new_noblk/' | tee -a $out/summary.txt
timeout 600 python tools/bench_configs.py --config c5 --reps 3 --check 1e5 > $out/c5.json 2> $out/c5.err; echo "c5 rc $?" | tee -a $out/summary.txt
python - $out/c5.json <<'PY' | tee -a $out/summary.txt
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d['kernels'], 'kernel_ms', d['kernel_ms'], d['kernel_ms_stats'], 'pairs', d.get('velocity_pairs'), 'check', (d.get('check') or {}).get('max_abs_diff'))
PY
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print('value',d['value'],'ms_per_step',d['ms_per_step'],'kernel',d['timed_reps']['kernel_ms'],'long_run',(d.get('long_run') or {}).get('value'))
for s in d.get('secondary',[]): print(s.get('kernels'),s.get('kernel_ms'),(s.get('roofline') or {}).get('frac'),(s.get('check') or {}).get('passed'))
PY
