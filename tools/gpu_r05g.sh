#!/bin/bash
# round 5, final validation call: whole GPU suite (new fixtures included), smoke(), the driver's bench command, a 2-rank rehearsal of bench.py --gpus 2 on the one GPU (gloo),
# the C3 5-day streaming run (regression of the slab ring with this round's engine changes)
out=gpurun_out/r05g; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; grep -E "passed|failed" $out/pytest.log | tee -a $out/summary.txt; grep -E "^FAILED|^ERROR" $out/pytest.log | head -20 | tee -a $out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" | tee -a $out/summary.txt; tail -2 $out/smoke.log | tee -a $out/summary.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r=d['roofline']
print('value',d['value'],'ms_per_step',d['ms_per_step'],'kernel',d['timed_reps']['kernel_ms'],'long_run',(d.get('long_run') or {}).get('value'))
print('roofline frac',r['frac'],'achieved',r['achieved'],r['unit'],'sclk',r.get('sclk_mhz'),'valu_busy',r.get('valu_busy_frac'),'stale',r.get('counters_stale'))
re_=d.get('repeat_execute') or {}
print('repeat_execute e2e later',re_.get('value_end_to_end_later_calls'),'wall',re_.get('wall_ms_later_calls'),'err',re_.get('error'))
for s in d.get('secondary',[]): print(s.get('kernels'),s.get('kernel_ms'),(s.get('roofline') or {}).get('frac'),'stale',(s.get('roofline') or {}).get('counters_stale'),(s.get('check') or {}).get('passed'), (s.get('velocity_pairs') or {}).get('frac_incl_pack'))
print(d.get('legs_wall_s'))
PY
timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --secondary 0 --no-cpu-baseline --long-run 0 --particles 2e6 > $out/bench_2ranks_rehearsal.json 2> $out/bench_2ranks_rehearsal.err; echo "bench 2 ranks rc $?" | tee -a $out/summary.txt; tail -c 600 $out/bench_2ranks_rehearsal.json | tee -a $out/summary.txt
timeout 600 python tools/bench_configs.py --config c3 --steps 120 --nt 6 --nslots 3 --reps 0 > $out/c3_stream_5days.json 2> $out/c3_stream.err; echo "c3 stream rc $?" | tee -a $out/summary.txt
python - $out/c3_stream_5days.json <<'PY' | tee -a $out/summary.txt
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print('c3 stream', d['kernels'], 'wall', d['wall_s'], 'kernel_ms', d['kernel_ms'], 'launches', d['launches'], d['stream_host_s'])
PY
