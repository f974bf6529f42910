#!/bin/bash
# round 6: HEAD the way the driver runs it -- the GPU suite serially with -x, smoke(), the default bench line -- plus the --gpus 2 rehearsal (gloo, one GPU)
out=gpurun_out/${OUT:-r06u}; mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/ -x -q -m gpu > $out/pytest_all.log 2>&1; echo "pytest -x -q -m gpu rc $?" | tee -a $out/summary.txt; grep -E "^FAILED" $out/pytest_all.log | cut -c1-200 | tee -a $out/summary.txt; tail -1 $out/pytest_all.log | tee -a $out/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" | tee -a $out/summary.txt; tail -1 $out/smoke.log | tee -a $out/summary.txt
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_default.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "check", (d.get("check") or {}).get("passed"), "frac", d["roofline"]["frac"], "stale", d["roofline"].get("counters_stale"), "kernel", d["timed_reps"]["kernel_ms"]["median"])
print("with_output", (d.get("with_output") or {}).get("wall_s"))
for s in d.get("secondary") or []:
    print(s.get("kernels"), s.get("kernel_ms"), (s.get("roofline") or {}).get("frac"), (s.get("check") or {}).get("passed"), (s.get("roofline") or {}).get("counters_stale"), ((s.get("roofline") or {}).get("attainable") or {}).get("frac_of_peak"))
print(d.get("legs_wall_s"))
PY
PARCELS_AMD_BENCH_REHEARSAL=1 timeout 900 python bench.py --gpus 2 --particles 2e6 --steps 8 --warmup 2 > $out/bench_gpus2_rehearsal.json 2> $out/bench_gpus2_rehearsal.err; echo "bench --gpus 2 rehearsal rc $?" | tee -a $out/summary.txt; tail -c 600 $out/bench_gpus2_rehearsal.json | tee -a $out/summary.txt
echo finished | tee -a $out/summary.txt
