#!/bin/bash
# round 6: HEAD once more on a fresh box -- the whole GPU suite, smoke(), the default bench line
out=gpurun_out/r06k; mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_all.log 2>&1; echo "pytest all rc $?" | tee -a $out/summary.txt; tail -3 $out/pytest_all.log | tee -a $out/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" | tee -a $out/summary.txt; tail -1 $out/smoke.log | tee -a $out/summary.txt
timeout 1200 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" | tee -a $out/summary.txt
python - $out/bench_default.json <<'PY' | tee -a $out/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "check", (d.get("check") or {}).get("passed"), "frac", d["roofline"]["frac"], "stale", d["roofline"].get("counters_stale"), "kernel", d["timed_reps"]["kernel_ms"]["median"])
print("with_output", {k:v for k,v in (d.get("with_output") or {}).items() if k in ("wall_s","value_incl_output","output_hidden_frac","async_not_slower_than_inline")})
for s in d.get("secondary") or []:
    print(s.get("kernels"), s.get("kernel_ms"), (s.get("roofline") or {}).get("frac"), (s.get("check") or {}).get("passed"), (s.get("roofline") or {}).get("counters_stale"), ((s.get("roofline") or {}).get("attainable") or {}).get("frac_of_peak"))
print(d.get("legs_wall_s"))
PY
